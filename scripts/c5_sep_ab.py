"""Where does the separator's time go at config-5 size (development aid)?  Times separator_kernel for the same scenes with and
without the entangle candidates and with and without static obstacles / bases in reach."""
import os, sys, dataclasses
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np, torch
from neptune_amd import abi, dist as ndist, scene
from neptune_amd.backend import BatchBackend
from bench import _config5_scene
import multiprocessing as mp
from concurrent.futures import ProcessPoolExecutor


def main():
    S = int(sys.argv[1]) if len(sys.argv) > 1 else 32
    with ProcessPoolExecutor(max_workers=S, mp_context=mp.get_context("spawn")) as ex:
        made = list(ex.map(_config5_scene, [(256, 100, s) for s in range(S)]))
    scs = [m[0] for m in made]; case = np.stack([m[1] for m in made])
    com, gue = ndist.stack_scenes(scs)
    for label, ent, cull in (("entangle on, presolve 4 m", True, 4.0), ("entangle off, presolve 4 m", False, 4.0), ("entangle on, no presolve", True, 0.0), ("entangle off, no presolve", False, 0.0),
                             ("entangle on, presolve 2 m", True, 2.0), ("entangle on, presolve 8 m", True, 8.0)):
        p = dataclasses.replace(scs[0]["par"], enable_entangle=ent)
        be = BatchBackend(p, scs[0]["statics"], n_scenes=S)
        for s in range(S):
            be.set_scene_statics(s, scs[s]["statics"])
        be.set_line_cull(cull)
        d_c = be.to_device(com); d_g = be.to_device(gue); d_e = torch.from_numpy(np.ascontiguousarray(case).reshape(-1)).to(be.device) if ent else None
        be.enable_timing(True)
        for _ in range(3):
            be.replan(d_c, d_g, d_ent=d_e)
        be.reset_timing()
        for _ in range(10):
            be.replan(d_c, d_g, d_ent=d_e)
        torch.cuda.synchronize()
        sol = be.solutions()
        read = np.mean([len(be.debug_lines(a, cap=30000)[0]) for a in range(0, S * 256, 97)])
        print("%-28s separator %.3f ms  qp %.3f ms  (%s)  lines made per replan %.1f of %.1f, rows solved %.0f" % (label, be.kernel_time_ms(1)[0], be.kernel_time_ms(2)[0], be.qp_kernel_name(), read, sol["stats"]["n_lines"].mean(), sol["stats"]["n_rows"].mean()))
        be.close()


if __name__ == "__main__":
    main()
