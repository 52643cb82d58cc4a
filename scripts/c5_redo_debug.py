"""Which config-5 replans does the presolve send to the redo pass, and why (development aid)?  With NEP_SEP_NO_REDO=1 the flagged
replans keep their presolved result: the parked lines and the movement are then evaluated here, on the host."""
import os, sys, dataclasses
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np, torch
from neptune_amd import abi, dist as ndist, scene
from neptune_amd.backend import BatchBackend
from bench import _config5_scene
import multiprocessing as mp
from concurrent.futures import ProcessPoolExecutor


def main():
    S = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 6
    with ProcessPoolExecutor(max_workers=S, mp_context=mp.get_context("spawn")) as ex:
        made = list(ex.map(_config5_scene, [(256, 100, s) for s in range(S)]))
    scs = [m[0] for m in made]; case = np.stack([m[1] for m in made])
    p = dataclasses.replace(scs[0]["par"], enable_entangle=True)
    be = BatchBackend(p, scs[0]["statics"], n_scenes=S)
    for s in range(S):
        be.set_scene_statics(s, scs[s]["statics"])
    com, gue = ndist.stack_scenes(scs)
    d_c = be.to_device(com); d_g = be.to_device(gue); d_e = torch.from_numpy(np.ascontiguousarray(case).reshape(-1)).to(be.device)
    T = p.T_span
    M4 = scene.A_POS_INV * np.array([T ** 3, T ** 2, T, 1.0])[:, None]
    cg = np.array(gue.reshape(-1)["coeff"])
    for rnd in range(rounds):
        be.replan(d_c, d_g, d_ent=d_e)
        n = be.redo_count(); lst = be.redo_list()
        sol = be.solutions(); st = sol["stats"]; co = np.array(sol["coeff"])
        print("round %d: redo count %d %s slots %s" % (rnd, n, be.redo_reasons, lst[:12].tolist()))
        for a in lst[:6]:
            a = int(a); K = int(sol[a]["K"])
            seg, nd = be.debug_lines(a, cap=30000)
            cpx = co[a, 0, :K] @ M4; cpy = co[a, 1, :K] @ M4; gx = cg[a, 0, :K] @ M4; gy = cg[a, 1, :K] @ M4
            mv = np.sqrt((cpx - gx) ** 2 + (cpy - gy) ** 2).max()
            val = nd[:, 0:1] * cpx[seg] + nd[:, 1:2] * cpy[seg] + nd[:, 2:3] - 1.0
            gdist = -(nd[:, 0:1] * gx[seg] + nd[:, 1:2] * gy[seg] + nd[:, 2:3] - 1.0) / np.hypot(nd[:, 0], nd[:, 1])[:, None]
            bad = np.nonzero(val.max(axis=1) > 0)[0]
            print("      host: final leading coefficients y", co[a, 1, :K, 0], " x", co[a, 0, :K, 0])
            print("      host: guess cp31 (%.6f, %.6f) solution cp31 (%.6f, %.6f); guess coeff y seg7 %s" % (gx[K - 1][3], gy[K - 1][3], cpx[K - 1][3], cpy[K - 1][3], cg[a, 1, K - 1]))
            print("   slot %d status %d iters %d rows %d lines(stat) %d lines(read) %d  moved %.3f m; violated lines: %d, their values %s, their distance from the guess %s" % (
                a, st[a]["status"], st[a]["iters"], st[a]["n_rows"], st[a]["n_lines"], len(seg), mv, len(bad), val.max(axis=1)[bad][:4], gdist.min(axis=1)[bad][:4]))
        cm = be.d_commit.view(-1, abi.TRAJ_REC_DTYPE.itemsize); v = d_c.view(-1, abi.TRAJ_REC_DTYPE.itemsize)
        v[:, 40:64].copy_(cm[:, 40:64]); v[:, 192:].copy_(cm[:, 192:])


if __name__ == "__main__":
    main()
