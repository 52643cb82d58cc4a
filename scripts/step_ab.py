"""Development aid: per-kernel HIP-event times of the back end's launch sequence on the bench's scenes under a debug option (A/B).
python scripts/step_ab.py <scenes> <option=value>[,<option=value>...] [agents=64] [statics=20]    e.g.  step_ab.py 128 sep_pack=4
options: the names of nep_batch_debug_set_option (include/neptune_backend_debug.h), plus cull=<radius>, polish=<0|1|3>, hull=<0|1|2>"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np
from neptune_amd import scene, dist as ndist
from neptune_amd.backend import BatchBackend


def main():
    S = int(sys.argv[1]) if len(sys.argv) > 1 else 128
    sets = (sys.argv[2] if len(sys.argv) > 2 else "").split(";")          # ("" = the defaults: "a=1;;b=2" runs a=1, the defaults, b=2)
    N = int(sys.argv[3]) if len(sys.argv) > 3 else 64
    M = int(sys.argv[4]) if len(sys.argv) > 4 else 20
    scs = scene.make_scenes(N, M, range(S), workers=min(S, 64))
    com, gue = ndist.stack_scenes(scs)
    for opts in sets:
        be = BatchBackend(scs[0]["par"], scs[0]["statics"], n_scenes=S)
        for s in range(1, S):
            be.set_scene_statics(s, scs[s]["statics"])
        for kv in [o for o in opts.split(",") if o]:
            k, v = kv.split("=")
            if k == "cull": be.set_line_cull(float(v))
            elif k == "polish": be.set_polish(int(v))
            elif k == "hull": be.set_hull_kernel(int(v))
            else: be.debug_option(k, int(v))
        d_com, d_gue = be.to_device(com), be.to_device(gue)
        for _ in range(5):
            be.replan(d_com, d_gue)
        be.enable_timing(True); be.reset_timing()
        for _ in range(30):
            be.replan(d_com, d_gue)
        t = [be.kernel_time_ms(i)[0] for i in range(4)]
        be.enable_timing(False)
        sol = be.solutions()
        print("%-28s hull %.4f  separator %.4f  qp(+redo+polish) %.4f  sequence %.4f ms | iters mean %.3f, polish %r, redo %d, digest %016x"
              % (opts or "(defaults)", t[0], t[1], t[2], t[3], sol["stats"]["iters"].mean(), be.polish_count(), be.redo_count(),
                 int(np.frombuffer(np.ascontiguousarray(sol["coeff"]).tobytes(), dtype=np.uint64).sum() & 0xFFFFFFFFFFFFFFFF)), flush=True)
        be.close()


if __name__ == "__main__":
    main()
