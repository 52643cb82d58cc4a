"""Development aid: the verified presolve against the every-row solve on FRONT-END guesses (where interior-point solves end on the loose
snapshot): statuses and coefficient differences, and how many solves each path lists / certifies in the polish pass.
python scripts/presolve_vs_full_fe.py [scenes=8] [radius=4.0]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np
from neptune_amd import scene, abi, dist as ndist
from neptune_amd.backend import BatchBackend


def main():
    S = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    radius = float(sys.argv[2]) if len(sys.argv) > 2 else 4.0
    N = 64
    scs = [scene.make_scene(N, 20, seed=200 + s) for s in range(S)]
    p = scs[0]["par"]
    com, gue = ndist.stack_scenes(scs)
    bb = BatchBackend(p, scs[0]["statics"], n_scenes=S)
    for s in range(1, S):
        bb.set_scene_statics(s, scs[s]["statics"])
    d_com = bb.to_device(com); d_g = bb.to_device(gue)
    bb.frontend(scene.frontend_cfg(p, beam_width=32), d_com, bb.to_device(np.stack([scene.frontend_starts(s) for s in scs])), d_g, None)
    out = {}
    for name, cull, pol in (("full", 0.0, 1), ("full_nopolish", 0.0, 0), ("presolve", radius, 1), ("presolve_polished", radius, 2)):
        bb.set_line_cull(cull); bb.set_polish(pol)
        bb.replan(d_com, d_g)
        out[name] = bb.solutions().copy(); print(name, "polish listed/certified", bb.polish_count(), "redo", bb.redo_count(),
                                                 "status counts", np.bincount(out[name]["stats"]["status"].astype(int), minlength=3).tolist())
    for a, b in (("full", "presolve"), ("full", "presolve_polished"), ("full_nopolish", "presolve"), ("full", "full_nopolish")):
        A, B = out[a], out[b]
        same = A["stats"]["status"] == B["stats"]["status"]
        ok = same & (A["stats"]["status"] != 2)
        d = np.abs(A["coeff"] - B["coeff"]).reshape(len(A), -1).max(axis=1)
        print("%s vs %s: status mismatches %d; coeff diff over equal non-failed: max %.3e, > 1e-7: %d, > 1e-6: %d, > 1e-5: %d of %d"
              % (a, b, int((~same).sum()), d[ok].max(), int((d[ok] > 1e-7).sum()), int((d[ok] > 1e-6).sum()), int((d[ok] > 1e-5).sum()), int(ok.sum())))
    bb.close()


if __name__ == "__main__":
    main()
