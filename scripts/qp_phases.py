"""Per-phase shader cycles of the QP kernel for one slot (NEP_QP_PROFILE=1)."""
import os, sys
os.environ["NEP_QP_PROFILE"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np, torch
from neptune_amd import scene, backend
N, S = int(sys.argv[1]) if len(sys.argv) > 1 else 64, int(sys.argv[2]) if len(sys.argv) > 2 else 20
sc = scene.make_scene(N, S, seed=0); p = sc["par"]
bb = backend.BatchBackend(p, sc["statics"])
if os.environ.get("NEP_CULL"): bb.set_line_cull(float(os.environ["NEP_CULL"]))
dc = bb.to_device(sc["committed"]); dg = bb.to_device(sc["guesses"])
for _ in range(3): bb.replan(dc, dg)
names = ["A rows(update+resid)", "B combine+qc", "C rd+M assembly", "D conv test", "E cholesky", "F pred solve+Ua", "G P2 affine", "H P4 corr rhs", "I corr solve+Ud", "J P5 step+z"]
for slot in (0, 1, 2, 3):
    c = bb.debug_phase_cycles(slot); it = max(c[12], 1)
    tot = sum(c[:10])
    print("   loop cycles %d, workgroup lifetime %d (outside the loops %d: line gather %d, mode staging+decode %d, start point %d, rest (verify, theta, states) %d)" % (c[10], c[11], c[11] - c[10], c[13], c[14], c[15], c[11] - c[10] - c[13] - c[14] - c[15]))
    print("slot %d: iters %d, total cycles %d (%.1f us at 2.4GHz... s_memtime is 100MHz? raw)" % (slot, it, tot, tot / 2400.0))
    for k, nme in enumerate(names): print("   %-22s %10d  per-iter %8d  %5.1f%%" % (nme, c[k], c[k] // it, 100.0 * c[k] / max(tot, 1)))
