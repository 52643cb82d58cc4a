"""Convergence history of one slot of the QP kernel (build with make PROFILE=1 EXTRA=-DNEP_QP_ITERDBG=<agent index>).
usage: qp_iterdbg.py <scene seed> <agent index>   (front-end guesses, one scene)"""
import os, sys
os.environ["NEP_QP_PROFILE"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import ctypes as C
import numpy as np, torch
from neptune_amd import abi, dist as ndist, scene
from neptune_amd.backend import BatchBackend
from neptune_amd._lib import lib
N, M = int(os.environ.get("NEP_AGENTS", 64)), int(os.environ.get("NEP_STATICS", 20))
W = int(os.environ.get("NEP_BEAM", 32))
seed, a = int(sys.argv[1]), int(sys.argv[2])
sc = scene.make_scene(N, M, seed=seed); p = sc["par"]
statics = scene.make_scene(N, M, seed=0)["statics"] if N == 64 else sc["statics"]
be = BatchBackend(p, statics, n_scenes=1)
if os.environ.get("NEP_CULL"):
    be.set_line_cull(float(os.environ["NEP_CULL"]))
d_com = be.to_device(sc["committed"]); d_gue = be.to_device(sc["guesses"])
if not os.environ.get("NEP_NO_FRONTEND"):
    d_start = be.to_device(scene.frontend_starts(sc))
    be.frontend(scene.frontend_cfg(p, beam_width=W), d_com, d_start, d_gue, None)
    be.replan(None, d_gue)
else:
    be.replan(d_com, d_gue)
sol = be.solutions()
print("slot", a, "status", sol["stats"]["status"][a], "iters", sol["stats"]["iters"][a], "obj", sol["stats"]["objective"][a])
hist = np.zeros(16 * max(N, 64), dtype=np.int64)
for s_ in range(N):
    hist[16 * s_:16 * s_ + 16] = be.debug_phase_cycles(s_)
h = hist[16:16 + 60 * 8].reshape(60, 8)
for it in range(60):
    if not h[it].any():
        continue
    v = h[it].view(np.float64)
    print("it %2d nrp %.3e nrd %.3e (qs %.3e) gap %.3e obj %.9g flag %d alpha_prev %.3e sigmamu_prev %.3e" % (it, v[0], v[1], v[2], v[3], v[4], h[it][5], v[6], v[7]))
