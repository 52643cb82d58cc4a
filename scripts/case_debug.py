"""Per-agent GPU-vs-oracle differences of one scene with front-end guesses (development aid).
usage: case_debug.py <agents> <statics> <seed> <beam>"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np, torch
from neptune_amd import abi, scene
from neptune_amd.backend import BatchBackend
from oracle import oracle
n, m, seed, W = [int(x) for x in sys.argv[1:5]]
sc = scene.make_scene(n, m, seed=seed); p = sc["par"]; N = p.num_agents
bb = BatchBackend(p, sc["statics"])
d_com = bb.to_device(sc["committed"]); d_start = bb.to_device(scene.frontend_starts(sc))
d_guess = torch.zeros(N * abi.GUESS_DTYPE.itemsize, dtype=torch.uint8, device=bb.device)
bb.frontend(scene.frontend_cfg(p, beam_width=W), d_com, d_start, d_guess, None)
bb.replan(d_com, d_guess)
sol = bb.solutions(); g = d_guess.cpu().numpy().view(abi.GUESS_DTYPE)
for a in range(N):
    K = int(g[a]["K"])
    if K == 0:
        continue
    r = oracle.replan(p, a + 1, sc["committed"], g[a], sc["statics"])
    os.environ["ORC_EXP_GAPTOL"] = "1e-12"
    r2 = oracle.replan(p, a + 1, sc["committed"], g[a], sc["statics"])
    os.environ.pop("ORC_EXP_GAPTOL")
    co = np.array(sol[a]["coeff"])[:, :K, :]
    print("agent %2d status gpu %d oracle %d | iters gpu %2d oracle %2d tight %2d | gpu-oracle %.2e gpu-tight %.2e oracle-tight %.2e | obj gpu %.10g oracle %.10g" % (
        a, sol[a]["stats"]["status"], r["status"], sol[a]["stats"]["iters"], r["iters"], r2["iters"], np.abs(co - r["coeff"]).max(), np.abs(co - r2["coeff"]).max(),
        np.abs(r["coeff"] - r2["coeff"]).max(), sol[a]["stats"]["objective"], r["objective"]))
