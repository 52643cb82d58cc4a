"""Prints GPU-vs-oracle detail for a few scenes (development aid for gpurun sessions)."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from neptune_amd import scene, abi, backend
from oracle import oracle

print(torch.cuda.get_device_name(0))
for (N, S, K, seed) in [(5, 0, 8, 0), (8, 20, 8, 3), (3, 4, 2, 22), (64, 20, 8, 0)]:
    sc = scene.make_scene(N, S, seed=seed, K=K)
    p = sc["par"]
    bb = backend.BatchBackend(p, sc["statics"])
    dc = bb.to_device(sc["committed"]); dg = bb.to_device(sc["guesses"])
    bb.enable_timing(True)
    for _ in range(3):
        bb.replan(dc, dg)
    torch.cuda.synchronize()
    sol = bb.solutions()
    print("scene N=%d S=%d K=%d: kernel ms hull %.3f sep %.3f qp %.3f total %.3f" % (N, S, K, bb.kernel_time_ms(0)[0], bb.kernel_time_ms(1)[0], bb.kernel_time_ms(2)[0], bb.kernel_time_ms(3)[0]))
    for a in range(min(N, 8)):
        r = oracle.replan(p, a + 1, sc["committed"], sc["guesses"][a], sc["statics"])
        st = sol[a]["stats"]; Kk = int(sol[a]["K"])
        co = np.array(sol[a]["coeff"])[:, :Kk, :]
        seg, nd = bb.debug_lines(a)
        leq = (len(nd) == len(r["line_nd"])) and np.array_equal(nd, r["line_nd"])
        print("  a%d gpu st %d it %d/%d lines %d lp %d/%d obj %.10g | orc st %d it %d lines %d obj %.10g | dcoef %.2e lines_eq %s" % (
            a + 1, st["status"], st["iters"], st["iters_first"], st["n_lines"], st["n_lp"], st["n_lp_failed"], st["objective"],
            r["status"], r["iters"], r["n_lines"], r["objective"], np.abs(co - r["coeff"]).max(), leq))
    bb.close()
