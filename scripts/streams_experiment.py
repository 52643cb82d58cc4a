"""Development aid: one step of the headline workload as G independent scene groups on G HIP streams (captured into one
graph), against the single-stream step.  Prints ms per step for each G."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np
import torch
from neptune_amd import scene, dist as ndist
from neptune_amd.backend import BatchBackend

N, M, S = 64, 20, 32
scs = [scene.make_scene(N, M, seed=s) for s in range(S)]
p = scs[0]["par"]
com, gue = ndist.stack_scenes(scs)
dev = torch.device("cuda", 0)
for G in (1, 2, 4):
    Sg = S // G
    bes = []
    for k in range(G):
        b = BatchBackend(p, scs[k * Sg]["statics"], n_scenes=Sg)
        for s in range(Sg):
            b.set_scene_statics(s, scs[k * Sg + s]["statics"])
        bes.append(b)
    d_com = [bes[k].to_device(np.ascontiguousarray(com[k * Sg:(k + 1) * Sg])) for k in range(G)]
    d_gue = [bes[k].to_device(np.ascontiguousarray(gue[k * Sg:(k + 1) * Sg])) for k in range(G)]
    streams = [torch.cuda.Stream(device=dev) for _ in range(G)]

    def step():
        cur = torch.cuda.current_stream(dev)
        for k in range(G):
            streams[k].wait_stream(cur)
            with torch.cuda.stream(streams[k]):
                bes[k].replan(d_com[k], d_gue[k])
                d_com[k].copy_(bes[k].d_commit)
        for k in range(G):
            cur.wait_stream(streams[k])
    for _ in range(5):
        step()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        step()
    g.replay(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(200):
        g.replay()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    it = np.concatenate([b.solutions()["stats"]["iters"] for b in bes]).mean()
    print("groups %d: %.4f ms/step, %.0f replans/s (iters %.2f)" % (G, dt / 200 * 1e3, S * N * 200 / dt, it), flush=True)
    for b in bes:
        b.close()
