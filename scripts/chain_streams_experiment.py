"""Development aid: one step of the chain leg (front end -> lines -> QP -> safety check + commit) as G independent scene
groups on G HIP streams, captured into one graph, against the single-stream step.  The front end is latency-bound (15 % VALU
activity), the back end issue-bound: side by side they should overlap.  Prints ms per step for each G."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np
import torch
from neptune_amd import abi, scene, dist as ndist
from neptune_amd.backend import BatchBackend

if __name__ == "__main__":
    N, M, S = 64, 20, int(sys.argv[1]) if len(sys.argv) > 1 else 128
    scs = scene.make_scenes(N, M, range(S))
    p = scs[0]["par"]
    com, gue = ndist.stack_scenes(scs)
    starts = np.stack([scene.frontend_starts(s) for s in scs])
    MOVING = os.environ.get("NEP_MOVING") == "1"        # the closed loop (bench.py's `moving` leg) instead of the chain
    cfg = scene.frontend_cfg(p, beam_width=32, pad_hold=1 if MOVING else 0)
    dev = torch.device("cuda", 0)
    STAGGER = len(sys.argv) > 2 and sys.argv[2] == "stagger"
    for G in (1, 2, 4, 8):
        Sg = S // G
        bes = []
        for k in range(G):
            b = BatchBackend(p, scs[k * Sg]["statics"], n_scenes=Sg)
            for s in range(Sg):
                b.set_scene_statics(s, scs[k * Sg + s]["statics"])
            bes.append(b)
        sl = [slice(k * Sg, (k + 1) * Sg) for k in range(G)]
        d_com = [bes[k].to_device(np.ascontiguousarray(com[sl[k]])) for k in range(G)]
        d_st = [bes[k].to_device(np.ascontiguousarray(starts[sl[k]])) for k in range(G)]
        d_gfe = [torch.zeros(Sg * N * abi.GUESS_DTYPE.itemsize, dtype=torch.uint8, device=dev) for k in range(G)]
        d_res = [torch.zeros(Sg * N * abi.FE_RESULT_DTYPE.itemsize, dtype=torch.uint8, device=dev) for k in range(G)]
        d_nxt = [torch.empty_like(d_com[k]) for k in range(G)]
        d_acc = [torch.zeros(Sg * N, dtype=torch.int32, device=dev) for k in range(G)]
        streams = [torch.cuda.Stream(device=dev) for _ in range(G)]
        d_alt = [torch.from_numpy(np.ascontiguousarray(starts[sl[k]]["pos"].reshape(Sg * N, 3)).copy()).to(dev) for k in range(G)]

        R = int(os.environ.get("NEP_ROUNDS", "1"))       # rounds of every group inside one captured graph (the groups drift apart: one's QP tail beside another's front end)

        def step():
            cur = torch.cuda.current_stream(dev)
            fe_done = None
            for k in range(G):
                streams[k].wait_stream(cur)
                if STAGGER and fe_done is not None:
                    streams[k].wait_event(fe_done)          # the front ends one after the other: each back end runs next to the following group's front end
                with torch.cuda.stream(streams[k]):
                    for r_ in range(R):
                        bes[k].frontend(cfg, d_com[k], d_st[k], d_gfe[k], d_res[k])
                        if r_ == 0:
                            fe_done = torch.cuda.Event(); fe_done.record(streams[k])
                        bes[k].replan(None, d_gfe[k])
                        bes[k].safety_commit(d_com[k], bes[k].d_commit, d_gfe[k], d_nxt[k], d_acc[k])
                        d_com[k].copy_(d_nxt[k])
                        if MOVING:
                            bes[k].next_starts(d_com[k], p.T_span, d_st[k], d_alt[k], 0.5)
            for k in range(G):
                cur.wait_stream(streams[k])
        for _ in range(3):
            step()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            step()
        g.replay(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(40):
            g.replay()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        st = np.concatenate([b.solutions()["stats"]["status"] for b in bes]).astype(int)
        print("groups %d, %d rounds per graph%s: %.4f ms/round, %.0f replans/s (status %s)" % (G, R, " staggered" if STAGGER else "", dt / 40 / R * 1e3, S * N * 40 * R / dt, np.bincount(st, minlength=3).tolist()), flush=True)
        for b in bes:
            b.close()
