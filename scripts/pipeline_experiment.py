"""Development aid: the headline's step with the scenes in flight as G groups (one handle each) whose rounds are PIPELINED on streams —
the geometry halves of all groups (hulls + separating lines: chip-filling, VALU-bound) in turn on one stream, each group's QP half
(a few long solves on a mostly idle chip at its end) on the group's own stream, ordered by events only:

    geometry stream : lines(g0) lines(g1) ... lines(g0, next round: waits solve(g0)) ...
    stream of g     : solve(g) after lines(g)

Every group's rounds stay a chain (lines k -> solve k -> gather -> lines k + 1); nothing joins the groups but the final synchronise.
Prints ms per step (= every group one round) for G = 1 (one sequence, one captured graph) and each pipelined G, host-launched and with
one captured graph per half.   python scripts/pipeline_experiment.py [S] [steps]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np
import torch
from neptune_amd import scene, dist as ndist
from neptune_amd.backend import BatchBackend


def main():
    N, M = 64, 20
    S = int(sys.argv[1]) if len(sys.argv) > 1 else 128
    STEPS = int(sys.argv[2]) if len(sys.argv) > 2 else 200
    GS = [int(x) for x in sys.argv[3].split(",")] if len(sys.argv) > 3 else [2, 3, 4]
    scs = scene.make_scenes(N, M, range(S), workers=min(S, 64))
    p = scs[0]["par"]
    com, gue = ndist.stack_scenes(scs)
    dev = torch.device("cuda", 0)


    def handles(bounds):
        bes = []
        for lo, hi in bounds:
            b = BatchBackend(p, scs[lo]["statics"], n_scenes=hi - lo)
            for s in range(lo, hi):
                b.set_scene_statics(s - lo, scs[s]["statics"])
            bes.append(b)
        d_com = [bes[k].to_device(np.ascontiguousarray(com[lo:hi])) for k, (lo, hi) in enumerate(bounds)]
        d_gue = [bes[k].to_device(np.ascontiguousarray(gue[lo:hi])) for k, (lo, hi) in enumerate(bounds)]
        return bes, d_com, d_gue


    def timed(fn, steps):
        for _ in range(10):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / steps * 1e3


    # ---- one sequence, one graph (the round-6 headline's shape) ----
    bes, d_com, d_gue = handles([(0, S)])
    def step1():
        bes[0].replan(d_com[0], d_gue[0]); d_com[0].copy_(bes[0].d_commit)
    for _ in range(5):
        step1()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        step1()
    ms = timed(g.replay, STEPS)
    ref_sol = bes[0].solutions().copy()
    print("one sequence, one graph: %.4f ms/step, %.2f M replans/s" % (ms, S * N / ms / 1e3), flush=True)
    bes[0].close()

    for G in GS:
        cut = [round(k * S / G) for k in range(G + 1)]
        bounds = [(cut[k], cut[k + 1]) for k in range(G)]
        bes, d_com, d_gue = handles(bounds)
        s_geo = torch.cuda.Stream(device=dev)
        prio = -1 if os.environ.get("PIPE_PRIO", "1") == "1" else 0
        s_qp = [torch.cuda.Stream(device=dev, priority=prio) for _ in range(G)]
        e_geo = [torch.cuda.Event() for _ in range(G)]
        e_qp = [torch.cuda.Event() for _ in range(G)]
        for k in range(G):
            e_qp[k].record(torch.cuda.current_stream(dev))

        def geo(k):
            bes[k].replan_lines(d_com[k], d_gue[k])
        def qp(k):
            bes[k].replan_solve(d_com[k], d_gue[k]); d_com[k].copy_(bes[k].d_commit)

        graphs = None
        def step():
            for k in range(G):
                s_geo.wait_event(e_qp[k])
                with torch.cuda.stream(s_geo):
                    graphs[0][k].replay() if graphs else geo(k)
                    e_geo[k].record(s_geo)
                s_qp[k].wait_event(e_geo[k])
                with torch.cuda.stream(s_qp[k]):
                    graphs[1][k].replay() if graphs else qp(k)
                    e_qp[k].record(s_qp[k])
        ms_e = timed(step, STEPS)
        torch.cuda.synchronize()
        # per-half graphs
        gg, gq = [], []
        for k in range(G):
            a = torch.cuda.CUDAGraph()
            with torch.cuda.graph(a):
                geo(k)
            b = torch.cuda.CUDAGraph()
            with torch.cuda.graph(b):
                qp(k)
            gg.append(a); gq.append(b)
        torch.cuda.synchronize()
        graphs = (gg, gq)
        ms_g = timed(step, STEPS)
        sol = np.concatenate([b.solutions() for b in bes])
        same = bool((sol["stats"]["status"] == ref_sol["stats"]["status"]).all())
        dco = float(np.abs(np.array(sol["coeff"]) - np.array(ref_sol["coeff"])).max())
        print("pipelined, %d groups %s: host-launched %.4f ms/step (%.2f M/s); a graph per half %.4f ms/step (%.2f M replans/s); statuses equal to the one-sequence run: %s, coeff diff max %.3g"
              % (G, [hi - lo for lo, hi in bounds], ms_e, S * N / ms_e / 1e3, ms_g, S * N / ms_g / 1e3, same, dco), flush=True)
        for b in bes:
            b.close()


if __name__ == "__main__":
    main()
