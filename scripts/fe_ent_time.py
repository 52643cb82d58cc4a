"""Development aid: wall time of frontend_kernel<true> (entangle check on) at config-5 size, first replan of S seeded scenes and
after a few closed-loop rounds (the state the bench's config5.chain leg times).   python scripts/fe_ent_time.py [scenes=32] [rounds=4]"""
import dataclasses, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np, torch
from neptune_amd import scene, abi
from neptune_amd.backend import BatchBackend


def main():
    S = int(sys.argv[1]) if len(sys.argv) > 1 else 32
    rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 4
    N = 256
    made = scene.make_scenes(N, 100, range(S), workers=min(S, 32))
    if os.environ.get("NEP_SCRIPT_BENDS"):      # the bench's config-5 inputs: 2-4 bend points per tether (scene.synthetic_entangle)
        for k_, m_ in enumerate(made): scene.synthetic_entangle(m_, seed=1000 + k_, frac=0.1)
    p = dataclasses.replace(made[0]["par"], enable_entangle=True)
    be = BatchBackend(p, made[0]["statics"], n_scenes=S)
    for s in range(S):
        be.set_scene_statics(s, made[s]["statics"])
        reps, long_ = scene.static_reps(made[s]["statics"]); be.set_static_reps(reps, long_, scene=s)
    com = np.stack([m["committed"] for m in made]); starts = np.stack([scene.frontend_starts(m) for m in made])
    d_c = be.to_device(com); d_s = be.to_device(starts)
    d_g = torch.zeros(S * N * abi.GUESS_DTYPE.itemsize, dtype=torch.uint8, device=be.device)
    d_r = torch.zeros(S * N * abi.FE_RESULT_DTYPE.itemsize, dtype=torch.uint8, device=be.device)
    d_case = torch.zeros(S * N * abi.NEP_MAX_POL * N, dtype=torch.int32, device=be.device)
    d_nx = torch.empty_like(d_c); d_ac = torch.zeros(S * N, dtype=torch.int32, device=be.device)
    cfg = scene.frontend_cfg(p, beam_width=32, entangle=True)
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    for r in range(rounds):
        e0.record(); be.frontend_ent(cfg, d_c, d_s, d_g, d_r, d_case); e1.record(); torch.cuda.synchronize()
        res = d_r.cpu().numpy().view(abi.FE_RESULT_DTYPE)
        print("round %d: frontend_ent of %d searches %.2f ms; entangled %d overflow %d; guesses checksum %d" % (
            r, S * N, e0.elapsed_time(e1), int(res["n_entangled"].sum()), int(res["ent_overflow"].sum()),
            int(d_g.view(torch.int64).sum().item() & 0xffffffffffff)), flush=True)
        us = be.fe_search_us(); big = res["_pad"].astype(np.int64) >> 8
        print("         search time us: mean %.0f p50 %.0f p99 %.0f max %.0f; searches with big records %d (children %d), their times %s" % (
            us.mean(), np.percentile(us, 50), np.percentile(us, 99), us.max(), int((big > 0).sum()), int(big.sum()), np.sort(us[big > 0]).astype(int).tolist()), flush=True)
        if r == 0: us_prev2 = us.copy(); us_max = us.copy()
        if r > 0:      # what the launch order is worth: list scheduling of this round's measured times on 768 workgroup slots
            import heapq
            def makespan(order):
                h = [0.0] * 768; heapq.heapify(h)
                end = 0.0
                for i in order:
                    t = heapq.heappop(h) + us[i]; end = max(end, t); heapq.heappush(h, t)
                return end
            n = len(us)
            for name, order in (("slot order", np.arange(n)), ("previous round's time in 256 us bins (what the kernel does)", np.argsort(-np.minimum(63, (us_prev / 256).astype(int)), kind="stable")),
                                ("previous round's time in 64 us bins", np.argsort(-np.minimum(63, (us_prev / 64).astype(int)), kind="stable")), ("previous round's time, exact", np.argsort(-us_prev, kind="stable")),
                                ("max of the two previous rounds' times", np.argsort(-np.maximum(us_prev, us_prev2), kind="stable")),
                                ("max of the previous rounds' times, all", np.argsort(-us_max, kind="stable")),
                                ("this round's time (perfect predictor)", np.argsort(-us, kind="stable"))):
                print("         simulated makespan on 768 slots, %s: %.2f ms (sum / 768 = %.2f ms)" % (name, makespan(order) / 1e3, us.sum() / 768e3))
        us_prev2 = us_prev.copy() if r > 0 else us.copy(); us_prev = us.copy(); us_max = np.maximum(us_max, us) if r > 0 else us.copy()
        be.replan(None, d_g, d_ent=d_case)
        e2 = torch.cuda.Event(enable_timing=True); e3 = torch.cuda.Event(enable_timing=True)
        e2.record(); be.safety_commit_ent(d_c, be.d_commit, d_g, d_nx, d_ac); e3.record(); torch.cuda.synchronize()
        print("         safety pass with the entangle re-check: %.3f ms; accepted %d" % (e2.elapsed_time(e3), int(d_ac.sum().item())), flush=True)
        d_c.copy_(d_nx)


if __name__ == "__main__":
    main()
