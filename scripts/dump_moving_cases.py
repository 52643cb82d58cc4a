"""Dumps problem instances from bench.py's `moving` closed loop for offline study with the CPU oracle: after a number of rounds,
the scenes (records, statics), guesses and device results of the replans whose first solve failed plus a random sample of
the others -> gpurun_out/moving_cases.pkl (development aid), and — what tests/golden/make_moving_hard_cases.py turns into the
committed fixture — the HARD replans alone as (guess, the device's separating lines, device status): gpurun_out/moving_hard_raw.npz.
Usage: python scripts/dump_moving_cases.py [scenes] [rounds]"""
import os, sys, pickle
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np, torch
from neptune_amd import abi, dist as ndist, scene
from neptune_amd.backend import BatchBackend


def main():
    N, M, S = 64, 20, int(sys.argv[1]) if len(sys.argv) > 1 else 128
    rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 60
    scs = scene.make_scenes(N, M, range(S))
    p = scs[0]["par"]
    com, gue = ndist.stack_scenes(scs)
    be = BatchBackend(p, scs[0]["statics"], n_scenes=S)
    for s in range(S):
        be.set_scene_statics(s, scs[s]["statics"])
    cfg = scene.frontend_cfg(p, beam_width=32, pad_hold=1)
    starts = np.stack([scene.frontend_starts(s) for s in scs])
    d_st = be.to_device(starts); d_alt = torch.from_numpy(np.ascontiguousarray(starts["pos"].reshape(S * N, 3)).copy()).to(be.device)
    d_com = be.to_device(com); d_nxt = torch.empty_like(d_com); d_acc = torch.zeros(S * N, dtype=torch.int32, device=be.device)
    d_g = torch.zeros(S * N * abi.GUESS_DTYPE.itemsize, dtype=torch.uint8, device=be.device)
    d_res = torch.zeros(S * N * abi.FE_RESULT_DTYPE.itemsize, dtype=torch.uint8, device=be.device)
    out = []
    raw = dict(guess=[], seg=[], nd=[], status=[], iters_first=[], iters=[], where=[], hard=[])
    for r in range(rounds):
        com_before = d_com.cpu().numpy().view(abi.TRAJ_REC_DTYPE).reshape(S, N).copy() if r >= rounds - 3 else None
        be.frontend(cfg, d_com, d_st, d_g, d_res)
        be.replan(None, d_g)
        if com_before is not None:
            sol = be.solutions(timing=True).reshape(S, N); g = d_g.cpu().numpy().view(abi.GUESS_DTYPE).reshape(S, N).copy()
            st = sol["stats"]
            hard = (st["iters_first"] > 14) | (st["status"] > 0)
            hard &= sol["K"] > 0
            rng = np.random.default_rng(r)
            pick = set(map(tuple, np.argwhere(hard)))
            for s_, a_ in rng.integers(0, [S, N], size=(150, 2)):
                pick.add((int(s_), int(a_)))
            # the hard replans (and a few easy ones for control) as stand-alone QPs: the guess and the lines the separator made
            easy = [tuple(map(int, x)) for x in rng.integers(0, [S, N], size=(8, 2))]
            for s_, a_ in sorted(set(map(tuple, np.argwhere(hard)))) + easy:
                if int(sol[s_, a_]["K"]) == 0:
                    continue
                sg, nd_ = be.debug_lines(int(s_) * N + int(a_), cap=16384)
                raw["guess"].append(g[s_, a_].copy()); raw["seg"].append(sg); raw["nd"].append(nd_)
                raw["status"].append(int(st[s_, a_]["status"])); raw["iters_first"].append(int(st[s_, a_]["iters_first"])); raw["iters"].append(int(st[s_, a_]["iters"]))
                raw["where"].append((r, int(s_), int(a_))); raw["hard"].append(bool(hard[s_, a_]))
            scenes_needed = sorted({s_ for s_, _ in pick})
            out.append(dict(round=r, picks=sorted(pick), committed={s_: com_before[s_] for s_ in scenes_needed}, guesses={(s_, a_): g[s_, a_] for s_, a_ in pick},
                            sol={(s_, a_): sol[s_, a_] for s_, a_ in pick}))
        be.safety_commit(d_com, be.d_commit, d_g, d_nxt, d_acc)
        d_com.copy_(d_nxt)
        be.next_starts(d_com, p.T_span, d_st, d_alt, 0.5)
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    pickle.dump(dict(par=p, statics=[s["statics"] for s in scs], cases=out), open(os.path.join(ROOT, "gpurun_out", "moving_cases.pkl"), "wb"))
    n = len(raw["status"])
    off = np.zeros(n + 1, dtype=np.int64)
    for k in range(n):
        off[k + 1] = off[k] + len(raw["seg"][k])
    np.savez_compressed(os.path.join(ROOT, "gpurun_out", "moving_hard_raw.npz"), guess=np.array(raw["guess"]), line_off=off,
                        line_seg=np.concatenate(raw["seg"]).astype(np.int8), line_nd=np.concatenate(raw["nd"]), status=np.array(raw["status"], dtype=np.int8),
                        iters_first=np.array(raw["iters_first"], dtype=np.int16), iters=np.array(raw["iters"], dtype=np.int16),
                        where=np.array(raw["where"], dtype=np.int32), hard=np.array(raw["hard"]),
                        bounds=np.array([p.x_min, p.x_max, p.y_min, p.y_max, p.z_min, p.z_max, p.v_max, p.a_max, p.T_span, p.weight]))
    print("dumped", sum(len(o["picks"]) for o in out), "replans of", len(out), "rounds;", n, "stand-alone QPs (", int(np.sum(raw["hard"])), "hard ), lines", int(off[-1]))


if __name__ == "__main__":
    main()
