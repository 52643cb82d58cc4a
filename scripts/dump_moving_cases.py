"""Dumps problem instances from bench.py's `moving` closed loop for offline study with the CPU oracle: after a number of rounds,
the scenes (records, statics), guesses and device results of the replans whose first solve failed plus a random sample of
the others -> gpurun_out/moving_cases.npz (development aid).  Usage: python scripts/dump_moving_cases.py [scenes] [rounds]"""
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
            scenes_needed = sorted({s_ for s_, _ in pick})
            out.append(dict(round=r, picks=sorted(pick), committed={s_: com_before[s_] for s_ in scenes_needed}, guesses={(s_, a_): g[s_, a_] for s_, a_ in pick},
                            sol={(s_, a_): sol[s_, a_] for s_, a_ in pick}))
        be.safety_commit(d_com, be.d_commit, d_g, d_nxt, d_acc)
        d_com.copy_(d_nxt)
        be.next_starts(d_com, p.T_span, d_st, d_alt, 0.5)
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    pickle.dump(dict(par=p, statics=[s["statics"] for s in scs], cases=out), open(os.path.join(ROOT, "gpurun_out", "moving_cases.pkl"), "wb"))
    print("dumped", sum(len(o["picks"]) for o in out), "replans of", len(out), "rounds")


if __name__ == "__main__":
    main()
