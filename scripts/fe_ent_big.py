"""Development aid: which config-5 searches carry children in big records (entangle states beyond the fixed record), and do
their outputs equal the oracle's, over a few closed-loop rounds.   python scripts/fe_ent_big.py [scenes=8] [check per round=4] [rounds=1]"""
import dataclasses, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from neptune_amd import scene, abi
from neptune_amd.backend import BatchBackend


def main():
    S = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    n_check = int(sys.argv[2]) if len(sys.argv) > 2 else 4
    N = 256
    made = scene.make_scenes(N, 100, range(S), workers=min(S, 32))
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
    cfg = scene.frontend_cfg(p, beam_width=32, entangle=True)
    import helpers
    from oracle import oracle
    rounds = int(sys.argv[3]) if len(sys.argv) > 3 else 1
    d_nx = torch.empty_like(d_c); d_ac = torch.zeros(S * N, dtype=torch.int32, device=be.device)
    n_same = n_diff = 0
    for rd in range(rounds):
        be.frontend_ent(cfg, d_c, d_s, d_g, d_r, d_case); torch.cuda.synchronize()
        res = d_r.cpu().numpy().view(abi.FE_RESULT_DTYPE).reshape(S, N)
        big = res["_pad"].astype(np.int64) >> 8
        print("round %d: searches re-run with big records: %d of %d (children stored in big records or taken through the global-memory pass: %d); ent_overflow %d" % (
            rd, (big > 0).sum(), S * N, big.sum(), res["ent_overflow"].sum()))
        where = np.argwhere(big > 0)
        print("   (scene, agent, such children):", [(int(s), int(a), int(big[s, a])) for s, a in where])
        got_g = d_g.cpu().numpy().view(abi.GUESS_DTYPE).reshape(S, N); got_case = d_case.cpu().numpy().reshape(S, N, abi.NEP_MAX_POL, N)
        recs = d_c.cpu().numpy().view(abi.TRAJ_REC_DTYPE).reshape(S, N)
        for s, a in where[:n_check]:
            sc = dict(made[s], par=p); t0 = time.time()
            hx, hn = oracle.hulls_of_scene(p, a + 1, recs[s], float(starts[s, a]["t_start"]), sc["statics"])
            ent = helpers.ent_inputs(sc, a, recs=recs[s], t0=float(starts[s, a]["t_start"]))
            g, r, case = oracle.frontend_beam_ent(p, cfg, a + 1, starts[s, a], hx, hn, sc["statics"], ent)
            same = np.array_equal(np.array(got_g[s, a]["coeff"]), np.array(g["coeff"])) and np.array_equal(got_case[s, a], case) and all(int(res[s, a][f]) == r[f] for f in ("status", "K", "n_children", "n_feasible", "n_collision_free", "n_entangled", "ent_overflow")) and float(res[s, a]["cost"]) == r["cost"]
            n_same += same; n_diff += not same
            print("   scene %d agent %d: guess, case block, counters and cost %s the oracle's (status %d, K %d, children pruned by the check %d; %.2f s)" % (s, a, "EQUAL" if same else "DIFFER FROM", int(res[s, a]["status"]), int(res[s, a]["K"]), r["n_entangled"], time.time() - t0), flush=True)
        be.replan(None, d_g, d_ent=d_case)
        be.safety_commit_ent(d_c, be.d_commit, d_g, d_nx, d_ac)
        d_c.copy_(d_nx)
    print("searches compared with the oracle: %d equal, %d different" % (n_same, n_diff))


if __name__ == "__main__":
    main()
