"""Development aid: which config-5 searches carry children in big records (entangle states beyond the fixed record), and do
their outputs equal the oracle's.   python scripts/fe_ent_big.py [scenes=8] [check=4]"""
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
    be.frontend_ent(cfg, d_c, d_s, d_g, d_r, d_case); torch.cuda.synchronize()
    res = d_r.cpu().numpy().view(abi.FE_RESULT_DTYPE).reshape(S, N)
    big = res["_pad"].astype(np.int64) >> 8
    print("searches with big records: %d of %d; children in big records %d; ent_overflow %d" % ((big > 0).sum(), S * N, big.sum(), res["ent_overflow"].sum()))
    where = np.argwhere(big > 0)
    print("(scene, agent, big children):", [(int(s), int(a), int(big[s, a])) for s, a in where])
    if n_check <= 0:
        return
    import helpers
    from oracle import oracle
    got_g = d_g.cpu().numpy().view(abi.GUESS_DTYPE).reshape(S, N); got_case = d_case.cpu().numpy().reshape(S, N, abi.NEP_MAX_POL, N)
    for s, a in where[:n_check]:
        sc = dict(made[s], par=p); t0 = time.time()
        hx, hn = oracle.hulls_of_scene(p, a + 1, sc["committed"], float(starts[s, a]["t_start"]), sc["statics"])
        ent = helpers.ent_inputs(sc, a, t0=float(starts[s, a]["t_start"]))
        g, r, case = oracle.frontend_beam_ent(p, cfg, a + 1, starts[s, a], hx, hn, sc["statics"], ent)
        same = np.array_equal(np.array(got_g[s, a]["coeff"]), np.array(g["coeff"])) and np.array_equal(got_case[s, a], case) and all(int(res[s, a][f]) == r[f] for f in ("status", "K", "n_children", "n_feasible", "n_collision_free", "n_entangled", "ent_overflow"))
        print("scene %d agent %d: oracle %s (%.1f s); n_entangled %d / %d" % (s, a, "SAME" if same else "DIFFERENT", time.time() - t0, int(res[s, a]["n_entangled"]), r["n_entangled"]), flush=True)


if __name__ == "__main__":
    main()
