"""Why does the config-5 chain leg end 16 % of its front-end searches with "no solution" (round-3 review, item 3b)?  The same chain step
as bench.py's config5.chain leg is run for a few rounds on the GPU; then, on the hulls of the LAST round and from the same points A, the
oracle's restatement of the reference's best-first search (orc_frontend_astar: KinodynamicSearch::run without the entangle pruning,
20 000 pops) and the oracle's beam without the entangle check are run on the host for every search of the first scenes, and the
outcomes are tabulated next to the device's (beam WITH the entangle check).  A search the reference's own A* cannot start either
(no collision-free first primitive from point A) is a property of the scenario, not of the beam.

  python scripts/fe_success_rates.py [scenes=4] [rounds=6] [host_scenes=2]   ->  stdout (kept under profiles/)
"""
import dataclasses
import os
import sys
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np, torch          # noqa: E402
from neptune_amd import abi, scene  # noqa: E402
from neptune_amd.backend import BatchBackend  # noqa: E402
from oracle import oracle          # noqa: E402

NAMES = {0: "depth reached", 1: "goal reached", 2: "beam died out", 3: "no solution"}


def main():
    S = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 6
    SH = int(sys.argv[3]) if len(sys.argv) > 3 else 2
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
    d_nx = torch.empty_like(d_c); d_ac = torch.zeros(S * N, dtype=torch.int32, device=be.device)
    cfg = scene.frontend_cfg(p, beam_width=32, entangle=True)
    cfg_plain = scene.frontend_cfg(p, beam_width=32, entangle=False)
    for r in range(rounds):
        be.frontend_ent(cfg, d_c, d_s, d_g, d_r, d_case)
        res = d_r.cpu().numpy().view(abi.FE_RESULT_DTYPE).copy()
        if r == rounds - 1:
            hulls = [be.debug_hulls(s) for s in range(min(SH, S))]
            break
        be.replan(None, d_g, d_ent=d_case)
        be.safety_commit_ent(d_c, be.d_commit, d_g, d_nx, d_ac)
        d_c.copy_(d_nx)
        st = be.solutions()["stats"]["status"]
        print("round %d: front end %s | back end ok %d relaxed %d failed %d | accepted %.3f"
              % (r, {NAMES[k]: int((res["status"] == k).sum()) for k in NAMES}, (st == 0).sum(), (st == 1).sum(), (st == 2).sum(), float(d_ac.float().mean().item())), flush=True)
    print("device beam WITH the entangle check, last round, all %d searches: %s; children pruned by the check %d; ent_overflow %d"
          % (S * N, {NAMES[k]: int((res["status"] == k).sum()) for k in NAMES}, int(res["n_entangled"].sum()), int(res["ent_overflow"].sum())))
    # the same searches once more with the widest beam this build has (64): how many of the lost ones would a wider-beam retry find?
    cfg64 = scene.frontend_cfg(p, beam_width=abi.NEP_FE_MAX_BEAM if hasattr(abi, "NEP_FE_MAX_BEAM") else 64, entangle=True)
    d_g2 = torch.zeros_like(d_g); d_r2 = torch.zeros_like(d_r); d_case2 = torch.zeros_like(d_case)
    be.frontend_ent(cfg64, d_c, d_s, d_g2, d_r2, d_case2)
    res64 = d_r2.cpu().numpy().view(abi.FE_RESULT_DTYPE).copy()
    lost32 = (res["status"] >= 2)
    print("device beam of width 64 WITH the entangle check, same inputs: %s; of the %d searches the width-32 beam ended without a plan (died out / no solution), width 64 finds a plan for %d"
          % ({NAMES[k]: int((res64["status"] == k).sum()) for k in NAMES}, int(lost32.sum()), int((lost32 & (res64["status"] < 2)).sum())))
    oracle.lib()
    jobs = [(s, a) for s in range(min(SH, S)) for a in range(N)]

    def one(job):
        s, a = job
        hx, hn = hulls[s]
        g1, r1 = oracle.frontend_astar(p, cfg_plain, a + 1, starts[s, a], hx, hn, made[s]["statics"], max_pops=20000)
        g2, r2 = oracle.frontend_beam(p, cfg_plain, a + 1, starts[s, a], hx, hn, made[s]["statics"])
        return r1["status"], int(g1["K"]), r2["status"], int(g2["K"])
    with ThreadPoolExecutor(min(128, os.cpu_count() or 1)) as ex:
        out = np.array(list(ex.map(one, jobs)))
    dev = res["status"].reshape(S, N)[:min(SH, S)].reshape(-1); devK = res["K"].reshape(S, N)[:min(SH, S)].reshape(-1)
    n = len(jobs)
    print("\nsame hulls, same points A, first %d scenes (%d searches):" % (min(SH, S), n))
    print("  %-46s %s" % ("device beam, entangle check ON (what the leg runs)", {NAMES[k]: int((dev == k).sum()) for k in NAMES}))
    print("  %-46s %s" % ("oracle beam, entangle check off", {NAMES[k]: int((out[:, 2] == k).sum()) for k in NAMES}))
    print("  %-46s %s" % ("oracle A* (reference's search, 20 000 pops), off", {NAMES[k]: int((out[:, 0] == k).sum()) for k in NAMES}))
    both_none = int(((dev == 3) & (out[:, 0] == 3)).sum())
    print("  searches with no solution on the device: %d; of these the reference's A* has no solution either: %d (no collision-free first primitive from point A: "
          "the scenario's, not the beam's); found by the A* but not by the device beam: %d" % (int((dev == 3).sum()), both_none, int(((dev == 3) & (out[:, 0] != 3)).sum())))
    print("  plans the device returns (K > 0): %d; the A* returns: %d; A* plans shorter than num_pol: %d" % (int((devK > 0).sum()), int((out[:, 1] > 0).sum()), int(((out[:, 1] > 0) & (out[:, 1] < p.num_pol)).sum())))


if __name__ == "__main__":
    main()
