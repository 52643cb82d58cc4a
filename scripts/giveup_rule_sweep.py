"""Development aid (CPU, the oracle): what the interior point's give-up rule costs and risks on the closed loop's hard replans
(tests/golden/moving_hard_cases.npz, HiGHS labels) and on ordinary front-end guesses.  The rule: from iteration FROM on, an affine step
shorter than 0.1 discards the predictor; more than MAX such iterations end the attempt (oracle knobs ORC_EXP_CORR_IT / ORC_EXP_CORR_MAX,
the kernels' kCorrFromIt / kCorrMaxCount).  Prints, per setting: device passes (iterations + discards) spent on the infeasible cases and on
the feasible ones, and how many feasible cases are lost.   python scripts/giveup_rule_sweep.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import helpers
from neptune_amd import abi, scene
from oracle import oracle


def main():
    oracle.build()
    p, cases = helpers.load_moving_hard_cases()
    # ordinary replans: front-end guesses of two 64-agent scenes (the oracle's own front end)
    for frm, mx in [tuple(int(x) for x in s.split(",")) for s in os.environ.get("SWEEP", "10,8 10,5 8,5 8,3 6,5 6,3 5,3").split()]:
        os.environ["ORC_EXP_CORR_IT"] = str(frm); os.environ["ORC_EXP_CORR_MAX"] = str(mx)
        oracle.pass_stats()
        tot = {0: [0, 0, 0], 1: [0, 0, 0], 2: [0, 0, 0], -1: [0, 0, 0]}      # label -> [cases, passes, status as labelled]
        lost = []
        for k, c in enumerate(cases):
            r = oracle.optimize(p, 1, c["coeff"], [], [], lines=(c["seg"], c["nd"]))
            it, tr = oracle.pass_stats()
            e = c["expected"]
            t = tot[e]; t[0] += 1; t[1] += it + tr
            if e >= 0 and r["status"] == e: t[2] += 1
            if e == 0 and r["status"] != 0: lost.append(k)
        print("from %2d max %d | infeasible: %d cases %5d passes (%d as labelled) | relaxed-only: %d cases %5d passes (%d) | feasible: %d cases %5d passes, lost %d %r | undecided: %d cases %5d passes"
              % (frm, mx, tot[2][0], tot[2][1], tot[2][2], tot[1][0], tot[1][1], tot[1][2], tot[0][0], tot[0][1], len(lost), lost, tot[-1][0], tot[-1][1]), flush=True)




def ordinary(n_scenes=4):
    """front-end guesses of ordinary scenes (the oracle's own beam search): statuses, passes and trajectories under two settings"""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    oracle.build()
    runs = {}
    work = []
    for seed in range(300, 300 + n_scenes):
        sc = scene.make_scene(64, 20, seed=seed); p = sc["par"]
        fe = scene.frontend_cfg(p, beam_width=32); starts = scene.frontend_starts(sc)
        for a in range(64):
            st = starts[a]
            hx, hn = oracle.hulls_of_scene(p, a + 1, sc["committed"], float(st["t_start"]), sc["statics"])
            g, r = oracle.frontend_beam(p, fe, a + 1, st, hx, hn, sc["statics"])
            if int(g["K"]) >= 1:
                work.append((sc, p, a, g))
    for frm, mx in ((10, 8), (6, 5), (6, 3)):
        os.environ["ORC_EXP_CORR_IT"] = str(frm); os.environ["ORC_EXP_CORR_MAX"] = str(mx)
        oracle.pass_stats()
        out = []
        for sc, p, a, g in work:
            res = oracle.replan(p, a + 1, sc["committed"], g, sc["statics"])
            it, tr = oracle.pass_stats()
            out.append((res["status"], it, tr, np.array(res["coeff"]).copy()))
        runs[(frm, mx)] = out
        print("ordinary, from %2d max %d: %d replans, statuses %r, passes %d (discards %d)" % (frm, mx, len(out), np.bincount([o[0] for o in out], minlength=3).tolist(),
                                                                                             sum(o[1] + o[2] for o in out), sum(o[2] for o in out)), flush=True)
    base = runs[(10, 8)]
    for key in ((6, 5), (6, 3)):
        oth = runs[key]
        ch = [(k, b[0], o[0]) for k, (b, o) in enumerate(zip(base, oth)) if b[0] != o[0]]
        dmax = max((float(np.abs(b[3] - o[3]).max()) for b, o in zip(base, oth) if b[0] == o[0] and b[0] != 2 and b[3].shape == o[3].shape), default=0.0)
        print("   %r against (10, 8): status changes %r; max coefficient difference among equal statuses %.3e" % (key, ch, dmax))


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "ordinary":
        ordinary(int(sys.argv[2]) if len(sys.argv) > 2 else 4)
    else:
        main()
