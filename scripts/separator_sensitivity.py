#!/usr/bin/env python3
"""How much does the QP optimum depend on WHICH vertex the separator LP returns?

The reference's LP (separator_glpk.cpp:248-373) has a zero objective: GLPK returns whichever vertex its simplex reaches,
and that choice — the direction of the half-plane `n.q + d - 1 <= 0` (solver_gurobi_poly.cpp:485-489) — is the one place
where the path's result is solver-defined.  GLPK is not available here, so instead of one other vertex this measures
the whole admissible set: every LP vertex is a line through two points of one set that touches the other set, the oracle
enumerates them all (oracle/neptune_oracle.c::orc_set_vertex_policy), and each replan is solved again with

  random   a pseudo-random admissible vertex per LP (three seeds)
  mingap   the admissible vertex with the smallest gap (the opposite extreme of the product's rule)
  worst    per LP the admissible vertex that leaves the max-gap optimum's own control points the least room
  bland    the vertex a textbook two-phase simplex with Bland's rule reaches
  glpk     the point a primal simplex of the class glp_simplex runs by default stops at — a feasible BASIS, in general not a vertex:
           free variables may stay non-basic at zero when phase 1 ends (one tight row 23 %, two 65 %, three 12 % of random LPs) —
           (standard start basis, projected
           steepest-edge pricing, Harris ratio test: oracle policy 5 = the product's separator rule 1,
           nep_batch_set_separator_rule) — the closest stand-in for the reference's own lines this image allows

against the product's rule (largest gap).  A replan is PROVABLY separator-independent when the optimum of its QP without
any line row satisfies the rows of the worst admissible vertex of every LP: then every choice yields that same point.

  python scripts/separator_sensitivity.py [--replans 200] [--out profiles/r03_separator_sensitivity.txt]
"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402


def ctrl_of(coeff, K, T):
    from neptune_amd import scene
    ref = np.zeros((8, 4, 2))
    ref[:K, :, 0] = scene.pos_ctrl_pts(coeff[0, :K], T); ref[:K, :, 1] = scene.pos_ctrl_pts(coeff[1, :K], T)
    return np.ascontiguousarray(ref)


def set_policy(oracle, policy, seed=0, ref=None):
    import ctypes as C
    L = oracle.lib()
    L.orc_set_vertex_policy.argtypes = [C.c_int, C.c_ulonglong, C.c_void_p]
    L.orc_set_vertex_policy.restype = None
    L.orc_set_vertex_policy(policy, seed, ref.ctypes.data if ref is not None else None)


def policy_stats(oracle):
    import ctypes as C
    a, b = C.c_long(0), C.c_long(0)
    oracle.lib().orc_vertex_policy_stats(C.byref(a), C.byref(b))
    return a.value, b.value


def rows_hold(p, coeff, K, seg, nd, tol=1e-9):
    from neptune_amd import scene
    if not len(seg):
        return True
    cx = scene.pos_ctrl_pts(coeff[0, :K], p.T_span); cy = scene.pos_ctrl_pts(coeff[1, :K], p.T_span)
    val = nd[:, 0:1] * cx[seg] + nd[:, 1:2] * cy[seg] + nd[:, 2:3] - 1.0
    scale = np.hypot(nd[:, 0], nd[:, 1])[:, None]
    return bool((val / np.maximum(scale, 1e-300)).max() <= tol)     # metres


def study_replan(oracle, scene, sc, a, guess=None):
    """-> dict of per-variant (status, rel cost change, max position change) for agent a (0-based) of scene sc"""
    p = sc["par"]
    g = sc["guesses"][a] if guess is None else guess
    K = int(g["K"]); T = p.T_span
    set_policy(oracle, 0)
    r0 = oracle.replan(p, a + 1, sc["committed"], g, sc["statics"])
    out = {"status0": r0["status"], "n_lines": r0["n_lines"]}
    nb, nl = scene.active_rows(p, r0["coeff"], K, r0["line_seg"], r0["line_nd"])
    out["active_line_rows"] = nl
    # provable independence: the line-free optimum against the worst admissible vertex of every LP
    ci = np.array(g["coeff"])[:, :K, :]
    rf = oracle.optimize(p, a + 1, ci, [], [], lines=(np.zeros(0, dtype=np.int32), np.zeros((0, 3))))
    indep = False
    if rf["status"] == 0 and r0["status"] == 0:
        set_policy(oracle, 3, ref=ctrl_of(rf["coeff"], K, T))
        rw = oracle.replan(p, a + 1, sc["committed"], g, sc["statics"])
        indep = rw["n_lp_failed"] == r0["n_lp_failed"] and rows_hold(p, rf["coeff"], K, rw["line_seg"], rw["line_nd"])
    out["provably_independent"] = bool(indep)
    pos0 = oracle.sample(r0["coeff"], T, p.dc)[:, :3]
    variants = [("random1", 1, 11), ("random2", 1, 22), ("random3", 1, 33), ("mingap", 2, 0), ("worst", 3, 0), ("bland", 4, 0), ("glpk", 5, 0)]
    n_lp = n_vert = 0
    for name, pol, seed in variants:
        set_policy(oracle, pol, seed, ref=ctrl_of(r0["coeff"], K, T) if pol == 3 else None)
        r = oracle.replan(p, a + 1, sc["committed"], g, sc["statics"])
        if pol == 1 and seed == 11:
            n_lp, n_vert = policy_stats(oracle)
        rec = {"status": r["status"]}
        if pol == 5:     # how many of its lines ARE the largest-gap lines (same supporting half-plane of the obstacle side)
            same = tot = 0
            if len(r["line_nd"]) == len(r0["line_nd"]) and len(r0["line_nd"]):
                a_, b_ = r["line_nd"], r0["line_nd"]
                la = np.column_stack([a_[:, 0], a_[:, 1], a_[:, 2] - 1.0]) / np.hypot(a_[:, 0], a_[:, 1])[:, None]
                lb = np.column_stack([b_[:, 0], b_[:, 1], b_[:, 2] - 1.0]) / np.hypot(b_[:, 0], b_[:, 1])[:, None]
                same = int((np.abs(la - lb).max(axis=1) < 1e-9).sum()); tot = len(a_)
            rec["same_lines"] = same; rec["lines"] = tot
        if r["status"] != 2 and r0["status"] != 2:
            rec["dcost"] = (r["objective"] - r0["objective"]) / (1.0 + abs(r0["objective"]))
            rec["dpos"] = float(np.abs(oracle.sample(r["coeff"], T, p.dc)[:, :3] - pos0).max())
            rec["dcoef"] = float(np.abs(r["coeff"] - r0["coeff"]).max())
        out[name] = rec
    set_policy(oracle, 0)
    out["lps"] = n_lp; out["vertices"] = n_vert
    return out


def summarise(name, recs, lines):
    n = len(recs)
    indep = sum(r["provably_independent"] for r in recs)
    noact = sum(r["active_line_rows"] == 0 for r in recs)
    lps = sum(r["lps"] for r in recs); verts = sum(r["vertices"] for r in recs)
    lines.append("%s: %d replans, %.1f lines each; %.2f admissible vertices per LP" % (name, n, np.mean([r["n_lines"] for r in recs]), verts / max(lps, 1)))
    lines.append("  provably separator-independent (line-free optimum clears the worst vertex of every LP): %d (%.1f %%);  no active line row at the max-gap optimum: %d (%.1f %%)"
                 % (indep, 100.0 * indep / n, noact, 100.0 * noact / n))
    lines.append("  %-8s %8s %12s %12s %12s %12s %12s %12s" % ("variant", "status!=", "dcost p50", "dcost p99", "dcost max", "dpos p50 m", "dpos p99 m", "dpos max m"))
    worst = {"dcost": 0.0, "dpos": 0.0, "status_changes": 0}
    for v in ("random1", "random2", "random3", "mingap", "worst", "bland", "glpk"):
        dep = [r for r in recs if not r["provably_independent"]]
        st = sum(r[v]["status"] != r["status0"] for r in recs)
        dc = np.array([abs(r[v]["dcost"]) for r in dep if "dcost" in r[v]]); dp = np.array([r[v]["dpos"] for r in dep if "dpos" in r[v]])
        if len(dc) == 0:
            dc = np.zeros(1); dp = np.zeros(1)
        lines.append("  %-8s %8d %12.3e %12.3e %12.3e %12.3e %12.3e %12.3e" % (v, st, np.percentile(dc, 50), np.percentile(dc, 99), dc.max(), np.percentile(dp, 50), np.percentile(dp, 99), dp.max()))
        worst["dcost"] = max(worst["dcost"], float(dc.max())); worst["dpos"] = max(worst["dpos"], float(dp.max())); worst["status_changes"] += st
        # the independent ones must not move at all (beyond solver noise): a check of the proof
        di = [r[v].get("dpos", 0.0) for r in recs if r["provably_independent"]]
        if di:
            key = "indep_dpos_glpk" if v == "glpk" else "indep_dpos"
            worst.setdefault(key, 0.0); worst[key] = max(worst[key], float(max(di)))
    sl = sum(r["glpk"].get("same_lines", 0) for r in recs); tl = sum(r["glpk"].get("lines", 0) for r in recs)
    unmoved = sum(1 for r in recs if r["glpk"].get("dpos", 1.0) < 1e-6)
    lines.append("  glpk: %d of %d lines (%.1f %%) are the largest-gap line; %d of %d replans (%.1f %%) end within 1e-6 m of the max-gap trajectory"
                 % (sl, tl, 100.0 * sl / max(tl, 1), unmoved, n, 100.0 * unmoved / n))
    worst["glpk_same_line_frac"] = sl / max(tl, 1); worst["glpk_unmoved_frac"] = unmoved / n
    lines.append("  (the %d provably independent replans moved by at most %.1e m under any VERTEX variant — random, mingap, worst, bland: the proof's check — and by "
                 "at most %.1e m under the glpk rule, whose answer is a feasible basis but in general NOT a vertex of the LP: free variables may stay "
                 "non-basic at zero when phase 1 ends, so the proof does not cover it; statistics above are over the other %d)"
                 % (indep, worst.get("indep_dpos", 0.0), worst.get("indep_dpos_glpk", 0.0), n - indep))
    return worst


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--replans", type=int, default=200)
    ap.add_argument("--out", default=os.path.join(ROOT, "profiles", "r03_separator_sensitivity.txt"))
    ap.add_argument("--json", default=os.path.join(ROOT, "profiles", "r03_separator_sensitivity.json"))
    args = ap.parse_args()
    from neptune_amd import scene
    from oracle import oracle
    lines = ["separator-vertex sensitivity (scripts/separator_sensitivity.py): every replan solved with other ADMISSIBLE vertices of",
             "each separator LP instead of the largest-gap one; dcost = |cost - cost_maxgap| / (1 + |cost_maxgap|), dpos = max |p(t) - p_maxgap(t)|", ""]
    summary = {}
    for name, N, S in (("config 2 (5 agents)", 5, 0), ("config 3 (8 agents + 20 obstacles)", 8, 20), ("config 4 (64 agents + 20 obstacles)", 64, 20)):
        recs = []
        seed = 100
        while len(recs) < args.replans:
            sc = scene.make_scene(N, S, seed=seed); seed += 1
            for a in range(N):
                recs.append(study_replan(oracle, scene, sc, a))
                if len(recs) >= args.replans:
                    break
        summary[name] = summarise(name, recs, lines)
        lines.append("")
        print("\n".join(lines[-12:]), flush=True)
    # front-end guesses (lattice paths that cut corners): config 3 and 4 scenes
    recs = []
    seed = 300
    while len(recs) < args.replans:
        N, S = (8, 20) if seed % 2 == 0 else (64, 20)
        sc = scene.make_scene(N, S, seed=seed); seed += 1
        p = sc["par"]; fe = scene.frontend_cfg(p, beam_width=16)
        for a in range(0, N, 1 if N == 8 else 4):
            st = scene.frontend_starts(sc)[a]
            hx, hn = oracle.hulls_of_scene(p, a + 1, sc["committed"], float(st["t_start"]), sc["statics"])
            g, res = oracle.frontend_beam(p, fe, a + 1, st, hx, hn, sc["statics"])
            if int(g["K"]) < 1:
                continue
            recs.append(study_replan(oracle, scene, sc, a, guess=g))
            if len(recs) >= args.replans:
                break
    summary["front-end guesses"] = summarise("front-end guesses (configs 3 and 4 scenes)", recs, lines)
    print("\n".join(lines[-11:]), flush=True)
    open(args.out, "w").write("\n".join(lines) + "\n")
    json.dump(summary, open(args.json, "w"), indent=1)


if __name__ == "__main__":
    main()
