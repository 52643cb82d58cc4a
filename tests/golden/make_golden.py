#!/usr/bin/env python3
"""Generates the golden fixtures that pin the CPU oracle (and through it the HIP path).

The reference holds no golden vectors for this path and its solvers (Gurobi, GLPK 4.65, CGAL)
are not available, so the fixtures are produced HERE, independently of both
the oracle and the product:

  minvo_kat.json   MINVO position/velocity control points of fixed cubics, computed from the
                   literal matrices of reference neptune/include/mader_types.hpp:152-163 with an
                   exact rational inverse (fractions.Fraction), no floating-point inversion.
  qp_cases.npz     For seeded scenes: the QP of reference solver_gurobi_poly.cpp:322-710 in its
                   own 12K-variable space (equalities written out), built with numpy from given
                   separating lines, solved by two independent SciPy solvers (trust-constr and
                   SLSQP); a case is kept only if the two agree to 1e-7 relative cost.
  lp_cases.npz     Random separator LPs (separator_glpk.cpp:248-373) with HiGHS feasibility.

Run from the repo root:  python tests/golden/make_golden.py
(uses the oracle only to obtain the *lines* for the scenes — lines are inputs of the QP cases,
so the QP golden values do not depend on any solver of ours).
"""
import json
import os
import sys
from fractions import Fraction as F

import numpy as np
from scipy import optimize as so

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
HERE = os.path.dirname(os.path.abspath(__file__))

A_POS = [["-3.4416308968564117698463178385282", "6.9895481477801393310755884158425", "-4.4622887507045296828778191411402", "0.91437149978080234369315348885721"],
         ["6.6792587327074839365081970754545", "-11.845989901556746914934592496138", "5.2523596690684613008670567069203", "0"],
         ["-6.6792587327074839365081970754545", "8.1917862965657040064115790301003", "-1.5981560640774179482548333908198", "0.085628500219197656306846511142794"],
         ["3.4416308968564117698463178385282", "-3.3353445427890959784633650997421", "0.80808514571348655231020075007109", "-0.0000000000000000084567769453869345852581318467855"]]
A_VEL = [["1.50000000000000", "-2.36602540378444", "0.933012701892219"], ["-3", "3", "0"],
         ["1.50000000000000", "-0.633974596215561", "0.0669872981077807"]]


def frac_inv(M):
    n = len(M)
    M = [[F(float(x)) for x in r] + [F(int(i == j)) for j in range(n)] for i, r in enumerate(M)]
    for c in range(n):
        p = max(range(c, n), key=lambda r: abs(M[r][c]))
        M[c], M[p] = M[p], M[c]
        pv = M[c][c]
        M[c] = [x / pv for x in M[c]]
        for r in range(n):
            if r != c:
                f = M[r][c]
                M[r] = [a - f * b for a, b in zip(M[r], M[c])]
    return [r[n:] for r in M]


APINV_F = frac_inv(A_POS)
AVINV_F = frac_inv(A_VEL)
APINV = np.array([[float(x) for x in r] for r in APINV_F])
AVINV = np.array([[float(x) for x in r] for r in AVINV_F])


def pos_inv_T(T):
    return APINV * np.array([T ** 3, T ** 2, T, 1.0])[:, None]


def vel_inv321_T(T):
    return AVINV * (np.array([3.0, 2.0, 1.0]) * np.array([T ** 2, T, 1.0]))[:, None]


def minvo_kat():
    rng = np.random.default_rng(1234)
    cases = [dict(T=0.5, P=[1.0, -2.0, 0.5, 3.0])]
    for _ in range(15):
        cases.append(dict(T=float(rng.choice([0.25, 0.5, 1.0, 0.4])), P=[float(x) for x in rng.normal(size=4) * 3]))
    out = []
    for c in cases:
        T = F(c["T"]); P = [F(x) for x in c["P"]]
        tp = [T ** 3, T ** 2, T, F(1)]
        q = [float(sum(P[j] * tp[j] * APINV_F[j][k] for j in range(4))) for k in range(4)]
        tv = [3 * T ** 2, 2 * T, F(1)]
        v = [float(sum(P[j] * tv[j] * AVINV_F[j][k] for j in range(3))) for k in range(3)]
        out.append(dict(T=c["T"], P=c["P"], pos_cp=q, vel_cp=v))
    json.dump(dict(A_POS_INV=APINV.tolist(), A_VEL_INV=AVINV.tolist(), cases=out),
              open(os.path.join(HERE, "minvo_kat.json"), "w"), indent=1)
    return out


# -------------------------------------------------------------------------------------------------
# QP in the reference's variable space (numpy restatement of solver_gurobi_poly.cpp:322-710)
# -------------------------------------------------------------------------------------------------
def build_qp(K, T, weight, mins, maxs, v_max, a_max, coeff_init, line_seg, line_nd, relaxed):
    n = 12 * K

    def var(ax, seg, j):
        return ax * 4 * K + seg * 4 + j
    M4 = pos_inv_T(T); V3 = vel_inv321_T(T)
    tp = np.array([T ** 3, T ** 2, T, 1.0]); qv = np.array([3 * T * T, 2 * T, 1.0, 0.0]); qa = np.array([6 * T, 2.0, 0, 0])
    final = np.array([tp @ coeff_init[ax, K - 1] for ax in range(3)])
    Pm = np.zeros((n, n)); q = np.zeros(n); c0 = 0.0
    for i in range(K):
        for ax in range(3):
            Pm[var(ax, i, 0), var(ax, i, 0)] += 2 * 36 * T
    for ax in range(3):
        idx = [var(ax, K - 1, j) for j in range(4)]
        Pm[np.ix_(idx, idx)] += 2 * weight * np.outer(tp, tp)
        q[idx] += -2 * weight * tp * final[ax]
        c0 += weight * final[ax] ** 2
        if relaxed:
            Pm[np.ix_(idx, idx)] += 2 * weight * (np.outer(qv, qv) + np.outer(qa, qa))
    E = []; e = []
    for k1 in range(1, 4):
        for ax in range(3):
            r = np.zeros(n); r[var(ax, 0, k1)] = 1; E.append(r); e.append(coeff_init[ax, 0, k1])
    for i in range(K - 1):
        for ax in range(3):
            r = np.zeros(n); r[[var(ax, i, j) for j in range(4)]] = tp; r[var(ax, i + 1, 3)] = -1; E.append(r); e.append(0)
            r = np.zeros(n); r[[var(ax, i, j) for j in range(4)]] = qv; r[var(ax, i + 1, 2)] = -1; E.append(r); e.append(0)
            r = np.zeros(n); r[[var(ax, i, j) for j in range(4)]] = qa; r[var(ax, i + 1, 1)] = -2; E.append(r); e.append(0)
    if not relaxed:
        for ax in range(3):
            r = np.zeros(n); r[[var(ax, K - 1, j) for j in range(4)]] = qv; E.append(r); e.append(0)
            r = np.zeros(n); r[[var(ax, K - 1, j) for j in range(4)]] = qa; E.append(r); e.append(0)
    G = []; h = []
    for i in range(K):
        for ax in range(3):
            idx = [var(ax, i, j) for j in range(4)]
            for k in range(4):
                r = np.zeros(n); r[idx] = M4[:, k]; G.append(r); h.append(maxs[ax]); G.append(-r); h.append(-mins[ax])
            for k in range(3):
                r = np.zeros(n); r[idx[:3]] = V3[:, k]; G.append(r); h.append(v_max); G.append(-r); h.append(v_max)
            r = np.zeros(n); r[idx[0]] = 6 * T; r[idx[1]] = 2; G.append(r); h.append(a_max); G.append(-r); h.append(a_max)
        for s, nd in zip(line_seg, line_nd):
            if s != i:
                continue
            for k in range(4):
                r = np.zeros(n)
                r[[var(0, i, j) for j in range(4)]] = nd[0] * M4[:, k]
                r[[var(1, i, j) for j in range(4)]] = nd[1] * M4[:, k]
                G.append(r); h.append(1 - nd[2])
    init_pos = coeff_init[:, 0, 3]
    has_qc = np.linalg.norm(init_pos - final) < 1.0
    Cq = np.zeros((n, n)); cq = np.zeros(n); cc = -0.01
    for ax in range(3):
        idx = [var(ax, K - 1, j) for j in range(4)]
        Cq[np.ix_(idx, idx)] += np.outer(tp, tp); cq[idx] += -tp * final[ax]; cc += final[ax] ** 2
    return dict(n=n, P=Pm, q=q, c0=c0, E=np.array(E), e=np.array(e, dtype=float), G=np.array(G), h=np.array(h, dtype=float),
                has_qc=bool(has_qc), Cq=Cq, cq=cq, cc=cc, final=final)


def solve_two_ways(Q, x0):
    """trust-constr and SLSQP on the equality-eliminated problem (exact null-space elimination
    of E via SVD keeps both solvers well conditioned); returns (theta, cost) or None."""
    E, e = Q["E"], Q["e"]
    U, s, Vt = np.linalg.svd(E, full_matrices=True)
    rank = int((s > 1e-10 * s[0]).sum())
    xp = np.linalg.lstsq(E, e, rcond=None)[0]
    if np.abs(E @ xp - e).max() > 1e-8:
        return None
    N = Vt[rank:].T
    nz = N.shape[1]
    Pr = N.T @ Q["P"] @ N; qr = N.T @ (Q["P"] @ xp + Q["q"]); cr = 0.5 * xp @ Q["P"] @ xp + Q["q"] @ xp + Q["c0"]
    Gr = Q["G"] @ N; hr = Q["h"] - Q["G"] @ xp

    def f(z): return 0.5 * z @ Pr @ z + qr @ z + cr
    def g(z): return Pr @ z + qr
    def th(z): return N @ z + xp
    def qc(z):
        t = th(z); return -(t @ Q["Cq"] @ t + 2 * Q["cq"] @ t + Q["cc"])
    def qcj(z):
        t = th(z); return -(2 * (Q["Cq"] @ t + Q["cq"])) @ N
    if nz == 0:
        t = xp
        feas = (Q["G"] @ t - Q["h"]).max() <= 1e-7 and (not Q["has_qc"] or qc(np.zeros(0)) >= -1e-9)
        return (t, float(cr), 0.0, 0.0) if feas else None
    z0 = N.T @ (x0 - xp)
    cons_tc = [so.LinearConstraint(Gr, -np.inf, hr)]
    cons_sl = [dict(type="ineq", fun=lambda z: hr - Gr @ z, jac=lambda z: -Gr)]
    if Q["has_qc"]:
        cons_tc.append(so.NonlinearConstraint(lambda z: -qc(z), -np.inf, 0.0, jac=lambda z: -qcj(z),
                                              hess=lambda z, v: v[0] * 2 * N.T @ Q["Cq"] @ N))
        cons_sl.append(dict(type="ineq", fun=qc, jac=qcj))
    r1 = so.minimize(f, z0, jac=g, hess=lambda z: Pr, method="trust-constr", constraints=cons_tc,
                     options=dict(gtol=1e-10, xtol=1e-12, barrier_tol=1e-11, maxiter=1500))
    r2 = so.minimize(f, z0, jac=g, method="SLSQP", constraints=cons_sl, options=dict(ftol=1e-15, maxiter=2000))
    ok1 = (Gr @ r1.x - hr).max() <= 1e-7 and (not Q["has_qc"] or qc(r1.x) >= -1e-7)
    ok2 = (Gr @ r2.x - hr).max() <= 1e-7 and (not Q["has_qc"] or qc(r2.x) >= -1e-7)
    if not (ok1 and ok2):
        print("  solvers: ok1=%s ok2=%s f1=%.9g f2=%.9g viol1=%.2e viol2=%.2e" % (ok1, ok2, r1.fun, r2.fun, (Gr @ r1.x - hr).max(), (Gr @ r2.x - hr).max()))
        return None
    if abs(r1.fun - r2.fun) > 5e-6 * (1 + abs(r1.fun)):
        print("  solvers disagree: f1=%.12g f2=%.12g" % (r1.fun, r2.fun))
        return None
    best = r1 if r1.fun <= r2.fun else r2
    zb, fb = best.x, float(best.fun)
    # active-set polish: with the active rows identified, the optimum solves a linear KKT system
    if not (Q["has_qc"] and qc(zb) < 1e-6):
        for tol in (1e-6, 1e-5, 1e-7):
            act = np.where(hr - Gr @ zb < tol)[0]
            Ga = Gr[act]
            KKT = np.block([[Pr, Ga.T], [Ga, np.zeros((len(act), len(act)))]])
            sol = np.linalg.lstsq(KKT, np.concatenate([-qr, hr[act]]), rcond=None)[0]
            zp, lam = sol[:nz], sol[nz:]
            if (Gr @ zp - hr).max() <= 1e-9 and (len(lam) == 0 or lam.min() >= -1e-7) and \
               np.abs(Pr @ zp + qr + Ga.T @ lam).max() <= 1e-7 * (1 + np.abs(qr).max()) and \
               (not Q["has_qc"] or qc(zp) >= 0) and f(zp) <= fb + 1e-9 * (1 + abs(fb)):
                zb, fb = zp, float(f(zp))
                break
        else:
            print("  (polish failed, keeping best-of-two)")
    return th(zb), fb, float(abs(r1.fun - r2.fun)), float(np.abs(th(r1.x) - th(r2.x)).max())


def linear_feasible(Q):
    """HiGHS: is {E th = e, G th <= h} non-empty?  (ignores the ball constraint)"""
    r = so.linprog(np.zeros(Q["n"]), A_ub=Q["G"], b_ub=Q["h"], A_eq=Q["E"], b_eq=Q["e"], bounds=(None, None), method="highs")
    return r.status == 0


def _one_case(tag, p, K, ci, seg, nd, cases):
    mins = [p.x_min, p.y_min, p.z_min]; maxs = [p.x_max, p.y_max, p.z_max]
    seg = np.asarray(seg, dtype=np.int32); nd = np.asarray(nd, dtype=np.float64).reshape(-1, 3)
    Q = build_qp(K, p.T_span, p.weight, mins, maxs, p.v_max, p.a_max, ci, seg, nd, relaxed=False)
    x0 = np.concatenate([ci[ax].reshape(-1) for ax in range(3)])
    sol = solve_two_ways(Q, x0) if linear_feasible(Q) else None
    status = 0
    if sol is None:
        Q = build_qp(K, p.T_span, p.weight, mins, maxs, p.v_max, p.a_max, ci, seg, nd, relaxed=True)
        sol = solve_two_ways(Q, x0) if linear_feasible(Q) else None
        status = 1 if sol is not None else 2
        if sol is None and linear_feasible(Q):
            print("  (skipped %s: relaxed problem linear-feasible but solvers disagree)" % tag); return
    if status == 2:
        theta, cost, dcost, dth = x0, float("nan"), 0.0, 0.0
    else:
        theta, cost, dcost, dth = sol
    G, h = Q["G"], Q["h"]
    n_active = int(((G @ theta - h) > -1e-6).sum()) if status != 2 else 0
    cases.append(dict(tag=tag, K=K, status=status, coeff_init=ci, line_seg=seg, line_nd=nd,
                      theta=theta.reshape(3, K, 4), cost=cost, dcost=dcost, dth=dth, qc=Q["has_qc"], n_active=n_active,
                      mins=mins, maxs=maxs, T=p.T_span, weight=p.weight, v_max=p.v_max, a_max=p.a_max))
    print("case %-28s K=%d status=%d cost=%.9g |d cost|=%.2e |d th|=%.2e lines=%d active=%d qc=%d" %
          (tag, K, status, cost, dcost, dth, len(seg), n_active, Q["has_qc"]), flush=True)


def tight_lines(ci, T, rng, per_seg=2, margin=(0.0, 0.05)):
    """Random lines the guess satisfies with a small margin, so that they bind at the optimum."""
    K = ci.shape[1]; M4 = pos_inv_T(T)
    seg, nd = [], []
    for i in range(K):
        q = np.stack([ci[0, i] @ M4, ci[1, i] @ M4], 1)  # [4][2]
        for _ in range(per_seg):
            ang = rng.uniform(0, 2 * np.pi); n = np.array([np.cos(ang), np.sin(ang)])
            c = (q @ n).max() + rng.uniform(*margin)
            s = 2.0 / rng.uniform(0.1, 2.0)
            seg.append(i); nd.append([s * n[0], s * n[1], 1 - s * c])
    return seg, nd


def qp_cases():
    from neptune_amd import scene
    from oracle import oracle
    cases = []
    # (a) scene-derived: lines from the separator on seeded scenes (mostly inactive constraints)
    for (N, S, K, seed, agents) in [(1, 0, 3, 0, [1]), (1, 0, 8, 1, [1]), (5, 0, 8, 2, [1, 3]), (8, 20, 8, 4, [2, 4, 6]),
                                    (8, 20, 5, 6, [1, 2]), (5, 0, 4, 7, [1]), (3, 4, 2, 9, [1, 2]), (3, 4, 1, 10, [1, 2])]:
        sc = scene.make_scene(N, S, seed=seed, K=K, separation="aabb")
        p = sc["par"]
        for a in agents:
            g = sc["guesses"][a - 1]
            ci = np.array(g["coeff"])[:, :K, :]
            r = oracle.replan(p, a, sc["committed"], g, sc["statics"])  # only to obtain the lines
            _one_case("scene N%d S%d seed%d a%d" % (N, S, seed, a), p, K, ci, r["line_seg"], r["line_nd"], cases)
    # (b) binding constraints: random tight lines around the guess
    rng = np.random.default_rng(2024)
    for t, (K, seed) in enumerate([(8, 11), (8, 12), (8, 13), (8, 14), (6, 15), (5, 16), (4, 17), (3, 18), (8, 19), (8, 20)]):
        sc = scene.make_scene(5, 0, seed=seed, K=K, separation="aabb")
        p = sc["par"]; a = int(rng.integers(0, 5))
        ci = np.array(sc["guesses"][a]["coeff"])[:, :K, :]
        seg, nd = tight_lines(ci, p.T_span, rng, per_seg=2 + t % 2)
        _one_case("tight K%d seed%d" % (K, seed), p, K, ci, seg, nd, cases)
    # (c) goal reached: hovering agent -> terminal ball constraint present
    p = scene.scaled_params(5, 0)
    for K in (8, 3):
        ci = np.zeros((3, K, 4)); ci[:, :, 3] = np.array([2.0, -3.0, 1.0])[:, None]
        _one_case("hover K%d" % K, p, K, ci, [], [], cases)
    # short hop with the ball constraint and binding lines
    for seed in (31, 32):
        rng2 = np.random.default_rng(seed)
        ci = scene.rollout(np.array([1.0, 1.0, 1.0]), np.array([0.3, -0.2, 0.0]), np.zeros(3), np.array([1.5, 0.6, 1.0]), p, 8)
        seg, nd = tight_lines(ci, p.T_span, rng2, per_seg=1)
        _one_case("hop qc seed%d" % seed, p, 8, ci, seg, nd, cases)
    # (d) failure paths: primary infeasible (cannot stop in time) -> relaxed solve
    for K, v0 in ((2, 1.9), (1, 1.0), (3, 1.95)):
        ci = scene.rollout(np.array([0.0, 0.0, 1.0]), np.array([v0, 0.5 * v0, 0.0]), np.array([1.0, 0.0, 0.0]), np.array([8.0, 3.0, 1.0]), p, K)
        _one_case("nostop K%d" % K, p, K, ci, [], [], cases)
    # (e) both infeasible: contradictory lines -> output == initial guess
    ci = scene.rollout(np.array([0.0, 0.0, 1.0]), np.zeros(3), np.zeros(3), np.array([4.0, 1.0, 1.0]), p, 8)
    _one_case("contradictory", p, 8, ci, [2, 2], [[2.0, 0.0, 1 - 2.0 * 0.0], [-2.0, 0.0, 1 + 2.0 * 50.0]], cases)
    # (f) rest-to-rest K=1/K=2 consistent equalities
    for K in (1, 2):
        ci = np.zeros((3, K, 4)); ci[:, :, 3] = np.array([-1.0, 4.0, 1.0])[:, None]
        _one_case("rest K%d" % K, p, K, ci, [], [], cases)
    return cases


def save_qp_cases(cases, name="qp_cases.npz"):
    d = {"n": np.array(len(cases))}
    for i, c in enumerate(cases):
        for k, v in c.items():
            d["c%d_%s" % (i, k)] = np.asarray(v)
    np.savez_compressed(os.path.join(HERE, name), **d)


def lp_cases():
    rng = np.random.default_rng(77)
    A_all, B_all, feas = [], [], []
    for t in range(300):
        nA = int(rng.integers(1, 9)); nB = 4
        ca = rng.uniform(-3, 3, size=2); cb = rng.uniform(-3, 3, size=2)
        A = ca + rng.normal(size=(nA, 2)) * rng.uniform(0.1, 1.5)
        B = cb + rng.normal(size=(nB, 2)) * rng.uniform(0.1, 1.5)
        if t % 7 == 0:
            B[:] = B[0]  # resting agent: coincident control points
        Aub = np.concatenate([-np.c_[A, np.ones(nA)], np.c_[B, np.ones(nB)]]); bub = -np.ones(nA + nB)
        r = so.linprog(np.zeros(3), A_ub=Aub, b_ub=bub, bounds=(None, None), method="highs")
        Ap = np.full((8, 2), np.nan); Ap[:nA] = A
        A_all.append(Ap); B_all.append(B); feas.append(r.status == 0)
    np.savez_compressed(os.path.join(HERE, "lp_cases.npz"), A=np.array(A_all), B=np.array(B_all), feasible=np.array(feas))
    print("lp cases: %d feasible of %d" % (sum(feas), len(feas)))


if __name__ == "__main__":
    minvo_kat()
    lp_cases()
    save_qp_cases(qp_cases())
