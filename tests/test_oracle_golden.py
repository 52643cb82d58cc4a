"""CPU: the oracle against the committed golden fixtures (tests/golden/, made by make_golden.py)."""
import numpy as np
import pytest

import helpers
from neptune_amd import abi, scene


def test_minvo_kat(oracle):
    kat = helpers.load_kat()
    for c in kat["cases"]:
        q = oracle.pos_ctrl_pts(c["P"], c["T"]); v = oracle.vel_ctrl_pts(c["P"], c["T"])
        np.testing.assert_allclose(q, c["pos_cp"], rtol=0, atol=5e-14 * (1 + np.abs(c["pos_cp"]).max()))
        np.testing.assert_allclose(v, c["vel_cp"], rtol=0, atol=5e-14 * (1 + np.abs(c["vel_cp"]).max()))
    # the survey's published known answer
    np.testing.assert_allclose(oracle.pos_ctrl_pts([1, -2, 0.5, 3], 0.5),
                               [3.0029164208, 3.0625329272, 2.9688574246, 2.8574380317], atol=1e-9)
    np.testing.assert_allclose(oracle.vel_ctrl_pts([1, -2, 0.5, 3], 0.5), [0.5966878365, -0.375, -0.8466878365], atol=1e-9)


def test_minvo_hull_contains_curve(oracle):
    rng = np.random.default_rng(0)
    for _ in range(50):
        P = rng.normal(size=4) * 2; T = 0.5
        q = oracle.pos_ctrl_pts(P, T)
        t = np.linspace(0, T, 101)
        y = P[0] * t ** 3 + P[1] * t ** 2 + P[2] * t + P[3]
        assert y.min() >= q.min() - 1e-9 and y.max() <= q.max() + 1e-9


def test_qp_against_golden(oracle):
    cases = helpers.load_qp_cases()
    assert len(cases) >= 30
    seen = set()
    for c in cases:
        p = helpers.params_of_case(c)
        r = oracle.optimize(p, 1, c["coeff_init"], [], [], lines=(c["line_seg"], c["line_nd"]))
        assert r["status"] == c["status"], c["tag"]
        seen.add(c["status"])
        th = helpers.golden_theta_out(c)
        assert np.abs(r["coeff"] - th).max() <= helpers.theta_tol(c), c["tag"]
        if c["status"] != 2:
            assert abs(r["objective"] - c["cost"]) <= 1e-4 * (1 + abs(c["cost"])) * 0.05, c["tag"]
    assert seen == {0, 1, 2}


def test_qp_against_golden_r2(oracle):
    """The sizes and inputs the first fixture set did not reach (tests/golden/make_golden_r2.py): BASELINE config 4
    (~510 lines) and config 5 (~2 050 lines) replans, K = 7, front-end (lattice) guesses.  Every
    fixture carries a KKT certificate, i.e. it is the optimum to ~1e-9 whatever SciPy's two solvers did."""
    cases = helpers.load_qp_cases("qp_cases_r2.npz")
    tags = [c["tag"] for c in cases]
    assert sum(t.startswith("c4 ") for t in tags) >= 6 and sum(t.startswith("c5 ") for t in tags) >= 2
    assert sum(t.startswith("fe ") for t in tags) >= 10 and any(c["K"] == 7 for c in cases)
    worst = 0.0
    for c in cases:
        p = helpers.params_of_case(c)
        r = oracle.optimize(p, 1, c["coeff_init"], [], [], lines=(c["line_seg"], c["line_nd"]))
        assert r["status"] == c["status"], c["tag"]
        th = helpers.golden_theta_out(c)
        err = np.abs(r["coeff"] - th).max(); worst = max(worst, err)
        assert err <= 1e-8, (c["tag"], err)      # observed: 6.4e-11 (the fixtures are certified optima)
        if c["status"] != 2:
            assert abs(r["objective"] - c["cost"]) <= 1e-8 * (1 + abs(c["cost"])), c["tag"]


def test_separator_feasibility_matches_highs(oracle):
    d = np.load(helpers.ROOT + "/tests/golden/lp_cases.npz")
    n_ok = 0
    for A, B, feas in zip(d["A"], d["B"], d["feasible"]):
        A = A[~np.isnan(A[:, 0])]
        ok, nd = oracle.separator(A, B)
        ok2, nd2 = oracle.separator(A, B, simplex=True)
        if ok:
            assert (A @ nd[:2] + nd[2]).min() >= 1 - 1e-9 and (B @ nd[:2] + nd[2]).max() <= -1 + 1e-9
            # vertex of the LP: at least three tight rows unless points coincide
            tight = (np.abs(A @ nd[:2] + nd[2] - 1) < 1e-7).sum() + (np.abs(B @ nd[:2] + nd[2] + 1) < 1e-7).sum()
            assert tight >= 2
            n_ok += 1
        if ok2:
            assert (A @ nd2[:2] + nd2[2]).min() >= 1 - 1e-6 and (B @ nd2[:2] + nd2[2]).max() <= -1 + 1e-6
        # gaps below the separator's 1e-7 floor are the only place the two may differ
        if ok != bool(feas):
            assert not ok  # never claims a separation HiGHS rejects
        assert ok2 == bool(feas) or not ok
    assert n_ok > 200


def test_glpk_class_simplex_on_golden_lps(oracle):
    """Separator rule 1 (orc_separator_glpk_class: primal simplex from the standard basis, projected steepest-edge pricing,
    Harris ratio test — the algorithm class glp_simplex runs with its defaults, separator_glpk.cpp:39-41, 336): feasibility
    equals HiGHS' on every golden LP, the returned point satisfies every row of the reference LP and is a vertex of it in
    GLPK's standard form (three non-basic variables: tight rows, or structurals left at zero)."""
    d = np.load(helpers.ROOT + "/tests/golden/lp_cases.npz")
    n_ok = n_same = 0
    pivots = []
    for A, B, feas in zip(d["A"], d["B"], d["feasible"]):
        A = A[~np.isnan(A[:, 0])]
        ok, nd, npiv = oracle.separator_glpk_class(A, B)
        assert ok == bool(feas)
        if not ok:
            assert (nd == 0).all()
            continue
        ra = A @ nd[:2] + nd[2]; rb = B @ nd[:2] + nd[2]
        assert ra.min() >= 1 - 3e-7 and rb.max() <= -1 + 3e-7                      # (tol_bnd = 1e-7, relative to the bound)
        tight = (np.abs(ra - 1) < 1e-9).sum() + (np.abs(rb + 1) < 1e-9).sum() + (nd == 0).sum()
        assert tight >= 3
        n_ok += 1; pivots.append(npiv)
        ok0, nd0 = oracle.separator(A, B)
        if ok0:
            l = np.array([nd[0], nd[1], nd[2] - 1]) / np.hypot(*nd[:2]); l0 = np.array([nd0[0], nd0[1], nd0[2] - 1]) / np.hypot(*nd0[:2])
            n_same += np.abs(l - l0).max() < 1e-9
    assert n_ok > 200 and max(pivots) <= 20
    assert 0 < n_same < n_ok           # a different rule: it meets the largest-gap line on some LPs, not on all
    # the rule switch of the restated path: orc_separator* follow it
    A = d["A"][0]; A = A[~np.isnan(A[:, 0])]; B = d["B"][0]
    oracle.set_separator_rule(1)
    try:
        ok1, nd1 = oracle.separator(A, B)
    finally:
        oracle.set_separator_rule(0)
    ok2, nd2, _ = oracle.separator_glpk_class(A, B)
    assert ok1 == ok2 and nd1.tobytes() == nd2.tobytes()


def test_hull_properties(oracle):
    rng = np.random.default_rng(3)
    for _ in range(100):
        n = int(rng.integers(1, 40))
        pts = rng.normal(size=(n, 2))
        if n > 4:
            pts[rng.integers(0, n)] = pts[0]  # duplicate
        h = oracle.convex_hull_2d(pts)
        ref = scene.hull_ccw_lexmin(pts)
        np.testing.assert_array_equal(h, ref)
        assert tuple(h[0]) == min(map(tuple, pts))
        if len(h) >= 3:
            e = np.roll(h, -1, 0) - h
            cr = e[:, 0] * np.roll(e, -1, 0)[:, 1] - e[:, 1] * np.roll(e, -1, 0)[:, 0]
            assert (cr > 0).all()  # strictly convex, counter-clockwise


def test_sampling_matches_reference_loop(oracle):
    co = np.random.default_rng(1).normal(size=(3, 8, 4))
    st = oracle.sample(co, 0.5, 0.05)
    # independent restatement of solver_gurobi_poly.cpp:911-934
    t = 0.0; i = 0; out = []
    while i < 8:
        dt = t - i * 0.5
        out.append([[co[ax, i] @ np.array([dt ** 3, dt ** 2, dt, 1]) for ax in range(3)],          # pos   (:921-923)
                    [co[ax, i] @ np.array([3 * dt ** 2, 2 * dt, 1, 0]) for ax in range(3)],       # vel   (:924-926)
                    [co[ax, i] @ np.array([6 * dt, 2, 0, 0]) for ax in range(3)],                 # accel (:927-929)
                    [6 * co[ax, i, 0] for ax in range(3)]])                                      # jerk  (:930-932)
        t += 0.05
        if t > (i + 1) * 0.5:
            i += 1
    out = np.array(out).reshape(len(out), 12)
    assert len(st) == len(out) and 80 <= len(st) <= 90
    np.testing.assert_allclose(st, out, atol=1e-12)


def test_reduced_model_matches_oracle(oracle):
    """The numpy model of the reduced structured interior point (what the HIP kernel implements)
    agrees with the full-space oracle on every golden case."""
    import reduced_ipm_ref as R
    for c in helpers.load_qp_cases():
        p = helpers.params_of_case(c)
        r = oracle.optimize(p, 1, c["coeff_init"], [], [], lines=(c["line_seg"], c["line_nd"]))
        st, th, obj, it = R.optimize(c["K"], p.T_span, p.weight, c["coeff_init"], c["mins"], c["maxs"], p.v_max, p.a_max,
                                     c["line_seg"], c["line_nd"])
        assert st == r["status"], c["tag"]
        assert np.abs(th - r["coeff"]).max() < 1e-6, c["tag"]   # the stated bar (two interior-point paths, same optimum)


def test_ordered_polygon_rule_agrees_with_all_pairs(oracle):
    """For a counter-clockwise convex polygon A the edge-only rule (A rows satisfied by convexity)
    picks the same LP vertex as the exhaustive all-pairs rule."""
    rng = np.random.default_rng(9)
    n_sep = 0
    for _ in range(300):
        A = scene.hull_ccw_lexmin(rng.normal(size=(int(rng.integers(3, 14)), 2)) * rng.uniform(0.3, 2.0) + rng.uniform(-3, 3, size=2))
        if len(A) < 3:
            continue
        B = rng.normal(size=(4, 2)) * rng.uniform(0.1, 1.0) + rng.uniform(-4, 4, size=2)
        ok1, nd1 = oracle.separator(A, B)
        ok2, nd2 = oracle.separator(A, B, ordered=True)
        assert ok1 == ok2
        if ok1:
            n_sep += 1
            np.testing.assert_allclose(nd1, nd2, rtol=1e-9, atol=1e-9)
            assert (A @ nd2[:2] + nd2[2]).min() >= 1 - 1e-9 and (B @ nd2[:2] + nd2[2]).max() <= -1 + 1e-9
    assert n_sep > 100


def test_hull_matches_qhull(oracle):
    """cu::convexHullOfPoints2d is CGAL's convex_hull_2 in the reference (absent here): the restatement is
    pinned against an independent implementation, Qhull (scipy.spatial), on points in general position —
    same extreme points, same counter-clockwise order, started at the lexicographically smallest."""
    from scipy.spatial import ConvexHull
    rng = np.random.default_rng(17)
    for _ in range(200):
        n = int(rng.integers(3, 49))
        pts = rng.normal(size=(n, 2)) * rng.uniform(0.1, 5.0)
        q = pts[ConvexHull(pts).vertices]                      # counter-clockwise in 2-D
        k = min(range(len(q)), key=lambda i: tuple(q[i]))
        np.testing.assert_array_equal(oracle.convex_hull_2d(pts), np.roll(q, -k, axis=0))


def test_hard_closed_loop_replans_against_highs_labels(oracle):
    """The hard replans of the closed loop (tests/golden/moving_hard_cases.npz: 120 of the 400 dumped from three rounds of bench.py's
    `moving` leg — every one the device did not solve at the first attempt, a sample of those that took more than 14 iterations, a few
    easy ones) against HiGHS on the reference's linear rows (solver_gurobi_poly.cpp:832-861 decides the status from them):
    where the rows are DECISIVELY infeasible the oracle must fail (first and relaxed problem -> FAILED; first only -> RELAXED or worse);
    where HiGHS finds an interior the oracle should solve — it does not always (feasible sets whose optimum is degenerate: the interior
    point's gap stalls): those are counted and bounded, not hidden."""
    p, cases = helpers.load_moving_hard_cases()
    n_lab = {0: 0, 1: 0, 2: 0}; missed = []
    for k, c in enumerate(cases):
        r = oracle.optimize(p, 1, c["coeff"], [], [], lines=(c["seg"], c["nd"]))
        e = c["expected"]
        if e < 0:
            continue
        n_lab[e] += 1
        if e == 2:
            assert r["status"] == abi.NEP_FAILED, (k, r["status"])
        elif e == 1:
            assert r["status"] == abi.NEP_RELAXED, (k, r["status"])
        elif r["status"] != abi.NEP_OK:
            missed.append((k, r["status"], c["t_first"]))
    assert n_lab[2] >= 30 and n_lab[1] >= 5 and n_lab[0] >= 20
    print("oracle gave up on %d of %d replans HiGHS finds strictly feasible: %r" % (len(missed), n_lab[0], missed))
    assert len(missed) <= 6
