"""GPU tests of the drop-in boundary: the per-agent handle (PolySolverGurobi's call sequence, solver_gurobi_poly.hpp:28-49) against the
batched interface and the oracle, its one-copy-in / one-copy-out data path, the call-sequence errors, the exact-signature shim and the
C++ host class driven through the reference's call sites (neptune.cpp:1514-1527)."""
import numpy as np
import pytest

import helpers
from neptune_amd import abi, scene
from gpu_util import _bounds, _solver, solver_lines_match, COEF_TOL, COST_RTOL

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def be():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from neptune_amd import backend
    return backend


def test_per_agent_api_matches_batch(be, oracle):
    sc = scene.make_scene(5, 3, seed=7)
    p = sc["par"]
    aid = 2
    hx, hn, h0, n0 = be.hulls_batch(sc["committed"], 0.0, p.num_pol, p.T_span, p.drone_radius)
    others = [j for j in range(5) if j != aid - 1]
    s = _solver(be, p, aid)
    s.setStaticObstVert(sc["statics"])
    g = sc["guesses"][aid - 1]; K = int(g["K"])
    s.setInitTrajectory(np.arange(K + 1) * p.T_span, np.array(g["coeff"])[:, :K, :])
    s.setHulls([[hx[j, i, :hn[j, i]] for i in range(p.num_pol)] for j in others])
    s.setHullsNoInflation([[h0[j, i, :n0[j, i]] for i in range(p.num_pol)] if j != aid - 1 else [] for j in range(5)])
    ok, obj = s.optimize()
    r = oracle.replan(p, aid, sc["committed"], g, sc["statics"])
    solver_lines_match(s, r)                            # the default path: verified presolve (near lines first, parked after)
    times, coeff, traj = s.generatePwpOut(12.5, p.dc)
    assert ok and np.abs(coeff - r["coeff"]).max() <= COEF_TOL
    np.testing.assert_allclose(times, 12.5 + np.arange(K + 1) * p.T_span)
    assert s.stats()["solve_us"] > 0
    # every row through the interior point (nep_backend_set_line_cull(h, 0)): the lines in the reference's call order, the same optimum
    s.setLineCull(0.0)
    ok2, obj2 = s.optimize()
    solver_lines_match(s, r, ordered=True)
    _, coeff2, _ = s.generatePwpOut(12.5, p.dc)
    assert ok2 and np.abs(coeff2 - r["coeff"]).max() <= COEF_TOL and np.abs(coeff2 - coeff).max() <= COEF_TOL
    assert abs(obj2 - obj) <= COST_RTOL * (1 + abs(obj))
    s.close()


def test_per_agent_one_copy_in_one_copy_out(be, oracle):
    """The per-agent handle stages its inputs in one page-locked arena (one host-to-device copy per replan) and gets the sampled
    states back with the solution (generatePwpOut at the schedule's dc does no device work).  Checked: the samples of either path
    equal the oracle's on the returned coefficients; a dc other than the schedule's takes the sampling kernel and the NEXT replan
    samples at it; hull lists beyond the arena's capacity re-lay it out without losing what the other setters wrote; the measured
    sequence (nep_backend_debug_time_sequence) returns the same status."""
    sc = scene.make_scene(5, 3, seed=7)
    p = sc["par"]
    aid = 2
    hx, hn, h0, n0 = be.hulls_batch(sc["committed"], 0.0, p.num_pol, p.T_span, p.drone_radius)
    others = [j for j in range(5) if j != aid - 1]
    g = sc["guesses"][aid - 1]; K = int(g["K"])
    hulls = [[hx[j, i, :hn[j, i]] for i in range(p.num_pol)] for j in others]
    hulls0 = [[h0[j, i, :n0[j, i]] for i in range(p.num_pol)] if j != aid - 1 else [] for j in range(5)]
    r = oracle.replan(p, aid, sc["committed"], g, sc["statics"])
    s = _solver(be, p, aid)
    s.setStaticObstVert(sc["statics"])
    times0 = np.arange(K + 1) * p.T_span; co0 = np.array(g["coeff"])[:, :K, :]

    def replan(hl, dc):
        s.setInitTrajectory(times0, co0); s.setHulls(hl); s.setHullsNoInflation(hulls0)
        ok, _ = s.optimize()
        return ok, s.generatePwpOut(0.0, dc)
    ok, (_, coeff, traj) = replan(hulls, p.dc)                       # states came back with the solution
    assert ok and np.abs(coeff - r["coeff"]).max() <= COEF_TOL
    np.testing.assert_allclose(traj, oracle.sample(coeff, p.T_span, p.dc), rtol=0, atol=1e-12)
    _, _, traj_b = s.generatePwpOut(0.0, 0.1)                         # another dc: the sampling kernel
    np.testing.assert_allclose(traj_b, oracle.sample(coeff, p.T_span, 0.1), rtol=0, atol=1e-12)
    ok, (_, coeff_c, traj_c) = replan(hulls, 0.1)                     # the next replan samples at the new dc by itself
    np.testing.assert_array_equal(coeff_c, coeff)
    np.testing.assert_allclose(traj_c, traj_b, rtol=0, atol=1e-12)
    # twelve hull lists (more than the arena was laid out for): the first four as before, the rest copies far away
    far = [[h + np.array([500.0, 500.0]) for h in hulls[k % 4]] for k in range(8)]
    ok, (_, coeff_d, _) = replan(hulls + far, p.dc)
    assert ok and np.abs(coeff_d - coeff).max() <= 1e-9
    st, us, uo = s.timeSequence(times0, co0, hulls, hulls0, dc=p.dc, n_iter=20)
    assert st == 0 and (us > 0).all() and (uo <= us).all()
    s.close()
    # a horizon of more than 128 states at the default dc (num_pol T_span / 0.05 + 1 = 161 at T_span = 1 s): the states that come back
    # with the solution must be ALL of generatePwpOut's time walk (solver_gurobi_poly.cpp:911-934), not the first 128
    s2 = be.PolySolver(p.num_pol, 3, aid, 1.0, p.pb, p.weight, 0.5, True)
    s2.setMaxValues(*_bounds(p)); s2.setMaxRuntime(0.05); s2.setTetherLength(p.tether_length); s2.setStaticObstVert([])
    co2 = co0 * np.array([0.125, 0.25, 0.5, 1.0])                     # the same path flown at half the speed: p2(t) = p(t / 2)
    s2.setInitTrajectory(times0 * 2.0, co2); s2.setHulls([])
    ok2, _ = s2.optimize()
    _, coeff2, traj2 = s2.generatePwpOut(0.0, 0.05)
    want2 = oracle.sample(coeff2, 1.0, 0.05)
    assert ok2 and K == 8 and len(want2) == 161 and traj2.shape == want2.shape
    np.testing.assert_allclose(traj2, want2, rtol=0, atol=1e-12)
    s2.close()


def test_exact_signature_shim_solves_through_the_c_abi(be, tmp_path):
    """tests/cpp/shim_signature_check.cpp on the GPU: `class PolySolverGurobi` (the reference's signatures) constructed and driven in
    Neptune's call order against the stand-in type declarations (tests/cpp/ref_types_min/README.md) — optimize() returns true, K
    segments and K T / dc + 1 states come back, times are shifted by t_start (solver_gurobi_poly.cpp:898)."""
    import test_abi
    if test_abi._find_eigen() is not None:
        pytest.skip("Eigen present")
    r = test_abi._build_shim_check(tmp_path / "shim_signature_check")
    assert r.returncode == 0 and "optimize -> 1" in r.stdout and "segments 4 states 4" in r.stdout and "t0 3.00" in r.stdout, (r.returncode, r.stdout, r.stderr)


def test_call_sequence_errors(be):
    from neptune_amd._lib import BackendError
    p = scene.scaled_params(2, 0)
    s = be.PolySolver(p.num_pol, 3, 1, p.T_span, p.pb, p.weight, 0.5, True)
    with pytest.raises(BackendError):
        s.optimize()                                 # before setMaxValues / setInitTrajectory
    with pytest.raises(BackendError):
        be.PolySolver(p.num_pol, 5, 1, p.T_span, p.pb, p.weight, 0.5, True)   # deg_pol != 3
    with pytest.raises(BackendError):
        be.PolySolver(p.num_pol, 3, 1, p.T_span, p.pb, p.weight, 0.5, False)  # bilinear variant
    s.close()


def test_entangle_through_per_agent_api(be, oracle):
    sc = scene.make_scene(4, 0, seed=13)
    case_id = scene.synthetic_entangle(sc, seed=2, frac=1.0)
    p = sc["par"]; aid = 1; N = 4
    hx, hn, h0, n0 = be.hulls_batch(sc["committed"], 0.0, p.num_pol, p.T_span, p.drone_radius)
    s = _solver(be, p, aid)
    g = sc["guesses"][aid - 1]; K = int(g["K"])
    s.setInitTrajectory(np.arange(K + 1) * p.T_span, np.array(g["coeff"])[:, :K, :])
    others = [j for j in range(N) if j != aid - 1]
    s.setHulls([[hx[j, i, :hn[j, i]] for i in range(p.num_pol)] for j in others])
    s.setHullsNoInflation([[h0[j, i, :n0[j, i]] for i in range(p.num_pol)] if j != aid - 1 else [] for j in range(N)])
    # eu::ent_state per knot: alphas (agent_id, case) + active_cases
    ent = []
    for i in range(K + 1):
        ii = min(i, abi.NEP_MAX_POL - 1)
        alphas = [(j + 1, int(case_id[aid - 1, ii, j])) for j in range(N) if case_id[aid - 1, ii, j]]
        ent.append(dict(alphas=alphas, active_cases=[1 if case_id[aid - 1, ii, j] else 0 for j in range(N)]))
    bend = [np.array(sc["committed"][j]["bend"])[:int(sc["committed"][j]["n_bend"])] for j in range(N)]
    s.setEntStateVector(ent, bend)
    ok, obj = s.optimize()
    r = oracle.replan(p, aid, sc["committed"], g, sc["statics"], case_id=case_id[aid - 1])
    solver_lines_match(s, r)
    t, coeff, traj = s.generatePwpOut(0.0, p.dc)
    assert np.abs(coeff - r["coeff"]).max() <= COEF_TOL
    s.close()


def test_cpp_host_class_reference_call_sequence(be, oracle):
    """tests/cpp/replan_example.cpp drives neptune_amd::PolySolver exactly as neptune.cpp:102-107,
    1514-1527 drives PolySolverGurobi; results match the oracle, failure leaves objective untouched."""
    import os, subprocess
    exe = os.path.join(helpers.ROOT, "tests", "cpp", "replan_example")
    if not os.path.exists(exe):
        import __graft_entry__ as g
        g.build()
    for c in helpers.load_qp_cases():
        if c["tag"] not in ("tight K8 seed12", "hop qc seed32", "nostop K1", "contradictory", "rest K2"):
            continue
        p = helpers.params_of_case(c); K = c["K"]
        txt = "%d %r %r\n" % (K, p.T_span, p.weight)
        txt += " ".join(repr(float(x)) for x in (p.x_min, p.x_max, p.y_min, p.y_max, p.z_min, p.z_max, p.v_max, p.a_max)) + "\n"
        txt += " ".join(repr(float(x)) for x in c["coeff_init"].reshape(-1)) + "\n"
        txt += "%d\n" % len(c["line_seg"])
        for s_, l in zip(c["line_seg"], c["line_nd"]):
            txt += "%d %r %r %r\n" % (int(s_), float(l[0]), float(l[1]), float(l[2]))
        out = subprocess.run([exe], input=txt, capture_output=True, text=True, check=True).stdout.split("\n")
        ok, obj, ns, t0 = out[0].split()
        coeff = np.array([[float(x) for x in ln.split()] for ln in out[1:1 + 3 * K]]).reshape(3, K, 4)
        r = oracle.optimize(p, 1, c["coeff_init"], [], [], lines=(c["line_seg"], c["line_nd"]))
        assert int(ok) == (0 if r["status"] == 2 else 1), c["tag"]
        assert np.abs(coeff - r["coeff"]).max() <= COEF_TOL, c["tag"]
        assert float(t0) == 3.25
        if r["status"] == 2:
            assert float(obj) == -12345.0           # objective_value untouched (solver_gurobi_poly.cpp:856-859)
        else:
            assert abs(float(obj) - r["objective"]) <= COST_RTOL * (1 + abs(r["objective"]))
