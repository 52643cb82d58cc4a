"""The opt-in pin against the REAL solvers of the reference (rows a9 and c of SURVEY.md 8): glp_simplex
(separator_glpk.cpp:336) and GRBModel::optimize (solver_gurobi_poly.cpp:823) are the two result-defining calls of the path,
and neither library is in this image.  Wherever they ARE (NEP_GLPK_LIB / a system libglpk; an importable, licensed gurobipy)
these tests drive them exactly as the reference does (oracle/reference_solvers.py) on the committed golden inputs and compare
with the oracle and — on a GPU box — with the HIP path; where they are not, they SKIP with that reason, and
test_absence_is_reported keeps the reason on record (bench.py prints the same probe as `reference_solvers`).
Tolerances: north star's 1e-4 relative on the cost; positions along the trajectory 1e-4 m."""
import json
import os

import numpy as np
import pytest

import helpers
from conftest import golden_path
from oracle import reference_solvers as rs

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _cases(name):
    return helpers.load_qp_cases(name)


def test_absence_is_reported():
    """never silently: what exists on this box is written down (and lands in gpurun_out/ on a GPU box)"""
    pr = rs.probe()
    assert set(pr) >= {"glpk", "gurobi", "eigen"}
    for k in ("glpk", "gurobi", "eigen"):
        assert isinstance(pr[k], str) and pr[k]
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "reference_solvers_probe.json"), "w") as f:
        json.dump(pr, f)
    print("reference solvers on this box:", pr)


def test_optimize_logic_of_the_gurobi_adapter_with_a_scipy_stand_in():
    """gurobi_optimize's own logic — the model matrices, first solve / relaxed re-solve / fall back to the guess, the z override —
    with SciPy in place of m_.optimize(): reproduces the committed golden statuses and costs.  What no box of ours can test
    is the gurobipy call itself (_gurobi_once)."""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    import make_golden as mg

    def stand_in(Q, time_limit):
        if not mg.linear_feasible(Q):
            return None
        r = mg.solve_two_ways(Q, stand_in.x0)
        return None if r is None else (r[0], r[1])
    picked = [c for c in _cases("qp_cases.npz") if c["K"] <= 4][:6] + [c for c in _cases("qp_cases.npz") if c["status"] != 0][:3]
    assert picked
    for c in picked:
        stand_in.x0 = np.concatenate([c["coeff_init"][ax].reshape(-1) for ax in range(3)])
        st, th, obj = rs.gurobi_optimize(c["K"], float(c["T"]), float(c["weight"]), c["mins"], c["maxs"], float(c["v_max"]), float(c["a_max"]),
                                         c["coeff_init"], c["line_seg"], c["line_nd"], solve_once=stand_in)
        assert st == c["status"], c["tag"]
        if st != 2:
            assert abs(obj - c["cost"]) <= 1e-6 * (1 + abs(c["cost"])), c["tag"]
            assert np.abs(th - helpers.golden_theta_out(c)).max() <= 1e-5, c["tag"]
        else:
            np.testing.assert_array_equal(th, c["coeff_init"])


@pytest.mark.skipif(rs.glpk_lib() is None, reason="libglpk absent on this box (reference: GLPK 4.65, submodules/separator/cmake/glpk.cmake.in:6) — a9 stays unpinned here")
def test_glpk_pin_of_the_separator(oracle):
    """glp_simplex on the 300 golden LPs, driven as separator_glpk.cpp:258-349 drives it: feasibility must equal the golden flag and
    the oracle's; every returned line must satisfy the LP's rows; and the line itself is compared with rule 1 (the GLPK-class
    simplex restated in the oracle and in separator_kernel<1>) and with rule 0 (largest gap) — the agreement rates are the
    pin this row has been missing and are written to gpurun_out/glpk_pin.json."""
    z = np.load(golden_path("lp_cases.npz"))
    same1 = same0 = nfeas = 0
    for A, B, feas in zip(z["A"], z["B"], z["feasible"]):
        A = A[~np.isnan(A[:, 0])]
        ok, nd = rs.glpk_separator(A, B)
        assert ok == bool(feas)
        ok0, nd0 = oracle.separator(A, B)
        ok1, nd1, _ = oracle.separator_glpk_class(A, B)
        assert ok0 == ok and ok1 == ok
        if not ok:
            continue
        nfeas += 1
        assert (A @ nd[:2] + nd[2] >= 1 - 1e-7).all() and (B @ nd[:2] + nd[2] <= -1 + 1e-7).all()
        same1 += bool(np.abs(nd - nd1).max() <= 1e-9 * (1 + np.abs(nd).max()))
        same0 += bool(np.abs(nd - nd0).max() <= 1e-9 * (1 + np.abs(nd).max()))
    rec = {"glpk": rs.probe()["glpk"], "feasible_lps": nfeas, "same_line_as_rule_1": same1, "same_line_as_rule_0": same0}
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    json.dump(rec, open(os.path.join(ROOT, "gpurun_out", "glpk_pin.json"), "w"))
    print("GLPK pin:", rec)
    assert same1 >= 0.9 * nfeas, rec          # rule 1 claims to be glp_simplex's algorithm class: it should reach GLPK's basis nearly always


@pytest.mark.skipif(rs.gurobi_module() is None, reason="gurobipy absent on this box (the reference links Gurobi, README.md:28) — c stays 'parity unpinned' here")
def test_gurobi_pin_of_the_qp(oracle):
    """the reference's model on the real Gurobi against the golden optimum and the oracle: same status, cost within 1e-4 relative
    (north star), positions within 1e-4 m"""
    worst = 0.0
    for name in ("qp_cases.npz", "qp_cases_r2.npz"):
        for c in _cases(name):
            if len(c["line_seg"]) > 600:
                continue                                             # (dense 8 600-row models: minutes each even for Gurobi's Python layer)
            st, th, obj = rs.gurobi_optimize(c["K"], float(c["T"]), float(c["weight"]), c["mins"], c["maxs"], float(c["v_max"]), float(c["a_max"]),
                                             c["coeff_init"], c["line_seg"], c["line_nd"])
            assert st == c["status"], c["tag"]
            if st == 2:
                continue
            rel = abs(obj - c["cost"]) / (1 + abs(c["cost"]))
            worst = max(worst, rel)
            assert rel <= 1e-4, (c["tag"], obj, c["cost"])
            r = oracle.optimize(helpers.params_of_case(c), 1, c["coeff_init"], [], [], lines=(c["line_seg"], c["line_nd"]))
            assert abs(r["objective"] - obj) <= 1e-4 * (1 + abs(obj)), c["tag"]
            tt = np.linspace(0.0, c["T"], 6)
            P = np.stack([tt ** 3, tt ** 2, tt, np.ones_like(tt)])
            assert np.abs(np.einsum("akj,jt->akt", th[:2] - r["coeff"][:2], P)).max() <= 1e-4, c["tag"]
    print("Gurobi pin: worst relative cost difference %.2e" % worst)


@pytest.mark.gpu
def test_reference_solvers_on_the_gpu_box(oracle):
    """the same probe and pins under `-m gpu` (the GPU image is the same as the build container's, but it is the box the judged
    numbers come from: its probe is the one on record in gpurun_out/reference_solvers_probe.json)"""
    test_absence_is_reported()
    if rs.glpk_lib() is not None:
        test_glpk_pin_of_the_separator(oracle)
    if rs.gurobi_module() is not None:
        test_gurobi_pin_of_the_qp(oracle)
