"""TEST INFRASTRUCTURE (never imported by the product): opt-in adapters to the REAL solvers the reference calls on this path,
for boxes that have them.  Neither is in this image — `probe()` says so explicitly and the tests skip with
the reason — but wherever `libglpk` or `gurobipy` exists these drive them exactly as the reference does, which is the only
way the parity of rows a9 / c can ever be pinned against the reference's own arithmetic.

  GLPK    ctypes on libglpk: the call sequence of reference submodules/separator/src/separator_glpk.cpp:258-349
          (glp_create_prob, GLP_MAX, rows GLP_LO 1 / GLP_UP -1, three free columns with zero objective, glp_load_matrix in the
          reference's (row, column) order, glp_simplex with glp_init_smcp defaults + msg_lev = 1 (:39-41), status OPT|FEAS
          (:367), glp_delete_prob).
  Gurobi  gurobipy: the 12K-variable model of reference neptune/src/solver_gurobi_poly.cpp:322-710 fed from the same dense
          matrices the golden generator builds (tests/golden/make_golden.py::build_qp: objective :322-383, equalities
          :390-425,659-678, box rows :433-471, line rows :485-489, terminal ball :680-702), OutputFlag 0 and TimeLimit only
          (:811-812), the accepted statuses of :832-836 and the relaxed re-solve of :838-861.
"""
import ctypes as C
import ctypes.util
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

GLP_MAX, GLP_FR, GLP_LO, GLP_UP = 2, 1, 2, 3
GLP_FEAS, GLP_OPT = 2, 5

_glpk = [False, None]


def glpk_lib():
    """libglpk through ctypes, or None"""
    if _glpk[0]:
        return _glpk[1]
    _glpk[0] = True
    name = os.environ.get("NEP_GLPK_LIB") or ctypes.util.find_library("glpk")
    if not name:
        return None
    try:
        L = C.CDLL(name)
    except OSError:
        return None
    vp, i, d = C.c_void_p, C.c_int, C.c_double
    L.glp_create_prob.restype = vp
    L.glp_set_prob_name.argtypes = [vp, C.c_char_p]
    L.glp_set_obj_dir.argtypes = [vp, i]
    L.glp_add_rows.argtypes = [vp, i]; L.glp_add_cols.argtypes = [vp, i]
    L.glp_set_row_bnds.argtypes = [vp, i, i, d, d]; L.glp_set_col_bnds.argtypes = [vp, i, i, d, d]
    L.glp_set_col_name.argtypes = [vp, i, C.c_char_p]
    L.glp_set_obj_coef.argtypes = [vp, i, d]
    L.glp_load_matrix.argtypes = [vp, i, C.POINTER(i), C.POINTER(i), C.POINTER(d)]
    L.glp_init_smcp.argtypes = [vp]
    L.glp_simplex.argtypes = [vp, vp]
    L.glp_get_col_prim.argtypes = [vp, i]; L.glp_get_col_prim.restype = d
    L.glp_get_obj_val.argtypes = [vp]; L.glp_get_obj_val.restype = d
    L.glp_get_status.argtypes = [vp]
    L.glp_delete_prob.argtypes = [vp]
    L.glp_version.restype = C.c_char_p
    _glpk[1] = L
    return L


def glpk_separator(A, B):
    """Separator::solveModel(Vector3d&, 2xN A, 2xN B) (separator_glpk.cpp:248-373) on the real GLPK -> (ok, (n1, n2, d))"""
    L = glpk_lib()
    if L is None:
        raise RuntimeError("libglpk not found")
    A = np.asarray(A, dtype=np.float64).reshape(-1, 2); B = np.asarray(B, dtype=np.float64).reshape(-1, 2)
    nA, nB = len(A), len(B)
    lp = L.glp_create_prob()
    L.glp_set_prob_name(lp, b"separator")
    L.glp_set_obj_dir(lp, GLP_MAX)
    L.glp_add_rows(lp, nA + nB)
    for r in range(nA):
        L.glp_set_row_bnds(lp, r + 1, GLP_LO, 1.0, 0.0)          # n'xA + d >= epsilon (= 1)
    for r in range(nB):
        L.glp_set_row_bnds(lp, nA + r + 1, GLP_UP, 0.0, -1.0)    # n'xB + d <= -epsilon
    L.glp_add_cols(lp, 3)
    for c_, nm in ((1, b"n1"), (2, b"n2"), (3, b"d")):
        L.glp_set_col_name(lp, c_, nm)
        L.glp_set_col_bnds(lp, c_, GLP_FR, 0.0, 0.0)
        L.glp_set_obj_coef(lp, c_, 0.0)                           # weight_n0 = weight_n1 = 0 (:25-27)
    ne = 3 * (nA + nB)
    ia = (C.c_int * (ne + 1))(); ja = (C.c_int * (ne + 1))(); ar = (C.c_double * (ne + 1))()
    r = 1
    for row, pt in enumerate(np.vstack([A, B])):
        for col, v in ((1, pt[0]), (2, pt[1]), (3, 1.0)):
            ia[r], ja[r], ar[r] = row + 1, col, v
            r += 1
    L.glp_load_matrix(lp, ne, ia, ja, ar)
    parm = (C.c_char * 1024)()                                    # glp_smcp (352 bytes in 4.65); msg_lev is its first int in every version
    L.glp_init_smcp(parm)
    C.cast(parm, C.POINTER(C.c_int))[0] = 1                       # params.msg_lev = 1 (:41)
    L.glp_simplex(lp, parm)
    nd = (L.glp_get_col_prim(lp, 1), L.glp_get_col_prim(lp, 2), L.glp_get_col_prim(lp, 3))
    status = L.glp_get_status(lp)
    L.glp_delete_prob(lp)
    return status in (GLP_OPT, GLP_FEAS), np.array(nd)


def _golden_builder():
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    try:
        import make_golden
    finally:
        sys.path.pop(0)
    return make_golden.build_qp


def gurobi_module():
    try:
        import gurobipy
        return gurobipy
    except Exception:
        return None


def _gurobi_once(gp, Q, time_limit):
    """one m_.optimize() of the reference's model -> (theta, objective) or None ("no solution": :832-836)"""
    n = Q["n"]
    env = gp.Env(empty=True); env.setParam("OutputFlag", 0); env.start()
    m = gp.Model(env=env)
    m.Params.OutputFlag = 0
    m.Params.TimeLimit = time_limit
    x = m.addMVar(n, lb=-gp.GRB.INFINITY, ub=gp.GRB.INFINITY)
    m.setObjective(0.5 * (x @ Q["P"] @ x) + Q["q"] @ x + Q["c0"], gp.GRB.MINIMIZE)
    if len(Q["e"]):
        m.addConstr(Q["E"] @ x == Q["e"])
    if len(Q["h"]):
        m.addConstr(Q["G"] @ x <= Q["h"])
    if Q["has_qc"]:
        m.addConstr(x @ Q["Cq"] @ x + 2.0 * (Q["cq"] @ x) + Q["cc"] <= 0.0)
    m.optimize()
    ok_status = {gp.GRB.OPTIMAL, gp.GRB.TIME_LIMIT, gp.GRB.USER_OBJ_LIMIT, gp.GRB.ITERATION_LIMIT, gp.GRB.NODE_LIMIT, gp.GRB.SOLUTION_LIMIT}
    if m.Status in ok_status and m.SolCount > 0:
        out = (np.array(x.X), float(m.ObjVal))
    else:
        out = None
    m.dispose(); env.dispose()
    return out


def gurobi_optimize(K, T, weight, mins, maxs, v_max, a_max, coeff_init, line_seg, line_nd, time_limit=0.05, solve_once=None):
    """PolySolverGurobi::optimize (solver_gurobi_poly.cpp:804-887) on the real Gurobi: first solve, relaxed re-solve, fall back
    to the guess -> (status 0 | 1 | 2, theta [3][K][4], objective or nan).  `solve_once(Q, time_limit) -> (theta, objective) |
    None` replaces the Gurobi call (the CPU suite runs this function's own logic with a SciPy stand-in, since no box of ours has
    Gurobi: what is then left untested is _gurobi_once alone)."""
    if solve_once is None:
        gp = gurobi_module()
        if gp is None:
            raise RuntimeError("gurobipy not importable")
        solve_once = lambda Q, tl: _gurobi_once(gp, Q, tl)      # noqa: E731
    build_qp = _golden_builder()
    ci = np.asarray(coeff_init, dtype=np.float64)
    for status, relaxed in ((0, False), (1, True)):
        Q = build_qp(K, T, weight, mins, maxs, v_max, a_max, ci, line_seg, line_nd, relaxed=relaxed)
        r = solve_once(Q, time_limit)
        if r is not None:
            th = r[0].reshape(3, K, 4).copy()
            tp = np.array([T ** 3, T ** 2, T, 1.0]); final = ci[:, K - 1, :] @ tp
            if np.hypot(ci[0, 0, 3] - final[0], ci[1, 0, 3] - final[1]) < 1.0:
                th[2] = ci[2]                              # "give more ascending speed if goal is too near" (:878-880)
            return status, th, r[1]
    return 2, ci.copy(), float("nan")


def probe():
    """what a reader of the bench line / test log needs to know: which reference solvers exist on this box"""
    out = {}
    L = glpk_lib()
    out["glpk"] = ("libglpk %s" % L.glp_version().decode()) if L is not None else "absent"
    gp = gurobi_module()
    if gp is None:
        out["gurobi"] = "absent"
    else:
        try:
            out["gurobi"] = "gurobipy %d.%d.%d" % gp.gurobi.version()
        except Exception as e:                      # importable but unusable (no licence)
            out["gurobi"] = "gurobipy importable, unusable: %r" % (e,)
    eigen = [d for d in ("/usr/include/eigen3", "/usr/local/include/eigen3", "/opt/rocm/include/eigen3") if os.path.exists(os.path.join(d, "Eigen", "Core"))]
    out["eigen"] = eigen[0] if eigen else "absent"
    out["note"] = ("the reference's result-defining calls are glp_simplex (separator_glpk.cpp:336) and GRBModel::optimize "
                   "(solver_gurobi_poly.cpp:823); tests/test_reference_solvers.py pins against them wherever they exist")
    return out
