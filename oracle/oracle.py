"""ctypes front of the CPU oracle (oracle/neptune_oracle.c).  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module; the
product (neptune_amd/) never does.
"""
import ctypes as C
import os
import subprocess

import numpy as np

from neptune_amd import abi

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

MAX_LINES = 8192


class orc_params(C.Structure):
    _fields_ = [("num_pol", C.c_int), ("id", C.c_int), ("num_agents", C.c_int),
                ("T_span", C.c_double), ("weight", C.c_double),
                ("mins", C.c_double * 3), ("maxs", C.c_double * 3),
                ("v_max", C.c_double), ("a_max", C.c_double),
                ("pb", C.POINTER(C.c_double))]


class orc_polys(C.Structure):
    _fields_ = [("n", C.c_int), ("off", C.POINTER(C.c_int)), ("xy", C.POINTER(C.c_double))]


class orc_ent(C.Structure):
    _fields_ = [("enabled", C.c_int), ("case_id", C.POINTER(C.c_int)),
                ("bend_off", C.POINTER(C.c_int)), ("bend_xy", C.POINTER(C.c_double)),
                ("hulls_noinfl", C.POINTER(orc_polys))]


class orc_result(C.Structure):
    _fields_ = [("status", C.c_int), ("iters", C.c_int), ("iters_first", C.c_int),
                ("n_lines", C.c_int), ("n_lp", C.c_int), ("n_lp_failed", C.c_int),
                ("n_rows", C.c_int), ("qc_active", C.c_int),
                ("objective", C.c_double),
                ("coeff", ((C.c_double * 4) * abi.NEP_MAX_POL) * 3),
                ("line_seg", C.c_int * MAX_LINES),
                ("line_nd", (C.c_double * 3) * MAX_LINES)]


def build(force=False):
    so = os.path.join(_HERE, "libneptune_oracle.so")
    src = os.path.join(_HERE, "neptune_oracle.c")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B", "libneptune_oracle.so"],
                              stdout=subprocess.DEVNULL)
    return so


def lib():
    global _LIB
    if _LIB is None:
        so = os.path.join(_HERE, "libneptune_oracle.so")
        if not os.path.exists(so):
            build()
        L = C.CDLL(so)
        L.orc_pos_ctrl_pts.argtypes = [C.POINTER(C.c_double), C.c_double, C.POINTER(C.c_double)]
        L.orc_vel_ctrl_pts.argtypes = [C.POINTER(C.c_double), C.c_double, C.POINTER(C.c_double)]
        L.orc_convex_hull_2d.argtypes = [C.c_int, C.c_void_p, C.c_void_p]
        L.orc_convex_hull_2d.restype = C.c_int
        L.orc_hull_of_interval.argtypes = [C.POINTER(abi.nep_pwp), C.c_double, C.c_double,
                                           C.c_double, C.POINTER(C.c_double), C.c_void_p,
                                           C.POINTER(C.c_int), C.c_void_p, C.POINTER(C.c_int)]
        L.orc_hull_of_interval.restype = C.c_int
        L.orc_inflate_static.argtypes = [C.c_int, C.c_void_p, C.c_double, C.c_void_p]
        L.orc_inflate_static.restype = C.c_int
        for f in (L.orc_separator, L.orc_separator_simplex, L.orc_separator_ordered):
            f.argtypes = [C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.POINTER(C.c_double)]
            f.restype = C.c_int
        L.orc_separator_glpk_class.argtypes = [C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_int)]
        L.orc_separator_glpk_class.restype = C.c_int
        L.orc_set_separator_rule.argtypes = [C.c_int]; L.orc_set_separator_rule.restype = None
        L.orc_set_qp_tolerances.argtypes = [C.c_double, C.c_double]; L.orc_set_qp_tolerances.restype = None
        L.orc_set_polish.argtypes = [C.c_int]; L.orc_set_polish.restype = None
        L.orc_pass_stats.argtypes = [C.POINTER(C.c_long), C.POINTER(C.c_long)]; L.orc_pass_stats.restype = None
        L.orc_last_polished.argtypes = []; L.orc_last_polished.restype = C.c_int
        L.orc_optimize.argtypes = [C.POINTER(orc_params), C.c_int, C.c_void_p, C.c_int,
                                   C.POINTER(orc_polys), C.POINTER(orc_polys), C.POINTER(orc_ent),
                                   C.c_int, C.c_void_p, C.c_void_p, C.POINTER(orc_result)]
        L.orc_optimize.restype = C.c_int
        L.orc_sample.argtypes = [C.c_int, C.c_void_p, C.c_double, C.c_double, C.c_void_p, C.c_int]
        L.orc_sample.restype = C.c_int
        L.orc_replan.argtypes = [C.POINTER(orc_params), C.c_double, C.c_int, C.c_void_p,
                                 C.c_void_p, C.POINTER(orc_polys), C.c_void_p,
                                 C.POINTER(orc_result), C.c_void_p, C.c_void_p]
        L.orc_replan.restype = C.c_int
        L.orc_gjk_collision.argtypes = [C.c_int, C.c_void_p, C.c_int, C.c_void_p]
        L.orc_gjk_collision.restype = C.c_int
        L.orc_trajs_and_pwp_in_collision.argtypes = [C.c_void_p, C.c_void_p, C.c_double, C.c_double]
        L.orc_trajs_and_pwp_in_collision.restype = C.c_int
        L.orc_safety_resolve.argtypes = [C.c_int, C.c_void_p, C.c_double, C.c_double, C.c_double, C.c_void_p, C.c_void_p]
        L.orc_safety_resolve.restype = None
        L.orc_safety_resolve_prev.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_double, C.c_double, C.c_double, C.c_void_p, C.c_void_p]
        L.orc_safety_resolve_prev.restype = None
        L.orc_fe_set_dump.argtypes = [C.c_void_p, C.c_int]; L.orc_fe_set_dump.restype = None
        _LIB = L
    return _LIB


def _c(a, dt=np.float64):
    return np.ascontiguousarray(a, dtype=dt)


def pos_ctrl_pts(P, T):
    P = _c(P); Q = np.zeros(4)
    lib().orc_pos_ctrl_pts(abi.dptr(P), float(T), abi.dptr(Q))
    return Q


def vel_ctrl_pts(P, T):
    P = _c(P); Q = np.zeros(3)
    lib().orc_vel_ctrl_pts(abi.dptr(P), float(T), abi.dptr(Q))
    return Q


def convex_hull_2d(pts):
    pts = _c(pts).reshape(-1, 2)
    out = np.zeros((max(len(pts), 1) + 2, 2))
    k = lib().orc_convex_hull_2d(len(pts), pts.ctypes.data, out.ctypes.data)
    return out[:k].copy()


def hull_of_interval(pwp, t0, t1, T_span, delta, with_overflow=False):
    d = _c(delta)
    h = np.zeros((abi.NEP_HULL_MAX_V, 2)); h0 = np.zeros((abi.NEP_HULL_MAX_V, 2))
    nv = C.c_int(0); nv0 = C.c_int(0)
    ov = lib().orc_hull_of_interval(C.byref(pwp), t0, t1, T_span, abi.dptr(d), h.ctypes.data,
                                    C.byref(nv), h0.ctypes.data, C.byref(nv0))
    if with_overflow:
        return h[:nv.value].copy(), h0[:nv0.value].copy(), bool(ov)
    return h[:nv.value].copy(), h0[:nv0.value].copy()


def inflate_static(verts, safe_dist):
    v = _c(verts).reshape(-1, 2)
    out = np.zeros((4 * len(v) + 2, 2))
    k = lib().orc_inflate_static(len(v), v.ctypes.data, float(safe_dist), out.ctypes.data)
    return out[:k].copy()


def separator(A, B, simplex=False, ordered=False):
    A = _c(A).reshape(-1, 2); B = _c(B).reshape(-1, 2)
    nd = np.zeros(3)
    f = lib().orc_separator_simplex if simplex else (lib().orc_separator_ordered if ordered else lib().orc_separator)
    ok = f(len(A), A.ctypes.data, len(B), B.ctypes.data, abi.dptr(nd))
    return bool(ok), nd


def separator_glpk_class(A, B):
    """the vertex a primal simplex of GLPK's default class reaches (orc_separator_glpk_class) -> (ok, nd, pivots)"""
    A = _c(A).reshape(-1, 2); B = _c(B).reshape(-1, 2)
    nd = np.zeros(3); n = C.c_int(0)
    ok = lib().orc_separator_glpk_class(len(A), A.ctypes.data, len(B), B.ctypes.data, abi.dptr(nd), C.byref(n))
    return bool(ok), nd, n.value


def set_separator_rule(rule):
    """0: the largest-gap vertex (default); 1: the GLPK-class simplex's vertex — for every separator call of the restated path
    made from this thread afterwards (checker of nep_batch_set_separator_rule)"""
    lib().orc_set_separator_rule(int(rule))


def pass_stats():
    """(interior-point iterations, discarded predictors) since the last call: a device solve's passes are their sum"""
    a, b = C.c_long(0), C.c_long(0)
    lib().orc_pass_stats(C.byref(a), C.byref(b))
    return int(a.value), int(b.value)


def set_polish(on=True):
    """the active-set polish of solves that end without the strict tests (on by default; the product's twin: nep_batch_set_polish)"""
    lib().orc_set_polish(1 if on else 0)


def last_polished():
    return bool(lib().orc_last_polished())


def set_qp_tolerances(residual_tol=1e-10, gap_tol=1e-11):
    """the interior point's strict tests, for every solve made from this thread afterwards (checker of nep_batch_set_tolerances)"""
    lib().orc_set_qp_tolerances(float(residual_tol), float(gap_tol))


class Polys:
    """CSR polygon list; keeps the numpy buffers alive."""

    def __init__(self, polys):
        self.off = np.zeros(len(polys) + 1, dtype=np.int32)
        for i, p in enumerate(polys):
            self.off[i + 1] = self.off[i] + len(p)
        self.xy = _c(np.concatenate([np.asarray(p, dtype=np.float64).reshape(-1, 2) for p in polys])
                     if len(polys) and self.off[-1] > 0 else np.zeros((0, 2)))
        self.c = orc_polys(len(polys), self.off.ctypes.data_as(C.POINTER(C.c_int)), abi.dptr(self.xy))


def make_params(p, agent_id):
    """p: neptune_amd.scene.Params-like object."""
    pb = _c(p.pb)
    par = orc_params(p.num_pol, agent_id, len(pb), p.T_span, p.weight,
                     (C.c_double * 3)(p.x_min, p.y_min, p.z_min),
                     (C.c_double * 3)(p.x_max, p.y_max, p.z_max), p.v_max, p.a_max, abi.dptr(pb))
    par._keep = pb
    return par


def result_dict(res, K):
    co = np.ctypeslib.as_array(res.coeff).copy()[:, :K, :]
    nl = res.n_lines
    return dict(status=res.status, iters=res.iters, iters_first=res.iters_first, n_lines=nl,
                n_lp=res.n_lp, n_lp_failed=res.n_lp_failed, n_rows=res.n_rows,
                qc_active=res.qc_active, objective=res.objective, coeff=co,
                line_seg=np.ctypeslib.as_array(res.line_seg)[:nl].copy(),
                line_nd=np.ctypeslib.as_array(res.line_nd)[:nl].copy())


def optimize(p, agent_id, coeff_init, hulls, statics, ent=None, lines=None):
    """PolySolverGurobi::optimize on host.  coeff_init [3][K][4]; hulls: list over
    (obstacle j, interval i) j-major of (V,2) arrays; statics: list of (V,2) inflated polygons;
    lines: optional (seg[], nd[][3]) override."""
    ci = np.zeros((3, abi.NEP_MAX_POL, 4)); K = coeff_init.shape[1]; ci[:, :K, :] = coeff_init
    par = make_params(p, agent_id)
    H = Polys(hulls); S = Polys(statics)
    n_obst = len(hulls) // p.num_pol
    res = orc_result()
    entc = None
    keep = []
    if ent is not None:
        case_id = _c(ent["case_id"], np.int32)
        bend = ent["bend"]  # list per agent of (nb,2)
        boff = np.zeros(len(bend) + 1, dtype=np.int32)
        for i, b in enumerate(bend):
            boff[i + 1] = boff[i] + len(b)
        bxy = _c(np.concatenate([np.asarray(b, dtype=np.float64).reshape(-1, 2) for b in bend])
                 if boff[-1] > 0 else np.zeros((0, 2)))
        H0 = Polys(ent["hulls_noinfl"])
        entc = orc_ent(1, case_id.ctypes.data_as(C.POINTER(C.c_int)),
                       boff.ctypes.data_as(C.POINTER(C.c_int)), abi.dptr(bxy), C.pointer(H0.c))
        keep += [case_id, boff, bxy, H0]
    if lines is None:
        on, oseg, ond = -1, None, None
    else:
        oseg = _c(lines[0], np.int32); ond = _c(lines[1]).reshape(-1, 3); on = len(oseg)
    lib().orc_optimize(C.byref(par), K, ci.ctypes.data, n_obst, C.byref(H.c), C.byref(S.c),
                       C.byref(entc) if entc is not None else None, on,
                       oseg.ctypes.data if oseg is not None else None,
                       ond.ctypes.data if ond is not None else None, C.byref(res))
    return result_dict(res, K)


def sample(coeff, T_span, dc, cap=512):
    K = coeff.shape[1]
    ci = np.zeros((3, abi.NEP_MAX_POL, 4)); ci[:, :K, :] = coeff
    st = np.zeros((cap, abi.NEP_STATE_DOUBLES))
    n = lib().orc_sample(K, ci.ctypes.data, T_span, dc, st.ctypes.data, cap)
    return st[:n].copy()


def replan(p, agent_id, recs, guess, statics, case_id=None, want_hulls=False):
    """Whole replan of one agent from committed-trajectory records (numpy structured arrays with
    abi.TRAJ_REC_DTYPE / abi.GUESS_DTYPE)."""
    par = make_params(p, agent_id)
    S = Polys(statics)
    recs = np.ascontiguousarray(recs)
    g = np.ascontiguousarray(guess)
    res = orc_result()
    hx = hn = None
    if want_hulls:
        hx = np.zeros((len(recs) * p.num_pol, abi.NEP_HULL_MAX_V, 2)); hn = np.zeros(len(recs) * p.num_pol, dtype=np.int32)
    cid = _c(case_id, np.int32) if case_id is not None else None
    lib().orc_replan(C.byref(par), p.drone_radius, len(recs), recs.ctypes.data, g.ctypes.data,
                     C.byref(S.c), cid.ctypes.data if cid is not None else None, C.byref(res),
                     hx.ctypes.data if want_hulls else None, hn.ctypes.data if want_hulls else None)
    out = result_dict(res, int(np.asarray(g["K"]).reshape(-1)[0]))
    if want_hulls:
        out["hull_xy"] = hx; out["hull_nv"] = hn
    return out


class orc_fe_cfg(C.Structure):
    _fields_ = [("num_pol", C.c_int), ("id", C.c_int), ("num_agents", C.c_int), ("num_samples", C.c_int), ("beam_width", C.c_int), ("pad_hold", C.c_int),
                ("T_span", C.c_double), ("j_max", C.c_double), ("v_max", C.c_double), ("a_max", C.c_double), ("voxel_size", C.c_double),
                ("bias", C.c_double), ("goal_size", C.c_double), ("cable_length", C.c_double), ("mins", C.c_double * 2),
                ("maxs", C.c_double * 2), ("pb", C.POINTER(C.c_double))]


def frontend_beam(p, fe, agent_id, start, hull_xy, hull_nv, statics):
    """The deterministic beam rule of include/neptune_frontend.h for one agent.  fe: abi.nep_fe_cfg; start: one
    FE_START_DTYPE record; hull_xy/hull_nv as replan(..., want_hulls=True) returns them.  -> (guess record, result dict)."""
    pb = _c(p.pb)
    cfg = orc_fe_cfg(p.num_pol, agent_id, p.num_agents, fe.num_samples, fe.beam_width, fe.pad_hold, p.T_span, fe.j_max, p.v_max, p.a_max,
                     fe.voxel_size, fe.bias, fe.goal_size, fe.cable_length, (C.c_double * 2)(p.x_min, p.y_min),
                     (C.c_double * 2)(p.x_max, p.y_max), abi.dptr(pb))
    S = Polys(statics)
    st = np.ascontiguousarray(start, dtype=abi.FE_START_DTYPE).reshape(1)
    hx = _c(hull_xy); hn = _c(hull_nv, np.int32)
    g = np.zeros(1, dtype=abi.GUESS_DTYPE); r = np.zeros(1, dtype=abi.FE_RESULT_DTYPE)
    f = lib().orc_frontend_beam
    f.restype = C.c_int
    rc = f(C.byref(cfg), C.c_void_p(st.ctypes.data), C.c_void_p(hx.ctypes.data), C.c_void_p(hn.ctypes.data), C.byref(S.c),
           C.c_void_p(g.ctypes.data), C.c_void_p(r.ctypes.data))
    if rc:
        raise RuntimeError("orc_frontend_beam: bad configuration")
    return g[0], {k: r[0][k].item() for k in abi.FE_RESULT_DTYPE.names}


class orc_fe_ent(C.Structure):
    _fields_ = [("num_samples", C.c_int), ("n_static", C.c_int), ("static_rep", C.POINTER(C.c_double)), ("static_longest", C.POINTER(C.c_double)),
                ("sampled", C.POINTER(C.c_double)), ("present", C.POINTER(C.c_int)), ("bend_n", C.POINTER(C.c_int)), ("bend_xy", C.POINTER(C.c_double)),
                ("init", C.c_void_p)]


def _fe_ent(p, ent):
    """ent: dict(num_samples, reps [S][2][2], longest [S][2], sampled [N][num_pol][ns+1][2], present [N], bend_n [N],
    bend_xy [N][NEP_MAX_BEND][2], init (one FE_ENT_STATE_DTYPE record or None)) -> (orc_fe_ent, keep-alive list)"""
    reps = _c(ent["reps"]).reshape(-1); lg = _c(ent["longest"]).reshape(-1)
    if reps.size == 0:
        reps = np.zeros(4); lg = np.zeros(2)
    sm = _c(ent["sampled"]); pr = _c(ent["present"], np.int32); bn = _c(ent["bend_n"], np.int32); bx = _c(ent["bend_xy"])
    init = np.ascontiguousarray(ent["init"], dtype=abi.FE_ENT_STATE_DTYPE).reshape(1) if ent.get("init") is not None else None
    E = orc_fe_ent(int(ent["num_samples"]), len(np.asarray(ent["reps"]).reshape(-1, 2, 2)), abi.dptr(reps), abi.dptr(lg), abi.dptr(sm),
                   pr.ctypes.data_as(C.POINTER(C.c_int)), bn.ctypes.data_as(C.POINTER(C.c_int)), abi.dptr(bx),
                   init.ctypes.data if init is not None else None)
    return E, [reps, lg, sm, pr, bn, bx, init]


def frontend_beam_ent(p, fe, agent_id, start, hull_xy, hull_nv, statics, ent):
    """orc_frontend_beam_ent: the beam with the entangle check on -> (guess record, result dict, case_id [NEP_MAX_POL][N])."""
    pb = _c(p.pb)
    cfg = orc_fe_cfg(p.num_pol, agent_id, p.num_agents, fe.num_samples, fe.beam_width, fe.pad_hold, p.T_span, fe.j_max, p.v_max, p.a_max,
                     fe.voxel_size, fe.bias, fe.goal_size, fe.cable_length, (C.c_double * 2)(p.x_min, p.y_min),
                     (C.c_double * 2)(p.x_max, p.y_max), abi.dptr(pb))
    S = Polys(statics)
    st = np.ascontiguousarray(start, dtype=abi.FE_START_DTYPE).reshape(1)
    hx = _c(hull_xy); hn = _c(hull_nv, np.int32)
    g = np.zeros(1, dtype=abi.GUESS_DTYPE); r = np.zeros(1, dtype=abi.FE_RESULT_DTYPE)
    case = np.zeros((abi.NEP_MAX_POL, p.num_agents), dtype=np.int32)
    E, keep = _fe_ent(p, ent)
    f = lib().orc_frontend_beam_ent
    f.restype = C.c_int
    rc = f(C.byref(cfg), C.c_void_p(st.ctypes.data), C.c_void_p(hx.ctypes.data), C.c_void_p(hn.ctypes.data), C.byref(S.c), C.byref(E),
           C.c_void_p(g.ctypes.data), C.c_void_p(r.ctypes.data), C.c_void_p(case.ctypes.data))
    if rc:
        raise RuntimeError("orc_frontend_beam_ent: bad configuration")
    return g[0], {k: r[0][k].item() for k in abi.FE_RESULT_DTYPE.names}, case


def ent_propagate_guess(p, agent_id, cable_length, guess, ent):
    """case ids [NEP_MAX_POL][N] and the first entangling segment (0: none) along a given guess (orc_ent_propagate_guess)."""
    pb = _c(p.pb)
    cfg = orc_fe_cfg(p.num_pol, agent_id, p.num_agents, 5, 1, 0, p.T_span, p.j_max, p.v_max, p.a_max, 0.2, 1.1, 0.2, cable_length,
                     (C.c_double * 2)(p.x_min, p.y_min), (C.c_double * 2)(p.x_max, p.y_max), abi.dptr(pb))
    E, keep = _fe_ent(p, ent)
    g = np.ascontiguousarray(guess, dtype=abi.GUESS_DTYPE).reshape(1)
    case = np.zeros((abi.NEP_MAX_POL, p.num_agents), dtype=np.int32)
    hit = C.c_int(0); na = C.c_int(0)
    lib().orc_ent_propagate_guess(C.byref(cfg), C.byref(E), C.c_void_p(g.ctypes.data), C.c_void_p(case.ctypes.data), C.byref(hit), C.byref(na))
    return case, hit.value, na.value


def entangle_check_pwp(p, agent_id, cable_length, coeff_x0, coeff_y0, ent):
    """KinodynamicSearch::entangleCheckGivenPwp on the first interval [a b c d] of a new trajectory."""
    pb = _c(p.pb)
    cfg = orc_fe_cfg(p.num_pol, agent_id, p.num_agents, 5, 1, 0, p.T_span, p.j_max, p.v_max, p.a_max, 0.2, 1.1, 0.2, cable_length,
                     (C.c_double * 2)(p.x_min, p.y_min), (C.c_double * 2)(p.x_max, p.y_max), abi.dptr(pb))
    E, keep = _fe_ent(p, ent)
    cx = _c(coeff_x0); cy = _c(coeff_y0)
    f = lib().orc_entangle_check_pwp
    f.restype = C.c_int
    return bool(f(C.byref(cfg), C.byref(E), abi.dptr(cx), abi.dptr(cy)))


def frontend_astar(p, fe, agent_id, start, hull_xy, hull_nv, statics, order=None, max_pops=20000):
    """KinodynamicSearch::run restated (best-first search to the goal, lattice order and pop budget as parameters)."""
    pb = _c(p.pb)
    cfg = orc_fe_cfg(p.num_pol, agent_id, p.num_agents, fe.num_samples, fe.beam_width, fe.pad_hold, p.T_span, fe.j_max, p.v_max, p.a_max,
                     fe.voxel_size, fe.bias, fe.goal_size, fe.cable_length, (C.c_double * 2)(p.x_min, p.y_min),
                     (C.c_double * 2)(p.x_max, p.y_max), abi.dptr(pb))
    S = Polys(statics)
    st = np.ascontiguousarray(start, dtype=abi.FE_START_DTYPE).reshape(1)
    hx = _c(hull_xy); hn = _c(hull_nv, np.int32)
    od = _c(order, np.int32) if order is not None else None
    g = np.zeros(1, dtype=abi.GUESS_DTYPE); r = np.zeros(1, dtype=abi.FE_RESULT_DTYPE)
    f = lib().orc_frontend_astar
    f.restype = C.c_int
    rc = f(C.byref(cfg), C.c_void_p(st.ctypes.data), C.c_void_p(hx.ctypes.data), C.c_void_p(hn.ctypes.data), C.byref(S.c),
           C.c_void_p(od.ctypes.data) if od is not None else None, C.c_int(max_pops), C.c_void_p(g.ctypes.data), C.c_void_p(r.ctypes.data))
    if rc:
        raise RuntimeError("orc_frontend_astar: bad configuration")
    return g[0], {k: r[0][k].item() for k in abi.FE_RESULT_DTYPE.names}


def hulls_of_scene(p, agent_id, recs, t_start, statics):
    """Interval hulls of every record as the oracle builds them for a replan at t_start:
    (hull_xy [N][num_pol][16][2], hull_nv [N][num_pol])."""
    g = np.zeros(1, dtype=abi.GUESS_DTYPE)
    g["K"] = 1; g["t_start"] = t_start
    out = replan(p, agent_id, recs, g[0], statics, want_hulls=True)
    # orc_replan lists the hulls in obstacle order (ids 1..N that are present, own id skipped): back to id order
    N = p.num_agents
    cx = out["hull_xy"].reshape(-1, p.num_pol, abi.NEP_HULL_MAX_V, 2); cn = out["hull_nv"].reshape(-1, p.num_pol)
    hx = np.zeros((N, p.num_pol, abi.NEP_HULL_MAX_V, 2)); hn = np.zeros((N, p.num_pol), dtype=np.int32)
    recs = np.asarray(recs)
    k = 0
    for aid in range(1, N + 1):
        m = [r for r in recs if int(r["id"]) == aid and r["valid"] and r["is_agent"]]
        if aid == agent_id or not m:
            continue
        hx[aid - 1] = cx[k]; hn[aid - 1] = cn[k]; k += 1
    return hx, hn


def gjk_collision(V1, V2):
    """gjk::collision(vertices1, vertices2)."""
    V1 = _c(V1).reshape(-1, 2); V2 = _c(V2).reshape(-1, 2)
    return bool(lib().orc_gjk_collision(len(V1), V1.ctypes.data, len(V2), V2.ctypes.data))


def safety_resolve(fresh, t_start, T_span, drone_radius):
    """fresh: [N] TRAJ_REC_DTYPE (every agent's new trajectory).  Returns (conflict [N][N] uint8, accept [N] int32)."""
    fresh = np.ascontiguousarray(fresh)
    n = len(fresh)
    conflict = np.zeros((n, n), dtype=np.uint8); accept = np.zeros(n, dtype=np.int32)
    lib().orc_safety_resolve(n, fresh.ctypes.data, t_start, T_span, drone_radius, conflict.ctypes.data, accept.ctypes.data)
    return conflict, accept


def safety_resolve_prev(prev, fresh, t_start, T_span, drone_radius):
    """safety_resolve with nep_batch_set_safety_check_prev: new trajectories must also clear the previous records."""
    prev = np.ascontiguousarray(prev); fresh = np.ascontiguousarray(fresh)
    n = len(fresh)
    conflict = np.zeros((n, n), dtype=np.uint8); accept = np.zeros(n, dtype=np.int32)
    lib().orc_safety_resolve_prev(n, prev.ctypes.data, fresh.ctypes.data, t_start, T_span, drone_radius, conflict.ctypes.data, accept.ctypes.data)
    return conflict, accept


def trajs_and_pwp_in_collision(other_rec, mine_rec, T_span, drone_radius):
    """Neptune::trajsAndPwpAreInCollision2d(other, mine.pwp, mine.times.front(), mine.times.back())."""
    o = np.ascontiguousarray(other_rec); m = np.ascontiguousarray(mine_rec)
    pw = abi.nep_pwp.from_buffer_copy(m["pwp"].tobytes())
    return bool(lib().orc_trajs_and_pwp_in_collision(o.ctypes.data, C.addressof(pw), T_span, drone_radius))
