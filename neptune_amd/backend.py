"""Host-side mirror of the reference's back-end interface over the C ABI.

`PolySolver` carries the public methods of the reference's `class PolySolverGurobi`
(reference neptune/include/solver_gurobi_poly.hpp:28-49) with the same names, argument meaning,
call order and failure behaviour (reference neptune/src/neptune.cpp:102-107,1514-1527), on numpy
data instead of Eigen/ROS types.  `BatchBackend` drives all local agents of a node per launch
(device-resident records; torch only supplies device memory and streams).
"""
import ctypes as C

import numpy as np

from . import abi
from ._lib import BackendError, check, lib


def _csr(polys):
    off = np.zeros(len(polys) + 1, dtype=np.int32)
    for k, p in enumerate(polys):
        off[k + 1] = off[k] + len(p)
    xy = (np.concatenate([np.asarray(p, dtype=np.float64).reshape(-1, 2) for p in polys])
          if len(polys) and off[-1] > 0 else np.zeros((0, 2)))
    return off, np.ascontiguousarray(xy, dtype=np.float64)


def make_pwp(times, coeff):
    """times [K+1], coeff [3][K][4] -> abi.nep_pwp (mt::PieceWisePol)."""
    p = abi.nep_pwp()
    K = coeff.shape[1]
    p.n_seg = K
    for i in range(K + 1):
        p.times[i] = float(times[i])
    arr = np.ctypeslib.as_array(p.coeff)
    arr[:, :K, :] = coeff
    return p


class PolySolver:
    """Drop-in for PolySolverGurobi (solver_gurobi_poly.hpp:25-49)."""

    def __init__(self, num_pol, deg_pol, id, T_span, pb, weight_term, rad_term, use_linear_constraints):
        self._pb = np.ascontiguousarray(pb, dtype=np.float64).reshape(-1, 2)
        cfg = abi.nep_backend_cfg(num_pol, deg_pol, id, len(self._pb), T_span, weight_term, rad_term,
                                  1 if use_linear_constraints else 0, 0, abi.dptr(self._pb))
        self._h = lib().nep_backend_create(C.byref(cfg))
        if not self._h:
            raise BackendError(lib().nep_last_error().decode())
        self.num_pol = num_pol
        self.T_span = T_span
        self.num_agents = len(self._pb)

    def close(self):
        if getattr(self, "_h", None):
            lib().nep_backend_destroy(self._h)
            self._h = None

    __del__ = close

    # ---- setters (same names as the reference) -----------------------------------------------
    def setMaxValues(self, x_min, x_max, y_min, y_max, z_min, z_max, v_max, a_max, j_max):
        check(lib().nep_backend_set_max_values(self._h, x_min, x_max, y_min, y_max, z_min, z_max, v_max, a_max, j_max))

    def setSeparatorRule(self, rule):
        """not in the reference: which LP vertex the separator returns (nep_backend_set_separator_rule)"""
        check(lib().nep_backend_set_separator_rule(self._h, int(rule)))

    def setTolerances(self, residual_tol=1e-10, gap_tol=1e-11):
        """not in the reference (which leaves Gurobi's defaults, 1e-6 / 1e-8): the interior point's strict tests (nep_backend_set_tolerances)"""
        check(lib().nep_backend_set_tolerances(self._h, float(residual_tol), float(gap_tol)))

    def setPolish(self, on=True):
        """not in the reference: the active-set polish of solves that end without the strict tests (nep_backend_set_polish; on by default,
        under the line presolve as well; 3: every-row solves only)"""
        check(lib().nep_backend_set_polish(self._h, int(on) if (on is not True and on is not False) else (1 if on else 0)))

    def setLineCull(self, radius):
        """not in the reference: the verified line presolve's radius (nep_backend_set_line_cull; 4 m by default, 0 = every row through the
        interior point, lines in the reference's call order)"""
        check(lib().nep_backend_set_line_cull(self._h, float(radius)))

    def debug_option(self, name, value):
        """development aid (include/neptune_backend_debug.h: nep_backend_debug_set_option)"""
        check(lib().nep_backend_debug_set_option(self._h, name.encode(), int(value)))

    def setMaxRuntime(self, runtime):
        check(lib().nep_backend_set_max_runtime(self._h, runtime))

    def setTetherLength(self, tether_length):
        check(lib().nep_backend_set_tether_length(self._h, tether_length))

    def setStaticObstVert(self, convex_hulls_of_static_obs):
        off, xy = _csr(convex_hulls_of_static_obs)
        check(lib().nep_backend_set_static_obst_vert(self._h, len(convex_hulls_of_static_obs), abi.iptr(off), abi.dptr(xy)))

    def setInitTrajectory(self, times, coeff):
        """pwp_init: times [K+1], coeff [3][K][4] ([a b c d] per interval, seconds)."""
        self._K = coeff.shape[1]
        p = make_pwp(times, np.asarray(coeff, dtype=np.float64))
        check(lib().nep_backend_set_init_trajectory(self._h, C.byref(p)))

    def setHulls(self, hulls):
        """hulls[j][i]: (V,2) vertices, j over the other agents present, i < num_pol."""
        flat = [h for obs in hulls for h in obs]
        off, xy = _csr(flat)
        check(lib().nep_backend_set_hulls(self._h, len(hulls), abi.iptr(off), abi.dptr(xy)))

    def setHullsNoInflation(self, hulls):
        """hulls[agent_id-1][i]; empty lists for self / unknown agents."""
        flat = []
        for obs in hulls:
            obs = list(obs) + [np.zeros((0, 2))] * (self.num_pol - len(obs))
            flat += obs[:self.num_pol]
        off, xy = _csr(flat)
        check(lib().nep_backend_set_hulls_no_inflation(self._h, len(hulls), abi.iptr(off), abi.dptr(xy)))

    def setBetasVector(self, vec_of_agents):
        """Dead in the reference (solver_gurobi_poly.cpp:290-305); accepted and ignored."""

    def setEntStateVector(self, ent_state_vec, bend_pts_for_agents):
        """ent_state_vec: list (K+1) of dicts {alphas: [(agent_id, case)], active_cases: [..]};
        bend_pts_for_agents: list per agent of (nb,2)."""
        if ent_state_vec is None:
            check(lib().nep_backend_set_ent_state_vector(self._h, None))
            return
        ns = len(ent_state_vec)
        na = max(len(e["active_cases"]) for e in ent_state_vec)
        aoff = np.zeros(ns + 1, dtype=np.int32)
        al = []
        act = np.zeros((ns, na), dtype=np.int32)
        for k, e in enumerate(ent_state_vec):
            al += [tuple(a) for a in e["alphas"]]
            aoff[k + 1] = len(al)
            act[k, :len(e["active_cases"])] = e["active_cases"]
        alphas = np.ascontiguousarray(np.array(al, dtype=np.int32).reshape(-1, 2))
        boff, bxy = _csr(bend_pts_for_agents)
        v = abi.nep_ent_view(ns, na, abi.iptr(aoff), abi.iptr(alphas), abi.iptr(act), abi.iptr(boff), abi.dptr(bxy))
        check(lib().nep_backend_set_ent_state_vector(self._h, C.byref(v)))

    # ---- solve ---------------------------------------------------------------------------------
    def optimize(self):
        """Returns (success, objective_value).  objective_value is None when the solve failed
        (the reference leaves its out-parameter untouched, solver_gurobi_poly.cpp:856-859)."""
        obj = C.c_double(float("nan"))
        st = check(lib().nep_backend_optimize(self._h, C.byref(obj)))
        self.status = st
        return st != abi.NEP_FAILED, (obj.value if st != abi.NEP_FAILED else None)

    def generatePwpOut(self, t_start, dc):
        """Returns (times [K+1], coeff [3][K][4], traj_out [n][12] = pos,vel,accel,jerk)."""
        p = abi.nep_pwp()
        cap = int(np.ceil(self.num_pol * self.T_span / dc)) + 3
        st = np.zeros((cap, abi.NEP_STATE_DOUBLES))
        n = C.c_int32(0)
        check(lib().nep_backend_generate_pwp_out(self._h, t_start, dc, C.byref(p), abi.dptr(st), cap, C.byref(n)))
        K = p.n_seg
        return (np.array(p.times[:K + 1]), np.ctypeslib.as_array(p.coeff)[:, :K, :].copy(), st[:n.value].copy())

    def stats(self):
        s = abi.nep_stats()
        check(lib().nep_backend_get_stats(self._h, C.byref(s)))
        return {f: getattr(s, f) for f, _ in abi.nep_stats._fields_}

    def timeSequence(self, times, coeff, hulls, hulls_no_inflation=None, t_start=0.0, dc=0.05, n_iter=100):
        """measurement aid (nep_backend_debug_time_sequence): the six-call drop-in sequence of one replan n_iter times inside the
        library, no Python between the calls -> (status, wall us per sequence [n_iter], wall us of optimize() alone [n_iter])"""
        pw = make_pwp(times, np.asarray(coeff, dtype=np.float64))
        off, xy = _csr([h for obs in hulls for h in obs])
        if hulls_no_inflation is not None:
            flat = []
            for obs in hulls_no_inflation:
                obs = list(obs) + [np.zeros((0, 2))] * (self.num_pol - len(obs))
                flat += obs[:self.num_pol]
            off0, xy0 = _csr(flat)
        us = np.zeros(n_iter); uo = np.zeros(n_iter)
        st = check(lib().nep_backend_debug_time_sequence(self._h, C.byref(pw), len(hulls), abi.iptr(off), abi.dptr(xy),
                                                         abi.iptr(off0) if hulls_no_inflation is not None else None,
                                                         abi.dptr(xy0) if hulls_no_inflation is not None else None, None,
                                                         float(t_start), float(dc), int(n_iter), abi.dptr(us), abi.dptr(uo)))
        return st, us, uo

    # ---- test hooks ----------------------------------------------------------------------------
    def debugSetLines(self, seg, nd):
        if seg is None:
            check(lib().nep_backend_debug_set_lines(self._h, -1, None, None))
            return
        seg = np.ascontiguousarray(seg, dtype=np.int32); nd = np.ascontiguousarray(nd, dtype=np.float64).reshape(-1, 3)
        check(lib().nep_backend_debug_set_lines(self._h, len(seg), abi.iptr(seg), abi.dptr(nd)))

    def debugGetLines(self, cap=8192):
        seg = np.zeros(cap, dtype=np.int32); nd = np.zeros((cap, 3)); n = C.c_int32(0)
        check(lib().nep_backend_debug_get_lines(self._h, cap, abi.iptr(seg), abi.dptr(nd), C.byref(n)))
        return seg[:n.value].copy(), nd[:n.value].copy()


def separator_batch(As, Bs, rule=0):
    """Batched separator::Separator::solveModel (2-D).  As/Bs: lists of (n,2) arrays.  rule: which LP vertex is returned
    (0 largest gap, 1 GLPK-class simplex: nep_batch_set_separator_rule)."""
    aoff, axy = _csr(As); boff, bxy = _csr(Bs)
    n = len(As)
    nd = np.zeros((n, 3)); ok = np.zeros(n, dtype=np.int32)
    check(lib().nep_separator_batch_rule(int(rule), n, abi.iptr(aoff), abi.dptr(axy), abi.iptr(boff), abi.dptr(bxy), abi.dptr(nd), abi.iptr(ok)))
    return ok.astype(bool), nd


def gjk_batch(polys, quads):
    """Batched gjk::collision(polygon, four points).  polys: list of (n,2); quads: (len(polys), 4, 2)."""
    aoff, axy = _csr(polys)
    q = np.ascontiguousarray(quads, dtype=np.float64).reshape(len(polys), 4, 2)
    hit = np.zeros(len(polys), dtype=np.int32)
    check(lib().nep_gjk_batch(len(polys), abi.iptr(aoff), abi.dptr(axy), abi.dptr(q), abi.iptr(hit)))
    return hit.astype(bool)


def hulls_batch(recs, t_start, num_pol, T_span, drone_radius):
    """Neptune::convexHullsOfCurve2d for committed-trajectory records (TRAJ_REC_DTYPE array)."""
    recs = np.ascontiguousarray(recs)
    n = len(recs)
    hx = np.zeros((n, num_pol, abi.NEP_HULL_MAX_V, 2)); hn = np.zeros((n, num_pol), dtype=np.int32)
    h0 = np.zeros((n, num_pol, abi.NEP_HULL_MAX_V, 2)); n0 = np.zeros((n, num_pol), dtype=np.int32)
    check(lib().nep_hulls_batch(n, recs.ctypes.data, t_start, num_pol, T_span, drone_radius, abi.dptr(hx), abi.iptr(hn), abi.dptr(h0), abi.iptr(n0)))
    return hx, hn, h0, n0


class BatchBackend:
    """All local agents of `n_scenes` scenes per launch sequence (nep_batch_* entry points)."""

    def __init__(self, par, statics, first_local=0, n_local=None, n_scenes=1, device=None):
        import torch
        self.torch = torch
        self.par = par
        N = par.num_agents
        n_local = N - first_local if n_local is None else n_local
        self.N, self.first_local, self.n_local, self.n_scenes = N, first_local, n_local, n_scenes
        self.device = device if device is not None else torch.device("cuda", torch.cuda.current_device())
        self._pb = np.ascontiguousarray(par.pb, dtype=np.float64)
        soff, sxy = _csr(statics)
        cfg = abi.nep_batch_cfg(N, first_local, n_local, par.num_pol, len(statics), 1 if par.enable_entangle else 0,
                                par.max_states, n_scenes, par.T_span, par.weight, par.dc, par.drone_radius,
                                par.x_min, par.x_max, par.y_min, par.y_max, par.z_min, par.z_max, par.v_max, par.a_max,
                                abi.dptr(self._pb), abi.iptr(soff), abi.dptr(sxy))
        with torch.cuda.device(self.device):
            self._h = lib().nep_batch_create(C.byref(cfg))
        if not self._h:
            raise BackendError(lib().nep_last_error().decode())
        self.slots = n_scenes * n_local
        S = self.slots
        self.d_solution = torch.zeros(S * abi.SOLUTION_DTYPE.itemsize, dtype=torch.uint8, device=self.device)
        self.d_states = torch.zeros(S * par.max_states * abi.NEP_STATE_DOUBLES, dtype=torch.float64, device=self.device)
        self.d_commit = torch.zeros(S * abi.TRAJ_REC_DTYPE.itemsize, dtype=torch.uint8, device=self.device)

    def close(self):
        if getattr(self, "_h", None):
            try:
                lib().nep_batch_destroy(self._h)
            except TypeError:          # (interpreter shutdown: the module globals are gone)
                pass
            self._h = None

    __del__ = close

    def to_device(self, arr):
        """numpy structured array -> device byte tensor."""
        a = np.ascontiguousarray(arr)
        return self.torch.from_numpy(a.view(np.uint8).reshape(-1).copy()).to(self.device)

    def replan(self, d_committed, d_guess, d_ent=None, stream=None, want_commit=True):
        """Enqueues one replan of every slot; tensors are device byte tensors."""
        torch = self.torch
        st = stream if stream is not None else torch.cuda.current_stream(self.device)
        check(lib().nep_batch_replan(self._h, d_committed.data_ptr() if d_committed is not None else None, d_guess.data_ptr(),
                                     d_ent.data_ptr() if d_ent is not None else None,
                                     self.d_solution.data_ptr(), self.d_states.data_ptr(),
                                     self.d_commit.data_ptr() if want_commit else None, st.cuda_stream))

    def replan_lines(self, d_committed, d_guess, d_ent=None, stream=None):
        """first half of replan(): interval hulls + separating lines into the handle's scratch (nep_batch_replan_lines)"""
        st = stream if stream is not None else self.torch.cuda.current_stream(self.device)
        check(lib().nep_batch_replan_lines(self._h, d_committed.data_ptr(), d_guess.data_ptr(),
                                           d_ent.data_ptr() if d_ent is not None else None, st.cuda_stream))

    def replan_solve(self, d_committed, d_guess, d_ent=None, stream=None, want_commit=True):
        """second half of replan(): the QPs on the lines replan_lines left (nep_batch_replan_solve)"""
        st = stream if stream is not None else self.torch.cuda.current_stream(self.device)
        check(lib().nep_batch_replan_solve(self._h, d_committed.data_ptr(), d_guess.data_ptr(),
                                           d_ent.data_ptr() if d_ent is not None else None,
                                           self.d_solution.data_ptr(), self.d_states.data_ptr(),
                                           self.d_commit.data_ptr() if want_commit else None, st.cuda_stream))

    # ---- sharded hulls (multi-GPU rounds): hulls of the local agents -> all-gather -> replan -------
    def hull_block_bytes(self):
        return int(lib().nep_batch_hull_block_bytes(self._h))

    def hulls(self, d_committed_local, d_guess, d_block, stream=None):
        """Interval hulls of the local agents' committed trajectories ([S][n_local] records) into d_block."""
        st = stream if stream is not None else self.torch.cuda.current_stream(self.device)
        check(lib().nep_batch_hulls(self._h, d_committed_local.data_ptr(), d_guess.data_ptr(), d_block.data_ptr(), st.cuda_stream))

    def replan_hulls(self, d_blocks, d_guess, d_ent=None, stream=None, want_commit=True):
        """Separator + QP of every slot against the gathered hull blocks of all ranks."""
        st = stream if stream is not None else self.torch.cuda.current_stream(self.device)
        n_blocks = self.N // self.n_local
        check(lib().nep_batch_replan_hulls(self._h, d_blocks.data_ptr(), n_blocks, d_guess.data_ptr(),
                                           d_ent.data_ptr() if d_ent is not None else None,
                                           self.d_solution.data_ptr(), self.d_states.data_ptr(),
                                           self.d_commit.data_ptr() if want_commit else None, st.cuda_stream))

    # ---- front end: hulls -> beam search over the jerk lattice -> guesses -------------
    def frontend(self, fe_cfg, d_committed, d_start, d_guess, d_result=None, stream=None):
        """fe_cfg: abi.nep_fe_cfg; d_start: device bytes of [slots] FE_START_DTYPE; d_guess (out): [slots] GUESS_DTYPE."""
        st = stream if stream is not None else self.torch.cuda.current_stream(self.device)
        check(lib().nep_batch_frontend(self._h, C.byref(fe_cfg), d_committed.data_ptr(), d_start.data_ptr(), d_guess.data_ptr(),
                                       d_result.data_ptr() if d_result is not None else None, st.cuda_stream))

    def frontend_hulls(self, fe_cfg, d_blocks, d_start, d_guess, d_result=None, stream=None):
        """the front end against all-gathered hull blocks (see hulls / replan_hulls)"""
        st = stream if stream is not None else self.torch.cuda.current_stream(self.device)
        check(lib().nep_batch_frontend_hulls(self._h, C.byref(fe_cfg), d_blocks.data_ptr(), self.N // self.n_local, d_start.data_ptr(),
                                             d_guess.data_ptr(), d_result.data_ptr() if d_result is not None else None, st.cuda_stream))

    # ---- entangle check on (include/neptune_frontend.h) ------------------------------------------------------------
    def set_static_reps(self, reps, longest, scene=-1):
        """staticObsRep_ [S][2][2] and staticObsLongestDist_ [S][2] (nep_batch_set_static_reps)"""
        r = np.ascontiguousarray(reps, dtype=np.float64).reshape(-1); l = np.ascontiguousarray(longest, dtype=np.float64).reshape(-1)
        if r.size == 0:
            r = np.zeros(4); l = np.zeros(2)
        check(lib().nep_batch_set_static_reps(self._h, scene, abi.dptr(r), abi.dptr(l)))

    def frontend_ent(self, fe_cfg, d_committed, d_start, d_guess, d_result=None, d_case_out=None, d_ent_init=None, stream=None):
        """the front end with per-node entangle states: guesses and the dense case block [slots][8][N] (int32) of nep_batch_replan"""
        st = stream if stream is not None else self.torch.cuda.current_stream(self.device)
        check(lib().nep_batch_frontend_ent(self._h, C.byref(fe_cfg), d_committed.data_ptr(), d_start.data_ptr(),
                                           d_ent_init.data_ptr() if d_ent_init is not None else None, d_guess.data_ptr(),
                                           d_result.data_ptr() if d_result is not None else None,
                                           d_case_out.data_ptr() if d_case_out is not None else None, st.cuda_stream))

    def frontend_ent_hulls(self, fe_cfg, d_blocks, d_start, d_guess, d_result=None, d_case_out=None, d_ent_init=None, stream=None):
        """frontend_ent against all-gathered hull blocks (sharded handle created with enable_entangle): nep_batch_frontend_ent_hulls"""
        st = stream if stream is not None else self.torch.cuda.current_stream(self.device)
        check(lib().nep_batch_frontend_ent_hulls(self._h, C.byref(fe_cfg), d_blocks.data_ptr(), self.N // self.n_local, d_start.data_ptr(),
                                                 d_ent_init.data_ptr() if d_ent_init is not None else None, d_guess.data_ptr(),
                                                 d_result.data_ptr() if d_result is not None else None,
                                                 d_case_out.data_ptr() if d_case_out is not None else None, st.cuda_stream))

    def set_ent_samples(self, ns):
        """num_sample_per_interval the hull blocks reserve room for (nep_batch_set_ent_samples)"""
        check(lib().nep_batch_set_ent_samples(self._h, int(ns)))

    def safety_commit_ent(self, d_prev, d_new, d_guess, d_final, d_accept=None, d_ent_init=None, ent_samples=3, cable_length=None, stream=None):
        """safety_commit plus entangleCheckGivenPwp on every new trajectory (nep_batch_safety_commit_ent)"""
        st = stream if stream is not None else self.torch.cuda.current_stream(self.device)
        cable = self.par.tether_length if cable_length is None else cable_length
        check(lib().nep_batch_safety_commit_ent(self._h, d_prev.data_ptr(), d_new.data_ptr(), d_guess.data_ptr(),
                                                d_ent_init.data_ptr() if d_ent_init is not None else None, ent_samples, float(cable),
                                                d_final.data_ptr(), d_accept.data_ptr() if d_accept is not None else None, st.cuda_stream))

    def next_starts(self, d_records, dt, d_start, d_alt_goal=None, switch_radius=0.0, stream=None):
        """point A of the next round on the device: d_start's clock advances by dt and its state becomes that of the committed
        trajectories d_records at the new time; with d_alt_goal ([slots][3] float64) arrived agents swap goals
        (nep_batch_next_starts)"""
        st = stream if stream is not None else self.torch.cuda.current_stream(self.device)
        check(lib().nep_batch_next_starts(self._h, d_records.data_ptr(), float(dt), d_start.data_ptr(),
                                          d_alt_goal.data_ptr() if d_alt_goal is not None else None, float(switch_radius), st.cuda_stream))

    def safety_commit(self, d_prev, d_new, d_guess, d_final, d_accept=None, stream=None):
        """Post-solve safety check + commit (nep_batch_safety_commit); tensors are device byte/int32 tensors."""
        st = stream if stream is not None else self.torch.cuda.current_stream(self.device)
        check(lib().nep_batch_safety_commit(self._h, d_prev.data_ptr(), d_new.data_ptr(), d_guess.data_ptr(), d_final.data_ptr(),
                                            d_accept.data_ptr() if d_accept is not None else None, st.cuda_stream))

    def set_scene_statics(self, scene, statics):
        """scene `scene` gets its own static obstacles (same count as the handle's set; nep_batch_set_scene_statics)"""
        soff, sxy = _csr(statics)
        check(lib().nep_batch_set_scene_statics(self._h, scene, len(statics), abi.iptr(soff), abi.dptr(sxy)))

    def check(self, stream=None):
        """waits for the stream and raises on a capacity overflow met by the kernels (nep_batch_check)"""
        st = stream if stream is not None else self.torch.cuda.current_stream(self.device)
        check(lib().nep_batch_check(self._h, st.cuda_stream))

    def qp_kernel_name(self):
        """the interior-point kernel this handle launches (nep_batch_qp_placement)"""
        return "qp_reg_kernel" if lib().nep_batch_qp_placement(self._h) == 1 else "qp_kernel"

    def set_hull_kernel(self, mode=0):
        """0: by batch size, 1: one hull per wave, 2: eight hulls per wave (nep_batch_set_hull_kernel)"""
        check(lib().nep_batch_set_hull_kernel(self._h, int(mode)))

    def set_launch_order(self, enable=True):
        """QP workgroups longest-expected-first (default) or in slot order: nep_batch_set_launch_order"""
        check(lib().nep_batch_set_launch_order(self._h, 1 if enable else 0))

    def launch_order(self):
        """the workgroup -> slot order of the last replan, or None when it ran in slot order (nep_batch_debug_launch_order)"""
        out = np.zeros(self.n_scenes * self.n_local, dtype=np.int32); n = C.c_int32(0)
        check(lib().nep_batch_debug_launch_order(self._h, abi.iptr(out), len(out), C.byref(n)))
        return out[:n.value].copy() if n.value else None

    def set_max_runtime(self, seconds):
        """wall-clock budget of one solve (Gurobi TimeLimit; 0 = off): nep_batch_set_max_runtime"""
        check(lib().nep_batch_set_max_runtime(self._h, float(seconds)))

    def set_line_cull(self, radius):
        """presolve: separating lines farther than `radius` metres from the guess are left out of the QP and verified
        afterwards, and a replan whose unconstrained minimiser is feasible returns it without iterating
        (nep_batch_set_line_cull); 0 turns it off"""
        check(lib().nep_batch_set_line_cull(self._h, float(radius)))

    def line_cull(self):
        """the presolve radius in force (0: off): nep_batch_get_line_cull — 4 m by default at every scene size"""
        return float(lib().nep_batch_get_line_cull(self._h))

    def redo_count(self):
        """replans the last replan sent through the presolve's redo pass (nep_batch_debug_redo_count)"""
        r = np.zeros(2, dtype=np.int32)
        n = int(check(lib().nep_batch_debug_redo_count(self._h, abi.iptr(r))))
        self.redo_reasons = {"parked_line_violated": int(r[0]), "moved_beyond_radius": int(r[1])}
        return n

    def reserve_row_scratch(self):
        """one worst-case row-scratch area per slot instead of the redo pass's pool (nep_batch_reserve_row_scratch)"""
        check(lib().nep_batch_reserve_row_scratch(self._h))

    def fe_search_us(self):
        """device time of the last front-end search of every slot, microseconds (nep_batch_fe_search_us)"""
        out = np.zeros(self.slots, dtype=np.float32)
        check(lib().nep_batch_fe_search_us(self._h, out.ctypes.data_as(C.POINTER(C.c_float)), int(self.slots)))
        return out

    def set_fe_ent_big_records(self, records):
        """records in the pool of big entangle-state records of the front end (0: default; nep_batch_set_fe_ent_big_records)"""
        check(lib().nep_batch_set_fe_ent_big_records(self._h, int(records)))

    def set_fe_ent_fast_caps(self, list_cap=40, add_cap=32, bend_cap=8):
        """what the fixed entangle-state record's path accepts before a child goes to a big record (results do not depend on it)"""
        check(lib().nep_batch_set_fe_ent_fast_caps(self._h, int(list_cap), int(add_cap), int(bend_cap)))

    def set_line_capacity(self, lines_per_segment):
        """lines a (replan, segment) bucket holds: 0 default budget, -1 the reference's worst case, n > 0 (nep_batch_set_line_capacity)"""
        check(lib().nep_batch_set_line_capacity(self._h, int(lines_per_segment)))

    def line_bucket_bytes(self):
        return int(lib().nep_batch_line_bucket_bytes(self._h))

    def row_scratch_bytes(self):
        return int(lib().nep_batch_row_scratch_bytes(self._h))

    def set_separator_pack(self, pack):
        """test hook: 0 segments per wave by launch size, -1 the unpacked separator, 1..8 forced (nep_batch_debug_set_separator_pack)"""
        check(lib().nep_batch_debug_set_separator_pack(self._h, int(pack)))

    def redo_list(self, cap=4096):
        out = np.zeros(cap, dtype=np.int32)
        n = int(check(lib().nep_batch_debug_redo_list(self._h, abi.iptr(out), cap)))
        return out[:n].copy()

    def set_separator_rule(self, rule):
        """which vertex of the separating-line LP is returned: 0 largest gap (default), 1 the one a primal simplex of GLPK's
        default class reaches (nep_batch_set_separator_rule)"""
        check(lib().nep_batch_set_separator_rule(self._h, int(rule)))

    def set_tolerances(self, residual_tol=1e-10, gap_tol=1e-11):
        """the interior point's strict tests (nep_batch_set_tolerances); (1e-6, 1e-8) = Gurobi's defaults, what the reference's solver stops at"""
        check(lib().nep_batch_set_tolerances(self._h, float(residual_tol), float(gap_tol)))

    def set_polish(self, on=True):
        """the active-set polish of solves that end without the strict tests (on by default, under the line presolve as well — 2 is a
        synonym; 3: every-row solves only, round 5's default; False / 0: off): nep_batch_set_polish"""
        check(lib().nep_batch_set_polish(self._h, int(on) if (on is not True and on is not False) else (1 if on else 0)))

    def debug_option(self, name, value):
        """development aid (include/neptune_backend_debug.h: nep_batch_debug_set_option)"""
        check(lib().nep_batch_debug_set_option(self._h, name.encode(), int(value)))

    def polish_count(self):
        """(replans listed for the polish pass of the last replan, replans it certified): nep_batch_debug_polish_count"""
        a = C.c_int32(0); b = C.c_int32(0)
        check(lib().nep_batch_debug_polish_count(self._h, C.byref(a), C.byref(b)))
        return a.value, b.value

    def polish_flags(self):
        """per slot of the last replan: 0 not listed for the polish pass; bit 0 / 1 the first / relaxed solve was left for it; bit 8 (256) the pass
        certified an optimum and wrote the result (nep_batch_debug_polish_flags)"""
        out = np.zeros(self.n_scenes * self.n_local, dtype=np.int32)
        check(lib().nep_batch_debug_polish_flags(self._h, abi.iptr(out), len(out)))
        return out

    def set_safety_check_prev(self, on=True):
        """also turn down new trajectories that collide with another agent's PREVIOUS record (nep_batch_set_safety_check_prev)"""
        check(lib().nep_batch_set_safety_check_prev(self._h, 1 if on else 0))

    def debug_conflicts(self, scene=0):
        out = np.zeros((self.N, self.N), dtype=np.uint8)
        check(lib().nep_batch_debug_conflicts(self._h, scene, out.ctypes.data_as(C.POINTER(C.c_uint8))))
        return out

    def solutions(self, timing=False):
        """the nep_solution records of the last replan.  stats.solve_us (the workgroup's measured device time, which differs
        from run to run) is zeroed unless timing=True, so that two runs' records can be compared byte for byte."""
        self.torch.cuda.synchronize(self.device)
        sol = self.d_solution.cpu().numpy().view(abi.SOLUTION_DTYPE).copy()
        if not timing:
            sol["stats"]["solve_us"] = 0.0
        return sol

    def active_rows(self, tol=1e-6):
        """(box rows, line rows) [slots][2] whose slack at the last replan's solutions is below tol (nep_batch_active_rows)"""
        out = self.torch.zeros(self.slots * 2, dtype=self.torch.int32, device=self.device)
        check(lib().nep_batch_active_rows(self._h, self.d_solution.data_ptr(), float(tol), out.data_ptr(), self.torch.cuda.current_stream(self.device).cuda_stream))
        self.torch.cuda.synchronize(self.device)
        return out.cpu().numpy().reshape(self.slots, 2)

    def states(self):
        self.torch.cuda.synchronize(self.device)
        return self.d_states.cpu().numpy().reshape(self.slots, self.par.max_states, abi.NEP_STATE_DOUBLES).copy()

    def commits(self):
        self.torch.cuda.synchronize(self.device)
        return self.d_commit.cpu().numpy().view(abi.TRAJ_REC_DTYPE).copy()

    def enable_timing(self, on=True):
        check(lib().nep_batch_enable_timing(self._h, 1 if on else 0))

    def reset_timing(self):
        check(lib().nep_batch_reset_timing(self._h))

    def kernel_time_ms(self, which):
        """which: 0 hulls, 1 separator, 2 QP, 3 whole sequence -> (avg ms per launch, launches)."""
        ms = C.c_double(0); n = C.c_int32(0)
        check(lib().nep_batch_kernel_time(self._h, which, C.byref(ms), C.byref(n)))
        return ms.value, n.value

    def debug_phase_cycles(self, slot=0):
        out = (C.c_int64 * 16)()
        check(lib().nep_batch_debug_phase_cycles(self._h, slot, out))
        return list(out)

    def debug_hulls(self, scene=0):
        hx = np.zeros((self.N, self.par.num_pol, abi.NEP_HULL_MAX_V, 2)); hn = np.zeros((self.N, self.par.num_pol), dtype=np.int32)
        check(lib().nep_batch_debug_hulls(self._h, scene, abi.dptr(hx), abi.iptr(hn)))
        return hx, hn

    def debug_lines(self, slot, cap=8192):
        seg = np.zeros(cap, dtype=np.int32); nd = np.zeros((cap, 3)); n = C.c_int32(0)
        check(lib().nep_batch_debug_lines(self._h, slot, cap, abi.iptr(seg), abi.dptr(nd), C.byref(n)))
        return seg[:n.value].copy(), nd[:n.value].copy()
