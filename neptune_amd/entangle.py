"""Tether entanglement-state propagation over include/neptune_entangle.h: the
real eu::ent_state inputs of the back end's entangle rows, accumulated along a guess the way the
front end does node by node (reference neptune/src/kinodynamic_search.cpp:707-895, entangle_utils.cpp)."""
import ctypes as C

import numpy as np

from . import abi
from ._lib import BackendError, lib


class EntangleError(BackendError):
    pass


def _ck(rc, what):
    if rc < 0:
        raise EntangleError("%s failed: %d" % (what, rc))
    return rc


def sample_points(rec_pwp, t_start, t_end, num_pol, num_samples):
    """Neptune::SamplePointsOfIntervals for one committed trajectory (abi.nep_pwp or the 'pwp' field of a
    TRAJ_REC_DTYPE record) -> [num_pol][num_samples+1][2]."""
    if not isinstance(rec_pwp, abi.nep_pwp):
        rec_pwp = abi.nep_pwp.from_buffer_copy(np.ascontiguousarray(rec_pwp).tobytes())
    out = np.zeros((num_pol, num_samples + 1, 2))
    _ck(lib().nep_ent_sample_points(C.byref(rec_pwp), float(t_start), float(t_end), num_pol, num_samples, abi.dptr(out)), "nep_ent_sample_points")
    return out


class State:
    """eu::ent_state in caller-owned arrays."""

    def __init__(self, n_active, cap=None):
        cap = cap if cap is not None else n_active + 16
        self.alphas = np.zeros((cap, 2), dtype=np.int32)
        self.betas = np.zeros(cap)
        self.bend_idx = np.zeros(cap, dtype=np.int32)
        self.active = np.zeros(n_active, dtype=np.int32)
        self.c = abi.nep_ent_state(0, 0, cap, n_active, abi.iptr(self.alphas), abi.dptr(self.betas), abi.iptr(self.bend_idx),
                                   abi.iptr(self.active))

    def as_lists(self):
        n, b = self.c.n_alpha, self.c.n_bend
        return ([tuple(int(v) for v in a) for a in self.alphas[:n]], [float(v) for v in self.betas[:n]],
                [int(v) for v in self.bend_idx[:b]], [int(v) for v in self.active])


class EntangleCheck:
    """What KinodynamicSearch holds for the check + the per-replan inputs."""

    def __init__(self, num_agents, agent_id, num_pol, num_samples, T_span, cable_length, pb, static_rep=(), static_longest=()):
        self.N, self.id, self.num_pol, self.ns = num_agents, agent_id, num_pol, num_samples
        self._pb = np.ascontiguousarray(pb, dtype=np.float64).reshape(num_agents, 2)
        S = len(static_rep)
        self.S = S
        self._rep = np.ascontiguousarray(static_rep, dtype=np.float64).reshape(S, 2, 2) if S else np.zeros((1, 2, 2))
        self._lng = np.ascontiguousarray(static_longest, dtype=np.float64).reshape(S, 2) if S else np.zeros((1, 2))
        self.cfg = abi.nep_ent_cfg(num_agents, agent_id, num_pol, num_samples, T_span, cable_length, S, 0, abi.dptr(self._pb),
                                   abi.dptr(self._rep), abi.dptr(self._lng))
        self.n_active = num_agents + S
        self.inputs = None

    def set_inputs(self, sampled, present, bendpts):
        """sampled [N][num_pol][ns+1][2]; present [N]; bendpts: list of (k,2) arrays per agent."""
        self._sampled = np.ascontiguousarray(sampled, dtype=np.float64).reshape(self.N, self.num_pol, self.ns + 1, 2)
        self._present = np.ascontiguousarray(present, dtype=np.int32).reshape(self.N)
        off = np.zeros(self.N + 1, dtype=np.int32)
        for j, b in enumerate(bendpts):
            off[j + 1] = off[j] + len(b)
        xy = np.zeros((max(int(off[-1]), 1), 2))
        for j, b in enumerate(bendpts):
            if len(b):
                xy[off[j]:off[j + 1]] = np.asarray(b, dtype=np.float64).reshape(-1, 2)
        self._boff, self._bxy = off, xy
        self.inputs = abi.nep_ent_inputs(abi.dptr(self._sampled), abi.iptr(self._present), abi.iptr(off), abi.dptr(xy))

    def new_state(self):
        return State(self.n_active)

    def propagate_segment(self, state, coeff_x, coeff_y, end_xy, index):
        """-> (entangled, arc_length); updates state in place (entanglesWithOtherAgents)."""
        cx = np.ascontiguousarray(coeff_x, dtype=np.float64); cy = np.ascontiguousarray(coeff_y, dtype=np.float64)
        e = np.ascontiguousarray(end_xy, dtype=np.float64)
        arc = C.c_double(0.0)
        rc = _ck(lib().nep_ent_propagate_segment(C.byref(self.cfg), C.byref(self.inputs), C.byref(state.c), abi.dptr(cx), abi.dptr(cy),
                                                 abi.dptr(e), int(index), C.byref(arc)), "nep_ent_propagate_segment")
        return bool(rc), arc.value

    def propagate_guess(self, init, guess):
        """guess: one GUESS_DTYPE record.  -> dict(alpha_off, alphas, active_cases [K+1][n_active], entangled_at,
        final: State, case_id [8][N]) — alpha_off/alphas/active_cases are the nep_ent_view fields."""
        g = np.ascontiguousarray(guess, dtype=abi.GUESS_DTYPE).reshape(1)
        K = int(g[0]["K"])
        cap = (K + 1) * (self.n_active + 16)
        alpha_off = np.zeros(K + 2, dtype=np.int32)
        alphas = np.zeros((cap, 2), dtype=np.int32)
        active = np.zeros((K + 1, self.n_active), dtype=np.int32)
        hit = C.c_int32(0)
        final = self.new_state()
        _ck(lib().nep_ent_propagate_guess(C.byref(self.cfg), C.byref(self.inputs), C.byref(init.c), g.ctypes.data, cap, abi.iptr(alpha_off),
                                          abi.iptr(alphas), abi.iptr(active), C.byref(hit), C.byref(final.c)), "nep_ent_propagate_guess")
        case_id = np.zeros((abi.NEP_MAX_POL, self.N), dtype=np.int32)
        _ck(lib().nep_ent_case_ids(K + 1, self.n_active, abi.iptr(alpha_off), abi.iptr(alphas), abi.iptr(active), self.N, abi.iptr(case_id)),
            "nep_ent_case_ids")
        return dict(alpha_off=alpha_off, alphas=alphas[:alpha_off[K + 1]].copy(), active_cases=active, entangled_at=int(hit.value),
                    final=final, case_id=case_id)
