"""Host side of the replan loop around the back end, over include/neptune_plan.h:
trajectory composition, the DynTraj wire format and the committed-plan deque.

Mirrors the reference names: mu::composePieceWisePol (neptune/src/utils.cpp:318-402),
pwp2PwpMsg / pwpMsg2Pwp + publishOwnTraj / trajCB (utils.cpp:180-261, neptune_ros.cpp:379-480),
mt::committedTrajectory and the plan handling of Neptune::replanFull / getNextGoal
(mader_types.hpp:674-738, neptune.cpp:860-891,1366-1425,1661-1720)."""
import ctypes as C

import numpy as np

from . import abi
from ._lib import BackendError, lib

_ERR = {abi_code: name for name, abi_code in (("NEP_E_ARG", -1), ("NEP_E_STATE", -2), ("NEP_E_HIP", -3), ("NEP_E_CAP", -4))}


class PlanError(BackendError):
    def __init__(self, code, what):
        super().__init__("%s: %s" % (what, _ERR.get(code, code)))
        self.code = code


def _ck(rc, what):
    if rc < 0:
        raise PlanError(int(rc), what)
    return rc


def make_pwp(times, coeff):
    """times [n+1], coeff [3][n][4] -> nep_pwp"""
    p = abi.nep_pwp()
    times = np.asarray(times, dtype=np.float64)
    coeff = np.asarray(coeff, dtype=np.float64)
    n = coeff.shape[1] if coeff.size else 0
    if n > abi.NEP_TRAJ_MAX_SEG:
        raise PlanError(-4, "make_pwp")
    p.n_seg = n
    a = np.ctypeslib.as_array(p.times)
    a[: len(times)] = times
    c = np.ctypeslib.as_array(p.coeff)
    if n:
        c[:, :n, :] = coeff
    return p


def pwp_arrays(p):
    n = p.n_seg
    times = np.array(np.ctypeslib.as_array(p.times)[: n + 1 if n else 0])
    coeff = np.array(np.ctypeslib.as_array(p.coeff)[:, :n, :])
    return times, coeff


def compose_piecewise_pol(t, dc, p1, p2):
    """mu::composePieceWisePol.  p1, p2: nep_pwp, adjusted in place like the reference's by-reference
    arguments.  Returns a new nep_pwp (n_seg == 0 for the reference's empty 'dummy')."""
    out = abi.nep_pwp()
    _ck(lib().nep_pwp_compose(float(t), float(dc), C.byref(p1), C.byref(p2), C.byref(out)), "nep_pwp_compose")
    return out


def compose_exact(t, p1, p2):
    """nep_pwp_compose_exact: the old trajectory from t until p2 starts, then p2, every interval on its own
    local time (what a closed loop should publish; the reference routine does not reproduce the flown path)."""
    out = abi.nep_pwp()
    _ck(lib().nep_pwp_compose_exact(float(t), C.byref(p1), C.byref(p2), C.byref(out)), "nep_pwp_compose_exact")
    return out


def eval_pwp(p, t):
    """position at time t of a nep_pwp whose intervals run on their own local time (held outside its span)"""
    times, co = pwp_arrays(p)
    n = p.n_seg
    k = int(np.searchsorted(times, t, side="right") - 1)
    k = min(max(k, 0), n - 1)
    dt = min(max(t - times[k], 0.0), times[k + 1] - times[k])
    return np.array([((co[ax, k, 0] * dt + co[ax, k, 1]) * dt + co[ax, k, 2]) * dt + co[ax, k, 3] for ax in range(3)])


def dyntraj_encode(rec, seq=0, stamp=(0, 0), frame_id=b""):
    """nep_traj_rec (ctypes struct or 1-element TRAJ_REC_DTYPE array) -> bytes of one mader_msgs/DynTraj."""
    rec = _as_rec(rec)
    hdr = abi.nep_wire_header(seq, stamp[0], stamp[1], 0, frame_id)
    n = _ck(lib().nep_dyntraj_wire_size(C.byref(rec), C.byref(hdr)), "nep_dyntraj_wire_size")
    buf = (C.c_uint8 * n)()
    _ck(lib().nep_dyntraj_encode(C.byref(rec), C.byref(hdr), buf, n), "nep_dyntraj_encode")
    return bytes(buf)


def dyntraj_decode(data):
    """bytes -> (nep_traj_rec, (seq, sec, nsec), bytes consumed)"""
    rec = abi.nep_traj_rec()
    hdr = abi.nep_wire_header()
    buf = (C.c_uint8 * len(data)).from_buffer_copy(data) if len(data) else (C.c_uint8 * 1)()
    n = _ck(lib().nep_dyntraj_decode(buf, len(data), C.byref(rec), C.byref(hdr)), "nep_dyntraj_decode")
    return rec, (hdr.seq, hdr.stamp_sec, hdr.stamp_nsec), int(n)


def _as_rec(rec):
    if isinstance(rec, abi.nep_traj_rec):
        return rec
    a = np.ascontiguousarray(rec, dtype=abi.TRAJ_REC_DTYPE).reshape(-1)
    return abi.nep_traj_rec.from_buffer_copy(a[:1].tobytes())


def rec_to_numpy(rec):
    return np.frombuffer(bytes(rec), dtype=abi.TRAJ_REC_DTYPE).copy()


class CommittedPlan:
    """mt::committedTrajectory plan_ plus deltaT_ and the three places replanFull touches them."""

    def __init__(self, dc, T_span, lower_bound_runtime, upper_bound_runtime, runtime_opt, factor_alpha, deltaT0=75):
        cfg = abi.nep_plan_cfg(dc, T_span, lower_bound_runtime, upper_bound_runtime, runtime_opt, factor_alpha,
                               deltaT0, 0)
        self._h = lib().nep_plan_create(C.byref(cfg))
        if not self._h:
            raise PlanError(-1, "nep_plan_create")

    def close(self):
        if self._h:
            lib().nep_plan_destroy(self._h)
            self._h = None

    __del__ = close

    def reset(self, state):
        s = np.ascontiguousarray(state, dtype=np.float64).reshape(12)
        _ck(lib().nep_plan_reset(self._h, abi.dptr(s)), "nep_plan_reset")

    def __len__(self):
        return _ck(lib().nep_plan_size(self._h), "nep_plan_size")

    def get(self, i):
        s = np.zeros(12)
        _ck(lib().nep_plan_get(self._h, int(i), abi.dptr(s)), "nep_plan_get")
        return s

    def to_array(self):
        return np.array([self.get(i) for i in range(len(self))]).reshape(-1, 12)

    def next_goal(self):
        """Neptune::getNextGoal -> (state, last_point)"""
        s = np.zeros(12)
        last = C.c_int32(0)
        _ck(lib().nep_plan_next_goal(self._h, abi.dptr(s), C.byref(last)), "nep_plan_next_goal")
        return s, bool(last.value)

    def select_a(self, state_pos, time_now):
        pos = np.ascontiguousarray(state_pos, dtype=np.float64).reshape(3)
        out = abi.nep_point_a()
        _ck(lib().nep_plan_select_a(self._h, abi.dptr(pos), float(time_now), C.byref(out)), "nep_plan_select_a")
        return out

    def splice(self, k_index_end, traj_out):
        t = np.ascontiguousarray(traj_out, dtype=np.float64).reshape(-1, 12)
        _ck(lib().nep_plan_splice(self._h, int(k_index_end), abi.dptr(t), t.shape[0]), "nep_plan_splice")

    def update_delta(self, elapsed_ms):
        _ck(lib().nep_plan_update_delta(self._h, float(elapsed_ms)), "nep_plan_update_delta")

    @property
    def deltaT(self):
        return lib().nep_plan_delta(self._h)
