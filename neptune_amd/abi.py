"""ctypes mirror of include/neptune_backend.h (the C ABI of the back-end path).

Only layouts and constants live here; no compute.  Field order and sizes must match the header
exactly (tests/test_abi.py checks sizeof() against the values the C library reports).
"""
import ctypes as C

import numpy as np

NEP_MAX_POL = 8
NEP_TRAJ_MAX_SEG = 16
NEP_HULL_MAX_V = 16
NEP_HULL_MAX_CP = 16
NEP_MAX_BEND = 8
NEP_STATE_DOUBLES = 12

NEP_OK, NEP_RELAXED, NEP_FAILED = 0, 1, 2


class nep_pwp(C.Structure):
    """mt::PieceWisePol (reference neptune/include/mader_types.hpp:462-548)."""
    _fields_ = [("n_seg", C.c_int32), ("_pad", C.c_int32),
                ("times", C.c_double * (NEP_TRAJ_MAX_SEG + 1)),
                ("coeff", ((C.c_double * 4) * NEP_TRAJ_MAX_SEG) * 3)]


class nep_traj_rec(C.Structure):
    """mader_msgs/DynTraj (reference mader_msgs/msg/DynTraj.msg:1-9) as a fixed-size record."""
    _fields_ = [("id", C.c_int32), ("is_agent", C.c_int32), ("n_bend", C.c_int32),
                ("valid", C.c_int32),
                ("bbox", C.c_double * 3), ("pos", C.c_double * 3),
                ("bend", (C.c_double * 2) * NEP_MAX_BEND),
                ("pwp", nep_pwp)]


class nep_backend_cfg(C.Structure):
    _fields_ = [("num_pol", C.c_int32), ("deg_pol", C.c_int32), ("id", C.c_int32),
                ("num_agents", C.c_int32),
                ("T_span", C.c_double), ("weight_term", C.c_double), ("rad_term", C.c_double),
                ("use_linear_constraints", C.c_int32), ("_pad", C.c_int32),
                ("pb", C.POINTER(C.c_double))]


class nep_ent_view(C.Structure):
    _fields_ = [("n_states", C.c_int32), ("n_active", C.c_int32),
                ("alpha_off", C.POINTER(C.c_int32)), ("alphas", C.POINTER(C.c_int32)),
                ("active_cases", C.POINTER(C.c_int32)),
                ("bend_off", C.POINTER(C.c_int32)), ("bend_xy", C.POINTER(C.c_double))]


class nep_stats(C.Structure):
    _fields_ = [("status", C.c_int32), ("iters", C.c_int32), ("iters_first", C.c_int32),
                ("n_lines", C.c_int32), ("n_lp", C.c_int32), ("n_lp_failed", C.c_int32),
                ("n_rows", C.c_int32), ("qc_active", C.c_int32),
                ("objective", C.c_double), ("solve_us", C.c_double)]


class nep_batch_cfg(C.Structure):
    _fields_ = [("num_agents", C.c_int32), ("first_local", C.c_int32), ("n_local", C.c_int32),
                ("num_pol", C.c_int32), ("n_static", C.c_int32), ("enable_entangle", C.c_int32),
                ("max_states", C.c_int32), ("n_scenes", C.c_int32),
                ("T_span", C.c_double), ("weight_term", C.c_double), ("dc", C.c_double),
                ("drone_radius", C.c_double),
                ("x_min", C.c_double), ("x_max", C.c_double), ("y_min", C.c_double),
                ("y_max", C.c_double), ("z_min", C.c_double), ("z_max", C.c_double),
                ("v_max", C.c_double), ("a_max", C.c_double),
                ("pb", C.POINTER(C.c_double)), ("static_off", C.POINTER(C.c_int32)),
                ("static_xy", C.POINTER(C.c_double))]


class nep_guess(C.Structure):
    _fields_ = [("K", C.c_int32), ("n_alpha", C.c_int32), ("t_start", C.c_double),
                ("coeff", ((C.c_double * 4) * NEP_MAX_POL) * 3)]


class nep_solution(C.Structure):
    _fields_ = [("stats", nep_stats), ("K", C.c_int32), ("n_states", C.c_int32),
                ("times", C.c_double * (NEP_MAX_POL + 1)),
                ("coeff", ((C.c_double * 4) * NEP_MAX_POL) * 3)]


# numpy structured dtypes with identical layout (used for device<->host staging through torch)
class nep_wire_header(C.Structure):
    """ROS Header fields of a DynTraj message on the wire (include/neptune_plan.h)."""
    _fields_ = [("seq", C.c_uint32), ("stamp_sec", C.c_uint32), ("stamp_nsec", C.c_uint32),
                ("_pad", C.c_uint32), ("frame_id", C.c_char_p)]


class nep_plan_cfg(C.Structure):
    """The yaml parameters replanFull's plan handling reads (neptune.cpp:1366-1425,1713-1720)."""
    _fields_ = [("dc", C.c_double), ("T_span", C.c_double), ("lower_bound_runtime", C.c_double),
                ("upper_bound_runtime", C.c_double), ("runtime_opt", C.c_double),
                ("factor_alpha", C.c_double), ("deltaT0", C.c_int32), ("_pad", C.c_int32)]


class nep_point_a(C.Structure):
    _fields_ = [("A", C.c_double * 12), ("k_index", C.c_int32), ("k_index_end", C.c_int32),
                ("runtime_search", C.c_double), ("t_start", C.c_double)]


class nep_ent_cfg(C.Structure):
    """include/neptune_entangle.h: what KinodynamicSearch holds for the entangle check."""
    _fields_ = [("num_agents", C.c_int32), ("id", C.c_int32), ("num_pol", C.c_int32), ("num_samples", C.c_int32),
                ("T_span", C.c_double), ("cable_length", C.c_double), ("n_static", C.c_int32), ("_pad", C.c_int32),
                ("pb", C.POINTER(C.c_double)), ("static_rep", C.POINTER(C.c_double)), ("static_longest", C.POINTER(C.c_double))]


class nep_ent_inputs(C.Structure):
    _fields_ = [("sampled", C.POINTER(C.c_double)), ("present", C.POINTER(C.c_int32)),
                ("bend_off", C.POINTER(C.c_int32)), ("bend_xy", C.POINTER(C.c_double))]


class nep_ent_state(C.Structure):
    _fields_ = [("n_alpha", C.c_int32), ("n_bend", C.c_int32), ("cap", C.c_int32), ("n_active", C.c_int32),
                ("alphas", C.POINTER(C.c_int32)), ("betas", C.POINTER(C.c_double)), ("bend_idx", C.POINTER(C.c_int32)),
                ("active_cases", C.POINTER(C.c_int32))]


class nep_fe_cfg(C.Structure):
    """include/neptune_frontend.h: the KinodynamicSearch setters the batched front end needs."""
    _fields_ = [("j_max", C.c_double), ("voxel_size", C.c_double), ("bias", C.c_double), ("goal_size", C.c_double),
                ("cable_length", C.c_double), ("num_samples", C.c_int32), ("beam_width", C.c_int32),
                ("pad_hold", C.c_int32), ("enable_entangle", C.c_int32), ("ent_samples", C.c_int32), ("_pad", C.c_int32)]


NEP_FE_ENT_CAP = 40


class nep_fe_ent_state(C.Structure):
    """eu::ent_state of a search node / of point A in a fixed-size record (include/neptune_frontend.h)."""
    _fields_ = [("n_alpha", C.c_int32), ("n_bend", C.c_int32), ("id", C.c_int16 * NEP_FE_ENT_CAP), ("cs", C.c_int8 * NEP_FE_ENT_CAP),
                ("beta", C.c_double * NEP_FE_ENT_CAP), ("bend", C.c_int8 * 8)]


class nep_fe_start(C.Structure):
    _fields_ = [("pos", C.c_double * 3), ("vel", C.c_double * 3), ("accel", C.c_double * 3), ("goal", C.c_double * 3),
                ("t_start", C.c_double)]


class nep_fe_result(C.Structure):
    _fields_ = [("status", C.c_int32), ("K", C.c_int32), ("depth", C.c_int32), ("n_children", C.c_int32),
                ("n_feasible", C.c_int32), ("n_collision_free", C.c_int32), ("goal_occupied", C.c_int32), ("_pad", C.c_int32),
                ("cost", C.c_double), ("dist_to_goal", C.c_double), ("n_entangled", C.c_int32), ("ent_overflow", C.c_int32)]


def np_dtype(struct):
    return np.dtype(struct)


TRAJ_REC_DTYPE = np.dtype(nep_traj_rec)
GUESS_DTYPE = np.dtype(nep_guess)
SOLUTION_DTYPE = np.dtype(nep_solution)
FE_START_DTYPE = np.dtype(nep_fe_start)
FE_RESULT_DTYPE = np.dtype(nep_fe_result)
FE_ENT_STATE_DTYPE = np.dtype(nep_fe_ent_state)


def dptr(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def iptr(a):
    return a.ctypes.data_as(C.POINTER(C.c_int32))
