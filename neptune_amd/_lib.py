"""Loader of the C-ABI shared library (include/neptune_backend.h).  Fails loudly: there is no
CPU path behind these entry points."""
import ctypes as C
import os

from . import abi

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("NEP_BACKEND_LIB") or os.path.join(_HERE, "libneptune_backend.so")   # (NEP_BACKEND_LIB: development aid, A/B builds)
_lib = None

# every symbol include/neptune_backend.h declares (the drop-in surface + the batched handle)
EXPORTS = [
    "nep_backend_create", "nep_backend_destroy", "nep_backend_set_max_values", "nep_backend_set_max_runtime",
    "nep_backend_set_tether_length", "nep_backend_set_static_obst_vert", "nep_backend_set_init_trajectory",
    "nep_backend_set_hulls", "nep_backend_set_hulls_no_inflation", "nep_backend_set_ent_state_vector",
    "nep_backend_optimize", "nep_backend_generate_pwp_out", "nep_backend_get_stats", "nep_inflate_static",
    "nep_separator_batch", "nep_separator_batch_rule", "nep_gjk_batch", "nep_hulls_batch", "nep_batch_create",
    "nep_batch_destroy", "nep_batch_set_scene_statics", "nep_batch_replan", "nep_batch_replan_lines", "nep_batch_replan_solve", "nep_batch_hull_block_bytes",
    "nep_batch_set_ent_samples", "nep_batch_hulls", "nep_batch_replan_hulls", "nep_comm_unique_id",
    "nep_comm_create", "nep_comm_destroy", "nep_comm_nranks", "nep_batch_exchange_hulls",
    "nep_batch_exchange_records", "nep_batch_exchange_slots", "nep_comm_reserve", "nep_batch_ent_bytes",
    "nep_batch_safety_commit", "nep_batch_set_line_cull", "nep_backend_set_line_cull", "nep_batch_get_line_cull", "nep_batch_reserve_row_scratch",
    "nep_batch_set_line_capacity", "nep_batch_set_separator_rule", "nep_backend_set_separator_rule",
    "nep_batch_set_tolerances", "nep_backend_set_tolerances", "nep_batch_set_polish", "nep_backend_set_polish", "nep_batch_set_max_runtime",
    "nep_batch_set_safety_check_prev", "nep_batch_wait", "nep_batch_check", "nep_abi_sizeof", "nep_last_error",
    "nep_version",
]
# every symbol include/neptune_backend_debug.h declares (test hooks, measurement aids, A/B knobs)
DEBUG_EXPORTS = [
    "nep_batch_debug_polish_count", "nep_batch_debug_polish_flags",
    "nep_backend_debug_time_sequence", "nep_backend_debug_set_lines", "nep_backend_debug_get_lines",
    "nep_debug_regroup_records", "nep_batch_debug_redo_count", "nep_batch_debug_redo_list",
    "nep_batch_line_bucket_bytes", "nep_batch_row_scratch_bytes", "nep_batch_active_rows",
    "nep_batch_debug_set_separator_pack", "nep_batch_qp_placement", "nep_batch_set_launch_order",
    "nep_batch_debug_launch_order", "nep_batch_set_hull_kernel", "nep_batch_debug_conflicts",
    "nep_batch_kernel_time", "nep_batch_enable_timing", "nep_batch_reset_timing", "nep_batch_debug_hulls",
    "nep_batch_debug_lines", "nep_batch_debug_phase_cycles", "nep_batch_fe_search_us",
    "nep_batch_set_fe_ent_fast_caps", "nep_batch_debug_set_option", "nep_backend_debug_set_option", "nep_debug_set_global_option",
]
# every symbol include/neptune_plan.h declares (host-only: no HIP call behind them)
PLAN_EXPORTS = [
    "nep_pwp_compose", "nep_pwp_compose_exact", "nep_dyntraj_wire_size", "nep_dyntraj_encode", "nep_dyntraj_decode", "nep_plan_create",
    "nep_plan_destroy", "nep_plan_reset", "nep_plan_size", "nep_plan_get", "nep_plan_next_goal",
    "nep_plan_select_a", "nep_plan_splice", "nep_plan_update_delta", "nep_plan_delta",
]
# every symbol include/neptune_entangle.h declares (host-only)
# include/neptune_frontend.h
FE_EXPORTS = ["nep_batch_frontend", "nep_batch_frontend_hulls", "nep_batch_set_static_reps", "nep_batch_set_fe_ent_big_records", "nep_batch_frontend_ent", "nep_batch_frontend_ent_hulls", "nep_batch_safety_commit_ent", "nep_batch_next_starts"]
ENT_EXPORTS = ["nep_ent_sample_points", "nep_ent_propagate_segment", "nep_ent_propagate_guess", "nep_ent_case_ids"]


class BackendError(RuntimeError):
    pass


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise BackendError("%s is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                           "(or make -C neptune_amd/csrc); there is no CPU fallback" % LIB_PATH)
    # The library links the HIP runtime; PyTorch ships its own copy of it.  Whichever is loaded first serves both, and it has to
    # be PyTorch's (device memory and streams come from there): a host-only entry point called before `import torch` — the scene
    # helpers inflate static obstacles through nep_inflate_static — would otherwise bind the system's copy, and the handle
    # created later would see no device.  (A C++ host has one runtime and no such order.)
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    L = C.CDLL(LIB_PATH)
    d, i, vp = C.c_double, C.c_int32, C.c_void_p
    pd, pi = C.POINTER(C.c_double), C.POINTER(C.c_int32)
    L.nep_last_error.restype = C.c_char_p
    L.nep_version.restype = C.c_char_p
    L.nep_abi_sizeof.argtypes = [i]; L.nep_abi_sizeof.restype = C.c_int
    L.nep_backend_create.argtypes = [C.POINTER(abi.nep_backend_cfg)]; L.nep_backend_create.restype = vp
    L.nep_backend_destroy.argtypes = [vp]; L.nep_backend_destroy.restype = None
    L.nep_backend_set_max_values.argtypes = [vp] + [d] * 9
    L.nep_backend_set_max_runtime.argtypes = [vp, d]
    L.nep_backend_set_tether_length.argtypes = [vp, d]
    L.nep_backend_set_static_obst_vert.argtypes = [vp, i, pi, pd]
    L.nep_backend_set_init_trajectory.argtypes = [vp, C.POINTER(abi.nep_pwp)]
    L.nep_backend_set_hulls.argtypes = [vp, i, pi, pd]
    L.nep_backend_set_hulls_no_inflation.argtypes = [vp, i, pi, pd]
    L.nep_backend_set_ent_state_vector.argtypes = [vp, C.POINTER(abi.nep_ent_view)]
    L.nep_backend_optimize.argtypes = [vp, pd]
    L.nep_backend_generate_pwp_out.argtypes = [vp, d, d, C.POINTER(abi.nep_pwp), pd, i, pi]
    L.nep_backend_get_stats.argtypes = [vp, C.POINTER(abi.nep_stats)]
    L.nep_backend_debug_set_lines.argtypes = [vp, i, pi, pd]
    L.nep_backend_debug_get_lines.argtypes = [vp, i, pi, pd, pi]
    L.nep_separator_batch.argtypes = [i, pi, pd, pi, pd, pd, pi]
    L.nep_inflate_static.argtypes = [i, pi, pd, d, pi, pd, i]
    L.nep_separator_batch_rule.argtypes = [i, i, pi, pd, pi, pd, pd, pi]
    L.nep_batch_set_separator_rule.argtypes = [vp, i]
    L.nep_backend_set_separator_rule.argtypes = [vp, i]
    L.nep_batch_set_tolerances.argtypes = [vp, C.c_double, C.c_double]
    L.nep_backend_set_tolerances.argtypes = [vp, C.c_double, C.c_double]
    L.nep_gjk_batch.argtypes = [i, pi, pd, pd, pi]
    L.nep_hulls_batch.argtypes = [i, vp, d, i, d, d, pd, pi, pd, pi]
    L.nep_batch_create.argtypes = [C.POINTER(abi.nep_batch_cfg)]; L.nep_batch_create.restype = vp
    L.nep_batch_destroy.argtypes = [vp]; L.nep_batch_destroy.restype = None
    L.nep_batch_replan.argtypes = [vp, vp, vp, vp, vp, vp, vp, vp]
    L.nep_batch_replan_lines.argtypes = [vp, vp, vp, vp, vp]
    L.nep_batch_replan_solve.argtypes = [vp, vp, vp, vp, vp, vp, vp, vp]
    L.nep_batch_ent_bytes.argtypes = [vp]; L.nep_batch_ent_bytes.restype = C.c_int64
    L.nep_batch_wait.argtypes = [vp, vp]
    L.nep_batch_kernel_time.argtypes = [vp, i, pd, pi]
    L.nep_batch_enable_timing.argtypes = [vp, i]
    L.nep_batch_reset_timing.argtypes = [vp]
    L.nep_batch_debug_hulls.argtypes = [vp, i, pd, pi]
    L.nep_batch_debug_lines.argtypes = [vp, i, i, pi, pd, pi]
    L.nep_batch_debug_phase_cycles.argtypes = [vp, i, C.POINTER(C.c_int64)]
    L.nep_batch_safety_commit.argtypes = [vp, vp, vp, vp, vp, vp, vp]
    L.nep_batch_debug_conflicts.argtypes = [vp, i, C.POINTER(C.c_uint8)]
    L.nep_batch_set_safety_check_prev.argtypes = [vp, i]
    L.nep_batch_set_line_cull.argtypes = [vp, d]; L.nep_backend_set_line_cull.argtypes = [vp, d]
    L.nep_batch_debug_set_option.argtypes = [vp, C.c_char_p, i]; L.nep_backend_debug_set_option.argtypes = [vp, C.c_char_p, i]; L.nep_debug_set_global_option.argtypes = [C.c_char_p, i]
    L.nep_batch_get_line_cull.argtypes = [vp]; L.nep_batch_get_line_cull.restype = d
    L.nep_batch_debug_redo_count.argtypes = [vp, pi]
    L.nep_batch_debug_redo_list.argtypes = [vp, pi, i]
    L.nep_batch_debug_set_separator_pack.argtypes = [vp, i]
    L.nep_batch_active_rows.argtypes = [vp, vp, d, vp, vp]
    L.nep_batch_reserve_row_scratch.argtypes = [vp]
    L.nep_batch_set_line_capacity.argtypes = [vp, i]
    L.nep_batch_set_fe_ent_big_records.argtypes = [vp, C.c_int64]
    L.nep_batch_set_fe_ent_fast_caps.argtypes = [vp, i, i, i]
    L.nep_batch_fe_search_us.argtypes = [vp, C.POINTER(C.c_float), i]
    L.nep_batch_line_bucket_bytes.argtypes = [vp]; L.nep_batch_line_bucket_bytes.restype = C.c_int64
    L.nep_batch_row_scratch_bytes.argtypes = [vp]; L.nep_batch_row_scratch_bytes.restype = C.c_int64
    L.nep_backend_debug_time_sequence.argtypes = [vp, vp, i, pi, pd, pi, pd, vp, d, d, i, pd, pd]
    L.nep_batch_set_max_runtime.argtypes = [vp, d]
    L.nep_batch_qp_placement.argtypes = [vp]
    L.nep_batch_set_launch_order.argtypes = [vp, i]
    L.nep_batch_set_hull_kernel.argtypes = [vp, i]
    L.nep_batch_debug_launch_order.argtypes = [vp, pi, i, pi]
    L.nep_batch_set_scene_statics.argtypes = [vp, i, i, pi, pd]
    L.nep_batch_check.argtypes = [vp, vp]
    L.nep_comm_unique_id.argtypes = [C.POINTER(C.c_uint8)]
    L.nep_comm_create.argtypes = [C.POINTER(C.c_uint8), i, i]; L.nep_comm_create.restype = vp
    L.nep_comm_destroy.argtypes = [vp]; L.nep_comm_destroy.restype = None
    L.nep_comm_nranks.argtypes = [vp]
    L.nep_comm_reserve.argtypes = [vp, C.c_int64, C.c_int64]
    L.nep_batch_exchange_hulls.argtypes = [vp, vp, vp, vp, vp]
    L.nep_batch_exchange_records.argtypes = [vp, vp, vp, vp, vp]
    L.nep_debug_regroup_records.argtypes = [vp, vp, i, i, i, vp]
    L.nep_batch_hull_block_bytes.argtypes = [vp]; L.nep_batch_hull_block_bytes.restype = C.c_int64
    L.nep_batch_hulls.argtypes = [vp, vp, vp, vp, vp]
    L.nep_batch_replan_hulls.argtypes = [vp, vp, i, vp, vp, vp, vp, vp, vp]
    ppwp, prec, phdr = C.POINTER(abi.nep_pwp), C.POINTER(abi.nep_traj_rec), C.POINTER(abi.nep_wire_header)
    pu8 = C.POINTER(C.c_uint8)
    L.nep_pwp_compose.argtypes = [d, d, ppwp, ppwp, ppwp]
    L.nep_pwp_compose_exact.argtypes = [d, ppwp, ppwp, ppwp]
    L.nep_dyntraj_wire_size.argtypes = [prec, phdr]; L.nep_dyntraj_wire_size.restype = C.c_int64
    L.nep_dyntraj_encode.argtypes = [prec, phdr, pu8, C.c_size_t]; L.nep_dyntraj_encode.restype = C.c_int64
    L.nep_dyntraj_decode.argtypes = [pu8, C.c_size_t, prec, phdr]; L.nep_dyntraj_decode.restype = C.c_int64
    L.nep_plan_create.argtypes = [C.POINTER(abi.nep_plan_cfg)]; L.nep_plan_create.restype = vp
    L.nep_plan_destroy.argtypes = [vp]; L.nep_plan_destroy.restype = None
    L.nep_plan_reset.argtypes = [vp, pd]
    L.nep_plan_size.argtypes = [vp]
    L.nep_plan_get.argtypes = [vp, i, pd]
    L.nep_plan_next_goal.argtypes = [vp, pd, pi]
    L.nep_plan_select_a.argtypes = [vp, pd, d, C.POINTER(abi.nep_point_a)]
    L.nep_plan_splice.argtypes = [vp, i, pd, i]
    L.nep_plan_update_delta.argtypes = [vp, d]
    L.nep_plan_delta.argtypes = [vp]
    pcfg, pin, pst = C.POINTER(abi.nep_ent_cfg), C.POINTER(abi.nep_ent_inputs), C.POINTER(abi.nep_ent_state)
    L.nep_ent_sample_points.argtypes = [ppwp, d, d, i, i, pd]
    L.nep_ent_propagate_segment.argtypes = [pcfg, pin, pst, pd, pd, pd, i, pd]
    L.nep_ent_propagate_guess.argtypes = [pcfg, pin, pst, vp, i, pi, pi, pi, pi, pst]
    L.nep_ent_case_ids.argtypes = [i, i, pi, pi, pi, i, pi]
    L.nep_batch_frontend.argtypes = [vp, C.POINTER(abi.nep_fe_cfg), vp, vp, vp, vp, vp]
    L.nep_batch_frontend_hulls.argtypes = [vp, C.POINTER(abi.nep_fe_cfg), vp, i, vp, vp, vp, vp]
    L.nep_batch_set_static_reps.argtypes = [vp, i, pd, pd]
    L.nep_batch_frontend_ent.argtypes = [vp, C.POINTER(abi.nep_fe_cfg), vp, vp, vp, vp, vp, vp, vp]
    L.nep_batch_safety_commit_ent.argtypes = [vp, vp, vp, vp, vp, i, d, vp, vp, vp]
    L.nep_batch_next_starts.argtypes = [vp, vp, d, vp, vp, d, vp]
    L.nep_batch_frontend_ent_hulls.argtypes = [vp, C.POINTER(abi.nep_fe_cfg), vp, i, vp, vp, vp, vp, vp, vp]
    L.nep_batch_exchange_slots.argtypes = [vp, vp, vp, vp, C.c_int64, vp]
    L.nep_batch_set_ent_samples.argtypes = [vp, i]
    L.nep_batch_set_polish.argtypes = [vp, i]; L.nep_backend_set_polish.argtypes = [vp, i]; L.nep_batch_debug_polish_count.argtypes = [vp, pi, pi]; L.nep_batch_debug_polish_flags.argtypes = [vp, pi, i]
    # the records this mirror builds with ctypes must have the library's layout (a library compiled from other headers would read
    # them at another stride): fail loudly at load, not as wrong numbers later
    L.nep_abi_sizeof.argtypes = [i]; L.nep_abi_sizeof.restype = i
    for which, struct in ((1, abi.nep_traj_rec), (5, abi.nep_guess), (6, abi.nep_solution), (11, abi.nep_fe_cfg), (12, abi.nep_fe_start),
                          (13, abi.nep_fe_result), (14, abi.nep_fe_ent_state)):
        if L.nep_abi_sizeof(which) != C.sizeof(struct):
            raise BackendError("%s: sizeof(%s) is %d in the library, %d in neptune_amd/abi.py — rebuild the library from this tree's headers"
                               % (LIB_PATH, struct.__name__, L.nep_abi_sizeof(which), C.sizeof(struct)))
    _lib = L
    return L


def check(rc):
    if rc < 0:
        raise BackendError("neptune backend error %d: %s" % (rc, lib().nep_last_error().decode()))
    return rc
