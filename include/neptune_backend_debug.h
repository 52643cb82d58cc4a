/* neptune_backend_debug.h — test hooks, measurement aids and A/B knobs of libneptune_backend.so.
 *
 * NOT part of the drop-in surface: nothing here is needed to replace PolySolverGurobi (include/neptune_backend.h holds the
 * entry points the reference's call sites bind, solver_gurobi_poly.hpp:28-49).  These exist for the parity tests (feed lines in,
 * read intermediates out), for bench.py (per-kernel HIP-event times, active-row counts) and for A/B runs of scheduling choices
 * whose results do not depend on them.  Same library, same ABI conventions (plain pointers and sizes, int status).            */
#ifndef NEPTUNE_BACKEND_DEBUG_H
#define NEPTUNE_BACKEND_DEBUG_H

#include "neptune_backend.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Measurement aid: the drop-in call sequence of one replan (neptune.cpp:1514-1527: setInitTrajectory -> setHulls ->
 * setHullsNoInflation -> setEntStateVector -> optimize -> generatePwpOut) n_iter times from the calling thread;
 * us_out[n_iter] = wall microseconds of each iteration, us_optimize_out (may be NULL) = of optimize() alone.  h0_off / ent may
 * be NULL (entangle check off).  Returns the last optimize's status or < 0.                                           */
int nep_backend_debug_time_sequence(nep_backend_t* h, const nep_pwp* init, int32_t n_obst, const int32_t* hull_off,
                                    const double* hull_xy, const int32_t* h0_off, const double* h0_xy, const nep_ent_view* ent,
                                    double t_start, double dc, int32_t n_iter, double* us_out, double* us_optimize_out);

/* Test hook: bypass the separator and use these lines for the next optimize():
 * seg[i] in [0,K), nd[i] = (n1,n2,d) in the reference's scaling (row: n.q + d - 1 <= 0).
 * n_lines < 0 restores the built-in separator.                                                 */
int nep_backend_debug_set_lines(nep_backend_t* h, int32_t n_lines, const int32_t* seg,
                                const double* nd);
/* Test hook: copy out the lines used by the last optimize() (order = row order).              */
int nep_backend_debug_get_lines(nep_backend_t* h, int32_t cap, int32_t* seg, double* nd,
                                int32_t* n_out);

/* Test hook: the regrouping step of nep_batch_exchange_records ([world][n_scenes][n_local] -> [n_scenes][world n_local]). */
int nep_debug_regroup_records(const nep_traj_rec* d_src, nep_traj_rec* d_dst, int32_t world, int32_t n_scenes,
                              int32_t n_local, void* stream);

/* nep_batch_debug_redo_count: replans the last call sent through the presolve's redo pass (test hook); by_reason (may be NULL)
 * receives how many of them had a parked line violated [0] and how many moved farther than the radius [1].            */
int nep_batch_debug_redo_count(nep_batch_t* h, int32_t* by_reason);
int nep_batch_debug_redo_list(nep_batch_t* h, int32_t* slots_out, int32_t cap);    /* the listed slots (test hook) */

/* Diagnostics: bytes of the line buckets / of the row scratch as sized now (nep_batch_set_line_capacity, nep_batch_reserve_row_scratch). */
int64_t nep_batch_line_bucket_bytes(nep_batch_t* h);
int64_t nep_batch_row_scratch_bytes(nep_batch_t* h);

/* Diagnostic ("how hard are these problems"): inequality rows of the QP (solver_gurobi_poly.cpp:433-489) whose slack at the
 * solutions of the last nep_batch_replan* is below tol: d_out [slots][2] int32 = (box rows, separating-line rows) per slot.
 * d_solution is what that replan wrote; the lines are the handle's own (parked ones included).  Asynchronous on `stream`.  */
int nep_batch_active_rows(nep_batch_t* h, const nep_solution* d_solution, double tol, int32_t* d_out, void* stream);

/* Test hook: which form of the presolve's separator the next replans launch — 0 (default) segments per wave picked from the
 * launch size, -1 the unpacked kernel (one segment per wave), 1..NEP_MAX_POL that many segments per wave.  The packed form's
 * list entries hold 8 191 candidates per segment (n_hull + N + S + 8 N with the entangle rows); larger scenes take the unpacked
 * kernel whatever is asked here.  Results do not depend on the form (GPU test).                                          */
int nep_batch_debug_set_separator_pack(nep_batch_t* h, int32_t pack);

/* Which placement of the interior point the handle runs: 1 = qp_reg_kernel (line-row state in registers, four workgroups
 * per CU: chosen when the expected lines per segment fit its register slots, e.g. BASELINE configs 1-4, or when the line
 * presolve is on — config 5 by default), 0 = qp_kernel (row state in LDS with a global spill: config-5 sized problems
 * with the presolve explicitly turned off).  Same solver, same results to rounding.                                    */
int nep_batch_qp_placement(nep_batch_t* h);

/* Launch order of the interior-point workgroups (a scheduling matter: results do not depend on it).  A batch of more than
 * 1 024 replans runs as several waves of workgroups over the chip and their durations spread about 1 : 3, so by default the
 * workgroups of a launch are ordered longest-expected-first, the expectation being the measured device time of the same
 * slot's previous replan (nep_stats.solve_us; the first replan of a handle runs in slot order).  enable = 0 keeps slot order.
 * nep_batch_debug_launch_order copies the order of the last replan (n_out = 0 when it ran in slot order).              */
int nep_batch_set_launch_order(nep_batch_t* h, int32_t enable);
int nep_batch_debug_launch_order(nep_batch_t* h, int32_t* order, int32_t cap, int32_t* n_out);

/* Development aids that were environment variables until round 5 (the library reads no environment variable that changes what it
 * computes or how it schedules; NEP_QP_PROFILE, read at handle creation for the per-phase cycle counters of `make PROFILE=1`, is the
 * one exception and changes no result).  Per handle, by name:
 *   "qp_kernel"    0 automatic | 1 qp_reg_kernel (row state in registers) | 2 qp_kernel (row state in LDS)
 *   "sep_skip"     1 (default) LPs whose line is known to be far by the box test are not solved | 0 they are solved and parked
 *   "sep_no_redo"  1: replans the presolve could not verify keep their first-pass result (inspection only: NOT the optimum)
 *   "sep_pack"     as nep_batch_debug_set_separator_pack
 *   "qp_lpt" / "fe_lpt"   launch order of the QP workgroups / the searches: 1 longest-expected-first (default) | 0 slot order
 *   "qp_key_decay" / "fe_key_decay"   bins an ordering key loses per launch (defaults 2 / 1; 0: the key is the last time's bin)
 *   "corr_from" / "corr_max"   the short-step give-up rule of the interior point (defaults 10 / 8):
 *                  these two DO change which hard replans are given up on; they exist for that trade's A/B only
 *   "qp_profile"   1: collect the phase counters (PROFILE builds)
 *   "presolve_kernel"  1 (default): the presolve's zero-iteration certificate runs as a kernel of its own before the interior point
 *                  (qp_presolve_kernel.hip, one wave per replan) | 0: only inside qp_reg_kernel<true>, as in rounds 3-5
 * Process-wide (launcher choices, results do not depend on them): "fe_three" (keep the three-workgroup front-end instantiation),
 * "fe_xcd" (0: no XCD placement of the searches), "polish_grid" (workgroups of the polish pass, default 256).
 * Unknown names are refused (NEP_E_ARG).                                                                                      */
int nep_batch_debug_set_option(nep_batch_t* h, const char* name, int32_t value);
int nep_backend_debug_set_option(nep_backend_t* h, const char* name, int32_t value);
int nep_debug_set_global_option(const char* name, int32_t value);

/* Which kernel builds the interval hulls (same hulls, bit for bit): 0 = by batch size (eight hulls per wave from ~2 000
 * trajectories per launch on, one per wave below: DESIGN.md section 6), 1 = one hull per wave, 2 = eight per wave.       */
int nep_batch_set_hull_kernel(nep_batch_t* h, int32_t mode);

/* Test hook: the conflict matrix [N][N] of one scene as the last nep_batch_safety_commit saw it. */
int nep_batch_debug_conflicts(nep_batch_t* h, int32_t scene, uint8_t* conflict_out);

/* Average device time (ms) of the dominant kernel over the launches since the last call,
 * measured with HIP events on the launch stream; *n_launch = launches averaged.               */
int nep_batch_kernel_time(nep_batch_t* h, int32_t which, double* avg_ms, int32_t* n_launch);
int nep_batch_enable_timing(nep_batch_t* h, int32_t on);
int nep_batch_reset_timing(nep_batch_t* h);

/* Test hooks: fetch intermediates of the last replan to host.                                 */
int nep_batch_debug_hulls(nep_batch_t* h, int32_t scene, double* hull_xy, int32_t* hull_nv);
int nep_batch_debug_lines(nep_batch_t* h, int32_t slot, int32_t cap, int32_t* seg, double* nd,
                          int32_t* n_out);

/* Development aid: shader-cycle counters of the QP kernel's phases for one slot (handle must be
 * created with NEP_QP_PROFILE set in the environment). */
int nep_batch_debug_phase_cycles(nep_batch_t* h, int32_t slot, int64_t* out16);

/* Device time of the last search of every slot, microseconds (what nep_stats.solve_us is for the QP): us [slots], host.  The
 * searches of a launch are started longest-expected-first — the previous search of the same slot is the predictor, as for the
 * QP's workgroups (nep_batch_set_launch_order switches both); the results do not depend on the order.               */
int nep_batch_fe_search_us(nep_batch_t* h, float* us, int32_t cap);

/* Test hook of the entangle-aware front end (include/neptune_frontend.h: big records): what the fixed record's path accepts before a
 * child goes to a big record — list entries (<= NEP_FE_ENT_CAP), new crossings per sampled step (<= 32), bend points (<= NEP_MAX_BEND).
 * The results do not depend on these.                                                                                          */
int nep_batch_set_fe_ent_fast_caps(nep_batch_t* h, int32_t list_cap, int32_t add_cap, int32_t bend_cap);

/* Test hook: the polish pass of the last replan (nep_batch_set_polish) — how many replans were listed for it and how many it certified. */
int nep_batch_debug_polish_count(nep_batch_t* h, int32_t* listed, int32_t* certified);
/* ... and per slot: flags [slots] (host) — 0 = not listed; bit 0 / bit 1: the first / the relaxed problem's solve was left for the pass; bit 8: the
 * pass certified an optimum and wrote the slot's status, objective and trajectory (nep_stats.iters / iters_first stay the interior point's;
 * nep_stats.solve_us includes the pass's device time).                                                                                  */
int nep_batch_debug_polish_flags(nep_batch_t* h, int32_t* flags, int32_t cap);

#ifdef __cplusplus
}
#endif
#endif /* NEPTUNE_BACKEND_DEBUG_H */
