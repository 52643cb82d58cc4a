// backend.hip — host side of the C ABI declared in include/neptune_backend.h.
//
// Owns device memory, tables and launch sequencing; all arithmetic of the path runs in the HIP
// kernels (geom_kernels.hip, qp_kernels.hip).  There is no CPU fallback: without a HIP device
// every compute entry point returns NEP_E_HIP.
#include <hip/hip_runtime.h>

#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "nep_device.h"
#include "../../include/neptune_plan.h"

using namespace nep;

namespace nep { DebugGlobals g_debug; }

namespace {

thread_local std::string g_err;
int fail(int code, const std::string& msg) { g_err = msg; return code; }
#define HIPCHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) return fail(NEP_E_HIP, std::string(#x) + ": " + hipGetErrorString(e_)); } while (0)
#define HIPCHK_NULL(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { g_err = std::string(#x) + ": " + hipGetErrorString(e_); return nullptr; } } while (0)

template <class T> struct DevBuf {
  T* p = nullptr; size_t n = 0;
  int ensure(size_t count) {
    if (count <= n && p) return 0;
    if (p) hipFree(p);
    p = nullptr; n = 0;
    if (count == 0) count = 1;
    hipError_t e = hipMalloc((void**)&p, count * sizeof(T));
    if (e != hipSuccess) { g_err = std::string("hipMalloc: ") + hipGetErrorString(e); return NEP_E_HIP; }
    n = count;
    return 0;
  }
  void release() { if (p) hipFree(p); p = nullptr; n = 0; }
};

// Page-locked host memory (the per-agent handle's staging arenas: one DMA in, one out per replan).  Growth keeps the contents.
struct PinnedArena {
  char* p = nullptr; size_t n = 0;
  int ensure(size_t bytes) {
    if (bytes <= n && p) return 0;
    char* q = nullptr;
    hipError_t e = hipHostMalloc((void**)&q, bytes ? bytes : 1, hipHostMallocDefault);
    if (e != hipSuccess) { g_err = std::string("hipHostMalloc: ") + hipGetErrorString(e); return NEP_E_HIP; }
    if (p) { std::memcpy(q, p, n); hipHostFree(p); }
    if (bytes > n) std::memset(q + n, 0, bytes - n);
    p = q; n = bytes;
    return 0;
  }
  void release() { if (p) hipHostFree(p); p = nullptr; n = 0; }
};

constexpr size_t kLdsBudget = 150 * 1024;  // of the 160 KiB per CU

// The separator treats hull lists and static obstacles as counter-clockwise convex polygons (only their edges are
// candidate lines and the polygon's own rows hold by convexity, geom_kernels.hip::separator_impl) — the order CGAL's
// convex_hull_2 and setStaticObst produce.  The reference LP itself is order-independent (separator_glpk.cpp:248-373),
// so anything a caller may legally pass is brought to that form at upload: clockwise input is reversed (the first
// vertex stays first: col(0) feeds the proximity cull, solver_gurobi_poly.cpp:559), non-convex input is refused.
// Returns false for a polygon that is not convex.
bool normalize_ccw(double* xy, int n) {
  if (n < 3) return true;
  double area2 = 0, scale = 0;
  for (int v = 0; v < n; v++) {
    const double* a = xy + 2 * v; const double* b = xy + 2 * ((v + 1) % n);
    area2 += a[0] * b[1] - a[1] * b[0];
    scale = std::fmax(scale, std::fmax(std::fabs(a[0]), std::fabs(a[1])));
  }
  if (area2 < 0) for (int lo = 1, hi = n - 1; lo < hi; lo++, hi--) { std::swap(xy[2 * lo], xy[2 * hi]); std::swap(xy[2 * lo + 1], xy[2 * hi + 1]); }
  const double tol = 1e-9 * (1.0 + scale) * (1.0 + scale);
  for (int v = 0; v < n; v++) {
    const double* o = xy + 2 * v; const double* a = xy + 2 * ((v + 1) % n); const double* b = xy + 2 * ((v + 2) % n);
    if ((a[0] - o[0]) * (b[1] - o[1]) - (a[1] - o[1]) * (b[0] - o[0]) < -tol) return false;
  }
  return true;
}

// wall_clock64() ticks per second of the current device (the constant-rate counter behind nep_stats.solve_us, the launch
// order's keys and the TimeLimit emulation): 100 MHz on gfx950, asked of the runtime rather than assumed
double wall_clock_hz() {
  int dev = 0, khz = 0;
  if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, dev) == hipSuccess && khz > 0) return (double)khz * 1e3;
  return 1e8;
}

// Everything both handle kinds share: tables, scratch and the launch sequence of one replan.
struct Engine {
  SceneParams sp{};
  int n_scenes = 1;
  DevBuf<QpTable> d_tables;
  DevBuf<int> d_sched_n, d_sched_seg; DevBuf<double> d_sched_dt;
  DevBuf<double> d_pb, d_static_xy, d_static_el; DevBuf<int> d_static_nv;
  DevBuf<double> d_hull_xy, d_hull0_xy, d_bend_xy, d_line_nd, d_row_scratch;
  DevBuf<int> d_hull_nv, d_hull0_nv, d_bend_n, d_line_cnt, d_line_far, d_lp_stats;
  DevBuf<int> d_presolved; bool presolve_kernel = true;      // qp_presolve_kernel's marks (one per slot); debug option "presolve_kernel" = 0: the test runs inside qp_reg_kernel<true> as in rounds 3-5
  DevBuf<int> d_line_skip, d_redo_list, d_redo_count;   // spatial presolve: skipped LPs per segment, replans listed for the redo pass
  DevBuf<double> d_polish_z; DevBuf<int> d_polish_flag, d_polish_list, d_polish_count; bool polish = true, polish_presolve = true, last_polish_armed = false;      // the active-set polish of solves that end without the strict tests (qp_polish_kernel.hip; nep_*_set_polish)
  DevBuf<long long> d_dbg; bool profile_phases = false;
  DevBuf<int> d_flags;
  // entangle-aware front end / safety re-check (nep_batch_frontend_ent, nep_batch_safety_commit_ent)
  DevBuf<double> d_sampled, d_srep, d_slong; DevBuf<int> d_present, d_entangles;
  DevBuf<nep_fe_ent_state> d_fe_nodes, d_fe_work, d_fe_saved; DevBuf<double> d_fe_arc, d_fe_packed; bool have_reps = false;
  // big records of the entangle-aware front end (ent_device.h): a pool per handle, 0 = the default budget (4 per slot, at least 4 096)
  DevBuf<double> d_fe_big_beta, d_fe_stf; DevBuf<long long> d_fe_stvox; DevBuf<unsigned> d_fe_xpool;
  DevBuf<unsigned char> d_fe_big, d_fe_big_check; DevBuf<int> d_fe_big_count, d_fe_big_check_count; long fe_big_records = 0;
  int fe_fast_cap = NEP_FE_ENT_CAP, fe_fast_add = 32, fe_fast_bend = NEP_MAX_BEND;
  DevBuf<unsigned char> d_conflict, d_conflict_prev;
  bool safety_check_prev = false;
  int lds_lines = 0, lds_rows = 0, rows_cap = 0; size_t lds_bytes = 0;
  DevBuf<double> d_fe_box;          // boxes of the front end's obstacles, made before every search launch
  DevBuf<int> d_fe_order, d_fe_order_key; DevBuf<float> d_fe_us; bool fe_history = false, fe_lpt = true;      // the same for the front end's searches (frontend_kernel)
  DevBuf<int> d_order, d_order_key; bool have_history = false, lpt = true, last_ordered = false;   // QP workgroups launched longest-expected-first (order_kernel)
  bool use_reg = false;        // the QP runs as qp_reg_kernel (row state in registers, four workgroups per CU)
  double clock_hz = 1e8;       // wall_clock64() rate of the handle's device (set_clock)
  // (the short-step give-up rule's two numbers: run-time values for A/B, nep_*_debug_set_option "corr_from" / "corr_max"; never read from the environment)
  void set_clock() { clock_hz = wall_clock_hz(); sp.us_per_tick = 1e6 / clock_hz; sp.corr_from_it = opt_corr_from; sp.corr_max_count = opt_corr_max; if (!(sp.tol_res > 0.0)) { sp.tol_res = 1e-10; sp.tol_gap = 1e-11; sp.tol_res_inv = 1e10; sp.tol_gap_inv = 1e11; sp.tol_gap_floor = 0.1 * 1e-11; } }      // (called by both create paths: the strict tests' defaults with it)
  double sched_dc = -1, tab_T = -1, tab_w = -1; int sched_cap = 0;
  std::vector<int> h_sched_n, h_sched_seg; std::vector<double> h_sched_dt;
  // timing
  bool timing = false; std::vector<hipEvent_t> ev; size_t ev_used = 0;

  int build_tables() {
    if (tab_T == sp.T_span && tab_w == sp.weight && d_tables.p) return 0;
    std::vector<QpTable> t(2 * (kMaxK + 1));
    std::memset(t.data(), 0, t.size() * sizeof(QpTable));
    for (int mode = 0; mode < 2; mode++) for (int K = 1; K <= kMaxK; K++) build_qp_table(K, sp.T_span, sp.weight, mode, &t[mode * (kMaxK + 1) + K]);
    if (int e = d_tables.ensure(t.size())) return e;
    HIPCHK(hipMemcpy(d_tables.p, t.data(), t.size() * sizeof(QpTable), hipMemcpyHostToDevice));
    tab_T = sp.T_span; tab_w = sp.weight;
    return 0;
  }
  int build_schedule(double dc, int cap) {
    if (dc == sched_dc && cap == sched_cap && d_sched_n.p) return 0;
    h_sched_n.assign(kMaxK + 1, 0); h_sched_seg.assign((size_t)(kMaxK + 1) * cap, 0); h_sched_dt.assign((size_t)(kMaxK + 1) * cap, 0.0);
    for (int K = 1; K <= kMaxK; K++) h_sched_n[K] = build_sample_schedule(K, sp.T_span, dc, cap, &h_sched_seg[(size_t)K * cap], &h_sched_dt[(size_t)K * cap]);
    if (int e = d_sched_n.ensure(h_sched_n.size())) return e;
    if (int e = d_sched_seg.ensure(h_sched_seg.size())) return e;
    if (int e = d_sched_dt.ensure(h_sched_dt.size())) return e;
    HIPCHK(hipMemcpy(d_sched_n.p, h_sched_n.data(), h_sched_n.size() * sizeof(int), hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(d_sched_seg.p, h_sched_seg.data(), h_sched_seg.size() * sizeof(int), hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(d_sched_dt.p, h_sched_dt.data(), h_sched_dt.size() * sizeof(double), hipMemcpyHostToDevice));
    sched_dc = dc; sched_cap = cap; sp.max_states = cap; sp.dc = dc;
    return 0;
  }
  // which interior-point kernel the next replan launches, and its LDS carve (see size_scratch)
  static constexpr double kAutoCullRadius = 4.0;
  bool fits_reg = true, cull_user_set = false, skip_lps = true, no_redo = false; int lds_lines_lds = 0, sep_pack = 0, force_kernel = 0, opt_qp_key_decay = 2, opt_fe_key_decay = 1, opt_corr_from = kCorrFromItDefault, opt_corr_max = kCorrMaxCountDefault;      // (skip_lps, no_redo, sep_pack, force_kernel: development aids behind nep_*_debug_set_option, include/neptune_backend_debug.h)
  // Row scratch (rows and coefficients beyond the register slots / the LDS carve): one area per slot in general.  With the presolve's
  // redo pass (skip_mode()) the first pass never needs one — a replan whose near lines exceed the slots is sent to the redo pass —
  // so the handle keeps a pool of kScratchPool areas for that pass (config 5: 1.9 GB instead of 15.3 per 32 scenes);
  // nep_batch_reserve_row_scratch asks for the worst case.
  static constexpr int kScratchPool = 1024;
  bool scratch_full = false; int scratch_chunks = 0;
  bool static_boxes_ok = false;      // the static polygons' entries of d_fe_box are those of the polygons now uploaded (set by a fe_box_kernel launch of run())
  int lines_cap_user = 0;      // 0: the default budget; -1: the reference's worst case; n > 0: n lines per segment (nep_batch_set_line_capacity)
  bool skip_mode() const { return sp.cull_radius > 0.0 && use_reg && sp.sep_rule == 0 && sp.skip_own == 1 && sp.n_hull == sp.num_agents && skip_lps && statics_boxy && !no_redo; }
  int size_row_scratch() {
    const long slots = (long)n_scenes * sp.n_local;
    const bool pooled = skip_mode() && !scratch_full && slots > kScratchPool;
    scratch_chunks = pooled ? kScratchPool : 0;
    return d_row_scratch.ensure((size_t)(pooled ? kScratchPool : slots) * (11L * (rows_cap / 4 + 2)));
  }
  void choose_placement() {
    use_reg = fits_reg || sp.cull_radius > 0.0;
    if (force_kernel == 1) use_reg = true; else if (force_kernel == 2) use_reg = false;      // (nep_*_debug_set_option "qp_kernel": tests, A/B)
    if (use_reg) { lds_lines = NEP_MAX_POL * 8 * qp_reg_slots(); lds_rows = 4 * lds_lines; lds_bytes = qp_reg_lds_bytes(); }
    else { lds_lines = lds_lines_lds; lds_rows = 4 * lds_lines; lds_bytes = qp_lds_fixed_bytes() + (size_t)(lds_lines + 2) * 11 * 8; }   // + the dummy line of the padded row groups
  }
  // sizes the scratch for (n_scenes x n_local) slots with sp.n_hull hull lists per scene
  int size_scratch() {
    const int N = sp.num_agents, np = sp.num_pol;
    const long slots = (long)n_scenes * sp.n_local;
    // Lines a segment's bucket holds.  The reference's worst case is one line per other agent's hull, per base, per static and per
    // (agent, bend segment) pair of the entangle rows — n_hull + N + S + 8 N.  The last term is by far the largest (2 048 of 2 660 at
    // config 5) and an entangle line needs an ACTIVE case for that agent and segment, which a tenth of the pairs have in the
    // survey's scenes: by default the buckets budget 2 N entangle lines per segment (config 5: 1 124 instead of 2 660, 1.8 GB
    // instead of 4.2 per 32 scenes).  A segment that gets more is flagged (nep_batch_check: NEP_E_CAP), never written past its
    // bucket; nep_batch_set_line_capacity(h, -1) restores the worst case, (h, n) sets n.
    const int worst_lines = sp.n_hull + N + sp.n_static + (sp.ent_enabled ? N * kBend : 0);
    if ((long)NEP_MAX_POL * worst_lines > 65535) return fail(NEP_E_CAP, "more than 65535 separator candidates per agent");
    sp.lines_cap = lines_cap_user < 0 ? worst_lines : (lines_cap_user > 0 ? lines_cap_user : sp.n_hull + N + sp.n_static + (sp.ent_enabled ? std::min(N * kBend, std::max(2 * N, 64)) : 0));
    if (sp.lines_cap > worst_lines) sp.lines_cap = worst_lines;
    if (sp.lines_cap < 8) sp.lines_cap = 8;
    const long lines_total = (long)NEP_MAX_POL * sp.lines_cap;
    // LDS carve of the QP kernel: 11 doubles per line (n1, n2, h + 4 x (s, lambda)).  The worst case
    // (every base and every obstacle close to every segment) almost never happens, so the carve is
    // sized for the expected count and the rest spills to global scratch; when the expectation
    // fits half a CU's LDS two workgroups share a CU.
    const size_t fixed = qp_lds_fixed_bytes();
    const long per_line = 11 * 8;
    const long l_half = ((long)(160 * 1024 / 2) - (long)fixed) / per_line - 2;
    const long l_full = ((long)kLdsBudget - (long)fixed) / per_line - 2;
    // expected lines per segment: every other agent's hull, plus a few nearby bases / statics / entangle lines
    const long expect = (long)NEP_MAX_POL * (sp.n_hull + 4 + sp.n_static / 8 + (sp.ent_enabled ? sp.num_agents / 8 : 0));
    long ll = expect <= l_half ? l_half : l_full;
    if (ll > lines_total) ll = lines_total;
    lds_lines_lds = (int)((ll + 1) & ~1L);
    // Placement of the row state.  When the expected lines per segment fit the register slots of qp_reg_kernel (8 per slot;
    // a few more only cost a trip through the global scratch) the row state lives in registers and four workgroups share a
    // CU.  Bigger problems (config 5: ~260 lines per segment) get the verified line presolve by default (kAutoCullRadius,
    // nep_batch_set_line_cull): the few dozen near lines fit the register slots, the parked ones are checked at the solution,
    // and only a replan that violates one is solved again with every row, the rows beyond the slots going through the per-slot
    // global scratch.  With the presolve turned off (radius 0) such problems keep the LDS placement of qp_kernel.
    // nep_*_debug_set_option(h, "qp_kernel", 1 | 2) overrides the choice (tests, A/B).
    fits_reg = expect / NEP_MAX_POL <= 8L * qp_reg_slots() + 8;
    // The verified line presolve is the handle's default at EVERY size (round 6; until then only where the lines did not fit the
    // register slots): it is what the reference's solver does inside optimize() (Gurobi's presolve drops redundant rows before the
    // barrier, solver_gurobi_poly.cpp:823), its result is the full problem's optimum by construction (parked lines and skipped LPs
    // are verified at the solution, a replan that fails the verification is solved again with every row), and the polish pass runs
    // under it by default, so that one handle has one optimum whatever path its solves take.  nep_batch_set_line_cull(h, 0) turns it off.
    if (!cull_user_set) sp.cull_radius = kAutoCullRadius;
    sp.fe_key_decay = opt_fe_key_decay; sp.qp_key_decay = opt_qp_key_decay;      // (nep_*_debug_set_option "qp_key_decay" / "fe_key_decay": A/B)
    // (the keys remember — a new key is the maximum of the measured bin and the old key less a decay — so a fresh buffer starts at zero;
    // only a fresh one: the per-agent handle sizes its scratch on every replan and a memset there was 30 us of its 100)
    if (lpt) { if (int e = d_order.ensure((size_t)slots)) return e; const int* was = d_order_key.p; if (int e = d_order_key.ensure((size_t)slots)) return e; if (d_order_key.p != was) hipMemset(d_order_key.p, 0, d_order_key.n * sizeof(int)); }
    if (lpt) { if (int e = d_fe_order.ensure((size_t)slots)) return e; const int* was = d_fe_order_key.p; if (int e = d_fe_order_key.ensure((size_t)slots)) return e; if (d_fe_order_key.p != was) hipMemset(d_fe_order_key.p, 0, d_fe_order_key.n * sizeof(int)); }
    if (int e = d_fe_us.ensure((size_t)slots)) return e;
    choose_placement();
    rows_cap = 4 * (int)lines_total; rows_cap = (rows_cap + 3) & ~3;
    if (int e = d_hull_xy.ensure((size_t)n_scenes * sp.n_hull * np * kHullV * 2)) return e;
    if (int e = d_fe_box.ensure((size_t)n_scenes * (N + (sp.n_static > 0 ? sp.n_static : 0)) * np * 4)) return e;
    static_boxes_ok = false;      // (the buffer may be a new one)
    if (int e = d_hull_nv.ensure((size_t)n_scenes * sp.n_hull * np)) return e;
    if (int e = d_hull0_xy.ensure((size_t)n_scenes * N * np * 2)) return e;
    if (int e = d_hull0_nv.ensure((size_t)n_scenes * N * np)) return e;
    if (int e = d_bend_xy.ensure((size_t)n_scenes * N * kBend * 2)) return e;
    if (int e = d_bend_n.ensure((size_t)n_scenes * N)) return e;
    if (int e = d_line_nd.ensure((size_t)slots * lines_total * 3)) return e;
    if (int e = d_line_cnt.ensure((size_t)slots * NEP_MAX_POL)) return e;
    if (int e = d_line_far.ensure((size_t)slots * NEP_MAX_POL)) return e;
    if (int e = d_lp_stats.ensure((size_t)slots * NEP_MAX_POL * 2)) return e;   // per (slot, segment): LPs attempted, LPs without a line
    if (int e = d_line_skip.ensure((size_t)slots * NEP_MAX_POL)) return e;
    if (int e = d_redo_list.ensure((size_t)slots)) return e;
    if (int e = d_presolved.ensure((size_t)slots)) return e;
    if (!d_redo_count.p) { if (int e = d_redo_count.ensure(4)) return e; HIPCHK(hipMemset(d_redo_count.p, 0, 4 * sizeof(int))); }      // [0] listed replans, [1] parked line violated, [2] moved beyond the radius
    if (!d_flags.p) { if (int e = d_flags.ensure(1)) return e; HIPCHK(hipMemset(d_flags.p, 0, sizeof(int))); }
    if (int e = d_polish_z.ensure((size_t)slots * 2 * 24)) return e;
    if (int e = d_polish_flag.ensure((size_t)slots)) return e;
    if (int e = d_polish_list.ensure((size_t)slots)) return e;
    if (!d_polish_count.p) { if (int e = d_polish_count.ensure(8)) return e; HIPCHK(hipMemset(d_polish_count.p, 0, 8 * sizeof(int))); }
    if (!profile_phases) profile_phases = getenv("NEP_QP_PROFILE") != nullptr;      // (profiling only: the per-phase cycle counters of make PROFILE=1; also nep_*_debug_set_option "qp_profile")
    if (profile_phases) { if (int e = d_dbg.ensure((size_t)slots * 32)) return e; }      // (the QP kernels use 16 per slot, the front end 32)
    if (int e = size_row_scratch()) return e;
    return 0;
  }
  void fill(ProblemSet& ps) {
    ps.pb = d_pb.p; ps.static_xy = d_static_xy.p; ps.static_nv = d_static_nv.p; ps.static_el = d_static_el.p;
    ps.hull_xy = d_hull_xy.p; ps.hull_nv = d_hull_nv.p; ps.hull0_xy = d_hull0_xy.p; ps.hull0_nv = d_hull0_nv.p;
    ps.bend_xy = d_bend_xy.p; ps.bend_n = d_bend_n.p;
    ps.fe_order = nullptr; ps.fe_order_key = (lpt && fe_lpt && d_fe_order_key.n >= (size_t)n_scenes * (size_t)sp.n_local) ? d_fe_order_key.p : nullptr; ps.fe_us = d_fe_us.p;
    ps.line_nd = d_line_nd.p; ps.line_cnt = d_line_cnt.p; ps.lp_stats = d_lp_stats.p;
    ps.line_far = sp.cull_radius > 0.0 ? d_line_far.p : nullptr;
    // LPs whose line is known to be far without solving them are skipped when the presolve is on, the rule is the largest gap
    // (box far => line far holds for that vertex only), the hull lists are the batch's (one per agent: the boxes are indexed
    // by agent) and the interior point is the register kernel (the one that verifies them): see separator_body / qp_reg_kernel
    const bool skip = sp.cull_radius > 0.0 && use_reg && sp.sep_rule == 0 && sp.skip_own == 1 && sp.n_hull == sp.num_agents && skip_lps && statics_boxy;
    ps.scratch_chunks = (skip && !no_redo) ? scratch_chunks : 0; ps.scratch_by_block = 0;
    ps.skip_box = skip ? d_fe_box.p : nullptr; ps.line_skip = skip ? d_line_skip.p : nullptr;
    ps.redo_list = skip ? d_redo_list.p : nullptr; ps.redo_count = skip ? d_redo_count.p : nullptr; ps.order_count = nullptr;
    ps.sep_pack = sep_pack;
    ps.row_scratch = d_row_scratch.p; ps.rows_cap = rows_cap; ps.lds_rows = lds_rows; ps.lds_lines = lds_lines;
    ps.dbg = profile_phases ? d_dbg.p : nullptr;
    ps.presolved = nullptr;      // (set by run() for a launch sequence in which qp_presolve_kernel goes first)
    ps.flags = d_flags.p;
    ps.fe_box = d_fe_box.p;
    // the polish pass finishes what the register kernel leaves.  Under the presolve only on request (nep_batch_set_polish(h, 2): on the
    // near lines, and a certified point goes through the presolve's verification of the parked lines and the skipped LPs again —
    // polish_slot): a pass over a handful of slots is 0.03-0.06 ms, 8 % of a presolved step
    const bool pol = polish && use_reg && (polish_presolve || !(sp.cull_radius > 0.0)) && d_polish_z.p != nullptr;
    ps.polish_z = pol ? d_polish_z.p : nullptr; ps.polish_flag = pol ? d_polish_flag.p : nullptr;
    ps.polish_list = pol ? d_polish_list.p : nullptr; ps.polish_count = pol ? d_polish_count.p : nullptr;
    last_polish_armed = pol;
  }
  // packs n polygons into the fixed-stride device layout (vertices, vertex counts, edge lengths)
  bool statics_boxy = true;      // every static polygon uploaded so far has an edge on each side of its bounding box (see pack_statics)
  int pack_statics(int n, const int32_t* off, const double* xy, std::vector<double>& sx, std::vector<int>& nv, std::vector<double>& el) {
    sx.assign((size_t)(n > 0 ? n : 1) * kHullV * 2, 0.0); nv.assign(n > 0 ? n : 1, 0);
    for (int j = 0; j < n; j++) {
      int c = off[j + 1] - off[j];
      if (c > kHullV) return fail(NEP_E_CAP, "static obstacle with more than NEP_HULL_MAX_V vertices");
      if (c < 0) return fail(NEP_E_ARG, "static obstacle offsets must not decrease");
      nv[j] = c;
      for (int v = 0; v < c; v++) { sx[((size_t)j * kHullV + v) * 2] = xy[2 * (off[j] + v)]; sx[((size_t)j * kHullV + v) * 2 + 1] = xy[2 * (off[j] + v) + 1]; }
      if (!normalize_ccw(&sx[(size_t)j * kHullV * 2], c)) return fail(NEP_E_ARG, "static obstacle polygon is not convex");
      // The spatial presolve skips the LP of an obstacle whose BOX is far (box_far, geom_kernels.hip): sound only when every side of
      // the box carries an edge of the polygon — true of inflated statics and interval hulls (hulls of axis-aligned squares), not of
      // an arbitrary convex polygon (a diamond's nearest edge line can lie at 0.71 of the box distance).  One polygon without that
      // property turns LP skipping off for the handle (parked lines and the zero-iteration test stay: they evaluate real lines).
      if (c > 0) {
        const double* q = &sx[(size_t)j * kHullV * 2];
        double lo[2] = {q[0], q[1]}, hi[2] = {q[0], q[1]};
        for (int v = 1; v < c; v++) for (int a = 0; a < 2; a++) { lo[a] = std::min(lo[a], q[2 * v + a]); hi[a] = std::max(hi[a], q[2 * v + a]); }
        const double tol = 1e-9 * (1.0 + (hi[0] - lo[0]) + (hi[1] - lo[1]));
        for (int a = 0; a < 2; a++) {
          int n_lo = 0, n_hi = 0;
          for (int v = 0; v < c; v++) { n_lo += q[2 * v + a] - lo[a] <= tol; n_hi += hi[a] - q[2 * v + a] <= tol; }
          if (n_lo < 2 || n_hi < 2) statics_boxy = false;
        }
      }
    }
    // edge lengths for the proximity cull (solver_gurobi_poly.cpp:566-572), the same IEEE operations the kernel used to
    // repeat per candidate: squares, one sum, sqrt (no contraction: three separate roundings)
    el.assign(sx.size() / 2, 0.0);
    for (int j = 0; j < n; j++)
      for (int v = 0; v + 1 < nv[j]; v++) {
        const double* q = &sx[((size_t)j * kHullV + v) * 2];
        volatile double ex = q[2] - q[0], ey = q[3] - q[1];
        volatile double a = ex * ex, b = ey * ey;
        volatile double c = a + b;
        el[(size_t)j * kHullV + v] = std::sqrt(c);
      }
    return 0;
  }
  int upload_statics(int n, const int32_t* off, const double* xy) {
    static_boxes_ok = false;
    std::vector<double> sx, el; std::vector<int> nv;
    statics_boxy = true;       // (a new shared set replaces every earlier polygon)
    if (int e = pack_statics(n, off, xy, sx, nv, el)) return e;
    if (int e = d_static_xy.ensure(sx.size())) return e;
    if (int e = d_static_el.ensure(el.size())) return e;
    if (int e = d_static_nv.ensure(nv.size())) return e;
    HIPCHK(hipMemcpy(d_static_el.p, el.data(), el.size() * sizeof(double), hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(d_static_xy.p, sx.data(), sx.size() * sizeof(double), hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(d_static_nv.p, nv.data(), nv.size() * sizeof(int), hipMemcpyHostToDevice));
    sp.n_static = n; sp.static_stride = 0;
    if (n_scenes > 0 && sp.num_agents > 0 && sp.num_pol > 0)      // (the front end's obstacle boxes: one per agent or static polygon and interval)
      if (int e = d_fe_box.ensure((size_t)n_scenes * (sp.num_agents + n) * sp.num_pol * 4)) return e;
    return 0;
  }
  // One static-obstacle set per scene (same polygon count S in every scene): the first call replicates the handle's
  // shared set into [n_scenes][S] arrays, then scene `scene` gets its own polygons.
  int upload_scene_statics(int scene, int n, const int32_t* off, const double* xy) {
    static_boxes_ok = false;
    const int S = sp.n_static;
    if (n != S) return fail(NEP_E_ARG, "every scene must have the handle's n_static polygons");
    if (S == 0) return 0;
    std::vector<double> sx, el; std::vector<int> nv;
    if (int e = pack_statics(n, off, xy, sx, nv, el)) return e;
    if (sp.static_stride == 0) {
      DevBuf<double> nxy, nel; DevBuf<int> nnv;
      if (int e = nxy.ensure((size_t)n_scenes * S * kHullV * 2)) return e;
      if (int e = nel.ensure((size_t)n_scenes * S * kHullV)) return e;
      if (int e = nnv.ensure((size_t)n_scenes * S)) return e;
      for (int s = 0; s < n_scenes; s++) {
        HIPCHK(hipMemcpy(nxy.p + (size_t)s * S * kHullV * 2, d_static_xy.p, (size_t)S * kHullV * 2 * sizeof(double), hipMemcpyDeviceToDevice));
        HIPCHK(hipMemcpy(nel.p + (size_t)s * S * kHullV, d_static_el.p, (size_t)S * kHullV * sizeof(double), hipMemcpyDeviceToDevice));
        HIPCHK(hipMemcpy(nnv.p + (size_t)s * S, d_static_nv.p, (size_t)S * sizeof(int), hipMemcpyDeviceToDevice));
      }
      d_static_xy.release(); d_static_el.release(); d_static_nv.release();
      d_static_xy = nxy; d_static_el = nel; d_static_nv = nnv;
      sp.static_stride = S;
      // the entangle check's representatives (nep_batch_set_static_reps) are indexed like the polygons: one set per scene from
      // now on.  A set uploaded while the statics were shared is replicated; every scene can then be given its own.
      if (have_reps && d_srep.p && d_srep.n >= (size_t)S * 4 && d_slong.n >= (size_t)S * 2) {
        DevBuf<double> nr, nl;
        if (int e = nr.ensure((size_t)n_scenes * S * 4)) return e;
        if (int e = nl.ensure((size_t)n_scenes * S * 2)) return e;
        for (int s = 0; s < n_scenes; s++) {
          HIPCHK(hipMemcpy(nr.p + (size_t)s * S * 4, d_srep.p, (size_t)S * 4 * sizeof(double), hipMemcpyDeviceToDevice));
          HIPCHK(hipMemcpy(nl.p + (size_t)s * S * 2, d_slong.p, (size_t)S * 2 * sizeof(double), hipMemcpyDeviceToDevice));
        }
        d_srep.release(); d_slong.release(); d_srep = nr; d_slong = nl;
      } else have_reps = false;
    }
    HIPCHK(hipMemcpy(d_static_xy.p + (size_t)scene * S * kHullV * 2, sx.data(), (size_t)S * kHullV * 2 * sizeof(double), hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(d_static_el.p + (size_t)scene * S * kHullV, el.data(), (size_t)S * kHullV * sizeof(double), hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(d_static_nv.p + (size_t)scene * S, nv.data(), (size_t)S * sizeof(int), hipMemcpyHostToDevice));
    return 0;
  }
  hipEvent_t next_event() {
    if (ev_used == ev.size()) { hipEvent_t e; hipEventCreate(&e); ev.push_back(e); }
    return ev[ev_used++];
  }
  // separator + QP (+ hulls when recs != nullptr) on `st`
  // phases: 1 = the geometry half (hulls, boxes, separating lines into the handle's scratch), 2 = the QP half on the lines that are there,
  // 3 = both (nep_batch_replan); the halves of one round may be enqueued on different streams, ordered by the caller (nep_batch_replan_lines)
  int run(const nep_traj_rec* d_recs, int n_rec, ProblemSet& ps, hipStream_t st, int phases = 3) {
    const int slots = n_scenes * sp.n_local;
    SampleSched sc{d_sched_n.p, d_sched_seg.p, d_sched_dt.p};
    const bool geo = (phases & 1) != 0, qp = (phases & 2) != 0;
    if (timing) hipEventRecord(next_event(), st);
    // the hull kernel makes the hulls' boxes itself (and zeroes the redo counters) when it is the eight-hulls-per-wave kernel over one hull list
    // per agent and the static polygons' boxes are in place from an earlier fe_box_kernel launch: one launch less per round
    const bool fused_boxes = d_recs && geo && ps.skip_box != nullptr && !ps.lines_override && static_boxes_ok && ps.hull_pb <= 0
                             && n_rec == sp.num_agents && sp.n_hull == sp.num_agents && hulls_grouped(sp, n_scenes, n_rec);
    // ... and, in one wave more, this round's launch order of the QP workgroups (order_kernel's counting sort: it only needs the previous
    // round's measured times), when the same call goes on to the QP half
    const bool want_order = qp && lpt && have_history && slots > 1024 && d_order_key.n >= (size_t)slots && d_order.n >= (size_t)slots;
    const bool fused_order = fused_boxes && want_order;
    if (fused_order) { ps.order = d_order.p; ps.order_key = d_order_key.p; }
    if (d_recs && geo) launch_hulls(d_recs, n_scenes, n_rec, ps.guess, sp, ps, st, fused_boxes);
    if (timing) hipEventRecord(next_event(), st);
    if (ps.lines_override) { ps.skip_box = nullptr; ps.line_skip = nullptr; ps.redo_list = nullptr; ps.redo_count = nullptr; }
    if (scratch_chunks > 0 && ps.scratch_chunks == 0) {      // (a pooled handle asked for a replan without the redo pass — lines from the host, a rule or hull layout that cannot skip LPs: one area per slot after all)
      scratch_full = true;
      HIPCHK(hipStreamSynchronize(st));
      if (int e = size_row_scratch()) return e;
      ps.row_scratch = d_row_scratch.p;
    }
    const bool skip = ps.skip_box != nullptr;
    if (!ps.lines_override && geo) {
      if (skip && !fused_boxes) { launch_boxes(n_scenes, sp, ps, st); static_boxes_ok = true; }      // (zeroes the redo counters as well)
      launch_separator(slots, sp, ps, st);
    }
    if (timing) hipEventRecord(next_event(), st);
    if (!qp) { if (timing) hipEventRecord(next_event(), st); HIPCHK(hipGetLastError()); return 0; }
    ps.order = nullptr; last_ordered = false;
    ps.order_key = (lpt && d_order_key.n >= (size_t)slots) ? d_order_key.p : nullptr;
    if (ps.order_key && have_history && slots > 1024 && d_order.n >= (size_t)slots) {   // (more than one wave of workgroups)
      if (!fused_order) launch_qp_order(slots, d_order_key.p, d_order.p, st, ps.polish_count);      // (zeroes the polish pass's counters on its way; fused_order: the hull launch has done both)
      ps.order = d_order.p; last_ordered = true;
    } else if (ps.polish_count && !(use_reg && slots == 1)) launch_qp_polish_zero(ps.polish_count, st);      // (a one-workgroup launch — the per-agent handle — sets the counters itself: qp_reg_kernel's last lines)
    // the presolve's zero-iteration certificate as a kernel of its own, one wave per replan: the replans it finishes (nine in ten of the
    // bench's scenes) cost the interior-point launch an immediate return (qp_presolve_kernel.hip)
    if (use_reg && presolve_kernel && ps.line_far != nullptr && !ps.lines_override && d_presolved.n >= (size_t)slots) {
      ps.presolved = d_presolved.p;
      launch_qp_presolve(slots, sp, ps, d_tables.p, sc, d_presolved.p, st);
    }
    if (use_reg) launch_qp_reg(slots, sp, ps, d_tables.p, sc, lds_bytes, st);
    else launch_qp(slots, sp, ps, d_tables.p, sc, lds_bytes, st);
    if (skip && !no_redo) {      // (NEP_SEP_NO_REDO, read in size_scratch: development aid — the flagged replans keep their presolved result for inspection)
      // the presolve's redo pass: replans whose solution did not verify the skipped / parked lines (listed by the kernel above; the
      // list is empty nearly always) get every LP solved and every row through the interior point
      launch_separator_redo(slots, sp, ps, st);
      ProblemSet pr = ps;
      pr.line_far = nullptr; pr.line_skip = nullptr; pr.order = d_redo_list.p; pr.order_count = d_redo_count.p;
      pr.scratch_by_block = ps.scratch_chunks > 0 ? 1 : 0;
      launch_qp_reg(slots, sp, pr, d_tables.p, sc, lds_bytes, st);
    }
    if (ps.polish_list) launch_qp_polish(slots, sp, ps, d_tables.p, sc, st);      // (the slots the QP kernel listed: nearly always none — workgroups beyond the count return at once)
    have_history = ps.order_key != nullptr;
    if (timing) hipEventRecord(next_event(), st);
    HIPCHK(hipGetLastError());
    return 0;
  }
  void release() {
    d_tables.release(); d_sched_n.release(); d_sched_seg.release(); d_sched_dt.release(); d_pb.release(); d_static_xy.release();
    d_static_nv.release(); d_static_el.release(); d_hull_xy.release(); d_hull0_xy.release(); d_bend_xy.release(); d_line_nd.release(); d_row_scratch.release(); d_order.release(); d_order_key.release(); d_fe_order.release(); d_fe_order_key.release(); d_fe_us.release(); d_fe_box.release();
    d_sampled.release(); d_srep.release(); d_slong.release(); d_present.release(); d_entangles.release(); d_fe_nodes.release(); d_fe_work.release(); d_fe_saved.release(); d_fe_arc.release(); d_fe_packed.release(); d_fe_big.release(); d_fe_big_beta.release(); d_fe_stf.release(); d_fe_stvox.release(); d_fe_xpool.release(); d_fe_big_count.release(); d_fe_big_check.release(); d_fe_big_check_count.release();
    d_presolved.release(); d_line_skip.release(); d_redo_list.release(); d_redo_count.release(); d_polish_z.release(); d_polish_flag.release(); d_polish_list.release(); d_polish_count.release(); d_flags.release(); d_conflict.release(); d_conflict_prev.release(); d_hull_nv.release(); d_hull0_nv.release(); d_bend_n.release(); d_line_cnt.release(); d_line_far.release(); d_lp_stats.release();
    for (auto e : ev) hipEventDestroy(e);
    ev.clear();
  }
};

bool have_device() { int n = 0; return hipGetDeviceCount(&n) == hipSuccess && n > 0; }


__global__ void sample_kernel(const nep_solution* __restrict__ sol, int K, const int* __restrict__ seg, const double* __restrict__ dt, int ns, double* __restrict__ out) {
  const int s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= ns) return;
  const int i = seg[s]; const double d = dt[s];
  double* st = out + (long)s * NEP_STATE_DOUBLES;
  for (int ax = 0; ax < 3; ax++) {  // solver_gurobi_poly.cpp:921-929
    const double* c = sol->coeff[ax][i];
    st[ax] = ((c[0] * (d * d * d) + c[1] * (d * d)) + c[2] * d) + c[3];
    st[3 + ax] = (c[0] * (3 * d * d) + c[1] * (2 * d)) + c[2];
    st[6 + ax] = c[0] * (6 * d) + c[1] * 2;
    st[9 + ax] = c[0] * 6;
  }
}

// Test hook behind both debug line readers: the lines of one slot in row order — per segment the lines at the front of the
// bucket (all of them, or with the presolve on the near ones) and then the parked far ones, which the separator writes from
// the back of the bucket in order of appearance.  LPs without a separating line left (0, 0, 0) and are skipped.
int read_lines(Engine& E, size_t slot, int n_seg, int32_t cap, int32_t* seg, double* nd, int32_t* n_out) {
  std::vector<int> cnt(NEP_MAX_POL), far(NEP_MAX_POL, 0); std::vector<double> buf((size_t)NEP_MAX_POL * E.sp.lines_cap * 3);
  HIPCHK(hipMemcpy(cnt.data(), E.d_line_cnt.p + slot * NEP_MAX_POL, cnt.size() * sizeof(int), hipMemcpyDeviceToHost));
  if (E.sp.cull_radius > 0.0) HIPCHK(hipMemcpy(far.data(), E.d_line_far.p + slot * NEP_MAX_POL, far.size() * sizeof(int), hipMemcpyDeviceToHost));
  HIPCHK(hipMemcpy(buf.data(), E.d_line_nd.p + slot * NEP_MAX_POL * E.sp.lines_cap * 3, buf.size() * sizeof(double), hipMemcpyDeviceToHost));
  for (int& c : cnt) c = line_count(c);      // (an overflowed bucket's count is stored as -1 - n)
  int n = 0;
  for (int s = 0; s < n_seg && s < NEP_MAX_POL; s++)
    for (int c = 0; c < cnt[s] + far[s]; c++) {
      const size_t q = c < cnt[s] ? (size_t)c : (size_t)E.sp.lines_cap - 1 - (size_t)(c - cnt[s]);
      const double* e = &buf[((size_t)s * E.sp.lines_cap + q) * 3];
      if (e[0] == 0.0 && e[1] == 0.0 && e[2] == 0.0) continue;   // LP without a separating line
      if (n < cap) { seg[n] = s; for (int k = 0; k < 3; k++) nd[3 * n + k] = e[k]; }
      n++;
    }
  *n_out = n;
  return 0;
}

}  // namespace

// =================================================================================================
// per-agent handle
// =================================================================================================
// One replan through this handle is ONE host-to-device copy, two kernels (separator, QP with generatePwpOut's samples in its
// tail) and ONE device-to-host copy: the setters write straight into a page-locked arena whose layout the device mirrors,
//   [guess][hull vertex counts: cap x num_pol][entangle block: cases, col(0) of the uninflated hulls, bend points][hull vertices: used],
// (the copy ends with the vertices of the hull lists actually set; the entangle block travels only after setEntStateVector),
// and the solution comes back together with its sampled states.  neptune.cpp:1504-1528 is the call sequence this serves.
struct InLayout { size_t guess, nv, cas, h0xy, h0nv, bend, bendn, xy, cap_end; };
static InLayout in_layout(int N, int np, int cap) {
  auto up = [](size_t v) { return (v + 255) & ~(size_t)255; };
  InLayout L{};
  size_t o = 0;
  L.guess = o; o = up(o + sizeof(nep_guess));
  L.nv = o; o = up(o + (size_t)cap * np * sizeof(int));
  L.cas = o; o = up(o + (size_t)NEP_MAX_POL * N * sizeof(int));
  L.h0xy = o; o = up(o + (size_t)N * np * 2 * sizeof(double));
  L.h0nv = o; o = up(o + (size_t)N * np * sizeof(int));
  L.bend = o; o = up(o + (size_t)N * kBend * 2 * sizeof(double));
  L.bendn = o; o = up(o + (size_t)N * sizeof(int));
  L.xy = o; o = up(o + (size_t)cap * np * kHullV * 2 * sizeof(double));
  L.cap_end = o;
  return L;
}
struct nep_backend {
  Engine eng;
  nep_backend_cfg cfg{};
  std::vector<double> pb;
  double max_runtime = 0.05, tether = 0, j_max = 0;
  bool have_bounds = false, have_init = false, have_hulls = false, solved = false;
  nep_guess guess{}; std::vector<double> guess_times;
  int n_obst = 0;
  PinnedArena hin, hout; DevBuf<char> d_in, d_out;
  int in_cap = 0; InLayout L{};                 // hull lists the arenas are laid out for
  bool have_h0 = false, have_ent = false, states_fresh = false;
  int override_n = -1; std::vector<int> ov_seg; std::vector<double> ov_nd;
  nep_solution h_sol{}; nep_stats stats{};
  hipStream_t stream = nullptr;
  // (re)lays the input arena out for `cap` hull lists, keeping what the setters have written
  int lay_out(int cap) {
    const int N = cfg.num_agents, np = cfg.num_pol;
    if (cap <= in_cap && hin.p) return 0;
    if (cap < 8) cap = 8;
    const InLayout nl = in_layout(N, np, cap);
    PinnedArena fresh;
    if (int e = fresh.ensure(nl.cap_end)) return e;
    std::memset(fresh.p, 0, nl.cap_end);
    if (hin.p) {
      std::memcpy(fresh.p + nl.guess, hin.p + L.guess, sizeof(nep_guess));
      std::memcpy(fresh.p + nl.nv, hin.p + L.nv, (size_t)in_cap * np * sizeof(int));
      std::memcpy(fresh.p + nl.cas, hin.p + L.cas, L.xy - L.cas);                    // the whole entangle block (same sizes: they depend on N only)
      std::memcpy(fresh.p + nl.xy, hin.p + L.xy, (size_t)in_cap * np * kHullV * 2 * sizeof(double));
      hin.release();
    }
    hin = fresh; L = nl; in_cap = cap;
    return d_in.ensure(nl.cap_end);
  }
  template <class T> T* in(size_t off) { return (T*)(hin.p + off); }
};

extern "C" {

const char* nep_last_error(void) { return g_err.c_str(); }
const char* nep_version(void) { return "neptune-amd-backend 0.1 (gfx950)"; }

nep_backend_t* nep_backend_create(const nep_backend_cfg* cfg) {
  if (!cfg || !cfg->pb) { g_err = "null cfg"; return nullptr; }
  if (cfg->deg_pol != 3) { g_err = "deg_pol must be 3 (reference yaml: only 3 is supported)"; return nullptr; }
  if (!cfg->use_linear_constraints) { g_err = "use_linear_constraints=false (bilinear variant) is out of scope"; return nullptr; }
  if (cfg->num_pol < 1 || cfg->num_pol > NEP_MAX_POL) { g_err = "num_pol out of range"; return nullptr; }
  if (cfg->id < 1 || cfg->id > cfg->num_agents) { g_err = "id out of range"; return nullptr; }
  if (!have_device()) { g_err = "no HIP device: the back end has no CPU path"; return nullptr; }
  nep_backend* h = new nep_backend();
  h->cfg = *cfg; h->pb.assign(cfg->pb, cfg->pb + 2 * cfg->num_agents); h->cfg.pb = nullptr;
  Engine& E = h->eng;
  E.sp.num_agents = cfg->num_agents; E.sp.num_pol = cfg->num_pol; E.sp.n_static = 0; E.sp.n_hull = 0; E.sp.ent_enabled = 0;
  E.sp.n_local = 1; E.sp.first_local = cfg->id - 1; E.sp.skip_own = 0; E.sp.T_span = cfg->T_span; E.sp.weight = cfg->weight_term;
  E.sp.drone_radius = 0; E.n_scenes = 1;
  E.lines_cap_user = -1;      // (one replan at a time: the buckets are sized for the reference's worst case, there is no flag to poll)
  E.set_clock();
  HIPCHK_NULL(hipStreamCreate(&h->stream));
  if (h->lay_out(cfg->num_agents > 8 ? cfg->num_agents : 8)) { delete h; return nullptr; }
  if (E.d_pb.ensure(h->pb.size())) { delete h; return nullptr; }
  HIPCHK_NULL(hipMemcpy(E.d_pb.p, h->pb.data(), h->pb.size() * sizeof(double), hipMemcpyHostToDevice));
  int32_t off0[1] = {0};
  if (E.upload_statics(0, off0, nullptr)) { delete h; return nullptr; }
  if (E.build_tables()) { delete h; return nullptr; }
  return h;
}

void nep_backend_destroy(nep_backend_t* h) {
  if (!h) return;
  h->eng.release(); h->d_in.release(); h->d_out.release(); h->hin.release(); h->hout.release();
  if (h->stream) hipStreamDestroy(h->stream);
  delete h;
}

int nep_backend_set_max_values(nep_backend_t* h, double x_min, double x_max, double y_min, double y_max, double z_min,
                               double z_max, double v_max, double a_max, double j_max) {
  if (!h) return fail(NEP_E_ARG, "null handle");
  SceneParams& sp = h->eng.sp;
  sp.mins[0] = x_min; sp.mins[1] = y_min; sp.mins[2] = z_min; sp.maxs[0] = x_max; sp.maxs[1] = y_max; sp.maxs[2] = z_max;
  sp.v_max = v_max; sp.a_max = a_max; h->j_max = j_max;
  sp.long_length = std::sqrt((x_max - x_min) * (x_max - x_min) + (y_max - y_min) * (y_max - y_min));
  h->have_bounds = true;
  return 0;
}
int nep_backend_set_max_runtime(nep_backend_t* h, double s) {
  if (!h) return fail(NEP_E_ARG, "null handle");
  h->max_runtime = s;
  h->eng.sp.time_limit_ticks = s > 0 ? (long long)(s * h->eng.clock_hz) : 0;
  return 0;
}
int nep_backend_set_tether_length(nep_backend_t* h, double t) { if (!h) return fail(NEP_E_ARG, "null handle"); h->tether = t; return 0; }

int nep_backend_set_static_obst_vert(nep_backend_t* h, int32_t n, const int32_t* off, const double* xy) {
  if (!h || n < 0 || (n > 0 && (!off || !xy))) return fail(NEP_E_ARG, "bad static obstacle arguments");
  int32_t off0[1] = {0};
  return h->eng.upload_statics(n, n ? off : off0, xy);
}

int nep_backend_set_init_trajectory(nep_backend_t* h, const nep_pwp* p) {
  if (!h || !p) return fail(NEP_E_ARG, "null argument");
  if (p->n_seg < 1 || p->n_seg > h->cfg.num_pol) return fail(NEP_E_CAP, "initial trajectory must have 1..num_pol intervals");
  std::memset(&h->guess, 0, sizeof(h->guess));
  h->guess.K = p->n_seg; h->guess.t_start = 0.0;
  for (int ax = 0; ax < 3; ax++) for (int i = 0; i < p->n_seg; i++) for (int j = 0; j < 4; j++) h->guess.coeff[ax][i][j] = p->coeff[ax][i][j];
  h->guess_times.assign(p->times, p->times + p->n_seg + 1);
  *h->in<nep_guess>(h->L.guess) = h->guess;
  h->have_init = true; h->solved = false; h->states_fresh = false;
  return 0;
}

int nep_backend_set_hulls(nep_backend_t* h, int32_t n_obst, const int32_t* off, const double* xy) {
  if (!h || n_obst < 0 || (n_obst > 0 && (!off || !xy))) return fail(NEP_E_ARG, "bad hull arguments");
  const int np = h->cfg.num_pol;
  if (int e = h->lay_out(n_obst)) return e;
  int* nv = h->in<int>(h->L.nv); double* hx = h->in<double>(h->L.xy);
  for (int p = 0; p < n_obst * np; p++) {
    const int c = off[p + 1] - off[p];
    if (c > kHullV) return fail(NEP_E_CAP, "hull with more than NEP_HULL_MAX_V vertices");
    if (c < 0) return fail(NEP_E_ARG, "hull offsets must not decrease");
    nv[p] = c;
    double* q = hx + (size_t)p * kHullV * 2;
    std::memcpy(q, xy + 2 * (size_t)off[p], (size_t)c * 2 * sizeof(double));
    if (c < kHullV) std::memset(q + 2 * c, 0, (size_t)(kHullV - c) * 2 * sizeof(double));
    if (!normalize_ccw(q, c)) return fail(NEP_E_ARG, "hull polygon is not convex");
  }
  h->n_obst = n_obst; h->have_hulls = true;
  return 0;
}

int nep_backend_set_hulls_no_inflation(nep_backend_t* h, int32_t n_agents, const int32_t* off, const double* xy) {
  if (!h || n_agents != h->cfg.num_agents || !off) return fail(NEP_E_ARG, "hullsNoInflation must be indexed by agent id (num_agents lists)");
  const int np = h->cfg.num_pol;
  double* x0 = h->in<double>(h->L.h0xy); int* n0 = h->in<int>(h->L.h0nv);
  for (int p = 0; p < n_agents * np; p++) {
    const int c = off[p + 1] - off[p];
    n0[p] = c;
    if (c > 0) { x0[(size_t)p * 2] = xy[2 * off[p]]; x0[(size_t)p * 2 + 1] = xy[2 * off[p] + 1]; }  // only col(0) is read (:722-734)
    else { x0[(size_t)p * 2] = 0.0; x0[(size_t)p * 2 + 1] = 0.0; }
  }
  h->have_h0 = true;
  return 0;
}

int nep_backend_set_ent_state_vector(nep_backend_t* h, const nep_ent_view* e) {
  if (!h) return fail(NEP_E_ARG, "null handle");
  if (!e) { h->have_ent = false; return 0; }
  const int N = h->cfg.num_agents;
  if (e->n_active < N) return fail(NEP_E_ARG, "active_cases rows shorter than num_agents");
  int* cas = h->in<int>(h->L.cas);
  std::memset(cas, 0, (size_t)NEP_MAX_POL * N * sizeof(int));
  // case id of (knot i, agent j): solver_gurobi_poly.cpp:624-631 (last matching alpha wins)
  for (int i = 0; i < e->n_states && i < NEP_MAX_POL; i++)
    for (int j = 0; j < N; j++) {
      if (e->active_cases[(size_t)i * e->n_active + j] != 1) continue;
      int cid = 0;
      for (int a = e->alpha_off[i]; a < e->alpha_off[i + 1]; a++) if (e->alphas[2 * a] == j + 1) cid = e->alphas[2 * a + 1];
      cas[(size_t)i * N + j] = cid;
    }
  double* bend = h->in<double>(h->L.bend); int* bend_n = h->in<int>(h->L.bendn);
  std::memset(bend, 0, (size_t)N * kBend * 2 * sizeof(double));
  for (int j = 0; j < N; j++) {
    const int nb = e->bend_off[j + 1] - e->bend_off[j];
    if (nb > kBend) return fail(NEP_E_CAP, "more than NEP_MAX_BEND bend points");
    bend_n[j] = nb;
    for (int b = 0; b < nb; b++) { bend[((size_t)j * kBend + b) * 2] = e->bend_xy[2 * (e->bend_off[j] + b)]; bend[((size_t)j * kBend + b) * 2 + 1] = e->bend_xy[2 * (e->bend_off[j] + b) + 1]; }
  }
  h->have_ent = true;
  return 0;
}

int nep_backend_debug_set_lines(nep_backend_t* h, int32_t n, const int32_t* seg, const double* nd) {
  if (!h) return fail(NEP_E_ARG, "null handle");
  h->override_n = n;
  if (n > 0) { h->ov_seg.assign(seg, seg + n); h->ov_nd.assign(nd, nd + 3 * n); } else { h->ov_seg.clear(); h->ov_nd.clear(); }
  return 0;
}

int nep_backend_optimize(nep_backend_t* h, double* objective_value) {
  if (!h) return fail(NEP_E_ARG, "null handle");
  if (!h->have_bounds) return fail(NEP_E_STATE, "optimize before setMaxValues");
  if (!h->have_init) return fail(NEP_E_STATE, "optimize before setInitTrajectory");
  Engine& E = h->eng;
  const int N = h->cfg.num_agents, np = h->cfg.num_pol;
  E.sp.n_hull = h->have_hulls ? h->n_obst : 0;
  E.sp.ent_enabled = h->have_ent ? 1 : 0;
  {
    // the samples generatePwpOut returns are made by the QP kernel's tail at the schedule's dc: the schedule must hold every state of the
    // horizon (ceil(num_pol T_span / dc) + 1, as generatePwpOut's own path sizes it) — a fixed cap of 128 cut trajectories longer than
    // 6.3 s short of what the reference's time walk returns (solver_gurobi_poly.cpp:911-934)
    const double dc0 = E.sched_dc > 0 ? E.sched_dc : 0.05;
    const int need = (int)std::ceil(np * E.sp.T_span / dc0) + 3;
    if (int e = E.build_schedule(dc0, need)) return e;
  }
  if (h->override_n >= 0) {  // line buckets sized for the override
    int per_seg[NEP_MAX_POL] = {0}; for (int l = 0; l < h->override_n; l++) { int s = h->ov_seg[l]; if (s < 0 || s >= NEP_MAX_POL) return fail(NEP_E_ARG, "override line segment out of range"); per_seg[s]++; }
    int mx = 8; for (int s = 0; s < NEP_MAX_POL; s++) if (per_seg[s] > mx) mx = per_seg[s];
    E.sp.n_hull = mx;  // only used to size lines_cap below
  }
  if (int e = E.size_scratch()) return e;
  if (h->override_n >= 0) E.sp.n_hull = 0;
  const auto t_host0 = std::chrono::steady_clock::now();
  ProblemSet ps{};
  E.fill(ps);
  const InLayout& L = h->L;
  const bool with_hulls = h->have_hulls && h->n_obst > 0 && h->override_n < 0;
  // one copy in: the guess, the vertex counts and — the arena's tail — the vertices of the hull lists in use; with the entangle
  // rows the block between them goes along (one contiguous range from the start), without them the two ends travel separately
  // only when the block in between is bigger than what it would cost to carry it
  const size_t xy_used = with_hulls ? (size_t)h->n_obst * np * kHullV * 2 * sizeof(double) : 0;
  if (h->have_ent) {
    if (!h->have_h0) return fail(NEP_E_STATE, "setEntStateVector without setHullsNoInflation");
    HIPCHK(hipMemcpyAsync(h->d_in.p, h->hin.p, L.xy + xy_used, hipMemcpyHostToDevice, h->stream));
  } else {
    HIPCHK(hipMemcpyAsync(h->d_in.p, h->hin.p, L.cas, hipMemcpyHostToDevice, h->stream));
    if (xy_used) HIPCHK(hipMemcpyAsync(h->d_in.p + L.xy, h->hin.p + L.xy, xy_used, hipMemcpyHostToDevice, h->stream));
  }
  const size_t out_states = (sizeof(nep_solution) + 255) & ~(size_t)255;
  const size_t out_bytes = out_states + (size_t)E.sp.max_states * NEP_STATE_DOUBLES * sizeof(double);
  if (int e = h->d_out.ensure(out_bytes)) return e;
  if (int e = h->hout.ensure(out_bytes)) return e;
  ps.guess = (const nep_guess*)(h->d_in.p + L.guess); ps.solution = (nep_solution*)h->d_out.p;
  ps.states = (double*)(h->d_out.p + out_states); ps.commit = nullptr; ps.case_id = nullptr;
  if (with_hulls) { ps.hull_xy = (double*)(h->d_in.p + L.xy); ps.hull_nv = (int*)(h->d_in.p + L.nv); }
  if (h->have_ent) {
    ps.case_id = (const int*)(h->d_in.p + L.cas);
    ps.hull0_xy = (double*)(h->d_in.p + L.h0xy); ps.hull0_nv = (int*)(h->d_in.p + L.h0nv);
    ps.bend_xy = (double*)(h->d_in.p + L.bend); ps.bend_n = (int*)(h->d_in.p + L.bendn);
  }
  ps.lines_override = 0;
  if (h->override_n >= 0) {
    std::vector<double> nd((size_t)NEP_MAX_POL * E.sp.lines_cap * 3, 0.0); std::vector<int> cnt(NEP_MAX_POL, 0);
    for (int l = 0; l < h->override_n; l++) { int s = h->ov_seg[l]; int c = cnt[s]++; for (int k = 0; k < 3; k++) nd[((size_t)s * E.sp.lines_cap + c) * 3 + k] = h->ov_nd[3 * l + k]; }
    HIPCHK(hipMemcpyAsync(E.d_line_nd.p, nd.data(), nd.size() * sizeof(double), hipMemcpyHostToDevice, h->stream));
    HIPCHK(hipMemcpyAsync(E.d_line_cnt.p, cnt.data(), cnt.size() * sizeof(int), hipMemcpyHostToDevice, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));  // host vectors go out of scope
    ps.lines_override = 1;
  }
  if (int e = E.run(nullptr, 0, ps, h->stream)) return e;
  // one copy out: the solution and generatePwpOut's samples at the schedule's dc (the QP kernel's tail wrote them)
  HIPCHK(hipMemcpyAsync(h->hout.p, h->d_out.p, out_bytes, hipMemcpyDeviceToHost, h->stream));
  HIPCHK(hipStreamSynchronize(h->stream));
  h->h_sol = *(const nep_solution*)h->hout.p;
  h->stats = h->h_sol.stats;
  h->stats.solve_us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t_host0).count();   // what the caller's clock around optimize() sees (neptune.cpp:1504,1528), set-up checks excluded
  h->solved = true; h->states_fresh = true;
  if (h->stats.status != NEP_FAILED && objective_value) *objective_value = h->stats.objective;  // :882 (untouched on failure)
  return h->stats.status;
}

int nep_backend_generate_pwp_out(nep_backend_t* h, double t_start, double dc, nep_pwp* pwp_out, double* states_out,
                                 int32_t states_cap, int32_t* n_states_out) {
  if (!h || !pwp_out) return fail(NEP_E_ARG, "null argument");
  if (!h->have_init) return fail(NEP_E_STATE, "generatePwpOut before setInitTrajectory");
  const int K = h->guess.K;
  std::memset(pwp_out, 0, sizeof(*pwp_out));
  pwp_out->n_seg = K;
  for (int i = 0; i <= K; i++) pwp_out->times[i] = h->guess_times[i] + t_start;  // :898
  const double (*co)[NEP_MAX_POL][4] = h->solved ? h->h_sol.coeff : h->guess.coeff;   // pwp_out_ = pwp_init_ until solved
  for (int ax = 0; ax < 3; ax++) for (int i = 0; i < K; i++) for (int j = 0; j < 4; j++) pwp_out->coeff[ax][i][j] = co[ax][i][j];
  if (n_states_out) *n_states_out = 0;
  if (states_out && states_cap > 0 && h->solved && h->states_fresh && dc == h->eng.sched_dc) {
    // the samples came back with the solution (the QP kernel's tail made them at this dc): no device work here
    int ns = h->h_sol.n_states; if (ns > states_cap) ns = states_cap;
    const size_t out_states = (sizeof(nep_solution) + 255) & ~(size_t)255;
    std::memcpy(states_out, h->hout.p + out_states, (size_t)ns * NEP_STATE_DOUBLES * sizeof(double));
    if (n_states_out) *n_states_out = ns;
    return 0;
  }
  if (states_out && states_cap > 0) {
    Engine& E = h->eng;
    DevBuf<nep_solution> d_sol_tmp; DevBuf<double> d_states_tmp;
    struct Rel { DevBuf<nep_solution>& a; DevBuf<double>& b; ~Rel() { a.release(); b.release(); } } rel{d_sol_tmp, d_states_tmp};
    if (int e = d_sol_tmp.ensure(1)) return e;
    {  // another dc than the schedule's, or no solve yet: stage the trajectory to sample (the solution, else the guess)
      nep_solution s{}; s.K = K; std::memcpy(s.coeff, h->solved ? h->h_sol.coeff : h->guess.coeff, sizeof(s.coeff));
      HIPCHK(hipMemcpy(d_sol_tmp.p, &s, sizeof(s), hipMemcpyHostToDevice));
    }
    int cap = (int)std::ceil(h->cfg.num_pol * h->cfg.T_span / dc) + 3;
    if (int e = E.build_schedule(dc, cap)) return e;
    int ns = E.h_sched_n[K]; if (ns > states_cap) ns = states_cap;
    if (int e = d_states_tmp.ensure((size_t)cap * NEP_STATE_DOUBLES)) return e;
    hipLaunchKernelGGL(sample_kernel, dim3((ns + 63) / 64), dim3(64), 0, h->stream, d_sol_tmp.p, K, E.d_sched_seg.p + (size_t)K * cap, E.d_sched_dt.p + (size_t)K * cap, ns, d_states_tmp.p);
    HIPCHK(hipMemcpyAsync(states_out, d_states_tmp.p, (size_t)ns * NEP_STATE_DOUBLES * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    h->states_fresh = false;      // (the schedule now belongs to this dc: the next optimize samples at it)
    if (n_states_out) *n_states_out = ns;
  }
  return 0;
}

// Measurement aid (bench.py's per_agent_api leg): the drop-in call sequence of one replan, as Neptune::replanFull issues it
// (neptune.cpp:1514-1527: setInitTrajectory -> setHulls -> setHullsNoInflation -> setEntStateVector -> optimize ->
// generatePwpOut), n_iter times from one host thread with the caller's buffers; us_out[i] = wall time of iteration i as a C++
// caller's clock around the six calls sees it, us_optimize_out[i] (may be NULL) = of optimize() alone.  h0_off / ent may be
// NULL (entangle check off).  Returns the status of the last optimize, or < 0.
int nep_backend_debug_time_sequence(nep_backend_t* h, const nep_pwp* init, int32_t n_obst, const int32_t* hull_off, const double* hull_xy,
                                    const int32_t* h0_off, const double* h0_xy, const nep_ent_view* ent, double t_start, double dc,
                                    int32_t n_iter, double* us_out, double* us_optimize_out) {
  if (!h || !init || n_iter < 1 || !us_out) return fail(NEP_E_ARG, "bad arguments");
  const int cap = (int)std::ceil(h->cfg.num_pol * h->cfg.T_span / dc) + 3;
  std::vector<double> states((size_t)cap * NEP_STATE_DOUBLES);
  nep_pwp out; int32_t ns = 0; int status = 0;
  for (int it = 0; it < n_iter; it++) {
    const auto t0 = std::chrono::steady_clock::now();
    if (int e = nep_backend_set_init_trajectory(h, init)) return e;
    if (int e = nep_backend_set_hulls(h, n_obst, hull_off, hull_xy)) return e;
    if (h0_off) if (int e = nep_backend_set_hulls_no_inflation(h, h->cfg.num_agents, h0_off, h0_xy)) return e;
    if (int e = nep_backend_set_ent_state_vector(h, ent)) return e;
    const auto t1 = std::chrono::steady_clock::now();
    double obj = 0.0;
    status = nep_backend_optimize(h, &obj);
    if (status < 0) return status;
    const auto t2 = std::chrono::steady_clock::now();
    if (int e = nep_backend_generate_pwp_out(h, t_start, dc, &out, states.data(), cap, &ns)) return e;
    const auto t3 = std::chrono::steady_clock::now();
    us_out[it] = std::chrono::duration<double, std::micro>(t3 - t0).count();
    if (us_optimize_out) us_optimize_out[it] = std::chrono::duration<double, std::micro>(t2 - t1).count();
  }
  return status;
}

int nep_backend_get_stats(nep_backend_t* h, nep_stats* out) { if (!h || !out) return fail(NEP_E_ARG, "null argument"); *out = h->stats; return 0; }

int nep_backend_debug_get_lines(nep_backend_t* h, int32_t cap, int32_t* seg, double* nd, int32_t* n_out) {
  if (!h || !n_out) return fail(NEP_E_ARG, "null argument");
  return read_lines(h->eng, 0, h->guess.K, cap, seg, nd, n_out);
}

// =================================================================================================
// stand-alone kernels
// =================================================================================================
int nep_separator_batch_rule(int32_t rule, int32_t n_prob, const int32_t* a_off, const double* a_xy, const int32_t* b_off, const double* b_xy,
                             double* nd_out, int32_t* solved_out) {
  if (rule != 0 && rule != 1) return fail(NEP_E_ARG, "separator rule: 0 largest gap, 1 GLPK-class simplex");
  if (n_prob < 0 || !a_off || !b_off || !nd_out || !solved_out) return fail(NEP_E_ARG, "bad arguments");
  if (!have_device()) return fail(NEP_E_HIP, "no HIP device: the back end has no CPU path");
  if (n_prob == 0) return 0;
  for (int p = 0; p < n_prob; p++) {
    if (b_off[p + 1] - b_off[p] != 4) return fail(NEP_E_ARG, "set B must be the 4 control points of a segment");
    if (a_off[p + 1] - a_off[p] > kHullV) return fail(NEP_E_CAP, "set A has more than NEP_HULL_MAX_V points");
  }
  DevBuf<int> da, db, ds; DevBuf<double> dax, dbx, dnd;
  const int na = a_off[n_prob], nb = b_off[n_prob];
  int e = 0;
  if ((e = da.ensure(n_prob + 1)) || (e = db.ensure(n_prob + 1)) || (e = ds.ensure(n_prob)) || (e = dax.ensure((size_t)2 * na)) || (e = dbx.ensure((size_t)2 * nb)) || (e = dnd.ensure((size_t)3 * n_prob))) return e;
  HIPCHK(hipMemcpy(da.p, a_off, (n_prob + 1) * sizeof(int), hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(db.p, b_off, (n_prob + 1) * sizeof(int), hipMemcpyHostToDevice));
  if (na) HIPCHK(hipMemcpy(dax.p, a_xy, (size_t)2 * na * sizeof(double), hipMemcpyHostToDevice));
  if (nb) HIPCHK(hipMemcpy(dbx.p, b_xy, (size_t)2 * nb * sizeof(double), hipMemcpyHostToDevice));
  launch_separator_explicit(n_prob, da.p, dax.p, db.p, dbx.p, dnd.p, ds.p, rule, nullptr);
  HIPCHK(hipGetLastError());
  HIPCHK(hipMemcpy(nd_out, dnd.p, (size_t)3 * n_prob * sizeof(double), hipMemcpyDeviceToHost));
  HIPCHK(hipMemcpy(solved_out, ds.p, (size_t)n_prob * sizeof(int), hipMemcpyDeviceToHost));
  da.release(); db.release(); ds.release(); dax.release(); dbx.release(); dnd.release();
  return 0;
}

int nep_separator_batch(int32_t n_prob, const int32_t* a_off, const double* a_xy, const int32_t* b_off, const double* b_xy,
                        double* nd_out, int32_t* solved_out) {
  return nep_separator_batch_rule(0, n_prob, a_off, a_xy, b_off, b_xy, nd_out, solved_out);
}

int nep_gjk_batch(int32_t n_prob, const int32_t* a_off, const double* a_xy, const double* b_xy, int32_t* hit_out) {
  if (n_prob < 0 || !a_off || !b_xy || !hit_out) return fail(NEP_E_ARG, "bad arguments");
  if (!have_device()) return fail(NEP_E_HIP, "no HIP device: the back end has no CPU path");
  if (n_prob == 0) return 0;
  DevBuf<int> da, dh; DevBuf<double> dax, dbx;
  const int na = a_off[n_prob];
  int e = 0;
  if ((e = da.ensure(n_prob + 1)) || (e = dh.ensure(n_prob)) || (e = dax.ensure((size_t)2 * (na > 0 ? na : 1))) || (e = dbx.ensure((size_t)8 * n_prob))) return e;
  HIPCHK(hipMemcpy(da.p, a_off, (n_prob + 1) * sizeof(int), hipMemcpyHostToDevice));
  if (na) HIPCHK(hipMemcpy(dax.p, a_xy, (size_t)2 * na * sizeof(double), hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(dbx.p, b_xy, (size_t)8 * n_prob * sizeof(double), hipMemcpyHostToDevice));
  launch_gjk_explicit(n_prob, da.p, dax.p, dbx.p, dh.p, nullptr);
  HIPCHK(hipGetLastError());
  HIPCHK(hipMemcpy(hit_out, dh.p, (size_t)n_prob * sizeof(int), hipMemcpyDeviceToHost));
  da.release(); dh.release(); dax.release(); dbx.release();
  return 0;
}

int nep_hulls_batch(int32_t n_traj, const nep_traj_rec* trajs, double t_start, int32_t num_pol, double T_span, double drone_radius,
                    double* hull_xy, int32_t* hull_nv, double* hull0_xy, int32_t* hull0_nv) {
  if (n_traj < 0 || !trajs || !hull_xy || !hull_nv || !hull0_xy || !hull0_nv || num_pol < 1) return fail(NEP_E_ARG, "bad arguments");
  if (!have_device()) return fail(NEP_E_HIP, "no HIP device: the back end has no CPU path");
  if (n_traj == 0) return 0;
  DevBuf<nep_traj_rec> dr; DevBuf<double> dh, dh0; DevBuf<int> dn, dn0, dfl;
  const size_t np = (size_t)n_traj * num_pol;
  int e = 0;
  if ((e = dr.ensure(n_traj)) || (e = dh.ensure(np * kHullV * 2)) || (e = dh0.ensure(np * kHullV * 2)) || (e = dn.ensure(np)) || (e = dn0.ensure(np)) || (e = dfl.ensure(1))) return e;
  HIPCHK(hipMemset(dfl.p, 0, sizeof(int)));
  HIPCHK(hipMemcpy(dr.p, trajs, (size_t)n_traj * sizeof(nep_traj_rec), hipMemcpyHostToDevice));
  HIPCHK(hipMemset(dh.p, 0, np * kHullV * 2 * sizeof(double))); HIPCHK(hipMemset(dh0.p, 0, np * kHullV * 2 * sizeof(double)));
  launch_hulls_explicit(dr.p, n_traj, t_start, num_pol, T_span, drone_radius, dh.p, dn.p, dh0.p, dn0.p, dfl.p, nullptr);
  HIPCHK(hipGetLastError());
  HIPCHK(hipMemcpy(hull_xy, dh.p, np * kHullV * 2 * sizeof(double), hipMemcpyDeviceToHost));
  HIPCHK(hipMemcpy(hull0_xy, dh0.p, np * kHullV * 2 * sizeof(double), hipMemcpyDeviceToHost));
  HIPCHK(hipMemcpy(hull_nv, dn.p, np * sizeof(int), hipMemcpyDeviceToHost));
  HIPCHK(hipMemcpy(hull0_nv, dn0.p, np * sizeof(int), hipMemcpyDeviceToHost));
  int flags = 0;
  HIPCHK(hipMemcpy(&flags, dfl.p, sizeof(int), hipMemcpyDeviceToHost));
  dr.release(); dh.release(); dh0.release(); dn.release(); dn0.release(); dfl.release();
  if (flags & NEP_FLAG_HULL_OVERFLOW) return fail(NEP_E_CAP, "an interval overlaps more than NEP_HULL_MAX_CP/4 committed segments (or its hull has more than NEP_HULL_MAX_V vertices)");
  return 0;
}

}  // extern "C"

// =================================================================================================
// batched handle
// =================================================================================================
struct nep_batch {
  Engine eng;
  nep_batch_cfg cfg{};
  int slots = 0;
  const nep_traj_rec* fe_committed = nullptr;   // the records nep_batch_frontend built this round's hulls from
  int ent_ns = 3;                               // num_sample_per_interval the hull blocks reserve room for (nep_batch_set_ent_samples; yaml: 3)
};

extern "C" {

nep_batch_t* nep_batch_create(const nep_batch_cfg* c) {
  if (!c || !c->pb) { g_err = "null cfg"; return nullptr; }
  if (c->num_pol < 1 || c->num_pol > NEP_MAX_POL || c->num_agents < 1 || c->n_local < 1 || c->first_local < 0 ||
      c->first_local + c->n_local > c->num_agents || c->n_scenes < 1 || c->max_states < 1) { g_err = "bad batch configuration"; return nullptr; }
  if (!have_device()) { g_err = "no HIP device: the back end has no CPU path"; return nullptr; }
  nep_batch* h = new nep_batch();
  h->cfg = *c; h->cfg.pb = nullptr; h->cfg.static_off = nullptr; h->cfg.static_xy = nullptr;
  Engine& E = h->eng; SceneParams& sp = E.sp;
  sp.num_agents = c->num_agents; sp.num_pol = c->num_pol; sp.n_hull = c->num_agents; sp.ent_enabled = c->enable_entangle;
  sp.n_local = c->n_local; sp.first_local = c->first_local; sp.skip_own = 1;
  sp.T_span = c->T_span; sp.weight = c->weight_term; sp.drone_radius = c->drone_radius;
  sp.mins[0] = c->x_min; sp.mins[1] = c->y_min; sp.mins[2] = c->z_min; sp.maxs[0] = c->x_max; sp.maxs[1] = c->y_max; sp.maxs[2] = c->z_max;
  sp.v_max = c->v_max; sp.a_max = c->a_max;
  sp.long_length = std::sqrt((c->x_max - c->x_min) * (c->x_max - c->x_min) + (c->y_max - c->y_min) * (c->y_max - c->y_min));
  E.n_scenes = c->n_scenes; h->slots = c->n_scenes * c->n_local;
  E.set_clock();
  bool ok = true;
  ok = ok && !E.d_pb.ensure((size_t)2 * c->num_agents);
  if (ok) ok = hipMemcpy(E.d_pb.p, c->pb, (size_t)2 * c->num_agents * sizeof(double), hipMemcpyHostToDevice) == hipSuccess;
  int32_t off0[1] = {0};
  if (ok) ok = !E.upload_statics(c->n_static, c->n_static ? c->static_off : off0, c->static_xy);
  if (ok) ok = !E.build_tables();
  if (ok) ok = !E.build_schedule(c->dc, c->max_states);
  if (ok) ok = !E.size_scratch();
  if (!ok) { if (g_err.empty()) g_err = "batch setup failed"; E.release(); delete h; return nullptr; }
  return h;
}

void nep_batch_destroy(nep_batch_t* h) { if (!h) return; h->eng.release(); delete h; }

int64_t nep_batch_ent_bytes(const nep_batch_t* h) { return h ? (int64_t)h->slots * NEP_MAX_POL * h->cfg.num_agents * sizeof(int32_t) : 0; }

int nep_batch_replan(nep_batch_t* h, const nep_traj_rec* d_committed, const nep_guess* d_guess, const void* d_ent,
                     nep_solution* d_solution, double* d_states, nep_traj_rec* d_commit, void* stream) {
  if (!h || !d_guess || !d_solution) return fail(NEP_E_ARG, "null argument");
  Engine& E = h->eng;
  ProblemSet ps{};
  E.fill(ps);
  ps.guess = d_guess; ps.solution = d_solution; ps.states = d_states; ps.commit = d_commit;
  ps.prev_commit = d_committed ? d_committed : h->fe_committed;   // (hulls reused from nep_batch_frontend: its records are this round's previous ones)
  ps.case_id = (E.sp.ent_enabled && d_ent) ? (const int*)d_ent : nullptr;
  ps.lines_override = 0;
  // d_committed == NULL: the interval hulls of this round are already in the handle's scratch (nep_batch_frontend)
  return E.run(d_committed, h->cfg.num_agents, ps, (hipStream_t)stream);
}

// The two halves of nep_batch_replan as calls of their own (include/neptune_backend.h): hulls + separating lines, then the QP on them
int nep_batch_replan_lines(nep_batch_t* h, const nep_traj_rec* d_committed, const nep_guess* d_guess, const void* d_ent, void* stream) {
  if (!h || !d_guess || !d_committed) return fail(NEP_E_ARG, "null argument");
  Engine& E = h->eng;
  ProblemSet ps{};
  E.fill(ps);
  ps.guess = d_guess;
  ps.case_id = (E.sp.ent_enabled && d_ent) ? (const int*)d_ent : nullptr;
  ps.lines_override = 0;
  return E.run(d_committed, h->cfg.num_agents, ps, (hipStream_t)stream, 1);
}
int nep_batch_replan_solve(nep_batch_t* h, const nep_traj_rec* d_committed, const nep_guess* d_guess, const void* d_ent,
                           nep_solution* d_solution, double* d_states, nep_traj_rec* d_commit, void* stream) {
  if (!h || !d_guess || !d_solution || !d_committed) return fail(NEP_E_ARG, "null argument");
  Engine& E = h->eng;
  ProblemSet ps{};
  E.fill(ps);
  ps.guess = d_guess; ps.solution = d_solution; ps.states = d_states; ps.commit = d_commit;
  ps.prev_commit = d_committed;
  ps.case_id = (E.sp.ent_enabled && d_ent) ? (const int*)d_ent : nullptr;
  ps.lines_override = 0;
  return E.run(d_committed, h->cfg.num_agents, ps, (hipStream_t)stream, 2);
}

// ---- sharded hulls: a rank computes the interval hulls of its own agents' committed trajectories
// into one block, the blocks of all ranks are all-gathered, and every rank runs separator + QP
// against the gathered blocks (kernels address them through hull_ref, nep_device.h) -------------
namespace {
struct HullBlock { size_t xy, nv, xy0, nv0, bend, bend_n, samp, present, bytes; };
// ent_ns > 0 (handle created with enable_entangle): the block also carries what the entangle check reads of a committed
// trajectory — its ent_ns + 1 samples per interval (Neptune::SamplePointsOfIntervals) and whether it exists — so that the
// entangle-aware front end and the safety pass's re-check run against gathered blocks like everything else
HullBlock hull_block_layout(int n_scenes, int per, int np, int ent_ns = 0) {
  auto up = [](size_t v) { return (v + 255) & ~(size_t)255; };
  HullBlock b{};
  const size_t e = (size_t)n_scenes * per;
  size_t o = 0;
  b.xy = o; o = up(o + e * np * kHullV * 2 * sizeof(double));
  b.nv = o; o = up(o + e * np * sizeof(int));
  b.xy0 = o; o = up(o + e * np * 2 * sizeof(double));
  b.nv0 = o; o = up(o + e * np * sizeof(int));
  b.bend = o; o = up(o + e * kBend * 2 * sizeof(double));
  b.bend_n = o; o = up(o + e * sizeof(int));
  b.samp = o; b.present = o;
  if (ent_ns > 0) { o = up(o + e * np * (ent_ns + 1) * 2 * sizeof(double)); b.present = o; o = up(o + e * sizeof(int)); }
  b.bytes = o;
  return b;
}
HullBlock block_of(const nep_batch* h);
void point_at_block(ProblemSet& ps, const HullBlock& b, void* base) {
  char* p = (char*)base;
  ps.hull_xy = (double*)(p + b.xy); ps.hull_nv = (int*)(p + b.nv); ps.hull0_xy = (double*)(p + b.xy0); ps.hull0_nv = (int*)(p + b.nv0);
  ps.bend_xy = (double*)(p + b.bend); ps.bend_n = (int*)(p + b.bend_n);
}
}  // namespace

namespace { HullBlock block_of(const nep_batch* h) { return hull_block_layout(h->cfg.n_scenes, h->cfg.n_local, h->cfg.num_pol, h->cfg.enable_entangle ? h->ent_ns : 0); } }

int64_t nep_batch_hull_block_bytes(const nep_batch_t* h) {
  return h ? (int64_t)block_of(h).bytes : 0;
}

int nep_batch_hulls(nep_batch_t* h, const nep_traj_rec* d_committed_local, const nep_guess* d_guess, void* d_block, void* stream) {
  if (!h || !d_committed_local || !d_guess || !d_block) return fail(NEP_E_ARG, "null argument");
  Engine& E = h->eng;
  const HullBlock b = block_of(h);
  ProblemSet ps{};
  E.fill(ps);
  point_at_block(ps, b, d_block);
  ps.guess = d_guess;
  launch_hulls(d_committed_local, h->cfg.n_scenes, h->cfg.n_local, d_guess, E.sp, ps, (hipStream_t)stream);
  if (h->cfg.enable_entangle)      // what the entangle check reads of my agents' trajectories travels in the same block
    launch_ent_sample(d_committed_local, h->cfg.n_scenes, h->cfg.n_local, &d_guess->t_start, (long)sizeof(nep_guess) * h->cfg.n_local, h->cfg.num_pol, h->ent_ns,
                      E.sp.T_span, (double*)((char*)d_block + b.samp), (int*)((char*)d_block + b.present), (hipStream_t)stream);
  HIPCHK(hipGetLastError());
  return 0;
}

int nep_batch_set_ent_samples(nep_batch_t* h, int32_t ns) {
  if (!h || ns < 1 || ns > 8) return fail(NEP_E_ARG, "ent_samples out of range (1..8)");
  h->ent_ns = ns;
  return 0;
}

int nep_batch_replan_hulls(nep_batch_t* h, const void* d_blocks, int32_t n_blocks, const nep_guess* d_guess, const void* d_ent,
                           nep_solution* d_solution, double* d_states, nep_traj_rec* d_commit, void* stream) {
  if (!h || !d_blocks || !d_guess || !d_solution) return fail(NEP_E_ARG, "null argument");
  if (n_blocks < 1 || n_blocks * h->cfg.n_local != h->cfg.num_agents) return fail(NEP_E_ARG, "n_blocks * n_local must equal num_agents");
  Engine& E = h->eng;
  const HullBlock b = block_of(h);
  ProblemSet ps{};
  E.fill(ps);
  point_at_block(ps, b, const_cast<void*>(d_blocks));
  ps.hull_pb = h->cfg.n_local; ps.hull_bstride = (long)b.bytes;
  ps.hull_pb_magic = (h->cfg.n_local > 0 && h->cfg.num_agents < 65536) ? (1ull << 32) / (unsigned long long)h->cfg.n_local + 1ull : 0ull;
  ps.guess = d_guess; ps.solution = d_solution; ps.states = d_states; ps.commit = d_commit;
  ps.case_id = (E.sp.ent_enabled && d_ent) ? (const int*)d_ent : nullptr;
  ps.lines_override = 0;
  return E.run(nullptr, 0, ps, (hipStream_t)stream);
}

// hulls -> front-end beam search; the guesses land where nep_batch_replan reads them
int nep_batch_frontend(nep_batch_t* h, const nep_fe_cfg* cfg, const nep_traj_rec* d_committed, const nep_fe_start* d_start,
                       nep_guess* d_guess, nep_fe_result* d_result, void* stream) {
  if (!h || !cfg || !d_committed || !d_start || !d_guess) return fail(NEP_E_ARG, "null argument");
  if (cfg->num_samples < 2 || cfg->num_samples > NEP_FE_MAX_SAMPLES || cfg->beam_width < 1 || cfg->beam_width > NEP_FE_MAX_BEAM ||
      !(cfg->voxel_size > 0.0) || !(cfg->j_max > 0.0)) return fail(NEP_E_ARG, "bad front-end configuration");
  Engine& E = h->eng;
  ProblemSet ps{};
  E.fill(ps);
  h->fe_committed = d_committed;
  launch_hulls_ts(d_committed, h->cfg.n_scenes, h->cfg.num_agents, &d_start->t_start, (long)sizeof(nep_fe_start), E.sp, ps, (hipStream_t)stream);
  launch_frontend(h->slots, E.sp, ps, *cfg, d_start, d_guess, d_result, nullptr, (hipStream_t)stream, E.lpt ? E.d_fe_order.p : nullptr, E.fe_history);
  E.fe_history = true;
  HIPCHK(hipGetLastError());
  return 0;
}

// the same against all-gathered hull blocks (multi-GPU rounds: nep_batch_hulls -> all-gather -> this -> nep_batch_replan_hulls)
int nep_batch_frontend_hulls(nep_batch_t* h, const nep_fe_cfg* cfg, const void* d_blocks, int32_t n_blocks, const nep_fe_start* d_start,
                             nep_guess* d_guess, nep_fe_result* d_result, void* stream) {
  if (!h || !cfg || !d_blocks || !d_start || !d_guess) return fail(NEP_E_ARG, "null argument");
  if (n_blocks < 1 || n_blocks * h->cfg.n_local != h->cfg.num_agents) return fail(NEP_E_ARG, "n_blocks * n_local must equal num_agents");
  if (cfg->num_samples < 2 || cfg->num_samples > NEP_FE_MAX_SAMPLES || cfg->beam_width < 1 || cfg->beam_width > NEP_FE_MAX_BEAM ||
      !(cfg->voxel_size > 0.0) || !(cfg->j_max > 0.0)) return fail(NEP_E_ARG, "bad front-end configuration");
  Engine& E = h->eng;
  const HullBlock b = block_of(h);
  ProblemSet ps{};
  E.fill(ps);
  point_at_block(ps, b, const_cast<void*>(d_blocks));
  ps.hull_pb = h->cfg.n_local; ps.hull_bstride = (long)b.bytes;
  ps.hull_pb_magic = (h->cfg.n_local > 0 && h->cfg.num_agents < 65536) ? (1ull << 32) / (unsigned long long)h->cfg.n_local + 1ull : 0ull;
  h->fe_committed = nullptr;
  launch_frontend(h->slots, E.sp, ps, *cfg, d_start, d_guess, d_result, nullptr, (hipStream_t)stream, E.lpt ? E.d_fe_order.p : nullptr, E.fe_history);
  E.fe_history = true;
  HIPCHK(hipGetLastError());
  return 0;
}

int nep_batch_safety_commit(nep_batch_t* h, const nep_traj_rec* d_prev, const nep_traj_rec* d_new, const nep_guess* d_guess,
                            nep_traj_rec* d_final, int32_t* d_accept, void* stream) {
  if (!h || !d_prev || !d_new || !d_guess || !d_final) return fail(NEP_E_ARG, "null argument");
  Engine& E = h->eng;
  const int N = h->cfg.num_agents;
  if (E.sp.n_hull != N) return fail(NEP_E_STATE, "safety check needs the batched (all-agent) hull layout");
  if (int e = E.d_conflict.ensure((size_t)h->cfg.n_scenes * N * N)) return e;
  ProblemSet ps{};
  E.fill(ps);
  ps.guess = d_guess;
  if (E.safety_check_prev) { if (int e = E.d_conflict_prev.ensure((size_t)h->cfg.n_scenes * N * N)) return e; }
  launch_safety(d_prev, d_new, h->cfg.n_scenes, N, E.sp, ps, E.d_conflict.p, E.safety_check_prev ? E.d_conflict_prev.p : nullptr, nullptr, d_final, d_accept, (hipStream_t)stream);
  HIPCHK(hipGetLastError());
  return 0;
}

// ---- entangle check on: front end with per-node entangle states, safety pass with entangleCheckGivenPwp ----
namespace {
bool fe_cfg_ok(const nep_fe_cfg* cfg) {
  return cfg->num_samples >= 2 && cfg->num_samples <= NEP_FE_MAX_SAMPLES && cfg->beam_width >= 1 && cfg->beam_width <= NEP_FE_MAX_BEAM &&
         cfg->voxel_size > 0.0 && cfg->j_max > 0.0;
}
int ent_prepare(nep_batch* h, int ns, int beam_width, const nep_traj_rec* d_recs, const double* ts0, long ts_scene_stride, FeEntArgs& ea, hipStream_t st) {
  Engine& E = h->eng;
  const int N = h->cfg.num_agents, S = h->cfg.n_scenes, np = h->cfg.num_pol;
  if (ns < 1 || ns > 8) return fail(NEP_E_ARG, "ent_samples out of range");
  if (E.sp.n_static > 0 && !E.have_reps) return fail(NEP_E_STATE, "entangle check with static obstacles needs nep_batch_set_static_reps first");
  if (E.sp.n_static > 0 && E.d_srep.n < (size_t)(E.sp.static_stride ? S : 1) * E.sp.n_static * 4) return fail(NEP_E_STATE, "static representatives do not cover every scene's obstacle set: call nep_batch_set_static_reps after nep_batch_set_scene_statics");
  if (int e = E.d_sampled.ensure((size_t)S * N * np * (ns + 1) * 2)) return e;
  if (int e = E.d_present.ensure((size_t)S * N)) return e;
  if (int e = E.d_fe_work.ensure((size_t)std::max(h->slots * 256, S * N))) return e;
  if (beam_width > 0) { if (int e = E.d_fe_nodes.ensure((size_t)h->slots * (np + 1) * beam_width)) return e; }
  if (!E.d_srep.p) { if (int e = E.d_srep.ensure(4)) return e; if (int e2 = E.d_slong.ensure(2)) return e2; }
  {   // the re-check's big records (three times the search's bound: entangleCheckGivenPwp); a sixteenth of the search's pool, at least 256
    const long n_rec = std::max(256L, (E.fe_big_records > 0 ? E.fe_big_records : std::max(4096L, 4L * h->slots)) / 16);
    const size_t rec = (size_t)ent_big_rec_bytes(N, E.sp.n_static, 3);
    if (int e = E.d_fe_big_check.ensure(rec * (size_t)n_rec)) return e;
    if (int e = E.d_fe_big_check_count.ensure(1)) return e;
    ea.big_check = EntBigPool{E.d_fe_big_check.p, E.d_fe_big_check_count.p, (int)n_rec, (int)rec, ent_big_cap(N, E.sp.n_static, 3), ent_big_add_lim(N, E.sp.n_static, 3)};
  }
  launch_ent_sample(d_recs, S, N, ts0, ts_scene_stride, np, ns, E.sp.T_span, E.d_sampled.p, E.d_present.p, st, E.d_fe_big_check_count.p);
  ea.sampled = E.d_sampled.p; ea.present = E.d_present.p; ea.srep = E.d_srep.p; ea.slong = E.d_slong.p;
  ea.nodes = E.d_fe_nodes.p; ea.work = E.d_fe_work.p; ea.ns = ns; ea.init = nullptr; ea.case_out = nullptr; ea.saved = nullptr; ea.saved_arc = nullptr;
  ea.fast_cap = E.fe_fast_cap; ea.fast_add = E.fe_fast_add; ea.fast_bend = E.fe_fast_bend;
  return 0;
}
}  // namespace

int nep_batch_set_static_reps(nep_batch_t* h, int32_t scene, const double* rep, const double* longest) {
  if (!h || !rep || !longest || scene < -1 || scene >= h->cfg.n_scenes) return fail(NEP_E_ARG, "bad arguments");
  Engine& E = h->eng;
  const int S = E.sp.n_static;
  if (S == 0) { E.have_reps = true; return 0; }
  const int sets = E.sp.static_stride ? h->cfg.n_scenes : 1;
  HIPCHK(hipDeviceSynchronize());
  if (E.d_srep.n < (size_t)sets * S * 4) {
    DevBuf<double> nr, nl;
    if (int e = nr.ensure((size_t)sets * S * 4)) return e;
    if (int e = nl.ensure((size_t)sets * S * 2)) return e;
    HIPCHK(hipMemset(nr.p, 0, (size_t)sets * S * 4 * sizeof(double))); HIPCHK(hipMemset(nl.p, 0, (size_t)sets * S * 2 * sizeof(double)));
    E.d_srep.release(); E.d_slong.release(); E.d_srep = nr; E.d_slong = nl;
  }
  for (int s = 0; s < sets; s++) {
    if (scene >= 0 && sets > 1 && s != scene) continue;
    HIPCHK(hipMemcpy(E.d_srep.p + (size_t)s * S * 4, rep, (size_t)S * 4 * sizeof(double), hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(E.d_slong.p + (size_t)s * S * 2, longest, (size_t)S * 2 * sizeof(double), hipMemcpyHostToDevice));
  }
  E.have_reps = true;
  return 0;
}

// scratch of the entangle-aware front end: what every surviving child arrived with (kept for the depth's installs: the winners are
// not propagated twice) and the packed per-(agent, interval) records of the entangle check
static int fe_ent_scratch(nep_batch* h, const nep_fe_cfg& cfg, FeEntArgs& ea) {
  Engine& E = h->eng;
  const size_t cap = frontend_children_cap(cfg, h->cfg.num_pol);
  if (int e = E.d_fe_saved.ensure((size_t)h->slots * cap)) return e;
  if (int e = E.d_fe_arc.ensure((size_t)h->slots * cap)) return e;
  ea.saved = E.d_fe_saved.p; ea.saved_arc = E.d_fe_arc.p;
  if (int e = E.d_fe_stf.ensure((size_t)h->slots * cap)) return e;
  if (int e = E.d_fe_stvox.ensure((size_t)h->slots * cap)) return e;
  ea.st_f = E.d_fe_stf.p; ea.st_vox = E.d_fe_stvox.p;
  {   // where a round's lists of new crossings go when the LDS pool is full (cross_round): one block of kEntAddCap words per (child, step)
    const size_t xw = frontend_ent_xpool_words(E.sp, cfg, ea.ns);
    if (int e = E.d_fe_xpool.ensure((size_t)h->slots * xw)) return e;
    ea.xpool = E.d_fe_xpool.p; ea.xpool_stride = (int)xw;
  }
  ea.pk_stride = 2 + 2 * kBend + 2 * (ea.ns + 1);      // (kEntPkHead + the samples: an even number of doubles)
  if (int e = E.d_fe_packed.ensure((size_t)h->cfg.n_scenes * h->cfg.num_agents * h->cfg.num_pol * ea.pk_stride)) return e;
  ea.packed = E.d_fe_packed.p;
  // the pool of big records: states beyond the fixed record's capacities (rare: a handful of children in a few of 8 192 config-5
  // searches), sized by the reference's own bound on a list (num_agents + statics)
  const int N = h->cfg.num_agents, S = E.sp.n_static;
  const long n_rec = E.fe_big_records > 0 ? E.fe_big_records : std::max(4096L, 4L * h->slots);
  const size_t rec = (size_t)ent_big_rec_bytes(N, S);
  if (int e = E.d_fe_big.ensure(rec * (size_t)n_rec)) return e;
  if (int e = E.d_fe_big_count.ensure(2 + 256)) return e;      // [0] records claimed, [1] searches listed for the big-record instantiation, [2..] the list
  ea.redo_count = E.d_fe_big_count.p + 1; ea.redo_list = E.d_fe_big_count.p + 2; ea.redo_cap = 256;
  if (int e = E.d_fe_big_beta.ensure((size_t)ea.redo_cap * 256 * 120)) return e;      // (kEntBigLdsCap betas per thread of the big-record instantiation: 63 MB)
  ea.big_beta = E.d_fe_big_beta.p;
  ea.big = EntBigPool{E.d_fe_big.p, E.d_fe_big_count.p, (int)n_rec, (int)rec, ent_big_cap(N, S), ent_big_add_lim(N, S)};
  ea.fast_cap = E.fe_fast_cap; ea.fast_add = E.fe_fast_add; ea.fast_bend = E.fe_fast_bend;
  return 0;
}
int nep_batch_set_fe_ent_big_records(nep_batch_t* h, int64_t records) {
  if (!h || records < 0) return fail(NEP_E_ARG, "bad arguments");
  h->eng.fe_big_records = (long)records;
  return 0;
}
int nep_batch_set_fe_ent_fast_caps(nep_batch_t* h, int32_t list_cap, int32_t add_cap, int32_t bend_cap) {
  if (!h || list_cap < 0 || list_cap > NEP_FE_ENT_CAP || add_cap < 0 || add_cap > 32 || bend_cap < 0 || bend_cap > NEP_MAX_BEND) return fail(NEP_E_ARG, "fast-path capacities out of range");
  h->eng.fe_fast_cap = list_cap; h->eng.fe_fast_add = add_cap; h->eng.fe_fast_bend = bend_cap;
  return 0;
}

int nep_batch_frontend_ent(nep_batch_t* h, const nep_fe_cfg* cfg, const nep_traj_rec* d_committed, const nep_fe_start* d_start,
                           const nep_fe_ent_state* d_ent_init, nep_guess* d_guess, nep_fe_result* d_result, int32_t* d_case_out, void* stream) {
  if (!h || !cfg || !d_committed || !d_start || !d_guess) return fail(NEP_E_ARG, "null argument");
  if (!fe_cfg_ok(cfg)) return fail(NEP_E_ARG, "bad front-end configuration");
  if (!cfg->enable_entangle || !h->cfg.enable_entangle) return fail(NEP_E_STATE, "nep_batch_frontend_ent needs enable_entangle in the front-end configuration and in the handle");
  if (h->cfg.n_local != h->cfg.num_agents) return fail(NEP_E_STATE, "the entangle-aware front end runs on an unsharded handle (n_local == num_agents)");
  Engine& E = h->eng;
  ProblemSet ps{};
  E.fill(ps);
  h->fe_committed = d_committed;
  launch_hulls_ts(d_committed, h->cfg.n_scenes, h->cfg.num_agents, &d_start->t_start, (long)sizeof(nep_fe_start), E.sp, ps, (hipStream_t)stream);
  FeEntArgs ea{};
  if (int e = ent_prepare(h, cfg->ent_samples, cfg->beam_width, d_committed, &d_start->t_start, (long)sizeof(nep_fe_start) * E.sp.n_local, ea, (hipStream_t)stream)) return e;
  ea.init = d_ent_init; ea.case_out = d_case_out;
  if (int e = fe_ent_scratch(h, *cfg, ea)) return e;
  launch_frontend(h->slots, E.sp, ps, *cfg, d_start, d_guess, d_result, &ea, (hipStream_t)stream, E.lpt ? E.d_fe_order.p : nullptr, E.fe_history);
  E.fe_history = true;
  HIPCHK(hipGetLastError());
  return 0;
}

// the same against all-gathered hull blocks (which carry the samples and the presence flags of every agent's trajectory when the
// handle was created with enable_entangle: nep_batch_hulls): the entangle-aware front end of a sharded handle
int nep_batch_frontend_ent_hulls(nep_batch_t* h, const nep_fe_cfg* cfg, const void* d_blocks, int32_t n_blocks, const nep_fe_start* d_start,
                                 const nep_fe_ent_state* d_ent_init, nep_guess* d_guess, nep_fe_result* d_result, int32_t* d_case_out, void* stream) {
  if (!h || !cfg || !d_blocks || !d_start || !d_guess) return fail(NEP_E_ARG, "null argument");
  if (n_blocks < 1 || n_blocks * h->cfg.n_local != h->cfg.num_agents) return fail(NEP_E_ARG, "n_blocks * n_local must equal num_agents");
  if (!fe_cfg_ok(cfg)) return fail(NEP_E_ARG, "bad front-end configuration");
  if (!cfg->enable_entangle || !h->cfg.enable_entangle) return fail(NEP_E_STATE, "nep_batch_frontend_ent_hulls needs enable_entangle in the front-end configuration and in the handle");
  if (cfg->ent_samples != h->ent_ns) return fail(NEP_E_ARG, "ent_samples differs from what the hull blocks were made with (nep_batch_set_ent_samples)");
  Engine& E = h->eng;
  if (E.sp.n_static > 0 && !E.have_reps) return fail(NEP_E_STATE, "entangle check with static obstacles needs nep_batch_set_static_reps first");
  const HullBlock b = block_of(h);
  ProblemSet ps{};
  E.fill(ps);
  point_at_block(ps, b, const_cast<void*>(d_blocks));
  ps.hull_pb = h->cfg.n_local; ps.hull_bstride = (long)b.bytes;
  ps.hull_pb_magic = (h->cfg.n_local > 0 && h->cfg.num_agents < 65536) ? (1ull << 32) / (unsigned long long)h->cfg.n_local + 1ull : 0ull;
  h->fe_committed = nullptr;
  if (int e = E.d_fe_work.ensure((size_t)std::max(h->slots * 256, h->cfg.n_scenes * h->cfg.num_agents))) return e;
  if (int e = E.d_fe_nodes.ensure((size_t)h->slots * (h->cfg.num_pol + 1) * cfg->beam_width)) return e;
  if (!E.d_srep.p) { if (int e = E.d_srep.ensure(4)) return e; if (int e2 = E.d_slong.ensure(2)) return e2; }
  FeEntArgs ea{};
  ea.sampled = (const double*)((const char*)d_blocks + b.samp); ea.present = (const int*)((const char*)d_blocks + b.present);
  ea.srep = E.d_srep.p; ea.slong = E.d_slong.p; ea.nodes = E.d_fe_nodes.p; ea.work = E.d_fe_work.p; ea.ns = h->ent_ns;
  ea.init = d_ent_init; ea.case_out = d_case_out;
  if (int e = fe_ent_scratch(h, *cfg, ea)) return e;
  launch_frontend(h->slots, E.sp, ps, *cfg, d_start, d_guess, d_result, &ea, (hipStream_t)stream, E.lpt ? E.d_fe_order.p : nullptr, E.fe_history);
  E.fe_history = true;
  HIPCHK(hipGetLastError());
  return 0;
}

int nep_batch_safety_commit_ent(nep_batch_t* h, const nep_traj_rec* d_prev, const nep_traj_rec* d_new, const nep_guess* d_guess,
                                const nep_fe_ent_state* d_ent_init, int32_t ent_samples, double cable_length, nep_traj_rec* d_final,
                                int32_t* d_accept, void* stream) {
  if (!h || !d_prev || !d_new || !d_guess || !d_final) return fail(NEP_E_ARG, "null argument");
  Engine& E = h->eng;
  const int N = h->cfg.num_agents;
  if (E.sp.n_hull != N) return fail(NEP_E_STATE, "the entangle re-check needs the batched (all-agent) hull layout");
  if (!h->cfg.enable_entangle) return fail(NEP_E_STATE, "handle created without enable_entangle");
  if (int e = E.d_conflict.ensure((size_t)h->cfg.n_scenes * N * N)) return e;
  if (int e = E.d_entangles.ensure((size_t)h->cfg.n_scenes * N)) return e;
  ProblemSet ps{};
  E.fill(ps);
  ps.guess = d_guess;
  if (E.safety_check_prev) { if (int e = E.d_conflict_prev.ensure((size_t)h->cfg.n_scenes * N * N)) return e; }
  // everybody's NEW trajectory counts as received while optimising: their samples and bend points feed the re-check
  FeEntArgs ea{};
  if (int e = ent_prepare(h, ent_samples, 0, d_new, &d_guess->t_start, (long)sizeof(nep_guess) * E.sp.n_local, ea, (hipStream_t)stream)) return e;
  ea.init = d_ent_init;
  ea.pk_stride = 2 + 2 * kBend + 2 * (ea.ns + 1);
  if (int e = E.d_fe_packed.ensure((size_t)h->cfg.n_scenes * N * h->cfg.num_pol * ea.pk_stride)) return e;
  ea.packed = E.d_fe_packed.p;
  launch_hulls(d_new, h->cfg.n_scenes, N, d_guess, E.sp, ps, (hipStream_t)stream);     // (with the bend points: ent_enabled)
  launch_ent_check(E.sp, ps, ea, d_new, h->cfg.n_scenes, cable_length, E.d_entangles.p, (hipStream_t)stream);
  launch_safety(d_prev, d_new, h->cfg.n_scenes, N, E.sp, ps, E.d_conflict.p, E.safety_check_prev ? E.d_conflict_prev.p : nullptr, E.d_entangles.p, d_final, d_accept, (hipStream_t)stream);
  HIPCHK(hipGetLastError());
  return 0;
}

int nep_batch_next_starts(nep_batch_t* h, const nep_traj_rec* d_records, double dt, nep_fe_start* d_start, double* d_alt_goal,
                          double switch_radius, void* stream) {
  if (!h || !d_records || !d_start || !(dt >= 0.0)) return fail(NEP_E_ARG, "bad arguments");
  launch_next_starts(d_records, h->cfg.n_scenes, h->cfg.num_agents, h->cfg.first_local, h->cfg.n_local, dt, d_start, d_alt_goal, switch_radius, (hipStream_t)stream);
  HIPCHK(hipGetLastError());
  return 0;
}

int nep_batch_set_scene_statics(nep_batch_t* h, int32_t scene, int32_t n_static, const int32_t* static_off, const double* static_xy) {
  if (!h || scene < 0 || scene >= h->cfg.n_scenes || n_static < 0 || (n_static > 0 && (!static_off || !static_xy))) return fail(NEP_E_ARG, "bad arguments");
  HIPCHK(hipDeviceSynchronize());     // the previous set may still be read by kernels in flight
  return h->eng.upload_scene_statics(scene, n_static, static_off, static_xy);
}

// Test hook: replans the last nep_batch_replan* sent through the presolve's redo pass (0 when LP skipping is off).
int nep_batch_debug_redo_count(nep_batch_t* h, int32_t* by_reason) {
  if (!h) return fail(NEP_E_ARG, "null handle");
  int n[3] = {0, 0, 0};
  HIPCHK(hipDeviceSynchronize());
  if (h->eng.d_redo_count.p) HIPCHK(hipMemcpy(n, h->eng.d_redo_count.p, 3 * sizeof(int), hipMemcpyDeviceToHost));
  if (by_reason) { by_reason[0] = n[1]; by_reason[1] = n[2]; }
  return n[0];
}

// Test hook: the slots on the redo list of the last replan (at most cap).
int nep_batch_debug_redo_list(nep_batch_t* h, int32_t* slots_out, int32_t cap) {
  if (!h || !slots_out || cap < 0) return fail(NEP_E_ARG, "bad arguments");
  int n = 0;
  HIPCHK(hipDeviceSynchronize());
  if (!h->eng.d_redo_count.p) return 0;
  HIPCHK(hipMemcpy(&n, h->eng.d_redo_count.p, sizeof(int), hipMemcpyDeviceToHost));
  if (n > cap) n = cap;
  if (n > 0) HIPCHK(hipMemcpy(slots_out, h->eng.d_redo_list.p, (size_t)n * sizeof(int), hipMemcpyDeviceToHost));
  return n;
}

// Test hook: segments per wave of the presolve's separator (0: by launch size, -1: the unpacked kernel, 1..NEP_MAX_POL: forced)
int nep_batch_debug_set_separator_pack(nep_batch_t* h, int32_t pack) {
  if (!h || pack < -1 || pack > NEP_MAX_POL) return fail(NEP_E_ARG, "pack must be -1, 0 or 1..NEP_MAX_POL");
  h->eng.sep_pack = pack;
  return 0;
}

int nep_batch_qp_placement(nep_batch_t* h) { return h ? (h->eng.use_reg ? 1 : 0) : NEP_E_ARG; }

int nep_batch_set_launch_order(nep_batch_t* h, int32_t enable) {
  if (!h) return fail(NEP_E_ARG, "null handle");
  if (enable) { if (int e = h->eng.d_order.ensure((size_t)h->slots)) return e; if (int e = h->eng.d_order_key.ensure((size_t)h->slots)) return e; }
  if (enable) { if (int e = h->eng.d_fe_order.ensure((size_t)h->slots)) return e; if (int e = h->eng.d_fe_order_key.ensure((size_t)h->slots)) return e; }
  if (!enable) { h->eng.have_history = false; h->eng.fe_history = false; }
  if (enable && !h->eng.lpt) {      // (the keys remember: switched on, they start from zero)
    HIPCHK(hipDeviceSynchronize());
    HIPCHK(hipMemset(h->eng.d_order_key.p, 0, h->eng.d_order_key.n * sizeof(int))); HIPCHK(hipMemset(h->eng.d_fe_order_key.p, 0, h->eng.d_fe_order_key.n * sizeof(int)));
    h->eng.have_history = false; h->eng.fe_history = false;
  }
  h->eng.lpt = enable != 0;
  return 0;
}
int nep_batch_fe_search_us(nep_batch_t* h, float* us, int32_t cap) {
  if (!h || !us || cap < h->slots) return fail(NEP_E_ARG, "bad arguments");
  HIPCHK(hipDeviceSynchronize());
  HIPCHK(hipMemcpy(us, h->eng.d_fe_us.p, (size_t)h->slots * sizeof(float), hipMemcpyDeviceToHost));
  return 0;
}
int nep_batch_debug_launch_order(nep_batch_t* h, int32_t* order, int32_t cap, int32_t* n_out) {
  if (!h || !order || !n_out || cap < 0) return fail(NEP_E_ARG, "bad arguments");
  *n_out = 0;
  if (!h->eng.last_ordered) return 0;
  if (cap < h->slots) return fail(NEP_E_CAP, "order buffer smaller than the slot count");
  HIPCHK(hipDeviceSynchronize());
  HIPCHK(hipMemcpy(order, h->eng.d_order.p, (size_t)h->slots * sizeof(int), hipMemcpyDeviceToHost));
  *n_out = h->slots;
  return 0;
}

// Development aids that used to be environment variables (rounds 2-5: NEP_QP_KERNEL, NEP_SEP_SKIP, NEP_SEP_NO_REDO, NEP_QP_LPT, NEP_FE_LPT,
// NEP_QP_KEY_DECAY, NEP_FE_KEY_DECAY, NEP_SEP_UNPACKED / NEP_SEP_PACK, NEP_CORR_FROM / NEP_CORR_MAX, NEP_FE_THREE, NEP_FE_XCD,
// NEP_POLISH_GRID, NEP_HULL_KERNEL): explicit setters now, so that a caller's environment cannot change what the library computes.
static int engine_option(Engine& E, const char* name, int32_t v) {
  if (!name) return fail(NEP_E_ARG, "null option name");
  const std::string n(name);
  if (n == "qp_kernel") { if (v < 0 || v > 2) return fail(NEP_E_ARG, "qp_kernel: 0 automatic, 1 register placement, 2 LDS placement"); E.force_kernel = v; E.choose_placement(); }
  else if (n == "sep_skip") E.skip_lps = v != 0;                 // 0: a presolved replan's far LPs are solved too (parked), none skipped
  else if (n == "sep_no_redo") E.no_redo = v != 0;               // 1: flagged replans keep their presolved result (inspection)
  else if (n == "qp_lpt") E.lpt = v != 0;
  else if (n == "fe_lpt") E.fe_lpt = v != 0;
  else if (n == "qp_key_decay") E.opt_qp_key_decay = v;
  else if (n == "fe_key_decay") E.opt_fe_key_decay = v;
  else if (n == "sep_pack") { if (v < -1 || v > NEP_MAX_POL) return fail(NEP_E_ARG, "sep_pack: -1 unpacked, 0 automatic, 1..NEP_MAX_POL"); E.sep_pack = v; }
  else if (n == "corr_from") { if (v < 1) return fail(NEP_E_ARG, "corr_from >= 1"); E.opt_corr_from = v; E.sp.corr_from_it = v; }
  else if (n == "corr_max") { if (v < 1) return fail(NEP_E_ARG, "corr_max >= 1"); E.opt_corr_max = v; E.sp.corr_max_count = v; }
  else if (n == "qp_profile") E.profile_phases = v != 0;
  else if (n == "presolve_kernel") E.presolve_kernel = v != 0;      // 0: the zero-iteration test only inside qp_reg_kernel<true> (rounds 3-5); same results up to the last place of the objective
  else return fail(NEP_E_ARG, "unknown debug option: " + n);
  return 0;
}
int nep_batch_debug_set_option(nep_batch_t* h, const char* name, int32_t value) { if (!h) return fail(NEP_E_ARG, "null handle"); return engine_option(h->eng, name, value); }
int nep_backend_debug_set_option(nep_backend_t* h, const char* name, int32_t value) { if (!h) return fail(NEP_E_ARG, "null handle"); return engine_option(h->eng, name, value); }
int nep_debug_set_global_option(const char* name, int32_t value) {
  if (!name) return fail(NEP_E_ARG, "null option name");
  const std::string n(name);
  if (n == "fe_three") g_debug.fe_three = value != 0;
  else if (n == "fe_xcd") g_debug.fe_xcd = value;
  else if (n == "polish_grid") { if (value < 1) return fail(NEP_E_ARG, "polish_grid >= 1"); g_debug.polish_grid = value; }
  else return fail(NEP_E_ARG, "unknown global debug option: " + n);
  return 0;
}

int nep_batch_set_hull_kernel(nep_batch_t* h, int32_t mode) {
  if (!h || mode < 0 || mode > 2) return fail(NEP_E_ARG, "mode: 0 automatic, 1 one hull per wave, 2 eight hulls per wave");
  h->eng.sp.hull_mode = mode;
  return 0;
}

int nep_batch_set_max_runtime(nep_batch_t* h, double seconds) {
  if (!h || !(seconds >= 0.0)) return fail(NEP_E_ARG, "bad arguments");
  h->eng.sp.time_limit_ticks = seconds > 0 ? (long long)(seconds * h->eng.clock_hz) : 0;
  return 0;
}

int nep_batch_set_line_cull(nep_batch_t* h, double radius) {
  if (!h || !(radius >= 0.0)) return fail(NEP_E_ARG, "bad arguments");
  h->eng.sp.cull_radius = radius; h->eng.cull_user_set = true;
  h->eng.choose_placement();      // (a culled problem's near lines fit the register kernel whatever the scene size)
  HIPCHK(hipDeviceSynchronize());
  return h->eng.size_row_scratch();      // (the row scratch follows the mode: a pool with the redo pass, one area per slot without — never inside a capture)
}
double nep_batch_get_line_cull(nep_batch_t* h) { return h ? h->eng.sp.cull_radius : -1.0; }
// (the per-agent handle sizes its scratch and picks its placement at every optimize(): nothing to re-size here)
int nep_backend_set_line_cull(nep_backend_t* h, double radius) {
  if (!h || !(radius >= 0.0)) return fail(NEP_E_ARG, "bad arguments");
  h->eng.sp.cull_radius = radius; h->eng.cull_user_set = true;
  return 0;
}

int nep_batch_set_separator_rule(nep_batch_t* h, int32_t rule) {
  if (!h || (rule != 0 && rule != 1)) return fail(NEP_E_ARG, "separator rule: 0 largest gap, 1 GLPK-class simplex");
  h->eng.sp.sep_rule = rule;
  HIPCHK(hipDeviceSynchronize());
  return h->eng.size_row_scratch();
}
// Row scratch for the worst case (one area per slot) whatever the mode: nep_batch_check reports NEP_E_CAP when the pool of the
// presolve's redo pass ran out (more than 1 024 replans of one launch listed with rows beyond the register slots).
int nep_batch_reserve_row_scratch(nep_batch_t* h) {
  if (!h) return fail(NEP_E_ARG, "null handle");
  h->eng.scratch_full = true;
  HIPCHK(hipDeviceSynchronize());
  return h->eng.size_row_scratch();
}
// Lines per (replan, segment) the line buckets hold: 0 the default budget, -1 the reference's worst case, n > 0 that many (see
// size_scratch).  Re-sizes the buffers: not inside a graph capture.
int nep_batch_set_line_capacity(nep_batch_t* h, int32_t lines_per_segment) {
  if (!h || lines_per_segment < -1) return fail(NEP_E_ARG, "bad arguments");
  HIPCHK(hipDeviceSynchronize());
  h->eng.lines_cap_user = lines_per_segment;
  return h->eng.size_scratch();
}
int64_t nep_batch_line_bucket_bytes(nep_batch_t* h) { return h ? (int64_t)h->slots * NEP_MAX_POL * h->eng.sp.lines_cap * 3 * (int64_t)sizeof(double) : 0; }
int64_t nep_batch_row_scratch_bytes(nep_batch_t* h) { return h ? (int64_t)(h->eng.d_row_scratch.n * sizeof(double)) : 0; }
int nep_backend_set_separator_rule(nep_backend_t* h, int32_t rule) {
  if (!h || (rule != 0 && rule != 1)) return fail(NEP_E_ARG, "separator rule: 0 largest gap, 1 GLPK-class simplex");
  h->eng.sp.sep_rule = rule;
  return 0;
}

namespace { int set_tol(Engine& E, double res, double gap) {
  if (!(res >= 1e-12 && res <= 1e-6) || !(gap >= 1e-13 && gap <= 1e-7)) return fail(NEP_E_ARG, "tolerances: residuals in [1e-12, 1e-6], relative gap in [1e-13, 1e-7]");
  E.sp.tol_res = res; E.sp.tol_gap = gap;
  E.sp.tol_res_inv = res == 1e-10 ? 1e10 : (res == 1e-9 ? 1e9 : 1.0 / res); E.sp.tol_gap_inv = gap == 1e-11 ? 1e11 : (gap == 1e-10 ? 1e10 : 1.0 / gap); E.sp.tol_gap_floor = 0.1 * gap;      // (the defaults' reciprocals as the literals the kernels were validated with)
  return 0;
} }
int nep_batch_set_tolerances(nep_batch_t* h, double residual_tol, double gap_tol) { if (!h) return fail(NEP_E_ARG, "null handle"); return set_tol(h->eng, residual_tol, gap_tol); }
int nep_backend_set_tolerances(nep_backend_t* h, double residual_tol, double gap_tol) { if (!h) return fail(NEP_E_ARG, "null handle"); return set_tol(h->eng, residual_tol, gap_tol); }

// (on: 0 off; 1 — the default — and 2 everywhere, under the presolve as well; 3: every-row solves only, round 5's default)
int nep_batch_set_polish(nep_batch_t* h, int32_t on) { if (!h) return fail(NEP_E_ARG, "null handle"); h->eng.polish = on != 0; h->eng.polish_presolve = on == 1 || on == 2; return 0; }
int nep_backend_set_polish(nep_backend_t* h, int32_t on) { if (!h) return fail(NEP_E_ARG, "null handle"); h->eng.polish = on != 0; h->eng.polish_presolve = on == 1 || on == 2; return 0; }
// Test hook: the last replan's polish pass — replans listed for it (solves that ended without the strict tests), replans certified
int nep_batch_debug_polish_count(nep_batch_t* h, int32_t* listed, int32_t* certified) {
  if (!h) return fail(NEP_E_ARG, "null handle");
  int c[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  HIPCHK(hipDeviceSynchronize());
  // (a launch whose pass was not armed — polish off, or the LDS-placement kernel, which has no hooks — leaves the counters of an earlier
  // launch behind: reported as (0, 0), round-5 advisor finding)
  if (h->eng.d_polish_count.p && h->eng.last_polish_armed) HIPCHK(hipMemcpy(c, h->eng.d_polish_count.p, sizeof(c), hipMemcpyDeviceToHost));
  if (listed) *listed = c[0];
  if (certified) *certified = c[3];
  return 0;
}
// Test hook: per slot of the last replan, 0 = not listed for the polish pass, else the flag word — bit m: the solve of mode m (0 first
// problem, 1 relaxed) was left for the pass; bit 8: the pass certified an optimum and wrote the slot's result.
int nep_batch_debug_polish_flags(nep_batch_t* h, int32_t* flags, int32_t cap) {
  if (!h || !flags || cap < h->slots) return fail(NEP_E_ARG, "bad arguments");
  for (int i = 0; i < h->slots; i++) flags[i] = 0;
  if (!h->eng.d_polish_count.p || !h->eng.last_polish_armed) return 0;
  int c[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  HIPCHK(hipDeviceSynchronize());
  HIPCHK(hipMemcpy(c, h->eng.d_polish_count.p, sizeof(c), hipMemcpyDeviceToHost));
  const int n = c[0] < h->slots ? c[0] : h->slots;
  if (n <= 0) return 0;
  std::vector<int> list(n), fl(h->slots);
  HIPCHK(hipMemcpy(list.data(), h->eng.d_polish_list.p, (size_t)n * sizeof(int), hipMemcpyDeviceToHost));
  HIPCHK(hipMemcpy(fl.data(), h->eng.d_polish_flag.p, (size_t)h->slots * sizeof(int), hipMemcpyDeviceToHost));
  for (int i = 0; i < n; i++) if (list[i] >= 0 && list[i] < h->slots) flags[list[i]] = fl[list[i]];
  return 0;
}
int nep_batch_set_safety_check_prev(nep_batch_t* h, int32_t on) { if (!h) return fail(NEP_E_ARG, "null handle"); h->eng.safety_check_prev = on != 0; return 0; }

int nep_batch_debug_conflicts(nep_batch_t* h, int32_t scene, uint8_t* conflict_out) {
  if (!h || !conflict_out || scene < 0 || scene >= h->cfg.n_scenes) return fail(NEP_E_ARG, "bad arguments");
  const size_t nn = (size_t)h->cfg.num_agents * h->cfg.num_agents;
  if (h->eng.d_conflict.n < (size_t)h->cfg.n_scenes * nn) return fail(NEP_E_STATE, "no safety check has run");
  HIPCHK(hipDeviceSynchronize());
  HIPCHK(hipMemcpy(conflict_out, h->eng.d_conflict.p + (size_t)scene * nn, nn, hipMemcpyDeviceToHost));
  return 0;
}

int nep_batch_wait(nep_batch_t* h, void* stream) { if (!h) return fail(NEP_E_ARG, "null handle"); HIPCHK(hipStreamSynchronize((hipStream_t)stream)); return 0; }

int nep_batch_check(nep_batch_t* h, void* stream) {
  if (!h) return fail(NEP_E_ARG, "null handle");
  HIPCHK(hipStreamSynchronize((hipStream_t)stream));
  int flags = 0;
  if (h->eng.d_flags.p) {
    HIPCHK(hipMemcpy(&flags, h->eng.d_flags.p, sizeof(int), hipMemcpyDeviceToHost));
    if (flags) HIPCHK(hipMemset(h->eng.d_flags.p, 0, sizeof(int)));
  }
  if (flags & NEP_FLAG_LINES) return fail(NEP_E_CAP, "a segment got more separating lines than its bucket holds: nep_batch_set_line_capacity(h, -1) sizes the buckets for the reference's worst case");
  if (flags & NEP_FLAG_SCRATCH) return fail(NEP_E_CAP, "the presolve's redo pass listed more replans with rows beyond the register slots than the handle has scratch areas for: nep_batch_reserve_row_scratch");
  if (flags & NEP_FLAG_ENT_POOL) return fail(NEP_E_CAP, "the pool of big entangle-state records ran out (front end: children of a search were pruned for it, by claim order; safety re-check: a trajectory was turned down): nep_batch_set_fe_ent_big_records");
  if (flags & NEP_FLAG_ENT_BETA) return fail(NEP_E_ARG, "an entangle state passed to the front end has a non-zero beta for an agent crossing (the reference's calculateBetaForCase makes it 0.0)");
  if (flags & NEP_FLAG_HULL_OVERFLOW) return fail(NEP_E_CAP, "an interval overlaps more than NEP_HULL_MAX_CP/4 committed segments (or its hull has more than NEP_HULL_MAX_V vertices)");
  return 0;
}

int nep_batch_enable_timing(nep_batch_t* h, int32_t on) { if (!h) return fail(NEP_E_ARG, "null handle"); h->eng.timing = on != 0; h->eng.ev_used = 0; return 0; }

// which: 0 hull kernel, 1 separator kernel, 2 QP kernel, 3 whole sequence
int nep_batch_kernel_time(nep_batch_t* h, int32_t which, double* avg_ms, int32_t* n_launch) {
  if (!h || !avg_ms || which < 0 || which > 3) return fail(NEP_E_ARG, "bad arguments");
  Engine& E = h->eng;
  const size_t calls = E.ev_used / 4;
  double tot = 0;
  for (size_t c = 0; c < calls; c++) {
    float ms = 0;
    hipEvent_t a = E.ev[4 * c + (which == 3 ? 0 : which)], b = E.ev[4 * c + (which == 3 ? 3 : which + 1)];
    HIPCHK(hipEventSynchronize(b));
    HIPCHK(hipEventElapsedTime(&ms, a, b));
    tot += ms;
  }
  *avg_ms = calls ? tot / calls : 0.0;
  if (n_launch) *n_launch = (int32_t)calls;
  return 0;
}

int nep_batch_reset_timing(nep_batch_t* h) { if (!h) return fail(NEP_E_ARG, "null handle"); h->eng.ev_used = 0; return 0; }

int nep_batch_debug_hulls(nep_batch_t* h, int32_t scene, double* hull_xy, int32_t* hull_nv) {
  if (!h || !hull_xy || !hull_nv || scene < 0 || scene >= h->cfg.n_scenes) return fail(NEP_E_ARG, "bad arguments");
  Engine& E = h->eng; const size_t np = (size_t)E.sp.n_hull * E.sp.num_pol;
  HIPCHK(hipDeviceSynchronize());
  HIPCHK(hipMemcpy(hull_xy, E.d_hull_xy.p + (size_t)scene * np * kHullV * 2, np * kHullV * 2 * sizeof(double), hipMemcpyDeviceToHost));
  HIPCHK(hipMemcpy(hull_nv, E.d_hull_nv.p + (size_t)scene * np, np * sizeof(int), hipMemcpyDeviceToHost));
  return 0;
}

int nep_batch_debug_lines(nep_batch_t* h, int32_t slot, int32_t cap, int32_t* seg, double* nd, int32_t* n_out) {
  if (!h || !n_out || slot < 0 || slot >= h->slots) return fail(NEP_E_ARG, "bad arguments");
  Engine& E = h->eng;
  HIPCHK(hipDeviceSynchronize());
  return read_lines(E, (size_t)slot, NEP_MAX_POL, cap, seg, nd, n_out);
}

// Diagnostic: active inequality rows (slack < tol) at the solutions of the last nep_batch_replan*, per slot: d_out [slots][2] =
// (box rows, separating-line rows).  Reads the handle's line buckets as that replan left them and d_solution as the caller got it.
int nep_batch_active_rows(nep_batch_t* h, const nep_solution* d_solution, double tol, int32_t* d_out, void* stream) {
  if (!h || !d_solution || !d_out || !(tol >= 0.0)) return fail(NEP_E_ARG, "bad arguments");
  Engine& E = h->eng;
  ProblemSet ps{};
  E.fill(ps);
  ps.solution = const_cast<nep_solution*>(d_solution);
  launch_active_rows(h->slots, E.sp, ps, tol, d_out, (hipStream_t)stream);
  HIPCHK(hipGetLastError());
  return 0;
}

// development aid: per-phase shader cycles of the QP kernel (only with NEP_QP_PROFILE set at create)
int nep_batch_debug_phase_cycles(nep_batch_t* h, int32_t slot, int64_t* out16) {
  if (!h || !out16 || slot < 0 || slot >= 2 * h->slots) return fail(NEP_E_ARG, "bad arguments");      // (the front end keeps 32 values per slot: rows 2 slot and 2 slot + 1)
  if (!h->eng.profile_phases) return fail(NEP_E_STATE, "NEP_QP_PROFILE was not set when the handle was created");
  HIPCHK(hipDeviceSynchronize());
  HIPCHK(hipMemcpy(out16, h->eng.d_dbg.p + (size_t)slot * 16, 16 * sizeof(long long), hipMemcpyDeviceToHost));
  return 0;
}

}  // extern "C"
// hooks for exchange.hip (the RCCL step lives in its own translation unit)
namespace nep {
void set_last_error(const std::string& msg) { g_err = msg; }
int64_t batch_hull_block_bytes(const struct ::nep_batch* h) { return nep_batch_hull_block_bytes(h); }
void batch_dims(const struct ::nep_batch* h, int* n_scenes, int* n_local, int* num_agents) { *n_scenes = h->cfg.n_scenes; *n_local = h->cfg.n_local; *num_agents = h->cfg.num_agents; }
}
extern "C" {

// layout self-description for the ctypes mirror (tests/test_abi.py)
int nep_abi_sizeof(int32_t which) {
  switch (which) {
    case 0: return (int)sizeof(nep_pwp);
    case 1: return (int)sizeof(nep_traj_rec);
    case 2: return (int)sizeof(nep_backend_cfg);
    case 3: return (int)sizeof(nep_stats);
    case 4: return (int)sizeof(nep_batch_cfg);
    case 5: return (int)sizeof(nep_guess);
    case 6: return (int)sizeof(nep_solution);
    case 7: return (int)sizeof(nep_ent_view);
    case 8: return (int)sizeof(nep_wire_header);
    case 9: return (int)sizeof(nep_plan_cfg);
    case 10: return (int)sizeof(nep_point_a);
    case 11: return (int)sizeof(nep_fe_cfg);
    case 12: return (int)sizeof(nep_fe_start);
    case 13: return (int)sizeof(nep_fe_result);
    case 14: return (int)sizeof(nep_fe_ent_state);
    default: return -1;
  }
}

}  // extern "C"
