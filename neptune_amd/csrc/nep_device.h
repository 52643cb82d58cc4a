// nep_device.h — device-side views shared by the HIP kernels and the host launcher.
#pragma once
#include <hip/hip_runtime.h>

#include <mutex>

#include "nep_tables.h"
#include "../../include/neptune_frontend.h"
#include "../../include/neptune_backend_debug.h"

namespace nep {

constexpr int kHullV = NEP_HULL_MAX_V;   // 16
constexpr int kHullCP = NEP_HULL_MAX_CP; // 16
constexpr int kBend = NEP_MAX_BEND;      // 8

// Process-wide A/B knobs of the launchers (nep_debug_set_global_option, include/neptune_backend_debug.h): scheduling and placement
// only — results do not depend on them — and never read from the environment (the library behaves the same under any environment).
struct DebugGlobals { bool fe_three = false; int fe_xcd = 1; int polish_grid = 256; };
extern DebugGlobals g_debug;

// Scene-level constants (setMaxValues, ctor arguments; solver_gurobi_poly.cpp:25-175)
constexpr int kCorrFromItDefault = 10, kCorrMaxCountDefault = 8;      // (qp_common.h: kCorrFromIt, kCorrMaxCount)
struct SceneParams {
  int num_agents;     // N (pb.size())
  int num_pol;        // planning intervals
  int n_static;       // S
  int n_hull;         // hull lists per scene: N in batch mode, n_obst in per-agent mode
  int ent_enabled;
  int hull_mode;      // 0: pick the hull kernel by batch size; 1: one hull per wave (hull_kernel); 2: eight per wave (hull_group_kernel)
  int max_states;
  int lines_cap;      // capacity of one (slot, segment) line bucket
  int n_local;        // slots per scene
  int first_local;    // index of first local agent (batch); per-agent mode: id-1
  int skip_own;       // 1: hull list is indexed by agent, own entry skipped
  int static_stride;  // 0: one static-obstacle set for every scene; S: scene s reads polygons [s*S, (s+1)*S) (nep_batch_set_scene_statics)
  double T_span, weight, dc, drone_radius;
  double mins[3], maxs[3], v_max, a_max;
  double long_length; // solver_gurobi_poly.cpp:173
  double cull_radius; // > 0: separating lines farther than this from the guess are presolved away (verified after the solve)
  long long time_limit_ticks;   // > 0: wall-clock budget of ONE solve in wall_clock64() ticks (setMaxRuntime -> Gurobi TimeLimit, solver_gurobi_poly.cpp:812)
  int corr_from_it, corr_max_count;      // the interior point's give-up rule (qp_common.h: kCorrFromIt, kCorrMaxCount; NEP_CORR_FROM / NEP_CORR_MAX: development A/B)
  double tol_res, tol_gap, tol_res_inv, tol_gap_inv, tol_gap_floor;      // (tol_gap_floor = 0.1 tol_gap: the centring target's floor)      // the interior point's strict tests: residuals (absolute, the dual one scaled), relative gap; their reciprocals for the merit (nep_batch_set_tolerances)
  int sep_rule;                 // which vertex of the separator LP is returned: 0 the largest-gap one (default), 1 the one a primal simplex of GLPK's default class reaches (nep_batch_set_separator_rule)
  int qp_key_decay;             // the same for the QP workgroups' key (8 us bins; 2: chain QP launch 1.63 -> 1.55 ms, crossing 2.29 -> 2.26; NEP_QP_KEY_DECAY for A/Bs, 0 = the last solve's bin)
  int fe_key_decay;             // front end's launch-order key: bins an old key loses per launch (0: the key is the last search's bin)
  double us_per_tick;           // microseconds per wall_clock64() tick of this device (hipDeviceAttributeWallClockRate; 0.01 on gfx950: 100 MHz)
};

// Buffers of one problem set (n_scenes x n_local slots).  All device pointers.
struct ProblemSet {
  // inputs
  const nep_guess* guess;        // [slots]
  const double* pb;              // [N][2]
  const double* static_xy;       // [S][kHullV][2]   (x n_scenes when sp.static_stride != 0)
  const int* static_nv;          // [S]
  const double* static_el;       // [S][kHullV] length of edge v -> v+1 (the proximity cull's square roots, taken once at upload)
  const int* case_id;            // [slots][NEP_MAX_POL][N] or null
  // per scene (batch: written by the hull kernel; per-agent: uploaded by setHulls)
  double* hull_xy;               // [scenes][n_hull][num_pol][kHullV][2]
  int* hull_nv;                  // [scenes][n_hull][num_pol]
  double* hull0_xy;              // [scenes][N][num_pol][2]  col(0) of the uninflated hull
  int* hull0_nv;                 // [scenes][N][num_pol]
  double* bend_xy;               // [scenes][N][kBend][2]
  int* bend_n;                   // [scenes][N]
  // Sharded hulls (nep_batch_replan_hulls): the six arrays above are the first of N / hull_pb
  // equal blocks, hull_bstride bytes apart, each holding hull_pb agents per scene — the layout an
  // all-gather of per-rank blocks produces.  hull_pb == 0: one block (everything above as stated).
  int hull_pb;
  unsigned long long hull_pb_magic;   // 2^32 / hull_pb + 1 (host): j / hull_pb = (j * magic) >> 32 for j < 65536; 0: divide
  long hull_bstride;
  // separator output
  double* line_nd;               // [slots][NEP_MAX_POL][lines_cap][3]
  int* line_cnt;                 // [slots][NEP_MAX_POL]  lines at the front of the bucket (all of them, or the near ones); -1 - n: the bucket overflowed (line_count())
  int* line_far;                 // [slots][NEP_MAX_POL]  presolved-away lines parked at the back of the bucket, or null
  int* lp_stats;                 // [slots][NEP_MAX_POL][2]  (LPs attempted, LPs without a line) per segment, written by the separator
  // spatial presolve (line presolve on, largest-gap rule, batched hull layout): LPs whose line is known to be far from the guess
  // without solving them are skipped; a replan whose solution does not verify them is listed for the redo pass
  const double* skip_box;        // = fe_box when LPs may be skipped, else null
  int* line_skip;                // [slots][NEP_MAX_POL] LPs skipped per segment, or null
  int* redo_list;                // [slots] replans to solve again with every LP and every row, or null
  int* redo_count;               // [1]
  const int* order_count;        // [1] or null: only the first *order_count workgroups of the QP launch have work (the redo pass)
  int lines_override;            // 1: line buckets were filled by the host (test hook)
  int sep_pack;                  // host only: segments per wave of the presolve's separator — 0 picked by launch size, -1 the unpacked kernel, 1..NEP_MAX_POL forced (nep_batch_debug_set_separator_pack; NEP_SEP_PACK / NEP_SEP_UNPACKED at create)
  const int* order;              // [slots] workgroup -> slot (longest expected solve first, see order_kernel) or null: identity
  int* order_key;                // [slots] this launch's measured device time in 8 us bins (the next launch's ordering key) or null
  // front end: the same for the searches (frontend_kernel: longest expected search first; key bins of 8 us, 256 us with the entangle check)
  const int* fe_order;           // [slots] workgroup -> slot or null: identity
  int* fe_order_key;             // [slots] or null
  float* fe_us;                  // [slots] device time of the last search of every slot, microseconds, or null
  // QP scratch when the row state does not fit LDS
  double* row_scratch;           // [slots or scratch_chunks][11][rows_cap / 4 + 2]
  int scratch_chunks;            // 0: one scratch area per slot.  > 0 (presolve with the redo pass): that many areas, used by the redo pass only, by launch index
  int scratch_by_block;          // 1: this launch addresses the scratch by blockIdx.x (the redo pass)
  int rows_cap;
  int lds_rows;                  // rows that fit the dynamic LDS carve
  int lds_lines;
  // outputs
  nep_solution* solution;        // [slots]
  double* states;                // [slots][max_states][12] or null
  nep_traj_rec* commit;          // [slots] or null
  const nep_traj_rec* prev_commit; // [scenes][N] records before this round, or null: what a failed replan's commit slot carries over
  // active-set polish (qp_polish_kernel.hip): solves that ended without the strict tests leave their last iterate and are listed
  double* polish_z;              // [slots][2][24] last iterate of the first / relaxed solve (axis stride nz), or null: no polish
  int* polish_flag;              // [slots] bit m: mode m's solve ended on the loose snapshot or gave up
  int* polish_list; int* polish_count;      // [slots] listed slots; counters (see qp_polish_kernel)
  int* presolved;                // [slots] 1: qp_presolve_kernel finished this replan (its certificate held); the interior-point kernel returns at once.  Null: that kernel did not run
  long long* dbg;                // [slots][16] phase cycle counters (development aid) or null
  int* flags;                    // [1] sticky NEP_FLAG_* bits raised by the kernels (capacity overflows), or null
  double* fe_box;                // [scenes][num_agents + n_static][num_pol][4] (x0, x1, y0, y1) of the front end's obstacles (fe_box_kernel)
};
constexpr int NEP_FLAG_ENT_POOL = 16;       // the safety pass's entangle re-check needed a big record and the pool had none left (nep_batch_set_fe_ent_big_records): the trajectory was turned down
constexpr int NEP_FLAG_LINES = 8;           // a segment got more separating lines than its bucket holds (nep_batch_set_line_capacity)
constexpr int NEP_FLAG_SCRATCH = 4;         // more replans went through the presolve's redo pass with rows beyond the register slots than the handle has scratch areas for (nep_batch_reserve_row_scratch)
constexpr int NEP_FLAG_ENT_BETA = 2;        // an entangle state handed to the front end carries a non-zero beta for an agent crossing (the reference's rule makes it 0.0)
constexpr int NEP_FLAG_HULL_OVERFLOW = 1;   // an interval overlapped more committed segments than NEP_HULL_MAX_CP / 4, or its hull has more than NEP_HULL_MAX_V vertices

// a (slot, segment) line count as the separator stores it: n, or -1 - n when the bucket overflowed (the replan then fails)
__host__ __device__ inline int line_count(int stored) { return stored < 0 ? -1 - stored : stored; }
// entry index (scene-major inside its block) and byte offset of the block of agent j's hull data
struct HullRef { long e; long boff; };
__host__ __device__ inline HullRef hull_ref(const ProblemSet& ps, int per_scene, int scene, int j) {
  if (ps.hull_pb <= 0) return HullRef{(long)scene * per_scene + j, 0L};
  // (j / hull_pb by the host's magic number, exact for j, hull_pb < 65536: an integer division is a forty-instruction float
  // sequence on the VALU, and the separator takes a hull reference per candidate)
  const int b = ps.hull_pb_magic ? (int)(((unsigned long long)(unsigned)j * ps.hull_pb_magic) >> 32) : j / ps.hull_pb;
  return HullRef{(long)scene * ps.hull_pb + (j - b * ps.hull_pb), (long)b * ps.hull_bstride};
}
template <typename T> __host__ __device__ inline T* blk(T* base, long boff) { return (T*)((char*)base + boff); }

// hipFuncAttributeMaxDynamicSharedMemorySize belongs to a (device, function) pair: the size configured so far is kept
// per device (handles may live on several GPUs of one process) and updated under a lock (handles may be driven from
// several host threads).
struct DynLdsAttr {
  std::mutex mu; size_t configured[64] = {};
  hipError_t ensure(const void* fn, size_t bytes) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    std::lock_guard<std::mutex> lk(mu);
    if (bytes <= configured[dev]) return hipSuccess;
    const hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    if (e == hipSuccess) configured[dev] = bytes;
    return e;
  }
};

struct SampleSched {             // per K: n, seg[], dt[]
  const int* n;                  // [kMaxK+1]
  const int* seg;                // [kMaxK+1][max_states]
  const double* dt;              // [kMaxK+1][max_states]
};

// boxes: the hull kernel also writes the hulls' boxes (ps.fe_box, entries of the agents: what fe_box_kernel would make of them) and zeroes the
// presolve's redo counters — only honoured by the eight-hulls-per-wave kernel (hulls_grouped) with one hull list per agent of the scene
bool hulls_grouped(const SceneParams& sp, int n_scenes, int n_rec);
void launch_hulls(const nep_traj_rec* recs, int n_scenes, int n_rec, const nep_guess* guess,
                  const SceneParams& sp, const ProblemSet& ps, hipStream_t st, bool boxes = false);
void launch_hulls_ts(const nep_traj_rec* recs, int n_scenes, int n_rec, const double* ts0, long ts_slot_stride,
                     const SceneParams& sp, const ProblemSet& ps, hipStream_t st, bool boxes = false);
void launch_hulls_explicit(const nep_traj_rec* recs, int n_traj, double t_start, int num_pol,
                           double T_span, double drone_radius, double* hull_xy, int* hull_nv,
                           double* hull0_xy, int* hull0_nv, int* flags, hipStream_t st);
void launch_separator(int n_slots, const SceneParams& sp, const ProblemSet& ps, hipStream_t st);
void launch_separator_redo(int n_slots, const SceneParams& sp, const ProblemSet& ps, hipStream_t st);
void launch_boxes(int n_scenes, const SceneParams& sp, const ProblemSet& ps, hipStream_t st);
void launch_active_rows(int n_slots, const SceneParams& sp, const ProblemSet& ps, double tol, int* out, hipStream_t st);
void launch_separator_explicit(int n_prob, const int* a_off, const double* a_xy, const int* b_off,
                               const double* b_xy, double* nd, int* solved, int rule, hipStream_t st);
void launch_qp(int n_slots, const SceneParams& sp, const ProblemSet& ps, const QpTable* tables,
               const SampleSched& sched, size_t lds_bytes, hipStream_t st);
size_t qp_lds_fixed_bytes();
// the register-resident placement of the same solver (qp_reg_kernel.hip)
void launch_qp_order(int n_slots, const int* key, int* order, hipStream_t st, int* zero_these = nullptr);      // (zero_these: four ints zeroed on the way, or null)
void launch_qp_polish_zero(int* counters, hipStream_t st);
void launch_order_xcd(int n_slots, const int* key, int* order, hipStream_t st);      // (key may be null: the XCD placement alone; n_slots % 8 == 0)
void launch_qp_reg(int n_slots, const SceneParams& sp, const ProblemSet& ps, const QpTable* tables,
                   const SampleSched& sched, size_t lds_bytes, hipStream_t st);
void launch_qp_polish(int n_slots, const SceneParams& sp, const ProblemSet& ps, const QpTable* tables, const SampleSched& sched, hipStream_t st);
void launch_qp_presolve(int n_slots, const SceneParams& sp, const ProblemSet& ps, const QpTable* tables, const SampleSched& sched, int* presolved, hipStream_t st);
int qp_reg_slots();
size_t qp_reg_lds_bytes();
// pool of big records (ent_device.h: a search node's entangle state beyond the fixed record's capacities), claimed by atomic
// increments of *count during a launch; the launch's first kernel zeroes the counter
struct EntBigPool { unsigned char* base; int* count; int n_rec; int rec_bytes; int cap; int add_lim; };
// (mult: 1 for the search, 3 for the safety pass's re-check — entangleCheckGivenPwp prunes at three times the search's bound, kinodynamic_search.cpp:944-948)
__host__ __device__ inline int ent_big_cap(int N, int S, int mult = 1) { int c = (N + S) * mult; if (c < NEP_FE_ENT_CAP) c = NEP_FE_ENT_CAP; return (c + 3) & ~3; }      // (any fixed record can be copied into one)
__host__ __device__ inline int ent_big_add_lim(int N, int S, int mult = 1) { return ((N + S) * mult + 2 + 31) & ~31; }
__host__ __device__ inline int ent_big_rec_bytes(int N, int S, int mult = 1) { const int c = ent_big_cap(N, S, mult), a = ent_big_add_lim(N, S, mult); return (((8 + 13 * c + 3) & ~3) + 4 * a + a / 8 + 7) & ~7; }
// entangle inputs / scratch of the front end and of the safety pass's entangle re-check (device pointers)
struct FeEntArgs {
  const double* sampled;         // [scenes][N][num_pol][ns+1][2]   SampledPtsForAll_ (ent_sample_kernel); with sharded hulls: block 0's
  const int* present;            // [scenes][N]                     (both indexed through hull_ref: they travel in the hull blocks)
  const double* srep;            // [S][2][2] staticObsRep_ (x n_scenes with per-scene statics)
  const double* slong;           // [S][2]    staticObsLongestDist_
  const nep_fe_ent_state* init;  // [slots] entangle state at point A, or null (empty)
  nep_fe_ent_state* nodes;       // [slots][num_pol+1][beam_width] states of the installed nodes
  nep_fe_ent_state* work;        // [scenes * N] one working record per thread of ent_check_kernel
  nep_fe_ent_state* saved;       // [slots][children cap] the state every surviving child of the depth at hand arrived with (frontend_children_cap)
  double* saved_arc;             // [slots][children cap] and its sampled arc length
  double* st_f; long long* st_vox;      // [slots][children cap] f and voxel key of the children that survive the propagation pass: parked here while the LDS arrays that hold them (s_f, s_vox) are lent to the crossing lists
  int* case_out;                 // [slots][NEP_MAX_POL][N] (out) or null
  int ns;                        // num_sample_per_interval
  double* packed;                // [scenes][N][num_pol][pk_stride] one record per (agent, interval) of what the check reads (ent_pack_kernel), or null
  int pk_stride;
  EntBigPool big;                // front end: where states beyond the fixed record go (base == null: such children are pruned and flagged)
  EntBigPool big_check;          // the same for the safety pass's re-check (records of three times the bound; its counter is zeroed by ent_sample_kernel)
  int big_lds_off;               // big-record instantiation: byte offset, in the dynamic LDS, of its per-thread lists (kEntBigLdsBytes each), or 0: none (the launch's LDS would not hold them)
  double* big_beta;              // [redo_cap][256][kEntBigLdsCap] betas of those lists
  int* redo_list; int* redo_count; int redo_cap;      // front end: searches in which a child outgrew the fixed record, listed by frontend_kernel<true, W> for frontend_kernel<true, 2, true> (beyond redo_cap: they stay flagged)
  unsigned* xpool; int xpool_stride;      // front end (fast instantiation): per slot, room for the lists of new crossings of a round's (child, step) pairs that do not fit the LDS pool ([slots][xpool_stride] words: kEntAddCap per pair), see cross_round
  int fast_cap, fast_add, fast_bend;      // front end: what the fixed record's path accepts (<= NEP_FE_ENT_CAP, 32, NEP_MAX_BEND; nep_batch_set_fe_ent_fast_caps — the tests shrink them to drive ordinary scenes through the big records)
};
size_t frontend_children_cap(const nep_fe_cfg& fc, int num_pol);
size_t frontend_ent_xpool_words(const SceneParams& sp, const nep_fe_cfg& fc, int ent_ns);      // words per slot of FeEntArgs::xpool      // children per depth of one search (beam_width x lattice)
// (order_buf: [slots] scratch for the launch order, used when the previous launch left its keys — have_history — and the launch is
// more than one wave of workgroups; null: slot order)
void launch_frontend(int n_slots, const SceneParams& sp, const ProblemSet& ps, const nep_fe_cfg& fc, const nep_fe_start* starts,
                     nep_guess* guess_out, nep_fe_result* res_out, const FeEntArgs* ea, hipStream_t st, int* order_buf = nullptr, bool have_history = false);
void launch_ent_sample(const nep_traj_rec* recs, int n_scenes, int N, const double* ts0, long ts_scene_stride, int num_pol, int ns, double T_span,
                       double* sampled, int* present, hipStream_t st, int* zero_this = nullptr);
void launch_ent_check(const SceneParams& sp, const ProblemSet& ps, const FeEntArgs& ea, const nep_traj_rec* fresh, int n_scenes, double cable, int* entangles, hipStream_t st);
void launch_next_starts(const nep_traj_rec* recs, int n_scenes, int N, int first_local, int n_local, double dt, nep_fe_start* starts,
                        double* alt, double r_switch, hipStream_t st);
void launch_gjk_explicit(int n_prob, const int* a_off, const double* a_xy, const double* b_xy, int* hit, hipStream_t st);
void launch_safety(const nep_traj_rec* prev, const nep_traj_rec* fresh, int n_scenes, int N, const SceneParams& sp, const ProblemSet& ps,
                   unsigned char* conflict, unsigned char* conflict_prev, const int* entangles, nep_traj_rec* final_out, int* accept_out, hipStream_t st);

}  // namespace nep
