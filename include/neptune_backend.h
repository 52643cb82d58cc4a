/* neptune_backend.h — C ABI of the MI355X-native NEPTUNE back-end trajectory optimizer.
 *
 * This is the drop-in boundary for ONE path of caomuqing/neptune: the back-end optimizer that
 * `Neptune::replanFull` drives (reference: neptune/src/neptune.cpp:1512-1529), i.e. the public
 * surface of `class PolySolverGurobi` (neptune/include/solver_gurobi_poly.hpp:25-49) plus the
 * separating-line LP it calls (submodules/separator/include/separator.hpp:18-48) and the hull
 * construction that feeds it (neptune/src/neptune.cpp:224-452, 639-664).
 *
 * Plain C: POD structs, plain pointers and sizes, no C++/torch/Eigen types.  All floating point
 * is IEEE fp64.  Every entry point returns an int status (>= 0 ok, < 0 misuse / HIP error; text
 * via nep_last_error()) and never throws across the boundary.
 *
 * Two levels:
 *   (1) per-agent handle  `nep_backend_*`  — one call per PolySolverGurobi method, host buffers
 *       in / host buffers out, blocking (what a cgo/ctypes/C++ shim of the reference binds);
 *   (2) batched handle    `nep_batch_*`    — all local agents of a node in one launch sequence on
 *       a HIP stream, device-resident inputs/outputs (what bench.py and the multi-GPU driver use).
 */
#ifndef NEPTUNE_BACKEND_H
#define NEPTUNE_BACKEND_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ------------------------------------------------------------------------------------------ */
/* Limits (compile-time capacities of the fixed-size records)                                  */
/* ------------------------------------------------------------------------------------------ */
#define NEP_MAX_POL 8       /* num_pol upper bound; every shipped yaml uses 8
                               (neptune/param/neptune_mtlp_benchmark.yaml:73)                   */
#define NEP_TRAJ_MAX_SEG 16 /* committed trajectory = remainder of previous plan + <= 8 new
                               segments (neptune/src/utils.cpp:318-402)                         */
#define NEP_HULL_MAX_V 16   /* vertices of one interval hull (V <= 16)              */
#define NEP_HULL_MAX_CP 16  /* MINVO control points feeding one interval hull: <= 4 committed
                               segments overlap one planning interval x 4 control points (a
                               wave holds their 64 inflated corners); more is NEP_E_CAP        */
#define NEP_MAX_BEND 8      /* tether bend points kept per agent                               */
#define NEP_STATE_DOUBLES 12 /* mt::state pos,vel,accel,jerk (mader_types.hpp:35-41)           */

/* status of one replan (mirrors PolySolverGurobi::optimize, solver_gurobi_poly.cpp:804-887)   */
#define NEP_OK 0            /* first solve succeeded                      -> optimize()==true  */
#define NEP_RELAXED 1       /* first failed, relaxed re-solve succeeded   -> optimize()==true  */
#define NEP_FAILED 2        /* both failed: output == initial guess       -> optimize()==false */

/* error codes (< 0) */
#define NEP_E_ARG (-1)
#define NEP_E_STATE (-2)    /* call-sequence misuse (e.g. optimize before setInitTrajectory)   */
#define NEP_E_HIP (-3)
#define NEP_E_CAP (-4)      /* a capacity above was exceeded                                   */

/* ------------------------------------------------------------------------------------------ */
/* POD records                                                                                 */
/* ------------------------------------------------------------------------------------------ */

/* mt::PieceWisePol (mader_types.hpp:462-548) with [a b c d] per interval, t in real seconds
 * measured from the interval start (solver_gurobi_poly.cpp:921, kinodynamic_search.cpp:535-545). */
typedef struct nep_pwp {
  int32_t n_seg;                              /* number of intervals (coeff_x.size())           */
  int32_t _pad;
  double times[NEP_TRAJ_MAX_SEG + 1];         /* n_seg+1 knots                                   */
  double coeff[3][NEP_TRAJ_MAX_SEG][4];       /* [axis x,y,z][interval][a b c d]                 */
} nep_pwp;

/* Committed trajectory of one agent as exchanged between agents.  Mirrors mader_msgs/DynTraj
 * (mader_msgs/msg/DynTraj.msg:1-9, PieceWisePolTraj.msg:1-4); this is the all-gather record.   */
typedef struct nep_traj_rec {
  int32_t id;                                 /* 1-based agent id                                */
  int32_t is_agent;                           /* only is_agent==1 produces hulls (neptune.cpp:332) */
  int32_t n_bend;                             /* bend points incl. base (neptune_ros.cpp:457-476) */
  int32_t valid;                              /* 0: not (yet) received -> skipped like an id
                                                 missing from trajs_ (neptune.cpp:244-262)        */
  double bbox[3];
  double pos[3];
  double bend[NEP_MAX_BEND][2];
  nep_pwp pwp;
} nep_traj_rec;

/* Constructor + setMaxValues/setMaxRuntime/setTetherLength arguments
 * (solver_gurobi_poly.cpp:25-27,140-185; call site neptune.cpp:102-107).                        */
typedef struct nep_backend_cfg {
  int32_t num_pol;                /* <= NEP_MAX_POL */
  int32_t deg_pol;                /* must be 3 (yaml: "Only 3 is supported") */
  int32_t id;                     /* 1-based */
  int32_t num_agents;             /* pb.size() */
  double T_span;
  double weight_term;
  double rad_term;                /* stored, unused by the QP (solver_gurobi_poly.cpp:32,703-706) */
  int32_t use_linear_constraints; /* must be 1; the bilinear variant is out of scope */
  int32_t _pad;
  const double* pb;               /* [num_agents][2] base positions, copied */
} nep_backend_cfg;

/* eu::ent_state of one knot (entangle_utils.hpp:23-29), flattened: only the fields the back end
 * reads (solver_gurobi_poly.cpp:624-636): alphas (agent_id, case) pairs and active_cases[].    */
typedef struct nep_ent_view {
  int32_t n_states;               /* K+1 */
  int32_t n_active;               /* length of each active_cases row (num_agents + statics)     */
  const int32_t* alpha_off;       /* [n_states+1] CSR offsets into alphas                        */
  const int32_t* alphas;          /* [alpha_off[n_states]][2]                                    */
  const int32_t* active_cases;    /* [n_states][n_active]                                        */
  const int32_t* bend_off;        /* [num_agents+1] CSR offsets into bend_xy                     */
  const double* bend_xy;          /* [bend_off[num_agents]][2]  bendPtsForAgents                 */
} nep_ent_view;

typedef struct nep_stats {
  int32_t status;                 /* NEP_OK / NEP_RELAXED / NEP_FAILED */
  int32_t iters;                  /* interior-point iterations of the accepted solve            */
  int32_t iters_first;            /* iterations spent in the first (failed or accepted) solve   */
  int32_t n_lines;                /* separating lines that produced constraint rows             */
  int32_t n_lp;                   /* separator LPs attempted (Separator::getNumOfLPsRun)        */
  int32_t n_lp_failed;            /* LPs without a separating line -> constraint skipped
                                     (solver_gurobi_poly.cpp:483-494)                           */
  int32_t n_rows;                 /* inequality rows of the QP                                  */
  int32_t qc_active;              /* terminal ball constraint present (solver_gurobi_poly.cpp:699) */
  double objective;               /* objective_value (solver_gurobi_poly.cpp:882)               */
  double solve_us;                /* batched handle: device time of the replan's QP workgroup; per-agent handle: wall time of
                                     optimize() as the caller's clock sees it (neptune.cpp:1504,1528), microseconds */
} nep_stats;

/* ------------------------------------------------------------------------------------------ */
/* (1) per-agent handle: one entry per PolySolverGurobi method                                 */
/* ------------------------------------------------------------------------------------------ */
typedef struct nep_backend nep_backend_t;

/* PolySolverGurobi::PolySolverGurobi (solver_gurobi_poly.cpp:25-134).  NULL on failure.        */
nep_backend_t* nep_backend_create(const nep_backend_cfg* cfg);
void nep_backend_destroy(nep_backend_t* h);

/* setMaxValues (solver_gurobi_poly.cpp:140-175) */
int nep_backend_set_max_values(nep_backend_t* h, double x_min, double x_max, double y_min,
                               double y_max, double z_min, double z_max, double v_max,
                               double a_max, double j_max);
/* setMaxRuntime (:182-185) -> Gurobi's TimeLimit (:812): the wall-clock budget of ONE solve (the first and the relaxed
 * re-solve each get it).  A solve that has not converged when the budget is spent (device wall clock, checked once per
 * interior-point iteration) counts as "no solution" — the reference accepts a time-limited solve only with an incumbent
 * (:832-836), which a barrier QP does not have before it converges.  The interior point is also bounded at 60
 * iterations (tens of microseconds), so the reference's 0.05 s never binds in practice.                        */
int nep_backend_set_max_runtime(nep_backend_t* h, double seconds);
/* setTetherLength (:177-180), stored only, as in the reference. */
int nep_backend_set_tether_length(nep_backend_t* h, double tether_length);
/* setStaticObstVert (:316-320): n polygons, CSR offsets (in vertices) + xy[off[n]][2].         */
int nep_backend_set_static_obst_vert(nep_backend_t* h, int32_t n_obst, const int32_t* vert_off,
                                     const double* xy);
/* setInitTrajectory (:187-244).  K = pwp->n_seg <= num_pol.                                    */
int nep_backend_set_init_trajectory(nep_backend_t* h, const nep_pwp* pwp_init);
/* setHulls (:246-281): hulls[j][i], j < n_obst (other agents present), i < num_pol.  CSR over
 * (j*num_pol+i).                                                                                */
int nep_backend_set_hulls(nep_backend_t* h, int32_t n_obst, const int32_t* vert_off,
                          const double* xy);
/* setHullsNoInflation (:283-288): id-indexed, num_agents x num_pol, empty polygons allowed.    */
int nep_backend_set_hulls_no_inflation(nep_backend_t* h, int32_t n_agents,
                                       const int32_t* vert_off, const double* xy);
/* setEntStateVector (:307-314).  ent == NULL clears (entangle check off).                      */
int nep_backend_set_ent_state_vector(nep_backend_t* h, const nep_ent_view* ent);
/* optimize (:804-887).  Returns NEP_OK / NEP_RELAXED / NEP_FAILED (or < 0); *objective_value is
 * written only when the return is NEP_OK or NEP_RELAXED, exactly like the reference.           */
int nep_backend_optimize(nep_backend_t* h, double* objective_value);
/* generatePwpOut (:889-936): pwp_out->times shifted by t_start; states_out[cap][12] receives
 * pos,vel,accel,jerk every dc; *n_states_out = number written.                                 */
int nep_backend_generate_pwp_out(nep_backend_t* h, double t_start, double dc, nep_pwp* pwp_out,
                                 double* states_out, int32_t states_cap, int32_t* n_states_out);
/* Neptune::getPlanningStats side channel (neptune.cpp:1812-1819) + solver counters.            */
int nep_backend_get_stats(nep_backend_t* h, nep_stats* out);

/* ------------------------------------------------------------------------------------------ */
/* Stand-alone kernels of the path (batched, host buffers): used by the parity tests           */
/* ------------------------------------------------------------------------------------------ */

/* Neptune::setStaticObst (neptune.cpp:639-664), the one-time host step in front of setStaticObstVert: every vertex of
 * footprint j (CSR vert_off / xy, as nep_backend_set_static_obst_vert takes them) is pushed out by
 * safe_dist = 2 drone_radius + 0.2 to four corners and the convex hull of those (cu::convexHullOfPoints2d,
 * cgal_utils.cpp:157-174: counter-clockwise from the lexicographically smallest point, collinear points dropped) is written
 * to out_off [n_obst + 1] / out_xy [cap][2] — what nep_backend_set_static_obst_vert / nep_batch_cfg expect.  Host only (no
 * HIP call, as in the reference).  Returns the number of vertices written, NEP_E_CAP when cap or NEP_HULL_MAX_V is exceeded. */
int nep_inflate_static(int32_t n_obst, const int32_t* vert_off, const double* xy, double drone_radius,
                       int32_t* out_off, double* out_xy, int32_t cap);

/* Separator::solveModel 2-D, 2-set (separator_glpk.cpp:248-373), batched: problem p has point
 * set A = a_xy[a_off[p]..a_off[p+1]) and B = b_xy[b_off[p]..b_off[p+1]).  nd_out[p] = (n1,n2,d),
 * solved_out[p] = 1/0.  The 3-set overload (:375-498) is the same LP with A := A u A+.  As at
 * every call site of the path, B must hold exactly 4 points (a segment's control points) and A at
 * most NEP_HULL_MAX_V.                                                                          */
int nep_separator_batch(int32_t n_prob, const int32_t* a_off, const double* a_xy,
                        const int32_t* b_off, const double* b_xy, double* nd_out,
                        int32_t* solved_out);
/* The same with the vertex rule named (see nep_batch_set_separator_rule): 0 = nep_separator_batch, 1 = GLPK-class simplex. */
int nep_separator_batch_rule(int32_t rule, int32_t n_prob, const int32_t* a_off, const double* a_xy,
                             const int32_t* b_off, const double* b_xy, double* nd_out,
                             int32_t* solved_out);

/* Batched gjk::collision(vertices1, vertices2) (gjk.cpp:76-149) with vertices1 = polygon p of the CSR
 * (a_off, a_xy) and vertices2 = the four points b_xy[p][4][2] — the call shape of the safety check
 * (neptune.cpp:794) and of the front end's collision test (kinodynamic_search.cpp:1514-1553).    */
int nep_gjk_batch(int32_t n_prob, const int32_t* a_off, const double* a_xy, const double* b_xy,
                  int32_t* hit_out);

/* Neptune::convexHullsOfCurve2d (neptune.cpp:269-452) for n_traj committed trajectories over
 * num_pol intervals of [t_start, t_start+num_pol*T_span]: inflated hull and uninflated hull per
 * (traj, interval).  hull_xy[(j*num_pol+i)][NEP_HULL_MAX_V][2], hull_nv[(j*num_pol+i)].        */
int nep_hulls_batch(int32_t n_traj, const nep_traj_rec* trajs, double t_start, int32_t num_pol,
                    double T_span, double drone_radius, double* hull_xy, int32_t* hull_nv,
                    double* hull0_xy, int32_t* hull0_nv);

/* ------------------------------------------------------------------------------------------ */
/* (2) batched handle: all local agents of one node per launch sequence                        */
/* ------------------------------------------------------------------------------------------ */
typedef struct nep_batch nep_batch_t;

typedef struct nep_batch_cfg {
  int32_t num_agents;             /* N: agents in the scene (ids 1..N)                          */
  int32_t first_local;            /* 0-based index of the first agent solved by this handle     */
  int32_t n_local;                /* agents solved by this handle (N/G when sharded)            */
  int32_t num_pol;
  int32_t n_static;               /* S: inflated static obstacle polygons                       */
  int32_t enable_entangle;        /* 0/1                                                        */
  int32_t max_states;             /* capacity of the sampled state list per agent               */
  int32_t n_scenes;               /* independent scenes processed per launch (>= 1); agent slot
                                     (scene s, local agent a) = s*n_local + a                   */
  double T_span, weight_term, dc, drone_radius;
  double x_min, x_max, y_min, y_max, z_min, z_max, v_max, a_max;
  const double* pb;               /* [N][2] bases (host, copied)                                */
  const int32_t* static_off;      /* [S+1] (host, copied)                                       */
  const double* static_xy;        /* [static_off[S]][2] (host, copied)                          */
} nep_batch_cfg;

/* Front-end result for one agent: initial guess + replan start time + entangle inputs.        */
typedef struct nep_guess {
  int32_t K;                                  /* num_pol_init_ (solver_gurobi_poly.cpp:194)      */
  int32_t n_alpha;                            /* alphas per knot are stored dense below          */
  double t_start;                             /* neptune.cpp:1422                                 */
  double coeff[3][NEP_MAX_POL][4];            /* pwp_init, times are i*T_span                     */
} nep_guess;

typedef struct nep_solution {
  nep_stats stats;
  int32_t K;
  int32_t n_states;
  double times[NEP_MAX_POL + 1];              /* t_start + i*T_span                               */
  double coeff[3][NEP_MAX_POL][4];            /* pwp_out                                          */
} nep_solution;

nep_batch_t* nep_batch_create(const nep_batch_cfg* cfg);
void nep_batch_destroy(nep_batch_t* h);

/* Static obstacles (nep_batch_cfg and nep_backend_set_static_obst_vert alike): each polygon must be convex with
 * at most NEP_HULL_MAX_V vertices — what Neptune::setStaticObst hands over (neptune.cpp:639-664).  The separator
 * works on counter-clockwise polygons; clockwise input is reversed at upload (first vertex kept: it feeds the
 * proximity cull, solver_gurobi_poly.cpp:559), non-convex input is refused with NEP_E_ARG.  The same holds for the
 * hull lists of nep_backend_set_hulls.
 *
 * nep_batch_set_scene_statics gives scene `scene` (0 <= scene < n_scenes) its own set of n_static == cfg.n_static
 * polygons (host CSR, copied); scenes never set keep the set of nep_batch_cfg.  Blocking (device synchronize).  */
int nep_batch_set_scene_statics(nep_batch_t* h, int32_t scene, int32_t n_static, const int32_t* static_off,
                                const double* static_xy);

/* One full back-end replan for every (scene, local agent) slot, enqueued on `stream`
 * (a hipStream_t passed as void*; NULL = default stream), asynchronous.
 *   d_committed : device, [n_scenes][N] nep_traj_rec — snapshot of every agent's committed
 *                 trajectory (own entry ignored); NULL = reuse the interval hulls the handle built
 *                 in nep_batch_frontend for this round (include/neptune_frontend.h)
 *   d_guess     : device, [n_scenes][n_local] nep_guess
 *   d_ent       : device or NULL, entangle inputs (layout: see nep_batch_ent_bytes)
 *   d_solution  : device, [n_scenes][n_local] nep_solution                        (out)
 *   d_states    : device, [n_scenes][n_local][max_states][12] or NULL             (out)
 *   d_commit    : device or NULL, [n_scenes][n_local] nep_traj_rec: the new trajectory as the
 *                 record the agent would publish (neptune_ros.cpp:434-480)         (out)
 * A replan that fails (status NEP_FAILED: both solves infeasible, or a guess with K < 1 or K > num_pol — a
 * front-end miss — which is never solved) publishes nothing, as in the reference (neptune_ros.cpp:651-663): its
 * d_commit slot receives the agent's PREVIOUS record (from d_committed, or from the records nep_batch_frontend
 * built this round's hulls from); when neither is known (nep_batch_replan_hulls) the slot is left exactly as the
 * caller passed it, so hand in the buffer that still holds the previous round's records.  d_solution of such a
 * slot: status NEP_FAILED, coefficients = the guess (all zero and K = 0 for an unusable guess).
 * Lifetime: with d_committed == NULL the handle reads the record buffer that was passed to the preceding
 * nep_batch_frontend / nep_batch_frontend_ent call (the pointer is kept, nothing is copied): that buffer must stay
 * allocated and unchanged until this replan has completed on `stream` — do not gather new records into it in between. */
int nep_batch_replan(nep_batch_t* h, const nep_traj_rec* d_committed, const nep_guess* d_guess,
                     const void* d_ent, nep_solution* d_solution, double* d_states,
                     nep_traj_rec* d_commit, void* stream);
/* The same round as two enqueues, so that a host with several scene groups in flight (one handle per group) can put the
 * halves on different streams: nep_batch_replan_lines = interval hulls + separating-line LPs into the handle's scratch
 * (the chip-filling, VALU-bound half), nep_batch_replan_solve = the QPs on those lines + sampled states + commit records (a few
 * long solves on a mostly idle chip at its end: another group's _lines fits under it).  _lines then _solve with the same
 * arguments == nep_batch_replan, bit for bit; the caller orders the two (same stream, or an event), keeps d_committed /
 * d_guess / d_ent unchanged in between, and does not start the handle's next _lines before its _solve has completed.
 * bench.py's headline runs its scene groups this way.                                            */
int nep_batch_replan_lines(nep_batch_t* h, const nep_traj_rec* d_committed, const nep_guess* d_guess,
                           const void* d_ent, void* stream);
int nep_batch_replan_solve(nep_batch_t* h, const nep_traj_rec* d_committed, const nep_guess* d_guess,
                           const void* d_ent, nep_solution* d_solution, double* d_states,
                           nep_traj_rec* d_commit, void* stream);

/* Sharded hulls for multi-GPU rounds.  Instead of every rank rebuilding the hulls of all N
 * committed trajectories, a rank computes the interval hulls of ITS n_local agents (every scene)
 * into one block of nep_batch_hull_block_bytes() bytes, the blocks of all ranks are all-gathered
 * (rank order = agent-id order, dist.shard), and nep_batch_replan_hulls runs the separator and the
 * QP against the n_blocks = N / n_local gathered blocks.  Bit-identical to nep_batch_replan.
 *   d_committed_local : device, [n_scenes][n_local] nep_traj_rec of the local agents
 *   d_block           : device, one block (out);   d_blocks : device, n_blocks consecutive blocks */
int64_t nep_batch_hull_block_bytes(const nep_batch_t* h);
/* Handles created with enable_entangle: the block also carries what the entangle check reads of a committed trajectory — its
 * samples (Neptune::SamplePointsOfIntervals, neptune.cpp:500-565: ns + 1 points per interval) and whether it exists — so
 * that nep_batch_frontend_ent_hulls and nep_batch_safety_commit_ent work on sharded handles.  ns = num_sample_per_interval,
 * 3 by default (the reference's yaml); set it before the first nep_batch_hull_block_bytes / nep_batch_hulls call.     */
int nep_batch_set_ent_samples(nep_batch_t* h, int32_t ns);
int nep_batch_hulls(nep_batch_t* h, const nep_traj_rec* d_committed_local, const nep_guess* d_guess,
                    void* d_block, void* stream);
int nep_batch_replan_hulls(nep_batch_t* h, const void* d_blocks, int32_t n_blocks,
                           const nep_guess* d_guess, const void* d_ent, nep_solution* d_solution,
                           double* d_states, nep_traj_rec* d_commit, void* stream);

/* ------------------------------------------------------------------------------------------ */
/* The exchange step of a multi-GPU round (RCCL all-gather over xGMI), for hosts without PyTorch */
/* ------------------------------------------------------------------------------------------ */
/* One process per GPU; rank r owns agents [r n_local, (r+1) n_local) of every scene (nep_batch_cfg.first_local).  The
 * reference exchanges committed trajectories on the /trajs topic (neptune_ros.cpp:434-480 publish, :379-430
 * receive); here one all-gather per round on the caller's stream does.  RCCL is bound at run time (dlopen of
 * librccl.so: the copy the process already holds, or the system's); without it these calls return NEP_E_HIP and
 * nothing else of the library is affected.
 *   nep_comm_unique_id      rank 0 creates the 128-byte id (ncclGetUniqueId) and ships it to the others by any means
 *   nep_comm_create         ncclCommInitRank on the calling thread's current HIP device; NULL on failure
 *   nep_batch_exchange_hulls    d_block (this rank's block from nep_batch_hulls) -> d_blocks (world consecutive
 *                               blocks, rank order = agent-id order: what nep_batch_replan_hulls reads)
 *   nep_batch_exchange_records  d_commit_local [n_scenes][n_local] (nep_batch_replan's d_commit) ->
 *                               d_committed_all [n_scenes][N]: the literal "all-gather of committed trajectories"
 * Both are asynchronous on `stream`; a round is  nep_batch_hulls -> nep_batch_exchange_hulls ->
 * nep_batch_replan_hulls  (or  nep_batch_replan -> nep_batch_exchange_records).  Being plain enqueues on the caller's
 * stream, they can be captured into a HIP graph together with the kernels around them (RCCL collectives are capturable):
 * bench.py captures the whole per-rank step, exchange included, and replays it.                                  */
typedef struct nep_comm nep_comm_t;
int nep_comm_unique_id(uint8_t id_out[128]);
nep_comm_t* nep_comm_create(const uint8_t id[128], int32_t world, int32_t rank);
void nep_comm_destroy(nep_comm_t* c);
int nep_comm_nranks(nep_comm_t* c);                /* ranks that joined (ncclCommCount), or < 0 */
int nep_batch_exchange_hulls(nep_batch_t* h, nep_comm_t* c, const void* d_block, void* d_blocks, void* stream);
int nep_batch_exchange_records(nep_batch_t* h, nep_comm_t* c, const nep_traj_rec* d_commit_local,
                               nep_traj_rec* d_committed_all, void* stream);
/* The same all-gather + regrouping for ANY per-slot array: d_local [n_scenes][n_local][bytes_per_slot] of every rank ->
 * d_all [n_scenes][N][bytes_per_slot] (bytes_per_slot a positive multiple of 8) — e.g. the entangle states at point A that
 * nep_batch_safety_commit_ent reads of every agent.                                                               */
int nep_batch_exchange_slots(nep_batch_t* h, nep_comm_t* c, const void* d_local, void* d_all, int64_t bytes_per_slot, void* stream);
/* nep_batch_exchange_records and nep_batch_exchange_slots regroup through a staging buffer owned by the communicator (one per
 * kind of exchange, so a records exchange and a slots exchange may be in flight on different streams; two exchanges of the SAME
 * kind must be ordered on one stream).  A buffer grows on demand — but never while `stream` is capturing (the call then returns
 * NEP_E_ARG): reserve the gathered sizes (world x one rank's piece, in bytes; 0 = leave as is) before capturing a graph, or run
 * the exchange once eagerly.  A graph captured earlier holds the pointer it was captured with: do not grow a buffer afterwards. */
int nep_comm_reserve(nep_comm_t* c, int64_t records_bytes, int64_t slots_bytes);

/* Dense per-slot entangle block consumed by nep_batch_replan when enable_entangle != 0:
 *   int32 case_id[NEP_MAX_POL][N]   (0 = no active case for that agent at that segment,
 *                                    else the alphas case id, solver_gurobi_poly.cpp:624-631)
 * bend points come from d_committed[j].bend / n_bend.                                          */
int64_t nep_batch_ent_bytes(const nep_batch_t* h);

/* SURVEY §8(f) rank 1 — post-solve safety check + commit for a bulk-synchronous round
 * (Neptune::safetyCheckAfterReplan / trajsAndPwpAreInCollision2d, neptune.cpp:719-806;
 * gjk::collision, gjk.cpp:76-149).  d_prev, d_new: [n_scenes][N] records before / after the round
 * (d_new = the all-gathered d_commit of nep_batch_replan); every other agent's new trajectory
 * counts as "received while optimizing".  An agent keeps its new trajectory unless it collides
 * (either direction) with an accepted lower id; otherwise its previous record is kept
 * (neptune_ros.cpp:651-663).  d_final [n_scenes][N] receives the records to replan against next,
 * d_accept [n_scenes][N] (may be NULL) the 1/0 flags.  Asynchronous on `stream`; overwrites the
 * handle's hull scratch.  d_guess as passed to nep_batch_replan (supplies t_start).           */
int nep_batch_safety_commit(nep_batch_t* h, const nep_traj_rec* d_prev, const nep_traj_rec* d_new,
                            const nep_guess* d_guess, nep_traj_rec* d_final, int32_t* d_accept,
                            void* stream);
/* Test hook: the conflict matrix [N][N] of one scene from the last safety check. */
/* Presolve of the separating-line rows (off by default, radius = 0).  With radius > 0 (metres) a line whose
 * boundary lies farther than `radius` from all four control points of the guess's segment is left out of the
 * QP; after the solve every left-out line is checked against the solution's control points and, if one is
 * violated, the replan is solved again with all lines — the optimum is that of the full problem (what a QP
 * presolve does with redundant rows; Gurobi runs one inside PolySolverGurobi::optimize).  nep_stats.n_lines
 * still counts every line, n_rows the rows actually solved for.  The debug line readers return, per segment,
 * the near lines in call order followed by the parked ones in call order.
 * The presolve also tries the minimiser of the cost without any inequality row (one small matrix-vector product
 * with a host-built inverse): if every box row, every line and the terminal ball hold there, that point with zero
 * multipliers satisfies the KKT conditions of the full problem and is returned as the optimum without a single
 * interior-point iteration (nep_stats.iters == 0); otherwise the interior point runs as usual.
 * Default (round 6): ON with radius 4 m for every handle, batched and per-agent, at every scene size — this is the library's
 * one solve path, as the presolve inside GRBModel::optimize is the reference's (solver_gurobi_poly.cpp:823); the result is the
 * full problem's optimum by construction and the polish pass (below) runs under it.  Until round 5 it was on only for scenes
 * whose lines do not fit the register slots of the interior-point kernel (config 5).  An explicit call (any radius, 0 included)
 * overrides the default: radius 0 = every separating-line row through the interior point, lines in the reference's call order
 * (what the line-order parity tests ask for); nep_batch_get_line_cull returns the radius in force.  The per-agent handle has the
 * same setter (its hull lists are the caller's, so no LP is skipped there: parked lines and the zero-iteration test only).       */
int nep_batch_set_line_cull(nep_batch_t* h, double radius);
int nep_backend_set_line_cull(nep_backend_t* h, double radius);
double nep_batch_get_line_cull(nep_batch_t* h);
/* With the presolve on (largest-gap rule, batched handle) the separator does not even SOLVE the LPs whose line must be far: a
 * hull or inflated static whose bounding box is farther than the radius from the box of the guess's control points along x or
 * y (the sides of those boxes are edges of the polygons, so the largest-gap line lies at least that far away).  Such LPs are
 * counted as attempted and solved (n_lp, n_lines: sets that far apart are separable) and verified through the distance the
 * solution's control points moved from the guess's; a replan that does not verify is solved again by a redo pass with every
 * LP and every row.  Precondition of the box test, checked at upload: every static polygon has an edge on each side of its
 * bounding box (true of the inflated statics nep_inflate_static makes and of the interval hulls: hulls of axis-aligned squares);
 * a handle that is given any other convex polygon (a diamond, say) solves every LP — parked lines and the zero-iteration test
 * still apply, they evaluate real lines.  The debug line readers (neptune_backend_debug.h) do not see lines that were never made.       */
/* Row scratch (rows and coefficients of a replan beyond the interior-point kernel's register slots or LDS carve).  In general one
 * worst-case area per slot.  With the presolve's redo pass (line presolve on, largest-gap rule, batched hull layout, box-edged
 * statics) the first pass never needs one — a replan whose near lines exceed the slots goes to the redo pass unsolved — so a
 * handle of more than 1 024 slots keeps a pool of 1 024 areas for that pass (config 5, 32 scenes: 1.9 GB instead of 15.3).  A
 * launch that lists more such replans than the pool holds fails them and raises a sticky flag: nep_batch_check returns
 * NEP_E_CAP; nep_batch_reserve_row_scratch switches the handle to one area per slot for good.  Setters that change the mode
 * (nep_batch_set_line_cull, nep_batch_set_separator_rule) re-size the scratch: never call them inside a graph capture.        */
int nep_batch_reserve_row_scratch(nep_batch_t* h);
/* Separating lines a (replan, segment) bucket holds.  The reference's worst case is n_hull + N + S + 8 N (one line per hull, base,
 * static and (agent, bend segment) pair of the entangle rows); by default the buckets budget 2 N entangle lines per segment instead
 * of 8 N (an entangle line needs an active case for that agent and segment).  A segment that gets more raises a sticky flag —
 * nep_batch_check returns NEP_E_CAP, nothing is written past a bucket — and (h, -1) sizes for the worst case; (h, n) sets n; (h, 0)
 * the default.  Re-sizes buffers: never inside a graph capture.                                                              */
int nep_batch_set_line_capacity(nep_batch_t* h, int32_t lines_per_segment);

/* Which vertex of the separating-line LP the separator returns.  The LP (separator_glpk.cpp:248-373) has a zero objective:
 * the reference gets "whatever vertex glp_simplex reaches", and the spline QP's optimum depends on it.
 *   0 (default)  the vertex with the largest geometric gap between the two point sets — a rule that depends on the LP only;
 *   1            the vertex a primal simplex of the class glp_simplex runs by default reaches: standard start basis (row
 *                variables basic, n1 = n2 = d = 0 non-basic), projected steepest-edge pricing, Harris' two-pass ratio test,
 *                tol_bnd = tol_dj = 1e-7, no presolve, no scaling (glp_init_smcp defaults, separator_glpk.cpp:39-41, 336).
 *                GLPK 4.65's source is not in the reference tree, so this is the documented algorithm class, not a clone of
 *                GLPK's pivot sequence; for callers who want simplex-reached lines.
 * Both return a point that satisfies every row of the reference LP, or "no line" (the constraint is then skipped,
 * solver_gurobi_poly.cpp:483-494).  Bit-identical to oracle/'s restatement of either rule.                           */
int nep_batch_set_separator_rule(nep_batch_t* h, int32_t rule);
int nep_backend_set_separator_rule(nep_backend_t* h, int32_t rule);

/* The interior point's stopping tests.  Default (round 6; 1e-9 / 1e-10 until then): primal residual <= 1e-10, dual residual <= 1e-10 x the
 * cost's scale, duality gap <= 1e-11 (1 + |objective|) — at these, strictly converged solves of the same replan over two row sets (the
 * presolved and the every-row problem) or on two hosts (device and oracle) agree to 3e-8 in the coefficients where 1e-9 / 1e-10 left up
 * to 2e-6 on weakly determined optima (scripts/parity_sweep.py, NEP_TOL), for 0.3 iterations more per iterating solve.  Three orders
 * tighter than the solver the reference calls: PolySolverGurobi never touches
 * Gurobi's parameters beyond OutputFlag and TimeLimit (solver_gurobi_poly.cpp:811-812), so its barrier stops at Gurobi's
 * defaults, BarConvTol = 1e-8 (relative gap) with FeasibilityTol = OptimalityTol = 1e-6.  nep_*_set_tolerances(h, 1e-6, 1e-8)
 * stops where the reference's solver does: about half an iteration less per replan (the last iteration of a solve shrinks the
 * gap by 1e4 or more), answers within the reference's own accuracy instead of 1e-8 of the optimum.  The loose-snapshot rule
 *  keeps its thresholds (1e-6, 1e-6, 1e-7).  Same knob in oracle/ (orc_set_qp_tolerances).           */
int nep_batch_set_tolerances(nep_batch_t* h, double residual_tol, double gap_tol);
int nep_backend_set_tolerances(nep_backend_t* h, double residual_tol, double gap_tol);

/* The active-set polish (on by default).  An interior-point solve that never passes the strict tests ends either on its best LOOSELY
 * converged iterate (the dual residual sits on a rounding floor above 1e-9: the answer is then within ~1e-4 of the optimum in the
 * coefficients) or by giving up — which happens on infeasible problems, and on feasible ones whose optimum is degenerate.  With
 * the polish such a solve is finished exactly: the rows active at its last iterate define an equality-constrained QP that is solved
 * directly; rows with a negative multiplier leave, violated rows enter, a few times; a point that satisfies every row with
 * non-negative multipliers is the optimum (KKT) and the solve counts as converged — NEP_OK for the first problem even if the
 * interior point had gone on to the relaxed one, which is what a solver that finds the optimum reports (solver_gurobi_poly.cpp:
 * 832-861).  Without a certificate nothing changes.  Not applied to problems with the terminal ball row.  on = 1 (the default; 2 is
 * accepted as a synonym): on every path of the register-resident interior point, the line presolve included — there the pass works
 * on the near lines and accepts a certified point only if it passes the presolve's own verification (parked lines, movement bound
 * of the skipped LPs) again, so that presolved and every-row solves agree to 1e-6 on loose exits as well (9e-5 otherwise), at
 * 0.03-0.06 ms per launch; on = 3: round 5's default (every-row solves only, not under the presolve); on = 0: off.
 * SCOPE: the pass finishes solves of qp_reg_kernel.  The LDS-placement kernel (qp_kernel: only reachable with the presolve turned
 * off, nep_batch_set_line_cull(h, 0), at scene sizes whose lines exceed the register slots — config 5 with every row) keeps its
 * loose exits: nep_batch_debug_polish_count reports (0, 0) for such a launch, and a parity check against oracle/ must call
 * orc_set_polish(0) for it.  oracle/ runs the same rule (orc_set_polish).                                                            */
int nep_batch_set_polish(nep_batch_t* h, int32_t on);
int nep_backend_set_polish(nep_backend_t* h, int32_t on);

/* setMaxRuntime for the batched handle (0 = no wall-clock limit, the default): see nep_backend_set_max_runtime. */
int nep_batch_set_max_runtime(nep_batch_t* h, double seconds);

/* on != 0: nep_batch_safety_commit additionally turns down a new trajectory that collides with the
 * PREVIOUS record of any other agent (its hulls on the round's grid).  The spline QP keeps the two
 * apart wherever a separating line was found; where the LP had no solution the reference skips the
 * constraint (solver_gurobi_poly.cpp:483-494) and nothing else certifies the pair — an agent that is
 * turned down this round keeps flying exactly that previous trajectory.  Off by default.          */
int nep_batch_set_safety_check_prev(nep_batch_t* h, int32_t on);

/* Blocks until everything enqueued by this handle on `stream` has finished. */
int nep_batch_wait(nep_batch_t* h, void* stream);

/* Blocks like nep_batch_wait, then reports (and clears) capacity overflows the kernels met since the last call:
 * NEP_E_CAP when a planning interval overlapped more than NEP_HULL_MAX_CP / 4 committed segments or an interval hull
 * had more than NEP_HULL_MAX_V vertices (the reference uses every segment, neptune.cpp:392-449; the kernels flag the
 * overflow instead of under-covering silently), else 0.  The asynchronous entry points cannot return this themselves. */
int nep_batch_check(nep_batch_t* h, void* stream);

/* sizeof() of the POD records as compiled (0 nep_pwp, 1 nep_traj_rec, 2 nep_backend_cfg,
 * 3 nep_stats, 4 nep_batch_cfg, 5 nep_guess, 6 nep_solution, 7 nep_ent_view; from neptune_plan.h:
 * 8 nep_wire_header, 9 nep_plan_cfg, 10 nep_point_a; from neptune_frontend.h: 11 nep_fe_cfg,
 * 12 nep_fe_start, 13 nep_fe_result): lets a foreign-language binding verify its struct mirror. */
int nep_abi_sizeof(int32_t which);

const char* nep_last_error(void);
const char* nep_version(void);

#ifdef __cplusplus
}
#endif
#endif /* NEPTUNE_BACKEND_H */
