/* neptune_frontend.h — batched front-end initial-guess generator (SURVEY §8 f, rank 2).
 *
 * GPU counterpart of KinodynamicSearch (reference neptune/src/kinodynamic_search.cpp) for the batched
 * pipeline: every (scene, agent) slot searches the jerk lattice for a kinodynamically feasible,
 * collision-free K-segment cubic from its point A toward its goal and writes the nep_guess the back
 * end starts from, so that hulls -> guess -> separator -> QP runs on the device without a host
 * round trip (Neptune::replanFull, neptune.cpp:1302-1725, between point-A selection and commit).
 *
 * What is kept from the reference, rule by rule (kinodynamic_search.cpp):
 *   motion primitives   constant jerk on a num_samples x num_samples lattice in [-j_max, j_max]^2 over
 *                       T_span (:1045-1097, :1240-1275); primitive [a b c d] = [j/6, a0/2, v0, p0]
 *   pruning of a child  |end - start| < 1e-5, acceleration bounds, MINVO position control points in
 *                       the box and within the tether length of the base, MINVO velocity control points
 *                       in bounds (not for the first segment, as in the reference's initial overload),
 *                       the four "cannot brake in time" tests (:1079-1152, :1262-1330)
 *   collision           gjk::collision of the child's control polygon with the hull of every other
 *                       agent's committed trajectory over the child's interval (index clamped to
 *                       num_pol) and with every inflated static obstacle (:1514-1553)
 *   costs               g = travelled chord length, h = distance to the goal, order by g + bias*h
 *                       (:1177-1179, CompareCost); one node per voxel of size voxel_size (:1170-1173)
 *   termination         a node within goal_size of the goal ends the search (:1719-1737); the plan is
 *                       the first num_pol segments of the path (:521-553); z = Neptune::getInitialZPwp's clamped B-spline
 *                       ramp from A's height state to the goal height (neptune.cpp:1727-1810; as there,
 *                       the first three control points are placed for a clamped basis but evaluated with
 *                       the uniform one, so only a start at rest is reproduced exactly)
 * What differs — this is a level-synchronous BEAM over the same lattice, not the reference's A*:
 *   the reference's open list is a wall-clock-bounded best-first search whose expansion order is
 *   shuffled with a time seed (:321-322, :1462-1463) and whose closed-set update toggles with a call
 *   counter (:1190-1203); it is not reproducible even against itself.  Here depth d keeps the
 *   beam_width best children (ties by parent rank, then lattice index), a voxel holds one node per
 *   depth and is closed to later depths, the depth is num_pol, and without reaching the goal the best
 *   node of the last depth is returned.  The deterministic rule is stated in oracle/ and the kernel
 *   matches it bit for bit.
 */
#ifndef NEPTUNE_FRONTEND_H_
#define NEPTUNE_FRONTEND_H_

#include <stdint.h>

#include "neptune_backend.h"

#ifdef __cplusplus
extern "C" {
#endif

#define NEP_FE_MAX_BEAM 64
#define NEP_FE_MAX_SAMPLES 5
#define NEP_FE_ENT_CAP 40        /* (a fixed ABI constant: nep_fe_ent_state's layout depends on it; nep_abi_sizeof(14) lets a client verify)
                                     crossings of the FIXED entangle-state record (nep_fe_ent_state: the state at point A
                                     the caller passes in, and a search node's state on the fast path).  It is not a
                                     limit of the search: the reference prunes a node at num_agents + statics crossings
                                     (kinodynamic_search.cpp:850-854) and so does the front end — a node whose list, new
                                     crossings of one sampled step (more than 32) or bend points (more than NEP_MAX_BEND)
                                     outgrow the fixed record is carried, with all its descendants, in a "big record" of
                                     a per-handle pool in device memory that is sized by that very bound (round 4; before,
                                     such a node was pruned and flagged).  nep_fe_result.ent_overflow now only says that
                                     the POOL ran out (nep_batch_set_fe_ent_big_records)                              */

#define NEP_FE_GOAL_REACHED 1     /* status codes of KinodynamicSearch::run (:1637-1639)          */
#define NEP_FE_DEPTH_REACHED 0    /* (RUNTIME_REACHED there): best node of the last depth         */
#define NEP_FE_EMPTY 2            /* the beam died out: best node of the last non-empty depth     */
#define NEP_FE_NO_SOLUTION 3      /* no feasible first segment: K = 0, nothing to optimise        */

/* setMaxValuesAndSamples / setXYZMinMaxAndRa / setBias / setGoalSize / setTetherLength
 * (call sites neptune.cpp:92-97); bounds, bases and T_span come from the batch handle.          */
typedef struct nep_fe_cfg {
  double j_max;                   /* par_.j_max                                                    */
  double voxel_size;              /* par_.a_star_fraction_voxel_size                               */
  double bias;                    /* 1.1 (neptune.cpp:96)                                          */
  double goal_size;               /* par_.goal_radius                                              */
  double cable_length;            /* par_.tetherLength                                             */
  int32_t num_samples;            /* par_.a_star_samp_x, <= NEP_FE_MAX_SAMPLES                     */
  int32_t beam_width;             /* <= NEP_FE_MAX_BEAM (this build's search width)                */
  int32_t pad_hold;               /* 1: a guess shorter than num_pol (goal reached early, beam died out) is
                                     extended with segments that hold its end point, so that the back end
                                     also separates the place where the vehicle will WAIT from everybody's
                                     committed trajectory (the reference leaves that unchecked,
                                     kinodynamic_search.cpp:1805-1813)                               */
  int32_t enable_entangle;        /* par_.enable_entangle_check: nep_batch_frontend_ent only                       */
  int32_t ent_samples;            /* num_sample_per_interval (yaml: 3), 1..8                                       */
  int32_t _pad;
} nep_fe_cfg;

/* eu::ent_state of one search node / of point A (entangle_utils.hpp:23-29) in a fixed-size record: the crossing list
 * (agent or static id, case), its betas and the bend-point indices; active_cases[i] is the number of list entries of
 * agent i and is not stored.  beta of an AGENT crossing (id <= num_agents) is 0.0, as the reference's calculateBetaForCase
 * returns it (entangle_utils.cpp:1713-1719); only static crossings carry one.  States made by nep_ent_propagate_* honour that;
 * a state that does not is flagged by nep_batch_check (the search does not read such betas).                              */
typedef struct nep_fe_ent_state {
  int32_t n_alpha, n_bend;
  int16_t id[NEP_FE_ENT_CAP];
  int8_t cs[NEP_FE_ENT_CAP];
  double beta[NEP_FE_ENT_CAP];
  int8_t bend[NEP_MAX_BEND];
} nep_fe_ent_state;

/* Point A and the goal of one slot (setUp, kinodynamic_search.cpp:190-227).                      */
typedef struct nep_fe_start {
  double pos[3], vel[3], accel[3];   /* A; x,y enter the search, z the height profile               */
  double goal[3];                    /* G_term.pos                                                  */
  double t_start;                    /* neptune.cpp:1422-1423                                       */
} nep_fe_start;

typedef struct nep_fe_result {
  int32_t status;                 /* NEP_FE_*                                                      */
  int32_t K;                      /* segments of the guess (<= num_pol)                            */
  int32_t depth;                  /* depth at which the search stopped                             */
  int32_t n_children;             /* lattice children generated                                    */
  int32_t n_feasible;             /* ... that passed the kinodynamic tests                         */
  int32_t n_collision_free;       /* ... and the collision tests                                   */
  int32_t goal_occupied;          /* setUp's goal_occupied_ (:210-226)                             */
  int32_t _pad;                   /* entangle check on: bit 3 = the pool of big records ran out (ent_overflow); bits 8
                                     and up = children of this search whose state went into a big record; else 0    */
  double cost;                    /* g + bias*h of the returned node                               */
  double dist_to_goal;
  int32_t n_entangled;            /* children pruned by entanglesWithOtherAgents (entangle check on)        */
  int32_t ent_overflow;           /* 1: a child needed a big record and the handle's pool had none left (pruned), or
                                     more than 256 searches of one launch needed big records (the ones beyond keep
                                     the fixed record's limits: _pad bits 0-2 say which) — deviations from the
                                     reference's rule; 0 in every test and bench leg                                */
} nep_fe_result;

/* Front end of every slot of the batch handle, asynchronous on `stream`.
 *   d_committed : device, [n_scenes][N] nep_traj_rec (as nep_batch_replan)
 *   d_start     : device, [n_scenes][n_local] nep_fe_start
 *   d_guess     : device, [n_scenes][n_local] nep_guess     (out: what nep_batch_replan consumes)
 *   d_result    : device, [n_scenes][n_local] nep_fe_result (out, may be NULL)
 * The interval hulls are rebuilt into the handle's scratch exactly as nep_batch_replan does.      */
int nep_batch_frontend(nep_batch_t* h, const nep_fe_cfg* cfg, const nep_traj_rec* d_committed,
                       const nep_fe_start* d_start, nep_guess* d_guess, nep_fe_result* d_result,
                       void* stream);

/* The same against all-gathered hull blocks (multi-GPU rounds, include/neptune_backend.h:
 * nep_batch_hulls -> all-gather -> nep_batch_frontend_hulls -> nep_batch_replan_hulls).            */
int nep_batch_frontend_hulls(nep_batch_t* h, const nep_fe_cfg* cfg, const void* d_blocks, int32_t n_blocks,
                             const nep_fe_start* d_start, nep_guess* d_guess, nep_fe_result* d_result,
                             void* stream);

/* ---- enable_entangle_check on ---------------------------------------------------------------------------------
 * The search node carries the tether's entangle state (eu::ent_state) the way the reference's does:
 *   pruning     entanglesWithOtherAgents on every child (kinodynamic_search.cpp:1160-1164, :1345-1353, body :707-895):
 *               crossings of the child's sampled step with every other agent's tether polyline (its committed
 *               trajectory sampled ent_samples times per interval, Neptune::SamplePointsOfIntervals neptune.cpp:500-565,
 *               and its bend points from the record) and with the static representatives; cancellation against the
 *               accumulated list; a second active case for an agent, too many crossings or a tether longer than
 *               cable_length prune the child
 *   costs       g = sampled arc length, h = distance to the goal + 0.3 per crossing + 1.0 per bend point (:1177-1182)
 *   voxel       (ix, iy, getIz(state)) (:1170-1173, :2006-2031)
 *   collision   additionally the other agents' 0.7 m base squares (collidesWithBases2d, :1583-1628, :1675)
 *   end point   only a node whose active cases are all <= 1 may end the plan (:1693-1700)
 * and the states along the returned path give the dense case block nep_batch_replan's d_ent takes (the case of
 * (segment i, agent j) from the state at the START of segment i, solver_gurobi_poly.cpp:624-631) — guesses AND entangle
 * cases are then device-made.  Bit-identical to oracle/'s orc_frontend_beam_ent.
 *
 * nep_batch_set_static_reps   staticObsRep_ / staticObsLongestDist_ (setStaticObstRep, kinodynamic_search.cpp:385-390):
 *                             rep [n_static][2][2] (col(0), col(1)), longest [n_static][2]; scene = -1: every scene
 * nep_batch_frontend_ent      d_ent_init: [slots] state at point A or NULL (empty); d_case_out: [slots][NEP_MAX_POL][N]
 *                             (out, may be NULL).  Unsharded handle (n_local == num_agents), created with enable_entangle;
 *                             sharded handles: nep_batch_frontend_ent_hulls below.
 * nep_batch_safety_commit_ent nep_batch_safety_commit plus KinodynamicSearch::entangleCheckGivenPwp (:897-985) as
 *                             Neptune::safetyCheckAfterReplan calls it (neptune.cpp:746-754): every new trajectory is
 *                             re-checked from the state at its start against everybody's NEW trajectories and bend
 *                             points; as in the reference only its FIRST interval is examined (the loop returns at the
 *                             end of its first pass, :983), with three times the search's crossing capacity and no
 *                             tether-length test.  An entangling trajectory is turned down like a colliding one.   */
int nep_batch_set_static_reps(nep_batch_t* h, int32_t scene, const double* rep, const double* longest);
/* Big records of the entangle-aware front end (see NEP_FE_ENT_CAP above).
 * nep_batch_set_fe_ent_big_records   records in the handle's pool (0: the default, 4 per slot and at least 4 096); a record
 *                                    holds num_agents + statics crossings (13 bytes each) and the scratch of one sampled
 *                                    step; takes effect at the next front-end call (not while a stream is capturing).
 * nep_batch_set_fe_ent_fast_caps     (neptune_backend_debug.h) what the fixed record's path accepts before a child goes to a big record: list
 *                                    entries (<= NEP_FE_ENT_CAP), new crossings per sampled step (<= 32), bend points
 *                                    (<= NEP_MAX_BEND).  The results do not depend on these — the tests set them to
 *                                    0 / 1 / 2 to drive ordinary scenes through the big records.
 * The FIRST entangle-aware front-end call of a handle (and the first after a setting that changes a size) allocates the search's
 * device scratch — per-child states, the big-record pool and its betas, ≈ 0.3 GB at BASELINE configs[4] with 32 scenes — so it
 * must be made eagerly, before any stream capture: inside a capture the allocation fails with NEP_E_HIP.  Later calls with the
 * same sizes allocate nothing and can be captured (bench_legs/ does exactly that: warm-up steps first, then the graph).       */
int nep_batch_set_fe_ent_big_records(nep_batch_t* h, int64_t records);
int nep_batch_frontend_ent(nep_batch_t* h, const nep_fe_cfg* cfg, const nep_traj_rec* d_committed, const nep_fe_start* d_start,
                           const nep_fe_ent_state* d_ent_init, nep_guess* d_guess, nep_fe_result* d_result,
                           int32_t* d_case_out, void* stream);
/* nep_batch_frontend_ent against all-gathered hull blocks (multi-GPU rounds: the blocks of a handle created with
 * enable_entangle carry every agent's samples and presence flag next to its hulls and bend points, nep_batch_hulls);
 * cfg->ent_samples must equal the handle's nep_batch_set_ent_samples value (3 by default).  Bit-identical to
 * nep_batch_frontend_ent on an unsharded handle.                                                                      */
int nep_batch_frontend_ent_hulls(nep_batch_t* h, const nep_fe_cfg* cfg, const void* d_blocks, int32_t n_blocks, const nep_fe_start* d_start,
                                 const nep_fe_ent_state* d_ent_init, nep_guess* d_guess, nep_fe_result* d_result,
                                 int32_t* d_case_out, void* stream);
/* On a sharded handle nep_batch_safety_commit_ent takes, like the records, the entangle states of ALL agents:
 * d_ent_init [n_scenes][N] (gather them with nep_batch_exchange_slots); d_guess stays [n_scenes][n_local] (it supplies the
 * round's clock).  Every rank then computes the same accept vector.                                                   */
int nep_batch_safety_commit_ent(nep_batch_t* h, const nep_traj_rec* d_prev, const nep_traj_rec* d_new, const nep_guess* d_guess,
                                const nep_fe_ent_state* d_ent_init, int32_t ent_samples, double cable_length,
                                nep_traj_rec* d_final, int32_t* d_accept, void* stream);

/* Point A of the NEXT round for every slot, on the device: d_start[slot].t_start advances by dt and pos / vel / accel become
 * the state of the agent's committed trajectory (d_records [n_scenes][N], e.g. nep_batch_safety_commit's d_final) at that
 * time — Neptune::replanFull's choice of A "deltaT ahead on the committed plan" (neptune.cpp:1366-1399) for a
 * bulk-synchronous loop in which every agent replans from the same clock.  Evaluated like generatePwpOut's samples
 * (solver_gurobi_poly.cpp:921-929); beyond the trajectory's last knot the vehicle rests at its end point; an agent without a
 * valid record keeps its state.  The goal is left alone unless d_alt_goal ([slots][3], may be NULL) is given: then an agent
 * that has arrived (within switch_radius of its goal, slower than 0.05 m/s) swaps goal and alternate goal, so that a fleet
 * keeps flying back and forth (what bench.py's `moving` leg runs).  Asynchronous on `stream`; with it a whole round
 * (front end -> lines + QP -> safety check + commit -> next point A) is a fixed launch sequence on fixed buffers, i.e. one
 * HIP graph.                                                                                                          */
int nep_batch_next_starts(nep_batch_t* h, const nep_traj_rec* d_records, double dt, nep_fe_start* d_start,
                          double* d_alt_goal, double switch_radius, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* NEPTUNE_FRONTEND_H_ */
