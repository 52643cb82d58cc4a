/* neptune_oracle.h — CPU restatement (plain C, fp64) of the NEPTUNE back-end path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under neptune_amd/ (the product) may include, link or call
 * this; only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg do, as the checker.
 *
 * PARITY STATUS: "parity unpinned" against a live reference run — the reference path needs
 * Gurobi (closed), GLPK 4.65 (downloaded at build time, submodules/separator/cmake/
 * glpk.cmake.in:6), CGAL 4.14.2 and Eigen, none of which exist in this image, and the reference
 * has no tests/golden vectors for this path.  What pins this oracle instead:
 * tests/golden/ (MINVO known answers; QP optima cross-checked by two independent SciPy solvers
 * on the full 12K-variable formulation; LP feasibility cross-checked with HiGHS).
 *
 * Every function cites the reference lines it follows.
 */
#ifndef NEPTUNE_ORACLE_H
#define NEPTUNE_ORACLE_H

#include "../include/neptune_backend.h"
#include "../include/neptune_frontend.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct orc_params {
  int num_pol, id, num_agents;
  double T_span, weight;
  double mins[3], maxs[3], v_max, a_max;
  const double* pb; /* [num_agents][2] */
} orc_params;

/* one obstacle polygon list in CSR */
typedef struct orc_polys {
  int n;
  const int* off;   /* [n+1] */
  const double* xy; /* [off[n]][2] */
} orc_polys;

typedef struct orc_ent {
  int enabled;
  const int* case_id; /* [K][num_agents]: 0 = nothing to add for (segment, agent) */
  const int* bend_off; /* [num_agents+1] */
  const double* bend_xy;
  const orc_polys* hulls_noinfl; /* num_agents*num_pol polygons, id-indexed */
} orc_ent;

typedef struct orc_result {
  int status; /* NEP_OK / NEP_RELAXED / NEP_FAILED */
  int iters, iters_first;
  int n_lines, n_lp, n_lp_failed, n_rows, qc_active;
  double objective;
  double coeff[3][NEP_MAX_POL][4];
  /* lines actually used, row order (segment-major, reference loop order) */
  int line_seg[8192];
  double line_nd[8192][3];
} orc_result;

/* MINVO control points.  solver_gurobi_poly.cpp:232-243 (position), :456-461 (velocity). */
void orc_pos_ctrl_pts(const double P[4], double T, double Q[4]);
void orc_vel_ctrl_pts(const double P[4], double T, double Qv[3]);

/* cu::convexHullOfPoints2d (cgal_utils.cpp:157-174): CCW extreme points, starting at the
 * lexicographically smallest point; returns vertex count. */
int orc_convex_hull_2d(int n, const double (*pts)[2], double (*out)[2]);

/* Neptune::vertexesOfInterval2d + convexHullOfInterval2d (neptune.cpp:288-309, 349-452) for one
 * committed trajectory and one interval.  hull / hull0 receive the inflated / uninflated hulls. */
int orc_hull_of_interval(const nep_pwp* pwp, double t0, double t1, double T_span,
                          const double delta[2], double (*hull)[2], int* nv, double (*hull0)[2],
                          int* nv0);

/* Study knob (scripts/separator_sensitivity.py), not part of the restated path: which admissible vertex of the
 * separator LP is returned by the calling thread from now on.  0 (default) largest gap = the product's rule; 1 seeded
 * pseudo-random; 2 smallest gap; 3 least room for the reference control points ref_ctrl[seg][4][2] of the segment being
 * built; 4 the vertex of a two-phase Bland simplex. */
void orc_set_vertex_policy(int policy, unsigned long long seed, const double* ref_ctrl);
void orc_vertex_policy_stats(long* n_lps, long* n_vertices);

/* Neptune::setStaticObst inflation (neptune.cpp:639-664). */
int orc_inflate_static(int nv, const double (*v)[2], double safe_dist, double (*out)[2]);

/* separator::Separator::solveModel 2-D (separator_glpk.cpp:248-373).  Deterministic rule: the
 * LP vertex (two tight rows of one set + one of the other) of maximum geometric gap; returns 1
 * if separable.  nd = (n1, n2, d) with n.a+d >= 1 on A and n.b+d <= -1 on B. */
int orc_separator(int nA, const double (*A)[2], int nB, const double (*B)[2], double nd[3]);
/* Same rule when A is a convex polygon given in boundary order (hulls, inflated statics): only
 * its edges are candidate pairs. */
int orc_separator_ordered(int nA, const double (*A)[2], int nB, const double (*B)[2], double nd[3]);

/* Same LP solved by a textbook two-phase primal simplex with Bland's rule (the algorithm class
 * GLPK's glp_simplex implements; GLPK's own pivoting rules are not reproducible without its
 * source).  Used only to cross-check feasibility in tests. */
int orc_separator_simplex(int nA, const double (*A)[2], int nB, const double (*B)[2],
                          double nd[3]);
/* the vertex a primal simplex of GLPK's default class reaches (see neptune_oracle.c); n_pivots may be NULL */
int orc_separator_glpk_class(int nA, const double (*A)[2], int nB, const double (*B)[2], double nd[3], int* n_pivots);
/* which separator rule the restated path uses from now on (thread-local): 0 largest gap, 1 GLPK-class simplex */
void orc_set_separator_rule(int rule);

/* the interior point's strict tests (thread-local; defaults 1e-10 / 1e-11 as the device's; checker of nep_batch_set_tolerances) */
void orc_set_qp_tolerances(double residual_tol, double gap_tol);
/* the active-set polish of interior-point solves that end without passing the strict tests (qp_solve; on by default) */
void orc_set_polish(int on);
int orc_last_polished(void);
void orc_pass_stats(long* iters, long* trig);      /* test hook: interior-point iterations / discarded predictors since the last call */

/* PolySolverGurobi::optimize (solver_gurobi_poly.cpp:804-887) for one agent.
 *   K, coeff_init: setInitTrajectory (:187-244)
 *   hulls: setHulls, polygons j*num_pol+i (:246-281)
 *   statics: setStaticObstVert (:316-320)
 *   ent: setEntStateVector + setHullsNoInflation (:283-314), may be NULL
 *   override_n >= 0: use the given lines instead of the separator (test hook). */
int orc_optimize(const orc_params* par, int K, const double coeff_init[3][NEP_MAX_POL][4],
                 int n_obst, const orc_polys* hulls, const orc_polys* statics,
                 const orc_ent* ent, int override_n, const int* override_seg,
                 const double (*override_nd)[3], orc_result* out);

/* PolySolverGurobi::generatePwpOut sampling loop (:911-934); returns number of states. */
int orc_sample(int K, const double coeff[3][NEP_MAX_POL][4], double T_span, double dc,
               double* states, int cap);

/* Whole replan of one agent from committed-trajectory records: hulls (neptune.cpp:224-285) for
 * every other valid agent, then orc_optimize.  case_id as in orc_ent (or NULL). */
int orc_replan(const orc_params* par, double drone_radius, int n_rec, const nep_traj_rec* recs,
               const nep_guess* guess, const orc_polys* statics, const int* case_id,
               orc_result* out, double* hull_xy_out, int* hull_nv_out);

/* SURVEY §8(f) rank 1 (post-solve safety check): gjk::collision (gjk.cpp:76-149),
 * Neptune::trajsAndPwpAreInCollision2d (neptune.cpp:767-806) and the bulk-synchronous resolution. */
int orc_gjk_collision(int n1, const double (*V1)[2], int n2, const double (*V2)[2]);
int orc_trajs_and_pwp_in_collision(const nep_traj_rec* other, const nep_pwp* mine, double T_span, double drone_radius);
void orc_safety_resolve(int n, const nep_traj_rec* fresh, double t_start, double T_span, double drone_radius, unsigned char* conflict, int* accept);
void orc_safety_resolve_prev(int n, const nep_traj_rec* prev, const nep_traj_rec* fresh, double t_start, double T_span, double drone_radius,
                             unsigned char* conflict, int* accept);

/* the deterministic beam rule of include/neptune_frontend.h (front-end initial
 * guess).  hull_xy [num_agents][num_pol][NEP_HULL_MAX_V][2], hull_nv [num_agents][num_pol]: interval
 * hulls of every agent's committed trajectory (own entry ignored), as orc_replan returns them. */
typedef struct orc_fe_cfg {
  int num_pol, id, num_agents, num_samples, beam_width, pad_hold;
  double T_span, j_max, v_max, a_max, voxel_size, bias, goal_size, cable_length;
  double mins[2], maxs[2];
  const double* pb;
} orc_fe_cfg;
/* KinodynamicSearch::run restated (best-first, unbounded depth) with the lattice order and the pop budget as
 * parameters; order == NULL: jx-major.  res->n_children = pops, res->n_feasible = nodes created, res->depth = index
 * of the returned node (may exceed num_pol: the guess keeps the first num_pol segments). */
int orc_frontend_astar(const orc_fe_cfg* c, const nep_fe_start* st, const double* hull_xy, const int* hull_nv, const orc_polys* statics,
                       const int* order, int max_pops, nep_guess* guess, nep_fe_result* res);
int orc_frontend_beam(const orc_fe_cfg* c, const nep_fe_start* st, const double* hull_xy, const int* hull_nv,
                      const orc_polys* statics, nep_guess* guess, nep_fe_result* res);
/* enable_entangle_check: what the search holds besides the hulls (setUp, kinodynamic_search.cpp:190-257; setStaticObstRep
 * :385-390): the other agents' sampled committed trajectories, their tether bend points, the static representatives,
 * and the entangle state at point A (NULL = empty). */
typedef struct orc_fe_ent {
  int num_samples;                 /* num_sample_per_interval */
  int n_static;
  const double* static_rep;        /* [n_static][2][2] */
  const double* static_longest;    /* [n_static][2] */
  const double* sampled;           /* [num_agents][num_pol][num_samples+1][2] */
  const int* present;              /* [num_agents] */
  const int* bend_n;               /* [num_agents] */
  const double* bend_xy;           /* [num_agents][NEP_MAX_BEND][2] */
  const nep_fe_ent_state* init;    /* or NULL */
} orc_fe_ent;
int orc_frontend_beam_ent(const orc_fe_cfg* c, const nep_fe_start* st, const double* hull_xy, const int* hull_nv,
                          const orc_polys* statics, const orc_fe_ent* E, nep_guess* guess, nep_fe_result* res, int* case_out);
/* test hook: case ids per (knot, agent) along a given guess, as the restatement above accumulates them */
int orc_ent_propagate_guess(const orc_fe_cfg* c, const orc_fe_ent* E, const nep_guess* g, int* case_out, int* hit_at, int* n_alpha_final);
/* entangleCheckGivenPwp for the first interval [a b c d] of a new trajectory (1 = entangles) */
int orc_entangle_check_pwp(const orc_fe_cfg* c, const orc_fe_ent* E, const double cx0[4], const double cy0[4]);

#ifdef __cplusplus
}
#endif
#endif
