/* neptune_entangle.h — tether entanglement-state propagation along a trajectory (SURVEY §8 f, rank 4).
 *
 * Host-only C ABI (no HIP call), exported by libneptune_backend.so.  It produces the REAL inputs of
 * the back end's entangle rows — PolySolverGurobi::setEntStateVector's eu::ent_state per knot
 * (solver_gurobi_poly.cpp:307-314, read at :620-637) — from a guess trajectory, the other agents'
 * committed trajectories and the tether bend points, the way the front end accumulates them node by
 * node:
 *
 *   nep_ent_sample_points      Neptune::SamplePointsOfIntervals          neptune/src/neptune.cpp:500-565
 *   nep_ent_propagate_segment  KinodynamicSearch::entanglesWithOtherAgents  kinodynamic_search.cpp:707-895
 *                              with eu::entangleHSigToAddAgentInd        entangle_utils.cpp:1129-1228
 *                                   eu::entangleHSigToAddStatic          :1231-1277
 *                                   eu::addAlphaBetaToList / breakcondition   :1402-1534, :1608-1647
 *                                   eu::updateBendPts                    :1536-1604
 *                                   eu::getBendPt2d / calculateBetaForCase / getTetherLength  :1649-1743
 *   nep_ent_propagate_guess    the chain of node states KinodynamicSearch::recoverEntStateVector
 *                              returns for a path (kinodynamic_search.cpp:582-603), for a given guess
 *   nep_ent_case_ids           the (knot, agent) -> case id reduction of solver_gurobi_poly.cpp:624-631
 *                              in the dense layout nep_batch_replan consumes
 */
#ifndef NEPTUNE_ENTANGLE_H_
#define NEPTUNE_ENTANGLE_H_

#include <stdint.h>

#include "neptune_backend.h"

#ifdef __cplusplus
extern "C" {
#endif

/* What KinodynamicSearch holds for the check: constructor arguments (kinodynamic_search.cpp:95-128),
 * setTetherLength (:259), setStaticObstRep (:385-390).                                            */
typedef struct nep_ent_cfg {
  int32_t num_agents;             /* pb.size()                                                      */
  int32_t id;                     /* 1-based id of the planning agent                               */
  int32_t num_pol;                /* planning intervals                                             */
  int32_t num_samples;            /* num_sample_per_interval (yaml: 3)                              */
  double T_span;
  double cable_length;            /* tether length (cablelength_)                                   */
  int32_t n_static;               /* staticObsRep_.size()                                           */
  int32_t _pad;
  const double* pb;               /* [num_agents][2] bases                                          */
  const double* static_rep;       /* [n_static][2][2]: col(0) = (x,y), col(1) = (x,y)               */
  const double* static_longest;   /* [n_static][2] staticObsLongestDist_                            */
} nep_ent_cfg;

/* Per-replan inputs: SampledPtsForAll_ and bendPtsForAgents_ (setUp, kinodynamic_search.cpp:190-257). */
typedef struct nep_ent_inputs {
  const double* sampled;          /* [num_agents][num_pol][num_samples+1][2]                        */
  const int32_t* present;         /* [num_agents]; 0 = SampledPtsForAll_[i].empty()                 */
  const int32_t* bend_off;        /* [num_agents+1] CSR offsets into bend_xy                        */
  const double* bend_xy;          /* [bend_off[num_agents]][2]                                      */
} nep_ent_inputs;

/* eu::ent_state (entangle_utils.hpp:23-29) in caller-owned arrays of capacity `cap` entries.       */
typedef struct nep_ent_state {
  int32_t n_alpha;                /* alphas.size() == betas.size()                                  */
  int32_t n_bend;                 /* bendPointsIdx.size()                                           */
  int32_t cap;
  int32_t n_active;               /* active_cases.size() (>= num_agents + n_static)                 */
  int32_t* alphas;                /* [cap][2] (agent or static id, case)                            */
  double* betas;                  /* [cap]                                                          */
  int32_t* bend_idx;              /* [cap]                                                          */
  int32_t* active_cases;          /* [n_active]                                                     */
} nep_ent_state;

/* Positions of one committed trajectory sampled on the planning grid; out: [num_pol][num_samples+1][2]. */
int nep_ent_sample_points(const nep_pwp* traj, double t_start, double t_end, int32_t num_pol,
                          int32_t num_samples, double* out);

/* One node expansion's entangle update for the segment with coefficients coeff_x/coeff_y ([a b c d],
 * local time) ending at end_xy, `index` = 1-based segment number.  Updates *state in place and adds
 * the sampled arc length to *arc_length.  Returns 1 when the reference's function returns true
 * (too many crossings, a second active case for an agent, tether too short), 0 otherwise,
 * NEP_E_ARG / NEP_E_CAP on bad input / capacity.                                                  */
int nep_ent_propagate_segment(const nep_ent_cfg* cfg, const nep_ent_inputs* in, nep_ent_state* state,
                              const double coeff_x[4], const double coeff_y[4], const double end_xy[2],
                              int32_t index, double* arc_length);

/* States at every knot of a K-segment guess, starting from *init (knot 0), in the flattened form
 * nep_backend_set_ent_state_vector takes (nep_ent_view): alpha_off [K+2], alphas [alpha_cap][2],
 * active_cases [K+1][n_active].  *entangled_at = first 1-based segment whose update returned 1
 * (the states from there on repeat the last good one), 0 if none.  *final receives the state at
 * the last propagated knot (may be NULL).                                                          */
int nep_ent_propagate_guess(const nep_ent_cfg* cfg, const nep_ent_inputs* in, const nep_ent_state* init,
                            const nep_guess* guess, int32_t alpha_cap, int32_t* alpha_off,
                            int32_t* alphas, int32_t* active_cases, int32_t* entangled_at,
                            nep_ent_state* final_state);

/* case_id[i][j] for knots i < NEP_MAX_POL, agents j < num_agents (0 = no single active case).     */
int nep_ent_case_ids(int32_t n_states, int32_t n_active, const int32_t* alpha_off, const int32_t* alphas,
                     const int32_t* active_cases, int32_t num_agents, int32_t* case_id);

#ifdef __cplusplus
}
#endif
#endif /* NEPTUNE_ENTANGLE_H_ */
