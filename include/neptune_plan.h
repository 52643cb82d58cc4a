/* neptune_plan.h — ROS-free host side of the replan loop around the back end (SURVEY §8 f, rank 3).
 *
 * Plain C ABI, host memory only (no HIP call is made by anything declared here), exported by the
 * same libneptune_backend.so as include/neptune_backend.h.  It lets a Neptune-like loop
 * (select point A -> guess -> back end -> splice plan -> compose committed trajectory -> publish)
 * run on the shim without ROS:
 *
 *   nep_pwp_compose        mu::composePieceWisePol               neptune/src/utils.cpp:318-402
 *   nep_dyntraj_*          mader_msgs/DynTraj on the ROS1 wire   mader_msgs/msg/DynTraj.msg:1-9,
 *                          (pwp2PwpMsg / pwpMsg2Pwp,             PieceWisePolTraj.msg:1-4, CoeffPoly3.msg:1-4,
 *                           publishOwnTraj / trajCB)             utils.cpp:180-261, neptune_ros.cpp:379-480
 *   nep_plan_*             mt::committedTrajectory plan_ and the point-A selection / splice /
 *                          goal pop of Neptune::replanFull       mader_types.hpp:674-738,
 *                                                                neptune.cpp:860-891,1366-1425,1661-1719
 */
#ifndef NEPTUNE_PLAN_H_
#define NEPTUNE_PLAN_H_

#include <stddef.h>
#include <stdint.h>

#include "neptune_backend.h"

#ifdef __cplusplus
extern "C" {
#endif

/* ---------------------------------------------------------------------------------------------
 * mu::composePieceWisePol(t, dc, p1, p2)   (utils.cpp:318-402)
 *
 * Joins the part of the previously committed trajectory p1 that lies after t with the new
 * trajectory p2.  Like the reference (which takes p1, p2 by non-const reference) the function
 * ADJUSTS p1->times[0] / p2->times[0] in place (:321-335).  Results:
 *   - |t - p2.times[0]| < 1e-5                 -> out = p2                       (:337-340)
 *   - gap / t outside both                     -> out->n_seg = 0 ("dummy")       (:342-355)
 *   - otherwise out.times = {t, p1 knots in (t, p2.times[0]), p2 knots > t}, with the interval
 *     ending at p2.times[0] taken from p1's last interval                        (:357-401)
 * `dc` is accepted and unused, as in the reference.
 * Returns NEP_OK, NEP_E_ARG (null / malformed n_seg), NEP_E_CAP (> NEP_TRAJ_MAX_SEG intervals).  */
int nep_pwp_compose(double t, double dc, nep_pwp* p1, nep_pwp* p2, nep_pwp* out);

/* Extension (no reference counterpart): the composition that reproduces the flown path.  The reference
 * routine above gives the stretch that ends at p2.times[0] the coefficients of p1's LAST interval
 * (utils.cpp:388-393) and leaves the first, partial interval on its source interval's local time, so
 * between a commit and the next point A a published trajectory can sit metres away from the vehicle.
 * Here every interval of out carries the coefficients of the source interval that covers it, re-based
 * (Taylor shift) to its own first knot; past p1's last knot its end point is held.  Knots: t, p1's knots
 * inside (t, p2.times[0]), p2.times[0], p2's knots.  With t >= p2.times[0] the result is p2 from t on.
 * Returns NEP_OK, NEP_E_ARG, NEP_E_CAP.                                                          */
int nep_pwp_compose_exact(double t, const nep_pwp* p1, const nep_pwp* p2, nep_pwp* out);

/* ---------------------------------------------------------------------------------------------
 * mader_msgs/DynTraj in ROS1 serialisation (little endian; every array = uint32 count + items;
 * string = uint32 length + bytes; bool = 1 byte).  Field order:
 *   Header{uint32 seq; uint32 stamp.sec; uint32 stamp.nsec; string frame_id}
 *   string[] function; float32[] bbox; float64 pos[3]; int32 id; uint8 is_agent;
 *   Vector3[] bendpt (float64 x,y,z);  pwp{float64[] times; CoeffPoly3[] coeff_x, coeff_y, coeff_z}
 * The record's bend[][2] carries (x, y) of bendpt; z is written as 0.0 (neptune_ros.cpp:457-476).
 * `function` is written as three empty strings (neptune_ros.cpp:436-441) and skipped on decode.
 * bbox goes through float32 on the wire exactly as in the reference: decode(encode(r)).bbox ==
 * (double)(float)r.bbox.                                                                        */
typedef struct nep_wire_header {
  uint32_t seq;
  uint32_t stamp_sec;
  uint32_t stamp_nsec;
  uint32_t _pad;
  const char* frame_id;          /* may be NULL (= "")                                            */
} nep_wire_header;

/* number of bytes nep_dyntraj_encode writes for this record/header; < 0 on error                 */
int64_t nep_dyntraj_wire_size(const nep_traj_rec* rec, const nep_wire_header* hdr);
/* serialises rec into buf[cap]; returns bytes written, NEP_E_ARG, or NEP_E_CAP if cap is short   */
int64_t nep_dyntraj_encode(const nep_traj_rec* rec, const nep_wire_header* hdr, uint8_t* buf,
                           size_t cap);
/* parses one message.  Returns bytes consumed, NEP_E_ARG on a truncated / malformed message
 * (incl. coeff_x/y/z of different lengths, on which the reference aborts, utils.cpp:231-236),
 * NEP_E_CAP if the message holds more intervals / bend points than the record can.  rec->valid is
 * set to 1.  seq/stamp are returned through hdr_out (frame_id is not returned), may be NULL.    */
int64_t nep_dyntraj_decode(const uint8_t* buf, size_t len, nep_traj_rec* rec,
                           nep_wire_header* hdr_out);

/* ---------------------------------------------------------------------------------------------
 * mt::committedTrajectory plan_ (a deque of mt::state) and the three places replanFull /
 * getNextGoal touch it.  A state is 12 doubles: pos[3], vel[3], accel[3], jerk[3] — the layout of
 * nep_batch_replan's d_states / nep_backend_generate_pwp_out's traj_out.                        */
typedef struct nep_plan nep_plan_t;

typedef struct nep_plan_cfg {
  double dc;                          /* par_.dc                                                  */
  double T_span;                      /* par_.T_span                                              */
  double lower_bound_runtime;         /* par_.lower_bound_runtime_snlopt                          */
  double upper_bound_runtime;         /* par_.upper_bound_runtime_snlopt                          */
  double runtime_opt;                 /* par_.runtime_opt                                         */
  double factor_alpha;                /* par_.factor_alpha                                        */
  int32_t deltaT0;                    /* initial deltaT_ (neptune.hpp:133: 75)                    */
  int32_t _pad;
} nep_plan_cfg;

typedef struct nep_point_a {
  double A[12];                       /* selected state, vel/accel zeroed when future_index < 0   */
  int32_t k_index;                    /* index of A in the plan                                   */
  int32_t k_index_end;                /* states after A                                           */
  double runtime_search;              /* front-end budget, saturated (neptune.cpp:1406-1419)      */
  double t_start;                     /* k_index*dc + time_now (neptune.cpp:1422-1423)            */
} nep_point_a;

nep_plan_t* nep_plan_create(const nep_plan_cfg* cfg);
void nep_plan_destroy(nep_plan_t* p);
/* plan_.clear(); plan_.push_back(state) — how Neptune seeds the plan with the current state     */
int nep_plan_reset(nep_plan_t* p, const double state[12]);
int32_t nep_plan_size(const nep_plan_t* p);
int nep_plan_get(const nep_plan_t* p, int32_t i, double state_out[12]);
/* Neptune::getNextGoal (neptune.cpp:860-891): returns front(); pops it when size > 1.
 * *last_point = 1 when the plan held a single state.                                            */
int nep_plan_next_goal(nep_plan_t* p, double goal_out[12], int32_t* last_point);
/* point-A selection (neptune.cpp:1366-1399) + runtime_search / t_start (:1406-1423).
 * state_pos = the agent's measured position (A.pos is replaced by it when the head of the plan
 * is > 1 m away, :1395-1398).                                                                   */
int nep_plan_select_a(nep_plan_t* p, const double state_pos[3], double time_now, nep_point_a* out);
/* plan splice (neptune.cpp:1661-1687): erases A and everything after it, appends traj_out.
 * Returns NEP_OK, or NEP_E_STATE ("Already published the point A": plan_size-1-k_index_end < 0). */
int nep_plan_splice(nep_plan_t* p, int32_t k_index_end, const double* traj_out, int32_t n_states);
/* deltaT_ update after a replan that took elapsed_ms (neptune.cpp:1713-1720)                    */
int nep_plan_update_delta(nep_plan_t* p, double elapsed_ms);
int32_t nep_plan_delta(const nep_plan_t* p);

#ifdef __cplusplus
}
#endif
#endif /* NEPTUNE_PLAN_H_ */
