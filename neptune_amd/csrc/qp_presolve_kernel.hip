// qp_presolve_kernel.hip — the zero-iteration half of the verified line presolve as a kernel of its own (round 6).
//
// Under the presolve (nep_batch_set_line_cull, the handle's default) most replans need no interior-point iteration: the minimiser of
// the cost over the equality-reduced variables, z* = -Hax^-1 g per axis, satisfies every box row, every near separating line and
// the terminal ball; with zero multipliers that point meets the KKT conditions of the full problem, so it IS what
// PolySolverGurobi::optimize returns (solver_gurobi_poly.cpp:823-882) — 91 % of the replans of the bench's 64-agent scenes, 87 % at
// config 5.  Until round 5 that test ran at the top of qp_reg_kernel<true>: a workgroup of 256 threads with 36.7 KB of LDS and 128
// registers per lane — an interior point's resources — held one of a CU's four slots for ~20 us to do a 24-variable
// matrix-vector product and a pass over ~500 rows (profiles/r06_qp_phases_default.txt: line gather 26 %, start point 16 %, outputs
// 18 % of the workgroups' lifetime, the iteration loop a third).  Here ONE WAVE per replan does exactly that and nothing else, eight
// waves per SIMD deep (<= 64 registers, 6 KB of LDS): every replan of a launch is resident at once, the kernel lasts as long as one
// wave's chain of global round trips.  A replan whose certificate holds is finished here — trajectory, statistics, sampled states and
// commit record written as qp_reg_kernel writes them — and marked in ps.presolved; the interior-point kernel that follows returns at
// once for marked slots, so its workgroups are the replans that do iterate (one in eleven), all resident from the start.
//
// The kernel only ever ACCEPTS or ABSTAINS: anything unusual — K < 3, an overflowed line bucket, a violated row, a parked line crossed
// or a control point moved beyond the radius the skipped LPs were proven for — leaves the slot unmarked and untouched, and
// qp_reg_kernel<true> handles it exactly as before (iterations, second attempt, redo list, polish list).  Same formulas as that
// kernel's own test (qp_reg_kernel.hip: "presolve: the minimiser without inequality rows"); the two may round differently in the
// last place, which decides nothing but who writes a result that both would accept.
#include <hip/hip_runtime.h>

#include "nep_device.h"
#include "nep_tables.h"

namespace nep {

namespace {
// MINVO position basis inverse on [0, 1] (the literals of nep_tables.h::kAPosInv): control points of the solution and of the guess
__constant__ double cPreAPosInv[4][4] = {
    {-0.03203276669713047, -0.09273093424558249, 0.3420572455666699, 1.1023313949144335},
    {-0.05111494245568798, -0.046272612998418894, 0.5458234872124772, 1.0979806946005568},
    {-0.07454781852812224, 0.203951949894552, 0.796048050105448, 1.0745478185281223},
    {1.0, 1.0, 0.9999999999999996, 0.9999999999999993}};

__device__ __forceinline__ double pre_wave_max(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_xor(v, o));
  return v;
}
__device__ __forceinline__ double pre_wave_sum(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}
}  // namespace

#ifndef NEP_PRE_WAVES
#define NEP_PRE_WAVES 8
#endif
__global__ __launch_bounds__(64, NEP_PRE_WAVES) void qp_presolve_kernel(SceneParams sp, ProblemSet ps, const QpTable* __restrict__ tables, SampleSched sched, int* __restrict__ presolved) {
  const int lane = threadIdx.x;
  const int slot = blockIdx.x;
  const long long t0 = (long long)wall_clock64();
  __shared__ double sCoef[96], sTheta[96], sInit[9], sFin[3], sZ[24], sG[24], sA[3 * kMaxR], sCp[64];
  __shared__ int sCnt[3 * NEP_MAX_POL + 4];
  const nep_guess* __restrict__ g = ps.guess + slot;
  nep_solution* __restrict__ sol = ps.solution + slot;
  const int K = g->K;
  const double T = sp.T_span, wgt = sp.weight;
  // every exit before the certificate leaves the slot to the interior-point kernel
  if (lane == 0) presolved[slot] = 0;
  if (K < 3 || K > NEP_MAX_POL || K > sp.num_pol) return;
  const int nz = K - 2, n = 3 * nz, R = 8 * K;
  const QpTable* __restrict__ tb = tables + K;      // mode 0: the first problem (terminal v = a = 0 eliminated)

  // ---- line counts of the eight buckets: near (solved for), parked (to verify), skipped LPs (verified by movement) ----
  int cn = 0, cf = 0, cs = 0; bool ovf = false;
  if (lane < NEP_MAX_POL && lane < K) {
    const int raw = ps.line_cnt[(long)slot * NEP_MAX_POL + lane];
    ovf = raw < 0; cn = line_count(raw);
    cf = ps.line_far[(long)slot * NEP_MAX_POL + lane];
    cs = ps.line_skip ? ps.line_skip[(long)slot * NEP_MAX_POL + lane] : 0;
  }
  if (lane < 96) sCoef[lane] = (&g->coeff[0][0][0])[lane];
  if (lane + 64 < 96) sCoef[lane + 64] = (&g->coeff[0][0][0])[lane + 64];
  if (__ballot(ovf) != 0ull) return;                    // a bucket overflowed: that replan fails (qp_reg_kernel: sI[27])
  if (lane < NEP_MAX_POL) { sCnt[lane] = cn; sCnt[NEP_MAX_POL + lane] = cf; sCnt[2 * NEP_MAX_POL + lane] = cs; }
  if (lane < 24) { sZ[lane] = 0.0; sG[lane] = 0.0; }
  __syncthreads();
  int L_near = 0, n_far = 0, n_skip = 0;
#pragma unroll
  for (int i = 0; i < NEP_MAX_POL; i++) { L_near += sCnt[i]; n_far += sCnt[NEP_MAX_POL + i]; n_skip += sCnt[2 * NEP_MAX_POL + i]; }
  if (lane < 9) sInit[lane] = sCoef[((lane / 3) * 8 + 0) * 4 + 1 + (lane % 3)];                       // b0, c0, d0 per axis (:390-396)
  if (lane < 3) { const double* c = sCoef + (lane * 8 + (K - 1)) * 4; sFin[lane] = ((T * T * T) * c[0] + (T * T) * c[1] + T * c[2]) + c[3]; }      // final_pos_ (:226-228)
  __syncthreads();
  const double dix = sCoef[3] - sFin[0], diy = sCoef[32 + 3] - sFin[1], diz = sCoef[64 + 3] - sFin[2];
  const bool has_qc = sqrt(dix * dix + diy * diy + diz * diz) < 1.0;      // the terminal ball row (:697-702)
  const bool z_override = sqrt(dix * dix + diy * diy) < 1.0;             // :879-880

  // ---- gradient of the cost at the origin of the reduced space, and the minimiser without inequality rows ----
  if (lane < n) {
    const int ax = lane / nz, c = lane - ax * nz;
    sG[lane] = (tb->Gi[c][0] * sInit[ax * 3] + tb->Gi[c][1] * sInit[ax * 3 + 1] + tb->Gi[c][2] * sInit[ax * 3 + 2]) - 2 * wgt * tb->ep[c] * sFin[ax];
  }
  __syncthreads();
  if (lane < n) {
    const int ax = lane / nz, c = lane - ax * nz;
    double v = 0;
    for (int e = 0; e < nz; e++) v -= tb->HaxInv[c][e] * sG[ax * nz + e];
    sZ[lane] = v;
  }
  __syncthreads();
  // ---- every base row at z*: a = U.init + B.z (positions of the 4 K control points, 3 K velocities, K accelerations per axis) ----
  double viol = -1.0;
#pragma nounroll
  for (int t = lane; t < 3 * R; t += 64) {
    const int ax = t / R, rho = t - ax * R;
    double a = tb->U[rho][0] * sInit[ax * 3] + tb->U[rho][1] * sInit[ax * 3 + 1] + tb->U[rho][2] * sInit[ax * 3 + 2];
    double v = 0;
#pragma nounroll
    for (int c = 0; c < nz; c++) v = __builtin_fma(tb->B[rho][c], sZ[ax * nz + c], v);
    a += v;
    sA[ax * kMaxR + rho] = a;
    const double hi = rho < 4 * K ? sp.maxs[ax] : (rho < 7 * K ? sp.v_max : sp.a_max);
    const double lo = rho < 4 * K ? sp.mins[ax] : (rho < 7 * K ? -sp.v_max : -sp.a_max);
    viol = fmax(viol, fmax(a - hi, lo - a));
  }
  __syncthreads();
  // ---- the near separating lines, read where the separator left them: n . q + d - 1 <= 0 at the segment's four control points ----
#pragma nounroll
  for (int e = lane; e < L_near; e += 64) {
    int i = 0, off = 0, acc = 0;      // i = the segment whose bucket holds line e (the number of inclusive prefix sums <= e), off = lines before it
#pragma unroll
    for (int j = 0; j < NEP_MAX_POL - 1; j++) { acc += sCnt[j]; if (e >= acc) { i = j + 1; off = acc; } }
    const double* nd = ps.line_nd + (((long)slot * NEP_MAX_POL + i) * sp.lines_cap + (e - off)) * 3;
    const double n1 = nd[0], n2 = nd[1], h = 1.0 - nd[2];
#pragma unroll
    for (int k = 0; k < 4; k++) viol = fmax(viol, (n1 * sA[4 * i + k] + n2 * sA[kMaxR + 4 * i + k]) - h);
  }
  if (lane == 0 && has_qc) {
    double c = -0.10 * 0.10;
    for (int ax = 0; ax < 3; ax++) {
      double pe = (tb->up[0] * sInit[ax * 3] + tb->up[1] * sInit[ax * 3 + 1] + tb->up[2] * sInit[ax * 3 + 2]) - sFin[ax];
      for (int e = 0; e < nz; e++) pe += tb->ep[e] * sZ[ax * nz + e];
      c += pe * pe;
    }
    viol = fmax(viol, c);
  }
  if (!(pre_wave_max(viol) <= 0.0)) return;             // some row is violated at z*: the interior point's job

  // ---- the trajectory: theta = Th z + ThU init (coefficients [a b c d] per segment and axis) ----
  for (int t = lane; t < 96; t += 64) sTheta[t] = 0.0;
  __syncthreads();
#pragma nounroll
  for (int t = lane; t < 12 * K; t += 64) {
    const int ax = t / (4 * K), r = t - ax * 4 * K;
    double v = tb->ThU[r][0] * sInit[ax * 3] + tb->ThU[r][1] * sInit[ax * 3 + 1] + tb->ThU[r][2] * sInit[ax * 3 + 2];
#pragma nounroll
    for (int c = 0; c < nz; c++) v += tb->Th[r][c] * sZ[ax * nz + c];
    sTheta[(ax * 8 + r / 4) * 4 + (r % 4)] = v;
  }
  __syncthreads();
  // ---- verification against what the presolve left out (qp_reg_kernel: "the far lines against the solution"): the returned
  // trajectory's position control points must stay within the radius of the guess's (the skipped LPs' lines are farther than that from
  // the guess) and on the right side of every parked line ----
  bool bad = false;
  if (lane < 8 * K) {
    const int rho = lane >> 1, ax = lane & 1, sg = rho >> 2, k = rho & 3;
    const double c0 = (T * T * T) * cPreAPosInv[0][k], c1 = (T * T) * cPreAPosInv[1][k], c2 = T * cPreAPosInv[2][k], c3 = cPreAPosInv[3][k];
    const double* Q = sTheta + (ax * 8 + sg) * 4;
    const double v = ((Q[0] * c0 + Q[1] * c1) + Q[2] * c2) + Q[3] * c3;
    sCp[rho * 2 + ax] = v;
    if (n_skip > 0) {
      const double* P = sCoef + (ax * 8 + sg) * 4;
      const double gq = ((P[0] * c0 + P[1] * c1) + P[2] * c2) + P[3] * c3;
      double d2 = (v - gq) * (v - gq);
      d2 += __shfl_xor(d2, 1);                          // (x and y of a control point sit on neighbouring lanes)
      bad = d2 > sp.cull_radius * sp.cull_radius;
    }
  }
  __syncthreads();
#pragma nounroll
  for (int e = lane; e < n_far; e += 64) {
    int i = 0, off = 0, acc = 0;
#pragma unroll
    for (int j = 0; j < NEP_MAX_POL - 1; j++) { acc += sCnt[NEP_MAX_POL + j]; if (e >= acc) { i = j + 1; off = acc; } }
    const double* nd = ps.line_nd + (((long)slot * NEP_MAX_POL + i) * sp.lines_cap + ((long)sp.lines_cap - 1 - (e - off))) * 3;      // parked lines sit at the back of the bucket
    const double n1 = nd[0], n2 = nd[1], dd = nd[2];
#pragma unroll
    for (int k = 0; k < 4; k++) bad = bad || (n1 * sCp[(4 * i + k) * 2] + n2 * sCp[(4 * i + k) * 2 + 1] + dd - 1.0 > 0.0);
  }
  if (__ballot(bad) != 0ull) return;                    // not verified: qp_reg_kernel solves it (and lists it for the redo pass if need be)

  // ---- the certificate holds: this is the optimum.  Outputs as qp_reg_kernel writes them. ----
  double o_share = 0.0;
  if (lane < n) {
    const int ax = lane / nz, c = lane - ax * nz;
    double hz = 0;
    for (int e = 0; e < nz; e++) hz += tb->Hax[c][e] * sZ[ax * nz + e];
    o_share = sZ[lane] * (0.5 * hz + sG[lane]);
  }
  o_share = pre_wave_sum(o_share);
  if (z_override) { if (lane < 32) sTheta[64 + lane] = sCoef[64 + lane]; }      // :879-880
  __syncthreads();
  for (int t = lane; t < 96; t += 64) (&sol->coeff[0][0][0])[t] = ((t % 32) / 4 < K) ? sTheta[t] : 0.0;
  if (lane <= NEP_MAX_POL) sol->times[lane] = (lane <= K) ? g->t_start + lane * T : 0.0;
  const int ns_all = sched.n[K];
  const int ns = ns_all < sp.max_states ? ns_all : sp.max_states;
  if (lane == 0) {
    double o = 0;      // the cost's constant term, in the reference's summation order (:322-383), then the quadratic's value at z*
    for (int ax = 0; ax < 3; ax++) {
      for (int r = 0; r < K; r++) { const double a = tb->Pp[r][0] * sInit[ax * 3] + tb->Pp[r][1] * sInit[ax * 3 + 1] + tb->Pp[r][2] * sInit[ax * 3 + 2]; o += 36 * T * a * a; }
      const double pe = (tb->up[0] * sInit[ax * 3] + tb->up[1] * sInit[ax * 3 + 1] + tb->up[2] * sInit[ax * 3 + 2]) - sFin[ax];
      o += wgt * pe * pe;
    }
    int n_lp = 0, n_lpf = 0;
    if (ps.lp_stats) for (int i = 0; i < NEP_MAX_POL; i++) { n_lp += ps.lp_stats[((long)slot * NEP_MAX_POL + i) * 2]; n_lpf += ps.lp_stats[((long)slot * NEP_MAX_POL + i) * 2 + 1]; }
    const int L_all = L_near + n_far + n_skip;
    sol->stats.status = NEP_OK; sol->stats.iters = 0; sol->stats.iters_first = 0;
    sol->stats.n_lines = L_all - n_lpf; sol->stats.n_lp = n_lp; sol->stats.n_lp_failed = n_lpf;
    sol->stats.n_rows = 48 * K + 4 * (L_near < L_all ? L_near : L_near - n_lpf); sol->stats.qc_active = has_qc ? 1 : 0;
    sol->stats.objective = o + o_share;
    sol->K = K; sol->n_states = ns;
  }
  if (ps.states) {      // generatePwpOut's samples (:911-934)
#pragma nounroll
    for (int s = lane; s < ns; s += 64) {
      const int i = sched.seg[K * sp.max_states + s]; const double dt = sched.dt[K * sp.max_states + s];
      double* st = ps.states + ((long)slot * sp.max_states + s) * NEP_STATE_DOUBLES;
#pragma nounroll
      for (int ax = 0; ax < 3; ax++) {
        const double* c = sTheta + (ax * 8 + i) * 4;
        st[ax] = ((c[0] * (dt * dt * dt) + c[1] * (dt * dt)) + c[2] * dt) + c[3];
        st[3 + ax] = (c[0] * (3 * dt * dt) + c[1] * (2 * dt)) + c[2];
        st[6 + ax] = c[0] * (6 * dt) + c[1] * 2;
        st[9 + ax] = c[0] * 6;
      }
    }
  }
  if (ps.commit) {      // the record the agent publishes (neptune_ros.cpp:434-480)
    nep_traj_rec* cr = ps.commit + slot;
    const int own = sp.first_local + (slot % sp.n_local);
    if (lane == 0) {
      cr->id = own + 1; cr->is_agent = 1; cr->n_bend = 1; cr->valid = 1;
      for (int a = 0; a < 3; a++) { cr->bbox[a] = 2 * sp.drone_radius; cr->pos[a] = sTheta[(a * 8) * 4 + 3]; }
      cr->bend[0][0] = ps.pb[2 * own]; cr->bend[0][1] = ps.pb[2 * own + 1];
      cr->pwp.n_seg = K;
    }
    if (lane <= NEP_TRAJ_MAX_SEG) cr->pwp.times[lane] = (lane <= K) ? g->t_start + lane * T : 0.0;
#pragma nounroll
    for (int e = lane; e < 3 * NEP_TRAJ_MAX_SEG * 4; e += 64) {
      const int ax = e / (NEP_TRAJ_MAX_SEG * 4), r = e % (NEP_TRAJ_MAX_SEG * 4), seg = r / 4, j = r % 4;
      (&cr->pwp.coeff[0][0][0])[e] = (seg < K) ? sTheta[(ax * 8 + seg) * 4 + j] : 0.0;
    }
  }
  if (lane == 0) {
    const double us_ = (double)((long long)wall_clock64() - t0) * sp.us_per_tick;
    sol->stats.solve_us = us_;
    if (ps.order_key) { const int ko = ps.order_key[slot] - sp.qp_key_decay; ps.order_key[slot] = (sp.qp_key_decay > 0 && ko > 0) ? ko : 0; }      // (a slot solved here costs the interior-point launch nothing: its key decays to the back of the order)
    presolved[slot] = 1;
  }
}

void launch_qp_presolve(int n_slots, const SceneParams& sp, const ProblemSet& ps, const QpTable* tables, const SampleSched& sched, int* presolved, hipStream_t st) {
  if (n_slots <= 0 || !presolved) return;
  hipLaunchKernelGGL(qp_presolve_kernel, dim3(n_slots), dim3(64), 0, st, sp, ps, tables, sched, presolved);
}

}  // namespace nep
