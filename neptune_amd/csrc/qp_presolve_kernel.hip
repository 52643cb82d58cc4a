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
// wave's chain of global round trips.  A replan whose certificate holds is finished here — trajectory and statistics written as qp_reg_kernel writes
// them — and marked in ps.presolved; the slot's workgroup of the interior-point launch that follows only samples the states and writes
// the commit record from the returned coefficients (pure stores, overlapped with the iterating replans' arithmetic) and returns, so that
// launch's interior-point work is the replans that do iterate (one in eleven), all resident from the start.
//
// The kernel only ever ACCEPTS or ABSTAINS: anything unusual — K < 3, an overflowed line bucket, a violated row, or a control
// point moved beyond the radius that verifies the parked lines and the skipped LPs — leaves the slot unmarked and untouched, and
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
#define NEP_PRE_WAVES 6      // (8: 64 registers with 19 spilled, 28.5 us per 8 192 replans and a step of 0.504 ms; 6: 80 registers, none spilled, 0.486 ms — same box)
#endif
// The minimiser without inequality rows is linear in v = (b0, c0, d0, f) of an axis (nep_tables.h: RowMap, ThMap, ObjQ, built on the
// host from the same tables qp_reg_kernel uses: z* = -HaxInv (Gi init - 2 w ep f), row = U init + B z*, theta = ThU init + Th z*):
// a lane loads ONE 32-byte row of each map as soon as K is known — no chain g -> z* -> rows — and the guess's twelve numbers come
// through the scalar unit (wave-uniform addresses).
__global__ __launch_bounds__(64, NEP_PRE_WAVES) void qp_presolve_kernel(SceneParams sp, ProblemSet ps, const QpTable* __restrict__ tables, SampleSched sched, int* __restrict__ presolved) {
  __builtin_amdgcn_s_setprio(3);      // (latency-bound waves: when another scene group's hull / separator waves share the SIMD — bench.py's pipelined groups — these issue first)
  const int lane = threadIdx.x;
  const int slot = blockIdx.x;
  const long long t0 = (long long)wall_clock64();
  __shared__ double sCoef[96], sTheta[96], sA[2 * 32];
  __shared__ int sCnt[3 * NEP_MAX_POL + 4];
  const nep_guess* __restrict__ g = ps.guess + slot;
  nep_solution* __restrict__ sol = ps.solution + slot;
  const int K = g->K;
  const double T = sp.T_span;
  // every exit before the certificate leaves the slot to the interior-point kernel
  if (lane == 0) presolved[slot] = 0;
  if (K < 3 || K > NEP_MAX_POL || K > sp.num_pol) return;
  const int R = 8 * K;
  const QpTable* __restrict__ tb = tables + K;      // mode 0: the first problem (terminal v = a = 0 eliminated)

  // ---- everything this wave reads that does not depend on anything but K and the slot, issued together ----
  int cn = 0, cf = 0, cs = 0; bool ovf = false;
  if (lane < NEP_MAX_POL && lane < K) {
    const int raw = ps.line_cnt[(long)slot * NEP_MAX_POL + lane];
    ovf = raw < 0; cn = line_count(raw);
    cf = ps.line_far[(long)slot * NEP_MAX_POL + lane];
    cs = ps.line_skip ? ps.line_skip[(long)slot * NEP_MAX_POL + lane] : 0;
  }
  const double c_lo = (&g->coeff[0][0][0])[lane], c_hi = lane + 64 < 96 ? (&g->coeff[0][0][0])[lane + 64] : 0.0;
  double rm0 = 0, rm1 = 0, rm2 = 0, rm3 = 0, tm0 = 0, tm1 = 0, tm2 = 0, tm3 = 0;
  if (lane < R) { const double* q = tb->RowMap[lane]; rm0 = q[0]; rm1 = q[1]; rm2 = q[2]; rm3 = q[3]; }
  if (lane < 4 * K) { const double* q = tb->ThMap[lane]; tm0 = q[0]; tm1 = q[1]; tm2 = q[2]; tm3 = q[3]; }
  int lpv = 0;
  if (ps.lp_stats && lane < 2 * NEP_MAX_POL) lpv = ps.lp_stats[(long)slot * NEP_MAX_POL * 2 + lane];
  // v = (b0, c0, d0, f) per axis: wave-uniform addresses (scalar loads)
  double vv[3][4];
#pragma unroll
  for (int ax = 0; ax < 3; ax++) {
    const double* c0 = g->coeff[ax][0]; const double* cK = g->coeff[ax][K - 1];
    vv[ax][0] = c0[1]; vv[ax][1] = c0[2]; vv[ax][2] = c0[3];                                    // b0, c0, d0 (:390-396)
    vv[ax][3] = ((T * T * T) * cK[0] + (T * T) * cK[1] + T * cK[2]) + cK[3];                    // final_pos_ (:226-228)
  }
  if (__ballot(ovf) != 0ull) return;                    // a bucket overflowed: that replan fails (qp_reg_kernel: sI[27])
  sCoef[lane] = c_lo; if (lane + 64 < 96) sCoef[lane + 64] = c_hi;
  if (lane < NEP_MAX_POL) { sCnt[lane] = cn; sCnt[NEP_MAX_POL + lane] = cf; sCnt[2 * NEP_MAX_POL + lane] = cs; }
  const double dix = vv[0][2] - vv[0][3], diy = vv[1][2] - vv[1][3], diz = vv[2][2] - vv[2][3];
  const bool has_qc = sqrt(dix * dix + diy * diy + diz * diz) < 1.0;      // the terminal ball row (:697-702)
  const bool z_override = sqrt(dix * dix + diy * diy) < 1.0;             // :879-880

  // ---- every base row at z* (positions of the 4 K control points, 3 K velocities, K accelerations: one row per lane, three axes) and
  // the trajectory's coefficients [a b c d] per segment and axis ----
  double viol = -1.0;
  if (lane < R) {
#pragma unroll
    for (int ax = 0; ax < 3; ax++) {
      const double a = ((rm0 * vv[ax][0] + rm1 * vv[ax][1]) + rm2 * vv[ax][2]) + rm3 * vv[ax][3];
      const double hi = lane < 4 * K ? sp.maxs[ax] : (lane < 7 * K ? sp.v_max : sp.a_max);
      const double lo = lane < 4 * K ? sp.mins[ax] : (lane < 7 * K ? -sp.v_max : -sp.a_max);
      viol = fmax(viol, fmax(a - hi, lo - a));
      if (ax < 2 && lane < 4 * K) sA[ax * 32 + lane] = a;                                      // x, y of the position control points: the line rows read them
    }
  }
  sTheta[lane] = 0.0; if (lane + 64 < 96) sTheta[lane + 64] = 0.0;
  __syncthreads();
  if (lane < 4 * K) {
#pragma unroll
    for (int ax = 0; ax < 3; ax++) sTheta[(ax * 8 + lane / 4) * 4 + (lane % 4)] = ((tm0 * vv[ax][0] + tm1 * vv[ax][1]) + tm2 * vv[ax][2]) + tm3 * vv[ax][3];
  }
  __syncthreads();
  int L_near = 0, n_far = 0, n_skip = 0;
#pragma unroll
  for (int i = 0; i < NEP_MAX_POL; i++) { L_near += sCnt[i]; n_far += sCnt[NEP_MAX_POL + i]; n_skip += sCnt[2 * NEP_MAX_POL + i]; }
  // ---- the near separating lines, read where the separator left them: n . q + d - 1 <= 0 at the segment's four control points ----
#pragma unroll 2
  for (int e = lane; e < L_near; e += 64) {
    int i = 0, off = 0, acc = 0;      // i = the segment whose bucket holds line e (the number of inclusive prefix sums <= e), off = lines before it
#pragma unroll
    for (int j = 0; j < NEP_MAX_POL - 1; j++) { acc += sCnt[j]; if (e >= acc) { i = j + 1; off = acc; } }
    const double* nd = ps.line_nd + (((long)slot * NEP_MAX_POL + i) * sp.lines_cap + (e - off)) * 3;
    const double n1 = nd[0], n2 = nd[1], h = 1.0 - nd[2];
#pragma unroll
    for (int k = 0; k < 4; k++) viol = fmax(viol, (n1 * sA[4 * i + k] + n2 * sA[32 + 4 * i + k]) - h);
  }
  if (has_qc) {      // the terminal ball (:697-702): |p(end) - f|^2 <= 0.1^2, p(end) from the returned coefficients
    double c = -0.10 * 0.10;
#pragma unroll
    for (int ax = 0; ax < 3; ax++) { const double* q = sTheta + (ax * 8 + (K - 1)) * 4; const double pe = (((T * T * T) * q[0] + (T * T) * q[1] + T * q[2]) + q[3]) - vv[ax][3]; c += pe * pe; }
    viol = fmax(viol, c);
  }
  // ---- verification against what the presolve left out.  A parked line lies farther than the radius r from each of the guess's four
  // control points of its segment (separator: -worst > r |n|), a skipped LP's line at least as far (its point sets' boxes are r apart and
  // the box sides are polygon edges): a solution control point within r of the guess's is on the right side of every one of them —
  // n . Q + d - 1 <= (n . B + d - 1) + |n| |Q - B| < 0 — so the movement bound verifies BOTH, and no parked line is read here (qp_reg_kernel
  // reads each of them: 96 MB per launch of 8 192 config-4 replans).  One part in 1e9 of slack for the roundings of the two tests; a
  // replan that moved farther is the interior-point kernel's (which lists it for the redo pass if need be). ----
  bool bad = false;
  if (lane < 8 * K) {
    const int rho = lane >> 1, ax = lane & 1, sg = rho >> 2, k = rho & 3;
    const double c0 = (T * T * T) * cPreAPosInv[0][k], c1 = (T * T) * cPreAPosInv[1][k], c2 = T * cPreAPosInv[2][k], c3 = cPreAPosInv[3][k];
    const double* Q = sTheta + (ax * 8 + sg) * 4;
    const double v = ((Q[0] * c0 + Q[1] * c1) + Q[2] * c2) + Q[3] * c3;
    const double* P = sCoef + (ax * 8 + sg) * 4;
    const double gq = ((P[0] * c0 + P[1] * c1) + P[2] * c2) + P[3] * c3;
    double d2 = (v - gq) * (v - gq);
    d2 += __shfl_xor(d2, 1);                            // (x and y of a control point sit on neighbouring lanes)
    bad = (n_far > 0 || n_skip > 0) && !(d2 <= sp.cull_radius * sp.cull_radius * (1.0 - 1e-9));
  }
  if (!(pre_wave_max(viol) <= 0.0)) return;             // some row is violated at z*: the interior point's job
  if (__ballot(bad) != 0ull) return;                    // moved too far to be sure: qp_reg_kernel solves it and checks every parked line

  // ---- the certificate holds: this is the optimum.  Trajectory and statistics as qp_reg_kernel writes them. ----
  int n_lp = (lane & 1) == 0 ? lpv : 0, n_lpf = (lane & 1) == 1 ? lpv : 0;
#pragma unroll
  for (int o_ = 8; o_ > 0; o_ >>= 1) { n_lp += __shfl_xor(n_lp, o_); n_lpf += __shfl_xor(n_lpf, o_); }      // (lanes 0..15 hold the values: the sums land in lane 0)
  if (z_override) { if (lane < 32) sTheta[64 + lane] = sCoef[64 + lane]; }      // :879-880
  __syncthreads();
  for (int t = lane; t < 96; t += 64) (&sol->coeff[0][0][0])[t] = ((t % 32) / 4 < K) ? sTheta[t] : 0.0;
  if (lane <= NEP_MAX_POL) sol->times[lane] = (lane <= K) ? g->t_start + lane * T : 0.0;
  const int ns_all = sched.n[K];
  const int ns = ns_all < sp.max_states ? ns_all : sp.max_states;
  if (lane == 0) {
    double o = 0;      // the cost at z*: v' ObjQ v per axis (the first problem's cost, :322-383)
    for (int ax = 0; ax < 3; ax++) for (int a = 0; a < 4; a++) { double r_ = 0; for (int b = 0; b < 4; b++) r_ += tb->ObjQ[a][b] * vv[ax][b]; o += vv[ax][a] * r_; }
    const int L_all = L_near + n_far + n_skip;
    sol->stats.status = NEP_OK; sol->stats.iters = 0; sol->stats.iters_first = 0;
    sol->stats.n_lines = L_all - n_lpf; sol->stats.n_lp = n_lp; sol->stats.n_lp_failed = n_lpf;
    sol->stats.n_rows = 48 * K + 4 * (L_near < L_all ? L_near : L_near - n_lpf); sol->stats.qc_active = has_qc ? 1 : 0;
    sol->stats.objective = o;
    sol->K = K; sol->n_states = ns;
  }
  // (the sampled states and the commit record — 9.7 KB per replan, 80 MB per launch of 8 192: a third of this kernel's 70 us when they
  // were written here — are written from the returned coefficients by the slot's workgroup of the interior-point launch that follows,
  // qp_reg_kernel's first lines: there they overlap the iterating replans' arithmetic instead of standing alone)
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
