// qp_common.h — constants, LDS carve and wave-level primitives shared by the two placements of the interior-point
// kernel (qp_kernels.hip: line-row state in LDS, any line count; qp_reg_kernel.hip: line-row state in registers).
#pragma once
#include <hip/hip_runtime.h>

#include <type_traits>

#include "nep_device.h"

namespace nep {

constexpr int BS = 256;
constexpr int SBS = 9;            // LDS row stride of the base-row table B (8 used; odd -> lanes on consecutive rows hit distinct banks)
constexpr int MS = 25;            // LDS row stride of the normal matrix (n <= 24; odd -> no bank conflicts)
constexpr int kMaxIt = 60;
// Mehrotra's second-order term dsa * dla extrapolates the affine step to its full length.  From iteration kCorrFromIt on (the
// usual solve has ended by then), whenever less than kCorrMinStep of that step is admissible, the predictor is discarded: the
// corrector is formed as if the affine direction were zero (dsa = -rp, dla = -lambda + w rp: what is left of the term vanishes
// with the primal residual), sigma as computed.  With the full term marginally feasible problems cycle (gap down 10x, back up
// over three short steps, for ever) and infeasible ones blow up to 1e18 and idle to the iteration cap; without it the former
// converge and the latter stall within a few iterations (oracle: qp_solve).
// (round 5 tried from the sixth iteration / five times at most — scripts/giveup_rule_sweep.py: on the closed loop's hard replans the 38
// decisively infeasible ones end after 37.7 passes instead of 49.3 and none of the 39 strictly feasible ones is lost, `moving` 1.91 ->
// 2.01 M replans/s — and took it back: in a fleet flown to its goals (scripts/closed_loop_failures.py) the QP then gives up on 1.54 % of
// the replans instead of 0.81 %; 10 / 5: 1.22 %, 8 / 5: 1.10 %.  Every earlier give-up loses solves the reference's solver would return.
// The two are run-time values (SceneParams::corr_from_it / corr_max_count; NEP_CORR_FROM / NEP_CORR_MAX for such A/Bs).)
constexpr int kCorrFromIt = 10;      // (the default of SceneParams::corr_from_it: nep_device.h's kCorrFromItDefault says the same)
constexpr double kCorrMinStep = 0.1;
constexpr int kCorrMaxCount = 8;            // a solve that needs this more often is given up (converging ones: at most five times in 16 000)
// start point of the rows: slack = max(h - a.x0, kSlackFloor), lambda = kMu0 / slack.  Chosen on this path's two kinds of
// guesses (oracle sweep): 1 / 1 needs 11.3 iterations on front-end guesses and 5.8 on near-optimal ones, 0.1 / 2
// needs 8.6 and 6.1.
constexpr double kSlackFloor = 0.1, kMu0 = 2.0;
// fraction of the step to the boundary: 1 - mu clamped to [0.999, 0.99999].  The late iterations shrink the residuals by
// (1 - fraction) each, so a fraction closer to one saves one of them (oracle sweep: 6.1 -> 5.2 iterations on the bench
// guesses); early on, while mu is large, hugging the boundary costs centrality: a fixed 0.99999 left a few front-end-guess
// replans per thousand iterating to the cap.
#ifndef NEP_STEPFRAC_MAX
#define NEP_STEPFRAC_MAX 0.99999
#endif
constexpr double kStepFracMin = 0.999, kStepFracMax = NEP_STEPFRAC_MAX;
// offsets (doubles) inside QpTable's tail Gi, ep, ev, ea, up, uv, ua, Nt, Pp, res_u, Zp, HaxInv
constexpr int tGi = 0, tUp = 48, tUv = 51, tUa = 54, tNt = 57, tPp = 121, tResU = 145, tZp = 151, tHi = 151 + kNZ * 4 * kMaxK, kSmallTab = tHi + kNZ * kNZ;
static_assert(offsetof(QpTable, up) - offsetof(QpTable, Gi) == tUp * 8 && offsetof(QpTable, Nt) - offsetof(QpTable, Gi) == tNt * 8 &&
              offsetof(QpTable, Pp) - offsetof(QpTable, Gi) == tPp * 8 && offsetof(QpTable, res_u) - offsetof(QpTable, Gi) == tResU * 8 && offsetof(QpTable, Zp) - offsetof(QpTable, Gi) == tZp * 8 && offsetof(QpTable, HaxInv) - offsetof(QpTable, Gi) == tHi * 8 &&
              sizeof(QpTable) - offsetof(QpTable, Gi) == kSmallTab * 8, "QpTable tail layout");

// ---- LDS carve (in doubles) -------------------------------------------------------------------
constexpr int oB = 0;                       // [64][8]
constexpr int oOff = oB + kMaxR * SBS;      // [64][3]
constexpr int oAccL = oOff + kMaxR * 3;     // [32][8] line accumulators per control point
constexpr int oDc = oAccL + 32 * 8;         // [64][4] combined weights Dxx,Dxy,Dyy,Dzz
constexpr int oTc = oDc + kMaxR * 4;        // [64][6] combined T_lambda[3], T1[3]
constexpr int oM = oTc + kMaxR * 6;         // [24][25]
constexpr int oHax = oM + 24 * MS;          // [8][8]
constexpr int oZ = oHax + 64;               // [24]
constexpr int oG = oZ + 24;                 // [24]
constexpr int oRd = oG + 24;                // [24]
constexpr int oRhs = oRd + 24;              // [24]
constexpr int oDxa = oRhs + 24;             // [24]
constexpr int oDx = oDxa + 24;              // [24]
constexpr int oGq = oDx + 24;               // [24]
constexpr int oZl = oGq + 24;               // [24] loose snapshot
constexpr int oInvD = oZl + 24;             // [24]
constexpr int oEp = oInvD + 24;             // [8]
constexpr int oCoef = oEp + 8;              // [3][8][4] initial guess
constexpr int oTheta = oCoef + 96;          // [3][8][4] result
constexpr int oInit = oTheta + 96;          // [3][3] b0,c0,d0 per axis
constexpr int oScal = oInit + 9;            // scalars, see enum
constexpr int oRed = oScal + 32;            // [3][16] reduction scratch (one slot per reduction of an iteration)
constexpr int oU = oRed + 48;               // [24] D^-1 g of the terminal ball row (Sherman-Morrison, qp_reg_kernel) + [8] its dot products
constexpr int oFixedEnd = oU + 32;
constexpr int kFixedDoubles = (oFixedEnd + 1) & ~1;

enum { sFinal0 = 0, sFinal1, sFinal2, sMu, sSigKeep, sAlpha, sObj0, sObj, sSq, sLq, sRpq, sWq, sDsqA, sDlqA, sDsq, sDlq, sQscale, sNrp, sSumSl, sObjLoose, sSigMu, sPe0, sPe1, sPe2, sBestMerit, sInvMt, sObjOut, sMtD };      // (32 slots)

// Line stride of the carve that lets two workgroups share a CU (backend.hip::size_scratch's l_half, rounded the same way):
// the normal case, instantiated with the stride as a compile-time constant so that the row passes' LDS accesses take
// immediate offsets instead of one address add each.
constexpr int kLdsLinesHalf = (int)(((((160 * 1024 / 2) - (kFixedDoubles * 8 + 64 * 4)) / (11 * 8) - 2) + 1) & ~1);
constexpr int kLLHalf = kLdsLinesHalf + 2;

// Cross-lane primitives on the VALU (DPP) and scalar (v_readlane) paths: HIP's __shfl* go through
// ds_bpermute (an LDS round trip per 32-bit half), which dominated the first version's reductions
// and the register Cholesky.
template <int CTRL>
__device__ __forceinline__ double dpp(double v) {
  // (mov_dpp, not update_dpp(old = src): the latter first copies the source into the destination — four instructions per double
  // instead of two; every lane of these permutations has a valid source, so there is no "old" value to keep)
  const int lo = __builtin_amdgcn_mov_dpp(__double2loint(v), CTRL, 0xf, 0xf, false);
  const int hi = __builtin_amdgcn_mov_dpp(__double2hiint(v), CTRL, 0xf, 0xf, false);
  return __hiloint2double(hi, lo);
}
constexpr int DPP_XOR1 = 0xB1;         // quad_perm [1,0,3,2]
constexpr int DPP_XOR2 = 0x4E;         // quad_perm [2,3,0,1]
constexpr int DPP_HALF_MIRROR = 0x141; // lane i <-> 7-i inside each 8
constexpr int DPP_ROW_MIRROR = 0x140;  // lane i <-> 15-i inside each 16
__device__ __forceinline__ double bcast(double v, int lane) {   // lane: wave-uniform
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), lane), hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
  return __hiloint2double(hi, lo);
}
// sum over each aligned group of 8 lanes (result in all 8)
__device__ __forceinline__ double slice_sum(double v) { v += dpp<DPP_XOR1>(v); v += dpp<DPP_XOR2>(v); v += dpp<DPP_HALF_MIRROR>(v); return v; }
__device__ __forceinline__ double wave_sum(double v) {
  v = slice_sum(v); v += dpp<DPP_ROW_MIRROR>(v);
  return (bcast(v, 0) + bcast(v, 16)) + (bcast(v, 32) + bcast(v, 48));
}
__device__ __forceinline__ double wave_max(double v) {
  v = fmax(v, dpp<DPP_XOR1>(v)); v = fmax(v, dpp<DPP_XOR2>(v)); v = fmax(v, dpp<DPP_HALF_MIRROR>(v)); v = fmax(v, dpp<DPP_ROW_MIRROR>(v));
  return fmax(fmax(bcast(v, 0), bcast(v, 16)), fmax(bcast(v, 32), bcast(v, 48)));
}


// MINVO position basis inverse on [0, 1] (mader_types.hpp:152-157 inverted; the literals of nep_tables.h::kAPosInv): the guess's
// control points, for the presolve's movement test
__constant__ double cQpAPosInv[4][4] = {
    {-0.03203276669713047, -0.09273093424558249, 0.3420572455666699, 1.1023313949144335},
    {-0.05111494245568798, -0.046272612998418894, 0.5458234872124772, 1.0979806946005568},
    {-0.07454781852812224, 0.203951949894552, 0.796048050105448, 1.0745478185281223},
    {1.0, 1.0, 0.9999999999999996, 0.9999999999999993}};

// x / d for 0 <= x < 1024 and a wave-uniform 1 <= d <= 64 as one multiply and a shift (exact in that range; the magic numbers
// come from constant memory through the scalar unit).  A real integer division expands to a float reciprocal sequence on
// the VALU which the compiler hoists out of the iteration loops and then has to keep (spill) across them.
__constant__ unsigned int cDivMagic[65] = {0, 65537, 32769, 21846, 16385, 13108, 10923, 9363, 8193, 7282, 6554, 5958, 5462, 5042, 4682, 4370, 4097, 3856, 3641, 3450, 3277, 3121, 2979, 2850, 2731, 2622, 2521, 2428, 2341, 2260, 2185, 2115, 2049, 1986, 1928, 1873, 1821, 1772, 1725, 1681, 1639, 1599, 1561, 1525, 1490, 1457, 1425, 1395, 1366, 1338, 1311, 1286, 1261, 1237, 1214, 1192, 1171, 1150, 1130, 1111, 1093, 1075, 1058, 1041, 1025};
__device__ __forceinline__ int div_small(int x, int d) { return (int)(((unsigned)x * cDivMagic[d]) >> 16); }

// 1/a: v_rcp_f64 seed + Newton steps (the IEEE divide expands to ~3x the work).  The row passes
// only use it inside the Newton direction (weights lam/s, ratio tests): one step suffices there,
// the residuals that decide convergence never go through it.
__device__ __forceinline__ double frcp(double a) {
  double r = __builtin_amdgcn_rcp(a);
  double e = __builtin_fma(-a, r, 1.0); r = __builtin_fma(r, e, r);
#ifdef NEP_FRCP_TWO_STEPS
  e = __builtin_fma(-a, r, 1.0); r = __builtin_fma(r, e, r);
#endif
  return r;
}

// 1/a to the last bit or two (v_rcp_f64 + two Newton steps: 8 issue slots) for the iteration's SCALARS — the affine step length,
// mu_aff / mu, the step length — which every thread of the workgroup computes for itself after a reduction: an IEEE division
// is a ~32-instruction sequence, and four of them on all 256 threads were 8 % of the kernel's VALU instructions.
__device__ __forceinline__ double frcp2(double a) {
  double r = __builtin_amdgcn_rcp(a);
  double e = __builtin_fma(-a, r, 1.0); r = __builtin_fma(r, e, r);
  e = __builtin_fma(-a, r, 1.0); r = __builtin_fma(r, e, r);
  return r;
}

// 1/sqrt(a): v_rsq_f64 seed + two Newton steps (the IEEE sqrt + divide pair costs ~4x more and sits
// on the Cholesky's critical path once per column).
__device__ __forceinline__ double frsqrt(double a) {
  double y = __builtin_amdgcn_rsq(a);
  double h = 0.5 * a;
  double e = __builtin_fma(-h * y, y, 0.5); y = __builtin_fma(y, e, y);
  e = __builtin_fma(-h * y, y, 0.5); y = __builtin_fma(y, e, y);
  return y;
}

// Workgroup reductions of one max and NS (0..2) sums in two halves, so that the barrier between
// them can be shared with other hand-offs: reduce_put before the barrier, reduce_get after it.
template <int NS>
__device__ __forceinline__ void reduce_put(double mx, double s0, double s1, double* red) {
  mx = wave_max(mx);
  if (NS > 0) s0 = wave_sum(s0);
  if (NS > 1) s1 = wave_sum(s1);
  if ((threadIdx.x & 63) == 0) { double* o = red + 4 * (threadIdx.x >> 6); o[0] = mx; if (NS > 0) o[1] = s0; if (NS > 1) o[2] = s1; }
}
template <int NS>
__device__ __forceinline__ void reduce_get(double& mx, double& s0, double& s1, const double* red) {
  mx = fmax(fmax(red[0], red[4]), fmax(red[8], red[12]));
  if (NS > 0) s0 = (red[1] + red[5]) + (red[9] + red[13]);
  if (NS > 1) s1 = (red[2] + red[6]) + (red[10] + red[14]);
}

// One max and up to three sums across the workgroup in one round trip.  red: LDS [16].
__device__ __forceinline__ void block_reduce4(double& mx, double& s0, double& s1, double& s2, double* red) {
  mx = wave_max(mx); s0 = wave_sum(s0); s1 = wave_sum(s1); s2 = wave_sum(s2);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) { double* o = red + 4 * (threadIdx.x >> 6); o[0] = mx; o[1] = s0; o[2] = s1; o[3] = s2; }
  __syncthreads();
  mx = fmax(fmax(red[0], red[4]), fmax(red[8], red[12]));
  s0 = (red[1] + red[5]) + (red[9] + red[13]);
  s1 = (red[2] + red[6]) + (red[10] + red[14]);
  s2 = (red[3] + red[7]) + (red[11] + red[15]);
}


// ---- wave-level dense SPD solve (wave 0 of the workgroup) ---------------------------------------
// Cholesky with one matrix row per lane: lane i keeps L[i][0..i] in registers, the pivot column is
// broadcast lane->wave with v_readlane; no workgroup barrier.  L goes back to LDS (lower triangle of
// sM, stride MS) together with 1/L[i][i]; returns false on a non-positive pivot.  Kept out of line
// so that its register rows do not inflate the row passes' allocation.
typedef __attribute__((address_space(3))) double* lds_dptr;
// Branch-free on purpose: per-lane predicates (lane >= k ...) would become exec-mask juggling per
// update; entries a lane does not own are simply allowed to hold garbage — they are never
// broadcast, stored into the used triangle, or selected.
template <int N>
__device__ __forceinline__ bool chol_impl(lds_dptr sM, lds_dptr sInvD, int lane) {
  const int row = lane < N ? lane : N - 1;
  double Lr[N];
  bool ok = true;
#pragma unroll
  for (int j = 0; j < N; j++) Lr[j] = sM[row * MS + j];
  double dinv = 0.0;
#pragma unroll
  for (int j = 0; j < N; j++) {
    const double d = bcast(Lr[j], j);
    if (!(d > 0.0)) ok = false;
    const double inv = frsqrt(d);
    Lr[j] *= inv;
    dinv = lane == j ? inv : dinv;
#pragma unroll
    for (int k = j + 1; k < N; k++) Lr[k] = __builtin_fma(-Lr[j], bcast(Lr[j], k), Lr[k]);
  }
  if (lane < N) {
#pragma unroll
    for (int j = 0; j < N; j++) sM[lane * MS + j] = Lr[j];   // (entries right of the diagonal are scratch)
    sInvD[lane] = dinv;
  }
  return ok;
}
template <int N>
__device__ __forceinline__ double solve_impl(lds_dptr sM, lds_dptr sInvD, int lane, double b) {
  // (select-free substitution: see solve_pad)
  const int row = lane < N ? lane : N - 1;
  const double dinv = sInvD[row];
  {
    double Lr[N];                   // row i of L, zero from the diagonal on
#pragma unroll
    for (int j = 0; j < N; j++) { const double v = sM[row * MS + j]; Lr[j] = j < lane ? v : 0.0; }
#pragma unroll
    for (int j = 0; j < N; j++) b = __builtin_fma(-Lr[j], bcast(b * dinv, j), b);
    b *= dinv;
  }
  __builtin_amdgcn_sched_barrier(0);   // the column loads below must not be hoisted over the forward pass (register footprint of the callee)
  {
    double Uc[N];                   // column i of L, zero down to the diagonal
#pragma unroll
    for (int j = 0; j < N; j++) { const double v = sM[j * MS + row]; Uc[j] = j > lane ? v : 0.0; }
#pragma unroll
    for (int j = N - 1; j >= 0; j--) b = __builtin_fma(-Uc[j], bcast(b * dinv, j), b);
    b *= dinv;
  }
  return b;
}

// The same two routines for an n x n block inside a fixed size N >= n (the block is extended by the identity on the fly,
// nothing is written to LDS for it), inlined: qp_reg_kernel keeps its row state in registers across the factorisation and
// cannot afford an out-of-line call's save/restore.  N is the smallest of a few sizes that holds n (12 and 6 are exact for
// the usual K = 8 replan).
template <int N>
__device__ __forceinline__ bool chol_pad(lds_dptr sM, lds_dptr sInvD, int n, int lane) {
  const int row = lane < N ? lane : N - 1;
  const bool rin = row < n;
  double Lr[N];
  bool ok = true;
#pragma unroll
  for (int j = 0; j < N; j++) { const double v = sM[row * MS + j]; Lr[j] = (rin && j < n) ? v : (row == j ? 1.0 : 0.0); }
  double dinv = 0.0;
#pragma unroll
  for (int j = 0; j < N; j++) {
    const double d = bcast(Lr[j], j);
    if (!(d > 0.0)) ok = false;
    const double inv = frsqrt(d);
    Lr[j] *= inv;
    dinv = lane == j ? inv : dinv;
#pragma unroll
    for (int k = j + 1; k < N; k++) Lr[k] = __builtin_fma(-Lr[j], bcast(Lr[j], k), Lr[k]);
  }
  if (lane < n) {
#pragma unroll
    for (int j = 0; j < N; j++) if (j < n) sM[lane * MS + j] = Lr[j];
    sInvD[lane] = dinv;
  }
  return ok;
}
template <int N>
__device__ __forceinline__ double solve_pad(lds_dptr sM, lds_dptr sInvD, int n, int lane, double b) {
  // Entries a lane must not use are loaded as ZERO (row i of L right of and on the diagonal, column i above and on it), so a
  // substitution step is one multiply, one broadcast and one fused multiply-add for every lane, with no per-step selects:
  // lane j's running value stops changing at step j (its later coefficients are zero) and x_j = b_j * dinv_j is taken once
  // after the loop.  Same operations on the entries that matter as the select form: same bits.
  const int row = lane < N ? lane : N - 1;
  const bool rin = row < n;
  const double dinv = rin ? sInvD[row] : 1.0;
  b = lane < n ? b : 0.0;
  {
    double Lr[N];
#pragma unroll
    for (int j = 0; j < N; j++) { const double v = sM[row * MS + j]; Lr[j] = (rin && j < n && j < lane) ? v : 0.0; }
#pragma unroll
    for (int j = 0; j < N; j++) b = __builtin_fma(-Lr[j], bcast(b * dinv, j), b);
    b *= dinv;
  }
  __builtin_amdgcn_sched_barrier(0);
  {
    double Uc[N];
#pragma unroll
    for (int j = 0; j < N; j++) { const double v = sM[j * MS + row]; Uc[j] = (rin && j < n && j > lane) ? v : 0.0; }
#pragma unroll
    for (int j = N - 1; j >= 0; j--) b = __builtin_fma(-Uc[j], bcast(b * dinv, j), b);
    b *= dinv;
  }
  return b;
}
// The same two routines for any n <= 24 with the matrix left in LDS (lane i = row i, the pivot column is read back by all
// lanes: LDS operations of one wave execute in order, so a lane's write is visible to the reads that follow it).  A dozen
// registers instead of 2 N + ...: qp_reg_kernel uses them for the sizes that are rare (terminal ball active: 3 nz; relaxed
// re-solve: nz = K), so that the register allocator never has to place a 24-entry row next to the row state.
__device__ __forceinline__ bool chol_lds(lds_dptr sM, lds_dptr sInvD, int n, int lane) {
  bool ok = true;
  const bool rin = lane < n;
  for (int j = 0; j < n; j++) {
    const double d = sM[j * MS + j];
    if (!(d > 0.0)) ok = false;
    const double inv = frsqrt(d);
    double lij = 0.0;
    if (rin && lane >= j) { lij = sM[lane * MS + j] * inv; sM[lane * MS + j] = lij; }
    if (lane == j) sInvD[j] = inv;
    for (int k = j + 1; k < n; k++) {
      const double lkj = sM[k * MS + j];
      if (rin && lane >= k) sM[lane * MS + k] = __builtin_fma(-lij, lkj, sM[lane * MS + k]);
    }
  }
  return ok;
}
__device__ __forceinline__ double solve_lds(lds_dptr sM, lds_dptr sInvD, lds_dptr tmp, int n, int lane, double b) {
  const bool rin = lane < n;
  b = rin ? b : 0.0;
  for (int j = 0; j < n; j++) {               // L y = b
    if (lane == j) tmp[j] = b * sInvD[j];
    const double xj = tmp[j];
    if (rin && lane > j) b = __builtin_fma(-sM[lane * MS + j], xj, b);
    if (lane == j) b = xj;
  }
  for (int j = n - 1; j >= 0; j--) {          // L' x = y
    if (lane == j) tmp[j] = b * sInvD[j];
    const double xj = tmp[j];
    if (lane < j) b = __builtin_fma(-sM[j * MS + lane], xj, b);
    if (lane == j) b = xj;
  }
  return b;
}
// wave 0's block (2 nz or 3 nz) / wave 1's block (nz): n is wave-uniform.  tmp: 24 doubles of LDS free during the solve.
__device__ __forceinline__ bool chol_n(lds_dptr sM, lds_dptr sInvD, int n, int lane) {
  n = __builtin_amdgcn_readfirstlane(n);
  if (n <= 6) return chol_pad<6>(sM, sInvD, n, lane);
  if (n <= 8) return chol_pad<8>(sM, sInvD, n, lane);
  if (n <= 12) return chol_pad<12>(sM, sInvD, n, lane);
  return chol_lds(sM, sInvD, n, lane);
}
__device__ __forceinline__ double solve_n(lds_dptr sM, lds_dptr sInvD, lds_dptr tmp, int n, int lane, double b) {
  n = __builtin_amdgcn_readfirstlane(n);
  if (n <= 6) return solve_pad<6>(sM, sInvD, n, lane, b);
  if (n <= 8) return solve_pad<8>(sM, sInvD, n, lane, b);
  if (n <= 12) return solve_pad<12>(sM, sInvD, n, lane, b);
  return solve_lds(sM, sInvD, tmp, n, lane, b);
}

// One straight-line instantiation per size (no per-step size tests).  Sizes: 3 nz (ball constraint
// couples the axes), or the two diagonal blocks 2 nz (x,y) and nz (z) factored by two waves at once.
#define NEP_SIZE_SWITCH(CALL)                                                                      \
  switch (n) {                                                                                     \
    case 1: return CALL(1); case 2: return CALL(2); case 3: return CALL(3); case 4: return CALL(4);  \
    case 5: return CALL(5); case 6: return CALL(6); case 7: return CALL(7); case 8: return CALL(8);  \
    case 9: return CALL(9); case 10: return CALL(10); case 12: return CALL(12); case 14: return CALL(14); \
    case 15: return CALL(15); case 16: return CALL(16); case 18: return CALL(18); case 21: return CALL(21); \
    default: return CALL(24);                                                                      \
  }
static __device__ __noinline__ bool chol_wave(unsigned m_off, unsigned d_off, int n) {   // LDS byte offsets of the block / its 1/diag
  // arguments of a real call arrive in VGPRs: make the wave-uniform ones scalar again
  m_off = __builtin_amdgcn_readfirstlane(m_off); d_off = __builtin_amdgcn_readfirstlane(d_off); n = __builtin_amdgcn_readfirstlane(n);
  const lds_dptr sM = (lds_dptr)m_off; const lds_dptr sInvD = (lds_dptr)d_off;
  const int lane = threadIdx.x & 63;
#define NEP_CHOL(N) chol_impl<N>(sM, sInvD, lane)
  NEP_SIZE_SWITCH(NEP_CHOL)
#undef NEP_CHOL
}

// x = (L L')^-1 b with L from chol_wave; lane i holds b[i] / returns x[i].
static __device__ __noinline__ double solve_wave(unsigned m_off, unsigned d_off, int n, double b) {
  m_off = __builtin_amdgcn_readfirstlane(m_off); d_off = __builtin_amdgcn_readfirstlane(d_off); n = __builtin_amdgcn_readfirstlane(n);
  const lds_dptr sM = (lds_dptr)m_off; const lds_dptr sInvD = (lds_dptr)d_off;
  const int lane = threadIdx.x & 63;
#define NEP_SOLVE(N) solve_impl<N>(sM, sInvD, lane, b)
  NEP_SIZE_SWITCH(NEP_SOLVE)
#undef NEP_SOLVE
}

}  // namespace nep
