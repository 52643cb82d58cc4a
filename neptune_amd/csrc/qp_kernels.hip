// qp_kernels.hip — batched primal-dual interior point for the back-end spline QP on gfx950.
//
// Replaces PolySolverGurobi::addObjective/addConstraints/optimize/generatePwpOut
// (reference neptune/src/solver_gurobi_poly.cpp:322-710, 804-936): one 256-thread workgroup per
// agent replan, every interior-point iteration inside one launch (no host round trips).
//
// Problem in reduced variables (nep_tables.h): x = [z_x; z_y; z_z], n = 3*nz <= 24.  Every
// inequality row is  alpha . (B[rho] z_axis + off[rho][axis]) <= h  for one of R = 8K "base
// rows" rho (4K position control points, 3K velocity control points, K accelerations):
//   box rows   alpha = +-e_axis                    (solver_gurobi_poly.cpp:433-471), 6R rows
//   line rows  alpha = (n1, n2, 0), rho = 4*seg+k  (:485-489 and the three other call sites)
// so the normal matrix  H + G' W G  collapses to  Hax (+) sum_rho D[rho] (x) B[rho]'B[rho]  with
// 4 weights per base row; rows only ever touch three R x 3 arrays staged in LDS.
//
// Thread roles: t < 3R owns the two box rows of (rho, axis); every thread also owns slice t&7
// of the lines of control point (seg,k) = (t>>5, (t>>3)&3): the 8 slices of one control point
// sit in consecutive lanes, so the scatter onto base rows is three xor-shuffles, not atomics.
// Row state (s, lambda) lives in LDS (global scratch only if it does not fit).
#include <hip/hip_runtime.h>

#include "nep_device.h"

namespace nep {

constexpr int BS = 256;
constexpr int MS = 24;            // matrix stride / max n
constexpr int kBoxRows = 6 * kMaxR;  // 384: line rows start here in the state arrays
constexpr int kMaxIt = 60;

// ---- LDS carve (in doubles) -------------------------------------------------------------------
constexpr int oB = 0;                       // [64][8]
constexpr int oU = oB + kMaxR * kNZ;        // [64][3]
constexpr int oOff = oU + kMaxR * 3;        // [64][3]
constexpr int oCp = oOff + kMaxR * 3;       // [64][3] base-row values at z
constexpr int oUa = oCp + kMaxR * 3;        // [64][3] B dx_aff
constexpr int oUd = oUa + kMaxR * 3;        // [64][3] B dx
constexpr int oAccL = oUd + kMaxR * 3;      // [32][8] line accumulators per control point
constexpr int oAccB = oAccL + 32 * 8;       // [192][4] box accumulators per (rho,axis)
constexpr int oM = oAccB + 192 * 4;         // [24][24]
constexpr int oHax = oM + MS * MS;          // [8][8]
constexpr int oTh = oHax + 64;              // [32][8]
constexpr int oThU = oTh + 32 * kNZ;        // [32][3]
constexpr int oZ = oThU + 32 * 3;           // [24]
constexpr int oG = oZ + MS;                 // [24]
constexpr int oRd = oG + MS;                // [24]
constexpr int oRhs = oRd + MS;              // [24]
constexpr int oDxa = oRhs + MS;             // [24]
constexpr int oDx = oDxa + MS;              // [24]
constexpr int oGq = oDx + MS;               // [24]
constexpr int oZl = oGq + MS;               // [24] loose snapshot
constexpr int oInvD = oZl + MS;             // [24]
constexpr int oEp = oInvD + MS;             // [8]
constexpr int oCoef = oEp + 8;              // [3][8][4] initial guess
constexpr int oTheta = oCoef + 96;          // [3][8][4] result
constexpr int oInit = oTheta + 96;          // [3][3] b0,c0,d0 per axis
constexpr int oScal = oInit + 9;            // scalars, see enum
constexpr int oRed = oScal + 32;            // [8] reduction scratch
constexpr int oFixedEnd = oRed + 8;
constexpr int kFixedDoubles = (oFixedEnd + 1) & ~1;

enum { sFinal0 = 0, sFinal1, sFinal2, sMu, sSigma, sAlpha, sObj0, sObj, sSq, sLq, sRpq, sWq, sDsqA, sDlqA, sDsq, sDlq, sQscale, sNrp, sSumSl, sObjLoose };

size_t qp_lds_fixed_bytes() { return (size_t)kFixedDoubles * sizeof(double) + 64 * sizeof(int); }

__device__ __forceinline__ double wave_min(double v) { for (int o = 32; o > 0; o >>= 1) v = fmin(v, __shfl_xor(v, o)); return v; }
__device__ __forceinline__ double wave_max(double v) { for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_xor(v, o)); return v; }
__device__ __forceinline__ double wave_sum(double v) { for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o); return v; }

// block-wide reductions (all threads must call; result broadcast).  red: LDS [8]
template <int OP>
__device__ __forceinline__ double block_reduce(double v, double* red) {
  v = OP == 0 ? wave_min(v) : (OP == 1 ? wave_max(v) : wave_sum(v));
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  const double a = red[0], b = red[1], c = red[2], d = red[3];
  return OP == 0 ? fmin(fmin(a, b), fmin(c, d)) : (OP == 1 ? fmax(fmax(a, b), fmax(c, d)) : (a + b) + (c + d));
}
__device__ __forceinline__ double slice_sum(double v) { v += __shfl_xor(v, 1); v += __shfl_xor(v, 2); v += __shfl_xor(v, 4); return v; }

struct RowCtx {
  // box role
  bool has_box; int q, brho, bax; double bhi, blo;
  // line role
  bool has_line; int lbeg, lend, lk, lrho, slice;
  const double *n1, *n2, *lh;
};

// Enumerates this thread's rows: f(state_index, rho, ax, ay, az, h, is_line)
template <class F>
__device__ __forceinline__ void for_rows(const RowCtx& c, F&& f) {
  if (c.has_box) {
    const double ex = c.bax == 0 ? 1.0 : 0.0, ey = c.bax == 1 ? 1.0 : 0.0, ez = c.bax == 2 ? 1.0 : 0.0;
    f(2 * c.q, c.brho, ex, ey, ez, c.bhi, false);
    f(2 * c.q + 1, c.brho, -ex, -ey, -ez, -c.blo, false);
  }
  if (c.has_line) {
    for (int l = c.lbeg + c.slice; l < c.lend; l += 8) f(kBoxRows + 4 * l + c.lk, c.lrho, c.n1[l], c.n2[l], 0.0, c.lh[l], true);
  }
}

__global__ __launch_bounds__(BS) void qp_kernel(SceneParams sp, ProblemSet ps, const QpTable* __restrict__ tables, SampleSched sched) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
  double* sB = smem + oB; double* sU = smem + oU; double* sOff = smem + oOff; double* sCp = smem + oCp;
  double* sUa = smem + oUa; double* sUd = smem + oUd; double* sAccL = smem + oAccL; double* sAccB = smem + oAccB;
  double* sM = smem + oM; double* sHax = smem + oHax; double* sTh = smem + oTh; double* sThU = smem + oThU;
  double* sZ = smem + oZ; double* sG = smem + oG; double* sRd = smem + oRd; double* sRhs = smem + oRhs;
  double* sDxa = smem + oDxa; double* sDx = smem + oDx; double* sGq = smem + oGq; double* sZl = smem + oZl;
  double* sInvD = smem + oInvD; double* sEp = smem + oEp; double* sCoef = smem + oCoef; double* sTheta = smem + oTheta;
  double* sInit = smem + oInit; double* sc = smem + oScal; double* sRed = smem + oRed;
  int* sI = (int*)(smem + kFixedDoubles);      // [0..8] line offsets, [16] flags
  double* dyn = smem + kFixedDoubles + 32;     // dynamic part: lines then row state

  const int tid = threadIdx.x;
  const int slot = blockIdx.x;
  const nep_guess* __restrict__ g = ps.guess + slot;
  const int K = g->K;
  const double T = sp.T_span, wgt = sp.weight;
  nep_solution* __restrict__ sol = ps.solution + slot;

  // ---- stage the guess ------------------------------------------------------------------------
  if (tid < 96) sCoef[tid] = (&g->coeff[0][0][0])[tid];
  if (tid == 0) {
    int o = 0;
    for (int i = 0; i < NEP_MAX_POL; i++) { sI[i] = o; int c = (i < K) ? ps.line_cnt[(long)slot * NEP_MAX_POL + i] : 0; o += c; }
    sI[NEP_MAX_POL] = o;
  }
  __syncthreads();
  const int L = sI[NEP_MAX_POL];
  const int m_rows = kBoxRows + 4 * L;
  const bool in_lds = (L <= ps.lds_lines) && (m_rows <= ps.lds_rows);
  double* lines = in_lds ? dyn : ps.row_scratch + (long)slot * (2L * ps.rows_cap + 3L * (ps.rows_cap / 4));
  const int lstride = in_lds ? ps.lds_lines : ps.rows_cap / 4;
  double* sN1 = lines; double* sN2 = lines + lstride; double* sLh = lines + 2 * lstride;
  double* stS = in_lds ? dyn + 3 * ps.lds_lines : lines + 3 * lstride;
  double* stL = stS + (in_lds ? ps.lds_rows : ps.rows_cap);
  if (tid < 9) {
    if (tid < 3) {
      const double* c = sCoef + (tid * 8 + (K - 1)) * 4;
      sc[sFinal0 + tid] = ((T * T * T) * c[0] + (T * T) * c[1] + T * c[2]) + c[3];   // final_pos_ (:226-228)
    }
    sInit[tid] = sCoef[((tid / 3) * 8 + 0) * 4 + 1 + (tid % 3)];                    // b0,c0,d0 (:390-396)
  }
  for (int i = 0; i < K; i++) {  // gather the separator's buckets into one segment-major list
    const int beg = sI[i], cnt = sI[i + 1] - beg;
    const double* src = ps.line_nd + ((long)slot * NEP_MAX_POL + i) * sp.lines_cap * 3;
    for (int l = tid; l < cnt; l += BS) { sN1[beg + l] = src[3 * l]; sN2[beg + l] = src[3 * l + 1]; sLh[beg + l] = 1.0 - src[3 * l + 2]; }
  }
  __syncthreads();
  const double f0 = sc[sFinal0], f1 = sc[sFinal1], f2 = sc[sFinal2];
  const double dix = sCoef[3] - f0, diy = sCoef[32 + 3] - f1, diz = sCoef[64 + 3] - f2;
  const bool has_qc = sqrt(dix * dix + diy * diy + diz * diz) < 1.0;   // :697-702
  const bool z_override = sqrt(dix * dix + diy * diy) < 1.0;           // :879-880
  const int mt = m_rows - (kBoxRows - 6 * 8 * K) + (has_qc ? 1 : 0);   // actual inequality count (+ball)

  // ---- thread roles ---------------------------------------------------------------------------
  const int R = 8 * K;
  RowCtx rc;
  rc.has_box = tid < 3 * R; rc.q = tid; rc.bax = rc.has_box ? tid / R : 0; rc.brho = rc.has_box ? tid % R : 0;
  {
    const int rho = rc.brho, ax = rc.bax;
    rc.bhi = rho < 4 * K ? sp.maxs[ax] : (rho < 7 * K ? sp.v_max : sp.a_max);
    rc.blo = rho < 4 * K ? sp.mins[ax] : (rho < 7 * K ? -sp.v_max : -sp.a_max);
  }
  const int pair = tid >> 3, li = pair >> 2;
  rc.lk = pair & 3; rc.slice = tid & 7; rc.has_line = li < K; rc.lrho = 4 * li + rc.lk;
  rc.lbeg = rc.has_line ? sI[li] : 0; rc.lend = rc.has_line ? sI[li + 1] : 0;
  rc.n1 = sN1; rc.n2 = sN2; rc.lh = sLh;

  int status = NEP_FAILED, iters_total = 0, iters_first = 0;
  double objective = 0.0;

  for (int mode = 0; mode < 2; mode++) {
    const QpTable* __restrict__ tb = tables + mode * (kMaxK + 1) + K;
    const int nz = tb->nz, n = 3 * nz;
    __syncthreads();
    for (int e = tid; e < kMaxR * kNZ; e += BS) sB[e] = (&tb->B[0][0])[e];
    for (int e = tid; e < 32 * kNZ; e += BS) sTh[e] = (&tb->Th[0][0])[e];
    if (tid < kMaxR * 3) sU[tid] = (&tb->U[0][0])[tid];
    if (tid < 96) sThU[tid] = (&tb->ThU[0][0])[tid];
    if (tid < 64) sHax[tid] = (&tb->Hax[0][0])[tid];
    if (tid < 8) sEp[tid] = tb->ep[tid];
    __syncthreads();
    if (tid < 3 * R) { const int rho = tid % R, ax = tid / R; sOff[rho * 3 + ax] = sU[rho * 3] * sInit[ax * 3] + sU[rho * 3 + 1] * sInit[ax * 3 + 1] + sU[rho * 3 + 2] * sInit[ax * 3 + 2]; }
    bool converged = false;
    int it = 0;
    if (nz == 0) {
      // ---- K <= 2 with the terminal rows: a single point, feasible or not (tolerance 1e-6) ------
      __syncthreads();
      double viol = 0.0;
      for_rows(rc, [&](int, int rho, double ax, double ay, double az, double h, bool) {
        const double a = ax * sOff[rho * 3] + ay * sOff[rho * 3 + 1] + az * sOff[rho * 3 + 2];
        viol = fmax(viol, a - h);
      });
      if (tid < 6) {  // terminal v = a = 0 must hold at the least-squares point
        const int ax = tid / 2, e = tid % 2;
        viol = fmax(viol, fabs(tb->res_u[e][0] * sInit[ax * 3] + tb->res_u[e][1] * sInit[ax * 3 + 1] + tb->res_u[e][2] * sInit[ax * 3 + 2]));
      }
      if (tid == 0 && has_qc) {
        double c = -0.10 * 0.10;
        for (int ax = 0; ax < 3; ax++) { const double pe = tb->up[0] * sInit[ax * 3] + tb->up[1] * sInit[ax * 3 + 1] + tb->up[2] * sInit[ax * 3 + 2] - sc[sFinal0 + ax]; c += pe * pe; }
        viol = fmax(viol, c);
      }
      viol = block_reduce<1>(viol, sRed);
      converged = viol <= 1e-6;
      if (tid == 0) {
        double o = 0;
        for (int ax = 0; ax < 3; ax++) {
          for (int r = 0; r < K; r++) { const double a = tb->Pp[r][0] * sInit[ax * 3] + tb->Pp[r][1] * sInit[ax * 3 + 1] + tb->Pp[r][2] * sInit[ax * 3 + 2]; o += 36 * T * a * a; }
          const double pe = tb->up[0] * sInit[ax * 3] + tb->up[1] * sInit[ax * 3 + 1] + tb->up[2] * sInit[ax * 3 + 2] - sc[sFinal0 + ax];
          o += wgt * pe * pe;
        }
        sc[sObj] = o;
      }
    } else {
      // ---- start point: projection of the guess, floored slacks, centred duals ------------------
      if (tid < n) {
        const int ax = tid / nz, c = tid % nz;
        double z = 0;
        for (int r = 0; r < K; r++) {
          const double ap = tb->Pp[r][0] * sInit[ax * 3] + tb->Pp[r][1] * sInit[ax * 3 + 1] + tb->Pp[r][2] * sInit[ax * 3 + 2];
          z += tb->Nt[c][r] * (sCoef[(ax * 8 + r) * 4] - ap);
        }
        sZ[tid] = z;
        sG[tid] = (tb->Gi[c][0] * sInit[ax * 3] + tb->Gi[c][1] * sInit[ax * 3 + 1] + tb->Gi[c][2] * sInit[ax * 3 + 2]) - 2 * wgt * sEp[c] * sc[sFinal0 + ax];
      }
      if (tid == 0) {
        double o = 0;
        for (int ax = 0; ax < 3; ax++) {
          for (int r = 0; r < K; r++) { const double a = tb->Pp[r][0] * sInit[ax * 3] + tb->Pp[r][1] * sInit[ax * 3 + 1] + tb->Pp[r][2] * sInit[ax * 3 + 2]; o += 36 * T * a * a; }
          const double pe = tb->up[0] * sInit[ax * 3] + tb->up[1] * sInit[ax * 3 + 1] + tb->up[2] * sInit[ax * 3 + 2] - sc[sFinal0 + ax];
          o += wgt * pe * pe;
          if (mode == 1) {
            const double ve = tb->uv[0] * sInit[ax * 3] + tb->uv[1] * sInit[ax * 3 + 1] + tb->uv[2] * sInit[ax * 3 + 2];
            const double ae = tb->ua[0] * sInit[ax * 3] + tb->ua[1] * sInit[ax * 3 + 1] + tb->ua[2] * sInit[ax * 3 + 2];
            o += wgt * (ve * ve + ae * ae);
          }
        }
        sc[sObj0] = o;
        sI[16] = 0;  // loose snapshot present
        sI[17] = 0;  // stall counter
      }
      __syncthreads();
      if (tid < 3 * R) { const int rho = tid % R, ax = tid / R; double v = sOff[rho * 3 + ax]; for (int c = 0; c < nz; c++) v += sB[rho * kNZ + c] * sZ[ax * nz + c]; sCp[rho * 3 + ax] = v; }
      __syncthreads();
      for_rows(rc, [&](int r, int rho, double ax, double ay, double az, double h, bool) {
        const double a = ax * sCp[rho * 3] + ay * sCp[rho * 3 + 1] + az * sCp[rho * 3 + 2];
        const double sl = h - a; const double s = sl > 0.1 ? sl : 0.1;
        stS[r] = s; stL[r] = 1.0 / s;
      });
      if (tid == 0) {
        double qs = 1.0; for (int e = 0; e < n; e++) qs = fmax(qs, fabs(sG[e]));
        sc[sQscale] = qs;
        if (has_qc) {
          double c = -0.10 * 0.10;
          for (int ax = 0; ax < 3; ax++) { double pe = tb->up[0] * sInit[ax * 3] + tb->up[1] * sInit[ax * 3 + 1] + tb->up[2] * sInit[ax * 3 + 2] - sc[sFinal0 + ax]; for (int e = 0; e < nz; e++) pe += sEp[e] * sZ[ax * nz + e]; c += pe * pe; }
          const double sq = (-c > 1e-3) ? -c : 1e-3;
          sc[sSq] = sq; sc[sLq] = 1.0 / sq;
        } else { sc[sSq] = 1.0; sc[sLq] = 0.0; }
      }
      __syncthreads();

      for (it = 0; it < kMaxIt; it++) {
        // ---- (A) base-row values at z ---------------------------------------------------------
        if (tid < 3 * R) { const int rho = tid % R, ax = tid / R; double v = sOff[rho * 3 + ax]; for (int c = 0; c < nz; c++) v += sB[rho * kNZ + c] * sZ[ax * nz + c]; sCp[rho * 3 + ax] = v; }
        __syncthreads();
        // ---- (P1) residuals, weights, scatter onto base rows ----------------------------------
        double bTl = 0, bD = 0, bT1 = 0;                                  // box: (rho,axis)
        double lTx = 0, lTy = 0, lDxx = 0, lDxy = 0, lDyy = 0, l1x = 0, l1y = 0;  // lines: (seg,k)
        double nrp = 0, sumsl = 0;
        for_rows(rc, [&](int r, int rho, double ax, double ay, double az, double h, bool is_line) {
          const double s = stS[r], lam = stL[r];
          const double a = ax * sCp[rho * 3] + ay * sCp[rho * 3 + 1] + az * sCp[rho * 3 + 2];
          const double rp = a + s - h;
          const double w = lam / s;
          const double v = lam - w * rp;
          nrp = fmax(nrp, fabs(rp)); sumsl += s * lam;
          if (is_line) { lTx += lam * ax; lTy += lam * ay; lDxx += w * ax * ax; lDxy += w * ax * ay; lDyy += w * ay * ay; l1x += v * ax; l1y += v * ay; }
          else { const double sg = ax + ay + az; bTl += lam * sg; bD += w; bT1 += v * sg; }
        });
        lTx = slice_sum(lTx); lTy = slice_sum(lTy); lDxx = slice_sum(lDxx); lDxy = slice_sum(lDxy); lDyy = slice_sum(lDyy); l1x = slice_sum(l1x); l1y = slice_sum(l1y);
        if (rc.slice == 0) { double* o = sAccL + pair * 8; o[0] = lTx; o[1] = lTy; o[2] = lDxx; o[3] = lDxy; o[4] = lDyy; o[5] = l1x; o[6] = l1y; }
        if (tid < 192) { double* o = sAccB + tid * 4; o[0] = bTl; o[1] = bD; o[2] = bT1; }
        nrp = block_reduce<1>(nrp, sRed);
        sumsl = block_reduce<2>(sumsl, sRed);   // (syncs inside also publish sAcc*)
        // ---- ball constraint (scalar row, thread 0) -------------------------------------------
        if (tid == 0) {
          double rpq = 0;
          if (has_qc) {
            double c = -0.10 * 0.10;
            for (int ax = 0; ax < 3; ax++) {
              double pe = tb->up[0] * sInit[ax * 3] + tb->up[1] * sInit[ax * 3 + 1] + tb->up[2] * sInit[ax * 3 + 2] - sc[sFinal0 + ax];
              for (int e = 0; e < nz; e++) pe += sEp[e] * sZ[ax * nz + e];
              c += pe * pe;
              for (int e = 0; e < nz; e++) sGq[ax * nz + e] = 2 * pe * sEp[e];
            }
            rpq = c + sc[sSq];
            sc[sWq] = sc[sLq] / sc[sSq];
          }
          sc[sRpq] = rpq;
          sc[sMu] = (sumsl + (has_qc ? sc[sSq] * sc[sLq] : 0.0)) / mt;
          sc[sNrp] = fmax(nrp, fabs(rpq));
        }
        // ---- dual residual, normal matrix, predictor right-hand side ---------------------------
        if (tid < n) {
          const int ax = tid / nz, c = tid % nz;
          double v = sG[tid];
          for (int e = 0; e < nz; e++) v += sHax[c * kNZ + e] * sZ[ax * nz + e];
          double t1 = 0;
          for (int rho = 0; rho < R; rho++) {
            double tl = sAccB[(ax * R + rho) * 4], tt = sAccB[(ax * R + rho) * 4 + 2];
            if (ax < 2 && rho < 4 * K) { tl += sAccL[rho * 8 + ax]; tt += sAccL[rho * 8 + 5 + ax]; }
            v += sB[rho * kNZ + c] * tl; t1 += sB[rho * kNZ + c] * tt;
          }
          sRd[tid] = v; sRhs[tid] = t1;   // rhs completed after the ball terms are known
        }
        for (int e = tid; e < n * n; e += BS) {
          const int i = e / n, j = e % n;
          const int ai = i / nz, ci = i % nz, aj = j / nz, cj = j % nz;
          double v = 0;
          if (ai == aj) {
            v = sHax[ci * kNZ + cj];
            for (int rho = 0; rho < R; rho++) {
              double d = sAccB[(ai * R + rho) * 4 + 1];
              if (ai < 2 && rho < 4 * K) d += sAccL[rho * 8 + (ai == 0 ? 2 : 4)];
              v += d * sB[rho * kNZ + ci] * sB[rho * kNZ + cj];
            }
          } else if (ai < 2 && aj < 2) {
            for (int rho = 0; rho < 4 * K; rho++) v += sAccL[rho * 8 + 3] * sB[rho * kNZ + ci] * sB[rho * kNZ + cj];
          }
          sM[i * MS + j] = v;
        }
        __syncthreads();
        if (has_qc) {
          const double lq = sc[sLq], wq = sc[sWq];
          for (int e = tid; e < n * n; e += BS) {
            const int i = e / n, j = e % n;
            double v = wq * sGq[i] * sGq[j];
            if (i / nz == j / nz) v += lq * 2 * sEp[i % nz] * sEp[j % nz];
            sM[i * MS + j] += v;
          }
          if (tid < n) sRd[tid] += lq * sGq[tid];
          __syncthreads();
        }
        // ---- convergence test (thread 0) ------------------------------------------------------
        if (tid == 0) {
          double nrd = 0, o = sc[sObj0];
          for (int e = 0; e < n; e++) nrd = fmax(nrd, fabs(sRd[e]));
          for (int ax = 0; ax < 3; ax++) for (int a = 0; a < nz; a++) { double v = 0; for (int b = 0; b < nz; b++) v += sHax[a * kNZ + b] * sZ[ax * nz + b]; o += 0.5 * sZ[ax * nz + a] * v + sG[ax * nz + a] * sZ[ax * nz + a]; }
          sc[sObj] = o;
          const double gap = sc[sMu] * mt, nr = sc[sNrp], qs = sc[sQscale];
          int flag = 0;
          if (nr <= 1e-9 && nrd <= 1e-9 * qs && gap <= 1e-10 * (1.0 + fabs(o))) flag = 1;
          else if (nr <= 1e-6 && nrd <= 1e-6 * qs && gap <= 1e-7 * (1.0 + fabs(o))) { flag = 2; sc[sObjLoose] = o; }
          if (!(sc[sMu] < 1e30) || !(nrd < 1e300)) flag = 3;  // diverged / NaN
          sI[18] = flag;
        }
        __syncthreads();
        const int flag = sI[18];
        if (flag == 1) { converged = true; break; }
        if (flag == 3) break;
        if (flag == 2) { if (tid < n) sZl[tid] = sZ[tid]; if (tid == 0) sI[16] = 1; }
        // ---- Cholesky of M (right-looking, in LDS) --------------------------------------------
        bool chol_ok = true;
        for (int j = 0; j < n; j++) {
          const double d = sM[j * MS + j];
          if (!(d > 0.0)) { chol_ok = false; break; }
          const double inv = 1.0 / sqrt(d);
          __syncthreads();
          if (tid >= j && tid < n) sM[tid * MS + j] *= inv;
          if (tid == 0) sInvD[j] = inv;
          __syncthreads();
          for (int e = tid; e < n * n; e += BS) { const int i = e / n, k = e % n; if (k > j && i >= k) sM[i * MS + k] -= sM[i * MS + j] * sM[k * MS + j]; }
          __syncthreads();
        }
        if (!chol_ok) break;
        // ---- predictor / corrector -----------------------------------------------------------
        double alpha = 1.0;
        for (int pass = 0; pass < 2; pass++) {
          const double mu = sc[sMu];
          if (pass == 1) {
            // corrector right-hand side: T1' = sum (rc/s - w rp) alpha with rc = s lam - sigma mu + ds_a dl_a
            const double sm = sc[sSigma] * mu;
            double b1 = 0, c1x = 0, c1y = 0;
            for_rows(rc, [&](int r, int rho, double ax, double ay, double az, double h, bool is_line) {
              const double s = stS[r], lam = stL[r];
              const double a = ax * sCp[rho * 3] + ay * sCp[rho * 3 + 1] + az * sCp[rho * 3 + 2];
              const double rp = a + s - h, w = lam / s;
              const double ga = ax * sUa[rho * 3] + ay * sUa[rho * 3 + 1] + az * sUa[rho * 3 + 2];
              const double dsa = -rp - ga, dla = -lam + w * (rp + ga);
              const double rcv = s * lam - sm + dsa * dla;
              const double v = rcv / s - w * rp;
              if (is_line) { c1x += v * ax; c1y += v * ay; } else b1 += v * (ax + ay + az);
            });
            c1x = slice_sum(c1x); c1y = slice_sum(c1y);
            __syncthreads();
            if (rc.slice == 0) { sAccL[pair * 8 + 5] = c1x; sAccL[pair * 8 + 6] = c1y; }
            if (tid < 192) sAccB[tid * 4 + 2] = b1;
            __syncthreads();
            if (tid < n) {
              const int ax = tid / nz, c = tid % nz;
              double t1 = 0;
              for (int rho = 0; rho < R; rho++) { double tt = sAccB[(ax * R + rho) * 4 + 2]; if (ax < 2 && rho < 4 * K) tt += sAccL[rho * 8 + 5 + ax]; t1 += sB[rho * kNZ + c] * tt; }
              sRhs[tid] = t1;
            }
            __syncthreads();
          }
          // rhs = -rd + B'T1 (+ gq (rcq/sq - wq rpq))
          if (tid < 64) {
            double b = 0;
            if (tid < n) {
              b = -sRd[tid] + sRhs[tid];
              if (has_qc) {
                const double sq = sc[sSq], lq = sc[sLq];
                const double rcq = (pass == 0) ? sq * lq : sq * lq - sc[sSigma] * mu + sc[sDsqA] * sc[sDlqA];
                b += sGq[tid] * (rcq / sq - sc[sWq] * sc[sRpq]);
              }
            }
            // L y = b, L' x = y inside wave 0 (lane i holds entry i)
            for (int j = 0; j < n; j++) { const double xj = __shfl(b, j) * sInvD[j]; if (tid == j) b = xj; else if (tid > j && tid < n) b -= sM[tid * MS + j] * xj; }
            for (int j = n - 1; j >= 0; j--) { const double xj = __shfl(b, j) * sInvD[j]; if (tid == j) b = xj; else if (tid < j) b -= sM[j * MS + tid] * xj; }
            if (tid < n) (pass == 0 ? sDxa : sDx)[tid] = b;
          }
          __syncthreads();
          double* sDir = pass == 0 ? sDxa : sDx; double* sUu = pass == 0 ? sUa : sUd;
          if (tid < 3 * R) { const int rho = tid % R, ax = tid / R; double v = 0; for (int c = 0; c < nz; c++) v += sB[rho * kNZ + c] * sDir[ax * nz + c]; sUu[rho * 3 + ax] = v; }
          __syncthreads();
          // step length
          const double sm = (pass == 0) ? 0.0 : sc[sSigma] * mu;
          double amin = 1.0;
          for_rows(rc, [&](int r, int rho, double ax, double ay, double az, double h, bool) {
            const double s = stS[r], lam = stL[r];
            const double a = ax * sCp[rho * 3] + ay * sCp[rho * 3 + 1] + az * sCp[rho * 3 + 2];
            const double rp = a + s - h, w = lam / s;
            const double ga = ax * sUa[rho * 3] + ay * sUa[rho * 3 + 1] + az * sUa[rho * 3 + 2];
            double ds, dl;
            if (pass == 0) { ds = -rp - ga; dl = -lam + w * (rp + ga); }
            else {
              const double dsa = -rp - ga, dla = -lam + w * (rp + ga);
              const double rcv = s * lam - sm + dsa * dla;
              const double gd = ax * sUd[rho * 3] + ay * sUd[rho * 3 + 1] + az * sUd[rho * 3 + 2];
              ds = -rp - gd; dl = -rcv / s + w * (rp + gd);
            }
            if (ds < 0) amin = fmin(amin, -s / ds);
            if (dl < 0) amin = fmin(amin, -lam / dl);
          });
          if (tid == 0 && has_qc) {
            const double sq = sc[sSq], lq = sc[sLq], wq = sc[sWq], rpq = sc[sRpq];
            double gd = 0; for (int e = 0; e < n; e++) gd += sGq[e] * sDir[e];
            const double rcq = (pass == 0) ? sq * lq : sq * lq - sm + sc[sDsqA] * sc[sDlqA];
            const double dsq = -rpq - gd, dlq = -rcq / sq + wq * (rpq + gd);
            if (pass == 0) { sc[sDsqA] = dsq; sc[sDlqA] = dlq; } else { sc[sDsq] = dsq; sc[sDlq] = dlq; }
            if (dsq < 0) amin = fmin(amin, -sq / dsq);
            if (dlq < 0) amin = fmin(amin, -lq / dlq);
          }
          alpha = block_reduce<0>(amin, sRed);
          if (pass == 0) {
            // mu_aff -> sigma
            double part = 0;
            for_rows(rc, [&](int r, int rho, double ax, double ay, double az, double h, bool) {
              const double s = stS[r], lam = stL[r];
              const double a = ax * sCp[rho * 3] + ay * sCp[rho * 3 + 1] + az * sCp[rho * 3 + 2];
              const double rp = a + s - h, w = lam / s;
              const double ga = ax * sUa[rho * 3] + ay * sUa[rho * 3 + 1] + az * sUa[rho * 3 + 2];
              const double ds = -rp - ga, dl = -lam + w * (rp + ga);
              part += (s + alpha * ds) * (lam + alpha * dl);
            });
            if (tid == 0 && has_qc) part += (sc[sSq] + alpha * sc[sDsqA]) * (sc[sLq] + alpha * sc[sDlqA]);
            part = block_reduce<2>(part, sRed);
            if (tid == 0) { const double rr = (part / mt) / mu; sc[sSigma] = rr * rr * rr; }
            __syncthreads();
          }
        }
        alpha = fmin(1.0, 0.995 * alpha);
        if (tid == 0) { if (alpha < 1e-8) sI[17]++; else sI[17] = 0; }
        // ---- update ---------------------------------------------------------------------------
        {
          const double sm = sc[sSigma] * sc[sMu];
          for_rows(rc, [&](int r, int rho, double ax, double ay, double az, double h, bool) {
            const double s = stS[r], lam = stL[r];
            const double a = ax * sCp[rho * 3] + ay * sCp[rho * 3 + 1] + az * sCp[rho * 3 + 2];
            const double rp = a + s - h, w = lam / s;
            const double ga = ax * sUa[rho * 3] + ay * sUa[rho * 3 + 1] + az * sUa[rho * 3 + 2];
            const double dsa = -rp - ga, dla = -lam + w * (rp + ga);
            const double rcv = s * lam - sm + dsa * dla;
            const double gd = ax * sUd[rho * 3] + ay * sUd[rho * 3 + 1] + az * sUd[rho * 3 + 2];
            const double ds = -rp - gd, dl = -rcv / s + w * (rp + gd);
            stS[r] = s + alpha * ds; stL[r] = lam + alpha * dl;
          });
        }
        __syncthreads();
        if (tid < n) sZ[tid] += alpha * sDx[tid];
        if (tid == 0 && has_qc) { sc[sSq] += alpha * sc[sDsq]; sc[sLq] += alpha * sc[sDlq]; }
        __syncthreads();
        if (sI[17] >= 3) break;
      }
      if (!converged && sI[16]) { __syncthreads(); if (tid < n) sZ[tid] = sZl[tid]; if (tid == 0) sc[sObj] = sc[sObjLoose]; converged = true; }
    }
    iters_total = it; if (mode == 0) iters_first = it;
    __syncthreads();
    if (converged) {
      status = mode;   // NEP_OK / NEP_RELAXED
      objective = sc[sObj];
      if (tid < 12 * K) {  // theta = Th z + ThU init
        const int ax = tid / (4 * K), r = tid % (4 * K);
        double v = sThU[r * 3] * sInit[ax * 3] + sThU[r * 3 + 1] * sInit[ax * 3 + 1] + sThU[r * 3 + 2] * sInit[ax * 3 + 2];
        for (int c = 0; c < nz; c++) v += sTh[r * kNZ + c] * sZ[ax * nz + c];
        sTheta[(ax * 8 + r / 4) * 4 + (r % 4)] = v;
      }
      break;
    }
  }
  __syncthreads();
  // ---- outputs -----------------------------------------------------------------------------------
  if (status == NEP_FAILED) { if (tid < 96) sTheta[tid] = sCoef[tid]; }                    // :856-859
  else if (z_override) { if (tid < 32) sTheta[64 + tid] = sCoef[64 + tid]; }             // :879-880
  __syncthreads();
  if (tid < 96) (&sol->coeff[0][0][0])[tid] = ((tid % 32) / 4 < K) ? sTheta[tid] : 0.0;
  if (tid <= NEP_MAX_POL) sol->times[tid] = (tid <= K) ? g->t_start + tid * T : 0.0;       // :898 (times = i*T_span + t_start)
  const int ns_all = sched.n[K];
  const int ns = ns_all < sp.max_states ? ns_all : sp.max_states;
  if (tid == 0) {
    sol->stats.status = status; sol->stats.iters = iters_total; sol->stats.iters_first = iters_first;
    sol->stats.n_lines = L; sol->stats.n_lp = ps.lp_stats ? ps.lp_stats[2 * slot] : 0; sol->stats.n_lp_failed = ps.lp_stats ? ps.lp_stats[2 * slot + 1] : 0;
    sol->stats.n_rows = 48 * K + 4 * L; sol->stats.qc_active = has_qc ? 1 : 0;
    sol->stats.objective = objective; sol->stats.solve_us = 0.0;
    sol->K = K; sol->n_states = ns;
  }
  if (ps.states) {  // generatePwpOut's samples (:911-934)
    for (int s = tid; s < ns; s += BS) {
      const int i = sched.seg[K * sp.max_states + s]; const double dt = sched.dt[K * sp.max_states + s];
      double* st = ps.states + ((long)slot * sp.max_states + s) * NEP_STATE_DOUBLES;
      for (int ax = 0; ax < 3; ax++) {
        const double* c = sTheta + (ax * 8 + i) * 4;
        st[ax] = ((c[0] * (dt * dt * dt) + c[1] * (dt * dt)) + c[2] * dt) + c[3];
        st[3 + ax] = (c[0] * (3 * dt * dt) + c[1] * (2 * dt)) + c[2];
        st[6 + ax] = c[0] * (6 * dt) + c[1] * 2;
        st[9 + ax] = c[0] * 6;
      }
    }
  }
  if (ps.commit) {  // the record the agent would publish (neptune_ros.cpp:434-480)
    nep_traj_rec* cr = ps.commit + slot;
    const int own = sp.first_local + (slot % sp.n_local);
    if (tid == 0) {
      cr->id = own + 1; cr->is_agent = 1; cr->n_bend = 1; cr->valid = 1;
      for (int a = 0; a < 3; a++) { cr->bbox[a] = 2 * sp.drone_radius; cr->pos[a] = sTheta[(a * 8) * 4 + 3]; }
      cr->bend[0][0] = ps.pb[2 * own]; cr->bend[0][1] = ps.pb[2 * own + 1];
      cr->pwp.n_seg = K;
    }
    if (tid <= NEP_TRAJ_MAX_SEG) cr->pwp.times[tid] = (tid <= K) ? g->t_start + tid * T : 0.0;
    for (int e = tid; e < 3 * NEP_TRAJ_MAX_SEG * 4; e += BS) {
      const int ax = e / (NEP_TRAJ_MAX_SEG * 4), r = e % (NEP_TRAJ_MAX_SEG * 4), seg = r / 4, j = r % 4;
      (&cr->pwp.coeff[0][0][0])[e] = (seg < K) ? sTheta[(ax * 8 + seg) * 4 + j] : 0.0;
    }
  }
}

void launch_qp(int n_slots, const SceneParams& sp, const ProblemSet& ps, const QpTable* tables,
               const SampleSched& sched, size_t lds_bytes, hipStream_t st) {
  if (n_slots <= 0) return;
  static size_t configured = 0;
  if (lds_bytes > configured) {
    hipFuncSetAttribute((const void*)qp_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
    configured = lds_bytes;
  }
  hipLaunchKernelGGL(qp_kernel, dim3(n_slots), dim3(BS), lds_bytes, st, sp, ps, tables, sched);
}

}  // namespace nep
