// qp_polish_kernel.hip — the active-set polish of interior-point solves that ended without passing the strict tests.
//
// PolySolverGurobi::optimize (reference neptune/src/solver_gurobi_poly.cpp:804-887) hands the QP to a solver that returns its
// optimum; the status it reports (:832-861) is decided by whether the rows are feasible.  The interior point of qp_reg_kernel /
// qp_kernel ends a solve in one of three ways: the strict tests pass (nearly always); the iterate is only LOOSELY converged (the
// dual residual sits on its rounding floor: the best iterate of a three-iteration window is returned, up to 8e-5 from the optimum
// in the coefficients); or the iteration gives up — on infeasible problems, but also on feasible ones whose optimum is degenerate
// (no strict complementarity: the gap stalls near 1e-2; tests/golden/moving_hard_cases.npz has them with HiGHS's verdict).
// The last two are finished here, exactly, by a kernel of its own over the few slots the QP kernel listed (ps.polish_list):
//
//   active set  = the rows whose slack at the solve's last iterate is below 1e-6 (1 + |rhs|);
//   repeat <= 6 times: solve the equality-constrained QP on a maximal independent subset of it (Schur complement of the
//   block-diagonal Hessian, Cholesky with dependent rows left out); drop the row with the most negative multiplier, else add the
//   most violated row, else stop;
//   certificate: every row satisfied to 1e-9 (1 + |rhs|), every multiplier >= -1e-9 (1 + max|nu|) — the KKT conditions of a
//   strictly convex QP: the point IS the optimum, whatever iterate it was found from; an infeasible problem can never pass.
//
// A certified mode-0 problem is NEP_OK (also when the interior point had gone on to the relaxed solve), a certified relaxed one
// NEP_RELAXED; without a certificate everything stays as the QP kernel left it.  Not for problems with the terminal ball row (a
// quadratic constraint, :680-702).  Under the line presolve the pass works on the near lines and accepts a certified point only if it
// also passes the presolve's own verification (parked lines, movement bound of the skipped LPs): see polish_slot.
// oracle/neptune_oracle.c::qp_solve runs the same rule; the two agree to the accuracy of a 24 x 24 solve (tests compare them).
#include <hip/hip_runtime.h>

#include <cstdlib>

#include "nep_device.h"

namespace nep {
namespace {
constexpr int PN = 24;          // reduced variables: 3 axes x nz <= 8
constexpr int PA = 40;          // active rows carried (an independent subset has at most PN)
constexpr int PST = PA + 1;     // LDS row stride of the Schur complement
constexpr int kPolRounds = 6;   // a certificate takes one to three rounds (oracle: the same bound)
constexpr int kPolLines = 768;  // lines staged in LDS (a config-4 replan has ~510; beyond: read where they lie)
constexpr double kActTol = 1e-6, kFeasTol = 1e-9, kDualTol = 1e-9, kPivTol = 1e-12, kStartTol = 1e-4;
// MINVO position basis inverse on [0, 1] (qp_common.h::cQpAPosInv, the literals of nep_tables.h::kAPosInv): the control points of the
// polished trajectory, for the presolve's parked-line and movement tests
__constant__ double cPolAPosInv[4][4] = {
    {-0.03203276669713047, -0.09273093424558249, 0.3420572455666699, 1.1023313949144335},
    {-0.05111494245568798, -0.046272612998418894, 0.5458234872124772, 1.0979806946005568},
    {-0.07454781852812224, 0.203951949894552, 0.796048050105448, 1.0745478185281223},
    {1.0, 1.0, 0.9999999999999996, 0.9999999999999993}};
}  // namespace

// one listed slot (a workgroup of 256); returns true when a problem of the slot was certified and its outputs rewritten
__device__ bool polish_slot(const SceneParams& sp, const ProblemSet& ps, const QpTable* __restrict__ tables, const SampleSched& sched, int slot) {
  const int flags = ps.polish_flag[slot];
  const int lane = threadIdx.x;      // (256 threads: the row scans, the table loads and the Schur complement on all of them, the small dense algebra on the first wave's)
  constexpr int NT = 256;
  const nep_guess* g = ps.guess + slot;
  nep_solution* sol = ps.solution + slot;
  const int K = g->K;
  if (K < 1 || K > NEP_MAX_POL || K > sp.num_pol) return false;
  const int R = 8 * K;
  const double T = sp.T_span;

  __shared__ double sB[kMaxR][kNZ], sOff[kMaxR][3], sHi[kNZ][kNZ], sG[PN], sZ[PN], sZ0[PN], sInit[9], sFin[3], sCoef[96], sTheta[96];
  __shared__ double sA[PA][PN + 1], sHA[PA][PN + 1], sS[PA][PST], sRhs[PA], sNu[PA], sRowH[PA], sRed[256];
  __shared__ int sAct[PA], sDrop[PA], sCnt[NEP_MAX_POL + 1], sI[8], sRedI[256], sDropped[12];
  __shared__ double sNd[kPolLines][3];      // the slot's separating lines (n1, n2, d): every scan of the rows reads them (from global memory a scan was a chain of round trips: most of a polish)

  if (lane < 96) sCoef[lane] = ((lane % 32) / 4 < K) ? (&g->coeff[0][0][0])[lane] : 0.0;
  if (lane <= NEP_MAX_POL) {
    int o = 0;
    for (int i = 0; i < NEP_MAX_POL; i++) { if (i == lane) sCnt[i] = o; o += i < K ? line_count(ps.line_cnt[(long)slot * NEP_MAX_POL + i]) : 0; }
    if (lane == NEP_MAX_POL) sCnt[NEP_MAX_POL] = o;
  }
  __syncthreads();
  if (lane < 9) sInit[lane] = sCoef[((lane / 3) * 8 + 0) * 4 + 1 + (lane % 3)];      // b0, c0, d0 per axis (:390-396)
  if (lane < 3) { const double* c = sCoef + (lane * 8 + (K - 1)) * 4; sFin[lane] = ((T * T * T) * c[0] + (T * T) * c[1] + T * c[2]) + c[3]; }      // final_pos_ (:226-228)
  __syncthreads();
  {
    const double dx = sCoef[3] - sFin[0], dy = sCoef[32 + 3] - sFin[1], dz = sCoef[64 + 3] - sFin[2];
    if (sqrt(dx * dx + dy * dy + dz * dz) < 1.0) return false;      // the terminal ball row is part of this problem (:697-702): not polished
  }
  const int L = sCnt[NEP_MAX_POL];
  const int n_rows = 6 * R + 4 * L;
  const double* bucket0 = ps.line_nd + (long)slot * NEP_MAX_POL * sp.lines_cap * 3;
  for (int l = lane; l < L && l < kPolLines; l += NT) {
    int i = 0;
    for (int j = 1; j < NEP_MAX_POL; j++) i += (l >= sCnt[j]) ? 1 : 0;
    const double* nd = bucket0 + ((long)i * sp.lines_cap + (l - sCnt[i])) * 3;
    sNd[l][0] = nd[0]; sNd[l][1] = nd[1]; sNd[l][2] = nd[2];
  }
  auto line_nd = [&](int l, int i, double& n1, double& n2, double& d) {
    if (l < kPolLines) { n1 = sNd[l][0]; n2 = sNd[l][1]; d = sNd[l][2]; }
    else { const double* nd = bucket0 + ((long)i * sp.lines_cap + (l - sCnt[i])) * 3; n1 = nd[0]; n2 = nd[1]; d = nd[2]; }
  };

  for (int mode = 0; mode < 2; mode++) {
    if (!((flags >> mode) & 1)) { if (mode == 0) continue; else break; }
    const QpTable* __restrict__ tb = tables + mode * (kMaxK + 1) + K;
    const int nz = mode == 1 ? K : (K > 2 ? K - 2 : 0), n = 3 * nz;
    if (nz == 0) continue;
    __syncthreads();
    for (int e = lane; e < kMaxR * kNZ; e += NT) sB[e / kNZ][e % kNZ] = (e / kNZ) < R ? tb->B[e / kNZ][e % kNZ] : 0.0;
    for (int e = lane; e < kMaxR * 3; e += NT) { const int rho = e / 3, ax = e % 3; sOff[rho][ax] = rho < R ? tb->U[rho][0] * sInit[ax * 3] + tb->U[rho][1] * sInit[ax * 3 + 1] + tb->U[rho][2] * sInit[ax * 3 + 2] : 0.0; }
    if (lane < kNZ * kNZ) sHi[lane / kNZ][lane % kNZ] = tb->HaxInv[lane / kNZ][lane % kNZ];
    if (lane < PN) {
      const int ax = lane / kNZ, e = lane % kNZ;      // (kNZ = 8: the axis blocks are padded to eight here; entries e >= nz stay zero)
      double gv = 0.0;
      if (e < nz && ax < 3) gv = (tb->Gi[e][0] * sInit[ax * 3] + tb->Gi[e][1] * sInit[ax * 3 + 1] + tb->Gi[e][2] * sInit[ax * 3 + 2]) - 2.0 * sp.weight * tb->ep[e] * sFin[ax];
      sG[lane] = gv;
      sZ[lane] = (e < nz) ? ps.polish_z[((long)slot * 2 + mode) * PN + ax * nz + e] : 0.0;
    }
    __syncthreads();
    // (from here on a variable is addressed as [axis][e] with stride kNZ: n_pad = 24 slots, unused ones zero)
    bool finite = true;
    for (int c = 0; c < PN; c++) finite = finite && (fabs(sZ[c]) < 1e100);
    if (!finite) continue;
    if (lane < PN) { const int ax = lane / kNZ, e = lane % kNZ; double v = 0.0; for (int c = 0; c < kNZ; c++) v -= sHi[e][c] * sG[ax * kNZ + c]; sZ0[lane] = v; }      // the minimiser without rows: -H^-1 g

    // a row of the problem: r < 6R: box row of (axis, base row, side); else line row (line l, control point k)
    auto row_eval = [&](int r, const double* z, double& rhs) -> double {      // -> slack h - a.z, rhs = the reference row's right-hand side
      if (r < 6 * R) {
        const int side = r & 1, q = r >> 1, ax = (q >= R ? 1 : 0) + (q >= 2 * R ? 1 : 0), rho = q - ax * R;      // (no integer division: forty instructions on the VALU)
        const double hi = rho < 4 * K ? (ax == 0 ? sp.maxs[0] : ax == 1 ? sp.maxs[1] : sp.maxs[2]) : (rho < 7 * K ? sp.v_max : sp.a_max);      // (selected, not indexed: a kernel argument indexed by a variable is copied to scratch memory)
        const double lo = rho < 4 * K ? (ax == 0 ? sp.mins[0] : ax == 1 ? sp.mins[1] : sp.mins[2]) : (rho < 7 * K ? -sp.v_max : -sp.a_max);
        double v = sOff[rho][ax];
        for (int c = 0; c < kNZ; c++) v += sB[rho][c] * z[ax * kNZ + c];
        rhs = side ? -lo : hi;
        return side ? v - lo : hi - v;
      }
      const int q = r - 6 * R, l = q >> 2, k = q & 3;
      int i = 0;
      for (int j = 1; j < NEP_MAX_POL; j++) i += (l >= sCnt[j]) ? 1 : 0;
      double n1, n2, d; line_nd(l, i, n1, n2, d);
      const int rho = 4 * i + k;
      double vx = sOff[rho][0], vy = sOff[rho][1];
      for (int c = 0; c < kNZ; c++) { vx += sB[rho][c] * z[c]; vy += sB[rho][c] * z[kNZ + c]; }
      rhs = 1.0 - d;
      return rhs - (n1 * vx + n2 * vy);
    };
    auto row_vec = [&](int r, int slot_a, double& h) {      // the row in z-space, a.z <= h, written to sA[slot_a] (LDS: a private array indexed by a variable would live in scratch memory)
      for (int c = 0; c < PN; c++) sA[slot_a][c] = 0.0;
      if (r < 6 * R) {
        const int side = r & 1, q = r >> 1, ax = (q >= R ? 1 : 0) + (q >= 2 * R ? 1 : 0), rho = q - ax * R;      // (no integer division: forty instructions on the VALU)
        const double hi = rho < 4 * K ? (ax == 0 ? sp.maxs[0] : ax == 1 ? sp.maxs[1] : sp.maxs[2]) : (rho < 7 * K ? sp.v_max : sp.a_max);      // (selected, not indexed: a kernel argument indexed by a variable is copied to scratch memory)
        const double lo = rho < 4 * K ? (ax == 0 ? sp.mins[0] : ax == 1 ? sp.mins[1] : sp.mins[2]) : (rho < 7 * K ? -sp.v_max : -sp.a_max);
        for (int c = 0; c < kNZ; c++) sA[slot_a][ax * kNZ + c] = side ? -sB[rho][c] : sB[rho][c];
        h = side ? sOff[rho][ax] - lo : hi - sOff[rho][ax];
        return;
      }
      const int q = r - 6 * R, l = q >> 2, k = q & 3;
      int i = 0;
      for (int j = 1; j < NEP_MAX_POL; j++) i += (l >= sCnt[j]) ? 1 : 0;
      double n1, n2, d; line_nd(l, i, n1, n2, d);
      const int rho = 4 * i + k;
      for (int c = 0; c < kNZ; c++) { sA[slot_a][c] = n1 * sB[rho][c]; sA[slot_a][kNZ + c] = n2 * sB[rho][c]; }
      h = (1.0 - d) - (n1 * sOff[rho][0] + n2 * sOff[rho][1]);
    };

    // ---- the active set at the solve's last iterate ----
    if (lane == 0) sI[0] = 0;
    __syncthreads();
    double vmax0 = 0.0;                            // the start point's worst violation (see below)
    for (int r0 = 0; r0 < n_rows; r0 += NT) {      // (in row order: ballots inside a wave, the waves' counts in front of each other)
      const int r = r0 + lane, wv = lane >> 6, ln = lane & 63;
      bool act = false;
      if (r < n_rows) { double rhs; const double s = row_eval(r, sZ, rhs); act = s < kActTol * (1.0 + fabs(rhs)); vmax0 = fmax(vmax0, -s / (1.0 + fabs(rhs))); }
      const unsigned long long m = __ballot(act);
      if (ln == 0) sI[4 + wv] = __popcll(m);
      __syncthreads();
      const int base = sI[0];
      int off = base; for (int w_ = 0; w_ < wv; w_++) off += sI[4 + w_];
      if (act) { const int p = off + __popcll(m & ((1ull << ln) - 1ull)); if (p < PA) sAct[p] = r; }
      __syncthreads();
      if (lane == 0) sI[0] = base + sI[4] + sI[5] + sI[6] + sI[7];
      __syncthreads();
    }
    int na = sI[0];
    if (na > PA) continue;                                     // (more candidate rows than the kernel carries: left as it is)
    {   // a start point that violates a row by more than 1e-4 (1 + |rhs|) is not "nearly there": an interior point that gives up on
        // a feasible problem has long driven the primal residual down; what is left are the infeasible problems, which no polish
        // can certify — the full number of rounds each was most of this kernel's time in the closed loop
      sRed[lane] = vmax0;
      __syncthreads();
      if (lane < 64) sRed[lane] = fmax(fmax(sRed[lane], sRed[lane + 64]), fmax(sRed[lane + 128], sRed[lane + 192]));
      __syncthreads();
      double vm = 0.0; for (int t = 0; t < 64; t++) vm = fmax(vm, sRed[t]);
      __syncthreads();
      if (vm > kStartTol) continue;
    }
    bool certified = false;
    int n_dropped = 0;                   // (rows that left with a negative multiplier: sDropped, LDS)
    for (int round = 0; round < kPolRounds; round++) {      // (a certificate takes one to three rounds; the bound is the oracle's)
      __syncthreads();
      // rows of the active set in z-space, H^-1 A', the Schur complement S = A H^-1 A' and its right-hand side A z0 - h
      if (lane < na) {
        double h;
        row_vec(sAct[lane], lane, h);
        double az0 = 0.0;
        for (int c = 0; c < PN; c++) az0 += sA[lane][c] * sZ0[c];
        for (int ax = 0; ax < 3; ax++) for (int e = 0; e < kNZ; e++) { double v = 0.0; for (int c = 0; c < kNZ; c++) v += sHi[e][c] * sA[lane][ax * kNZ + c]; sHA[lane][ax * kNZ + e] = v; }
        sRhs[lane] = az0 - h; sRowH[lane] = h;
      }
      __syncthreads();
      for (int e = lane; e < na * na; e += NT) { const int i = e / na, j = e - i * na; double v = 0.0; for (int c = 0; c < PN; c++) v += sA[i][c] * sHA[j][c]; sS[i][j] = v; }
      __syncthreads();
      // Cholesky, column by column; a row whose pivot vanishes is dependent on the rows before it: left out (nu = 0).  The small dense
      // algebra runs on the first wave alone, its steps separated by wave-level fences (LDS operations of one wave are in order): as
      // workgroup barriers the ~200 steps of a polish were half its time
      if (lane < 64) {
        auto wsync = [] { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup"); };
        if (lane < na) sDrop[lane] = 0;
        for (int j = 0; j < na; j++) {
          wsync();
          if (lane == 0) {
            double d = sS[j][j]; const double d0 = d;
            for (int k = 0; k < j; k++) d -= sS[j][k] * sS[j][k];
            if (!(d > kPivTol * d0) || !(d0 > 0.0)) { sDrop[j] = 1; for (int k = 0; k < j; k++) sS[j][k] = 0.0; sS[j][j] = 1.0; sRhs[j] = 0.0; }
            else sS[j][j] = sqrt(d);
          }
          wsync();
          if (lane > j && lane < na) {
            double v = 0.0;
            if (!sDrop[j]) { v = sS[lane][j]; for (int k = 0; k < j; k++) v -= sS[lane][k] * sS[j][k]; v /= sS[j][j]; }
            sS[lane][j] = v;
          }
        }
        wsync();
        // L y = rhs, L' nu = y, a column at a time with the lanes on the rows
        if (lane < na) sNu[lane] = sRhs[lane];
        for (int i = 0; i < na; i++) {
          wsync();
          const double yi = sNu[i] / sS[i][i];
          wsync();
          if (lane == i) sNu[i] = yi;
          else if (lane > i && lane < na) sNu[lane] -= sS[lane][i] * yi;
        }
        for (int i = na - 1; i >= 0; i--) {
          wsync();
          const double xi = sDrop[i] ? 0.0 : sNu[i] / sS[i][i];
          wsync();
          if (lane == i) sNu[i] = xi;
          else if (lane < i) sNu[lane] -= sS[i][lane] * xi;
        }
      }
      __syncthreads();
      if (lane < PN) { double v = sZ0[lane]; for (int i = 0; i < na; i++) v -= sHA[i][lane] * sNu[i]; sZ[lane] = v; }      // z = z0 - H^-1 A' nu
      __syncthreads();
      // multipliers: the rows with a negative one leave, all at once (one at a time a certificate took a round per weakly active row
      // of the start set; a row that is missed comes back through the violation test)
      {
        double numax = 0.0; for (int i = 0; i < na; i++) numax = fmax(numax, fabs(sNu[i]));
        const double wv = -kDualTol * (1.0 + numax);
        int n_neg = 0; for (int i = 0; i < na; i++) n_neg += (!sDrop[i] && sNu[i] < wv) ? 1 : 0;
        if (n_neg > 0) {
          __syncthreads();
          if (lane == 0) {
            int keep = 0, nd_ = n_dropped;
            for (int i = 0; i < na; i++) { if (!sDrop[i] && sNu[i] < wv) { if (nd_ < 12) sDropped[nd_++] = sAct[i]; } else sAct[keep++] = sAct[i]; }
          }
          n_dropped = n_dropped + n_neg < 12 ? n_dropped + n_neg : 12;
          na -= n_neg;
          continue;
        }
      }
      // every row at the new point: the most violated one enters
      double vmax = 0.0; int vrow = -1;
      for (int r = lane; r < n_rows; r += NT) { double rhs; const double s = row_eval(r, sZ, rhs); const double v = -s / (1.0 + fabs(rhs)); if (v > kFeasTol && v > vmax) { vmax = v; vrow = r; } }
      sRed[lane] = vmax; sRedI[lane] = vrow;
      __syncthreads();
      // (two levels: sixteen threads take sixteen entries each, then one takes their sixteen — on one thread the 256 entries were a
      // chain of dependent LDS reads, a quarter of a round; largest violation first, the lower row on a tie: the same whoever reduces)
      if (lane < 16) { double bv = 0.0; int br = -1; for (int t = lane * 16; t < lane * 16 + 16; t++) if (sRedI[t] >= 0 && (sRed[t] > bv || (sRed[t] == bv && sRedI[t] < br))) { bv = sRed[t]; br = sRedI[t]; } sRed[lane * 16] = bv; sRedI[lane * 16] = br; }
      __syncthreads();
      if (lane == 0) { double bv = 0.0; int br = -1; for (int t = 0; t < NT; t += 16) if (sRedI[t] >= 0 && (sRed[t] > bv || (sRed[t] == bv && sRedI[t] < br))) { bv = sRed[t]; br = sRedI[t]; } sI[1] = br; }
      __syncthreads();
      const int viol = sI[1];
      if (viol < 0) { certified = true; break; }
      bool have = false; for (int i = 0; i < na; i++) have = have || sAct[i] == viol;
      for (int i = 0; i < n_dropped; i++) have = have || sDropped[i] == viol;      // (a row that left with a negative multiplier comes back: the iteration would go round in circles)
      if (have || na >= PA) break;                              // (a row of the set still violated: dependent rows in conflict — no certificate)
      __syncthreads();
      if (lane == 0) { int pos = na; while (pos > 0 && sAct[pos - 1] > viol) { sAct[pos] = sAct[pos - 1]; pos--; } sAct[pos] = viol; }
      na++;
    }
    if (!certified) continue;
    // ---- the optimum: coefficients, objective, outputs — as the interior-point kernels write them ----
    __syncthreads();
    for (int t = lane; t < 12 * K; t += NT) {      // theta = Th z + ThU init
      const int ax = t / (4 * K), r = t - ax * 4 * K;
      double v = tb->ThU[r][0] * sInit[ax * 3] + tb->ThU[r][1] * sInit[ax * 3 + 1] + tb->ThU[r][2] * sInit[ax * 3 + 2];
      for (int c = 0; c < kNZ; c++) v += c < nz ? tb->Th[r][c] * sZ[ax * kNZ + c] : 0.0;
      sTheta[(ax * 8 + r / 4) * 4 + (r % 4)] = v;
    }
    for (int t = lane; t < 96; t += NT) if ((t % 32) / 4 >= K) sTheta[t] = 0.0;
    __syncthreads();
    if (ps.line_far && !ps.lines_override) {
      // Under the line presolve the rows above are the NEAR lines only (section 7 of DESIGN.md).  The certified point is the optimum of
      // the full problem if it also satisfies what the presolve set aside, tested as qp_reg_kernel tests its own solution: every parked
      // line at the trajectory's position control points, and — where LPs were skipped — every control point within cull_radius of the
      // guess's.  A point that fails either is dropped (the slot keeps the interior point's verified result).
      if (lane == 0) sI[2] = 0;
      __syncthreads();
      if (lane == 0) {      // (the parked lines' counts, as prefix sums: sDrop's storage — the rounds are over)
        int o = 0, sk = 0;
        for (int i = 0; i < NEP_MAX_POL; i++) { sDrop[i] = o; if (i < K) { o += ps.line_far[(long)slot * NEP_MAX_POL + i]; sk += ps.line_skip ? ps.line_skip[(long)slot * NEP_MAX_POL + i] : 0; } }
        sDrop[NEP_MAX_POL] = o; sDrop[NEP_MAX_POL + 1] = sk;
      }
      __syncthreads();
      const int n_far = sDrop[NEP_MAX_POL], n_skip = sDrop[NEP_MAX_POL + 1];
      double* sCp = sRed;      // [4 K][2]
      if (lane < 8 * K) {
        const int rho = lane >> 1, ax = lane & 1, sg = rho >> 2, k = rho & 3;
        const double c0 = (T * T * T) * cPolAPosInv[0][k], c1 = (T * T) * cPolAPosInv[1][k], c2 = T * cPolAPosInv[2][k], c3 = cPolAPosInv[3][k];
        const double* Q = sTheta + (ax * 8 + sg) * 4;
        const double v = ((Q[0] * c0 + Q[1] * c1) + Q[2] * c2) + Q[3] * c3;
        sCp[rho * 2 + ax] = v;
        if (n_skip > 0) {
          const double* P = sCoef + (ax * 8 + sg) * 4;
          const double gq = ((P[0] * c0 + P[1] * c1) + P[2] * c2) + P[3] * c3;
          double d2 = (v - gq) * (v - gq);
          d2 += __shfl_xor(d2, 1);                        // (x and y of a control point sit on neighbouring lanes)
          if (d2 > sp.cull_radius * sp.cull_radius) sI[2] = 1;
        }
      }
      __syncthreads();
      bool bad = false;
      for (int e = lane; e < n_far; e += NT) {
        int i = 0;
        for (int j = 1; j < NEP_MAX_POL; j++) i += (e >= sDrop[j]) ? 1 : 0;
        const double* nd = bucket0 + ((long)i * sp.lines_cap + ((long)sp.lines_cap - 1 - (e - sDrop[i]))) * 3;
        for (int k = 0; k < 4; k++) bad = bad || (nd[0] * sCp[(4 * i + k) * 2] + nd[1] * sCp[(4 * i + k) * 2 + 1] + nd[2] - 1.0 > 0.0);
      }
      if (bad) sI[2] = 1;
      __syncthreads();
      const bool refuse = sI[2] != 0;
      __syncthreads();
      if (refuse) continue;
    }
    double obj = 0.0;
    if (lane == 0) {      // the reference's objective on the returned coefficients (:322-383; relaxed: :838-861)
      for (int ax = 0; ax < 3; ax++) {
        for (int i = 0; i < K; i++) { const double a = sTheta[(ax * 8 + i) * 4]; obj += 36.0 * T * a * a; }
        const double* c = sTheta + (ax * 8 + (K - 1)) * 4;
        const double pe = ((T * T * T) * c[0] + (T * T) * c[1] + T * c[2]) + c[3], ve = (3 * T * T) * c[0] + (2 * T) * c[1] + c[2], ae = (6 * T) * c[0] + 2 * c[1];
        obj += sp.weight * (pe - sFin[ax]) * (pe - sFin[ax]);
        if (mode == 1) obj += sp.weight * (ve * ve + ae * ae);
      }
    }
    const double dix = sCoef[3] - sFin[0], diy = sCoef[32 + 3] - sFin[1];
    if (sqrt(dix * dix + diy * diy) < 1.0) { if (lane < 32) sTheta[64 + lane] = sCoef[64 + lane]; }      // :879-880
    __syncthreads();
    for (int t = lane; t < 96; t += NT) (&sol->coeff[0][0][0])[t] = sTheta[t];
    if (lane <= NEP_MAX_POL) sol->times[lane] = (lane <= K) ? g->t_start + lane * T : 0.0;
    const int ns_all = sched.n[K];
    const int ns = ns_all < sp.max_states ? ns_all : sp.max_states;
    if (lane == 0) {
      sol->stats.status = mode; sol->stats.objective = obj; sol->K = K; sol->n_states = ns;
    }
    if (ps.states) {
      for (int s = lane; s < ns; s += NT) {      // generatePwpOut's samples (:911-934)
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
    if (ps.commit) {      // the record the agent publishes (neptune_ros.cpp:434-480), as the interior-point kernels write it
      nep_traj_rec* cr = ps.commit + slot;
      const int own = sp.first_local + (slot % sp.n_local);
      if (lane == 0) {
        cr->id = own + 1; cr->is_agent = 1; cr->n_bend = 1; cr->valid = 1;
        for (int a = 0; a < 3; a++) { cr->bbox[a] = 2 * sp.drone_radius; cr->pos[a] = sTheta[(a * 8) * 4 + 3]; }
        cr->bend[0][0] = ps.pb[2 * own]; cr->bend[0][1] = ps.pb[2 * own + 1];
        cr->pwp.n_seg = K;
      }
      if (lane <= NEP_TRAJ_MAX_SEG) cr->pwp.times[lane] = (lane <= K) ? g->t_start + lane * T : 0.0;
      for (int e = lane; e < 3 * NEP_TRAJ_MAX_SEG * 4; e += NT) {
        const int ax = e / (NEP_TRAJ_MAX_SEG * 4), r = e % (NEP_TRAJ_MAX_SEG * 4), seg = r / 4, j = r % 4;
        (&cr->pwp.coeff[0][0][0])[e] = (seg < K) ? sTheta[(ax * 8 + seg) * 4 + j] : 0.0;
      }
    }
    return true;      // (a certified first problem is the answer: the relaxed one is not looked at)
  }
  return false;
}

// ps.polish_count: [0] slots listed by the QP kernel(s) of this launch sequence, [3] how many of them this kernel certified.  They are
// zeroed at the START of the next launch sequence — by order_kernel on its way, else by qp_polish_zero_kernel — so that a step
// has no memset node, no kernel behind this one and no "last workgroup done" counter (512 atomics on one address were 0.08 ms of
// every step, listed slots or not); until then the host reads them (nep_batch_debug_polish_count).
__global__ __launch_bounds__(256) void qp_polish_kernel(SceneParams sp, ProblemSet ps, const QpTable* __restrict__ tables, SampleSched sched) {
  __builtin_amdgcn_s_setprio(3);      // (latency-bound waves: when another scene group's hull / separator waves share the SIMD — bench.py's pipelined groups — these issue first)
  const int n_listed = ps.polish_count[0];
  int n_ok = 0;
  for (int e = blockIdx.x; e < n_listed; e += gridDim.x) {
    const int slot = ps.polish_list[e];
    const long long t0 = (long long)wall_clock64();
    const bool ok = polish_slot(sp, ps, tables, sched, slot);
    __syncthreads();
    // (round-5 advisor finding) the pass's device time belongs to the replan's: nep_stats.solve_us and the next launch's ordering key count
    // it, and a slot the pass certified is marked (bit 8 of its flag word: nep_batch_debug_polish_flags) — its status, objective and
    // trajectory are the pass's, its iteration counts still the interior point's
    if (threadIdx.x == 0) {
      nep_solution* sol = ps.solution + slot;
      const double us_ = sol->stats.solve_us + (double)((long long)wall_clock64() - t0) * sp.us_per_tick;
      sol->stats.solve_us = us_;
      if (ps.order_key) { const double k_ = us_ * 0.125; const int kn = k_ > 63.0 ? 63 : (int)k_; if (kn > ps.order_key[slot]) ps.order_key[slot] = kn; }
      if (ok) ps.polish_flag[slot] |= 0x100;
    }
    n_ok += ok ? 1 : 0;
    __syncthreads();
  }
  if (threadIdx.x == 0 && n_ok) atomicAdd(ps.polish_count + 3, n_ok);
}
__global__ void qp_polish_zero_kernel(int* __restrict__ c) { if (threadIdx.x < 4) c[threadIdx.x] = 0; }
void launch_qp_polish_zero(int* counters, hipStream_t st) { if (counters) hipLaunchKernelGGL(qp_polish_zero_kernel, dim3(1), dim3(64), 0, st, counters); }

// a fixed small grid walks the list (it holds a per cent of the slots at most)
void launch_qp_polish(int n_slots, const SceneParams& sp, const ProblemSet& ps, const QpTable* tables, const SampleSched& sched, hipStream_t st) {
  if (n_slots <= 0 || !ps.polish_list) return;
  const int g_env = g_debug.polish_grid > 0 ? g_debug.polish_grid : 256;      // (A/B: nep_debug_set_global_option "polish_grid")
  hipLaunchKernelGGL(qp_polish_kernel, dim3(n_slots < g_env ? n_slots : g_env), dim3(256), 0, st, sp, ps, tables, sched);
}

}  // namespace nep
