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
// Work split per iteration (3 row passes, 8 workgroup barriers):
//   * thread t < 3R owns the two box rows of (rho, axis), state in registers;
//   * every thread owns slice t&7 of the lines of control point (seg,k) = (t>>5, (t>>3)&3): the 8
//     slices of one control point sit in consecutive lanes, so the scatter onto base rows is three
//     xor-shuffles, not atomics; line-row state (s, lambda) is staged in LDS;
//   * all threads assemble the n x n normal matrix from the 4x64 base-row weights;
//   * wave 0 factors it in registers (lane i = row i, pivots broadcast with v_readlane) and runs
//     the forward/backward substitutions without a barrier.
// The LDS carve is sized for the expected line count so that two workgroups share a CU; rows
// beyond it spill to a global scratch (correct, slower).
#include <hip/hip_runtime.h>

#include "qp_common.h"

namespace nep {

size_t qp_lds_fixed_bytes() { return (size_t)kFixedDoubles * sizeof(double) + 64 * sizeof(int); }

// CULL: the separator parked far lines at the back of the buckets (nep_batch_set_line_cull) — solve with the near ones,
// verify the far ones, solve again with all of them if one is violated.  !CULL: every line, one solve.
template <bool CULL>
__global__ __launch_bounds__(BS, 2) void qp_kernel(SceneParams sp, ProblemSet ps, const QpTable* __restrict__ tables, SampleSched sched) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
  double* sB = smem + oB; double* sOff = smem + oOff; double* sAccL = smem + oAccL;
  double* sDc = smem + oDc; double* sTc = smem + oTc;
  double* sM = smem + oM; double* sHax = smem + oHax;
  double* sZ = smem + oZ; double* sG = smem + oG; double* sRd = smem + oRd; double* sRhs = smem + oRhs;
  double* sDxa = smem + oDxa; double* sDx = smem + oDx; double* sGq = smem + oGq; double* sZl = smem + oZl;
  double* sEp = smem + oEp; double* sCoef = smem + oCoef; double* sTheta = smem + oTheta;
  double* sInit = smem + oInit; double* sc = smem + oScal; double* sRed = smem + oRed;
  int* sI = (int*)(smem + kFixedDoubles);      // [0..8] line offsets, [16..] flags
  // (the dynamic part — line coefficients, then line-row state — starts at smem + kFixedDoubles + 32: ldyn below)

  const int tid = threadIdx.x;
  const int slot = ps.order ? ps.order[blockIdx.x] : (int)blockIdx.x;   // (launch order: see order_kernel)
  const long long t_wg0 = (long long)wall_clock64();          // this workgroup's lifetime goes to stats.solve_us (wall-clock ticks: sp.us_per_tick)
  const nep_guess* __restrict__ g = ps.guess + slot;
  // A guess without segments (front-end miss: K = 0) or with more than the handle plans for cannot be solved: such a slot
  // reports NEP_FAILED with an empty solution and publishes nothing (neptune_ros.cpp:651-663); K_ok guards every use of K.
  const int K_in = g->K;
  const bool K_ok = K_in >= 1 && K_in <= NEP_MAX_POL && K_in <= sp.num_pol;
  const int K = K_ok ? K_in : 1;
  const double T = sp.T_span, wgt = sp.weight;
  nep_solution* __restrict__ sol = ps.solution + slot;

  // ---- stage the guess ------------------------------------------------------------------------
  if (tid < 96) sCoef[tid] = (&g->coeff[0][0][0])[tid];
  // base rows >= R are never written: start them (and every accumulator) from zero rather than from stale LDS
  for (int e = tid; e < kMaxR * 6; e += BS) sTc[e] = 0.0;
  sDc[tid] = 0.0; sAccL[tid] = 0.0;
  // Presolved-away ("far") lines sit at the back of the buckets (separator_kernel); the first attempt solves with the
  // near lines only and checks the far ones against its solution; a violated one means all lines are solved for.
  const bool culled = CULL;
  int status = NEP_FAILED, iters_total = 0, iters_first = 0, L_used = 0, L_all = 0;
  double objective = 0.0;
  bool has_qc = false, z_override = false;
  // phase cycle counters: compiled in only with -DNEP_PROFILE_PHASES (they hold 26 VGPRs otherwise)
#ifdef NEP_PROFILE_PHASES
  long long tph[12] = {0,0,0,0,0,0,0,0,0,0,0,0}; long long tlast = 0;
  const bool prof = ps.dbg != nullptr;
  const long long tstart = prof ? clock64() : 0;
#define TICK(k) do { if (prof) { const long long t_ = clock64(); tph[k] += t_ - tlast; tlast = t_; } } while (0)
  long long tset[3] = {0, 0, 0}; long long tmark = tstart;
#define SETUP_TICK(k) do { if (prof) { const long long t_ = clock64(); tset[k] += t_ - tmark; tmark = t_; } } while (0)
#else
#define TICK(k) do { } while (0)
#define SETUP_TICK(k) do { } while (0)
#endif
  auto solve_once = [&](const bool use_far) -> bool {   // returns true when the far lines must be added
  __syncthreads();
  if (tid < NEP_MAX_POL) {   // one load per lane instead of a serial chain of global round trips
    sI[44 + tid] = (tid < K) ? ps.line_cnt[(long)slot * NEP_MAX_POL + tid] : 0;
    sI[32 + tid] = (tid < K && culled) ? ps.line_far[(long)slot * NEP_MAX_POL + tid] : 0;
  }
  __syncthreads();
  if (tid == 0) {
    int o = 0, nf = 0, all = 0, ovf = 0;
    for (int i = 0; i < NEP_MAX_POL; i++) {
      const int cn_raw = sI[44 + i], cf = sI[32 + i];
      ovf |= cn_raw < 0 ? 1 : 0;                                    // the segment's bucket overflowed: lines are missing (separator_body)
      const int cn = cn_raw < 0 ? -1 - cn_raw : cn_raw;
      sI[44 + i] = cn;
      sI[i] = o; sI[52 + i] = nf;
      o += cn + (use_far ? cf : 0); nf += cf; all += cn + cf;
    }
    sI[NEP_MAX_POL] = o; sI[41] = nf; sI[42] = all; sI[21] = 0; sI[27] = ovf;
  }
  __syncthreads();
  if (sI[27] != 0) return false;      // (a replan whose line bucket overflowed is not solved: it fails and keeps its previous trajectory — see qp_reg_kernel)
  const int L = sI[NEP_MAX_POL];
  L_used = L; L_all = sI[42];
  // Line coefficients and line-row state live in the LDS carve when the replan's L lines fit it
  // (the normal case: ds_read/ds_write through address_space(3) pointers), else in the per-slot
  // global spill; the solver body below is instantiated once per placement.
  const int LL = ps.lds_lines + 2;            // strides; the last two entries are a dummy line (tail of the 4-wide row groups)
  const int GL = ps.rows_cap / 4 + 2;
  double* gsp = ps.row_scratch ? ps.row_scratch + (long)slot * (11L * GL) : nullptr;
  typedef __attribute__((address_space(3))) double* lds_ptr;
  // the dynamic LDS region starts right after the (empty) static group segment
  const unsigned lds0 = __builtin_amdgcn_groupstaticsize();
  const lds_ptr ldyn = (lds_ptr)(unsigned int)(lds0 + (kFixedDoubles + 32) * sizeof(double));
  if (tid < 9) {
    if (tid < 3) {
      const double* c = sCoef + (tid * 8 + (K - 1)) * 4;
      sc[sFinal0 + tid] = ((T * T * T) * c[0] + (T * T) * c[1] + T * c[2]) + c[3];   // final_pos_ (:226-228)
    }
    sInit[tid] = sCoef[((tid / 3) * 8 + 0) * 4 + 1 + (tid % 3)];                    // b0,c0,d0 (:390-396)
  }
  __syncthreads();
  const double dix = sCoef[3] - sc[sFinal0], diy = sCoef[32 + 3] - sc[sFinal1], diz = sCoef[64 + 3] - sc[sFinal2];
  has_qc = sqrt(dix * dix + diy * diy + diz * diz) < 1.0;   // :697-702
  z_override = sqrt(dix * dix + diy * diy) < 1.0;           // :879-880
  const int mt = 48 * K + 4 * L + (has_qc ? 1 : 0);                    // inequality count (+ball)
  const double inv_mt = 1.0 / (double)mt;   // (see qp_reg_kernel: the iteration's scalar divisions)

  // ---- thread roles ---------------------------------------------------------------------------
  const int R = 8 * K;
  const bool has_box = tid < 3 * R;
  const int bax = has_box ? tid / R : 0, brho = has_box ? tid % R : 0;
  const double bhi = brho < 4 * K ? sp.maxs[bax] : (brho < 7 * K ? sp.v_max : sp.a_max);
  const double blo = brho < 4 * K ? sp.mins[bax] : (brho < 7 * K ? -sp.v_max : -sp.a_max);
  const int pair = tid >> 3, li = pair >> 2, lk = pair & 3, slice = tid & 7;
  const bool has_line = li < K;
  const int lrho = 4 * li + lk;
  const int lbeg = has_line ? sI[li] : 0, lend = has_line ? sI[li + 1] : 0;
  // box row state: [0] upper (alpha=+e), [1] lower (alpha=-e)
  double bs0 = 1, bl0 = 0, bs1 = 1, bl1 = 0;

  status = NEP_FAILED; iters_total = 0; iters_first = 0; objective = 0.0;

  auto run = [&](auto lds_tag, auto ll_tag) {
  constexpr bool LDSL = decltype(lds_tag)::value;
  constexpr int LLC = decltype(ll_tag)::value;
  const int LLe = LLC > 0 ? LLC : LL;         // (a constant in the common instantiation)
  // c: 0 n1, 1 n2, 2 h            (line coefficient)
  auto LNr = [&](int l, int c) -> double { if constexpr (LDSL) return ldyn[c * LLe + l]; else return gsp[c * GL + l]; };
  auto LNw = [&](int l, int c, double v) { if constexpr (LDSL) ldyn[c * LLe + l] = v; else gsp[c * GL + l] = v; };
  // c: 0 s, 1 lambda; k: control point of the segment
  auto STr = [&](int l, int k, int c) -> double { if constexpr (LDSL) return ldyn[(3 + c * 4 + k) * LLe + l]; else return gsp[(3 + c * 4 + k) * GL + l]; };
  auto STw = [&](int l, int k, int c, double v) { if constexpr (LDSL) ldyn[(3 + c * 4 + k) * LLe + l] = v; else gsp[(3 + c * 4 + k) * GL + l] = v; };
  // gather the separator's buckets into one segment-major list (one flat loop: a loop per segment costs a global
  // round trip each)
  for (int e = tid; e < L; e += BS) {
    int i = 0;
#pragma unroll
    for (int j = 1; j < NEP_MAX_POL; j++) i += (e >= sI[j]) ? 1 : 0;
    const int l = e - sI[i], cn = sI[44 + i];
    const double* src = ps.line_nd + ((long)slot * NEP_MAX_POL + i) * sp.lines_cap * 3;
    const long q = l < cn ? l : (long)sp.lines_cap - 1 - (l - cn);      // near lines from the front, far ones from the back
    LNw(e, 0, src[3 * q]); LNw(e, 1, src[3 * q + 1]); LNw(e, 2, 1.0 - src[3 * q + 2]);
  }
  const int LD = (LDSL ? LLe : GL) - 1;   // dummy line: harmless operands for the padded tail of a row group
  if (tid < 4) { STw(LD, tid, 0, 1.0); STw(LD, tid, 1, 1.0); if (tid == 0) { LNw(LD, 0, 0.0); LNw(LD, 1, 0.0); LNw(LD, 2, 1.0); } }
  __syncthreads();
  // This thread's line rows, four independent rows at a time (loads first, then the four bodies:
  // the dependent fp64 chains of different rows interleave).  body(valid, l, n1, n2, h, s, lam).
  auto for_lines4 = [&](auto&& body) {
    if (!has_line) return;
    for (int l0 = lbeg + slice; l0 < lend; l0 += 32) {
      int l[4]; bool v[4]; double n1[4], n2[4], h[4], s[4], lam[4];
#pragma unroll
      for (int u = 0; u < 4; u++) {
        l[u] = l0 + 8 * u; v[u] = l[u] < lend; if (!v[u]) l[u] = LD;
        n1[u] = LNr(l[u], 0); n2[u] = LNr(l[u], 1); h[u] = LNr(l[u], 2); s[u] = STr(l[u], lk, 0); lam[u] = STr(l[u], lk, 1);
      }
#pragma unroll
      for (int u = 0; u < 4; u++) body(v[u], l[u], n1[u], n2[u], h[u], s[u], lam[u]);
    }
  };
  SETUP_TICK(0);
  for (int mode = 0; mode < 2; mode++) {
#ifdef NEP_PROFILE_PHASES
    if (prof) tmark = clock64();
#endif
    const QpTable* __restrict__ tb = tables + mode * (kMaxK + 1) + K;
    const int nz = mode == 1 ? K : (K > 2 ? K - 2 : 0), n = 3 * nz;   // = tb->nz (nep_tables.h), without the round trip
    __syncthreads();
    for (int e = tid; e < kMaxR * kNZ; e += BS) sB[(e / kNZ) * SBS + (e % kNZ)] = (&tb->B[0][0])[e];
    if (tid < 64) sHax[tid] = (&tb->Hax[0][0])[tid];
    if (tid < 8) sEp[tid] = tb->ep[tid];
    // Gi .. res_u are contiguous in QpTable: one coalesced copy into the normal matrix's space (free until the first
    // assembly) instead of dependent global loads inside the start-point loops
    for (int e = tid; e < kSmallTab; e += BS) sM[e] = (&tb->Gi[0][0])[e];
    // guess minus the init part of theta (sTheta is free until the result is written): what the start point projects
    if (tid < 96) { const int ax = tid >> 5, r = tid & 31; sTheta[tid] = r < 4 * K ? sCoef[tid] - (tb->ThU[r][0] * sInit[ax * 3] + tb->ThU[r][1] * sInit[ax * 3 + 1] + tb->ThU[r][2] * sInit[ax * 3 + 2]) : 0.0; }
    if (tid < 3 * R) { const int rho = tid % R, ax = tid / R; sOff[rho * 3 + ax] = tb->U[rho][0] * sInit[ax * 3] + tb->U[rho][1] * sInit[ax * 3 + 1] + tb->U[rho][2] * sInit[ax * 3 + 2]; }
    __syncthreads();
    const double* tGiP = sM + tGi; const double* tUpP = sM + tUp; const double* tUvP = sM + tUv; const double* tUaP = sM + tUa;
    const double* tZpP = sM + tZp; const double* tPpP = sM + tPp; const double* tResP = sM + tResU;
    // a_r of the least-squares point per axis (sRhs is free here) and the init part of the end-position error
    if (tid < 24) { const int ax = tid >> 3, r = tid & 7; sRhs[tid] = tPpP[r * 3] * sInit[ax * 3] + tPpP[r * 3 + 1] * sInit[ax * 3 + 1] + tPpP[r * 3 + 2] * sInit[ax * 3 + 2]; }
    else if (tid >= 32 && tid < 35) { const int ax = tid - 32; sc[sPe0 + ax] = tUpP[0] * sInit[ax * 3] + tUpP[1] * sInit[ax * 3 + 1] + tUpP[2] * sInit[ax * 3 + 2] - sc[sFinal0 + ax]; }
    __syncthreads();
    // Normal-matrix entries owned by this thread, decoded once per mode.  Only the blocks that can be
    // non-zero are summed — xx, yx, yy, zz — each entry by a pair of adjacent lanes (even / odd base
    // rows); the z-x and z-y blocks only ever hold the ball constraint's rank-one term.
    int me_i[2], me_j[2], me_ci[2], me_cj[2], me_sel[2]; bool me_on[2];
    const int tri_n = nz * (nz + 1) / 2, n_ent = 3 * tri_n + nz * nz;
    {
      auto tri = [&](int e, int& r, int& c) { r = (int)((sqrt(8.0 * e + 1.0) - 1.0) * 0.5); while ((r + 1) * (r + 2) / 2 <= e) r++; while (r * (r + 1) / 2 > e) r--; c = e - r * (r + 1) / 2; };
#pragma unroll
      for (int u = 0; u < 2; u++) {
        const int e = (tid >> 1) + u * (BS / 2);
        me_on[u] = e < n_ent && nz > 0;
        int ci = 0, cj = 0, bi = 0, bj = 0, sel = 0;
        if (me_on[u]) {
          if (e < tri_n) { tri(e, ci, cj); sel = 0; }
          else if (e < tri_n + nz * nz) { const int f = e - tri_n; ci = f / nz; cj = f % nz; bi = 1; sel = 1; }
          else if (e < 2 * tri_n + nz * nz) { tri(e - tri_n - nz * nz, ci, cj); bi = 1; bj = 1; sel = 2; }
          else { tri(e - 2 * tri_n - nz * nz, ci, cj); bi = 2; bj = 2; sel = 3; }
        }
        me_ci[u] = ci; me_cj[u] = cj; me_i[u] = bi * nz + ci; me_j[u] = bj * nz + cj; me_sel[u] = sel;
      }
    }
    bool converged = false, have_loose = false;      // (have_loose: see the snapshot below)
    int it = 0;
    const long long t_solve0 = (long long)wall_clock64();   // m_.optimize() starts here: every solve (first and relaxed) has its own TimeLimit
    SETUP_TICK(1);
    if (nz == 0) {
      // ---- K <= 2 with the terminal rows: a single point, feasible or not (tolerance 1e-6) ------
      double viol = 0.0, d0 = 0, d1 = 0, d2 = 0;
      if (has_box) { const double a = sOff[brho * 3 + bax]; viol = fmax(a - bhi, blo - a); }
      { const double ox = sOff[lrho * 3], oy = sOff[lrho * 3 + 1]; for_lines4([&](bool v, int, double n1, double n2, double h, double, double) { viol = fmax(viol, v ? n1 * ox + n2 * oy - h : 0.0); }); }
      if (tid < 6) {  // terminal v = a = 0 must hold at the least-squares point
        const int ax = tid / 2, e = tid % 2;
        viol = fmax(viol, fabs(tResP[e * 3] * sInit[ax * 3] + tResP[e * 3 + 1] * sInit[ax * 3 + 1] + tResP[e * 3 + 2] * sInit[ax * 3 + 2]));
      }
      if (tid == 0 && has_qc) {
        double c = -0.10 * 0.10;
        for (int ax = 0; ax < 3; ax++) { const double pe = sc[sPe0 + ax]; c += pe * pe; }
        viol = fmax(viol, c);
      }
      block_reduce4(viol, d0, d1, d2, sRed);
      converged = viol <= 1e-6;
      if (tid == 0) {
        double o = 0;
        for (int ax = 0; ax < 3; ax++) {
          for (int r = 0; r < K; r++) { const double a = sRhs[ax * 8 + r]; o += 36 * T * a * a; }
          const double pe = sc[sPe0 + ax];
          o += wgt * pe * pe;
        }
        sc[sObj] = o;
      }
    } else {
      // ---- start point: projection of the guess, floored slacks, centred duals ------------------
      if (tid < n) {
        const int ax = tid / nz, c = tid % nz;
        double z = 0;
        // orthogonal projection of the guess's coefficients onto {Th z + ThU init} (rows >= 4K of Zp and of the difference are zero)
#pragma unroll 8
        for (int r = 0; r < 4 * kMaxK; r++) z = __builtin_fma(tZpP[c * 4 * kMaxK + r], sTheta[ax * 32 + r], z);
        sZ[tid] = z;
        sG[tid] = (tGiP[c * 3] * sInit[ax * 3] + tGiP[c * 3 + 1] * sInit[ax * 3 + 1] + tGiP[c * 3 + 2] * sInit[ax * 3 + 2]) - 2 * wgt * sEp[c] * sc[sFinal0 + ax];
      }
      if (tid == 0) {
        double o = 0;
        for (int ax = 0; ax < 3; ax++) {
          for (int r = 0; r < K; r++) { const double a = sRhs[ax * 8 + r]; o += 36 * T * a * a; }
          const double pe = sc[sPe0 + ax];
          o += wgt * pe * pe;
          if (mode == 1) {
            const double ve = tUvP[0] * sInit[ax * 3] + tUvP[1] * sInit[ax * 3 + 1] + tUvP[2] * sInit[ax * 3 + 2];
            const double ae = tUaP[0] * sInit[ax * 3] + tUaP[1] * sInit[ax * 3 + 1] + tUaP[2] * sInit[ax * 3 + 2];
            o += wgt * (ve * ve + ae * ae);
          }
        }
        sc[sObj0] = o;
        sI[16] = 0;  // loose snapshot present
        sI[17] = 0;  // stall counter
        sI[22] = -1; // iteration of the first loose hit
        sI[23] = 0;  // this iteration repeats the previous one with its predictor discarded (kCorrMinStep)
      }
      __syncthreads();
      // each thread keeps the values of its own base rows (and their step projections) in registers
      auto proj = [&](int rho, int ax, const double* vec) {   // fixed trip (loads in flight together); B's columns >= nz are zero
        double v = 0;
#pragma unroll
        for (int c = 0; c < kNZ; c++) v = __builtin_fma(sB[rho * SBS + c], vec[ax * nz + (c < nz ? c : nz - 1)], v);
        return v;
      };
      // Presolve (only with nep_batch_set_line_cull on): the minimiser of the cost without any inequality row, z* =
      // -Hax^-1 g per axis.  If every row holds there (and the terminal ball, if present), z* with zero multipliers
      // satisfies the KKT conditions of the full problem: it IS the optimum and no interior-point iteration is needed.
      // Far lines are then checked by the usual pass after the solve.  One extra row pass when the test fails.
      bool uncon = false;
      if constexpr (CULL) {
        if (tid < n) {
          const int ax = tid / nz, c = tid % nz;
          double v = 0;
          for (int e = 0; e < nz; e++) v -= sM[tHi + c * kNZ + e] * sG[ax * nz + e];
          sDx[tid] = v;
        }
        __syncthreads();
        double viol = -1.0, o_share = 0, d1 = 0, d2 = 0;
        if (has_box) { const double a = sOff[brho * 3 + bax] + proj(brho, bax, sDx); viol = fmax(a - bhi, blo - a); }
        {
          const double cx = has_line ? sOff[lrho * 3] + proj(lrho, 0, sDx) : 0.0, cy = has_line ? sOff[lrho * 3 + 1] + proj(lrho, 1, sDx) : 0.0;
          for_lines4([&](bool ok, int, double n1, double n2, double h, double, double) { viol = fmax(viol, ok ? (n1 * cx + n2 * cy) - h : -1.0); });
        }
        if (tid == BS - 1 && has_qc) {
          double c = -0.10 * 0.10;
          for (int ax = 0; ax < 3; ax++) { double pe = sc[sPe0 + ax]; for (int e = 0; e < nz; e++) pe += sEp[e] * sDx[ax * nz + e]; c += pe * pe; }
          viol = fmax(viol, c);
        }
        if (tid < n) {   // this coordinate's share of the objective at z*
          const int ax = tid / nz, c = tid % nz;
          double hz = 0;
          for (int e = 0; e < nz; e++) hz += sHax[c * kNZ + e] * sDx[ax * nz + e];
          o_share = sDx[tid] * (0.5 * hz + sG[tid]);
        }
        block_reduce4(viol, o_share, d1, d2, sRed);
        uncon = viol <= 0.0;
        if (uncon) { if (tid < n) sZ[tid] = sDx[tid]; if (tid == 0) sc[sObj] = sc[sObj0] + o_share; }
      }
      double cpb = has_box ? sOff[brho * 3 + bax] + proj(brho, bax, sZ) : 0.0, uab = 0.0, udb = 0.0;
      double cpx = has_line ? sOff[lrho * 3] + proj(lrho, 0, sZ) : 0.0, cpy = has_line ? sOff[lrho * 3 + 1] + proj(lrho, 1, sZ) : 0.0;
      double uax = 0.0, uay = 0.0, udx = 0.0, udy = 0.0;
      if (has_box) {
        const double a = cpb;
        double sl = bhi - a; bs0 = sl > kSlackFloor ? sl : kSlackFloor; bl0 = kMu0 * frcp(bs0);
        sl = a - blo; bs1 = sl > kSlackFloor ? sl : kSlackFloor; bl1 = kMu0 * frcp(bs1);
      }
      {
        const double cx = cpx, cy = cpy;
        for_lines4([&](bool, int l, double n1, double n2, double h, double, double) {
          const double sl = h - (n1 * cx + n2 * cy);
          const double s = sl > kSlackFloor ? sl : kSlackFloor;
          STw(l, lk, 0, s); STw(l, lk, 1, kMu0 * frcp(s));   // (the start duals need no exact division)
        });
      }
      if (tid == 0) {
        double qs = 1.0; for (int e = 0; e < n; e++) qs = fmax(qs, fabs(sG[e]));
        sc[sQscale] = qs;
        if (has_qc) {
          double c = -0.10 * 0.10;
          for (int ax = 0; ax < 3; ax++) { double pe = sc[sPe0 + ax]; for (int e = 0; e < nz; e++) pe += sEp[e] * sZ[ax * nz + e]; c += pe * pe; }
          const double sq = (-c > 1e-3) ? -c : 1e-3;
          sc[sSq] = sq; sc[sLq] = 1.0 / sq;
        } else { sc[sSq] = 1.0; sc[sLq] = 0.0; }
        sc[sDsq] = 0.0; sc[sDlq] = 0.0; sc[sDsqA] = 0.0; sc[sDlqA] = 0.0; sc[sRpq] = 0.0; sc[sWq] = 0.0;
      }
      __syncthreads();

      // Row direction of the previous solve, recomputed by the
      // merged update:  ds = -rp - gd ; dl = -rc/s + w (rp + gd), rc = s lam - sigma mu + dsa dla.
      double alpha_prev = 0.0, sm_prev = 0.0, sm_keep = 0.0;
      int n_nopred = 0;
      double* redA = sRed; double* redP2 = sRed + 16; double* redP5 = sRed + 32;
      const int n0 = has_qc ? n : 2 * nz;                       // wave 0 factors this block, wave 1 the z block
      const unsigned zoff_m = lds0 + (oM + 2 * nz * MS + 2 * nz) * 8, zoff_d = lds0 + (oInvD + 2 * nz) * 8;
      SETUP_TICK(2);
#ifdef NEP_PROFILE_PHASES
      if (prof) tph[10] -= clock64();     // [10]: cycles inside the iteration loops
#endif
      for (it = 0; it < kMaxIt && !uncon; it++) {
#ifdef NEP_PROFILE_PHASES
        if (prof) tlast = clock64();
#endif
        // ---- (A) apply the previous step, then residuals / weights / scatter onto base rows ----
        double bTl = 0, bD = 0, bT1 = 0;
        double lTx = 0, lTy = 0, lDxx = 0, lDxy = 0, lDyy = 0, l1x = 0, l1y = 0;
        double nrp = 0, sumsl = 0, dummy1 = 0;
        auto rowA = [&](double& s, double& lam, double a_old, double ga, double gd, double h, double& a_new_out) {
          const double rp0 = a_old + s - h, is = frcp(s), w0 = lam * is;
          const double dsa = -rp0 - ga, dla = -lam + w0 * (rp0 + ga);
          const double rcv = s * lam - sm_prev + dsa * dla;
          const double ds = -rp0 - gd, dl = -rcv * is + w0 * (rp0 + gd);
          s = __builtin_fma(alpha_prev, ds, s); lam = __builtin_fma(alpha_prev, dl, lam);
          a_new_out = __builtin_fma(alpha_prev, gd, a_old);
        };
        if (has_box) {
          double an;
          rowA(bs0, bl0, cpb, uab, udb, bhi, an);
          { const double rp = an + bs0 - bhi, w = bl0 * frcp(bs0), v = bl0 - w * rp; nrp = fmax(nrp, fabs(rp)); sumsl += bs0 * bl0; bTl += bl0; bD += w; bT1 += v; }
          rowA(bs1, bl1, -cpb, -uab, -udb, -blo, an);
          { const double rp = an + bs1 + blo, w = bl1 * frcp(bs1), v = bl1 - w * rp; nrp = fmax(nrp, fabs(rp)); sumsl += bs1 * bl1; bTl -= bl1; bD += w; bT1 -= v; }
        }
        for_lines4([&](bool ok, int l, double n1, double n2, double h, double s, double lam) {
          double an;
          rowA(s, lam, n1 * cpx + n2 * cpy, n1 * uax + n2 * uay, n1 * udx + n2 * udy, h, an);
          // The dummy line of a padded tail is (n1, n2, h, s, lambda) = (0, 0, 1, 1, 1): its activity and step are exactly zero,
          // so its slack stays 1, its residual 0 and everything it adds below is multiplied by n = 0 — only its multiplier
          // (which would drift) and its s*lambda need masking.
          STw(l, lk, 0, s); STw(l, lk, 1, ok ? lam : 1.0);
          const double rp = an + s - h, w = lam * frcp(s), v = lam - w * rp;
          nrp = fmax(nrp, fabs(rp)); sumsl += ok ? s * lam : 0.0;
          lTx += lam * n1; lTy += lam * n2; lDxx += w * n1 * n1; lDxy += w * n1 * n2; lDyy += w * n2 * n2; l1x += v * n1; l1y += v * n2;
        });
        cpb = __builtin_fma(alpha_prev, udb, cpb); cpx = __builtin_fma(alpha_prev, udx, cpx); cpy = __builtin_fma(alpha_prev, udy, cpy);   // base rows move with the step
        lTx = slice_sum(lTx); lTy = slice_sum(lTy); lDxx = slice_sum(lDxx); lDxy = slice_sum(lDxy); lDyy = slice_sum(lDyy); l1x = slice_sum(l1x); l1y = slice_sum(l1y);
        if (slice == 0) { double* o = sAccL + pair * 8; o[0] = lTx; o[1] = lTy; o[2] = lDxx; o[3] = lDxy; o[4] = lDyy; o[5] = l1x; o[6] = l1y; }
        if (has_box) { sTc[brho * 6 + bax] = bTl; sTc[brho * 6 + 3 + bax] = bT1; sDc[brho * 4 + (bax == 0 ? 0 : (bax == 1 ? 2 : 3))] = bD; }
        reduce_put<1>(nrp, sumsl, dummy1, redA);
        __syncthreads();                                                                       // barrier 1
        reduce_get<1>(nrp, sumsl, dummy1, redA);
        TICK(0);
        if (tid < R) {   // add the line sums onto the position rows (box threads wrote their part above)
          const int rho = tid; const double* al = sAccL + rho * 8;
          if (rho < 4 * K) {
            sDc[rho * 4 + 0] += al[2]; sDc[rho * 4 + 1] = al[3]; sDc[rho * 4 + 2] += al[4];
            sTc[rho * 6 + 0] += al[0]; sTc[rho * 6 + 1] += al[1]; sTc[rho * 6 + 3] += al[5]; sTc[rho * 6 + 4] += al[6];
          } else sDc[rho * 4 + 1] = 0.0;
        }
        // ---- ball constraint (scalar row, thread 255) --------------------------------------------
        if (tid == BS - 1) {
          double rpq = 0;
          if (has_qc) {
            // apply the previous step to the ball row (its slack moves linearly, infeasible-start)
            sc[sSq] += alpha_prev * sc[sDsq]; sc[sLq] += alpha_prev * sc[sDlq];
            double c = -0.10 * 0.10;
            for (int ax = 0; ax < 3; ax++) {
              double pe = sc[sPe0 + ax];
              for (int e = 0; e < nz; e++) pe += sEp[e] * sZ[ax * nz + e];
              c += pe * pe;
              for (int e = 0; e < nz; e++) sGq[ax * nz + e] = 2 * pe * sEp[e];
            }
            rpq = c + sc[sSq];
            sc[sWq] = sc[sLq] / sc[sSq];
          }
          sc[sRpq] = rpq;
          sc[sSumSl] = sumsl + (has_qc ? sc[sSq] * sc[sLq] : 0.0);
          sc[sMu] = sc[sSumSl] * inv_mt;
          sc[sNrp] = fmax(nrp, fabs(rpq));
        }
        __syncthreads();                                                                       // barrier 2
        TICK(1);
        // ---- dual residual + predictor rhs (8 partial sums per output), normal matrix -------------
        if (tid < 8 * n) {
          const int o = tid >> 3, sl8 = tid & 7, ax = o / nz, c = o % nz;
          double v = 0, t1 = 0;
          for (int rho = sl8; rho < R; rho += 8) { const double b = sB[rho * SBS + c]; v += b * sTc[rho * 6 + ax]; t1 += b * sTc[rho * 6 + 3 + ax]; }
          v = slice_sum(v); t1 = slice_sum(t1);
          if (sl8 == 0) {
            double hz = 0;
#pragma unroll
            for (int e = 0; e < kNZ; e++) hz += sHax[c * kNZ + e] * sZ[ax * nz + (e < nz ? e : nz - 1)];   // Hax's columns >= nz are zero
            const double zo = sZ[o], go = sG[o];
            v += go + hz;
            if (has_qc) v += sc[sLq] * sGq[o];
            sRd[o] = v; sRhs[o] = t1;
            sDxa[o] = zo * (0.5 * hz + go);       // this coordinate's share of the objective (sDxa is free here)
          }
        }
#pragma unroll
        for (int u = 0; u < 2; u++) {
          const int ci = me_ci[u], cj = me_cj[u], sel = me_sel[u], half = tid & 1;
          double a0 = 0.0, a1 = 0.0;
          if (me_on[u]) {
            // (Dxy is zero on the rows >= 4K, so every block can run over all R rows: uniform trip count)
            for (int q = 0; q < K; q++) {
              const int rho = half + 8 * q;
              double d[4], bi[4], bj[4];
#pragma unroll
              for (int w = 0; w < 4; w++) { d[w] = sDc[(rho + 2 * w) * 4 + sel]; bi[w] = sB[(rho + 2 * w) * SBS + ci]; bj[w] = sB[(rho + 2 * w) * SBS + cj]; }
              a0 = __builtin_fma(d[0] * bi[0], bj[0], a0); a1 = __builtin_fma(d[1] * bi[1], bj[1], a1);
              a0 = __builtin_fma(d[2] * bi[2], bj[2], a0); a1 = __builtin_fma(d[3] * bi[3], bj[3], a1);
            }
          }
          double v = a0 + a1;
          v += dpp<DPP_XOR1>(v);                      // the pair's two halves
          if (me_on[u] && half == 0) {
            if (sel != 1) v += sHax[ci * kNZ + cj];
            if (has_qc) { v += sc[sWq] * sGq[me_i[u]] * sGq[me_j[u]]; if (sel != 1) v += sc[sLq] * 2 * sEp[ci] * sEp[cj]; }
            sM[me_i[u] * MS + me_j[u]] = v;
          }
        }
        for (int e = tid; e < 2 * nz * nz; e += BS) {   // z-x and z-y blocks
          const int i = 2 * nz + e / (2 * nz), j = e % (2 * nz);
          sM[i * MS + j] = has_qc ? sc[sWq] * sGq[i] * sGq[j] : 0.0;
        }
        __syncthreads();                                                                       // barrier 3
        TICK(2);
        // ---- wave 0: convergence test, Cholesky of its block, predictor; wave 1: the z block -------
        if (tid < 64) {
          const double nrd = wave_max(tid < n ? fabs(sRd[tid]) : 0.0);
          const double o = sc[sObj0] + wave_sum(tid < n ? sDxa[tid] : 0.0);
          const double gap = sc[sMu] * mt, nr = sc[sNrp], qs = sc[sQscale];
          int flag = 0;
          if (nr <= sp.tol_res && nrd <= sp.tol_res * qs && gap <= sp.tol_gap * (1.0 + fabs(o))) flag = 1;      // (1e-10, 1e-10, 1e-11 unless nep_batch_set_tolerances says otherwise)
          else if (nr <= 1e-6 && nrd <= 1e-6 * qs && gap <= 1e-7 * (1.0 + fabs(o))) flag = 2;
          if (!(sc[sMu] < 1e30) || !(nrd < 1e300)) flag = 3;  // diverged / NaN
          if (sI[17] >= 3) flag = 3;                           // stalled
          if (sp.time_limit_ticks > 0 && (long long)wall_clock64() - t_solve0 > sp.time_limit_ticks) flag = 3;   // TimeLimit without an accepted iterate: "no solution" (:832-836)
          // Loosely converged iterates: keep the one closest to the strict tolerances (merit <= 1 is the strict test), and
          // stop three iterations after the first of them.  With mu that small the weights lambda/s amplify the rounding
          // of the row activities into the dual residual (floor ~1e-8 |g|): an iteration that has not passed the strict
          // test by then never will, and would idle to kMaxIt while its iterates get noisier.
          if (flag == 2 || (flag == 0 && sI[22] >= 0)) {
            const double merit = fmax(fmax(nr * sp.tol_res_inv, nrd / qs * sp.tol_res_inv), gap / (1.0 + fabs(o)) * sp.tol_gap_inv);
            const bool better = flag == 2 && (!sI[16] || merit < sc[sBestMerit]);
            const bool last = sI[22] >= 0 && it - sI[22] >= 3;
            if (sI[22] < 0 && tid == 0) sI[22] = it;
            flag = last ? (better ? 1 : 3) : (better ? 2 : 0);      // 3: leave the loop, the snapshot is the answer
            if (tid == 0 && flag == 2) sc[sBestMerit] = merit;
          }
          if (tid == 0) { sc[sObj] = o; if (flag == 2) sc[sObjLoose] = o; sI[18] = flag; }
#ifdef NEP_QP_ITERDBG
          if (tid == 0 && ps.dbg && slot == NEP_QP_ITERDBG && it < 60 && mode == 0) {   // development aid: convergence history of one slot
            long long* d = ps.dbg + 16 + it * 8;
            d[0] = __double_as_longlong(nr); d[1] = __double_as_longlong(nrd); d[2] = __double_as_longlong(qs); d[3] = __double_as_longlong(gap);
            d[4] = __double_as_longlong(o); d[5] = flag; d[6] = __double_as_longlong(alpha_prev); d[7] = __double_as_longlong(sm_prev);
          }
#endif
          TICK(3);
          if (flag != 1 && flag != 3) {                        // (uniform across the wave)
            const bool chol_ok = chol_wave(lds0 + oM * 8, lds0 + oInvD * 8, n0);
            if (tid == 0) sI[19] = chol_ok ? 1 : 0;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
            TICK(4);
            double b = 0;
            if (tid < n0) { b = -sRd[tid] + sRhs[tid]; if (has_qc) b += sGq[tid] * (sc[sLq] - sc[sWq] * sc[sRpq]); }   // rcq/sq = lq for the affine step
            b = solve_wave(lds0 + oM * 8, lds0 + oInvD * 8, n0, b);
            if (tid < n0) sDxa[tid] = b;
          }
        } else if (tid < 128) {
          bool chol_ok = true;
          if (!has_qc) {                                        // (runs also in the converging iteration: harmless)
            chol_ok = chol_wave(zoff_m, zoff_d, nz);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
            const int l1 = tid - 64;
            double b = l1 < nz ? -sRd[2 * nz + l1] + sRhs[2 * nz + l1] : 0.0;
            b = solve_wave(zoff_m, zoff_d, nz, b);
            if (l1 < nz) sDxa[2 * nz + l1] = b;
          }
          if (tid == 64) sI[20] = chol_ok ? 1 : 0;
        }
        __syncthreads();                                                                       // barrier 4
        const int flag = sI[18];
        if (flag == 1) { converged = true; break; }
        if (flag == 3) break;
        // (the snapshot comes before the factorisation's verdict: a pivot lost to rounding this late must not cost the loosely
        // converged iterate — the oracle keeps it the same way)
        if (flag == 2) { have_loose = true; if (tid < n) sZl[tid] = sZ[tid]; if (tid == 0) sI[16] = 1; }      // (every thread notes it for itself: sI[16] is read back below without a barrier in between, see qp_reg_kernel)
#ifdef NEP_QP_ITERDBG
        if ((!sI[19] || !sI[20]) && tid == 0 && ps.dbg && slot == NEP_QP_ITERDBG && mode == 0 && it < 59) ps.dbg[16 + (it + 1) * 8 + 5] = 100 + sI[19] * 10 + sI[20];
#endif
        if (!sI[19] || !sI[20]) break;
        // (nopred: see kCorrMinStep — every pass that re-derives the second-order term does so from uab / uax / uay)
        const bool nopred = sI[23] != 0;
        if (has_box) uab = nopred ? 0.0 : proj(brho, bax, sDxa);
        if (has_line) { uax = nopred ? 0.0 : proj(lrho, 0, sDxa); uay = nopred ? 0.0 : proj(lrho, 1, sDxa); }
        TICK(5);
        // ---- (P2) affine step: ratio test, the sum that gives mu_aff for any alpha, and the corrector's
        // right-hand side split as  va - sigma mu * vb  (sigma is only known after this pass's reduction):
        //   q = dsa/s;  dla = -lam - w dsa;  rcv/s - w rp = (lam + q dla - w rp) - sigma mu / s.
        // The first-order term of mu_aff needs no sum: s dla + lam dsa = -s lam for every row.
        double rmax = 0, c2 = 0, dmy = 0;
        {
          double b1a = 0, b1b = 0, vax = 0, vay = 0, vbx = 0, vby = 0;
          auto rowP2 = [&](bool ok, double s, double lam, double a, double ga, double h, double& va, double& vb) {
            const double rp = a + s - h, is = frcp(s), w = lam * is;
            const double dsa = -rp - ga, q = dsa * is, dla = -__builtin_fma(w, dsa, lam);
            rmax = fmax(rmax, fmax(-q, 1.0 + q));   // (the dummy line gives exactly 1, which the test rmax > 1 ignores, and dsa = 0)
            c2 += dsa * dla;
            va = __builtin_fma(q, dla, lam) - w * rp; vb = is;
          };
          if (has_box) {
            double va0, vb0, va1, vb1;
            rowP2(true, bs0, bl0, cpb, uab, bhi, va0, vb0); rowP2(true, bs1, bl1, -cpb, -uab, -blo, va1, vb1);
            b1a = va0 - va1; b1b = vb0 - vb1;
          }
          for_lines4([&](bool ok, int, double n1, double n2, double h, double s, double lam) {
            double va, vb;
            rowP2(ok, s, lam, n1 * cpx + n2 * cpy, n1 * uax + n2 * uay, h, va, vb);
            vax += va * n1; vay += va * n2; vbx += vb * n1; vby += vb * n2;     // (n1 = n2 = 0 on the dummy line)
          });
          vax = slice_sum(vax); vay = slice_sum(vay); vbx = slice_sum(vbx); vby = slice_sum(vby);
          // the T_lambda slots of sAccL / sTc were consumed before barrier 3: they carry vb now
          if (slice == 0) { double* o = sAccL + pair * 8; o[5] = vax; o[6] = vay; o[0] = vbx; o[1] = vby; }
          if (has_box) { sTc[brho * 6 + 3 + bax] = b1a; sTc[brho * 6 + bax] = b1b; }
        }
        if (tid == BS - 1 && has_qc) {
          const double sq = sc[sSq], lq = sc[sLq], wq = sc[sWq], rpq = sc[sRpq];
          double gd = 0; if (!nopred) for (int e = 0; e < n; e++) gd += sGq[e] * sDxa[e];
          const double dsq = -rpq - gd, dlq = -lq + wq * (rpq + gd);
          sc[sDsqA] = dsq; sc[sDlqA] = dlq;
          rmax = fmax(rmax, fmax(-dsq / sq, -dlq / lq));
          c2 += dsq * dlq;
        }
        reduce_put<1>(rmax, c2, dmy, redP2);
        __syncthreads();                                                                       // barrier 5
        reduce_get<1>(rmax, c2, dmy, redP2);
        double sm;
        {
          const double aaff = rmax > 1.0 ? frcp2(rmax) : 1.0;
          const double mu = sc[sMu];
          const double mua = ((1.0 - aaff) * sc[sSumSl] + aaff * aaff * c2) * inv_mt;
          const double rr = mua * frcp2(mu);
          sm = rr * rr * rr * mu;                              // sigma * mu, identical in every thread
          // never aim below a tenth of the gap the strict test asks for: with the long steps of kStepFracMax the centring
          // target would otherwise collapse by 1e5 per iteration, the last iterate would sit at mu ~ 1e-15 with weights
          // lambda/s ~ 1e17, and the rounding of that last step shows up as 1e-6 in the flat directions of the coefficients
          sm = fmax(sm, sp.tol_gap_floor * (1.0 + fabs(sc[sObj])) * inv_mt);
          if (nopred) sm = sm_keep;            // (sigma mu of the discarded predictor)
          else if (it >= sp.corr_from_it && aaff < kCorrMinStep) {
            // the affine step is too short for its second-order term to mean anything: the iteration is repeated from the same
            // point (a step of length zero) with the predictor discarded — identical in every thread, rare and late
            if (n_nopred >= sp.corr_max_count) break;             // (not going to end: give this attempt up)
            alpha_prev = 0.0; sm_keep = sm; sI[23] = 1; n_nopred++;
            it--;
            continue;
          }
        }
        TICK(6);
        TICK(7);
        // ---- corrector right-hand side (8 partial sums per entry), then the substitutions -----------
        if (tid < 8 * n) {
          const int o = tid >> 3, sl8 = tid & 7, ax = o / nz, c = o % nz;
          double t1 = 0;
          for (int q = 0; q < K; q++) {
            const int rho = sl8 + 8 * q;
            double ta = sTc[rho * 6 + 3 + ax], tb2 = sTc[rho * 6 + ax];
            if (q < 4) {   // position rows rho < 32 (those >= 4K carry zero line sums: sAccL rows are zeroed / never written)
              const double la = sAccL[rho * 8 + 5 + (ax & 1)], lb = sAccL[rho * 8 + (ax & 1)];
              const bool on = ax < 2 && rho < 4 * K;
              ta += on ? la : 0.0; tb2 += on ? lb : 0.0;
            }
            t1 += sB[rho * SBS + c] * __builtin_fma(-sm, tb2, ta);
          }
          t1 = slice_sum(t1);
          if (sl8 == 0) sRhs[o] = t1;
        }
        __syncthreads();                                                                       // barrier 6
        if (tid < 128) {
          const bool w0 = tid < 64;
          const int o = w0 ? tid : 2 * nz + (tid - 64);
          const bool mine = w0 ? tid < n0 : (!has_qc && tid - 64 < nz);
          double b = 0;
          if (mine) {
            b = -sRd[o] + sRhs[o];
            if (has_qc) { const double sq = sc[sSq], lq = sc[sLq]; const double rcq = sq * lq - sm + sc[sDsqA] * sc[sDlqA]; b += sGq[o] * (rcq / sq - sc[sWq] * sc[sRpq]); }
          }
          if (w0) { b = solve_wave(lds0 + oM * 8, lds0 + oInvD * 8, n0, b); if (mine) sDx[o] = b; }
          else if (!has_qc) { b = solve_wave(zoff_m, zoff_d, nz, b); if (mine) sDx[o] = b; }
        }
        __syncthreads();                                                                       // barrier 7
        if (has_box) udb = proj(brho, bax, sDx);
        if (has_line) { udx = proj(lrho, 0, sDx); udy = proj(lrho, 1, sDx); }
        TICK(8);
        // ---- (P5) step length of the combined direction ------------------------------------------
        rmax = 0;
        auto rowP5 = [&](bool ok, double s, double lam, double a, double ga, double gd, double h) {
          const double rp = a + s - h, is = frcp(s), w = lam * is;
          const double dsa = -rp - ga, dla = -lam + w * (rp + ga);
          const double rcv = s * lam - sm + dsa * dla;
          const double ds = -rp - gd, dl = -rcv * is + w * (rp + gd);
          rmax = fmax(rmax, ok ? fmax(-ds * is, -dl * frcp(lam)) : 0.0);
        };
        if (has_box) { rowP5(true, bs0, bl0, cpb, uab, udb, bhi); rowP5(true, bs1, bl1, -cpb, -uab, -udb, -blo); }
        for_lines4([&](bool ok, int, double n1, double n2, double h, double s, double lam) { rowP5(ok, s, lam, n1 * cpx + n2 * cpy, n1 * uax + n2 * uay, n1 * udx + n2 * udy, h); });
        if (tid == BS - 1 && has_qc) {
          const double sq = sc[sSq], lq = sc[sLq], wq = sc[sWq], rpq = sc[sRpq];
          double gd = 0; for (int e = 0; e < n; e++) gd += sGq[e] * sDx[e];
          const double rcq = sq * lq - sm + sc[sDsqA] * sc[sDlqA];
          const double dsq = -rpq - gd, dlq = -rcq / sq + wq * (rpq + gd);
          sc[sDsq] = dsq; sc[sDlq] = dlq;
          rmax = fmax(rmax, fmax(-dsq / sq, -dlq / lq));
        }
        reduce_put<0>(rmax, dmy, dmy, redP5);
        __syncthreads();                                                                       // barrier 8
        reduce_get<0>(rmax, dmy, dmy, redP5);
        {
          double alpha = rmax > 0.0 ? frcp2(rmax) : 1e30;
          alpha = fmin(1.0, fmin(fmax(1.0 - sc[sMu], kStepFracMin), kStepFracMax) * alpha);
          if (tid == 0) { if (alpha < 1e-8) sI[17]++; else sI[17] = 0; sI[23] = 0; }
          if (tid < n) sZ[tid] += alpha * sDx[tid];
          alpha_prev = alpha; sm_prev = sm;
        }
        TICK(9);
      }
#ifdef NEP_PROFILE_PHASES
      if (prof) tph[10] += clock64();
#endif
      if (uncon) converged = true;
      if (!converged && have_loose) { __syncthreads(); if (tid < n) sZ[tid] = sZl[tid]; if (tid == 0) sc[sObj] = sc[sObjLoose]; converged = true; }
    }
    iters_total = it; if (mode == 0) iters_first = it;
    __syncthreads();
    if (converged) {
      status = mode;   // NEP_OK / NEP_RELAXED
      objective = sc[sObj];
      if (tid < 12 * K) {  // theta = Th z + ThU init
        const int ax = tid / (4 * K), r = tid % (4 * K);
        double v = tb->ThU[r][0] * sInit[ax * 3] + tb->ThU[r][1] * sInit[ax * 3 + 1] + tb->ThU[r][2] * sInit[ax * 3 + 2];
        double th[kNZ];
#pragma unroll
        for (int c = 0; c < kNZ; c++) th[c] = tb->Th[r][c];   // all loads in flight at once (columns >= nz are zero)
#pragma unroll
        for (int c = 0; c < kNZ; c++) v += c < nz ? th[c] * sZ[ax * nz + c] : 0.0;
        sTheta[(ax * 8 + r / 4) * 4 + (r % 4)] = v;
      }
      break;
    }
  }
  };   // run
  if (L <= ps.lds_lines) { if (LL == kLLHalf) run(std::true_type{}, std::integral_constant<int, kLLHalf>{}); else run(std::true_type{}, std::integral_constant<int, 0>{}); }
  else run(std::false_type{}, std::integral_constant<int, 0>{});
  __syncthreads();
  if (!CULL) return false;
  if (use_far || sI[41] == 0 || status == NEP_FAILED) return false;
  {  // the far lines against the solution: position control points from the base rows of the converged mode
    const QpTable* __restrict__ tbv = tables + status * (kMaxK + 1) + K;
    const int nzv = tbv->nz;
    if (tid < 8 * K) {
      const int rho = tid >> 1, ax = tid & 1;
      double v = sOff[rho * 3 + ax];
      for (int c = 0; c < nzv; c++) v = __builtin_fma(sB[rho * SBS + c], sZ[ax * nzv + c], v);
      sAccL[rho * 2 + ax] = v;
    }
    __syncthreads();
    bool viol = false;
    const int F = sI[41];
    for (int e = tid; e < F; e += BS) {
      int i = 0;
#pragma unroll
      for (int j = 1; j < NEP_MAX_POL; j++) i += (e >= sI[52 + j]) ? 1 : 0;
      const double* src = ps.line_nd + ((long)slot * NEP_MAX_POL + i) * sp.lines_cap * 3;
      const long q = (long)sp.lines_cap - 1 - (e - sI[52 + i]);
      const double n1 = src[3 * q], n2 = src[3 * q + 1], dd = src[3 * q + 2];
#pragma unroll
      for (int k = 0; k < 4; k++) viol = viol || (n1 * sAccL[(4 * i + k) * 2] + n2 * sAccL[(4 * i + k) * 2 + 1] + dd - 1.0 > 0.0);
    }
    if (viol) sI[21] = 1;
    __syncthreads();
    return sI[21] != 0;
  }
  };   // solve_once
  // (one call site: a second inlined copy of the solver body pushes the compiler past its unrolling budget and the
  // row groups' private arrays into scratch)
#pragma nounroll
  for (int attempt = 0; attempt < (CULL ? 2 : 1) && K_ok; attempt++) { if (!solve_once(attempt == 1)) break; }
  __syncthreads();
  const int Ko = K_ok ? K : 0;      // segments of the output
  // ---- outputs -----------------------------------------------------------------------------------
  if (status == NEP_FAILED) { if (tid < 96) sTheta[tid] = sCoef[tid]; }                    // :856-859
  else if (z_override) { if (tid < 32) sTheta[64 + tid] = sCoef[64 + tid]; }             // :879-880
  __syncthreads();
  if (tid < 96) (&sol->coeff[0][0][0])[tid] = ((tid % 32) / 4 < Ko) ? sTheta[tid] : 0.0;
  if (tid <= NEP_MAX_POL) sol->times[tid] = (tid <= Ko) ? g->t_start + tid * T : 0.0;      // :898 (times = i*T_span + t_start)
  const int ns_all = sched.n[Ko];
  const int ns = ns_all < sp.max_states ? ns_all : sp.max_states;
  if (tid == 0) {
    sol->stats.status = status; sol->stats.iters = iters_total; sol->stats.iters_first = iters_first;
    // bucket entries of LPs without a separating line are (0,0,0) = null rows (constraint skipped)
    int n_lp = 0, n_lpf = 0;
    if (ps.lp_stats && !ps.lines_override) {
      int v[2 * NEP_MAX_POL];
#pragma unroll
      for (int i = 0; i < 2 * NEP_MAX_POL; i++) v[i] = ps.lp_stats[(long)slot * NEP_MAX_POL * 2 + i];   // one round trip
#pragma unroll
      for (int i = 0; i < NEP_MAX_POL; i++) { n_lp += v[2 * i]; n_lpf += v[2 * i + 1]; }
    }
    sol->stats.n_lines = L_all - n_lpf; sol->stats.n_lp = n_lp; sol->stats.n_lp_failed = n_lpf;
    sol->stats.n_rows = K_ok ? 48 * K + 4 * ((culled && L_used < L_all) ? L_used : L_used - n_lpf) : 0; sol->stats.qc_active = has_qc ? 1 : 0;   // rows solved for (null rows of failed LPs excluded)
    sol->stats.objective = objective; { const long long dt_ = (long long)wall_clock64() - t_wg0; const double us_ = (double)dt_ * sp.us_per_tick; sol->stats.solve_us = us_; if (ps.order_key) { const double k_ = us_ * 0.125; const int kn = k_ > 63.0 ? 63 : (int)k_, ko = ps.order_key[slot] - sp.qp_key_decay; ps.order_key[slot] = (sp.qp_key_decay > 0 && ko > kn) ? ko : kn; } }   // the per-replan device time, and the next launch's ordering key (8 us bins)
    sol->K = Ko; sol->n_states = ns;
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
#ifdef NEP_PROFILE_PHASES
  if (prof) tph[11] = clock64() - tstart;   // [11]: workgroup lifetime up to here
#ifndef NEP_QP_ITERDBG
  if (prof && tid == 0) { for (int k = 0; k < 12; k++) ps.dbg[(long)slot * 16 + k] = tph[k]; ps.dbg[(long)slot * 16 + 12] = iters_total; for (int k = 0; k < 3; k++) ps.dbg[(long)slot * 16 + 13 + k] = tset[k]; }
#endif
#endif
  if (ps.commit) {  // the record the agent would publish (neptune_ros.cpp:434-480)
    nep_traj_rec* cr = ps.commit + slot;
    const int own = sp.first_local + (slot % sp.n_local);
    if (status == NEP_FAILED) {
      // A failed replan publishes nothing: the agent keeps flying its committed trajectory (neptune_ros.cpp:651-663).  With
      // the previous records at hand (nep_batch_replan's d_committed) that record is carried over; otherwise d_commit[slot]
      // is left as the caller passed it (the usual round loop hands the buffer that holds the previous round's records).
      if (ps.prev_commit) {
        const double* src = (const double*)(ps.prev_commit + (long)(slot / sp.n_local) * sp.num_agents + own);
        for (int e = tid; e < (int)(sizeof(nep_traj_rec) / sizeof(double)); e += BS) ((double*)cr)[e] = src[e];
      }
      return;
    }
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
  const bool cull = ps.line_far != nullptr && !ps.lines_override;
  static DynLdsAttr attr[2];
  (void)attr[cull].ensure(cull ? (const void*)qp_kernel<true> : (const void*)qp_kernel<false>, lds_bytes);   // (a failure surfaces as the launch error)
  if (cull) hipLaunchKernelGGL(qp_kernel<true>, dim3(n_slots), dim3(BS), lds_bytes, st, sp, ps, tables, sched);
  else hipLaunchKernelGGL(qp_kernel<false>, dim3(n_slots), dim3(BS), lds_bytes, st, sp, ps, tables, sched);
}


// Launch order of the QP workgroups: longest expected solve first.  A launch is a handful of waves of workgroups over the chip
// (1 024 at a time at four per CU) whose durations spread 1 : 3 (iteration count; the terminal ball row's 3 nz x 3 nz
// factorisation); in slot order the long ones that happen to start last set the kernel's end while most CUs idle.  The previous
// replan of the same slot is the predictor (a receding-horizon replanner re-solves almost the same problem): a counting
// sort of the slots by its measured device time (stats.solve_us; the kernel leaves it in 8 us bins in ps.order_key), descending.  Results do not depend on the order
// (every workgroup owns its slot).
__global__ __launch_bounds__(1024) void order_kernel(int n, const int* __restrict__ key, int* __restrict__ order, int* __restrict__ zero_these) {
  if (zero_these && threadIdx.x < 4) zero_these[threadIdx.x] = 0;      // (the polish pass's counters of the launch sequence that starts here: no kernel or memset node of their own)
  // (sixteen sub-histograms by thread index: most keys fall into two or three bins, and one counter per bin would serialise
  // the whole launch's atomics on them)
  __shared__ int hist[64][16], tot[64];
  const int tid = threadIdx.x, sub = tid & 15;
  hist[tid >> 4][sub] = 0;
  __syncthreads();
  for (int i = tid; i < n; i += 1024) atomicAdd(&hist[63 - (key[i] & 63)][sub], 1);
  __syncthreads();
  if (tid < 64) { int o = 0; for (int u = 0; u < 16; u++) { const int c = hist[tid][u]; hist[tid][u] = o; o += c; } tot[tid] = o; }
  __syncthreads();
  if (tid == 0) { int o = 0; for (int b = 0; b < 64; b++) { const int c = tot[b]; tot[b] = o; o += c; } }
  __syncthreads();
  for (int i = tid; i < n; i += 1024) { const int b = 63 - (key[i] & 63); order[tot[b] + atomicAdd(&hist[b][sub], 1)] = i; }
}
// The same for launches whose neighbouring slots share data (the front end: the 64 or 256 searches of a scene read the same hulls,
// boxes and packed records): the dispatcher is observed to run block b on XCD b % 8 (MI355X_MICROARCH.md, a speed matter only), so
// in slot order every XCD's L2 sees every scene's data — 32 MB of it at config 5 against 4 MB of L2 — and a record read costs a
// trip to the Infinity Cache (measured: ~4 000 cycles per visited record in the entangle check's agent loop).  Here XCD x gets the
// x-th eighth of the slots — a few whole scenes — and, within it, the longest expected first: order[8 p + x] = the p-th slot of
// eighth x by descending key (key == null: in slot order).  n must be a multiple of 8.
__global__ __launch_bounds__(1024) void order_xcd_kernel(int n, const int* __restrict__ key, int* __restrict__ order) {
  __shared__ int hist[8][64];
  const int tid = threadIdx.x, chunk = n >> 3;
  if (tid < 512) hist[tid >> 6][tid & 63] = 0;
  __syncthreads();
  if (key) for (int i = tid; i < n; i += 1024) atomicAdd(&hist[i / chunk][63 - (key[i] & 63)], 1);
  __syncthreads();
  if (tid < 8 && key) { int o = 0; for (int b = 0; b < 64; b++) { const int c = hist[tid][b]; hist[tid][b] = o; o += c; } }
  __syncthreads();
  for (int i = tid; i < n; i += 1024) {
    const int c = i / chunk;
    const int pos = key ? atomicAdd(&hist[c][63 - (key[i] & 63)], 1) : i - c * chunk;
    order[pos * 8 + c] = i;
  }
}
void launch_order_xcd(int n_slots, const int* key, int* order, hipStream_t st) {
  if (n_slots > 0) hipLaunchKernelGGL(order_xcd_kernel, dim3(1), dim3(1024), 0, st, n_slots, key, order);
}
void launch_qp_order(int n_slots, const int* key, int* order, hipStream_t st, int* zero_these) {
  if (n_slots > 0) hipLaunchKernelGGL(order_kernel, dim3(1), dim3(1024), 0, st, n_slots, key, order, zero_these);
}

}  // namespace nep
