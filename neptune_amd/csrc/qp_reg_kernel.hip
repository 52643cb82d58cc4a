// qp_reg_kernel.hip — the interior point of qp_kernels.hip with the line-row state in REGISTERS.
//
// Same problem, same iteration, same thread roles as qp_kernel (see its header): one 256-thread workgroup per replan,
// thread (seg, k, slice) = (tid >> 5, (tid >> 3) & 3, tid & 7) owns the rows of control point k of the lines of segment
// seg whose index is congruent to slice modulo 8.  What differs is where a row's (s, lambda) lives.  qp_kernel keeps them
// in LDS (8 doubles per line: 32 KB of the 82 KB carve at config 4) and needs every VGPR it can get (256, and spills):
// two workgroups per CU by both limits, and the kernel is latency-bound.  The register file of a CU (512 KB) is three times
// its LDS, so here a thread keeps its first RS rows in RS x 2 doubles of registers — at config 4 a segment has 64-68 lines,
// i.e. 8 or 9 rows per thread — and only the line coefficients (n1, n2, h: 24 B per line) stay in LDS: 40 KB per
// workgroup, 128 VGPRs, FOUR workgroups per CU.  Rows beyond RS per thread, and coefficients beyond the LDS carve, go to
// the per-slot global scratch (correct for any count; config-5 sized problems are given to qp_kernel by the host).
// The solver body is one straight function whose phases scope their temporaries: no 4-wide row groups (the other three
// workgroups of the CU hide the latency instead), no out-of-line calls.
//
// One iteration, seven workgroup barriers (eight with the terminal ball row):
//   A1 + A2  all threads: apply the previous step to (s, lambda), then residuals / weights and the line rows' sums per
//            control point (three DPP steps per 8-lane group); the box rows' sums go to sTc / sDc directly            -> b1
//   assembly wave 1: B' [T_lambda | T1] (dual residual, predictor rhs); waves 2, 3: the blocks of M; one 16 x 16 MFMA
//            tile each over the 64 base rows, the line sums added as the operands are read; wave 0: the iteration's
//            scalars (and the ball row: then one more barrier before the tiles)                                       -> b3
//   factor   wave 0: Cholesky of the (x, y) block in registers + the predictor; wave 1: convergence test, then the z block -> b4
//   P2       all threads: affine ratio test, corrector right-hand side as  va - sigma mu vb                           -> b5
//            (late in a solve, an affine step shorter than kCorrMinStep sends the iteration back to A1 with a step of length
//            zero and the predictor discarded: qp_common.h)
//   rhs      144 threads: B' (va - sigma mu vb)                                                                       -> b6
//   solve    waves 0, 1: the corrector                                                                                -> b7
//   P5       all threads: step length of the combined direction                                                       -> b8
// Cycle counts per phase: DESIGN.md section 6 (make PROFILE=1, scripts/qp_reg_phases.py).  The workgroups of a launch are
// started longest-expected-first (ps.order, order_kernel in qp_kernels.hip); every workgroup leaves its measured duration
// in nep_stats.solve_us.
#include <hip/hip_runtime.h>

#include "qp_common.h"

static_assert(nep::kCorrFromIt == nep::kCorrFromItDefault && nep::kCorrMaxCount == nep::kCorrMaxCountDefault, "the give-up rule's constants are stated twice");

// line rows a thread keeps in registers (8 per segment and slot: 9 slots = 72 lines per segment before the global scratch is used)
#ifndef NEP_QP_REG_SLOTS
#define NEP_QP_REG_SLOTS 9
#endif

// 1: the contractions over the 64 base rows — the normal matrix's weighted Gram blocks  M_sel = sum_rho D_sel[rho] B[rho]' B[rho]
// (four weight sets) and  B' [T_lambda | T1] — are formed on the matrix cores as three 16x16 tiles of v_mfma_f64_16x16x4_f64
// (K = 64: 16 instructions per tile, one tile per wave); 0: the VALU versions (two matrix entries per thread, 32 base rows per
// lane pair; eight partial sums per output of the residual), kept for A/B.
#ifndef NEP_QP_MFMA
#define NEP_QP_MFMA 1
#endif
// workgroups per CU the register allocation is bounded for (4: 128 VGPRs, 3: 168)
#ifndef NEP_QP_REG_WGS
#define NEP_QP_REG_WGS 4
#endif
// 1: a row's 1 / s is kept next to (s, lambda) from pass A2 to the next iteration's pass A1 (four reciprocals per row and
// iteration fewer, a third register pair per row: needs the 168-VGPR budget of NEP_QP_REG_WGS = 3)
// 2: the same as a float (one register per row): the seed of the Newton step that otherwise starts from v_rcp_f64, whose result
// is no better than a float's 24 bits anyway
#ifndef NEP_QP_CACHE_IS
#define NEP_QP_CACHE_IS 0
#endif
// 1: the terminal ball row (a rank-one term w g g' that couples the three axes) is taken out of the normal matrix and put back
// by the Sherman-Morrison formula: with D = M - w g g' block diagonal (x, y | z), x = y - u (w g'y) / (1 + w g'u), y = D^-1 b,
// u = D^-1 g — the two blocks are factored in registers on two waves as for a replan without the row (sizes 12 and 6 at
// K = 8), one more substitution per iteration for u, one barrier more per solve.  0: the coupled 3 nz x 3 nz matrix, which
// at K = 8 (18 x 18, relaxed solve 24 x 24) is factored out of LDS at 1.5-3 x the iteration's duration.  Replans near their
// goal all have the row: a fifth of a closed loop's replans and its stragglers (bench.py's `moving` leg).
#ifndef NEP_QP_BALL_SM
#define NEP_QP_BALL_SM 1
#endif

namespace nep {
#if NEP_QP_CACHE_IS == 2
typedef float isd_t;
__device__ __forceinline__ double is_of(double s, isd_t c) { const double r = (double)c, e = __builtin_fma(-s, r, 1.0); return __builtin_fma(r, e, r); }
#elif NEP_QP_CACHE_IS == 1
typedef double isd_t;
__device__ __forceinline__ double is_of(double, isd_t c) { return c; }
#else
typedef double isd_t;
__device__ __forceinline__ double is_of(double s, isd_t) { return frcp(s); }
#endif

// (PST: the hooks of the polish pass under the presolve — an instantiation of their own, chosen when nep_batch_set_polish(h, 2) asks
// for it: they cost the presolve's kernel two more spilled registers and 3 % of its time, which the default does not pay)
template <bool CULL, int RS, bool PST = true>
__global__ __launch_bounds__(BS, NEP_QP_REG_WGS) void qp_reg_kernel(SceneParams sp, ProblemSet ps, const QpTable* __restrict__ tables, SampleSched sched) {
  __builtin_amdgcn_s_setprio(3);      // (latency-bound waves: when another scene group's hull / separator waves share the SIMD — bench.py's pipelined groups — these issue first)
  extern __shared__ __attribute__((aligned(16))) double smem[];
  double* sB = smem + oB; double* sOff = smem + oOff; double* sAccL = smem + oAccL;
  double* sDc = smem + oDc; double* sTc = smem + oTc;
  double* sM = smem + oM; double* sHax = smem + oHax;
  double* sZ = smem + oZ; double* sG = smem + oG; double* sRd = smem + oRd; double* sRhs = smem + oRhs;
  double* sDxa = smem + oDxa; double* sDx = smem + oDx; double* sGq = smem + oGq; double* sZl = smem + oZl;
  double* sEp = smem + oEp; double* sCoef = smem + oCoef; double* sTheta = smem + oTheta;
  double* sInit = smem + oInit; double* sc = smem + oScal; double* sRed = smem + oRed;
  int* sI = (int*)(smem + kFixedDoubles);      // [0..8] line offsets, [16..] flags (same map as qp_kernel)

  const int tid = threadIdx.x;
  if (ps.order_count && (int)blockIdx.x >= *ps.order_count) return;     // (the presolve's redo pass: a list that is empty nearly always)
  const int slot = ps.order ? ps.order[blockIdx.x] : (int)blockIdx.x;   // (launch order: see order_kernel)
  if (CULL && ps.presolved && ps.presolved[slot] != 0) {
    // qp_presolve_kernel certified and returned this replan's trajectory (round 6): what is left are the stores that depend on nothing
    // but the coefficients — generatePwpOut's samples (:911-934) and the record the agent publishes (neptune_ros.cpp:434-480) — written
    // here, beside the iterating replans' workgroups, and not by the wave that made the certificate
    if (gridDim.x == 1 && tid == 0 && ps.polish_list) { ps.polish_count[0] = 0; ps.polish_count[3] = 0; }      // (a one-workgroup launch is its own, empty, polish list)
    const nep_solution* __restrict__ so = ps.solution + slot;
    const int Kp = so->K;
    double* sTh = smem;                                       // [3][8][4]
    if (tid < 96) sTh[tid] = (&so->coeff[0][0][0])[tid];
    __syncthreads();
    if (ps.states) {
      const int ns_all = sched.n[Kp];
      const int ns = ns_all < sp.max_states ? ns_all : sp.max_states;
      for (int s = tid; s < ns; s += BS) {
        const int i = sched.seg[Kp * sp.max_states + s]; const double dt = sched.dt[Kp * sp.max_states + s];
        double* st = ps.states + ((long)slot * sp.max_states + s) * NEP_STATE_DOUBLES;
        for (int ax = 0; ax < 3; ax++) {
          const double* c = sTh + (ax * 8 + i) * 4;
          st[ax] = ((c[0] * (dt * dt * dt) + c[1] * (dt * dt)) + c[2] * dt) + c[3];
          st[3 + ax] = (c[0] * (3 * dt * dt) + c[1] * (2 * dt)) + c[2];
          st[6 + ax] = c[0] * (6 * dt) + c[1] * 2;
          st[9 + ax] = c[0] * 6;
        }
      }
    }
    if (ps.commit) {
      nep_traj_rec* cr = ps.commit + slot;
      const int own = sp.first_local + (slot % sp.n_local);
      const double t_start = ps.guess[slot].t_start, Tp = sp.T_span;
      if (tid == 0) {
        cr->id = own + 1; cr->is_agent = 1; cr->n_bend = 1; cr->valid = 1;
        for (int a = 0; a < 3; a++) { cr->bbox[a] = 2 * sp.drone_radius; cr->pos[a] = sTh[(a * 8) * 4 + 3]; }
        cr->bend[0][0] = ps.pb[2 * own]; cr->bend[0][1] = ps.pb[2 * own + 1];
        cr->pwp.n_seg = Kp;
      }
      if (tid <= NEP_TRAJ_MAX_SEG) cr->pwp.times[tid] = (tid <= Kp) ? t_start + tid * Tp : 0.0;
      for (int e = tid; e < 3 * NEP_TRAJ_MAX_SEG * 4; e += BS) {
        const int ax = e / (NEP_TRAJ_MAX_SEG * 4), r = e % (NEP_TRAJ_MAX_SEG * 4), seg = r / 4, j = r % 4;
        (&cr->pwp.coeff[0][0][0])[e] = (seg < Kp) ? sTh[(ax * 8 + seg) * 4 + j] : 0.0;
      }
    }
    return;
  }
  const long long t_wg0 = (long long)wall_clock64();          // this workgroup's lifetime goes to stats.solve_us (wall-clock ticks: sp.us_per_tick)
  const nep_guess* __restrict__ g = ps.guess + slot;
  const int K_in = g->K;
  const bool K_ok = K_in >= 1 && K_in <= NEP_MAX_POL && K_in <= sp.num_pol;   // (see qp_kernel)
  const int K = K_ok ? K_in : 1;
  const double T = sp.T_span, wgt = sp.weight;
  nep_solution* __restrict__ sol = ps.solution + slot;

  if (tid < 96) sCoef[tid] = (&g->coeff[0][0][0])[tid];
  for (int e = tid; e < kMaxR * 6; e += BS) sTc[e] = 0.0;
  sDc[tid] = 0.0; sAccL[tid] = 0.0;
  int status = NEP_FAILED, L_used = 0, L_all = 0;
  // (what only the last lines need — iteration counts, the objective, 1 / rows — waits in LDS, not in registers that would be
  // spilled to scratch for the whole solve: sI[31] iterations of the last solve, sI[32] of the first, sc[sObjOut], sc[sInvMt])
  if (tid == 0) { sI[31] = 0; sI[32] = 0; sI[26] = 0; sI[28] = 0; sI[29] = 0; sI[30] = 0; sc[sObjOut] = 0.0; }      // (sI[28]: the strict tests passed in this iteration; sI[29]: solves left for the polish pass, bit per mode)      // (written and read back by thread 0 only; a replan that never reaches a solve — K = 0 — reports zeros; sI[26]: sent to the redo pass)
  bool has_qc = false, z_override = false;

  // Line coefficients in LDS: [n1 | n2 | h][segment][SEGCAP], SEGCAP = 8 RS entries per segment whatever its line count — the
  // entries past a segment's last line hold the dummy line (0, 0, 1).  A thread's slot u is then a compile-time offset from
  // one base address (ds_read with an immediate), and a padded slot needs no redirect.  Lines beyond SEGCAP of a segment
  // (coefficients and row state) live in the per-slot global scratch.
  typedef __attribute__((address_space(3))) double* lds_ptr;
  constexpr int SEGCAP = 8 * RS, NS = NEP_MAX_POL * SEGCAP;
  const unsigned lds0 = __builtin_amdgcn_groupstaticsize();
  const lds_ptr ldyn = (lds_ptr)(unsigned int)(lds0 + (kFixedDoubles + 32) * sizeof(double));
  const int GL = ps.rows_cap / 4 + 2;
  double* __restrict__ gsp = ps.row_scratch + (long)(ps.scratch_by_block ? (int)blockIdx.x : slot) * (11L * GL);      // (the redo pass of a handle with pooled scratch: one area per listed replan)
  // phase cycle counters (make PROFILE=1, scripts/qp_phases.py): thread 0's clock, accumulated in 16 LDS words behind the carve
#ifdef NEP_PROFILE_PHASES
  long long* sProf = (long long*)(smem + kFixedDoubles + 32 + 3 * NS);
  const bool prof = ps.dbg != nullptr;
  long long tlast = 0;
  const long long tstart = clock64(), tstart_wall = (long long)wall_clock64();
  if (tid == 0) { for (int k = 0; k < 16; k++) sProf[k] = 0; tlast = tstart; }
#define TICK(k) do { if (prof && tid == 0) { const long long t_ = clock64(); sProf[k] += t_ - tlast; tlast = t_; } } while (0)
#else
#define TICK(k) do { } while (0)
#endif

  // ---- thread roles: derived where they are used (an opaque copy of the thread index keeps the compiler from parking role
  // indices, bounds and addresses in registers — or scratch — for the whole solve; see the iteration loop) ----
  const int R = 8 * K;
  auto otid = [&]() { int t = tid; asm volatile("" : "+v"(t)); return t; };
  // box rows of (axis, base row) = (t / R, t % R) for t < 3 R, R = 8 K: two comparisons instead of a division, so the role costs
  // nothing to re-derive and nothing to keep
  auto box_role = [&](int& ax, int& rho, double& hi, double& lo) {
    const int t = otid(), t8 = t >> 3;
    ax = (t8 >= K ? 1 : 0) + (t8 >= 2 * K ? 1 : 0);
    rho = ((t8 - ax * K) << 3) | (t & 7);
    hi = rho < 4 * K ? sp.maxs[ax] : (rho < 7 * K ? sp.v_max : sp.a_max);
    lo = rho < 4 * K ? sp.mins[ax] : (rho < 7 * K ? -sp.v_max : -sp.a_max);
    return t8 < 3 * K;
  };
  // line rows of control point (seg, k) = (t >> 5, (t >> 3) & 3), slice t & 7; base row of that control point: t >> 3


  // the mode's tables -> LDS (B, H of an axis, the end-point row, the small per-K tables in the normal matrix's place)
  auto stage_tables = [&](const QpTable* __restrict__ tb, int tm) {
    for (int e = tm; e < kMaxR * kNZ; e += BS) sB[(e / kNZ) * SBS + (e % kNZ)] = (&tb->B[0][0])[e];
    if (tm < 64) sHax[tm] = (&tb->Hax[0][0])[tm];
    if (tm < 8) sEp[tm] = tb->ep[tm];
    for (int e = tm; e < kSmallTab; e += BS) sM[e] = (&tb->Gi[0][0])[e];
  };
#pragma nounroll
  for (int attempt = 0; attempt < (CULL ? 2 : 1) && K_ok; attempt++) {
    const bool use_far = attempt == 1;
    __syncthreads();
    if (attempt == 0) stage_tables(tables + K, otid());         // the first solve's tables travel together with the line counts and the lines: one global latency instead of two
    if (tid <= NEP_MAX_POL) {   // line offsets per segment: nine threads read the eight counts (the same two cache lines) and each keeps its own prefix — one round trip and one barrier
      int o = 0, nf = 0, all = 0, over = 0, my_o = 0, my_nf = 0, my_cn = 0, my_cf = 0, nsk = 0, ovf = 0;
#pragma unroll
      for (int i = 0; i < NEP_MAX_POL; i++) {
        const int cn_raw = (i < K) ? ps.line_cnt[(long)slot * NEP_MAX_POL + i] : 0;
        ovf |= cn_raw < 0 ? 1 : 0;                                  // the segment's bucket overflowed: lines are missing (separator_body)
        const int cn = cn_raw < 0 ? -1 - cn_raw : cn_raw;
        const int cf = (i < K && CULL) ? ps.line_far[(long)slot * NEP_MAX_POL + i] : 0;
        const int cs = (i < K && CULL && ps.line_skip) ? ps.line_skip[(long)slot * NEP_MAX_POL + i] : 0;   // LPs the separator did not solve: their lines are known to be far (spatial presolve)
        if (i == tid) { my_o = o; my_nf = nf; my_cn = cn; my_cf = cf; }
        const int ct = cn + (use_far ? cf : 0);
        over |= ct > 8 * RS ? 1 : 0;
        o += ct; nf += cf; all += cn + cf + cs; nsk += cs;
      }
      if (tid < NEP_MAX_POL) { sI[tid] = my_o; sI[52 + tid] = my_nf; sI[44 + tid] = my_cn; sI[32 + tid] = my_cf; }
      else { sI[NEP_MAX_POL] = o; sI[41] = nf; sI[42] = all; sI[21] = 0; sI[43] = over; sI[40] = nsk; sI[25] = 0; sI[27] = ovf; }
    }
    __syncthreads();
    // A replan with a segment whose line bucket overflowed (NEP_FLAG_LINES is raised) is not solved: rows of the reference's problem are
    // missing, and a trajectory optimised without them must not be published.  It fails (status NEP_FAILED, output = the guess, the commit
    // slot keeps the previous record) — solver_gurobi_poly.cpp poses every line (:473-656).
    if (__builtin_amdgcn_readfirstlane(sI[27]) != 0) break;
    if (ps.scratch_chunks > 0 && __builtin_amdgcn_readfirstlane(sI[43]) != 0) {
      // rows beyond the register slots, on a handle whose row scratch is a small pool for the redo pass (the presolve's default: the
      // near lines of a replan fit the slots nearly always).  First pass: the replan goes to the redo pass as it is, unsolved;
      // redo pass: its area is the one of its place on the list — a list longer than the pool is flagged, those replans fail.
      if (CULL && ps.redo_count) {
        if (tid == 0) { sI[26] = 1; const int idx = atomicAdd(ps.redo_count, 1); ps.redo_list[idx] = slot; atomicAdd(ps.redo_count + 3, 1); }      // ([3]: listed unsolved, for the test hook)
        break;
      }
      if (!ps.scratch_by_block || (int)blockIdx.x >= ps.scratch_chunks) {
        if (tid == 0 && ps.flags) atomicOr(ps.flags, NEP_FLAG_SCRATCH);
        break;
      }
    }
    // (values every thread reads from LDS are wave-uniform: readfirstlane moves them, and what is computed from them, to SGPRs)
    const int L = __builtin_amdgcn_readfirstlane(sI[NEP_MAX_POL]);
    L_used = L; L_all = __builtin_amdgcn_readfirstlane(sI[42]);
    if (tid < 9) {
      if (tid < 3) {
        const double* c = sCoef + (tid * 8 + (K - 1)) * 4;
        sc[sFinal0 + tid] = ((T * T * T) * c[0] + (T * T) * c[1] + T * c[2]) + c[3];   // final_pos_ (:226-228)
      }
      sInit[tid] = sCoef[((tid / 3) * 8 + 0) * 4 + 1 + (tid % 3)];                    // b0,c0,d0 (:390-396)
    }
    // the separator's buckets -> the coefficient carve (and the global scratch for what exceeds a segment's SEGCAP entries)
    for (int e = tid; e < NS; e += BS) {
      const int i = e / SEGCAP, j = e - i * SEGCAP;
      const int cn = sI[44 + i], ct = sI[i + 1] - sI[i];     // near lines, lines of this attempt
      double n1 = 0.0, n2 = 0.0, h = 1.0;
      if (j < ct) {
        const double* src = ps.line_nd + ((long)slot * NEP_MAX_POL + i) * sp.lines_cap * 3;
        const long q = j < cn ? j : (long)sp.lines_cap - 1 - (j - cn);      // near lines from the front, far ones from the back
        n1 = src[3 * q]; n2 = src[3 * q + 1]; h = 1.0 - src[3 * q + 2];
      }
      ldyn[e] = n1; ldyn[NS + e] = n2; ldyn[2 * NS + e] = h;
    }
    if (sI[43]) {                                             // some segment has more than SEGCAP lines (uniform)
      for (int e = tid; e < L; e += BS) {
        int i = 0;
#pragma unroll
        for (int j = 1; j < NEP_MAX_POL; j++) i += (e >= sI[j]) ? 1 : 0;
        const int l = e - sI[i], cn = sI[44 + i];
        if (l < SEGCAP) continue;
        const double* src = ps.line_nd + ((long)slot * NEP_MAX_POL + i) * sp.lines_cap * 3;
        const long q = l < cn ? l : (long)sp.lines_cap - 1 - (l - cn);
        gsp[e] = src[3 * q]; gsp[GL + e] = src[3 * q + 1]; gsp[2 * GL + e] = 1.0 - src[3 * q + 2];
      }
    }
    __syncthreads();
    TICK(13);
    const double dix = sCoef[3] - sc[sFinal0], diy = sCoef[32 + 3] - sc[sFinal1], diz = sCoef[64 + 3] - sc[sFinal2];
    has_qc = __builtin_amdgcn_readfirstlane(sqrt(dix * dix + diy * diy + diz * diz) < 1.0 ? 1 : 0) != 0;   // :697-702
    z_override = __builtin_amdgcn_readfirstlane(sqrt(dix * dix + diy * diy) < 1.0 ? 1 : 0) != 0;           // :879-880
    const int mt = 48 * K + 4 * L + (has_qc ? 1 : 0);
    if (tid == 0) { sc[sInvMt] = 1.0 / (double)mt; sc[sMtD] = (double)mt; }               // (one division per attempt; the per-iteration means multiply by it)

    const int li = tid >> 5, slice = tid & 7;
    const int seg_cnt = sI[li + 1] - sI[li];                  // lines of my segment (0 for segments >= K)
    // slots in use by this wave (it covers segments 2w and 2w + 1): wave-uniform, so the slot loop branches on the scalar unit
    int n_u;
    {
      const int w2 = (tid >> 6) * 2;
      const int ca = sI[w2 + 1] - sI[w2], cb = sI[w2 + 2] - sI[w2 + 1];
      const int m = ((ca > cb ? ca : cb) + 7) >> 3;
      n_u = __builtin_amdgcn_readfirstlane(m < RS ? m : RS);
    }
    int my_cnt = seg_cnt - slice; my_cnt = my_cnt > 0 ? (my_cnt + 7) >> 3 : 0;       // my rows (slots u < my_cnt hold real lines)
    const bool over = __builtin_amdgcn_readfirstlane(sI[43]) != 0;
    const lds_ptr cbase = ldyn + (li * SEGCAP + slice);

    // One pass over this thread's line rows.  body(ok, n1, n2, h, s, lambda); padded slots see the dummy line (0, 0, 1): their
    // slack stays exactly 1 (activity and slack step are exactly zero), their multiplier follows lambda <- (1 - alpha) lambda +
    // alpha sigma mu, positive and finite, and everything it enters is multiplied by the zero normal — `ok` masks the two places
    // where a padded row would still count: the complementarity sum and the step-length test.
    auto for_rows = [&](double (&sl)[RS], double (&ll)[RS], isd_t (&il)[RS], auto&& body) {
      // (a block per slot would leave every row waiting for its own three ds_reads)
      // slots 0..3 and 4..7 four per scalar branch, 8 alone (a wave whose segments have at most 64 lines stops after 7): the rows'
      // coefficient loads are issued ahead of the rows' arithmetic, group by group
      static_assert(RS == 9, "slot grouping below");
#pragma unroll
      for (int u = 0; u < 8; u += 4) {
        if (u < n_u) {
          double c1[4], c2[4], c3[4];
#pragma unroll
          for (int v = 0; v < 4; v++) { c1[v] = cbase[8 * (u + v)]; c2[v] = cbase[NS + 8 * (u + v)]; c3[v] = cbase[2 * NS + 8 * (u + v)]; }
#pragma unroll
          for (int v = 0; v < 4; v++) body(u + v < my_cnt, c1[v], c2[v], c3[v], sl[u + v], ll[u + v], il[u + v]);
        }
      }
      if (8 < n_u) {
        constexpr int u = 8;
        const double n1 = cbase[8 * u], n2 = cbase[NS + 8 * u], h = cbase[2 * NS + 8 * u];
        body(u < my_cnt, n1, n2, h, sl[u], ll[u], il[u]);
      }
      if (over) {   // rows beyond the register slots: coefficients and state in the global scratch (rare: the role is re-derived here rather than kept)
        const int t_ = otid(), li = t_ >> 5, lk = (t_ >> 3) & 3;
        const int seg_end = sI[li + 1] - sI[li];
        for (int j = SEGCAP + (t_ & 7); j < seg_end; j += 8) {
          const int l = sI[li] + j;
          const double n1 = gsp[l], n2 = gsp[GL + l], h = gsp[2 * GL + l];
          double s = gsp[(3 + lk) * GL + l], lam = gsp[(7 + lk) * GL + l];
          isd_t isg = (isd_t)frcp(s);
          body(true, n1, n2, h, s, lam, isg);
          gsp[(3 + lk) * GL + l] = s; gsp[(7 + lk) * GL + l] = lam;
        }
      }
    };

    status = NEP_FAILED; if (tid == 0) { sI[31] = 0; sI[32] = 0; sc[sObjOut] = 0.0; if (CULL) sI[29] = 0; }      // (sI[29]: a second attempt supersedes what the first one left for the polish pass)

    for (int mode = 0; mode < 2; mode++) {
      const QpTable* __restrict__ tb = tables + mode * (kMaxK + 1) + K;
      const int nz = mode == 1 ? K : (K > 2 ? K - 2 : 0), n = 3 * nz;
      const int tm = otid();     // (the set-up below runs once per mode: nothing of it is worth hoisting out of the mode loop into registers)
      __syncthreads();
      if (mode == 1 || attempt == 1) stage_tables(tb, tm);      // (mode 0 of the first attempt: staged next to the line gather, above)
      if (tm < 96) { const int ax = tm >> 5, r = tm & 31; sTheta[tm] = r < 4 * K ? sCoef[tm] - (tb->ThU[r][0] * sInit[ax * 3] + tb->ThU[r][1] * sInit[ax * 3 + 1] + tb->ThU[r][2] * sInit[ax * 3 + 2]) : 0.0; }
      if (tm < 3 * R) { const int ax = div_small(tm, R), rho = tm - ax * R; sOff[rho * 3 + ax] = tb->U[rho][0] * sInit[ax * 3] + tb->U[rho][1] * sInit[ax * 3 + 1] + tb->U[rho][2] * sInit[ax * 3 + 2]; }
      // the iterate and the two directions are read eight entries at a time from an axis' first one (against B's or Hax's zero
      // columns): what lies beyond the 3 nz entries in use must be finite
      if (tm >= n && tm < 24) { sZ[tm] = 0.0; sDxa[tm] = 0.0; sDx[tm] = 0.0; }
      __syncthreads();
      const double* tGiP = sM + tGi; const double* tUpP = sM + tUp; const double* tUvP = sM + tUv; const double* tUaP = sM + tUa;
      const double* tZpP = sM + tZp; const double* tPpP = sM + tPp; const double* tResP = sM + tResU;
      if (tm < 24) { const int ax = tm >> 3, r = tm & 7; sRhs[tm] = tPpP[r * 3] * sInit[ax * 3] + tPpP[r * 3 + 1] * sInit[ax * 3 + 1] + tPpP[r * 3 + 2] * sInit[ax * 3 + 2]; }
      else if (tm >= 32 && tm < 35) { const int ax = tm - 32; sc[sPe0 + ax] = tUpP[0] * sInit[ax * 3] + tUpP[1] * sInit[ax * 3 + 1] + tUpP[2] * sInit[ax * 3 + 2] - sc[sFinal0 + ax]; }
      __syncthreads();
#if !NEP_QP_MFMA
      // normal-matrix entries owned by this thread (xx, yx, yy, zz blocks; a pair of lanes per entry), packed: ci | cj<<4 | sel<<8 | on<<12
      int me[2];
      {
        const int tri_n = nz * (nz + 1) / 2, n_ent = 3 * tri_n + nz * nz;
        auto tri = [&](int e, int& r, int& c) { r = (int)((sqrt(8.0 * e + 1.0) - 1.0) * 0.5); while ((r + 1) * (r + 2) / 2 <= e) r++; while (r * (r + 1) / 2 > e) r--; c = e - r * (r + 1) / 2; };
#pragma unroll
        for (int u = 0; u < 2; u++) {
          const int e = (tm >> 1) + u * (BS / 2);
          const bool on = e < n_ent && nz > 0;
          int ci = 0, cj = 0, sel = 0;
          if (on) {
            if (e < tri_n) { tri(e, ci, cj); sel = 0; }
            else if (e < tri_n + nz * nz) { const int f = e - tri_n; ci = f / nz; cj = f % nz; sel = 1; }
            else if (e < 2 * tri_n + nz * nz) { tri(e - tri_n - nz * nz, ci, cj); sel = 2; }
            else { tri(e - 2 * tri_n - nz * nz, ci, cj); sel = 3; }
          }
          me[u] = ci | (cj << 4) | (sel << 8) | ((on ? 1 : 0) << 12);
        }
      }
#endif
      bool converged = false, have_loose = false;      // (have_loose: a loosely converged iterate has been snapshot — every thread derives it from the workgroup-uniform verdict, see below)
      int it = 0;
      TICK(14);
      const long long t_solve0 = (long long)wall_clock64();   // m_.optimize() starts here: every solve (first and relaxed) has its own TimeLimit
      if (nz == 0) {
        // ---- K <= 2 with the terminal rows: a single point, feasible or not (tolerance 1e-6) ------
        double viol = 0.0, d0 = 0, d1 = 0, d2 = 0;
        { int ax, rho; double hi, lo; if (box_role(ax, rho, hi, lo)) { const double a = sOff[rho * 3 + ax]; viol = fmax(a - hi, lo - a); } }
        { const int lrho = otid() >> 3; const double ox = sOff[lrho * 3], oy = sOff[lrho * 3 + 1]; double du_[RS], dv_[RS]; isd_t dw_[RS]; for_rows(du_, dv_, dw_, [&](bool v, double n1, double n2, double h, double&, double&, isd_t&) { viol = fmax(viol, v ? n1 * ox + n2 * oy - h : 0.0); }); }
        if (tm < 6) {
          const int ax = tm / 2, e = tm % 2;
          viol = fmax(viol, fabs(tResP[e * 3] * sInit[ax * 3] + tResP[e * 3 + 1] * sInit[ax * 3 + 1] + tResP[e * 3 + 2] * sInit[ax * 3 + 2]));
        }
        if (tm == 0 && has_qc) {
          double c = -0.10 * 0.10;
          for (int ax = 0; ax < 3; ax++) { const double pe = sc[sPe0 + ax]; c += pe * pe; }
          viol = fmax(viol, c);
        }
        block_reduce4(viol, d0, d1, d2, sRed);
        converged = __builtin_amdgcn_readfirstlane(viol <= 1e-6 ? 1 : 0) != 0;
        if (tm == 0) {
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
        if (tm < n) {
          const int ax = div_small(tm, nz), c = tm - ax * nz;
          double z = 0;
#pragma unroll 8
          for (int r = 0; r < 4 * kMaxK; r++) z = __builtin_fma(tZpP[c * 4 * kMaxK + r], sTheta[ax * 32 + r], z);
          sZ[tm] = z;
          sG[tm] = (tGiP[c * 3] * sInit[ax * 3] + tGiP[c * 3 + 1] * sInit[ax * 3 + 1] + tGiP[c * 3 + 2] * sInit[ax * 3 + 2]) - 2 * wgt * sEp[c] * sc[sFinal0 + ax];
        }
        if (tm == 64) {   // (wave 1 has no share of the projection above: the cost's constant term goes there, in the reference's summation order)
          double o = 0;
          for (int ax = 0; ax < 3; ax++) {
#pragma unroll
            for (int r = 0; r < NEP_MAX_POL; r++) { const double a = r < K ? sRhs[ax * 8 + r] : 0.0; if (r < K) o += 36 * T * a * a; }
            const double pe = sc[sPe0 + ax];
            o += wgt * pe * pe;
            if (mode == 1) {
              const double ve = tUvP[0] * sInit[ax * 3] + tUvP[1] * sInit[ax * 3 + 1] + tUvP[2] * sInit[ax * 3 + 2];
              const double ae = tUaP[0] * sInit[ax * 3] + tUaP[1] * sInit[ax * 3 + 1] + tUaP[2] * sInit[ax * 3 + 2];
              o += wgt * (ve * ve + ae * ae);
            }
          }
          sc[sObj0] = o;
          sI[16] = 0; sI[17] = 0; sI[22] = -1; sI[23] = 0; sI[24] = 0;
        }
        __syncthreads();
        auto proj = [&](int rho, int ax, const double* vec) {   // B's columns >= nz are zero and the vectors' tails are kept finite (zeroed below)
          const double* b = sB + rho * SBS; const double* x = vec + ax * nz;
          double v = 0;
#pragma unroll
          for (int c = 0; c < kNZ; c++) v = __builtin_fma(b[c], x[c], v);
          return v;
        };
        bool uncon = false;
        if constexpr (CULL) {   // presolve: the minimiser without inequality rows, accepted when every row holds there (see qp_kernel)
          if (tm < n) {
            const int ax = div_small(tm, nz), c = tm - ax * nz;
            double v = 0;
            for (int e = 0; e < nz; e++) v -= sM[tHi + c * kNZ + e] * sG[ax * nz + e];
            sDx[tm] = v;
          }
          __syncthreads();
          double viol = -1.0, o_share = 0, d1 = 0, d2 = 0;
          { int ax, rho; double hi, lo; if (box_role(ax, rho, hi, lo)) { const double a = sOff[rho * 3 + ax] + proj(rho, ax, sDx); viol = fmax(a - hi, lo - a); } }
          {
            const int lrho = otid() >> 3; const bool has_line = (lrho >> 2) < K;
            const double cx = has_line ? sOff[lrho * 3] + proj(lrho, 0, sDx) : 0.0, cy = has_line ? sOff[lrho * 3 + 1] + proj(lrho, 1, sDx) : 0.0;
            double du_[RS], dv_[RS]; isd_t dw_[RS];
            for_rows(du_, dv_, dw_, [&](bool ok, double n1, double n2, double h, double&, double&, isd_t&) { viol = fmax(viol, ok ? (n1 * cx + n2 * cy) - h : -1.0); });
          }
          if (tm == BS - 1 && has_qc) {
            double c = -0.10 * 0.10;
            for (int ax = 0; ax < 3; ax++) { double pe = sc[sPe0 + ax]; for (int e = 0; e < nz; e++) pe += sEp[e] * sDx[ax * nz + e]; c += pe * pe; }
            viol = fmax(viol, c);
          }
          if (tm < n) {
            const int ax = div_small(tm, nz), c = tm - ax * nz;
            double hz = 0;
            for (int e = 0; e < nz; e++) hz += sHax[c * kNZ + e] * sDx[ax * nz + e];
            o_share = sDx[tm] * (0.5 * hz + sG[tm]);
          }
          block_reduce4(viol, o_share, d1, d2, sRed);
          uncon = __builtin_amdgcn_readfirstlane(viol <= 0.0 ? 1 : 0) != 0;
          if (uncon) { if (tm < n) sZ[tm] = sDx[tm]; if (tm == 0) sc[sObj] = sc[sObj0] + o_share; }
        }
        // (roles are re-derived from an opaque copy of the thread index in every phase: whatever the compiler could hoist out of
        // the iteration loop — row / entry indices, LDS addresses of four different roles — would otherwise sit in registers
        // next to the row state for the whole solve)
        // row state (its lifetime is one solve of one mode): box rows [0] upper (alpha = +e), [1] lower; line rows: slot u = line slice + 8 u of my segment
        double bs0 = 1, bl0 = 0, bs1 = 1, bl1 = 0;
        double sl[RS], ll[RS]; isd_t il[RS];        // (il: 1 / s of the row, kept only with NEP_QP_CACHE_IS)
        isd_t bi0 = 1, bi1 = 1;
        double cpb = 0.0, uab = 0.0, udb = 0.0, cpx = 0.0, cpy = 0.0;
        double uax = 0.0, uay = 0.0, udx = 0.0, udy = 0.0;
        {
          int ax, rho; double hi, lo;
          if (box_role(ax, rho, hi, lo)) {
            cpb = sOff[rho * 3 + ax] + proj(rho, ax, sZ);
            double slk = hi - cpb; bs0 = slk > kSlackFloor ? slk : kSlackFloor; bl0 = kMu0 * frcp(bs0);
            slk = cpb - lo; bs1 = slk > kSlackFloor ? slk : kSlackFloor; bl1 = kMu0 * frcp(bs1);
          }
          const int lrho = otid() >> 3;
          if ((lrho >> 2) < K) { cpx = sOff[lrho * 3] + proj(lrho, 0, sZ); cpy = sOff[lrho * 3 + 1] + proj(lrho, 1, sZ); }
        }
#pragma unroll
        for (int u = 0; u < RS; u++) { sl[u] = 1.0; ll[u] = 1.0; il[u] = (isd_t)1.0; }
        for_rows(sl, ll, il, [&](bool ok, double n1, double n2, double h, double& s, double& lam, isd_t& isv) {
          const double slk = h - (n1 * cpx + n2 * cpy);
          s = slk > kSlackFloor ? slk : kSlackFloor; const double is_ = frcp(s); isv = (isd_t)is_; lam = kMu0 * is_;      // (a padded row starts at s = 1, lambda = mu0)
        });
        if (tm >= 64 && tm < 128) {   // the dual residual's scale: a maximum, whatever the order
          const double qs = fmax(1.0, wave_max(tm - 64 < n ? fabs(sG[tm - 64]) : 0.0));
          if (tm == 64) sc[sQscale] = qs;
        }
        if (tm == 0) {
          if (has_qc) {
            double c = -0.10 * 0.10;
            for (int ax = 0; ax < 3; ax++) { double pe = sc[sPe0 + ax]; for (int e = 0; e < nz; e++) pe += sEp[e] * sZ[ax * nz + e]; c += pe * pe; }
            const double sq = (-c > 1e-3) ? -c : 1e-3;
            sc[sSq] = sq; sc[sLq] = 1.0 / sq;
          } else { sc[sSq] = 1.0; sc[sLq] = 0.0; }
          sc[sDsq] = 0.0; sc[sDlq] = 0.0; sc[sDsqA] = 0.0; sc[sDlqA] = 0.0; sc[sRpq] = 0.0; sc[sWq] = 0.0;
          sc[sAlpha] = 0.0; sc[sSigMu] = 0.0;        // step length and centring target of the previous iteration (read back in pass A)
        }
        __syncthreads();

        double* redA = sRed; double* redP2 = sRed + 16; double* redP5 = sRed + 32;
        TICK(15);
        const int n0 = (has_qc && !NEP_QP_BALL_SM) ? n : 2 * nz;  // wave 0 factors this block, wave 1 the z block
        const bool qc_full = has_qc && !NEP_QP_BALL_SM;           // the ball row inside the matrix (coupled system on wave 0)
        const bool qc_sm = has_qc && NEP_QP_BALL_SM;              // ... or put back after the block solves
        double* sU = smem + oU; double* sUd = smem + oU + 24;     // D^-1 g; [0..1] g'u (xy, z), [2..3] g'y of the predictor, [4..5] of the corrector
        for (it = 0; it < kMaxIt && !uncon; it++) {
          // ---- (A1) apply the previous step to the row state; (A2) residuals / weights / scatter onto base rows.  Two sweeps
          // over the rows instead of one: the step needs the two direction projections, the scatter seven accumulators — together
          // they do not fit next to the row state in 128 registers, one after the other they do (the second sweep re-reads three
          // coefficients per row) ----
          double nrp = 0, sumsl = 0, dummy1 = 0;
          if (it > 0) {     // (the first iteration has no step to apply: alpha = 0 would leave every row as it is)
            const double alpha_prev = sc[sAlpha], sm_prev = sc[sSigMu];
            auto rowA1 = [&](double& s, double& lam, double a_old, double ga, double gd, double h, isd_t isc) {
              const double rp0 = a_old + s - h, is = is_of(s, isc), w0 = lam * is;
              const double dsa = -rp0 - ga, dla = -lam + w0 * (rp0 + ga);
              const double rcv = s * lam - sm_prev + dsa * dla;
              const double ds = -rp0 - gd, dl = -rcv * is + w0 * (rp0 + gd);
              s = __builtin_fma(alpha_prev, ds, s); lam = __builtin_fma(alpha_prev, dl, lam);
            };
            {
              int ax, rho; double hi, lo;
              if (box_role(ax, rho, hi, lo)) { rowA1(bs0, bl0, cpb, uab, udb, hi, bi0); rowA1(bs1, bl1, -cpb, -uab, -udb, -lo, bi1); }
            }
            for_rows(sl, ll, il, [&](bool ok, double n1, double n2, double h, double& s, double& lam, isd_t& isv) {
              rowA1(s, lam, n1 * cpx + n2 * cpy, n1 * uax + n2 * uay, n1 * udx + n2 * udy, h, isv);
            });
            cpb = __builtin_fma(alpha_prev, udb, cpb); cpx = __builtin_fma(alpha_prev, udx, cpx); cpy = __builtin_fma(alpha_prev, udy, cpy);   // base rows move with the step
          }
          {
            {
              int ax, rho; double hi, lo;
              if (box_role(ax, rho, hi, lo)) {
                double bTl = 0, bD = 0, bT1 = 0;
                { const double is_ = frcp(bs0); bi0 = (isd_t)is_; const double rp = cpb + bs0 - hi, w = bl0 * is_, v = bl0 - w * rp; nrp = fmax(nrp, fabs(rp)); sumsl += bs0 * bl0; bTl += bl0; bD += w; bT1 += v; }
                { const double is_ = frcp(bs1); bi1 = (isd_t)is_; const double rp = -cpb + bs1 + lo, w = bl1 * is_, v = bl1 - w * rp; nrp = fmax(nrp, fabs(rp)); sumsl += bs1 * bl1; bTl -= bl1; bD += w; bT1 -= v; }
                sTc[rho * 6 + ax] = bTl; sTc[rho * 6 + 3 + ax] = bT1; sDc[rho * 4 + (ax == 0 ? 0 : (ax == 1 ? 2 : 3))] = bD;
              }
            }
            double lTx = 0, lTy = 0, lDxx = 0, lDxy = 0, lDyy = 0, l1x = 0, l1y = 0;
            for_rows(sl, ll, il, [&](bool ok, double n1, double n2, double h, double& s, double& lam, isd_t& isv) {
              const double is_ = frcp(s); isv = (isd_t)is_;
              const double rp = (n1 * cpx + n2 * cpy) + s - h, w = lam * is_, v = lam - w * rp;
              nrp = fmax(nrp, fabs(rp)); sumsl += ok ? s * lam : 0.0;
              lTx += lam * n1; lTy += lam * n2; lDxx += w * n1 * n1; lDxy += w * n1 * n2; lDyy += w * n2 * n2; l1x += v * n1; l1y += v * n2;
            });
            lTx = slice_sum(lTx); lTy = slice_sum(lTy); lDxx = slice_sum(lDxx); lDxy = slice_sum(lDxy); lDyy = slice_sum(lDyy); l1x = slice_sum(l1x); l1y = slice_sum(l1y);
            const int t = otid();
            if ((t & 7) == 0) { double* o = sAccL + (t >> 3) * 8; o[0] = lTx; o[1] = lTy; o[2] = lDxx; o[3] = lDxy; o[4] = lDyy; o[5] = l1x; o[6] = l1y; }
          }
          reduce_put<1>(nrp, sumsl, dummy1, redA);
          __syncthreads();                                                                       // barrier 1
          TICK(0);
          // (the line rows' sums per control point stay in sAccL: what reads the per-base-row weights and right-hand sides below
          // adds them on the fly — base rows < 32 are the position control points — instead of a combining phase and its barrier)
          {
            const int t = otid();
            if (t == 0) {   // ball constraint (scalar row) and the iteration's scalars (the only reader of this reduction)
              reduce_get<1>(nrp, sumsl, dummy1, redA);
              double rpq = 0;
              if (has_qc) {
                sc[sSq] += sc[sAlpha] * sc[sDsq]; sc[sLq] += sc[sAlpha] * sc[sDlq];
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
              sc[sMu] = sc[sSumSl] * sc[sInvMt];
              sc[sNrp] = fmax(nrp, fabs(rpq));
            }
          }
          if (has_qc) __syncthreads();                                                           // barrier 2 (the ball row's gradient and weight feed the assembly)
          TICK(1);
          // ---- dual residual + predictor rhs, normal matrix: three 16 x 16 tiles of v_mfma_f64_16x16x4_f64 over the 64 base rows
          // (K = 64: 16 instructions per tile), one tile per wave.  Rows of a tile: ci (B' as the A operand; rows 8..15 repeat rows
          // 0..7 and are not read back).  Columns: wave 1: Tl(x, y, z), T1(x, y, z) -> B' Tl, B' T1, the two sums the residual and
          // the predictor's right-hand side need; wave 2: D_xx B | D_yx B; wave 3: D_yy B | D_zz B -> the blocks of M.  An output
          // entry depends on its own row of A and its own column of B only, so the padding rows / columns carry whatever the loads
          // return (in-bounds LDS reads) and cost no selects.  C/D layout: column = lane & 15, row = (lane >> 4) + 4 reg. ----
#if NEP_QP_MFMA
          if (tid >= 64) {
            typedef double v4d __attribute__((ext_vector_type(4)));
            const int t = otid();
            const int l = t & 63, i16 = l & 15, kk = l >> 4;
            const double* px = sB + kk * SBS + (i16 & 7);               // B[rho][i16 & 7]: the A operand, and the B operand's factor in the M tiles
            v4d acc0 = {0.0, 0.0, 0.0, 0.0}, acc1 = {0.0, 0.0, 0.0, 0.0};
            if (t >= 128) {
              const int cj = l & 7;
              const int sel = (t >= 192 ? 2 : 0) + (i16 >> 3);         // weight set of this lane's column: 0 xx, 1 yx, 2 yy, 3 zz
              const double* pd = sDc + kk * 4 + sel;
              const double* pl = sAccL + kk * 8 + (sel < 3 ? 2 + sel : 7);   // line rows' share of the weight (xx, yx, yy; entry 7 of a row is never written: zero)
              // four base rows' operands are loaded together ahead of their four instructions: one LDS round trip per group
              // instead of two per pair (the register allocator, at its limit next to the row state, otherwise reuses the same few
              // registers for every load)
#pragma unroll
              for (int g = 0; g < 4; g++) {
                double xg[4], dg[4];
#pragma unroll
                for (int u = 0; u < 4; u++) { xg[u] = px[(16 * g + 4 * u) * SBS]; dg[u] = pd[(16 * g + 4 * u) * 4]; }
                if (g < 2) {
#pragma unroll
                  for (int u = 0; u < 4; u++) dg[u] += pl[(16 * g + 4 * u) * 8];     // base rows < 32: the position control points
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int u = 0; u < 4; u += 2) {
                  acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(xg[u], dg[u] * xg[u], acc0, 0, 0, 0);
                  acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(xg[u + 1], dg[u + 1] * xg[u + 1], acc1, 0, 0, 0);
                }
              }
              const int bi_ = sel == 0 ? 0 : (sel == 3 ? 2 : 1), bj_ = sel == 0 || sel == 1 ? 0 : (sel == 2 ? 1 : 2);
#pragma unroll
              for (int r = 0; r < 2; r++) {
                const int ci = kk + 4 * r;
                if (ci < nz && cj < nz) {
                  double v = acc0[r] + acc1[r];
                  const int mi = bi_ * nz + ci, mj = bj_ * nz + cj;
                  if (sel != 1) v += sHax[ci * kNZ + cj];
                  if (has_qc) { if (qc_full) v += sc[sWq] * sGq[mi] * sGq[mj]; if (sel != 1) v += sc[sLq] * 2 * sEp[ci] * sEp[cj]; }
                  sM[mi * MS + mj] = v;
                }
              }
            } else {
              // column i16 < 6 of this tile: (Tl | T1)[ax], the box rows' sums from sTc plus, on base rows < 32 and for x and y, the line rows' from sAccL
              const int ax = i16 < 3 ? i16 : (i16 < 6 ? i16 - 3 : 0);
              const double* pt = sTc + kk * 6 + (i16 < 6 ? i16 : 0);
              const double* pl = sAccL + kk * 8 + (i16 < 2 ? i16 : ((i16 == 3 || i16 == 4) ? i16 + 2 : 7));
#pragma unroll
              for (int g = 0; g < 4; g++) {
                double xg[4], tg[4];
#pragma unroll
                for (int u = 0; u < 4; u++) { xg[u] = px[(16 * g + 4 * u) * SBS]; tg[u] = pt[(16 * g + 4 * u) * 6]; }
                if (g < 2) {
#pragma unroll
                  for (int u = 0; u < 4; u++) tg[u] += pl[(16 * g + 4 * u) * 8];
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int u = 0; u < 4; u += 2) {
                  acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(xg[u], tg[u], acc0, 0, 0, 0);
                  acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(xg[u + 1], tg[u + 1], acc1, 0, 0, 0);
                }
              }
              if (i16 < 6) {
#pragma unroll 1
                for (int r = 0; r < 2; r++) {
                  const int ci = kk + 4 * r;
                  if (ci < nz) {
                    const int o = ax * nz + ci;
                    double v = r == 0 ? acc0[0] + acc1[0] : acc0[1] + acc1[1];
                    if (i16 >= 3) sRhs[o] = v;
                    else {
                      double hz = 0;
                      { const double* hx = sHax + ci * kNZ; const double* zx = sZ + ax * nz;
#pragma unroll
                        for (int e = 0; e < kNZ; e++) hz += hx[e] * zx[e]; }
                      const double zo = sZ[o], go = sG[o];
                      v += go + hz;
                      if (has_qc) v += sc[sLq] * sGq[o];
                      sRd[o] = v;
                      sDx[o] = zo * (0.5 * hz + go);      // the objective's terms (sDx is free until the corrector solve)
                    }
                  }
                }
              }
            }
          }
#else
          // ---- dual residual + predictor rhs (8 partial sums per output), normal matrix -------------
          {
            const int t = otid();
            if (t < 8 * n) {
              const int o = t >> 3, sl8 = t & 7, ax = o / nz, c = o % nz;
              double v = 0, t1 = 0;
              const int axl = ax & 1;
#pragma unroll
              for (int q = 0; q < NEP_MAX_POL; q++) {     // (base rows >= 8 K: zero B, zero sums)
                const int rho = sl8 + 8 * q;
                double tl = sTc[rho * 6 + ax], tv = sTc[rho * 6 + 3 + ax];
                if (q < 4) { const double la = sAccL[rho * 8 + axl], lv = sAccL[rho * 8 + 5 + axl]; tl += ax < 2 ? la : 0.0; tv += ax < 2 ? lv : 0.0; }
                const double b = sB[rho * SBS + c]; v += b * tl; t1 += b * tv;
              }
              v = slice_sum(v); t1 = slice_sum(t1);
              if (sl8 == 0) {
                double hz = 0;
                { const double* hx = sHax + c * kNZ; const double* zx = sZ + ax * nz;
#pragma unroll
                  for (int e = 0; e < kNZ; e++) hz += hx[e] * zx[e]; }
                const double zo = sZ[o], go = sG[o];
                v += go + hz;
                if (has_qc) v += sc[sLq] * sGq[o];
                sRd[o] = v; sRhs[o] = t1;
                sDx[o] = zo * (0.5 * hz + go);      // (sDx is free until the corrector solve)
              }
            }
          }
#pragma unroll 1
          for (int u = 0; u < 2; u++) {
            const int t = otid();
            int mu_ = u == 0 ? me[0] : me[1]; asm volatile("" : "+v"(mu_));
            const int ci = mu_ & 15, cj = (mu_ >> 4) & 15, sel = (mu_ >> 8) & 3, half = t & 1;
            const bool on = (mu_ >> 12) != 0;
            double a0 = 0.0, a1 = 0.0;
            if (on) {
              for (int q = 0; q < K; q++) {
                const int rho = half + 8 * q;
                double d[4], bi[4], bj[4];
#pragma unroll
                for (int w = 0; w < 4; w++) { d[w] = sDc[(rho + 2 * w) * 4 + sel] + ((q < 4 && sel < 3) ? sAccL[(rho + 2 * w) * 8 + 2 + sel] : 0.0); bi[w] = sB[(rho + 2 * w) * SBS + ci]; bj[w] = sB[(rho + 2 * w) * SBS + cj]; }
                a0 = __builtin_fma(d[0] * bi[0], bj[0], a0); a1 = __builtin_fma(d[1] * bi[1], bj[1], a1);
                a0 = __builtin_fma(d[2] * bi[2], bj[2], a0); a1 = __builtin_fma(d[3] * bi[3], bj[3], a1);
              }
            }
            double v = a0 + a1;
            v += dpp<DPP_XOR1>(v);
            if (on && half == 0) {
              const int bi_ = sel == 0 ? 0 : (sel == 3 ? 2 : 1), bj_ = sel == 0 || sel == 1 ? 0 : (sel == 2 ? 1 : 2);
              const int mi = bi_ * nz + ci, mj = bj_ * nz + cj;
              if (sel != 1) v += sHax[ci * kNZ + cj];
              if (has_qc) { if (qc_full) v += sc[sWq] * sGq[mi] * sGq[mj]; if (sel != 1) v += sc[sLq] * 2 * sEp[ci] * sEp[cj]; }
              sM[mi * MS + mj] = v;
            }
          }
#endif
          if (qc_full) {   // z-x and z-y blocks: only the ball row couples z to x and y (without it the two diagonal blocks are factored apart and these entries are never read)
            for (int e = otid(); e < 2 * nz * nz; e += BS) {
              const int qi = div_small(e, 2 * nz), i = 2 * nz + qi, j = e - qi * 2 * nz;
              sM[i * MS + j] = sc[sWq] * sGq[i] * sGq[j];
            }
          }
          __syncthreads();                                                                       // barrier 3
          TICK(2);
          // ---- wave 0: Cholesky of its block and the predictor; wave 1: convergence test, then the z block.  Wave 0 does not
          // wait for the verdict: on the one iteration that ends the solve its factorisation is wasted, on all the others the test
          // (two wave reductions and a dozen dependent LDS reads) is off the longest chain of the iteration ----
          if (tid < 64) {
            const int t = otid();
            const lds_dptr sMl = (lds_dptr)(unsigned)(lds0 + oM * 8), sInvDl = (lds_dptr)(unsigned)(lds0 + oInvD * 8);
            const bool chol_ok = chol_n(sMl, sInvDl, n0, t);
            if (t == 0) sI[19] = chol_ok ? 1 : 0;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
            double b = 0;
            if (t < n0) { b = -sRd[t] + sRhs[t]; if (has_qc) b += sGq[t] * (sc[sLq] - sc[sWq] * sc[sRpq]); }
            b = solve_n(sMl, sInvDl, (lds_dptr)(unsigned)(lds0 + oRed * 8), n0, t, b);
            if (t < n0) sDxa[t] = b;
            if (qc_sm) {   // u = D^-1 g on this block, and the two dot products the rank-one correction needs
              const double gt = t < n0 ? sGq[t] : 0.0;
              const double u = solve_n(sMl, sInvDl, (lds_dptr)(unsigned)(lds0 + oRed * 8), n0, t, gt);
              if (t < n0) sU[t] = u;
              const double gu = wave_sum(t < n0 ? gt * u : 0.0), gy = wave_sum(t < n0 ? gt * b : 0.0);
              if (t == 0) { sUd[0] = gu; sUd[2] = gy; }
            }
            TICK(3);
          } else if (tid < 128) {
            const int t = otid(), l1 = t - 64;
            const double nrd = wave_max(l1 < n ? fabs(sRd[l1]) : 0.0);
            const double o = sc[sObj0] + wave_sum(l1 < n ? sDx[l1] : 0.0);      // (the objective's terms: written next to rd, see above)
            const double gap = sc[sMu] * sc[sMtD], nr = sc[sNrp], qs = sc[sQscale];
            int flag = 0;
            if (nr <= sp.tol_res && nrd <= sp.tol_res * qs && gap <= sp.tol_gap * (1.0 + fabs(o))) flag = 1;      // (1e-10, 1e-10, 1e-11 unless nep_batch_set_tolerances says otherwise)
            else if (nr <= 1e-6 && nrd <= 1e-6 * qs && gap <= 1e-7 * (1.0 + fabs(o))) flag = 2;
            if ((!CULL || PST) && l1 == 0) sI[28] = flag;         // (1: the STRICT tests passed — flag 1 below may also mean "the loose window ends on this iterate")
            if (!(sc[sMu] < 1e30) || !(nrd < 1e300)) flag = 3;  // diverged / NaN
            if (sI[17] >= 3) flag = 3;                           // stalled
            if (sp.time_limit_ticks > 0 && (long long)wall_clock64() - t_solve0 > sp.time_limit_ticks) { flag = 3; if ((!CULL || PST) && l1 == 0) sI[30] = 1; }   // TimeLimit without an accepted iterate: "no solution" (:832-836) — and out of budget: nothing is left for the polish pass either
            if (flag == 2 || (flag == 0 && sI[22] >= 0)) {       // loosely converged iterates: see qp_kernel
              const double merit = fmax(fmax(nr * sp.tol_res_inv, nrd / qs * sp.tol_res_inv), gap / (1.0 + fabs(o)) * sp.tol_gap_inv);
              const bool better = flag == 2 && (!sI[16] || merit < sc[sBestMerit]);
              const bool last = sI[22] >= 0 && it - sI[22] >= 3;
              if (sI[22] < 0 && l1 == 0) sI[22] = it;
              flag = last ? (better ? 1 : 3) : (better ? 2 : 0);
              if (l1 == 0 && flag == 2) sc[sBestMerit] = merit;
            }
            if (l1 == 0) { sc[sObj] = o; if (flag == 2) sc[sObjLoose] = o; sI[18] = flag; }
            bool chol_ok = true;
            if (!qc_full && flag != 1 && flag != 3) {             // (uniform across the wave)
              const lds_dptr sMz = (lds_dptr)(unsigned)(lds0 + (oM + 2 * nz * MS + 2 * nz) * 8), sInvDz = (lds_dptr)(unsigned)(lds0 + (oInvD + 2 * nz) * 8);
              chol_ok = chol_n(sMz, sInvDz, nz, l1);
              __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
              double b = l1 < nz ? -sRd[2 * nz + l1] + sRhs[2 * nz + l1] : 0.0;
              if (qc_sm && l1 < nz) b += sGq[2 * nz + l1] * (sc[sLq] - sc[sWq] * sc[sRpq]);
              b = solve_n(sMz, sInvDz, (lds_dptr)(unsigned)(lds0 + (oRed + 24) * 8), nz, l1, b);
              if (l1 < nz) sDxa[2 * nz + l1] = b;
              if (qc_sm) {
                const double gt = l1 < nz ? sGq[2 * nz + l1] : 0.0;
                const double u = solve_n(sMz, sInvDz, (lds_dptr)(unsigned)(lds0 + (oRed + 24) * 8), nz, l1, gt);
                if (l1 < nz) sU[2 * nz + l1] = u;
                const double gu = wave_sum(l1 < nz ? gt * u : 0.0), gy = wave_sum(l1 < nz ? gt * b : 0.0);
                if (l1 == 0) { sUd[1] = gu; sUd[3] = gy; }
              }
            }
            if (l1 == 0) sI[20] = chol_ok ? 1 : 0;
          }
          __syncthreads();                                                                       // barrier 4
          TICK(4);
          const int flag = __builtin_amdgcn_readfirstlane(sI[18]);
          if (flag == 1) { converged = true; break; }
          if (flag == 3) break;
          // (every thread notes the snapshot for itself: the flag word sI[16] is written by thread 0 only now, and a loop that ends in
          // this very iteration — a pivot lost in the factorisation below — used to read it back a few lines further down without a
          // barrier in between: a wave that got there before thread 0's store took the "not converged" path while wave 0 took the
          // other, and the workgroup's waves went on to different modes.  Rare — a breakdown in the iteration of the first loose
          // iterate: about one replan in 10^5 at config 5 — and found through the presolve's movement test)
          if (flag == 2) { have_loose = true; if (tid < n) sZl[tid] = sZ[tid]; if (tid == 0) sI[16] = 1; }
          if (!__builtin_amdgcn_readfirstlane(sI[19]) || !__builtin_amdgcn_readfirstlane(sI[20])) break;
          if (qc_sm) {   // the rank-one term back in: x = y - u (w g'y) / (1 + w g'u)
            const double wq = sc[sWq];
            const double cA = wq * (sUd[2] + sUd[3]) / (1.0 + wq * (sUd[0] + sUd[1]));
            if (tid < n) sDxa[tid] = __builtin_fma(-cA, sU[tid], sDxa[tid]);
            __syncthreads();
          }
          // ---- (P2) affine step: ratio test, mu_aff, corrector right-hand side split as va - sigma mu vb (see qp_kernel) ----
          // (nopred: this iteration repeats the previous one with its predictor discarded — kCorrMinStep, below: every pass
          // that re-derives the second-order term does so from uab / uax / uay, so zeroing the three is all it takes)
          const bool nopred = __builtin_amdgcn_readfirstlane(sI[23]) != 0;
          const int n_nopred = __builtin_amdgcn_readfirstlane(sI[24]);      // (how often so far)
          double rmax = 0, c2 = 0, dmy = 0;
          {
            auto rowP2 = [&](double s, double lam, double a, double ga, double h, double& va, double& vb, isd_t isc) {
              const double rp = a + s - h, is = is_of(s, isc), w = lam * is;
              const double dsa = -rp - ga, q = dsa * is, dla = -__builtin_fma(w, dsa, lam);
              rmax = fmax(rmax, fmax(-q, 1.0 + q));
              c2 += dsa * dla;
              va = __builtin_fma(q, dla, lam) - w * rp; vb = is;
            };
            {
              int ax, rho; double hi, lo;
              if (box_role(ax, rho, hi, lo)) {
                uab = nopred ? 0.0 : proj(rho, ax, sDxa);
                double va0, vb0, va1, vb1;
                rowP2(bs0, bl0, cpb, uab, hi, va0, vb0, bi0); rowP2(bs1, bl1, -cpb, -uab, -lo, va1, vb1, bi1);
                sTc[rho * 6 + 3 + ax] = va0 - va1; sTc[rho * 6 + ax] = vb0 - vb1;
              }
            }
            const int t = otid();
            { const int rho = t >> 3; uax = nopred ? 0.0 : proj(rho, 0, sDxa); uay = nopred ? 0.0 : proj(rho, 1, sDxa); }
            double vax = 0, vay = 0, vbx = 0, vby = 0;
            for_rows(sl, ll, il, [&](bool, double n1, double n2, double h, double& s, double& lam, isd_t& isv) {
              double va, vb;
              rowP2(s, lam, n1 * cpx + n2 * cpy, n1 * uax + n2 * uay, h, va, vb, isv);
              vax += va * n1; vay += va * n2; vbx += vb * n1; vby += vb * n2;
            });
            vax = slice_sum(vax); vay = slice_sum(vay); vbx = slice_sum(vbx); vby = slice_sum(vby);
            if ((t & 7) == 0) { double* o = sAccL + (t >> 3) * 8; o[5] = vax; o[6] = vay; o[0] = vbx; o[1] = vby; }
            if (t == BS - 1 && has_qc) {
              const double sq = sc[sSq], lq = sc[sLq], wq = sc[sWq], rpq = sc[sRpq];
              double gd = 0;
              if (!nopred) {
                // (with the rank-one term put back by the Sherman-Morrison formula g'x is known in closed form, g'y / (1 + w g'u):
                // summed from the corrected vector it would be the small difference of two large dot products)
                if (qc_sm) gd = (sUd[2] + sUd[3]) / (1.0 + wq * (sUd[0] + sUd[1]));
                else for (int e = 0; e < n; e++) gd += sGq[e] * sDxa[e];
              }
              const double dsq = -rpq - gd, dlq = -lq + wq * (rpq + gd);
              sc[sDsqA] = dsq; sc[sDlqA] = dlq;
              rmax = fmax(rmax, fmax(-dsq / sq, -dlq / lq));
              c2 += dsq * dlq;
            }
          }
          reduce_put<1>(rmax, c2, dmy, redP2);
          __syncthreads();                                                                       // barrier 5
          TICK(5);
          reduce_get<1>(rmax, c2, dmy, redP2);
          double sm;
          {
            const double aaff = rmax > 1.0 ? frcp2(rmax) : 1.0;
            const double mu = sc[sMu];
            const double inv_mt = sc[sInvMt];
            const double mua = ((1.0 - aaff) * sc[sSumSl] + aaff * aaff * c2) * inv_mt;
            const double rr = mua * frcp2(mu);
            sm = rr * rr * rr * mu;
            sm = fmax(sm, sp.tol_gap_floor * (1.0 + fabs(sc[sObj])) * inv_mt);
            if (nopred) sm = sc[sSigKeep];       // (sigma mu of the discarded predictor)
            else if (__builtin_amdgcn_readfirstlane((int)(it >= sp.corr_from_it && aaff < kCorrMinStep))) {
              // the affine step is too short for its second-order term to mean anything: repeat the iteration from the same point
              // (a step of length zero) with the predictor discarded.  Rare and late, so the repeated assembly does not matter.
              if (n_nopred >= sp.corr_max_count) break;                 // (not going to end: give this attempt up)
              sc[sAlpha] = 0.0; sc[sSigKeep] = sm; sI[23] = 1; sI[24] = n_nopred + 1;      // (every thread stores the same values)
              it--;
              continue;
            }
          }
          {
            const int t = otid();
            if (t < 8 * n) {   // corrector right-hand side
              const int o = t >> 3, sl8 = t & 7, ax = o / nz, c = o % nz;
              double t1 = 0;
#pragma unroll
              for (int q = 0; q < NEP_MAX_POL; q++) {     // (base rows >= 8 K: zero B, zero sums)
                const int rho = sl8 + 8 * q;
                double ta = sTc[rho * 6 + 3 + ax], tb2 = sTc[rho * 6 + ax];
                if (q < 4) {
                  const double la = sAccL[rho * 8 + 5 + (ax & 1)], lb = sAccL[rho * 8 + (ax & 1)];
                  const bool on = ax < 2 && rho < 4 * K;
                  ta += on ? la : 0.0; tb2 += on ? lb : 0.0;
                }
                t1 += sB[rho * SBS + c] * __builtin_fma(-sm, tb2, ta);
              }
              t1 = slice_sum(t1);
              if (sl8 == 0) sRhs[o] = t1;
            }
          }
          __syncthreads();                                                                       // barrier 6
          TICK(6);
          if (tid < 128) {
            const int t = otid();
            const bool w0 = t < 64;
            const int o = w0 ? t : 2 * nz + (t - 64);
            const bool mine = w0 ? t < n0 : (!qc_full && t - 64 < nz);
            double b = 0;
            if (mine) {
              b = -sRd[o] + sRhs[o];
              if (has_qc) { const double sq = sc[sSq], lq = sc[sLq]; const double rcq = sq * lq - sm + sc[sDsqA] * sc[sDlqA]; b += sGq[o] * (rcq / sq - sc[sWq] * sc[sRpq]); }
            }
            if (w0) {
              const lds_dptr sMl = (lds_dptr)(unsigned)(lds0 + oM * 8), sInvDl = (lds_dptr)(unsigned)(lds0 + oInvD * 8);
              b = solve_n(sMl, sInvDl, (lds_dptr)(unsigned)(lds0 + oRed * 8), n0, t, b); if (mine) sDx[o] = b;
            } else if (!qc_full) {
              const lds_dptr sMz = (lds_dptr)(unsigned)(lds0 + (oM + 2 * nz * MS + 2 * nz) * 8), sInvDz = (lds_dptr)(unsigned)(lds0 + (oInvD + 2 * nz) * 8);
              b = solve_n(sMz, sInvDz, (lds_dptr)(unsigned)(lds0 + (oRed + 24) * 8), nz, t - 64, b); if (mine) sDx[o] = b;
            }
            if (qc_sm) { const double gy = wave_sum(mine ? sGq[o] * b : 0.0); if ((t & 63) == 0) sUd[4 + (t >> 6)] = gy; }
          }
          __syncthreads();                                                                       // barrier 7
          if (qc_sm) {
            const double wq = sc[sWq];
            const double cB = wq * (sUd[4] + sUd[5]) / (1.0 + wq * (sUd[0] + sUd[1]));
            if (tid < n) sDx[tid] = __builtin_fma(-cB, sU[tid], sDx[tid]);
            __syncthreads();
          }
          TICK(7);
          // ---- (P5) step length of the combined direction ------------------------------------------
          rmax = 0;
          {
            auto rowP5 = [&](bool ok, double s, double lam, double a, double ga, double gd, double h, isd_t isc) {
              const double rp = a + s - h, is = is_of(s, isc), w = lam * is;
              const double dsa = -rp - ga, dla = -lam + w * (rp + ga);
              const double rcv = s * lam - sm + dsa * dla;
              const double ds = -rp - gd, dl = -rcv * is + w * (rp + gd);
              rmax = fmax(rmax, ok ? fmax(-ds * is, -dl * frcp(lam)) : 0.0);
            };
            {
              int ax, rho; double hi, lo;
              if (box_role(ax, rho, hi, lo)) { udb = proj(rho, ax, sDx); rowP5(true, bs0, bl0, cpb, uab, udb, hi, bi0); rowP5(true, bs1, bl1, -cpb, -uab, -udb, -lo, bi1); }
            }
            const int t = otid();
            { const int rho = t >> 3; udx = proj(rho, 0, sDx); udy = proj(rho, 1, sDx); }
            for_rows(sl, ll, il, [&](bool ok, double n1, double n2, double h, double& s, double& lam, isd_t& isv) { rowP5(ok, s, lam, n1 * cpx + n2 * cpy, n1 * uax + n2 * uay, n1 * udx + n2 * udy, h, isv); });
            if (t == BS - 1 && has_qc) {
              const double sq = sc[sSq], lq = sc[sLq], wq = sc[sWq], rpq = sc[sRpq];
              double gd = 0;
              if (qc_sm) gd = (sUd[4] + sUd[5]) / (1.0 + wq * (sUd[0] + sUd[1]));
              else for (int e = 0; e < n; e++) gd += sGq[e] * sDx[e];
              const double rcq = sq * lq - sm + sc[sDsqA] * sc[sDlqA];
              const double dsq = -rpq - gd, dlq = -rcq / sq + wq * (rpq + gd);
              sc[sDsq] = dsq; sc[sDlq] = dlq;
              rmax = fmax(rmax, fmax(-dsq / sq, -dlq / lq));
            }
          }
          reduce_put<0>(rmax, dmy, dmy, redP5);
          __syncthreads();                                                                       // barrier 8
          TICK(8);
          reduce_get<0>(rmax, dmy, dmy, redP5);
          {
            double alpha = rmax > 0.0 ? frcp2(rmax) : 1e30;
            alpha = fmin(1.0, fmin(fmax(1.0 - sc[sMu], kStepFracMin), kStepFracMax) * alpha);
            const int t = otid();
            if (t == 0) { if (alpha < 1e-8) sI[17]++; else sI[17] = 0; sI[23] = 0; }
            sc[sAlpha] = alpha; sc[sSigMu] = sm;     // (every thread stores the same two values: pass A reads them back without a barrier in between)
            if (t < n) sZ[t] += alpha * sDx[t];
          }
          TICK(9);
        }
        if (uncon) converged = true;
        if (!converged && have_loose) { __syncthreads(); if (tid < n) sZ[tid] = sZl[tid]; if (tid == 0) sc[sObj] = sc[sObjLoose]; converged = true; }
        {
          // a solve that ends without the strict tests (the loose snapshot, or no convergence at all) on a problem without the ball row
          // leaves its last point for the active-set polish (qp_polish_kernel.hip), which finishes it exactly or leaves it alone.
          // Under the presolve: the first attempt's solves only (near lines; the polish pass re-verifies the parked lines and the
          // movement bound of the skipped LPs before it accepts a point) — a second attempt's rows are not what the pass would stage
          // (the presolve's instantiation keeps no per-iteration note of the strict tests — its registers are short as it is: a solve
          // counts as loose there when the loose window was ever opened (sI[22]; a strict pass inside the window is polished too, which
          // changes nothing but the last digits), and the TimeLimit is read off the clock again)
          // (round 6: the presolve's instantiation with the hooks keeps the per-iteration note of the strict tests too — sI[28], one LDS
          // store by one lane — so that ONLY solves that really ended loose or stalled are listed: with "the loose window was ever opened"
          // as the criterion, half of a closed loop's replans were listed, 3 837 of 8 192 per step, and the pass took 0.4 ms)
          bool leave;
          if constexpr (CULL && !PST) leave = false;
          else leave = !(CULL && use_far) && !(converged && __builtin_amdgcn_readfirstlane(sI[28]) == 1) && __builtin_amdgcn_readfirstlane(sI[30]) == 0;
          if ((!CULL || PST) && ps.polish_z && !has_qc && !uncon && leave) {
            __syncthreads();
            if (tid < n) ps.polish_z[((long)slot * 2 + mode) * 24 + tid] = sZ[tid];
            if (tid == 0) sI[29] |= 1 << mode;
          }
        }
      }
      if (tid == 0) { sI[31] = it; if (mode == 0) sI[32] = it; }
      __syncthreads();
      if (converged) {
        status = mode;   // NEP_OK / NEP_RELAXED
        if (tid == 0) sc[sObjOut] = sc[sObj];
        if (tid < 12 * K) {  // theta = Th z + ThU init
          const int ax = div_small(tid, 4 * K), r = tid - ax * 4 * K;
          double v = tb->ThU[r][0] * sInit[ax * 3] + tb->ThU[r][1] * sInit[ax * 3 + 1] + tb->ThU[r][2] * sInit[ax * 3 + 2];
          double th[kNZ];
#pragma unroll
          for (int c = 0; c < kNZ; c++) th[c] = tb->Th[r][c];
#pragma unroll
          for (int c = 0; c < kNZ; c++) v += c < nz ? th[c] * sZ[ax * nz + c] : 0.0;
          sTheta[(ax * 8 + r / 4) * 4 + (r % 4)] = v;
        }
        break;
      }
    }
    __syncthreads();
    if (!CULL) break;
    if (use_far || status == NEP_FAILED) break;
    const int n_skip = __builtin_amdgcn_readfirstlane(sI[40]);
    if (__builtin_amdgcn_readfirstlane(sI[41]) == 0 && n_skip == 0) break;
    {  // the far lines against the solution: its position control points, from the coefficients it returns (the MINVO matrix applied
       // to sTheta: what a caller evaluating the returned trajectory sees)
      if (tid < 8 * K) {
        const int rho = tid >> 1, ax = tid & 1, sg = rho >> 2, k = rho & 3;
        const double c0 = (T * T * T) * cQpAPosInv[0][k], c1 = (T * T) * cQpAPosInv[1][k], c2 = T * cQpAPosInv[2][k], c3 = cQpAPosInv[3][k];
        const double* Q = sTheta + (ax * 8 + sg) * 4;
        const double v = ((Q[0] * c0 + Q[1] * c1) + Q[2] * c2) + Q[3] * c3;
        sAccL[rho * 2 + ax] = v;
        if (n_skip > 0) {
          // The LPs the separator skipped have lines farther than cull_radius from every control point of the GUESS (their
          // point sets' boxes are that far apart and the box sides are polygon edges: separator_body).  A solution control point
          // that stays within cull_radius of the guess's — a point of the guess's control polygon — is therefore on the right
          // side of every one of them: nothing to evaluate.  One that moved farther sends the replan to the redo pass.
          const double* P = sCoef + (ax * 8 + sg) * 4;
          const double gq = ((P[0] * c0 + P[1] * c1) + P[2] * c2) + P[3] * c3;
          double d2 = (v - gq) * (v - gq);
          d2 += dpp<DPP_XOR1>(d2);                        // (x and y of a control point sit on neighbouring lanes)
          if (d2 > sp.cull_radius * sp.cull_radius) sI[25] = 1;
        }
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
      const bool v_far = __builtin_amdgcn_readfirstlane(sI[21]) != 0, v_move = __builtin_amdgcn_readfirstlane(sI[25]) != 0;
      if (!v_far && !v_move) break;
      if (n_skip > 0 || (ps.scratch_chunks > 0 && ps.redo_count)) {      // (pooled scratch: the second attempt's rows would not fit the slots — the redo pass has the room)
        // lines are missing from the buckets, so the full problem cannot be posed here: the redo pass solves every LP of this
        // replan and every row (separator_redo_kernel, then this kernel without the presolve); the solution written below is
        // overwritten by that pass whatever its outcome, but the COMMIT record is not written here at all (sI[26]): if the redo
        // solve fails the slot must keep the record it held, not this unverified trajectory marked valid
        if (tid == 0 && ps.redo_count) { sI[26] = 1; const int idx = atomicAdd(ps.redo_count, 1); ps.redo_list[idx] = slot; if (v_far) atomicAdd(ps.redo_count + 1, 1); if (v_move) atomicAdd(ps.redo_count + 2, 1); }   // ([1], [2]: by reason, for the test hook)
        break;
      }
      for (int e = tid; e < 256; e += BS) sAccL[e] = 0.0;   // (the accumulators were borrowed: the second attempt starts from zeros again)
    }
  }
  __syncthreads();
  const int Ko = K_ok ? K : 0;
  // ---- outputs (as qp_kernel) ---------------------------------------------------------------------
  if (status == NEP_FAILED) { if (tid < 96) sTheta[tid] = sCoef[tid]; }                    // :856-859
  else if (z_override) { if (tid < 32) sTheta[64 + tid] = sCoef[64 + tid]; }             // :879-880
  __syncthreads();
  if (tid < 96) (&sol->coeff[0][0][0])[tid] = ((tid % 32) / 4 < Ko) ? sTheta[tid] : 0.0;
  if (tid <= NEP_MAX_POL) sol->times[tid] = (tid <= Ko) ? g->t_start + tid * T : 0.0;
  const int ns_all = sched.n[Ko];
  const int ns = ns_all < sp.max_states ? ns_all : sp.max_states;
  if (tid == 0) {
    sol->stats.status = status; sol->stats.iters = sI[31]; sol->stats.iters_first = sI[32];
    int n_lp = 0, n_lpf = 0;
    if (ps.lp_stats && !ps.lines_override) {
      int v[2 * NEP_MAX_POL];
#pragma unroll
      for (int i = 0; i < 2 * NEP_MAX_POL; i++) v[i] = ps.lp_stats[(long)slot * NEP_MAX_POL * 2 + i];
#pragma unroll
      for (int i = 0; i < NEP_MAX_POL; i++) { n_lp += v[2 * i]; n_lpf += v[2 * i + 1]; }
    }
    sol->stats.n_lines = L_all - n_lpf; sol->stats.n_lp = n_lp; sol->stats.n_lp_failed = n_lpf;
    sol->stats.n_rows = K_ok ? 48 * K + 4 * ((CULL && L_used < L_all) ? L_used : L_used - n_lpf) : 0; sol->stats.qc_active = has_qc ? 1 : 0;
    sol->stats.objective = sc[sObjOut]; { const long long dt_ = (long long)wall_clock64() - t_wg0; const double us_ = (double)dt_ * sp.us_per_tick; sol->stats.solve_us = us_; if (ps.order_key) { const double k_ = us_ * 0.125; const int kn = k_ > 63.0 ? 63 : (int)k_, ko = ps.order_key[slot] - sp.qp_key_decay; ps.order_key[slot] = (sp.qp_key_decay > 0 && ko > kn) ? ko : kn; } }   // the per-replan device time, and the next launch's ordering key (8 us bins)
    sol->K = Ko; sol->n_states = ns;
    if (ps.polish_list) {
      const bool listed = sI[29] != 0 && !(CULL && sI[26] != 0);      // (a replan sent to the redo pass is listed there, if at all)
      // (a launch of one workgroup — the per-agent handle's — is its own list: no counter to zero beforehand, no kernel to do it)
      if (gridDim.x == 1) { ps.polish_count[0] = listed ? 1 : 0; ps.polish_count[3] = 0; if (listed) { ps.polish_flag[slot] = sI[29]; ps.polish_list[0] = slot; } }
      else if (listed) { ps.polish_flag[slot] = sI[29]; ps.polish_list[atomicAdd(ps.polish_count, 1)] = slot; }
    }
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
  if (prof && tid == 0) {   // [0..9] loop phases, [10] workgroup lifetime, [12] iterations, [13] line gather, [14] mode staging, [15] start point
    for (int k = 0; k < 16; k++) ps.dbg[(long)slot * 16 + k] = sProf[k];
    ps.dbg[(long)slot * 16 + 10] = clock64() - tstart; ps.dbg[(long)slot * 16 + 12] = sI[31];
    ps.dbg[(long)slot * 16 + 11] = (tstart_wall << 20) | ((long long)wall_clock64() - tstart_wall);   // start (100 MHz ticks) << 20 | duration
  }
#endif
  if (ps.commit) {  // the record the agent would publish (neptune_ros.cpp:434-480); a failed replan publishes nothing (see qp_kernel)
    nep_traj_rec* cr = ps.commit + slot;
    const int own = sp.first_local + (slot % sp.n_local);
    if (status == NEP_FAILED || (CULL && __builtin_amdgcn_readfirstlane(sI[26]) != 0)) {
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

constexpr int kRegSlots = NEP_QP_REG_SLOTS;

int qp_reg_slots() { return kRegSlots; }
// dynamic LDS of the register kernel: the fixed carve + the coefficient carve [3][NEP_MAX_POL][8 slots]
size_t qp_reg_lds_bytes() {
  size_t b = (size_t)kFixedDoubles * 8 + 64 * 4 + (size_t)3 * NEP_MAX_POL * 8 * kRegSlots * sizeof(double);
#ifdef NEP_PROFILE_PHASES
  b += 16 * sizeof(long long);
#endif
  return b;
}

void launch_qp_reg(int n_slots, const SceneParams& sp, const ProblemSet& ps, const QpTable* tables,
                   const SampleSched& sched, size_t lds_bytes, hipStream_t st) {
  if (n_slots <= 0) return;
  const bool cull = ps.line_far != nullptr && !ps.lines_override;
  const int which = !cull ? 0 : (ps.polish_z ? 2 : 1);
  static DynLdsAttr attr[3];
  (void)attr[which].ensure(which == 0 ? (const void*)qp_reg_kernel<false, kRegSlots> : which == 1 ? (const void*)qp_reg_kernel<true, kRegSlots, false> : (const void*)qp_reg_kernel<true, kRegSlots, true>, lds_bytes);
  if (which == 2) hipLaunchKernelGGL((qp_reg_kernel<true, kRegSlots, true>), dim3(n_slots), dim3(BS), lds_bytes, st, sp, ps, tables, sched);
  else if (which == 1) hipLaunchKernelGGL((qp_reg_kernel<true, kRegSlots, false>), dim3(n_slots), dim3(BS), lds_bytes, st, sp, ps, tables, sched);
  else hipLaunchKernelGGL((qp_reg_kernel<false, kRegSlots>), dim3(n_slots), dim3(BS), lds_bytes, st, sp, ps, tables, sched);
}

}  // namespace nep
