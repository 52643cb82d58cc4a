// geom_kernels.hip — MINVO control-point hulls and separating-line LPs for gfx950.
//
// Built with -ffp-contract=off: these kernels are compared BIT FOR BIT with the CPU oracle, so
// the compiler must not fuse multiply-adds (the expressions below are written in the same
// association order as the CPU checker under oracle/).  fp64 division and sqrt are IEEE-correct on
// CDNA4, so with contraction off the two sides produce identical bits.
//
// K1/K2 (reference neptune/src/neptune.cpp:269-452 + cgal_utils.cpp:157-174): one wavefront per
//   (scene, committed trajectory, planning interval) hull: rank sort across the 64 lanes, then
//   a monotone chain walked by lane 0 out of LDS.
// K3/K4 (reference submodules/separator/src/separator_glpk.cpp:248-498 called from
//   solver_gurobi_poly.cpp:477-495,521-553,556-593,715-764): one wavefront per (agent slot,
//   segment); each lane owns one candidate obstacle, stages its points in LDS (stride 17
//   doubles: conflict-free for ds_read_b64) and enumerates the LP's vertices; results are
//   compacted in reference loop order with a wave ballot.
#include <hip/hip_runtime.h>

#include <cstdlib>
#include <cstring>

#include "nep_device.h"
#include "../../include/neptune_frontend.h"
#include "ent_device.h"

namespace nep {

__constant__ double cAPosInv[4][4] = {
    {-0.03203276669713047, -0.09273093424558249, 0.3420572455666699, 1.1023313949144335},
    {-0.05111494245568798, -0.046272612998418894, 0.5458234872124772, 1.0979806946005568},
    {-0.07454781852812224, 0.203951949894552, 0.796048050105448, 1.0745478185281223},
    {1.0, 1.0, 0.9999999999999996, 0.9999999999999993}};

#define SEP_MIN_GAP 1e-7
#define NEP_INF (__builtin_huge_val())

// ---------------------------------------------------------------------------------------------
// hulls
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ double cross3(double ox, double oy, double ax, double ay, double bx, double by) {
  return (ax - ox) * (by - oy) - (ay - oy) * (bx - ox);
}
__device__ __forceinline__ bool lex_less(double ax, double ay, double bx, double by) {
  return ax < bx || (ax == bx && ay < by);
}

// Wave-cooperative convex hull of n <= 64 points held one per lane (px,py valid for lane<n).
// sxy: LDS [64][2] (sorted points), hxy: LDS [132][2] (hull + the upper chain's stack).  Returns
// the vertex count (uniform).  Rank sort and duplicate removal across the lanes, then Andrew's
// monotone chain with the lower chain on lane 0 and the upper chain on lane 1 at the same time.
// The two chains never test a triple that spans both (the reference formulation guards the upper
// chain's pops with k >= lo), so each lane evaluates exactly the cross products the serial walk
// would, in the same order — the output equals the one-stack walk bit for bit.  The top two stack
// entries are cached in registers (one LDS read per pop instead of four per test).
__device__ int wave_hull(int n, double px, double py, double* sxy, double* hxy) {
  const int lane = threadIdx.x & 63;
  // rank sort (lexicographic; ties by lane index so that ranks are a permutation)
  sxy[2 * lane] = px; sxy[2 * lane + 1] = py;
  __syncthreads();
  int rank = 0;
  if (lane < n) {
    for (int j = 0; j < n; j++) {
      const double qx = sxy[2 * j], qy = sxy[2 * j + 1];
      if (lex_less(qx, qy, px, py) || (qx == px && qy == py && j < lane)) rank++;
    }
  }
  __syncthreads();
  if (lane < n) { sxy[2 * rank] = px; sxy[2 * rank + 1] = py; }
  __syncthreads();
  // unique: sorted position `lane` survives if it differs from its predecessor; new position = number of survivors before it
  double ux = 0, uy = 0; bool keep = false;
  if (lane < n) { ux = sxy[2 * lane]; uy = sxy[2 * lane + 1]; keep = lane == 0 || ux != sxy[2 * lane - 2] || uy != sxy[2 * lane - 1]; }
  const unsigned long long km = __ballot(keep);
  const int m = __popcll(km);
  __syncthreads();
  if (keep) { const int pos = __popcll(km & ((1ull << lane) - 1ull)); sxy[2 * pos] = ux; sxy[2 * pos + 1] = uy; }
  __syncthreads();
  if (m == 0) return 0;
  if (m == 1) { if (lane == 0) { hxy[0] = sxy[0]; hxy[1] = sxy[1]; } __syncthreads(); return 1; }
  double* hu = hxy + 132;           // upper chain's stack (the lower chain's is hxy itself)
  int kc = 0;
  if (lane < 2) {
    double* st = lane == 0 ? hxy : hu;
    double ax = 0, ay = 0, bx = 0, by = 0;   // st[kc-2], st[kc-1]
    for (int i = 0; i < m; i++) {
      const int idx = lane == 0 ? i : m - 1 - i;
      const double x = sxy[2 * idx], y = sxy[2 * idx + 1];
      while (kc >= 2 && cross3(ax, ay, bx, by, x, y) <= 0.0) { kc--; bx = ax; by = ay; if (kc >= 2) { ax = st[2 * (kc - 2)]; ay = st[2 * (kc - 2) + 1]; } }
      st[2 * kc] = x; st[2 * kc + 1] = y; kc++;
      ax = bx; ay = by; bx = x; by = y;
    }
  }
  const int kl = __shfl(kc, 0), ku = __shfl(kc, 1);
  __syncthreads();
  // hull = lower chain, then the upper chain without its two end points (they are the lower chain's ends)
  if (lane >= 1 && lane < ku - 1) { hxy[2 * (kl + lane - 1)] = hu[2 * lane]; hxy[2 * (kl + lane - 1) + 1] = hu[2 * lane + 1]; }
  __syncthreads();
  return kl + ku - 2;
}

// Body shared by the batched and the stand-alone hull kernels: one wave computes the inflated
// and the uninflated hull of trajectory r over [ts + i T, ts + (i+1) T].  neptune.cpp:349-452.
__device__ void hull_body(const nep_traj_rec* __restrict__ r, double ts, int i, double T_span, double drone_radius,
                          long out, bool full0, double* __restrict__ hull_xy, int* __restrict__ hull_nv,
                          double* __restrict__ hull0_xy, int* __restrict__ hull0_nv, int* __restrict__ flags) {
  __shared__ __attribute__((aligned(16))) double sxy[128], hxy[264];   // hxy: hull [66][2] + upper-chain stack [66][2]
  __shared__ double cpx[kHullCP], cpy[kHullCP], stimes[NEP_TRAJ_MAX_SEG + 2];
  const int lane = threadIdx.x;
  if (!(r->valid && r->is_agent) || r->pwp.n_seg <= 0) {   // neptune.cpp:244-262, 332
    if (lane == 0) { hull_nv[out] = 0; if (hull0_nv) hull0_nv[out] = 0; }
    return;
  }
  const double t0 = ts + i * T_span, t1 = ts + (i + 1) * T_span;   // neptune.cpp:273-280
  const int n = r->pwp.n_seg;
  // std::lower_bound / upper_bound on the sorted knot vector (neptune.cpp:379-389) as two ballots
  const bool inr = lane <= n;
  const double tk = inr ? r->pwp.times[lane] : 0.0;
  if (inr) stimes[lane] = tk;
  int first = __popcll(__ballot(inr && tk < t0)) - 1;
  int last = __popcll(__ballot(inr && tk <= t1)) - 1;
  if (first < 0) first = 0; if (first > n - 1) first = n - 1;
  if (last < 0) last = 0; if (last > n - 1) last = n - 1;
  int nseg = last - first + 1; if (nseg < 0) nseg = 0;
  // The reference takes every committed segment that overlaps the interval (neptune.cpp:392-449); one wave holds the 4 x 4
  // inflated control points of NEP_HULL_MAX_CP / 4 segments.  More than that (a trajectory composed of several intervals
  // shorter than T_span) is a capacity overflow: flagged (NEP_E_CAP from nep_batch_check / nep_hulls_batch), never
  // silently under-covered.
  if (nseg > kHullCP / 4) { nseg = kHullCP / 4; if (lane == 0 && flags) atomicOr(flags, NEP_FLAG_HULL_OVERFLOW); }
  const int np0 = 4 * nseg;
  __syncthreads();
  if (lane < 2 * np0) {   // one lane per (segment, control point, axis)
    const int sl = lane >> 3, k = (lane >> 1) & 3, ax = lane & 1;
    const int s = first + sl;
    double _t;                                                     // neptune.cpp:399-424
    if (s != last) _t = stimes[s + 1] - stimes[s];
    else if (t1 > stimes[s + 1]) _t = stimes[s + 1] - stimes[s];
    else _t = t1 - stimes[s];
    if (_t > T_span) _t = T_span; else if (_t < 0) _t = 0;
    const double c0 = _t * _t * _t, c1 = _t * _t, c2 = _t, c3 = 1.0;
    const double* P = r->pwp.coeff[ax][s];                          // V = (P*C)*A^-1, :426-429
    const double v = (((P[0] * c0) * cAPosInv[0][k] + (P[1] * c1) * cAPosInv[1][k]) + (P[2] * c2) * cAPosInv[2][k]) + (P[3] * c3) * cAPosInv[3][k];
    if (ax == 0) cpx[sl * 4 + k] = v; else cpy[sl * 4 + k] = v;
  }
  __syncthreads();
  const double dx = r->bbox[0] / 2.0 + drone_radius, dy = r->bbox[1] / 2.0 + drone_radius;  // neptune.cpp:340
  const bool inflate = !(sqrt(dx * dx + dy * dy) < 1e-6);
  // inflated points: 4 corners per control point (:442-445)
  const int np = inflate ? 4 * np0 : np0;
  double px = 0, py = 0;
  if (lane < np) {
    if (inflate) {
      const int c = lane >> 2, q = lane & 3;
      const double x = cpx[c], y = cpy[c];
      px = (q < 2) ? x + dx : x - dx;
      py = (q == 0 || q == 3) ? y + dy : y - dy;
    } else { px = cpx[lane]; py = cpy[lane]; }
  }
  int k = wave_hull(np, px, py, sxy, hxy);
  if (k > kHullV) { k = kHullV; if (lane == 0 && flags) atomicOr(flags, NEP_FLAG_HULL_OVERFLOW); }
  if (lane < k) { hull_xy[(out * kHullV + lane) * 2] = hxy[2 * lane]; hull_xy[(out * kHullV + lane) * 2 + 1] = hxy[2 * lane + 1]; }
  if (lane == 0) hull_nv[out] = k;
  if (hull0_nv) {
    __syncthreads();
    px = (lane < np0) ? cpx[lane] : 0; py = (lane < np0) ? cpy[lane] : 0;
    int k0 = wave_hull(np0, px, py, sxy, hxy);
    if (k0 > kHullV) k0 = kHullV;
    if (full0) {
      if (lane < k0) { hull0_xy[(out * kHullV + lane) * 2] = hxy[2 * lane]; hull0_xy[(out * kHullV + lane) * 2 + 1] = hxy[2 * lane + 1]; }
    } else if (lane == 0) { hull0_xy[out * 2] = hxy[0]; hull0_xy[out * 2 + 1] = hxy[1]; }  // only col(0) is read (:722-734)
    if (lane == 0) hull0_nv[out] = k0;
  }
}

// One block (= one wave) per (scene, committed trajectory, planning interval).
__global__ __launch_bounds__(64) void hull_kernel(const nep_traj_rec* __restrict__ recs, int n_rec_per_scene,
                                                  const double* __restrict__ ts0, long ts_scene_stride,
                                                  int num_pol, double T_span, double drone_radius,
                                                  double* __restrict__ hull_xy, int* __restrict__ hull_nv,
                                                  double* __restrict__ hull0_xy, int* __restrict__ hull0_nv,
                                                  double* __restrict__ bend_xy, int* __restrict__ bend_n, int* __restrict__ flags) {
  const int lane = threadIdx.x;
  const int i = blockIdx.x % num_pol;
  const int jt = blockIdx.x / num_pol;          // scene*n_rec + j
  const int scene = jt / n_rec_per_scene;
  const nep_traj_rec* r = recs + jt;
  if (lane == 0 && i == 0 && bend_n) {          // bend points travel with the record (neptune_ros.cpp:457-476)
    int nb = r->n_bend; if (nb > kBend) nb = kBend; if (nb < 0) nb = 0;
    bend_n[jt] = (r->valid && r->is_agent) ? nb : 0;
    for (int b = 0; b < nb; b++) { bend_xy[((long)jt * kBend + b) * 2] = r->bend[b][0]; bend_xy[((long)jt * kBend + b) * 2 + 1] = r->bend[b][1]; }
  }
  // bulk-synchronous round: every agent of a scene replans from the same t_start (that of its first local slot)
  const double ts = *(const double*)((const char*)ts0 + (long)scene * ts_scene_stride);
  hull_body(r, ts, i, T_span, drone_radius, (long)jt * num_pol + i, false, hull_xy, hull_nv, hull0_xy, hull0_nv, flags);
}

// ---- the batched hulls, eight to a wave -----------------------------------------------------------
// hull_kernel above spends most of its instructions with two active lanes (the two monotone chains).  Here one wave takes
// ALL planning intervals of one committed trajectory: interval i is worked by the eight lanes 8 i .. 8 i + 7, each holding up
// to eight of the interval's (<= 64) inflated control points.  Same algorithm per hull — lexicographic sort (a bitonic network since
// round 6), duplicate removal, the two chains with the same cross products in the same order — so the output is the same bits; per
// wave the chains of eight hulls advance together and the sort's compare-exchanges are spread over all 64 lanes.
constexpr int kGrpPts = 8;                         // points per lane
constexpr int kGrpSxy = 2 * 64;                    // doubles per group: the sorted points

// Hull of the group's n points (point p = sub + 8 j in px[j], py[j]); sxy: this group's LDS.  Every lane of the wave must call
// it (workgroup barriers inside).  A chain's stack is a subsequence of the sorted points, so it is kept as a 64-bit mask over
// their indices in the chain lane's registers (push = set a bit, pop = clear the top one, the entry under the top = the next
// set bit) instead of two coordinate stacks in LDS: a wave of eight hulls needs 10 KB instead of 27, and three times as many
// waves fit a CU.  Writes the first min(k, cap) vertices to out_xy (lower chain, then the upper chain without its two end
// points: the lower chain's ends) and returns k.
// Lexicographic sort of the group's points as a bitonic network over 8 lanes x J registers (round 6; until then a rank sort: every
// point against every point, 64 x 64 lexicographic comparisons per hull — 4 100 of the wave's 6 300 VALU instructions,
// DESIGN.md section 12 item 2).  Element e = sub * J + r lives in register r of lane sub: compare-exchanges at distances below J are
// register to register, the others fetch the partner's point from lane sub ^ m (m = 1, 2: DPP quad permutes; m = 4: one
// ds_bpermute round).  A sort is a sort: the sorted sequence of the (distinct) points is unique, equal points are identical, so
// what the chains walk is the same array the rank sort produced — the hulls stay bit-identical to the oracle's.  Entries beyond n
// carry (+inf, +inf) and end up behind the points.
template <int M> __device__ __forceinline__ double grp_xor(double v) {
  if constexpr (M == 4) return __shfl_xor(v, 4);
  else {
    constexpr int ctrl = M == 1 ? 0xB1 : 0x4E;          // quad_perm [1,0,3,2] / [2,3,0,1]
    const int lo = __builtin_amdgcn_mov_dpp(__double2loint(v), ctrl, 0xf, 0xf, false);
    const int hi = __builtin_amdgcn_mov_dpp(__double2hiint(v), ctrl, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
  }
}
template <int J, int K_, int JJ> __device__ __forceinline__ void grp_bitonic_step(double (&kx)[J], double (&ky)[J], int sub) {
  if constexpr (JJ < J) {            // partner in the same lane: registers r and r ^ JJ
#pragma unroll
    for (int r = 0; r < J; r++) {
      const int l = r ^ JJ;
      if (l > r) {
        const bool asc = ((sub * J + r) & K_) == 0;
        const bool l_lt_r = (kx[l] < kx[r]) | ((kx[l] == kx[r]) & (ky[l] < ky[r]));
        const bool r_lt_l = (kx[r] < kx[l]) | ((kx[r] == kx[l]) & (ky[r] < ky[l]));
        const bool sw = asc ? l_lt_r : r_lt_l;
        const double tx = kx[r], ty = ky[r];
        kx[r] = sw ? kx[l] : tx; ky[r] = sw ? ky[l] : ty;
        kx[l] = sw ? tx : kx[l]; ky[l] = sw ? ty : ky[l];
      }
    }
  } else {                           // partner in lane sub ^ (JJ / J): I keep the smaller point when I am the pair's lower index in an ascending run (or the upper one in a descending run)
    constexpr int M = JJ / J;
    const bool asc = ((sub * J) & K_) == 0, lower = (sub & M) == 0;
    const bool want_min = lower == asc;
#pragma unroll
    for (int r = 0; r < J; r++) {
      const double ox = grp_xor<M>(kx[r]), oy = grp_xor<M>(ky[r]);
      const bool o_lt = (ox < kx[r]) | ((ox == kx[r]) & (oy < ky[r]));
      const bool m_lt = (kx[r] < ox) | ((kx[r] == ox) & (ky[r] < oy));
      const bool take = want_min ? o_lt : m_lt;
      kx[r] = take ? ox : kx[r]; ky[r] = take ? oy : ky[r];
    }
  }
}
template <int J, int K_, int JJ> __device__ __forceinline__ void grp_bitonic_merge(double (&kx)[J], double (&ky)[J], int sub) {
  grp_bitonic_step<J, K_, JJ>(kx, ky, sub);
  if constexpr (JJ > 1) grp_bitonic_merge<J, K_, JJ / 2>(kx, ky, sub);
}
template <int J, int K_> __device__ __forceinline__ void grp_bitonic(double (&kx)[J], double (&ky)[J], int sub) {
  if constexpr (K_ > 2) grp_bitonic<J, K_ / 2>(kx, ky, sub);
  grp_bitonic_merge<J, K_, K_ / 2>(kx, ky, sub);
}

template <int J>
__device__ __forceinline__ int group_hull(int n, const double (&px)[J], const double (&py)[J], double* sxy, int g, int sub,
                                          bool write, double* __restrict__ out_xy, int cap) {
  constexpr int kGrpPts = J;          // (points per lane in this instantiation: 8 for the inflated hull's <= 64 points, 2 for the <= 16 control points)
  __syncthreads();
  {
    double kx[J], ky[J];
#pragma unroll
    for (int j = 0; j < J; j++) { const bool in = sub + 8 * j < n; kx[j] = in ? px[j] : NEP_INF; ky[j] = in ? py[j] : NEP_INF; }
    grp_bitonic<J, 8 * J>(kx, ky, sub);
#pragma unroll
    for (int r = 0; r < J; r++) { const int e = sub * J + r; if (e < n) { sxy[2 * e] = kx[r]; sxy[2 * e + 1] = ky[r]; } }
  }
  __syncthreads();
  // unique: sorted position p survives if it differs from its predecessor; new position = number of survivors before it
  double ux[kGrpPts], uy[kGrpPts]; int pos[kGrpPts]; bool keep[kGrpPts];
  int m = 0;
#pragma unroll
  for (int j = 0; j < kGrpPts; j++) {
    const int p = sub + 8 * j;
    ux[j] = 0; uy[j] = 0; keep[j] = false;
    if (p < n) { ux[j] = sxy[2 * p]; uy[j] = sxy[2 * p + 1]; keep[j] = p == 0 || ux[j] != sxy[2 * p - 2] || uy[j] != sxy[2 * p - 1]; }
    const unsigned bits = (unsigned)(__ballot(keep[j]) >> (8 * g)) & 0xffu;
    pos[j] = m + __popc(bits & ((1u << sub) - 1u));
    m += __popc(bits);
  }
  __syncthreads();
#pragma unroll
  for (int j = 0; j < kGrpPts; j++) { if (keep[j]) { sxy[2 * pos[j]] = ux[j]; sxy[2 * pos[j] + 1] = uy[j]; } }
  __syncthreads();
  unsigned long long mask = 0;
  if (sub < 2 && m >= 2) {          // lane 0 of the group: lower chain (indices ascending); lane 1: upper chain (descending)
    const bool lower = sub == 0;
    int kc = 0;
    double ax = 0, ay = 0, bx = 0, by = 0;   // the two top entries
    for (int i = 0; i < m; i++) {
      const int idx = lower ? i : m - 1 - i;
      const double x = sxy[2 * idx], y = sxy[2 * idx + 1];
      // (the mask is kept in PUSH order — bit i = the i-th point this chain visited, i.e. sorted index i for the lower chain and
      // m - 1 - i for the upper one — so that the top of either stack is the highest set bit)
      while (kc >= 2 && cross3(ax, ay, bx, by, x, y) <= 0.0) {
        const int top = 63 - __clzll((long long)mask);
        mask &= ~(1ull << top);
        kc--; bx = ax; by = ay;
        if (kc >= 2) {
          const int t1 = 63 - __clzll((long long)mask);
          const unsigned long long rest = mask & ~(1ull << t1);
          const int t2 = 63 - __clzll((long long)rest);
          const int p2 = lower ? t2 : m - 1 - t2;
          ax = sxy[2 * p2]; ay = sxy[2 * p2 + 1];
        }
      }
      mask |= 1ull << i; kc++;
      ax = bx; ay = by; bx = x; by = y;
    }
  }
  const unsigned long long L = __shfl(mask, 8 * g), U = __shfl(mask, 8 * g + 1);
  const int kl = __popcll(L), ku = __popcll(U);
  const int k = m >= 2 ? kl + ku - 2 : m;
  if (write) {
    if (m == 1) { if (sub == 0 && cap > 0) { out_xy[0] = sxy[0]; out_xy[1] = sxy[1]; } }
    else if (m >= 2) {
#pragma unroll
      for (int j = 0; j < kGrpPts; j++) {
        const int p = sub + 8 * j;
        if (p < m) {
          const double x = sxy[2 * p], y = sxy[2 * p + 1];
          if ((L >> p) & 1ull) { const int o = __popcll(L & ((1ull << p) - 1ull)); if (o < cap) { out_xy[2 * o] = x; out_xy[2 * o + 1] = y; } }
          const int iu = m - 1 - p;                                  // the upper chain's push index of this point
          if (((U >> iu) & 1ull) && p != 0 && p != m - 1) {
            const int o = kl + __popcll(U & ((1ull << iu) - 1ull)) - 1;      // entries pushed before it that are still on the stack, minus the first
            if (o < cap) { out_xy[2 * o] = x; out_xy[2 * o + 1] = y; }
          }
        }
      }
    }
  }
  return k;
}

// One block (= one wave) per (scene, committed trajectory): interval i on lanes 8 i .. 8 i + 7 (num_pol <= 8).
__global__ __launch_bounds__(64) void hull_group_kernel(const nep_traj_rec* __restrict__ recs, int n_rec_per_scene,
                                                        const double* __restrict__ ts0, long ts_scene_stride,
                                                        int num_pol, double T_span, double drone_radius,
                                                        double* __restrict__ hull_xy, int* __restrict__ hull_nv,
                                                        double* __restrict__ hull0_xy, int* __restrict__ hull0_nv,
                                                        double* __restrict__ bend_xy, int* __restrict__ bend_n, int* __restrict__ flags,
                                                        double* __restrict__ box_out, int box_per_scene, int* __restrict__ zero4,
                                                        int n_traj, int ord_n, const int* __restrict__ ord_key, int* __restrict__ ord_out, int* __restrict__ ord_zero4) {
  // LDS of a wave: the groups' sorted points (8 KB) and their control points (2 KB, read again after the hull by the uninflated hull
  // of the entangle rows).  The knot vector lives in the last group's point area — it is dead before the first point is stored — so
  // that the wave takes exactly 10 KB: sixteen waves per CU, and the 8 192 waves of a 128-scene launch are two full rounds (with the
  // 144 bytes of a knot array of its own the CU held fifteen, and the launch was two rounds and a nearly empty third: 0.110 -> 0.099 ms)
  // (measured too: the control points in their group's point area as well — 8 KB a wave, twenty waves per CU, 1.6 rounds: 0.116 ms,
  // slower than both; sixteen waves and two full rounds it is)
  __shared__ __attribute__((aligned(16))) double s_sxy[8][kGrpSxy];
  __shared__ double s_cpx[8][kHullCP], s_cpy[8][kHullCP];
  double* stimes = &s_sxy[7][0];
  static_assert(NEP_TRAJ_MAX_SEG + 2 <= kGrpSxy, "the knot vector borrows a group's point area");
  const int lane = threadIdx.x, g = lane >> 3, sub = lane & 7;
  if (ord_n > 0 && blockIdx.x == 0) {
    // One wave more than there are trajectories — the FIRST, so that it is dispatched at once and ends long before the launch does (as the
    // last block it started in the second round of waves and ended 20 us after the hulls): the QP workgroups' launch order of this round — order_kernel's counting sort of the
    // slots by their key (63 - key: longest expected first; the order within a bin is whatever the atomics make it, as there), on 64
    // threads and in this launch's shadow instead of a launch of its own between the separator and the QP kernels.  It also zeroes
    // the polish pass's counters, as order_kernel does on its way.
    if (ord_zero4 && lane < 4) ord_zero4[lane] = 0;
    int* hist = (int*)&s_sxy[0][0];               // [64 bins][16 sub-histograms], then tot[64]: 4.25 KB of the wave's point area
    int* tot = hist + 64 * 16;
    for (int i = lane; i < 64 * 16; i += 64) hist[i] = 0;
    __syncthreads();
    const int su = lane & 15;
    // (sixteen keys a lane at a time, their reads in flight together: one wave walks 8 192 keys in two passes — read one at a time that is
    // 256 dependent round trips, longer than the whole hull launch)
    for (int i0 = 0; i0 < ord_n; i0 += 1024) {
      int kk[16];
#pragma unroll
      for (int u = 0; u < 16; u++) { const int i = i0 + 64 * u + lane; kk[u] = i < ord_n ? ord_key[i] : 0; }
#pragma unroll
      for (int u = 0; u < 16; u++) { const int i = i0 + 64 * u + lane; if (i < ord_n) atomicAdd(&hist[(63 - (kk[u] & 63)) * 16 + su], 1); }
    }
    __syncthreads();
    { int o = 0; for (int u = 0; u < 16; u++) { const int c = hist[lane * 16 + u]; hist[lane * 16 + u] = o; o += c; } tot[lane] = o; }
    __syncthreads();
    if (lane == 0) { int o = 0; for (int b = 0; b < 64; b++) { const int c = tot[b]; tot[b] = o; o += c; } }
    __syncthreads();
    for (int i0 = 0; i0 < ord_n; i0 += 1024) {
      int kk[16];
#pragma unroll
      for (int u = 0; u < 16; u++) { const int i = i0 + 64 * u + lane; kk[u] = i < ord_n ? ord_key[i] : 0; }
#pragma unroll
      for (int u = 0; u < 16; u++) { const int i = i0 + 64 * u + lane; if (i < ord_n) { const int b = 63 - (kk[u] & 63); ord_out[tot[b] + atomicAdd(&hist[b * 16 + su], 1)] = i; } }
    }
    return;
  }
  const int jt = (int)blockIdx.x - (ord_n > 0 ? 1 : 0);                      // scene * n_rec + j
  const int scene = jt / n_rec_per_scene;
  const nep_traj_rec* r = recs + jt;
  if (lane == 0 && bend_n) {                      // bend points travel with the record (neptune_ros.cpp:457-476)
    int nb = r->n_bend; if (nb > kBend) nb = kBend; if (nb < 0) nb = 0;
    bend_n[jt] = (r->valid && r->is_agent) ? nb : 0;
    for (int b = 0; b < nb; b++) { bend_xy[((long)jt * kBend + b) * 2] = r->bend[b][0]; bend_xy[((long)jt * kBend + b) * 2 + 1] = r->bend[b][1]; }
  }
  const bool act = g < num_pol;                   // this group has an interval
  const long out = (long)jt * num_pol + g;
  if (zero4 && jt == 0 && lane < 4) zero4[lane] = 0;      // (the presolve's redo list starts empty: fe_box_kernel's chore when that kernel runs)
  // the hull's box for the front end's shortlist and the separator's LP skipping (fe_box_kernel's output, made here when box_out is given:
  // one launch less per round).  [scene][box_per_scene][num_pol] x (x0, x1, y0, y1); an empty polygon gets a box nothing meets
  double* box = box_out ? box_out + (((long)scene * box_per_scene + (jt - scene * n_rec_per_scene)) * num_pol + g) * 4 : nullptr;
  if (!(r->valid && r->is_agent) || r->pwp.n_seg <= 0) {   // neptune.cpp:244-262, 332
    if (act && sub == 0) { hull_nv[out] = 0; if (hull0_nv) hull0_nv[out] = 0; if (box) { box[0] = NEP_INF; box[1] = -NEP_INF; box[2] = NEP_INF; box[3] = -NEP_INF; } }
    return;
  }
  // bulk-synchronous round: every agent of a scene replans from the same t_start (that of its first local slot)
  const double ts = *(const double*)((const char*)ts0 + (long)scene * ts_scene_stride);
  const double t0 = ts + g * T_span, t1 = ts + (g + 1) * T_span;   // neptune.cpp:273-280
  const int n = r->pwp.n_seg;
  if (lane <= n) stimes[lane] = r->pwp.times[lane];
  __syncthreads();
  // std::lower_bound / upper_bound on the sorted knot vector (neptune.cpp:379-389): counts of knots < t0 and <= t1
  int first = -1, last = -1;
  for (int k = 0; k <= n; k++) { const double tk = stimes[k]; first += tk < t0 ? 1 : 0; last += tk <= t1 ? 1 : 0; }
  if (first < 0) first = 0; if (first > n - 1) first = n - 1;
  if (last < 0) last = 0; if (last > n - 1) last = n - 1;
  int nseg = last - first + 1; if (nseg < 0) nseg = 0;
  if (nseg > kHullCP / 4) { nseg = kHullCP / 4; if (act && sub == 0 && flags) atomicOr(flags, NEP_FLAG_HULL_OVERFLOW); }   // (see hull_body)
  if (!act) nseg = 0;
  const int np0 = 4 * nseg;
  double* cpx = s_cpx[g]; double* cpy = s_cpy[g];
  for (int v = sub; v < 2 * np0; v += 8) {        // one value per (segment, control point, axis)
    const int sl = v >> 3, k = (v >> 1) & 3, ax = v & 1;
    const int s = first + sl;
    double _t;                                                     // neptune.cpp:399-424
    if (s != last) _t = stimes[s + 1] - stimes[s];
    else if (t1 > stimes[s + 1]) _t = stimes[s + 1] - stimes[s];
    else _t = t1 - stimes[s];
    if (_t > T_span) _t = T_span; else if (_t < 0) _t = 0;
    const double c0 = _t * _t * _t, c1 = _t * _t, c2 = _t, c3 = 1.0;
    const double* P = r->pwp.coeff[ax][s];                          // V = (P*C)*A^-1, :426-429
    const double val = (((P[0] * c0) * cAPosInv[0][k] + (P[1] * c1) * cAPosInv[1][k]) + (P[2] * c2) * cAPosInv[2][k]) + (P[3] * c3) * cAPosInv[3][k];
    if (ax == 0) cpx[sl * 4 + k] = val; else cpy[sl * 4 + k] = val;
  }
  __syncthreads();
  const double dx = r->bbox[0] / 2.0 + drone_radius, dy = r->bbox[1] / 2.0 + drone_radius;  // neptune.cpp:340
  const bool inflate = !(sqrt(dx * dx + dy * dy) < 1e-6);
  const int np = inflate ? 4 * np0 : np0;          // inflated points: 4 corners per control point (:442-445)
  double px[kGrpPts], py[kGrpPts];
#pragma unroll
  for (int j = 0; j < kGrpPts; j++) {
    const int p = sub + 8 * j;
    px[j] = 0; py[j] = 0;
    if (p < np) {
      if (inflate) {
        const int c = p >> 2, q = p & 3;
        const double x = cpx[c], y = cpy[c];
        px[j] = (q < 2) ? x + dx : x - dx;
        py[j] = (q == 0 || q == 3) ? y + dy : y - dy;
      } else { px[j] = cpx[p]; py[j] = cpy[p]; }
    }
  }
  if (box) {
    // The box of the hull's vertices is the box of the points it is the hull of (an extreme point is a vertex), and x -> x +- dx is monotone
    // in floating point too: min over the corners (x - dx) = (min x) - dx.  So: extremes of the control points over the group's lanes
    // (two points a lane, three exchange steps), then the corner offsets — the same four doubles fe_box_kernel reads off the vertices
    // (a hull of more than kHullV vertices is cut and flagged either way).
    double x0 = NEP_INF, x1 = -NEP_INF, y0 = NEP_INF, y1 = -NEP_INF;
#pragma unroll
    for (int j = 0; j < 2; j++) { const int p = sub + 8 * j; if (p < np0) { const double x = cpx[p], y = cpy[p]; x0 = fmin(x0, x); x1 = fmax(x1, x); y0 = fmin(y0, y); y1 = fmax(y1, y); } }
    // (towards the group's lane 0, which writes: lanes 0-3 take lanes 4-7 by a row shift, then two quad permutes — DPP moves only)
    auto shl4 = [](double v) { return __hiloint2double(__builtin_amdgcn_mov_dpp(__double2hiint(v), 0x104, 0xf, 0xf, false), __builtin_amdgcn_mov_dpp(__double2loint(v), 0x104, 0xf, 0xf, false)); };      // row_shl:4
    x0 = fmin(x0, shl4(x0)); x1 = fmax(x1, shl4(x1)); y0 = fmin(y0, shl4(y0)); y1 = fmax(y1, shl4(y1));
    x0 = fmin(x0, grp_xor<2>(x0)); x1 = fmax(x1, grp_xor<2>(x1)); y0 = fmin(y0, grp_xor<2>(y0)); y1 = fmax(y1, grp_xor<2>(y1));
    x0 = fmin(x0, grp_xor<1>(x0)); x1 = fmax(x1, grp_xor<1>(x1)); y0 = fmin(y0, grp_xor<1>(y0)); y1 = fmax(y1, grp_xor<1>(y1));
    if (act && sub == 0) {
      if (np0 <= 0) { box[0] = NEP_INF; box[1] = -NEP_INF; box[2] = NEP_INF; box[3] = -NEP_INF; }
      else if (inflate) { box[0] = x0 - dx; box[1] = x1 + dx; box[2] = y0 - dy; box[3] = y1 + dy; }
      else { box[0] = x0; box[1] = x1; box[2] = y0; box[3] = y1; }
    }
  }
  int k = group_hull<kGrpPts>(np, px, py, s_sxy[g], g, sub, act, hull_xy + out * kHullV * 2, kHullV);
  if (k > kHullV) { k = kHullV; if (act && sub == 0 && flags) atomicOr(flags, NEP_FLAG_HULL_OVERFLOW); }
  if (act && sub == 0) hull_nv[out] = k;
  if (hull0_nv) {
    double qx[2], qy[2];              // (<= 16 control points: two per lane)
#pragma unroll
    for (int j = 0; j < 2; j++) { const int p = sub + 8 * j; qx[j] = p < np0 ? cpx[p] : 0; qy[j] = p < np0 ? cpy[p] : 0; }
    int k0 = group_hull<2>(np0, qx, qy, s_sxy[g], g, sub, act, hull0_xy + out * 2, 1);      // only col(0) is read (:722-734)
    if (k0 > kHullV) k0 = kHullV;
    if (act && sub == 0) hull0_nv[out] = k0;
  }
}

bool hulls_grouped(const SceneParams& sp, int n_scenes, int n_rec) {
  return sp.num_pol <= 8 && (sp.hull_mode ? sp.hull_mode == 2 : (long)n_scenes * n_rec > 2048);
}
void launch_hulls(const nep_traj_rec* recs, int n_scenes, int n_rec, const nep_guess* guess,
                  const SceneParams& sp, const ProblemSet& ps, hipStream_t st, bool boxes) {
  launch_hulls_ts(recs, n_scenes, n_rec, &guess->t_start, (long)sizeof(nep_guess), sp, ps, st, boxes);
}
void launch_hulls_ts(const nep_traj_rec* recs, int n_scenes, int n_rec, const double* ts0, long ts_slot_stride,
                     const SceneParams& sp, const ProblemSet& ps, hipStream_t st, bool boxes) {
  int blocks = n_scenes * n_rec * sp.num_pol;
  if (blocks <= 0) return;
  // the uninflated hull is read only by the entangle rows (col(0), solver_gurobi_poly.cpp:722-734)
  const bool need0 = sp.ent_enabled != 0;
  // Eight hulls per wave when there are enough trajectories to fill the chip with such waves (a wave of eight takes ~60 us,
  // one hull per wave ~26 us: below ~2 000 trajectories the launch is one round of waves either way and the short waves
  // finish first — 0.026 against 0.058 ms for one 64-agent scene; 0.077 both at 32 scenes; 0.271 against 0.183 ms at 128).
  // nep_batch_set_hull_kernel forces one (tests, A/B).
  const bool grouped = sp.hull_mode ? sp.hull_mode == 2 : (long)n_scenes * n_rec > 2048;
  if (sp.num_pol <= 8 && grouped) {
    // (ps.order != null on the way in: the caller wants this round's QP launch order made in the launch's shadow — ps.order_key / ps.order_n)
    const bool ord = boxes && ps.order != nullptr && ps.order_key != nullptr;
    hipLaunchKernelGGL(hull_group_kernel, dim3(n_scenes * n_rec + (ord ? 1 : 0)), dim3(64), 0, st, recs, n_rec, ts0, ts_slot_stride * sp.n_local,
                       sp.num_pol, sp.T_span, sp.drone_radius, ps.hull_xy, ps.hull_nv, need0 ? ps.hull0_xy : nullptr,
                       need0 ? ps.hull0_nv : nullptr, need0 ? ps.bend_xy : nullptr, need0 ? ps.bend_n : nullptr, ps.flags,
                       boxes ? ps.fe_box : nullptr, sp.num_agents + sp.n_static, boxes ? ps.redo_count : nullptr,
                       n_scenes * n_rec, ord ? n_scenes * sp.n_local : 0, ord ? (const int*)ps.order_key : nullptr, ord ? (int*)ps.order : nullptr, ord ? ps.polish_count : nullptr);
    return;
  }
  hipLaunchKernelGGL(hull_kernel, dim3(blocks), dim3(64), 0, st, recs, n_rec, ts0, ts_slot_stride * sp.n_local,
                     sp.num_pol, sp.T_span, sp.drone_radius, ps.hull_xy, ps.hull_nv, need0 ? ps.hull0_xy : nullptr,
                     need0 ? ps.hull0_nv : nullptr, need0 ? ps.bend_xy : nullptr, need0 ? ps.bend_n : nullptr, ps.flags);
}

// ---------------------------------------------------------------------------------------------
// separator
// ---------------------------------------------------------------------------------------------
// Candidate bookkeeping: gaps are compared as num^2/len2 by cross-multiplication, so that only the
// winning candidate needs a square root and divisions (fp64 sqrt/div expand to ~30 VALU ops each).
// Running minima / maxima of the projections are fmin / fmax (one v_min_f64 / v_max_f64 each) rather than compare-and-select
// (three instructions): the inputs are finite, and where the two forms could differ — the sign of a zero — the value is
// only ever compared with zero or subtracted from it before a strict num > 0 test, so the lines are the same bits.
struct SepBest { bool have; double num, len2, tA, nx, ny, px, py; };   // (tA, nx, ny carry the candidate's sign: exact negations)

__device__ __forceinline__ void sep_consider(SepBest& b, double num, double len2, double tA, double nx, double ny, double px, double py) {
  // (before the first candidate b holds the floor itself, num = SEP_MIN_GAP over len2 = 1: num^2 * 1.0 > (MIN * MIN) * len2 is the
  // acceptance test of a first candidate, bit for bit, without a second comparison under a divergent branch; num > 0 is part of
  // the predicate rather than an early exit)
  const bool better = (num > 0.0) & ((num * num) * b.len2 > (b.num * b.num) * len2);
  if (better) { b.have = true; b.num = num; b.len2 = len2; b.tA = tA; b.nx = nx; b.ny = ny; b.px = px; b.py = py; }
}

// Point set A: interleaved (x,y) pairs (one ds_read_b128 per point); point set B: the segment's
// four control points, held in registers.
struct Pts4 { double x[4], y[4]; };

// One candidate pair (p,q) of set X (from_A: X = A).
// Edge p->q of a counter-clockwise convex polygon A (hulls, inflated statics): A lies on the left of
// its own edges, so the A rows need no projection — the edge is tight by construction and only the
// four B points decide the candidate.
__device__ __forceinline__ void sep_edge_ccw(double px, double py, double qx, double qy, const Pts4& B, SepBest& best) {
  const double ex = qx - px, ey = qy - py;
  const double nx = -ey, ny = ex;
  const double len2 = nx * nx + ny * ny;
  // (a degenerate pair, len2 = 0, needs no exit of its own: every projection is then exactly 0, so is the gap, and a candidate must have num > 0)
  double maxB = -NEP_INF;
#pragma unroll
  for (int i = 0; i < 4; i++) { const double t = nx * (B.x[i] - px) + ny * (B.y[i] - py); maxB = fmax(maxB, t); }
  sep_consider(best, 0.0 - maxB, len2, 0.0, nx, ny, px, py);
}

__device__ __forceinline__ void sep_pair(double px, double py, double qx, double qy, bool from_A,
                                         int nA, const double2* __restrict__ A, const Pts4& B, SepBest& best) {
  const double ex = qx - px, ey = qy - py;
  const double nx = -ey, ny = ex;
  const double len2 = nx * nx + ny * ny;
  double minA = NEP_INF, maxA = -NEP_INF, minB = NEP_INF, maxB = -NEP_INF;
#pragma unroll
  for (int i = 0; i < 4; i++) { const double t = nx * (B.x[i] - px) + ny * (B.y[i] - py); minB = fmin(minB, t); maxB = fmax(maxB, t); }
  // a pair of B with B on both sides of its line supports no candidate (np_ = nm = -inf below): skip the pass over A.
  // B is the same in every lane of the separator kernel, so this exit is wave-uniform there.
  if (!from_A && !(maxB <= 0.0) && !(minB >= 0.0)) return;
  for (int i = 0; i < nA; i++) { const double2 a = A[i]; const double t = nx * (a.x - px) + ny * (a.y - py); minA = fmin(minA, t); maxA = fmax(maxA, t); }
  double np_ = -NEP_INF, nm = -NEP_INF, tAp = 0.0, tAm = 0.0;
  if (from_A) {
    if (minA >= 0.0) { np_ = 0.0 - maxB; tAp = 0.0; }
    if (maxA <= 0.0) { nm = minB - 0.0; tAm = 0.0; }
  } else {
    if (maxB <= 0.0) { np_ = minA - 0.0; tAp = minA; }
    if (minB >= 0.0) { nm = 0.0 - maxA; tAm = maxA; }
  }
  const bool plus = np_ >= nm;        // (the side: sign +1 / -1, folded into tA and the normal)
  sep_consider(best, plus ? np_ : nm, len2, plus ? tAp : -tAm, plus ? nx : -nx, plus ? ny : -ny, px, py);
}

// nB == 4 in the path (the segment's control points); the stand-alone entry passes general B in A-like storage.
// kind of point set A: 0 = any points (every pair is a candidate line), 1 = counter-clockwise convex polygon (its edges), 2 = a base
// square in cand_eval's corner order (+,+) (+,-) (-,+) (-,-)
__device__ bool separator_impl(int nA, const double2* __restrict__ A, int kind, const Pts4& B, double nd[3]) {
  SepBest best; best.have = false; best.num = SEP_MIN_GAP; best.len2 = 1; best.tA = 0; best.nx = best.ny = best.px = best.py = 0;
  if (kind == 1 && nA >= 3) {
    for (int p = 0; p < nA - 1; p++) {
      const double2 a0 = A[p], a1 = A[p + 1];
      sep_edge_ccw(a0.x, a0.y, a1.x, a1.y, B, best);
      if (p == 0) { const double2 al = A[nA - 1]; sep_edge_ccw(al.x, al.y, a0.x, a0.y, B, best); }   // closing edge, same orientation
    }
  } else if (kind == 2) {
    // The all-pairs enumeration of a square, without its passes over A: of the six pairs the two diagonals have corners on both
    // sides (no candidate), and for the four edges the side of the other two corners is known — the corners on the edge project
    // to exactly 0 (their difference to the edge's first point is 0 in one coordinate and multiplies an exactly zero normal
    // component in the other), the far ones to -+4 r^2.  Same pairs in the same order with the same first points as the general
    // path: the same candidates, bit for bit, at a third of the instructions (two of five segments meet a base square).
    auto sq_edge = [&](int ip, int iq, bool plus) {
      const double2 p = A[ip], q = A[iq];
      const double ex = q.x - p.x, ey = q.y - p.y;
      const double nx = -ey, ny = ex;
      const double len2 = nx * nx + ny * ny;
      double minB = NEP_INF, maxB = -NEP_INF;
#pragma unroll
      for (int i = 0; i < 4; i++) { const double t = nx * (B.x[i] - p.x) + ny * (B.y[i] - p.y); minB = fmin(minB, t); maxB = fmax(maxB, t); }
      sep_consider(best, plus ? 0.0 - maxB : minB - 0.0, len2, plus ? 0.0 : -0.0, plus ? nx : -nx, plus ? ny : -ny, p.x, p.y);
    };
    sq_edge(0, 1, false); sq_edge(0, 2, true); sq_edge(1, 3, false); sq_edge(2, 3, true);
  } else {
    for (int p = 0; p < nA; p++) for (int q = p + 1; q < nA; q++) { const double2 a0 = A[p], a1 = A[q]; sep_pair(a0.x, a0.y, a1.x, a1.y, true, nA, A, B, best); }
  }
#pragma unroll
  for (int p = 0; p < 4; p++)
#pragma unroll
    for (int q = p + 1; q < 4; q++) sep_pair(B.x[p], B.y[p], B.x[q], B.y[q], false, nA, A, B, best);
  if (!best.have && nA > 0) {
    double cax = 0, cay = 0, cbx = 0, cby = 0;
    for (int i = 0; i < nA; i++) { cax += A[i].x; cay += A[i].y; }
    for (int i = 0; i < 4; i++) { cbx += B.x[i]; cby += B.y[i]; }
    cax /= nA; cay /= nA; cbx /= 4; cby /= 4;
    const double nx = cax - cbx, ny = cay - cby;
    const double len2 = nx * nx + ny * ny;
    if (len2 > 0.0) {
      double minA = NEP_INF, maxB = -NEP_INF;
      for (int i = 0; i < nA; i++) { const double t = nx * (A[i].x - cbx) + ny * (A[i].y - cby); minA = fmin(minA, t); }
      for (int i = 0; i < 4; i++) { const double t = nx * (B.x[i] - cbx) + ny * (B.y[i] - cby); maxB = fmax(maxB, t); }
      sep_consider(best, minA - maxB, len2, minA, nx, ny, cbx, cby);
    }
  }
  if (best.have) {   // the winning LP vertex in the reference's epsilon = 1 scaling
    const double len = sqrt(best.len2);
    const double g = best.num / len;
    const double s = 2.0 / g;
    const double n1 = s * (best.nx / len), n2 = s * (best.ny / len);
    nd[0] = n1; nd[1] = n2;
    nd[2] = (1.0 - s * (best.tA / len)) - (n1 * best.px + n2 * best.py);
    return true;
  }
  nd[0] = nd[1] = nd[2] = 0.0;
  return false;
}

// waves per SIMD the separator's register allocation is bounded for (3: 168 VGPRs, no spills: 0.615 ms per 4.2 M LPs; 4: 128
// VGPRs, 12 spilled: 0.555 ms; 5: 96 VGPRs, 42 spilled: 0.68 ms — same-box A/B)
#ifndef NEP_SEP_WAVES
#define NEP_SEP_WAVES 4
#endif
// Point sets A of a batch of 64 LPs are staged in one LDS pool, each lane's polygon at the exclusive prefix sum of the vertex
// counts (6-12 vertices for an interval hull, 4 for a base or a static): ~590 pairs on average instead of 64 x 13 reserved
// ones, so that a wave needs 10 KB and SIXTEEN waves share a CU (the allocation is bounded to 128 VGPRs for the same four
// waves per SIMD: 11 % faster than three).  A polygon that does not fit what is left of the pool is read where it lies
// (global memory / L2); the computed point sets (base squares, entangle segments) fall back to a private array.  The packed
// offsets are not bank-conflict-free as the 13-pair stride was; the LDS pipe has the slack (the kernel is bound by VALU issue).
#ifndef NEP_SEP_LDS
#define NEP_SEP_LDS (10 * 1024)
#endif
constexpr int kSepLdsTarget = NEP_SEP_LDS;
// Candidate c of segment seg, in the reference's loop order (solver_gurobi_poly.cpp:477-495 agents,
// :521-553 bases, :556-593 statics, :620-637 entangle): does the reference call the separator for
// it, and (stage == true) what is point set A.

// ---- separator rule 1: a primal simplex of the class GLPK's glp_simplex runs by default (nep_batch_set_separator_rule) ----
// The reference's separator hands its LP to glp_simplex with glp_init_smcp's defaults (separator_glpk.cpp:39-41, 336): primal
// simplex, projected steepest-edge pricing, Harris' two-pass ratio test, tol_bnd = tol_dj = 1e-7, no presolve, no scaling,
// from the standard basis (every row variable basic, the free structurals n1, n2, d non-basic at zero).  This is that
// documented algorithm class — stated line by line in oracle/neptune_oracle.c::orc_separator_glpk_class, which this function
// follows operation for operation (same loop orders, same association, no contraction): the two return the same bits.
// A basis leaves three variables non-basic; x = W v with W = G^-1 (G: gradients of the non-basics, by cofactors), the tableau
// is [A 1; B 1] W.  Rows 0 .. nA-1 are A's (r >= 1), rows nA .. nA+3 the four points of B (r <= -1).
__device__ __forceinline__ bool spx_inv3(const double (&G)[3][3], double (&W)[3][3]) {
  const double c00 = G[1][1] * G[2][2] - G[1][2] * G[2][1], c01 = G[1][2] * G[2][0] - G[1][0] * G[2][2], c02 = G[1][0] * G[2][1] - G[1][1] * G[2][0];
  const double det = (G[0][0] * c00 + G[0][1] * c01) + G[0][2] * c02;
  if (!(fabs(det) > 1e-300)) return false;
  const double id = 1.0 / det;
  W[0][0] = c00 * id; W[1][0] = c01 * id; W[2][0] = c02 * id;
  W[0][1] = (G[0][2] * G[2][1] - G[0][1] * G[2][2]) * id; W[1][1] = (G[0][0] * G[2][2] - G[0][2] * G[2][0]) * id; W[2][1] = (G[0][1] * G[2][0] - G[0][0] * G[2][1]) * id;
  W[0][2] = (G[0][1] * G[1][2] - G[0][2] * G[1][1]) * id; W[1][2] = (G[0][2] * G[1][0] - G[0][0] * G[1][2]) * id; W[2][2] = (G[0][0] * G[1][1] - G[0][1] * G[1][0]) * id;
  return true;
}
__device__ bool separator_glpk_class(int nA, const double2* A, const Pts4& B, double nd[3]) {
  constexpr int kMaxIt = 60;
  constexpr double kTolBnd = 1e-7, kTolDj = 1e-7, kTolPiv = 1e-9;
  nd[0] = nd[1] = nd[2] = 0.0;
  if (nA <= 0) return false;
  auto point = [&](int i, double& px, double& py) {     // row i's point (B's four live in registers: selected, not indexed)
    if (i < nA) { const double2 a = A[i]; px = a.x; py = a.y; }
    else { const int k = i - nA; px = k == 0 ? B.x[0] : (k == 1 ? B.x[1] : (k == 2 ? B.x[2] : B.x[3])); py = k == 0 ? B.y[0] : (k == 1 ? B.y[1] : (k == 2 ? B.y[2] : B.y[3])); }
  };
  int slot[3] = {-1, -2, -3};
  unsigned nb_rows = 0;
  for (int it = 0; it <= kMaxIt; it++) {
    double G[3][3], W[3][3], v[3], x[3];
#pragma unroll
    for (int j = 0; j < 3; j++) {
      if (slot[j] < 0) { const int k = -slot[j] - 1; G[j][0] = k == 0; G[j][1] = k == 1; G[j][2] = k == 2; v[j] = 0.0; }
      else { double px, py; point(slot[j], px, py); G[j][0] = px; G[j][1] = py; G[j][2] = 1.0; v[j] = slot[j] < nA ? 1.0 : -1.0; }
    }
    if (!spx_inv3(G, W)) return false;
#pragma unroll
    for (int k = 0; k < 3; k++) x[k] = (W[k][0] * v[0] + W[k][1] * v[1]) + W[k][2] * v[2];
    double d[3] = {0.0, 0.0, 0.0}; int n_inf = 0;
    auto cost_row = [&](int i, double px, double py, bool isA) {
      if ((nb_rows >> i) & 1u) return;
      const double r = (px * x[0] + py * x[1]) + x[2];
      const double bnd = isA ? 1.0 : -1.0, delta = kTolBnd * (1.0 + 1e-3 * fabs(bnd));
      double c = 0.0;
      if (isA) { if (r < bnd - delta) c = -1.0; } else { if (r > bnd + delta) c = 1.0; }
      if (c != 0.0) {
        n_inf++;
#pragma unroll
        for (int j = 0; j < 3; j++) d[j] += c * ((px * W[0][j] + py * W[1][j]) + W[2][j]);
      }
    };
    for (int i = 0; i < nA; i++) { const double2 a = A[i]; cost_row(i, a.x, a.y, true); }
#pragma unroll
    for (int k = 0; k < 4; k++) cost_row(nA + k, B.x[k], B.y[k], false);
    if (n_inf == 0) { nd[0] = x[0]; nd[1] = x[1]; nd[2] = x[2]; return true; }
    if (it == kMaxIt) break;
    int q = -1; double best = 0.0, sdir = 0.0;
#pragma unroll
    for (int j = 0; j < 3; j++) {
      double s_ = 0.0; bool elig = false;
      if (slot[j] < 0) { if (d[j] < -kTolDj) { s_ = 1.0; elig = true; } else if (d[j] > kTolDj) { s_ = -1.0; elig = true; } }
      else if (slot[j] < nA) { if (d[j] < -kTolDj) { s_ = 1.0; elig = true; } }
      else { if (d[j] > kTolDj) { s_ = -1.0; elig = true; } }
      if (elig) {
        double gamma = slot[j] < 0 ? 1.0 : 0.0;
#pragma unroll
        for (int k = 0; k < 3; k++) {
          const bool basic = slot[0] != -(k + 1) && slot[1] != -(k + 1) && slot[2] != -(k + 1);
          if (basic) gamma += W[k][j] * W[k][j];
        }
        if (!(gamma > 1e-300)) gamma = 1e-300;
        const double score = d[j] * d[j] / gamma;
        if (score > best) { best = score; q = j; sdir = s_; }
      }
    }
    if (q < 0) return false;
    const double wq0 = q == 0 ? W[0][0] : (q == 1 ? W[0][1] : W[0][2]), wq1 = q == 0 ? W[1][0] : (q == 1 ? W[1][1] : W[1][2]), wq2 = q == 0 ? W[2][0] : (q == 1 ? W[2][1] : W[2][2]);
    double tmax = NEP_INF;
    int p = -1;
    for (int pass = 0; pass < 2; pass++) {
      double piv = 0.0;
      auto ratio_row = [&](int i, double px, double py, bool isA) {
        if ((nb_rows >> i) & 1u) return;
        const double r = (px * x[0] + py * x[1]) + x[2];
        const double rho = sdir * ((px * wq0 + py * wq1) + wq2);
        if (!(fabs(rho) > kTolPiv)) return;
        const double bnd = isA ? 1.0 : -1.0, delta = kTolBnd * (1.0 + 1e-3 * fabs(bnd));
        double dist;
        if (isA) { const bool inf = r < bnd - delta; if (inf ? rho > 0 : rho < 0) dist = inf ? bnd - r : r - bnd; else return; }
        else { const bool inf = r > bnd + delta; if (inf ? rho < 0 : rho > 0) dist = inf ? r - bnd : bnd - r; else return; }
        const double arho = fabs(rho);
        if (pass == 0) { const double t = (dist + delta) / arho; if (t < tmax) tmax = t; }
        else { const double t = dist / arho; if (t <= tmax && arho > piv) { piv = arho; p = i; } }
      };
      for (int i = 0; i < nA; i++) { const double2 a = A[i]; ratio_row(i, a.x, a.y, true); }
#pragma unroll
      for (int k = 0; k < 4; k++) ratio_row(nA + k, B.x[k], B.y[k], false);
    }
    if (p < 0) return false;
    nb_rows |= 1u << p;
    const int old = q == 0 ? slot[0] : (q == 1 ? slot[1] : slot[2]);
    if (old >= 0) nb_rows &= ~(1u << old);
    if (q == 0) slot[0] = p; else if (q == 1) slot[1] = p; else slot[2] = p;
  }
  return false;
}

struct SepCtx {
  const SceneParams* sp; const ProblemSet* ps;
  int slot, scene, own, N, S, nH, total;
  double el[3];          // lengths of the control polygon's three edges (the terms of hulldist)
  double bb[4];          // box of the segment's four control points (x0, x1, y0, y1): the spatial presolve's far test
  const double* skip_box; // boxes of the hulls / statics (fe_box_kernel) when far LPs may be skipped, else null
  double skip_r;
};
// Spatial presolve (cx.skip_box != null): is polygon j's box farther than skip_r from the box of the segment's control points along
// x or along y?  Hulls and inflated statics are hulls of axis-aligned squares, so the sides of their boxes are EDGES of the
// polygon: the largest-gap line of such a pair has a gap of at least the box distance, i.e. it lies farther than skip_r from
// every control point of the guess — the LP need not be solved to know that its line is "far" (separator_body).
__device__ __forceinline__ bool box_far(const SepCtx& cx, int seg, int jbox) {
  const double* b = cx.skip_box + (((long)cx.scene * (cx.N + cx.S) + jbox) * cx.sp->num_pol + seg) * 4;
  const double x0 = b[0], x1 = b[1], y0 = b[2], y1 = b[3];
  return (x0 - cx.bb[1] > cx.skip_r) | (cx.bb[0] - x1 > cx.skip_r) | (y0 - cx.bb[3] > cx.skip_r) | (cx.bb[2] - y1 > cx.skip_r);
}
__device__ bool cand_eval(const SepCtx& cx, int seg, int c, const double* bx, const double* by, double hulldist,
                          int mode, double2* myA, int& nA, int& ordered, const double2*& Ause, bool* skip = nullptr) {
  const bool stage = mode == 1, cull_tests = mode == 0;   // mode 0: the reference's proximity culls; 1: stage the point set (myA, or in place when null); 2: vertex count only
  const SceneParams& sp = *cx.sp; const ProblemSet& ps = *cx.ps;
  const int N = cx.N, S = cx.S, nH = cx.nH;
  nA = 0; ordered = 0;       // (the kind of point set: see separator_impl)
  if (c < nH) {
    const int j = c;
    if (sp.skip_own && j == cx.own) return false;
    const HullRef hr = hull_ref(ps, nH, cx.scene, j);
    const long h = hr.e * sp.num_pol + seg;
    nA = blk(ps.hull_nv, hr.boff)[h];
    if (nA <= 0) return false;
    ordered = 1;
    if (skip && cx.skip_box) *skip = box_far(cx, seg, j);
    if (stage) {
      const double2* src = (const double2*)(blk(ps.hull_xy, hr.boff) + h * kHullV * 2);
      if (myA) for (int v = 0; v < nA; v++) myA[v] = src[v]; else Ause = src;
    }
    return true;
  } else if (c < nH + N) {
    const int j = c - nH;
    const double base_radius = 0.7;
    const double pbx = ps.pb[2 * j], pby = ps.pb[2 * j + 1];
    bool close_to_base = !cull_tests;      // (staging is only asked for candidates that passed this test in step 1)
    // (a base farther than 2.2 m from the box of the four control points along x or y is farther than that from each of them: the
    // per-point test below is false for all four — a round of 64 far bases skips it as a whole)
    if (cull_tests && !((pbx < cx.bb[0] - 2.2) | (pbx > cx.bb[1] + 2.2) | (pby < cx.bb[2] - 2.2) | (pby > cx.bb[3] + 2.2))) {
      // all four control points, no exits (lanes of a wave leave a loop at different points only to wait for each other); the square
      // root is taken where the coarse test does not already say "far" (sqrt(ddx^2 + ddy^2) >= max(|ddx|, |ddy|): beyond 2.2 on
      // either axis the reference's test is false)
      bool near_any = false; double d2[4];
#pragma unroll
      for (int k = 0; k < 4; k++) {
        const double ddx = bx[k] - pbx, ddy = by[k] - pby;
        d2[k] = ddx * ddx + ddy * ddy;
        const bool nr = !((fabs(ddx) > 2.2) | (fabs(ddy) > 2.2));
        if (!nr) d2[k] = 1e30;
        near_any |= nr;
      }
      if (near_any) {
#pragma unroll
        for (int k = 0; k < 4; k++) close_to_base |= sqrt(d2[k]) < base_radius * 3;
      }
    }
    if (!close_to_base) return false;
    nA = 4; ordered = 2;
    if (stage) {  // :536-540 (column order of base_hull)
      myA[0] = make_double2(pbx + base_radius, pby + base_radius);
      myA[1] = make_double2(pbx + base_radius, pby - base_radius);
      myA[2] = make_double2(pbx - base_radius, pby + base_radius);
      myA[3] = make_double2(pbx - base_radius, pby - base_radius);
    }
    return true;
  } else if (c < nH + N + S) {
    const long j = (long)cx.scene * sp.static_stride + (c - nH - N);
    const int nv = ps.static_nv[j];
    if (nv <= 0) return false;
    const double* src = ps.static_xy + j * kHullV * 2;
    if (cull_tests) {          // :558-578 (staging is only asked for candidates that passed this test in step 1)
      bool close_s = false;
      const double ddx = bx[0] - src[0], ddy = by[0] - src[1];
      double dist = sqrt(ddx * ddx + ddy * ddy);
      // (same values as the reference's per-call square roots; the reference leaves at the first negative remainder — the later
      // subtractions cannot un-set the verdict, so they are done anyway instead of a lane-divergent exit)
#pragma unroll
      for (int k = 0; k < 3; k++) { dist -= cx.el[k]; close_s |= dist < 0; }
      for (int k = 0; k < nv - 1; k++) { dist -= ps.static_el[j * kHullV + k]; close_s |= dist < 0; }
      if (!close_s) return false;
    }
    ordered = 1; nA = nv;
    if (skip && cx.skip_box) *skip = box_far(cx, seg, N + (c - nH - N));
    if (stage) { if (myA) for (int v = 0; v < nv; v++) myA[v] = make_double2(src[2 * v], src[2 * v + 1]); else Ause = (const double2*)src; }
    return true;
  } else if (c < cx.total) {
    const int e = c - nH - N - S;
    const int j = e / kBend, k = e % kBend + 1;
    if (j == cx.own) return false;
    // (modes 1 and 2 are asked for LISTED candidates only — they passed the tests below in step 1 —: the vertex count is 2 without a
    // read, and staging reads the two points alone: one round trip, two for k = 1, instead of the five of the chain of tests)
    if (mode == 2) { nA = 2; return true; }
    if (mode == 1) {
      const HullRef hr1 = hull_ref(ps, N, cx.scene, j);
      const double2* bp2 = (const double2*)(blk(ps.bend_xy, hr1.boff) + hr1.e * kBend * 2);
      if (k == 1) {  // :719-724
        const int nb1 = blk(ps.bend_n, hr1.boff)[hr1.e];
        const double2 h0 = ((const double2*)blk(ps.hull0_xy, hr1.boff))[hr1.e * sp.num_pol + seg];
        const double2 bl = bp2[nb1 - 1];
        myA[0] = make_double2((1 - sp.long_length) * bl.x + sp.long_length * h0.x, (1 - sp.long_length) * bl.y + sp.long_length * h0.y);
        myA[1] = h0;
      } else { myA[0] = bp2[k - 2]; myA[1] = bp2[k - 1]; }  // :725-730
      nA = 2;
      return true;
    }
    const int case_id = ps.case_id[((long)cx.slot * NEP_MAX_POL + seg) * N + j];
    const HullRef hr = hull_ref(ps, N, cx.scene, j);
    const int nb = blk(ps.bend_n, hr.boff)[hr.e];
    if (!(case_id != 0 && k != case_id && k <= nb)) return false;  // :631-636
    const double* bp = blk(ps.bend_xy, hr.boff) + hr.e * kBend * 2;
    const long h0 = hr.e * sp.num_pol + seg;
    double pAx, pAy, pBx, pBy;
    if (k == 1) {  // :719-724
      if (blk(ps.hull0_nv, hr.boff)[h0] <= 0) return false;
      const double hx0 = blk(ps.hull0_xy, hr.boff)[h0 * 2], hy0 = blk(ps.hull0_xy, hr.boff)[h0 * 2 + 1];
      pAx = (1 - sp.long_length) * bp[2 * (nb - 1)] + sp.long_length * hx0;
      pAy = (1 - sp.long_length) * bp[2 * (nb - 1) + 1] + sp.long_length * hy0;
      pBx = hx0; pBy = hy0;
    } else {  // :725-730
      pAx = bp[2 * (k - 2)]; pAy = bp[2 * (k - 2) + 1]; pBx = bp[2 * (k - 1)]; pBy = bp[2 * (k - 1) + 1];
    }
    const double ax_ = pAx - bx[0], ay_ = pAy - by[0], bx_ = pBx - bx[0], by_ = pBy - by[0];
    if (cull_tests && sqrt(ax_ * ax_ + ay_ * ay_) - hulldist > 0 && sqrt(bx_ * bx_ + by_ * by_) - hulldist > 0) return false;  // :743-745
    nA = 2;
    if (stage) { myA[0] = make_double2(pAx, pAy); myA[1] = make_double2(pBx, pBy); }
    return true;
  }
  return false;
}

// One wave per (agent slot, segment).  Step 1 runs the reference's proximity culls for every
// candidate of the segment and builds, with ordered wave ballots, the list of LPs the reference
// would actually call — in its loop order; step 2 hands those LPs out 64 at a time (one lane each,
// point set A staged in the lane's LDS slot), so that lanes are not parked on rejected candidates
// (with 63 other agents that is one full round plus a short tail instead of three rounds).  Line l
// lands in bucket (slot, seg) at its call rank; an LP without a separating line leaves (0,0,0)
// there — the QP kernel reads that as "constraint skipped" (solver_gurobi_poly.cpp:491-494).
// presolve: the handle's line presolve is on for this launch (sp.cull_radius > 0): lines far from the guess are parked, and LPs
// whose line is known to be far without solving them (box_far) are not solved at all — they are counted as attempted and
// solved (two point sets more than skip_r apart are separable), their number goes to ps.line_skip, and the QP kernel verifies
// them through the distance its solution moved from the guess (qp_reg_kernel).  presolve = false: every LP, every line in call
// order — the reference's loop, and what the redo pass runs for a replan whose presolve did not verify.
template <int RULE>
__device__ __forceinline__ void separator_body(const SceneParams& sp, const ProblemSet& ps, int pool_pairs, int slot, int seg, bool presolve) {
  extern __shared__ __attribute__((aligned(16))) double sdyn[];
  double2* sA = (double2*)sdyn;                      // [pool_pairs] (x,y) pairs: the batch's point sets A, packed
  double* sBx = sdyn + 2 * pool_pairs; double* sBy = sBx + 4;
  unsigned short* sAtt = (unsigned short*)(sBy + 4);
  const int lane = threadIdx.x;
  const nep_guess* g = ps.guess + slot;
  const int K = g->K;
  int* cnt_out = ps.line_cnt + (long)slot * NEP_MAX_POL + seg;
  int* lp_out = ps.lp_stats + ((long)slot * NEP_MAX_POL + seg) * 2;      // every wave writes its own pair: no memset, no atomics
  if (seg >= K || seg >= sp.num_pol) { if (lane == 0) { *cnt_out = 0; lp_out[0] = 0; lp_out[1] = 0; if (ps.line_far) ps.line_far[(long)slot * NEP_MAX_POL + seg] = 0; if (ps.line_skip) ps.line_skip[(long)slot * NEP_MAX_POL + seg] = 0; } return; }
  const double T = sp.T_span;
  SepCtx cx; cx.sp = &sp; cx.ps = &ps; cx.slot = slot; cx.scene = slot / sp.n_local; cx.own = sp.first_local + (slot % sp.n_local);
  cx.N = sp.num_agents; cx.S = sp.n_static; cx.nH = sp.n_hull;
  cx.total = cx.nH + cx.N + cx.S + ((sp.ent_enabled && ps.case_id) ? cx.N * kBend : 0);
  const int total = cx.total;
  if (lane < 4) {  // ctrlPtsInit_[seg] (solver_gurobi_poly.cpp:232-243)
    const double tp0 = T * T * T, tp1 = T * T, tp2 = T;
    const double m0 = tp0 * cAPosInv[0][lane], m1 = tp1 * cAPosInv[1][lane], m2 = tp2 * cAPosInv[2][lane], m3 = 1.0 * cAPosInv[3][lane];
    const double* Px = g->coeff[0][seg]; const double* Py = g->coeff[1][seg];
    sBx[lane] = ((Px[0] * m0 + Px[1] * m1) + Px[2] * m2) + Px[3] * m3;
    sBy[lane] = ((Py[0] * m0 + Py[1] * m1) + Py[2] * m2) + Py[3] * m3;
  }
  __syncthreads();
  const double* bx = sBx; const double* by = sBy;
  double hulldist = 0;  // :738-742
  for (int k = 0; k < 3; k++) { const double ex = bx[k + 1] - bx[k], ey = by[k + 1] - by[k]; cx.el[k] = sqrt(ex * ex + ey * ey); hulldist += cx.el[k]; }
  cx.bb[0] = fmin(fmin(bx[0], bx[1]), fmin(bx[2], bx[3])); cx.bb[1] = fmax(fmax(bx[0], bx[1]), fmax(bx[2], bx[3]));
  cx.bb[2] = fmin(fmin(by[0], by[1]), fmin(by[2], by[3])); cx.bb[3] = fmax(fmax(by[0], by[1]), fmax(by[2], by[3]));
  // (the largest-gap rule is what guarantees "box far => line far"; a simplex-reached vertex may lie anywhere between the sets)
  cx.skip_box = (presolve && RULE == 0 && sp.cull_radius > 0.0 && ps.line_far != nullptr) ? ps.skip_box : nullptr; cx.skip_r = sp.cull_radius;
  // ---- step 1: which LPs does the reference call, in order -------------------------------------
  int n_att = 0, n_skip = 0;                              // LPs to solve (listed), LPs known to give a far line (counted only)
  const int n_plain = cx.nH + cx.N + cx.S;                // hulls, bases, statics: one candidate per lane and round
  int c_first = 0;
  if (cx.skip_box) {
    // Spatial presolve: a hull candidate is decided by its box alone — empty (fe_box_kernel marks it with x0 = +inf: never called),
    // far (counted, not solved) or to be solved — one 32-byte load per candidate, fetched a round ahead of its use, instead
    // of the vertex count followed by the box; at config 5 these are 4 of a segment's rounds and nearly all of them are far.
    const double* bx0 = cx.skip_box + ((long)cx.scene * (cx.N + cx.S) * sp.num_pol + seg) * 4;
    auto load_box = [&](int j, double2& a, double2& b) {
      const bool v = j < cx.nH;
      const double2* q = (const double2*)(bx0 + (long)(v ? j : 0) * sp.num_pol * 4);
      a = q[0]; b = q[1];
    };
    double2 ca, cb, na, nb2;
    load_box(lane, ca, cb);
    for (int c0 = 0; c0 < cx.nH; c0 += 64) {
      const int j = c0 + lane;
      load_box(j + 64, na, nb2);
      const bool valid = j < cx.nH && !(sp.skip_own && j == cx.own) && ca.x < NEP_INF;
      const bool far = (ca.x - cx.bb[1] > cx.skip_r) | (cx.bb[0] - ca.y > cx.skip_r) | (cb.x - cx.bb[3] > cx.skip_r) | (cx.bb[2] - cb.y > cx.skip_r);
      const unsigned long long mask = __ballot(valid && !far);
      if (valid && !far) sAtt[n_att + __popcll(mask & ((1ull << lane) - 1ull))] = (unsigned short)j;
      n_att += __popcll(mask);
      n_skip += __popcll(__ballot(valid && far));
      ca = na; cb = nb2;
    }
    c_first = cx.nH;
  }
  for (int c0 = c_first; c0 < n_plain; c0 += 64) {
    const int c = c0 + lane;
    int nA; int ord; bool skp = false;
    const double2* unused = nullptr;
    const bool att = c < n_plain && cand_eval(cx, seg, c, bx, by, hulldist, 0, nullptr, nA, ord, unused, &skp);
    const unsigned long long mask = __ballot(att && !skp);
    if (att && !skp) sAtt[n_att + __popcll(mask & ((1ull << lane) - 1ull))] = (unsigned short)c;
    n_att += __popcll(mask);
    n_skip += __popcll(__ballot(att && skp));
  }
  // Entangle candidates (agent j, bend segment k), the reference's double loop (:624-636).  Nine agents in ten have no active
  // case: the agents that do are first gathered (one load per agent, ballot order = id order), then their (j, k) pairs are
  // handed out densely, one per lane — a few rounds instead of one per 64 (j, k) slots (32 of a config-5 segment's 42 rounds,
  // and still 0.6 of the kernel's 1.0 ms when every lane with a case walked its own eight k).  Pair order = (j, k) order.
  if (n_plain < total) {
    unsigned short* sAct = sAtt + (total + 8);
    int n_act = 0;
    for (int j0 = 0; j0 < cx.N; j0 += 64) {
      const int j = j0 + lane;
      const bool act = j < cx.N && j != cx.own && ps.case_id[((long)cx.slot * NEP_MAX_POL + seg) * cx.N + j] != 0;
      const unsigned long long mask = __ballot(act);
      if (act) sAct[n_act + __popcll(mask & ((1ull << lane) - 1ull))] = (unsigned short)j;
      n_act += __popcll(mask);
    }
    __syncthreads();
    for (int p0 = 0; p0 < n_act * kBend; p0 += 64) {
      const int pp = p0 + lane;
      int nA; int ord; const double2* unused = nullptr;
      const int c = pp < n_act * kBend ? n_plain + (int)sAct[pp / kBend] * kBend + (pp % kBend) : 0;
      const bool att = pp < n_act * kBend && cand_eval(cx, seg, c, bx, by, hulldist, 0, nullptr, nA, ord, unused);
      const unsigned long long mask = __ballot(att);
      if (att) sAtt[n_att + __popcll(mask & ((1ull << lane) - 1ull))] = (unsigned short)c;
      n_att += __popcll(mask);
    }
  }
  __syncthreads();
  // ---- step 2: the LPs ---------------------------------------------------------------------------
  Pts4 B4;
#pragma unroll
  for (int k = 0; k < 4; k++) { B4.x[k] = sBx[k]; B4.y[k] = sBy[k]; }
  double* bucket = ps.line_nd + ((long)slot * NEP_MAX_POL + seg) * sp.lines_cap * 3;
  int n_fail = 0;
  // Presolve (sp.cull_radius > 0): a line whose boundary is farther than cull_radius from all four control points of
  // the guess is parked at the back of the bucket ("far") and left out of the QP; the QP kernel verifies them against
  // its solution and re-solves with every line if one is violated, so the optimum is unchanged.  Near lines keep
  // their order at the front; far lines are written from the end of the bucket, in order of appearance.
  const bool cull = presolve && sp.cull_radius > 0.0 && ps.line_far != nullptr;
  int n_near = 0, n_far = 0;                              // wave-uniform running counts
  for (int a0 = 0; a0 < n_att; a0 += 64) {
    const int a = a0 + lane;
    const bool active = a < n_att;
    double nd[3] = {0.0, 0.0, 0.0};
    bool far = false;
    // this batch's pool: every lane's vertex count, its exclusive prefix sum across the wave, then the staging
    int c = 0, nA = 0; int ord = 0;
    if (active) { c = sAtt[a]; const double2* u_ = nullptr; cand_eval(cx, seg, c, bx, by, hulldist, 2, nullptr, nA, ord, u_); }
    int incl = nA;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const int v = __shfl_up(incl, o); if (lane >= o) incl += v; }
    if (active) {
      double2 priv[4];                                       // (base squares / entangle segments when the pool is full: rare)
      double2* myA = (incl <= pool_pairs) ? sA + (incl - nA) : nullptr;
      const bool made_here = (c >= cx.nH && c < cx.nH + cx.N) || c >= cx.nH + cx.N + cx.S;
      if (!myA && made_here) myA = priv;
      const double2* Ause = myA;
      cand_eval(cx, seg, c, bx, by, hulldist, 1, myA, nA, ord, Ause);
      const bool ok = RULE == 1 ? separator_glpk_class(nA, Ause, B4, nd) : separator_impl(nA, Ause, ord, B4, nd);   // (the LP vertex rule: sp.sep_rule)
      if (!ok) { n_fail++; nd[0] = nd[1] = nd[2] = 0.0; }
      if (cull) {
        double worst = -NEP_INF;                          // largest n.Q + d - 1 over the guess's control points (<= -2 for a solved LP)
#pragma unroll
        for (int k = 0; k < 4; k++) { const double v = (nd[0] * B4.x[k] + nd[1] * B4.y[k]) + (nd[2] - 1.0); if (v > worst) worst = v; }
        const double len = sqrt(nd[0] * nd[0] + nd[1] * nd[1]);
        far = !ok || -worst > sp.cull_radius * len;       // (a null row constrains nothing)
      }
    }
    if (!cull) {
      if (active && a < sp.lines_cap) { bucket[3 * a] = nd[0]; bucket[3 * a + 1] = nd[1]; bucket[3 * a + 2] = nd[2]; }
    } else {
      const unsigned long long mn = __ballot(active && !far), mf = __ballot(active && far);
      const unsigned long long below = (1ull << lane) - 1ull;
      if (active) {
        const long pos = far ? (long)sp.lines_cap - 1 - (n_far + __popcll(mf & below)) : (long)n_near + __popcll(mn & below);
        if (pos >= 0 && pos < sp.lines_cap) { bucket[3 * pos] = nd[0]; bucket[3 * pos + 1] = nd[1]; bucket[3 * pos + 2] = nd[2]; }      // (a bucket smaller than the worst case: an overflow is flagged below, never written past the bucket)
      }
      n_near += __popcll(mn); n_far += __popcll(mf);
    }
  }
  for (int o = 32; o > 0; o >>= 1) n_fail += __shfl_xor(n_fail, o);
  if (lane == 0) {
    const bool spill = cull ? n_near + n_far > sp.lines_cap : n_att > sp.lines_cap;
    if (spill) { if (ps.flags) atomicOr(ps.flags, NEP_FLAG_LINES); if (n_near > sp.lines_cap) n_near = sp.lines_cap; if (n_far > sp.lines_cap - n_near) n_far = sp.lines_cap - n_near; }
    // (a bucket that could not hold every line of its segment: the count goes out as -1 - n, and the QP kernels fail that replan — it
    // keeps its previous trajectory — instead of solving without the rows that did not fit; line_count() reads either form)
    const int n_out = cull ? n_near : (n_att < sp.lines_cap ? n_att : sp.lines_cap);
    *cnt_out = spill ? -1 - n_out : n_out;
    if (ps.line_far) ps.line_far[(long)slot * NEP_MAX_POL + seg] = cull ? n_far : 0;
    if (ps.line_skip) ps.line_skip[(long)slot * NEP_MAX_POL + seg] = n_skip;
    lp_out[0] = n_att + n_skip; lp_out[1] = n_fail;
  }
}
template <int RULE>
__global__ __launch_bounds__(64, NEP_SEP_WAVES) void separator_kernel(SceneParams sp, ProblemSet ps, int pool_pairs) {
  separator_body<RULE>(sp, ps, pool_pairs, blockIdx.x / NEP_MAX_POL, blockIdx.x % NEP_MAX_POL, true);
}
// The spatial presolve's separator (launched instead of separator_kernel<0> when LPs may be skipped, ps.skip_box != null): one wave
// takes kSepPack consecutive segments of a slot (all eight in a large launch).  With the skipping a segment is left with a dozen or
// two LPs (17 on average at config 5) — a quarter of a wave's lanes, and an LP batch costs the same whatever its fill — so the LPs of
// the wave's segments go to ONE list of (segment, candidate) entries, segment-major, and are solved 64 to a batch across the
// segments: three batches per slot instead of eight.  Same candidates, same culls, same per-LP arithmetic and the same order within
// every segment as separator_body (lines and counts are bit-identical: the presolve tests compare them).
// Round 6 — the kernel is bound by instruction issue at four waves per SIMD (every 1 000 instructions of a wave
// are 17 us of an 8 192-wave launch) once its loads stop being serial round trips:
//   step 1a  which candidates are LPs, CANDIDATE-major: what a lane reads of its candidate (a hull's eight boxes, a base, a static
//            polygon's first vertex / edges / box, an agent's eight case ids) is loaded once, every load of a round in flight, then
//            tested against each segment's control points (LDS); the verdicts are ballots per (segment, round of 64 candidates),
//            kept by lanes (one item per lane) when they fit, in LDS otherwise;
//   step 1b  the list: a DPP scan over the items' counts gives every item its place, and every lane writes its own item's entries
//            one set bit after the other (with entangle candidates: per segment, followed by the segment's (agent, bend segment)
//            pairs, four rounds of pairs at a time with their reads issued first); when the list cannot hold what is gathered, what
//            is there is solved first (one call site of the LP code);
//   step 2   64 LPs a batch: point sets staged at a DPP prefix sum of the vertex counts, separator_impl, line placement by the
//            runs of a segment's lanes.
// (segments per wave, chosen by the host from the launch's size — 2: 0.555, 3: 0.511, 4: 0.497, 8: 0.488 ms per 8 192 config-5
// replans against 0.639 unpacked, measured before the round-6 work; small launches keep more, shorter waves)
// inclusive prefix sum of an int over the wave's 64 lanes in the data-parallel-primitive moves of gfx9: four shifts within the rows of 16,
// then lane 15 of a row to the next row and lane 31 to the upper half (eight moves; the ds_bpermute form is six LDS-crossbar round trips)
__device__ __forceinline__ int wave_incl_scan(int v) {
  v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, false);      // row_shr:1
  v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, false);      // row_shr:2
  v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, false);      // row_shr:4
  v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, false);      // row_shr:8
  v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, false);      // row_bcast:15 -> rows 1 and 3
  v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, false);      // row_bcast:31 -> rows 2 and 3
  return v;
}
// entries of the packed kernel's LP list: what one segment's hulls, bases and statics can list at most (they are laid out a segment at
// a time; a round of 64 entangle pairs fits as well), and room for what all eight segments of a replan usually gather together (about
// 130 LPs at 64 agents + 20 obstacles), so that the list is placed in one go.  (Until round 6 the list had the unpacked kernel's
// capacity, every candidate of a segment with the entangle pairs: 5.3 KB of a config-5 wave's LDS for a list of some 140 entries.)
__host__ __device__ inline int sep_packed_cap(int n_plain) { return n_plain + 72 < 320 ? 320 : n_plain + 72; }
#ifdef NEP_SEP_PROF
__device__ unsigned long long g_sep_prof[16384 * 16];
#define SEP_PT(k) do { const long long t_ = clock64(); pa_[k] += t_ - pt_; pt_ = t_; } while (0)
#else
#define SEP_PT(k) do { } while (0)
#endif
__global__ __launch_bounds__(64, NEP_SEP_WAVES) void separator_packed_kernel(SceneParams sp, ProblemSet ps, int pool_pairs, int kSepPack) {
#ifdef NEP_SEP_PROF
  long long pt_ = clock64(); long long pa_[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#endif
  const int kSepGroups = (NEP_MAX_POL + kSepPack - 1) / kSepPack;
  extern __shared__ __attribute__((aligned(16))) double sdyn[];
  const int rh = (sp.n_hull + 63) >> 6, rb = (sp.num_agents + 63) >> 6, rs = (sp.n_static + 63) >> 6, rounds = rh + rb + rs;      // candidate rounds of a segment: hulls, bases, statics
  double2* sA = (double2*)sdyn;                      // [pool_pairs] the batch's point sets A, packed
  double* sBx = sdyn + 2 * pool_pairs; double* sBy = sBx + 4 * NEP_MAX_POL;      // [segment][4] control points
  double* sEl = sBy + 4 * NEP_MAX_POL;                // [segment][3] lengths of the control polygon's edges
  double* sBb = sEl + 3 * NEP_MAX_POL;                // [segment][4] box of the control points (x0, x1, y0, y1)
  unsigned long long* sMask = (unsigned long long*)(sBb + 4 * NEP_MAX_POL);      // [segment][round][2]: lanes whose candidate is an LP to solve / known to give a far line
  const bool ent_any = sp.ent_enabled && ps.case_id;
  unsigned long long* sEMask = sMask + 2 * NEP_MAX_POL * rounds;      // [segment][rb]: agents with an active entangle case (only with the entangle rows)
  int* sCnt = (int*)(sEMask + (ent_any ? NEP_MAX_POL * rb : 0));          // [segment][6]: near, far, failed, attempted, skipped (running)
  unsigned short* sAtt = (unsigned short*)(sCnt + 6 * NEP_MAX_POL);      // entries (segment << 13 | candidate)
  const int lane = threadIdx.x;
  const int slot = blockIdx.x / kSepGroups, grp = blockIdx.x % kSepGroups;
  const int seg_lo = grp * kSepPack, seg_hi = seg_lo + kSepPack < NEP_MAX_POL ? seg_lo + kSepPack : NEP_MAX_POL;
  const nep_guess* g = ps.guess + slot;
  const int K = g->K;
  const double T = sp.T_span;
  SepCtx cx; cx.sp = &sp; cx.ps = &ps; cx.slot = slot; cx.scene = slot / sp.n_local; cx.own = sp.first_local + (slot % sp.n_local);
  cx.N = sp.num_agents; cx.S = sp.n_static; cx.nH = sp.n_hull;
  cx.total = cx.nH + cx.N + cx.S + ((sp.ent_enabled && ps.case_id) ? cx.N * kBend : 0);
  const bool cull = sp.cull_radius > 0.0 && ps.line_far != nullptr;      // (the line presolve: far lines parked at the back of the bucket)
  cx.skip_box = cull ? ps.skip_box : nullptr; cx.skip_r = sp.cull_radius;      // (the spatial presolve on top: LPs known to give a far line are not solved)
  const int total = cx.total, cap = sep_packed_cap(cx.nH + cx.N + cx.S);
  int seg_end = seg_hi < K ? seg_hi : K; if (seg_end > sp.num_pol) seg_end = sp.num_pol;      // segments [seg_lo, seg_end) exist
  if (lane < 6 * NEP_MAX_POL) sCnt[lane] = 0;
  if (lane < 4 * (seg_hi - seg_lo)) {  // ctrlPtsInit_[seg] (solver_gurobi_poly.cpp:232-243)
    const int seg = seg_lo + (lane >> 2), k = lane & 3;
    // (the coefficients are read whether or not the segment exists — the arrays do —, so that these loads do not wait for K's)
    const double4 Px = *(const double4*)g->coeff[0][seg], Py = *(const double4*)g->coeff[1][seg];
    const double tp0 = T * T * T, tp1 = T * T, tp2 = T;
    const double m0 = tp0 * cAPosInv[0][k], m1 = tp1 * cAPosInv[1][k], m2 = tp2 * cAPosInv[2][k], m3 = 1.0 * cAPosInv[3][k];
    const double vx_ = ((Px.x * m0 + Px.y * m1) + Px.z * m2) + Px.w * m3, vy_ = ((Py.x * m0 + Py.y * m1) + Py.z * m2) + Py.w * m3;
    if (seg < seg_end) { sBx[seg * 4 + k] = vx_; sBy[seg * 4 + k] = vy_; }
  }
  __syncthreads();
  if (lane < 4 * (seg_hi - seg_lo) && seg_lo + (lane >> 2) < seg_end) {      // (one lane per value instead of every lane of the wave repeating a segment's three square roots and its box)
    const int seg = seg_lo + (lane >> 2), k = lane & 3;
    const double* bx_ = sBx + seg * 4; const double* by_ = sBy + seg * 4;
    if (k < 3) { const double ex = bx_[k + 1] - bx_[k], ey = by_[k + 1] - by_[k]; sEl[seg * 3 + k] = sqrt(ex * ex + ey * ey); }
    const double* q_ = k < 2 ? bx_ : by_;
    sBb[seg * 4 + k] = (k & 1) ? fmax(fmax(q_[0], q_[1]), fmax(q_[2], q_[3])) : fmin(fmin(q_[0], q_[1]), fmin(q_[2], q_[3]));
  }
  __syncthreads();
  SEP_PT(0);
  // ---- step 1a: which candidates of which segment are LPs to solve / LPs known to give a far line — the reference's proximity culls
  // (cand_eval, mode 0) and the spatial presolve's box test, as ballots per (segment, round of 64 candidates).  Candidate-major: what
  // a lane reads of its candidate (base position, static polygon's first vertex, edge lengths and box; an interval hull's eight boxes,
  // which lie side by side) is loaded ONCE, every load of a round issued before the first is used, and then tested against each
  // segment's control points (LDS).  The segment-major form of this step — one dependent global round trip after another, eight
  // segments in turn, four waves per SIMD to hide them — was two thirds of the kernel's time (0.097 of 0.144 ms per 8 192 replans).
  const int nsv = seg_end - seg_lo;
  // The ballots of a (segment, round) item: kept by lane `item` in two registers when the wave's items fit its lanes and there are no
  // entangle candidates (then the list is laid out from the registers: a scan over the items' counts, every lane writing its own
  // item's entries); in LDS otherwise ([segment][round][2]).
  const int n_items = nsv > 0 ? nsv * rounds : 0;
  const bool reg_items = cx.total == cx.nH + cx.N + cx.S && n_items <= 64;
  unsigned long long my_a = 0ull, my_k = 0ull;
  auto put_mask = [&](int sg, int round, unsigned long long ma, unsigned long long mk) {
    if (reg_items) { if (lane == (sg - seg_lo) * rounds + round) { my_a = ma; my_k = mk; } }
    else if (lane == 0) { sMask[(sg * rounds + round) * 2] = ma; sMask[(sg * rounds + round) * 2 + 1] = mk; }
  };
  if (nsv > 0) {
    const double rr = cx.skip_r;
    for (int c0 = 0; c0 < cx.nH; c0 += 64) {      // interval hulls: decided by their boxes alone (empty: x0 = +inf, never called)
      const int j = c0 + lane; const bool in = j < cx.nH;
      const bool valid0 = in && !(sp.skip_own && j == cx.own);
      if (cx.skip_box) {
        const double2* q = (const double2*)(cx.skip_box + ((long)cx.scene * (cx.N + cx.S) + (in ? j : 0)) * sp.num_pol * 4);
        for (int s0 = seg_lo; s0 < seg_end; s0 += 8) {
          double2 qa[8], qb[8];
#pragma unroll
          for (int u = 0; u < 8; u++) { const int sg = s0 + u < seg_end ? s0 + u : seg_end - 1; qa[u] = q[sg * 2]; qb[u] = q[sg * 2 + 1]; }
#pragma unroll
          for (int u = 0; u < 8; u++) {
            const int sg = s0 + u;
            if (sg < seg_end) {
              const double* bb = sBb + sg * 4;
              const bool valid = valid0 && qa[u].x < NEP_INF;
              const bool far = (qa[u].x - bb[1] > rr) | (bb[0] - qa[u].y > rr) | (qb[u].x - bb[3] > rr) | (bb[2] - qb[u].y > rr);
              const unsigned long long ma = __ballot(valid && !far), mk = __ballot(valid && far);
              put_mask(sg, c0 >> 6, ma, mk);
            }
          }
        }
      } else {
        for (int sg = seg_lo; sg < seg_end; sg++) {
          int nA; int ord; const double2* unused = nullptr;
          const bool att = in && cand_eval(cx, sg, j, sBx + sg * 4, sBy + sg * 4, 0.0, 0, nullptr, nA, ord, unused);
          const unsigned long long ma = __ballot(att);
          put_mask(sg, c0 >> 6, ma, 0ull);
        }
      }
    }
    SEP_PT(1);
    for (int j0 = 0; j0 < cx.N; j0 += 64) {      // bases (:521-553): within 3 x 0.7 m of one of the four control points
      const int j = j0 + lane; const bool in = j < cx.N;
      constexpr double base_radius = 0.7;
      const double pbx = ps.pb[2 * (in ? j : 0)], pby = ps.pb[2 * (in ? j : 0) + 1];
      for (int sg = seg_lo; sg < seg_end; sg++) {
        const double* bx_ = sBx + sg * 4; const double* by_ = sBy + sg * 4;
        // cand_eval's test — sqrt(d^2) < 3 x 0.7 for one of the four control points, behind its two coarse "farther than 2.2 m along an axis"
        // exits — without the square roots: for a correctly rounded square root, sqrt(x) < 0.7 * 3 (= 0x1.0ccccccccccccp+1) exactly when
        // x < kBaseNear2, the smallest double whose root rounds to that or more; a point within that distance passes both coarse tests
        // (2.1 < 2.2, roundings of 1e-16 apart), so the verdict is cand_eval's, bit for bit (the GPU tests compare the line lists).
        constexpr double kBaseNear2 = 0x1.1a3d70a3d70a2p+2;
        static_assert(base_radius * 3 == 0x1.0ccccccccccccp+1, "kBaseNear2 belongs to 0.7 * 3");
        bool close_to_base = false;
#pragma unroll
        for (int k = 0; k < 4; k++) { const double ddx = bx_[k] - pbx, ddy = by_[k] - pby; close_to_base |= ddx * ddx + ddy * ddy < kBaseNear2; }
        close_to_base &= in;
        const unsigned long long ma = __ballot(close_to_base);
        put_mask(sg, rh + (j0 >> 6), ma, 0ull);
      }
    }
    SEP_PT(2);
    // static polygons (:556-593): the perimeter cull, then the box.  With S <= 32 polygons a round of lanes takes 64 / S SEGMENTS of
    // them (lane = segment-in-round x S + polygon: 20 obstacles, three segments a round — three rounds instead of eight a third full)
    const int spr = (cx.S > 0 && cx.S <= 32) ? 64 / cx.S : 1;      // segments per round
    for (int c0 = 0; c0 < cx.S; c0 += 64) {
      const int qs = spr > 1 ? lane / cx.S : 0;                     // this lane's segment within the round
      const int js = spr > 1 ? lane - qs * cx.S : c0 + lane; const bool in = js < cx.S && qs < spr;
      const long j = (long)cx.scene * sp.static_stride + (in ? js : 0);
      const int nv = in ? ps.static_nv[j] : 0;
      const double* src = ps.static_xy + j * kHullV * 2;
      const double sx0 = src[0], sy0 = src[1];
      const double* elp = ps.static_el + j * kHullV;
      const double e0 = elp[0], e1 = elp[1], e2 = elp[2];      // (a polygon's first edges — all of a square's — ahead of the loop that consumes them)
      double perim = 0;
      if (nv > 1) perim += e0; if (nv > 2) perim += e1; if (nv > 3) perim += e2;
      for (int k = 3; k < nv - 1; k++) perim += elp[k];
      double2 qa = make_double2(0, 0), qb = qa;      // the polygon's box: the same in every interval
      if (cx.skip_box) { const double2* q = (const double2*)(cx.skip_box + (((long)cx.scene * (cx.N + cx.S) + cx.N + (in ? js : 0)) * sp.num_pol + seg_lo) * 4); qa = q[0]; qb = q[1]; }
      for (int sg0 = seg_lo; sg0 < seg_end; sg0 += spr) {
        const int sg = sg0 + qs;                         // (per lane when spr > 1)
        const bool on = nv > 0 && sg < seg_end;
        const int sgc = sg < seg_end ? sg : seg_end - 1;
        const double* bb = sBb + sgc * 4;
        const double el0 = sEl[sgc * 3], el1 = sEl[sgc * 3 + 1], el2 = sEl[sgc * 3 + 2];
        bool close_s = false;
        const double ddx = sBx[sgc * 4] - sx0, ddy = sBy[sgc * 4] - sy0;
        const double d2s = ddx * ddx + ddy * ddy;
        // (the cull below can only fire when the distance is less than what it subtracts — the control polygon's length plus the
        // polygon's edges; a round of polygons all farther than that, with a margin a million times the roundings, skips the
        // square roots and the chain as a whole: same verdicts)
        const double thr = ((el0 + el1) + el2 + perim) * 1.000001 + 1e-6;
        if (__ballot(on && d2s <= thr * thr) != 0ull)
        if (on) {          // :558-578
          double dist = sqrt(d2s);
          dist -= el0; close_s |= dist < 0; dist -= el1; close_s |= dist < 0; dist -= el2; close_s |= dist < 0;
          if (nv > 1) { dist -= e0; close_s |= dist < 0; }
          if (nv > 2) { dist -= e1; close_s |= dist < 0; }
          if (nv > 3) { dist -= e2; close_s |= dist < 0; }
          for (int k = 3; k < nv - 1; k++) { dist -= elp[k]; close_s |= dist < 0; }
        }
        const bool far = cx.skip_box != nullptr && ((qa.x - bb[1] > rr) | (bb[0] - qa.y > rr) | (qb.x - bb[3] > rr) | (bb[2] - qb.y > rr));
        const unsigned long long ma = __ballot(close_s && !far), mk = __ballot(close_s && far);
        if (spr > 1) {
          const unsigned long long sm = (1ull << cx.S) - 1ull;
          for (int u = 0; u < spr && sg0 + u < seg_end; u++) put_mask(sg0 + u, rh + rb, (ma >> (u * cx.S)) & sm, (mk >> (u * cx.S)) & sm);
        } else put_mask(sg0, rh + rb + (c0 >> 6), ma, mk);
      }
      if (spr > 1) break;
    }
    if (ent_any) {      // agents with an active entangle case, per segment (:624-631): an agent's case ids of the wave's segments in one go
      for (int j0 = 0; j0 < cx.N; j0 += 256) {      // (four rounds of agents x eight segments: 32 reads in flight)
        for (int s0 = seg_lo; s0 < seg_end; s0 += 8) {
          int cv[4][8];
#pragma unroll
          for (int w = 0; w < 4; w++) {
            const int j = j0 + 64 * w + lane;
            const int* cp = ps.case_id + ((long)cx.slot * NEP_MAX_POL + seg_lo) * cx.N + (j < cx.N ? j : 0);
#pragma unroll
            for (int u = 0; u < 8; u++) { const int sg = s0 + u < seg_end ? s0 + u : seg_end - 1; cv[w][u] = cp[(long)(sg - seg_lo) * cx.N]; }
          }
#pragma unroll
          for (int w = 0; w < 4; w++) {
            const int j = j0 + 64 * w + lane; const bool in = j < cx.N && j != cx.own;
            if (j0 + 64 * w < cx.N) {
#pragma unroll
              for (int u = 0; u < 8; u++) {
                if (s0 + u < seg_end) { const unsigned long long m = __ballot(in && cv[w][u] != 0); if (lane == 0) sEMask[(s0 + u) * rb + ((j0 >> 6) + w)] = m; }
              }
            }
          }
        }
      }
    }
    SEP_PT(3);
  }
  __syncthreads();
  int n_list = 0;
  // ---- the LPs gathered so far, 64 to a batch across the segments ----
  auto flush = [&]() {
    __syncthreads();
    SEP_PT(4);
    for (int a0 = 0; a0 < n_list; a0 += 64) {
      const int a = a0 + lane;
      const bool active = a < n_list;
      const int e = active ? (int)sAtt[a] : 0;
      const int sl = e >> 13, c = e & 8191;
      Pts4 B4;
#pragma unroll
      for (int k = 0; k < 4; k++) { B4.x[k] = sBx[sl * 4 + k]; B4.y[k] = sBy[sl * 4 + k]; }
      double nd[3] = {0.0, 0.0, 0.0};
      bool far = false, ok = true;
      int nA = 0; int ord = 0;
      if (active) { const double2* u_ = nullptr; cand_eval(cx, sl, c, sBx, sBy, 0.0, 2, nullptr, nA, ord, u_); }
      const int incl = wave_incl_scan(nA);
      if (active) {
        double2 priv[4];
        double2* myA = (incl <= pool_pairs) ? sA + (incl - nA) : nullptr;
        const bool made_here = (c >= cx.nH && c < cx.nH + cx.N) || c >= cx.nH + cx.N + cx.S;
        if (!myA && made_here) myA = priv;
        const double2* Ause = myA;
        cand_eval(cx, sl, c, sBx, sBy, 0.0, 1, myA, nA, ord, Ause);
#ifdef NEP_SEP_PROF
        SEP_PT(5);
#endif
        ok = separator_impl(nA, Ause, ord, B4, nd);
#ifdef NEP_SEP_PROF
        SEP_PT(6);
#endif
        if (!ok) { nd[0] = nd[1] = nd[2] = 0.0; }
        if (cull) {
          double worst = -NEP_INF;
#pragma unroll
          for (int k = 0; k < 4; k++) { const double v = (nd[0] * B4.x[k] + nd[1] * B4.y[k]) + (nd[2] - 1.0); if (v > worst) worst = v; }
          const double len = sqrt(nd[0] * nd[0] + nd[1] * nd[1]);
          far = !ok || -worst > sp.cull_radius * len;
        }
      }
      // Near lines keep their call order at the front of their segment's bucket, far ones from its end.  The list is segment-major, so
      // the lanes of a segment are a run of consecutive lanes: a lane's place is its rank among the near (far) lanes of its run on top
      // of the segment's running counts, and the run's last lane brings the counts forward — no loop over the segments.
      {
        const unsigned long long below = (1ull << lane) - 1ull;
        const unsigned long long mact = __ballot(active), mn = __ballot(active && !far), mf = __ballot(active && far), mx = __ballot(active && !ok);
        const int sl_prev = __builtin_amdgcn_update_dpp(-1, sl, 0x138, 0xf, 0xf, false);      // wave_shr:1 (lane 0 keeps -1)
        const unsigned long long ms = __ballot(active && (lane == 0 || sl != sl_prev));        // first lanes of the runs
        const unsigned long long upto = ms & (below | (1ull << lane));
        const int s0_ = upto ? 63 - __clzll((long long)upto) : 0;
        const unsigned long long run_below = below & ~((1ull << s0_) - 1ull), run_incl = run_below | (1ull << lane);
        const bool last = active && (lane == 63 || (((~mact | ms) >> (lane + 1)) & 1ull));
        const int base_n = active ? sCnt[sl * 6] : 0, base_f = active ? sCnt[sl * 6 + 1] : 0;
        if (active) {
          double* bucket = ps.line_nd + ((long)slot * NEP_MAX_POL + sl) * sp.lines_cap * 3;
          const long pos = far ? (long)sp.lines_cap - 1 - (base_f + __popcll(mf & run_below)) : (long)base_n + __popcll(mn & run_below);
          if (pos >= 0 && pos < sp.lines_cap) { bucket[3 * pos] = nd[0]; bucket[3 * pos + 1] = nd[1]; bucket[3 * pos + 2] = nd[2]; }
        }
        __syncthreads();
        if (last) { sCnt[sl * 6] = base_n + __popcll(mn & run_incl); sCnt[sl * 6 + 1] = base_f + __popcll(mf & run_incl); sCnt[sl * 6 + 2] += __popcll(mx & run_incl); }
        __syncthreads();
      }
      SEP_PT(7);
    }
    n_list = 0;
  };
  // ---- step 1b: the list of LPs from the ballots, segment-major, the reference's call order within a segment; ONE call site of flush()
  // (the lambda is inlined: with a call site in each round loop the kernel once carried four copies of the LP code, 70 KB)
  int seg = seg_lo - 1, ph = 4, c0 = 0, n_att = 0, n_skip = 0, n_act = 0;      // ph: 1 the ballots of step 1a, 2 entangle agents, 3 entangle pairs, 4 next segment
  const int n_plain = cx.nH + cx.N + cx.S;
  const double* bx = sBx; const double* by = sBy;
  double hulldist = 0;
  int tag = 0;
  unsigned short* sAct = sAtt + cap;
  bool more = true;
  // Without entangle candidates the list is the ballots read out in (segment, round) order: a flat walk, a few instructions a round
  // (the kernel is bound by instruction issue — every 1 000 instructions of a wave are 17 us of the launch —, and the general walk
  // below, whose state machine also serves the entangle rounds, cost 26 us for this).
  const bool ent_c = n_plain < total;
  int it_sg = seg_lo, it_r = 0;
  if (!ent_c && !reg_items && lane < nsv) {      // attempted / skipped per segment, straight from the ballots
    int na_ = 0, nk_ = 0;
    for (int r = 0; r < rounds; r++) { na_ += __popcll(sMask[((seg_lo + lane) * rounds + r) * 2]); nk_ += __popcll(sMask[((seg_lo + lane) * rounds + r) * 2 + 1]); }
    sCnt[(seg_lo + lane) * 6 + 3] = na_; sCnt[(seg_lo + lane) * 6 + 4] = nk_;
  }
  // (when every LP of the wave's segments fits the list at once — nearly always — an entry's place is known from the ballots' counts: a
  // scan over the (segment, round) items, one per lane, then the entries are written item by item with nothing carried from one item
  // to the next; the serial walk below, an LDS round trip and a scalar chain per round, was 18 % of the wave's cycles)
  bool placed = false;
  if (reg_items) {
    const int cnt_ = __popcll(my_a), cntk_ = __popcll(my_k);
    const int incl_ = wave_incl_scan(cnt_), inclk_ = wave_incl_scan(cntk_);
    const int tot_ = __builtin_amdgcn_readlane(incl_, 63);
    {      // attempted / skipped per segment: differences of the scans at the segments' last items (every lane takes part in the shuffles)
      const int hi_ = lane < nsv ? (lane + 1) * rounds - 1 : 0, lo_ = (lane < nsv && lane > 0) ? lane * rounds - 1 : 0;
      const int a_hi = __shfl(incl_, hi_), k_hi = __shfl(inclk_, hi_), a_lo = __shfl(incl_, lo_), k_lo = __shfl(inclk_, lo_);
      if (lane < nsv) { sCnt[(seg_lo + lane) * 6 + 3] = a_hi - (lane > 0 ? a_lo : 0); sCnt[(seg_lo + lane) * 6 + 4] = k_hi - (lane > 0 ? k_lo : 0); }
    }
    if (tot_ <= cap) {
      const int sgi_ = lane / rounds, r_ = lane - sgi_ * rounds;      // this lane's item
      const int cb_ = r_ < rh ? (r_ << 6) : (r_ < rh + rb ? cx.nH + ((r_ - rh) << 6) : cx.nH + cx.N + ((r_ - rh - rb) << 6));
      const int ebase_ = ((seg_lo + sgi_) << 13) | cb_, off_ = incl_ - cnt_;
      // (every lane writes the entries of ITS item, one set bit after the other: as many rounds as the fullest ballot has bits,
      // instead of one round per item)
      { unsigned long long m_ = my_a; int o_ = off_; while (m_) { const int b_ = __builtin_ctzll(m_); m_ &= m_ - 1ull; sAtt[o_++] = (unsigned short)(ebase_ + b_); } }
      n_list = tot_; placed = true;
    } else {      // (more LPs than the list holds at once: the ballots go to LDS for the serial walk below)
      if (lane < n_items) { sMask[(seg_lo * rounds + lane) * 2] = my_a; sMask[(seg_lo * rounds + lane) * 2 + 1] = my_k; }
      __syncthreads();
    }
  }
  while (more) {
    if (!ent_c) {
      if (placed) it_sg = seg_end;
      for (; it_sg < seg_end; ) {
        if (n_list > 0 && n_list + 64 > cap) break;
        const unsigned long long mv = sMask[(it_sg * rounds + it_r) * 2];
        const unsigned mlo = __builtin_amdgcn_readfirstlane((unsigned)mv), mhi = __builtin_amdgcn_readfirstlane((unsigned)(mv >> 32));
        const int cb_ = it_r < rh ? (it_r << 6) : (it_r < rh + rb ? cx.nH + ((it_r - rh) << 6) : cx.nH + cx.N + ((it_r - rh - rb) << 6));
        const unsigned rank = __builtin_amdgcn_mbcnt_hi(mhi, __builtin_amdgcn_mbcnt_lo(mlo, 0u));
        if ((((lane & 32) ? mhi : mlo) >> (lane & 31)) & 1u) sAtt[n_list + rank] = (unsigned short)((it_sg << 13) | (cb_ + lane));
        n_list += __popc(mlo) + __popc(mhi);
        if (++it_r == rounds) { it_r = 0; it_sg++; }
      }
      if (it_sg >= seg_end) more = false;
    } else
    for (;;) {
      if (ph == 4) {
        if (seg >= seg_lo && seg < seg_end && lane == 0) { sCnt[seg * 6 + 3] = n_att; sCnt[seg * 6 + 4] = n_skip; }
        seg++;
        if (seg >= seg_end) { more = false; break; }
        bx = sBx + seg * 4; by = sBy + seg * 4; tag = seg << 13;
        hulldist = 0;  // :738-742
        for (int k = 0; k < 3; k++) { cx.el[k] = sEl[seg * 3 + k]; hulldist += cx.el[k]; }
        for (int k = 0; k < 4; k++) cx.bb[k] = sBb[seg * 4 + k];
        n_att = 0; n_skip = 0; c0 = 0; ph = 1;
      }
      if (ph == 3 && c0 >= n_act * kBend) { ph = 4; continue; }
      if (ph == 3 && n_list > 0 && n_list + 64 > cap) break;      // (the round might not fit: what has been gathered is solved first; a round alone always fits)
      if (ph == 1) {
        // the segment's plain candidates (hulls, bases, statics) from step 1a's ballots: lane r takes round r's ballot and writes its
        // entries one set bit after the other, at the place a scan of the rounds' counts gives it (the serial form — a round per turn of
        // this loop, ballots through LDS and the scalar unit — was 30 % of the wave's cycles at config 5, 80 rounds a replan)
        int tot_seg = 0, skip_seg = 0;
        for (int r0 = 0; r0 < rounds; r0 += 64) {
          const int r = r0 + lane;
          const unsigned long long m0 = r < rounds ? sMask[(seg * rounds + r) * 2] : 0ull, k0_ = r < rounds ? sMask[(seg * rounds + r) * 2 + 1] : 0ull;
          tot_seg += __builtin_amdgcn_readlane(wave_incl_scan(__popcll(m0)), 63); skip_seg += __builtin_amdgcn_readlane(wave_incl_scan(__popcll(k0_)), 63);
        }
        if (n_list > 0 && n_list + tot_seg > cap) break;      // (a segment's plain candidates always fit an empty list)
        for (int r0 = 0; r0 < rounds; r0 += 64) {
          const int r = r0 + lane;
          unsigned long long m_ = r < rounds ? sMask[(seg * rounds + r) * 2] : 0ull;
          const int cnt_ = __popcll(m_), incl_ = wave_incl_scan(cnt_);
          const int cb_ = r < rh ? (r << 6) : (r < rh + rb ? cx.nH + ((r - rh) << 6) : cx.nH + cx.N + ((r - rh - rb) << 6));
          int o_ = n_list + incl_ - cnt_;
          while (m_) { const int b_ = __builtin_ctzll(m_); m_ &= m_ - 1ull; sAtt[o_++] = (unsigned short)(tag | (cb_ + b_)); }
          n_list += __builtin_amdgcn_readlane(incl_, 63);
        }
        n_att += tot_seg; n_skip += skip_seg;
        ph = n_plain < total ? 2 : 4;
      } else if (ph == 2) {      // entangle candidates (agent j, bend segment k): the agents with an active case first (step 1a's ballots), then their pairs densely
        SEP_PT(4);
        n_act = 0;
        __syncthreads();
        for (int ch = 0; ch < rb; ch++) {
          const unsigned long long mask = sEMask[seg * rb + ch];
          if ((mask >> lane) & 1ull) sAct[n_act + __popcll(mask & ((1ull << lane) - 1ull))] = (unsigned short)((ch << 6) + lane);
          n_act += __popcll(mask);
        }
        __syncthreads();
        ph = 3; c0 = 0;
        SEP_PT(9);
      } else {
        SEP_PT(4);
        // up to four rounds of (agent, bend segment) pairs at a time: what a pair reads — the agent's case id, its bend count, the two bend
        // points (or, for k = 1, col(0) of its uninflated hull) — is requested for all four rounds before the first is looked at, then the
        // one dependent read (k = 1: the last bend point, at index nb - 1); cand_eval's chain of tests in between used to make every
        // one of them a round trip of its own, four to five per round and thirty-odd rounds per replan at config 5
        const int n_pairs = n_act * kBend;
        int nr = (n_pairs - c0 + 63) >> 6; if (nr > 4) nr = 4;
        while (nr > 1 && n_list + 64 * nr > cap) nr--;      // (every round of the group must fit the list; one round always does)
        int cid[4], nbv[4], h0n[4], jj[4]; double2 X[4], Y[4], Z[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
          const int pp = c0 + 64 * u + lane; const bool in = u < nr && pp < n_pairs;
          const int j = in ? (int)sAct[pp / kBend] : 0, k = pp % kBend + 1;
          jj[u] = j;
          const HullRef hr = hull_ref(ps, cx.N, cx.scene, j);
          cid[u] = ps.case_id[((long)cx.slot * NEP_MAX_POL + seg) * cx.N + j];
          nbv[u] = blk(ps.bend_n, hr.boff)[hr.e];
          const double2* bp = (const double2*)(blk(ps.bend_xy, hr.boff) + hr.e * kBend * 2);
          const long h0 = hr.e * sp.num_pol + seg;
          h0n[u] = blk(ps.hull0_nv, hr.boff)[h0];
          X[u] = k == 1 ? ((const double2*)blk(ps.hull0_xy, hr.boff))[h0] : bp[k - 2];
          Y[u] = bp[k - 1];
        }
#pragma unroll
        for (int u = 0; u < 4; u++) {
          const int pp = c0 + 64 * u + lane; const int k = pp % kBend + 1;
          const HullRef hr = hull_ref(ps, cx.N, cx.scene, jj[u]);
          const double2* bp = (const double2*)(blk(ps.bend_xy, hr.boff) + hr.e * kBend * 2);
          int nbc = nbv[u]; if (nbc < 1) nbc = 1; if (nbc > kBend) nbc = kBend;
          Z[u] = bp[k == 1 ? nbc - 1 : 0];
        }
#pragma unroll
        for (int u = 0; u < 4; u++) {
          if (u < nr) {
            const int pp = c0 + 64 * u + lane; const int k = pp % kBend + 1;
            bool att = pp < n_pairs && cid[u] != 0 && k != cid[u] && k <= nbv[u];      // :631-636
            double pAx, pAy, pBx, pBy;
            if (k == 1) {  // :719-724
              att = att && h0n[u] > 0;
              pAx = (1 - sp.long_length) * Z[u].x + sp.long_length * X[u].x; pAy = (1 - sp.long_length) * Z[u].y + sp.long_length * X[u].y;
              pBx = X[u].x; pBy = X[u].y;
            } else { pAx = X[u].x; pAy = X[u].y; pBx = Y[u].x; pBy = Y[u].y; }  // :725-730
            const double ax_ = pAx - bx[0], ay_ = pAy - by[0], bx_ = pBx - bx[0], by_ = pBy - by[0];
            // :743-745  sqrt(d^2) - hulldist > 0 for both points.  The difference of two doubles is positive exactly when the first is the
            // larger, and the rounded root exceeds hulldist for every d^2 above hulldist^2 (1 + 1e-12), never below hulldist^2 (1 - 1e-12) (a
            // margin of ten thousand roundings): the two roots are only taken when a lane of the round falls in between
            const double d2a = ax_ * ax_ + ay_ * ay_, d2b = bx_ * bx_ + by_ * by_;
            const double h2 = hulldist * hulldist, h2hi = h2 * (1.0 + 1e-12), h2lo = h2 * (1.0 - 1e-12);
            bool ga = d2a > h2hi, gb = d2b > h2hi;
            if (__ballot(att && ((!ga && !(d2a < h2lo)) || (!gb && !(d2b < h2lo)))) != 0ull) { ga = sqrt(d2a) - hulldist > 0; gb = sqrt(d2b) - hulldist > 0; }
            if (att && ga && gb) att = false;
            const int c = n_plain + jj[u] * kBend + (k - 1);
            const unsigned long long mask = __ballot(att);
            if (att) sAtt[n_list + __popcll(mask & ((1ull << lane) - 1ull))] = (unsigned short)(tag | c);
            n_list += __popcll(mask); n_att += __popcll(mask);
          }
        }
        c0 += 64 * nr;
        SEP_PT(10);
      }
    }
    flush();
  }
  __syncthreads();
  if (lane < seg_hi - seg_lo) {
    const int seg = seg_lo + lane;
    const long o = (long)slot * NEP_MAX_POL + seg;
    int cn_ = sCnt[seg * 6], cf_ = sCnt[seg * 6 + 1];
    bool spill_ = false;
    if (cn_ + cf_ > sp.lines_cap) { spill_ = true; if (ps.flags) atomicOr(ps.flags, NEP_FLAG_LINES); if (cn_ > sp.lines_cap) cn_ = sp.lines_cap; if (cf_ > sp.lines_cap - cn_) cf_ = sp.lines_cap - cn_; }      // (bucket smaller than the worst case: flagged, and the replan fails — see separator_body)
    ps.line_cnt[o] = spill_ ? -1 - cn_ : cn_;
    if (ps.line_far) ps.line_far[o] = cf_;
    if (ps.line_skip) ps.line_skip[o] = sCnt[seg * 6 + 4];
    ps.lp_stats[o * 2] = sCnt[seg * 6 + 3] + sCnt[seg * 6 + 4]; ps.lp_stats[o * 2 + 1] = sCnt[seg * 6 + 2];
  }
  SEP_PT(8);
#ifdef NEP_SEP_PROF
  if (lane == 0 && blockIdx.x < 16384) { for (int k = 0; k < 12; k++) g_sep_prof[blockIdx.x * 16 + k] += (unsigned long long)pa_[k]; g_sep_prof[blockIdx.x * 16 + 15] += 1ull; }
#endif
}
#ifdef NEP_SEP_PROF
extern "C" int nep_debug_sep_prof(unsigned long long* out16, int reset) {      // sums over the blocks
  std::vector<unsigned long long> h((size_t)16384 * 16);
  (void)hipMemcpyFromSymbol(h.data(), HIP_SYMBOL(g_sep_prof), h.size() * 8);
  if (out16) { for (int k = 0; k < 16; k++) out16[k] = 0; for (size_t b = 0; b < 16384; b++) for (int k = 0; k < 16; k++) out16[k] += h[b * 16 + k]; }
  if (reset) { std::fill(h.begin(), h.end(), 0ull); (void)hipMemcpyToSymbol(HIP_SYMBOL(g_sep_prof), h.data(), h.size() * 8); }
  return 0;
}
#endif

// The redo pass of the presolve: every segment of the replans the QP kernel listed (ps.redo_list / ps.redo_count: a parked
// line violated, or the solution moved farther from the guess than the skipped LPs allow) with every LP solved and every line
// in call order, for the full re-solve that follows.  A fixed small grid walks the list (it is empty nearly always).
template <int RULE>
__global__ __launch_bounds__(64, NEP_SEP_WAVES) void separator_redo_kernel(SceneParams sp, ProblemSet ps, int pool_pairs) {
  const int n = *ps.redo_count * NEP_MAX_POL;
  for (int i = blockIdx.x; i < n; i += gridDim.x) {
    separator_body<RULE>(sp, ps, pool_pairs, ps.redo_list[i / NEP_MAX_POL], i % NEP_MAX_POL, false);
    __syncthreads();
  }
}

size_t separator_lds_bytes(const SceneParams& sp) {
  const int total = sp.n_hull + sp.num_agents + sp.n_static + (sp.ent_enabled ? sp.num_agents * kBend : 0);
  const size_t tail = 8 * sizeof(double) + (((size_t)(total + 8 + (sp.ent_enabled ? sp.num_agents : 0)) * sizeof(unsigned short) + 15) & ~(size_t)15);   // (+ the agents with an active entangle case)
  // the pool takes what is left of 10 KB (sixteen waves per CU); with very long candidate lists (config 5 with the entangle rows)
  // it keeps at least 64 x 8 pairs and the wave gets more LDS
  size_t pool = tail + 64 * 8 * 16 <= (size_t)kSepLdsTarget ? (size_t)kSepLdsTarget - tail : (size_t)64 * 8 * 16;
  pool &= ~(size_t)15;
  return pool + tail;
}
static int separator_pool_pairs(const SceneParams& sp) {
  const int total = sp.n_hull + sp.num_agents + sp.n_static + (sp.ent_enabled ? sp.num_agents * kBend : 0);
  const size_t tail = 8 * sizeof(double) + (((size_t)(total + 8 + (sp.ent_enabled ? sp.num_agents : 0)) * sizeof(unsigned short) + 15) & ~(size_t)15);
  return (int)((separator_lds_bytes(sp) - tail) / 16);
}

void launch_separator(int n_slots, const SceneParams& sp, const ProblemSet& ps, hipStream_t st) {
  if (n_slots <= 0) return;
  // (only with the spatial presolve: with every LP to solve a segment fills its wave by itself — 64 to 68 LPs — and the packed form is
  // slower, 0.53 against 0.41 ms per 4.2 M LPs: its step 1 is serial over the segments and its lanes hold different control points)
  const int total = sp.n_hull + sp.num_agents + sp.n_static + (sp.ent_enabled ? sp.num_agents * kBend : 0);
  // the packed kernel's list entries are (segment << 13 | candidate) in 16 bits: candidates beyond 8 191 (about 800 agents with the
  // entangle rows, 4 000 without) take the unpacked kernel, whose entries hold 65 535 (size_scratch refuses more)
  if (((ps.skip_box && ps.line_far && sp.cull_radius > 0.0 && ps.sep_pack >= 0) || (sp.cull_radius == 0.0 && ps.sep_pack >= 1)) && sp.sep_rule == 0 && total <= 8191) {
    // (the wave's LDS stays within the 10 KB sixteen waves per CU allow: the pool of point sets takes what the tables leave, 64 x 8 pairs at least)
    const size_t rounds_ = (size_t)((sp.n_hull + 63) / 64 + (sp.num_agents + 63) / 64 + (sp.n_static + 63) / 64);
    const size_t extras = 15 * NEP_MAX_POL * sizeof(double) + 2 * NEP_MAX_POL * rounds_ * sizeof(unsigned long long) + 6 * NEP_MAX_POL * sizeof(int)
                          + ((sp.ent_enabled && ps.case_id) ? (size_t)NEP_MAX_POL * ((sp.num_agents + 63) / 64) * sizeof(unsigned long long) : 0)
                          + (size_t)(sep_packed_cap(sp.n_hull + sp.num_agents + sp.n_static) + (sp.ent_enabled ? sp.num_agents : 0)) * sizeof(unsigned short);
    size_t pool_b = extras + 64 * 6 * 16 <= (size_t)kSepLdsTarget ? (size_t)kSepLdsTarget - extras : (size_t)64 * 6 * 16;
    pool_b &= ~(size_t)15;
    const int pairs = (int)(pool_b / 16);
    const size_t lds_p = (pool_b + extras + 15) & ~(size_t)15;
    static DynLdsAttr attr_p;
    (void)attr_p.ensure((const void*)separator_packed_kernel, lds_p);
    int pack = 1; while (pack < NEP_MAX_POL && (long)n_slots * (NEP_MAX_POL / (pack * 2)) >= 4096) pack *= 2;      // (at least ~4 000 waves while the launch allows it)
    if (ps.sep_pack >= 1 && ps.sep_pack <= NEP_MAX_POL) pack = ps.sep_pack;
    const int groups = (NEP_MAX_POL + pack - 1) / pack;
    hipLaunchKernelGGL(separator_packed_kernel, dim3(n_slots * groups), dim3(64), lds_p, st, sp, ps, pairs, pack);
    return;
  }
  const size_t lds = separator_lds_bytes(sp);
  static DynLdsAttr attr[2];
  if (sp.sep_rule == 1) {
    (void)attr[1].ensure((const void*)separator_kernel<1>, lds);
    hipLaunchKernelGGL(separator_kernel<1>, dim3(n_slots * NEP_MAX_POL), dim3(64), lds, st, sp, ps, separator_pool_pairs(sp));
  } else {
    (void)attr[0].ensure((const void*)separator_kernel<0>, lds);
    hipLaunchKernelGGL(separator_kernel<0>, dim3(n_slots * NEP_MAX_POL), dim3(64), lds, st, sp, ps, separator_pool_pairs(sp));
  }
}
void launch_separator_redo(int n_slots, const SceneParams& sp, const ProblemSet& ps, hipStream_t st) {
  if (n_slots <= 0 || !ps.redo_count || !ps.redo_list) return;
  const size_t lds = separator_lds_bytes(sp);
  static DynLdsAttr attr[2];
  const int grid = n_slots * NEP_MAX_POL < 1024 ? n_slots * NEP_MAX_POL : 1024;
  if (sp.sep_rule == 1) {
    (void)attr[1].ensure((const void*)separator_redo_kernel<1>, lds);
    hipLaunchKernelGGL(separator_redo_kernel<1>, dim3(grid), dim3(64), lds, st, sp, ps, separator_pool_pairs(sp));
  } else {
    (void)attr[0].ensure((const void*)separator_redo_kernel<0>, lds);
    hipLaunchKernelGGL(separator_redo_kernel<0>, dim3(grid), dim3(64), lds, st, sp, ps, separator_pool_pairs(sp));
  }
}

// Stand-alone batched LP (tests / nep_separator_batch): one lane per problem, A read from global
// memory into a private array, B = four points (what every call site of the path passes).
__global__ void separator_explicit_kernel(int n_prob, const int* __restrict__ a_off, const double* __restrict__ a_xy,
                                          const int* __restrict__ b_off, const double* __restrict__ b_xy,
                                          double* __restrict__ nd_out, int* __restrict__ solved, int rule) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= n_prob) return;
  double2 A[kHullV];
  Pts4 B;
  int nA = a_off[p + 1] - a_off[p];
  if (nA > kHullV) nA = kHullV;
  for (int i = 0; i < nA; i++) A[i] = make_double2(a_xy[2 * (a_off[p] + i)], a_xy[2 * (a_off[p] + i) + 1]);
  for (int i = 0; i < 4; i++) { B.x[i] = b_xy[2 * (b_off[p] + i)]; B.y[i] = b_xy[2 * (b_off[p] + i) + 1]; }
  double nd[3];
  const bool ok = rule == 1 ? separator_glpk_class(nA, A, B, nd) : separator_impl(nA, A, 0, B, nd);
  nd_out[3 * p] = nd[0]; nd_out[3 * p + 1] = nd[1]; nd_out[3 * p + 2] = nd[2];
  solved[p] = ok ? 1 : 0;
}

void launch_separator_explicit(int n_prob, const int* a_off, const double* a_xy, const int* b_off,
                               const double* b_xy, double* nd, int* solved, int rule, hipStream_t st) {
  if (n_prob <= 0) return;
  hipLaunchKernelGGL(separator_explicit_kernel, dim3((n_prob + 63) / 64), dim3(64), 0, st, n_prob, a_off, a_xy, b_off, b_xy, nd, solved, rule);
}

// Stand-alone hulls (tests / nep_hulls_batch): full vertex lists of both hulls.
__global__ __launch_bounds__(64) void hull_explicit_kernel(const nep_traj_rec* __restrict__ recs, double t_start, int num_pol,
                                                           double T_span, double drone_radius,
                                                           double* __restrict__ hull_xy, int* __restrict__ hull_nv,
                                                           double* __restrict__ hull0_xy, int* __restrict__ hull0_nv, int* __restrict__ flags) {
  const int i = blockIdx.x % num_pol;
  const int jt = blockIdx.x / num_pol;
  hull_body(recs + jt, t_start, i, T_span, drone_radius, (long)jt * num_pol + i, true, hull_xy, hull_nv, hull0_xy, hull0_nv, flags);
}

void launch_hulls_explicit(const nep_traj_rec* recs, int n_traj, double t_start, int num_pol,
                           double T_span, double drone_radius, double* hull_xy, int* hull_nv,
                           double* hull0_xy, int* hull0_nv, int* flags, hipStream_t st) {
  if (n_traj * num_pol <= 0) return;
  hipLaunchKernelGGL(hull_explicit_kernel, dim3(n_traj * num_pol), dim3(64), 0, st, recs, t_start, num_pol, T_span,
                     drone_radius, hull_xy, hull_nv, hull0_xy, hull0_nv, flags);
}

// ---------------------------------------------------------------------------------------------
// post-solve safety check (neptune.cpp:719-806, gjk.cpp:76-149)
// ---------------------------------------------------------------------------------------------
template <class VP> __device__ __forceinline__ int gjk_furthest(int n, VP V, double dx, double dy) {
  double mx = dx * V[0] + dy * V[1]; int idx = 0;
  for (int i = 1; i < n; i++) { const double p = dx * V[2 * i] + dy * V[2 * i + 1]; if (p > mx) { mx = p; idx = i; } }
  return idx;
}
// furthest of the four points of B along (dx, dy) — the same first-maximum rule as gjk_furthest, with the points
// in registers (an indexed private copy would live in scratch memory and turn every support query into
// memory round trips)
__device__ __forceinline__ void gjk_furthest4(const Pts4& B, double dx, double dy, double& px, double& py) {
  double mx = dx * B.x[0] + dy * B.y[0]; px = B.x[0]; py = B.y[0];
#pragma unroll
  for (int i = 1; i < 4; i++) { const double p = dx * B.x[i] + dy * B.y[i]; if (p > mx) { mx = p; px = B.x[i]; py = B.y[i]; } }
}
// gjk::collision(vertices1 = V1 [n1][2], vertices2 = the four control points in B)
// (VP: the vertices' pointer type — the front end passes an LDS-typed pointer for the obstacles it staged: through a generic pointer the
// support queries' reads are FLAT loads, a vector-memory round trip each even when the address is LDS)
template <class VP> __device__ bool gjk_collision(int n1, VP V1, const Pts4& B) {
  if (n1 <= 0) return false;
  double p1x = 0, p1y = 0, p2x = 0, p2y = 0;
  for (int i = 0; i < n1; i++) { p1x += V1[2 * i]; p1y += V1[2 * i + 1]; }
#pragma unroll
  for (int i = 0; i < 4; i++) { p2x += B.x[i]; p2y += B.y[i]; }
  p1x /= n1; p1y /= n1; p2x /= 4; p2y /= 4;
  double dx = p1x - p2x, dy = p1y - p2y;
  if (dx == 0 && dy == 0) dx = 1.0;
  double s0x, s0y, s1x = 0, s1y = 0;   // simplex columns 0 and 1 (column 2 is always `a`)
  int i1 = gjk_furthest(n1, V1, dx, dy);
  double qx, qy;
  gjk_furthest4(B, -dx, -dy, qx, qy);
  double ax = V1[2 * i1] - qx, ay = V1[2 * i1 + 1] - qy;
  s0x = ax; s0y = ay;
  if (ax * dx + ay * dy <= 0) return false;
  dx = -ax; dy = -ay;
  int index = 0;
  for (int iter = 0; iter < 64; iter++) {
    ++index;
    i1 = gjk_furthest(n1, V1, dx, dy); gjk_furthest4(B, -dx, -dy, qx, qy);
    ax = V1[2 * i1] - qx; ay = V1[2 * i1 + 1] - qy;
    if (index == 1) { s1x = ax; s1y = ay; }
    if (ax * dx + ay * dy <= 0) return false;
    const double aox = -ax, aoy = -ay;
    if (index < 2) {
      const double abx = s0x - ax, aby = s0y - ay;
      // tripleProduct(ab, ao, ab) = ao*(ab.ab) - ab*(ao.ab)
      const double ac_ = abx * abx + aby * aby, bc_ = aox * abx + aoy * aby;
      dx = aox * ac_ - abx * bc_; dy = aoy * ac_ - aby * bc_;
      if (sqrt(dx * dx + dy * dy) == 0) { dx = aby; dy = -abx; }
      continue;
    }
    const double bx_ = s1x, by_ = s1y, cx_ = s0x, cy_ = s0y;
    const double abx = bx_ - ax, aby = by_ - ay, acx = cx_ - ax, acy = cy_ - ay;
    // acperp = tripleProduct(ab, ac, ac) = ac*(ab.ac) - ab*(ac.ac)
    double t1 = abx * acx + aby * acy, t2 = acx * acx + acy * acy;
    const double apx = acx * t1 - abx * t2, apy = acy * t1 - aby * t2;
    if (apx * aox + apy * aoy >= 0) { dx = apx; dy = apy; }
    else {
      // abperp = tripleProduct(ac, ab, ab) = ab*(ac.ab) - ac*(ab.ab)
      t1 = acx * abx + acy * aby; t2 = abx * abx + aby * aby;
      const double bpx = abx * t1 - acx * t2, bpy = aby * t1 - acy * t2;
      if (bpx * aox + bpy * aoy < 0) return true;
      s0x = s1x; s0y = s1y;
      dx = bpx; dy = bpy;
    }
    s1x = ax; s1y = ay;   // simplex.col(1) = simplex.col(2)
    --index;
  }
  return false;
}

// conflict[scene][a][j] = agent a's new trajectory hits the interval hulls of agent j's new
// trajectory (trajsAndPwpAreInCollision2d on the round's interval grid).  One wave per (scene, a);
// lanes stride over j.  Hulls come from hull_kernel run on the new records.
__global__ __launch_bounds__(256) void safety_conflict_kernel(const nep_traj_rec* __restrict__ fresh, const nep_traj_rec* __restrict__ other, int N, int num_pol, double T_span,
                                                             const double* __restrict__ hull_xy, const int* __restrict__ hull_nv,
                                                             unsigned char* __restrict__ conflict) {
  // Four agents of a scene per workgroup, one per wave: the other agents' hulls they are tested against are the same, so the
  // staged copy in LDS is shared (17 KB per workgroup instead of per wave: the register file, not LDS, bounds the occupancy).
  __shared__ double sBx[4][NEP_MAX_POL * 4], sBy[4][NEP_MAX_POL * 4];
  __shared__ double2 sV[64][kHullV + 1];
  __shared__ int sNv[64];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int groups = (N + 3) >> 2;
  const int scene = blockIdx.x / groups, a = (blockIdx.x - scene * groups) * 4 + w;
  const bool mine = a < N;                                      // (a workgroup's last waves may have no agent: they still stage)
  const nep_traj_rec* ra = fresh + (long)scene * N + (mine ? a : 0);
  const int Ka = (mine && ra->valid) ? (ra->pwp.n_seg < num_pol ? ra->pwp.n_seg : num_pol) : 0;
  if (lane < 4 * NEP_MAX_POL) {  // my control points, P * A_rest_pos_basis_t_inverse_ (neptune.cpp:789)
    const int seg = lane >> 2, k = lane & 3;
    double vx = 0, vy = 0;
    if (seg < Ka) {
      const double tp0 = T_span * T_span * T_span, tp1 = T_span * T_span, tp2 = T_span;
      const double m0 = tp0 * cAPosInv[0][k], m1 = tp1 * cAPosInv[1][k], m2 = tp2 * cAPosInv[2][k], m3 = 1.0 * cAPosInv[3][k];
      const double* Px = ra->pwp.coeff[0][seg]; const double* Py = ra->pwp.coeff[1][seg];
      vx = ((Px[0] * m0 + Px[1] * m1) + Px[2] * m2) + Px[3] * m3;
      vy = ((Py[0] * m0 + Py[1] * m1) + Py[2] * m2) + Py[3] * m3;
    }
    sBx[w][lane] = vx; sBy[w][lane] = vy;
  }
  __syncthreads();
  // One GJK test per lane: the hulls of `ac` agents (all their intervals: contiguous in hull_xy) are brought into LDS with
  // coalesced loads, lane (jj, i) tests agent j0 + jj's interval i against my interval i, and a ballot gives every agent its
  // verdict.  (One agent per lane looping over the intervals read each hull with a 2 KB stride between lanes — 64 cache lines
  // per load instruction: the kernel was bound by that gather.)  Rows are padded to 17 pairs against bank conflicts.
  const int ac = 64 / num_pol;                                   // agents per round
  const int jj = lane / num_pol, ii = lane - jj * num_pol;
  for (int j0 = 0; j0 < N; j0 += ac) {
    const int na = N - j0 < ac ? N - j0 : ac, cnt = na * num_pol;            // hulls this round
    const long h0 = ((long)scene * N + j0) * num_pol;
    const double2* src = (const double2*)hull_xy + h0 * kHullV;
    for (int e = tid; e < cnt * kHullV; e += 256) sV[e / kHullV][e % kHullV] = src[e];
    if (tid < cnt) sNv[tid] = hull_nv[h0 + tid];
    __syncthreads();
    bool hit = false;
    const int j = j0 + jj;
    if (lane < cnt && j != a && ii < Ka) {
      const nep_traj_rec* rj = other + (long)scene * N + j;     // whose hulls these are (the new or the previous records)
      if (rj->valid && rj->is_agent) {
        Pts4 B;
#pragma unroll
        for (int k = 0; k < 4; k++) { B.x[k] = sBx[w][ii * 4 + k]; B.y[k] = sBy[w][ii * 4 + k]; }
        hit = gjk_collision(sNv[lane], (const double*)&sV[lane][0], B);
      }
    }
    const unsigned long long bal = __ballot(hit);
    if (mine && lane < cnt && ii == 0) conflict[((long)scene * N + a) * N + j] = ((bal >> lane) & ((1ull << num_pol) - 1ull)) ? 1 : 0;
    __syncthreads();
  }
}

// Agents visited by id; an agent keeps its new trajectory unless it conflicts (either direction)
// with an already accepted lower id.  One workgroup per scene; then the final records are written.
// The conflict matrix is first folded into one symmetric bit row per agent in LDS (all threads, coalesced passes over the
// two byte matrices), then one thread walks the ids — accepted(a) = !forced_bad(a) && (row(a) & accepted) == 0, a few words per
// agent — instead of three workgroup barriers and two strided byte gathers per agent (63 -> 8 us per launch of 128 scenes).
constexpr int kResolveParts = 8;
__global__ __launch_bounds__(256) void safety_resolve_kernel(const nep_traj_rec* __restrict__ prev, const nep_traj_rec* __restrict__ fresh, int N,
                                                             const unsigned char* __restrict__ conflict, const unsigned char* __restrict__ conflict_prev,
                                                             const int* __restrict__ entangles, nep_traj_rec* __restrict__ final_out, int* __restrict__ accept_out) {
  extern __shared__ int sAcc[];   // [N] accept flags, [N] forced-bad flags, [N][W] symmetric conflict rows (32-bit words), [W] accepted ids
  // (kResolveParts workgroups per scene: each works out the accept flags — a few thousand bytes — and writes its share of the final
  // records: the copy, 120 KB per 64-agent scene, is what takes time, and one workgroup per scene left half of the chip idle)
  const int tid = threadIdx.x, scene = blockIdx.x / kResolveParts, part = blockIdx.x % kResolveParts;
  const int W = (N + 31) >> 5;
  int* sBad = sAcc + N;
  unsigned* sRow = (unsigned*)(sBad + N);
  unsigned* sBits = sRow + N * W;      // [W] accepted ids so far, when they do not fit eight registers (N > 256)
  const unsigned char* Cm = conflict + (long)scene * N * N;
  const unsigned char* Cp = conflict_prev ? conflict_prev + (long)scene * N * N : nullptr;
  for (int e = tid; e < N * W + W; e += blockDim.x) sRow[e] = 0u;
  for (int a = tid; a < N; a += blockDim.x) sBad[a] = (entangles && entangles[(long)scene * N + a] != 0) ? 1 : 0;      // entangleCheckGivenPwp (neptune.cpp:746-754)
  __syncthreads();
  for (long e = tid; e < (long)N * N; e += blockDim.x) {
    const int a = (int)(e / N), j = (int)(e % N);
    if (Cm[e] && a != j) { atomicOr(&sRow[a * W + (j >> 5)], 1u << (j & 31)); atomicOr(&sRow[j * W + (a >> 5)], 1u << (a & 31)); }
    // (optional) the new trajectory must also clear what everybody else is flying now: whoever is turned down this round keeps
    // exactly that
    if (Cp && Cp[e] && a != j) sBad[a] = 1;
  }
  __syncthreads();
  if (tid == 0) {
    unsigned acc[8];                                  // accepted ids so far (N <= 256: the path's sizes; beyond that the words live in LDS)
#pragma unroll
    for (int w = 0; w < 8; w++) acc[w] = 0u;
    for (int a = 0; a < N; a++) {
      bool bad = sBad[a] != 0;
      if (W <= 8) {
#pragma unroll
        for (int w = 0; w < 8; w++) if (w < W) bad |= (sRow[a * W + w] & acc[w]) != 0u;
      } else {
        for (int w = 0; w < W; w++) bad |= (sRow[a * W + w] & sBits[w]) != 0u;
      }
      sAcc[a] = bad ? 0 : 1;
      if (!bad) {
        if (W <= 8) {
#pragma unroll
          for (int w = 0; w < 8; w++) if (w == (a >> 5)) acc[w] |= 1u << (a & 31);
        } else sBits[a >> 5] |= 1u << (a & 31);
      }
    }
  }
  __syncthreads();
  const int words = (int)(sizeof(nep_traj_rec) / sizeof(double));
  for (long e = (long)part * blockDim.x + tid; e < (long)N * words; e += (long)blockDim.x * kResolveParts) {
    const int a = (int)(e / words), w = (int)(e % words);
    const double* src = (const double*)((sAcc[a] ? fresh : prev) + (long)scene * N + a);
    ((double*)(final_out + (long)scene * N + a))[w] = src[w];
  }
  if (accept_out && part == 0) for (int a = tid; a < N; a += blockDim.x) accept_out[(long)scene * N + a] = sAcc[a];
}

void launch_safety(const nep_traj_rec* prev, const nep_traj_rec* fresh, int n_scenes, int N, const SceneParams& sp, const ProblemSet& ps,
                   unsigned char* conflict, unsigned char* conflict_prev, const int* entangles, nep_traj_rec* final_out, int* accept_out, hipStream_t st) {
  if (n_scenes * N <= 0) return;
  auto hulls_of = [&](const nep_traj_rec* recs) {      // interval hulls of one record set on the round's grid (eight per wave, as in the replan)
    if (sp.num_pol <= 8 && (sp.hull_mode ? sp.hull_mode == 2 : (long)n_scenes * N > 2048))
      hipLaunchKernelGGL(hull_group_kernel, dim3(n_scenes * N), dim3(64), 0, st, recs, N, &ps.guess->t_start, (long)sizeof(nep_guess) * sp.n_local, sp.num_pol, sp.T_span,
                         sp.drone_radius, ps.hull_xy, ps.hull_nv, (double*)nullptr, (int*)nullptr, (double*)nullptr, (int*)nullptr, ps.flags,
                         (double*)nullptr, 0, (int*)nullptr, n_scenes * N, 0, (const int*)nullptr, (int*)nullptr, (int*)nullptr);
    else
      hipLaunchKernelGGL(hull_kernel, dim3(n_scenes * N * sp.num_pol), dim3(64), 0, st, recs, N, &ps.guess->t_start, (long)sizeof(nep_guess) * sp.n_local, sp.num_pol, sp.T_span,
                         sp.drone_radius, ps.hull_xy, ps.hull_nv, (double*)nullptr, (int*)nullptr, (double*)nullptr, (int*)nullptr, ps.flags);
  };
  if (conflict_prev) {   // new trajectories against the hulls of the PREVIOUS records on the same grid
    hulls_of(prev);
    hipLaunchKernelGGL(safety_conflict_kernel, dim3(n_scenes * ((N + 3) / 4)), dim3(256), 0, st, fresh, prev, N, sp.num_pol, sp.T_span, ps.hull_xy, ps.hull_nv, conflict_prev);
  }
  hulls_of(fresh);
  hipLaunchKernelGGL(safety_conflict_kernel, dim3(n_scenes * ((N + 3) / 4)), dim3(256), 0, st, fresh, fresh, N, sp.num_pol, sp.T_span, ps.hull_xy, ps.hull_nv, conflict);
  // (2 N + (N + 1) ceil(N / 32)) ints of dynamic LDS: quadratic in N — above the 64 KB default from N ~ 690 on, so the limit is
  // raised explicitly (160 KB per CU: N up to ~1 100; beyond that the launch fails and HIPCHK(hipGetLastError()) reports it)
  const size_t resolve_lds = (size_t)(2 * N + (N + 1) * ((N + 31) / 32)) * sizeof(int);
  static DynLdsAttr resolve_attr;
  (void)resolve_attr.ensure((const void*)safety_resolve_kernel, resolve_lds);
  hipLaunchKernelGGL(safety_resolve_kernel, dim3(n_scenes * kResolveParts), dim3(256), resolve_lds, st, prev, fresh, N, conflict, conflict_prev, entangles, final_out, accept_out);
}


// ---------------------------------------------------------------------------------------------
// front-end initial guess (include/neptune_frontend.h) — the deterministic
// beam over KinodynamicSearch's jerk lattice (kinodynamic_search.cpp:1045-1228, :1240-1385, :1514-1553,
// :1629-1827).  One 256-thread workgroup per (scene, agent): per depth the lattice children of the
// beam are generated and pruned one per thread, one node per voxel survives, the beam_width best in
// (g + bias h, parent rank, lattice index) order form the next beam.  Arithmetic is written in the
// oracle's association order (this file is built -ffp-contract=off): guesses match bit for bit.
// ---------------------------------------------------------------------------------------------
__constant__ double cAVelInv[3][3] = {
    {-0.07735026918962577, 0.16666666666666635, 1.077350269189625},
    {-0.07735026918962577, 0.49999999999999967, 1.077350269189625},
    {1.0000000000000002, 1.0000000000000009, 1.0000000000000016}};

struct FeChild { double e[6], cx[4], cy[4], Qx[4], Qy[4], g, dist, f; int vx, vy; };

__device__ __forceinline__ void fe_pos_cps(const double P[4], double T, double Q[4]) {
  const double tp[4] = {T * T * T, T * T, T, 1.0};
#pragma unroll
  for (int k = 0; k < 4; k++) Q[k] = ((P[0] * (tp[0] * cAPosInv[0][k]) + P[1] * (tp[1] * cAPosInv[1][k])) + P[2] * (tp[2] * cAPosInv[2][k])) + P[3] * (tp[3] * cAPosInv[3][k]);
}
__device__ __forceinline__ void fe_vel_cps(const double P[4], double T, double Qv[3]) {
  const double tv[3] = {T * T, T, 1.0}; const double m321[3] = {3.0, 2.0, 1.0};
#pragma unroll
  for (int k = 0; k < 3; k++) Qv[k] = (P[0] * (m321[0] * (tv[0] * cAVelInv[0][k])) + P[1] * (m321[1] * (tv[1] * cAVelInv[1][k]))) + P[2] * (m321[2] * (tv[2] * cAVelInv[2][k]));
}

// Neptune::getInitialZPwp (neptune.cpp:1727-1810): the guess's height profile, one thread (same arithmetic as the oracle)
__device__ void fe_initial_z(double p0, double v0, double a0, double z_final, double T, int np, double v_max_z, double a_max_z, double (*co)[4]) {
  double q[NEP_MAX_POL + 3], v[NEP_MAX_POL + 2];
  if (np < 3) { for (int i = 0; i < np; i++) { co[i][0] = co[i][1] = co[i][2] = 0; co[i][3] = p0; } return; }
  if (v0 < -v_max_z) v0 = -v_max_z; else if (v0 > v_max_z) v0 = v_max_z;
  if (a0 < -a_max_z) a0 = -a_max_z; else if (a0 > a_max_z) a0 = a_max_z;
  for (int i = 0; i < np + 2; i++) v[i] = 0;
  q[0] = p0;
  q[1] = p0 + T * v0 / 3;
  q[2] = (3 * 3 * q[1] - 2 * T * (-a0 * T + v0) - 3 * (q[1] + (-2 * T) * v0)) / 6;
  q[np] = z_final;
  const double increment = (z_final - q[2]) / (np - 2);
  for (int i = 3; i <= np - 1; i++) q[i] = q[i - 1] + increment;
  for (int i = 3; i <= np; i++) {
    v[i - 1] = (q[i] - q[i - 1]) / T;
    if (v[i - 1] > v_max_z) { q[i] = q[i - 1] + v_max_z * T; v[i - 1] = v_max_z; }
    else if (v[i - 1] < -v_max_z) { q[i] = q[i - 1] - v_max_z * T; v[i - 1] = -v_max_z; }
  }
  for (int i = 2; i <= np - 1; i++) {
    const double a_i = (v[i] - v[i - 1]) / T;
    if (a_i > a_max_z) v[i] = v[i - 1] + a_max_z * T;
    else if (a_i < -a_max_z) v[i] = v[i - 1] - a_max_z * T;
    q[i + 1] = q[i] + v[i] * T;
  }
  q[np + 1] = q[np]; q[np + 2] = q[np];
  for (int i = 0; i < np; i++) {
    const double* s = q + i;
    const double c0 = (((1.0 / 6.0) * s[0] + (4.0 / 6.0) * s[1]) + (1.0 / 6.0) * s[2]) + (0.0 / 6.0) * s[3];
    const double c1 = (((-3.0 / 6.0) * s[0] + (0.0 / 6.0) * s[1]) + (3.0 / 6.0) * s[2]) + (0.0 / 6.0) * s[3];
    const double c2 = (((3.0 / 6.0) * s[0] + (-6.0 / 6.0) * s[1]) + (3.0 / 6.0) * s[2]) + (0.0 / 6.0) * s[3];
    const double c3 = (((-1.0 / 6.0) * s[0] + (3.0 / 6.0) * s[1]) + (-3.0 / 6.0) * s[2]) + (1.0 / 6.0) * s[3];
    co[i][3] = 1.0 * c0; co[i][2] = (1 / T) * c1; co[i][1] = (1 / (T * T)) * c2; co[i][0] = (1 / (T * T * T)) * c3;
  }
}

// per-lattice-value terms of the primitive (pure functions of the jerk sample: computed once per kernel)
struct FeLattice { const double* j6; const double* dp; const double* dv; const double* da; };   // LDS tables [num_samples]
__device__ __forceinline__ void fe_lattice_fill(const SceneParams& sp, const nep_fe_cfg& fc, double* tab, int k) {   // tab: [4][NEP_FE_MAX_SAMPLES]
  const double tau = sp.T_span, j_min = -fc.j_max, j_max = fc.j_max;
  const double delta = (j_max - j_min) / (fc.num_samples - 1);
  const double j = j_min + k * delta;
  tab[k] = j / 6; tab[NEP_FE_MAX_SAMPLES + k] = (((j * tau) * tau) * tau) / 6; tab[2 * NEP_FE_MAX_SAMPLES + k] = ((j * tau) * tau) / 2; tab[3 * NEP_FE_MAX_SAMPLES + k] = j * tau;
}

// one lattice child (expandAndAddToQueue); false when a kinodynamic test prunes it.  Norm tests are done on
// the squares (no square root on the rejection paths); both sides of the parity check state them that way.
__device__ bool fe_child(const SceneParams& sp, const nep_fe_cfg& fc, const FeLattice& L, const double* __restrict__ pe, double pg, bool first, int jx, int jy,
                         double gx, double gy, double bx, double by, FeChild& o) {
  // Branch-free: the tests are gathered in one flag instead of returning at the first failure.  The lanes of a wave run in
  // lock-step — an early return saves nothing while one lane's child is alive — and a body without exits is one block the
  // compiler can interleave (the divisions and square roots are long dependent chains).  What a failed child leaves in `o` is
  // not used.
  const double tau = sp.T_span, j_max = fc.j_max, v_max = sp.v_max, v_min = -sp.v_max, a_max = sp.a_max, a_min = -sp.a_max;
  const int jk[2] = {jx, jy};
#pragma unroll
  for (int ax = 0; ax < 2; ax++) {
    const double p = pe[ax], v = pe[2 + ax], a = pe[4 + ax];
    o.e[ax] = ((p + v * tau) + ((a * tau) * tau) / 2) + L.dp[jk[ax]];
    o.e[2 + ax] = (v + a * tau) + L.dv[jk[ax]];
    o.e[4 + ax] = a + L.da[jk[ax]];
  }
  double n2 = 0;
#pragma unroll
  for (int i = 0; i < 6; i++) n2 += (o.e[i] - pe[i]) * (o.e[i] - pe[i]);
  bool ok = !(n2 < 0.00001 * 0.00001);
  ok &= !(o.e[5] > a_max || o.e[5] < a_min || o.e[4] > a_max || o.e[4] < a_min);
  o.cx[0] = L.j6[jx]; o.cx[1] = pe[4] / 2; o.cx[2] = pe[2]; o.cx[3] = pe[0];
  o.cy[0] = L.j6[jy]; o.cy[1] = pe[5] / 2; o.cy[2] = pe[3]; o.cy[3] = pe[1];
  fe_pos_cps(o.cx, tau, o.Qx); fe_pos_cps(o.cy, tau, o.Qy);
  const double cable2 = fc.cable_length * fc.cable_length;
#pragma unroll
  for (int i = 0; i < 4; i++) {
    ok &= !(o.Qx[i] < sp.mins[0] || o.Qx[i] > sp.maxs[0] || o.Qy[i] < sp.mins[1] || o.Qy[i] > sp.maxs[1]);
    ok &= !((o.Qx[i] - bx) * (o.Qx[i] - bx) + (o.Qy[i] - by) * (o.Qy[i] - by) > cable2);
  }
  if (!first) {      // (wave-uniform)
    double Vx[3], Vy[3];
    fe_vel_cps(o.cx, tau, Vx); fe_vel_cps(o.cy, tau, Vy);
#pragma unroll
    for (int i = 0; i < 3; i++) ok &= !(Vx[i] < v_min || Vx[i] > v_max || Vy[i] < v_min || Vy[i] > v_max);
  }
#pragma unroll
  for (int ax = 0; ax < 2; ax++) {
    const double a = o.e[4 + ax], v = o.e[2 + ax];
    // (the reference divides by j_min = -j_max on one side and by j_max on the other: the quotients are each other's exact
    // negatives and v - (-q) is the same operation as v + q — one IEEE division per axis instead of two, the same bits)
    const double q = ((0.5 * a) * a) / j_max;
    ok &= !((a > 0 && v + q > v_max) || (a < 0 && v - q < v_min));
  }
  const double arc = sqrt((o.e[0] - pe[0]) * (o.e[0] - pe[0]) + (o.e[1] - pe[1]) * (o.e[1] - pe[1]));
  o.g = pg + arc;
  o.dist = sqrt((o.e[0] - gx) * (o.e[0] - gx) + (o.e[1] - gy) * (o.e[1] - gy));
  o.f = o.g + fc.bias * o.dist;
  o.vx = (int)round(o.e[0] / fc.voxel_size); o.vy = (int)round(o.e[1] / fc.voxel_size);
  return ok;
}

// The same child again for a node that is known to have passed every test of fe_child (the beam's winners are re-derived when
// they are installed, the children on the GJK work list when their turn comes: there is no room to keep 800 children's states):
// end state, coefficients, g, distance and f by the same arithmetic, without the kinodynamic tests (whose divisions and branches
// are most of fe_child); WITH_Q: the position control points and the voxel as well.
template <bool WITH_Q>
__device__ __forceinline__ void fe_child_again(const SceneParams& sp, const nep_fe_cfg& fc, const FeLattice& L, const double* __restrict__ pe, double pg, int jx, int jy,
                                               double gx, double gy, FeChild& o) {
  const double tau = sp.T_span;
  const int jk[2] = {jx, jy};
#pragma unroll
  for (int ax = 0; ax < 2; ax++) {
    const double p = pe[ax], v = pe[2 + ax], a = pe[4 + ax];
    o.e[ax] = ((p + v * tau) + ((a * tau) * tau) / 2) + L.dp[jk[ax]];
    o.e[2 + ax] = (v + a * tau) + L.dv[jk[ax]];
    o.e[4 + ax] = a + L.da[jk[ax]];
  }
  o.cx[0] = L.j6[jx]; o.cx[1] = pe[4] / 2; o.cx[2] = pe[2]; o.cx[3] = pe[0];
  o.cy[0] = L.j6[jy]; o.cy[1] = pe[5] / 2; o.cy[2] = pe[3]; o.cy[3] = pe[1];
  const double arc = sqrt((o.e[0] - pe[0]) * (o.e[0] - pe[0]) + (o.e[1] - pe[1]) * (o.e[1] - pe[1]));
  o.g = pg + arc;
  o.dist = sqrt((o.e[0] - gx) * (o.e[0] - gx) + (o.e[1] - gy) * (o.e[1] - gy));
  o.f = o.g + fc.bias * o.dist;
  if constexpr (WITH_Q) {
    fe_pos_cps(o.cx, tau, o.Qx); fe_pos_cps(o.cy, tau, o.Qy);
    o.vx = (int)round(o.e[0] / fc.voxel_size); o.vy = (int)round(o.e[1] / fc.voxel_size);
  }
}

// LDS arrays are sized for the configured beam width (not the maximum), so that narrower beams leave room for
// more workgroups per CU: cap = beam_width * num_samples^2 candidates per depth, hash tables a power of two above
// 1.25x (per-depth voxel table) / 2x (visited voxels, beam_width * num_pol keys) their load.
struct FeSizes { int cap, dd, vis, mb; };
__host__ __device__ inline FeSizes fe_sizes(int beam_width, int num_samples, int num_pol) {
  FeSizes z; z.cap = beam_width * num_samples * num_samples; if (z.cap < 64) z.cap = 64;
  z.dd = 64; while (z.dd * 4 < z.cap * 5) z.dd *= 2;
  z.vis = 64; while (z.vis < 2 * beam_width * num_pol) z.vis *= 2;
  z.mb = beam_width <= 32 ? 32 : NEP_FE_MAX_BEAM;      // stride of the per-rank arrays (the beam's width rounded up: a width-32 beam does not pay for 64 ranks of LDS)
  return z;
}
constexpr unsigned long long kFeEmpty = ~0ull;
constexpr int kFeObsLds = 24;   // shortlisted obstacles whose vertices are staged in LDS (the rest are read from global memory)

__device__ __forceinline__ unsigned fe_hash(long long vox) { return (unsigned)(((unsigned long long)vox * 0x9E3779B97F4A7C15ull) >> 40); }

// The serial loops of this kernel run out of LDS with every load independent of the previous
// iteration's result (dense arrays, no early exits): they are latency-bound otherwise.
// ENT (fc.enable_entangle): every child additionally carries its parent's entangle state through entanglesWithOtherAgents
// (ent_device.h; pruned when that returns true), g is the sampled arc length, h gains 0.3 per crossing and 1.0 per bend
// point (kinodynamic_search.cpp:1177-1182), a voxel is (ix, iy, getIz(state)) (:1170-1173), the other agents' base squares
// are obstacles (collidesWithBases2d, :1583-1628) and only nodes whose active cases are all <= 1 may end the plan
// (:1693-1700).  The states of the installed nodes stay in global memory ([depth][rank]); the path's states give the case
// ids the back end consumes (solver_gurobi_poly.cpp:624-631).
// workgroups per CU the front end's register allocation is bounded for (a 256-thread workgroup is one wave per SIMD).  The
// search is latency-bound (eight depths of ~10 barriers): three workgroups per CU at 168 VGPRs (13 spilled) take 3.06 ms per
// 8 192 searches against 4.13 ms for two at 187; four at 128 (53 spilled) take 3.78.  The entangle instantiation keeps all
// its registers (326).
#ifndef NEP_FE_WAVES
#define NEP_FE_WAVES 3
#endif
// Boxes of the front end's obstacles — the other agents' interval hulls and the static polygons — once per launch: the 64
// searches of a scene, eight depths each, used to re-derive them from the sixteen vertex slots every time (a fifth of a search).
// [scene][num_agents + n_static][num_pol] x (x0, x1, y0, y1); an empty polygon gets a box nothing meets.
__global__ __launch_bounds__(256) void fe_box_kernel(SceneParams sp, ProblemSet ps, int n_scenes) {
  const int N = sp.num_agents, S = sp.n_static, D = sp.num_pol;
  const long t = (long)blockIdx.x * 256 + threadIdx.x;
  if (ps.redo_count && t < 4) ps.redo_count[t] = 0;      // (the presolve's redo list starts empty: this kernel runs before the separator — no memset node in the step's graph)
  if (t >= (long)n_scenes * (N + S) * D) return;
  const int idx = (int)(t % D); const long r = t / D;
  const int j = (int)(r % (N + S)), scene = (int)(r / (N + S));
  int nv; const double* V;
  if (j < N) { const HullRef hr = hull_ref(ps, sp.n_hull, scene, j); const long h = hr.e * sp.num_pol + idx; nv = blk(ps.hull_nv, hr.boff)[h]; V = blk(ps.hull_xy, hr.boff) + h * kHullV * 2; }
  else { const long js = (long)scene * sp.static_stride + (j - N); nv = ps.static_nv[js]; V = ps.static_xy + js * kHullV * 2; }
  double x0 = __builtin_huge_val(), x1 = -__builtin_huge_val(), y0 = __builtin_huge_val(), y1 = -__builtin_huge_val();
  if (nv > 0) {
    double2 vv[kHullV];
#pragma unroll
    for (int i = 0; i < kHullV; i++) vv[i] = ((const double2*)V)[i];
    x0 = x1 = vv[0].x; y0 = y1 = vv[0].y;
#pragma unroll
    for (int i = 1; i < kHullV; i++) {
      const double vx = i < nv ? vv[i].x : vv[0].x, vy = i < nv ? vv[i].y : vv[0].y;
      x0 = fmin(x0, vx); x1 = fmax(x1, vx); y0 = fmin(y0, vy); y1 = fmax(y1, vy);
    }
  }
  double* o = ps.fe_box + t * 4;
  o[0] = x0; o[1] = x1; o[2] = y0; o[3] = y1;
}
// (also what the separator's spatial presolve reads: ps.skip_box)
void launch_boxes(int n_scenes, const SceneParams& sp, const ProblemSet& ps, hipStream_t st) {
  const long nb = (long)n_scenes * (sp.num_agents + sp.n_static) * sp.num_pol;
  if (nb > 0) hipLaunchKernelGGL(fe_box_kernel, dim3((unsigned)((nb + 255) / 256)), dim3(256), 0, st, sp, ps, n_scenes);
}

// What the entangle check reads of agent j in interval i — does its trajectory exist, its tether's bend points, its ns + 1 samples of
// the interval — gathered from the hull data (one array or all-gathered blocks: hull_ref) into ONE record per (scene, agent,
// interval): the front end's proofs and its crossing detection visit an agent with a single round trip, every load independent.
__global__ __launch_bounds__(256) void ent_pack_kernel(SceneParams sp, ProblemSet ps, FeEntArgs ea, int n_scenes) {
  const int N = sp.num_agents, D = sp.num_pol, ns = ea.ns;
  const long t = (long)blockIdx.x * 256 + threadIdx.x;
  if (t == 0 && ea.big.count) { *ea.big.count = 0; if (ea.redo_count) *ea.redo_count = 0; }      // (the big-record pool and the list of searches that need it start empty: this kernel runs before the search — no memset node in a captured step)
  if (t >= (long)n_scenes * N * D) return;
  const int i = (int)(t % D); const long r = t / D;
  const int j = (int)(r % N), scene = (int)(r / N);
  const HullRef hr = hull_ref(ps, sp.n_hull, scene, j);
  double* o = ea.packed + t * ea.pk_stride;
  const int present = blk(ea.present, hr.boff)[hr.e], nb = blk(ps.bend_n, hr.boff)[hr.e];
  *(int2*)o = int2{present, nb};
  const double* bp = blk(ps.bend_xy, hr.boff) + hr.e * kBend * 2;
  o[1] = 0.0;
  for (int k = 0; k < 2 * kBend; k++) o[kEntPkBend + k] = k < 2 * nb ? bp[k] : 0.0;
  const double* sm = blk(ea.sampled, hr.boff) + ((hr.e * D + i) * (ns + 1)) * 2;
  for (int k = 0; k < 2 * (ns + 1); k++) o[kEntPkHead + k] = sm[k];
}

#ifndef NEP_FE_ENT_WGS
#define NEP_FE_ENT_WGS 3      // (round 4: 3 = 168 registers, 48 spilled, 52 KB of LDS at config 5) // workgroups per CU of the entangle instantiation: 2 = working records in global memory, registers bounded to 256 (19.9 ms per 2 048 config-5 searches); 1 = working records in LDS, 139 KB (27.4 ms: the list surgery is latency-bound, a second workgroup hides more than LDS saves)
#endif
// BIG (with ENT): the instantiation that re-runs, after the launch proper, the few searches in which a child's entangle state outgrew
// the fixed record (ea.redo_list): such a child — pruned and listed by the plain instantiation — is carried in a big record here
// (ent_device.h), bounded by the reference's own rule only.  A kernel of its own because the big-record pass compiled into the
// search every slot runs costs that search 40 % (1.43 -> 1.98 ms per search at config 5: its 372 spilled registers and 1.8 KB of
// scratch per lane weigh on the fixed record's path although a launch executes it for a dozen searches in 8 192).
template <bool ENT, int WGS, bool BIG = false>
__global__ __launch_bounds__(256, WGS) void frontend_kernel(SceneParams sp, ProblemSet ps, nep_fe_cfg fc, const nep_fe_start* __restrict__ starts,
                                                       nep_guess* __restrict__ guess_out, nep_fe_result* __restrict__ res_out, FeEntArgs ea) {
  extern __shared__ __attribute__((aligned(16))) double fe_smem[];
  const int tid = threadIdx.x;
  // (launch order: longest expected search first — the previous search of the same slot is the predictor, as for the QP's
  // workgroups, order_kernel; a search's time spreads 1 : 4 with the depth it ends at, and with the entangle check a search that
  // carries its nodes in big records takes several times the others': started last it would set the kernel's end by itself)
  const long long t_wg0 = (long long)wall_clock64();
  if constexpr (BIG) { if ((int)blockIdx.x >= *ea.redo_count) return; }      // (the list holds at most gridDim.x = ea.redo_cap entries)
  const int slot = BIG ? ea.redo_list[blockIdx.x] : (ps.fe_order ? ps.fe_order[blockIdx.x] : (int)blockIdx.x), scene = slot / sp.n_local, own = sp.first_local + (slot % sp.n_local);
  const int N = sp.num_agents, S = sp.n_static, W = fc.beam_width, ns = fc.num_samples, NC = ns * ns, D = sp.num_pol;
  const FeSizes fz = fe_sizes(W, ns, D);
  const int kFeCap = fz.cap, kFeDd = fz.dd, kFeVis = fz.vis, MB = fz.mb;
  // ---- LDS carve ----
  double* s_f = fe_smem;                                   // [kFeCap] f of candidate id
  double* b_end = s_f + kFeCap;                            // [2][64][6]
  double* b_g = b_end + 2 * MB * 6;           // [2][64]
  double* b_dist = b_g + 2 * MB;              // [64]
  double* b_f = b_dist + MB;                  // [64]
  double* p_box = b_f + MB;                   // [64][4] box of every parent's children
  double* o_aabb = p_box + MB * 4;            // [N+S][4] boxes of the shortlisted obstacles, dense
  double* o_V = o_aabb + 4 * (N + S);                      // [kFeObsLds][16][2] their vertices (GJK walks them several times)
  // [kFeCap] f of the voxel winners, dense: written by the compaction and read by the rank count, when the shortlist's boxes and
  // vertices are dead — it lives in their storage when they are big enough (6.4 KB of the 51 a search held: with the per-rank arrays
  // at the beam's width the carve drops below 40 KB, four workgroups per CU instead of three; with the entangle check, where the same
  // storage is lent to the crossing lists in between, below 53 KB at config 5: three instead of two)
  const bool rf_alias = 4 * (N + S) + kFeObsLds * kHullV * 2 >= kFeCap;
  double* r_f = rf_alias ? o_aabb : o_V + kFeObsLds * kHullV * 2;
  double* s_lat = o_V + kFeObsLds * kHullV * 2 + (rf_alias ? 0 : kFeCap);   // [4][NEP_FE_MAX_SAMPLES] lattice tables
  long long* s_vox = (long long*)(s_lat + 4 * NEP_FE_MAX_SAMPLES);   // [kFeCap]
  unsigned long long* v_key = (unsigned long long*)(s_vox + kFeCap);   // [kFeVis] visited voxels
  int* d_slot = (int*)(v_key + kFeVis);                    // [kFeDd] voxel -> best candidate of the depth
  int* o_nv = d_slot + kFeDd;                              // [N+S] dense
  int* o_id = o_nv + (N + S);                              // [N+S] dense: obstacle index (agent j or N + static)
  int* s_i = o_id + (N + S);                               // [32] counters (+ the path's ranks with the entangle check on)
  unsigned short* r_id = (unsigned short*)(s_i + 32);      // [kFeCap] ids of the voxel winners, dense
  unsigned char* s_state = (unsigned char*)(r_id + kFeCap);   // [kFeCap] 0 dead, 1 alive, 2 lost its voxel
  signed char* p_parent = (signed char*)(s_state + kFeCap);   // [NEP_MAX_POL + 1][64]
  signed char* p_comb = p_parent + (NEP_MAX_POL + 1) * MB;

  const nep_fe_start* st = starts + slot;
  const double gx = st->goal[0], gy = st->goal[1];
  const double bx = ps.pb[2 * own], by = ps.pb[2 * own + 1];
  EntCtx ec;
  nep_fe_ent_state* my_work = nullptr;
  unsigned char* ent_lists = nullptr;
  unsigned char* b_valid = (unsigned char*)(p_comb + (NEP_MAX_POL + 1) * MB);   // [64] may a plan end at this rank
  auto ent_node = [&](int d, int r) -> nep_fe_ent_state* { return ea.nodes + (((long)slot * (D + 1) + d) * W + r); };
  // ENT: per parent of the depth at hand, who can matter to ANY of its children (bit sets over the agents / statics; filled next
  // to the shortlist from the parent's children box, see there)
  const int MW = (N + 31) >> 5, SW = (S + 31) >> 5;
  unsigned* m_ent = (unsigned*)(b_valid + MB + ((4 - ((3 * kFeCap) & 3)) & 3));      // [64][MW] agents whose tether a child's step may cross (aligned: every array before r_id is a multiple of four bytes)
  unsigned* m_base = m_ent + MB * MW;               // [64][MW] agents whose base square a child may be near
  unsigned* m_stat = m_base + MB * MW;              // [64][SW] static representatives a child's step may cross
  unsigned char* f_bits = (unsigned char*)(m_stat + MB * SW);      // [N] ENT: per agent, the sampled steps in which its moving tether segment sweeps over our base (EntCtx::f_bits)
  if constexpr (ENT) {
    ec.N = N; ec.S = S; ec.own = own; ec.num_pol = D; ec.ns = ea.ns; ec.T_span = sp.T_span; ec.cable = fc.cable_length;
    ec.pb = ps.pb; ec.srep = ea.srep + (long)scene * sp.static_stride * 4; ec.slong = ea.slong + (long)scene * sp.static_stride * 2;
    ec.sampled = ea.sampled; ec.present = ea.present;
    ec.ps = &ps; ec.scene = scene; ec.n_hull = sp.n_hull;
    ec.packed = ea.packed; ec.pk_stride = ea.pk_stride;
    ec.f_bits = (ea.packed && ea.ns <= 8) ? f_bits : nullptr;
#ifdef NEP_PROFILE_PHASES
    ec.prof = ps.dbg ? ps.dbg + (long)slot * 32 + 16 : nullptr;
#endif
    // the crossing lists of the children being merged (phase two of the propagation pass) live where the shortlist's boxes and
    // vertices and the winners' f values are: dead between a depth's GJK pass and its compaction / the next depth's shortlist
    ent_lists = (unsigned char*)o_aabb;      // [n_merge][kEntLdsBytes]
    my_work = NEP_FE_ENT_WGS == 1 ? (nep_fe_ent_state*)(((size_t)(f_bits + N) + 3 + 8 * MW + 7) & ~(size_t)7) + tid : ea.work + ((long)slot * 256 + tid);      // (one working record per thread, in LDS: the list surgery is a chain of dependent loads)
    if (tid == 0) {
      nep_fe_ent_state* root = ent_node(0, 0);
      if (ea.init) {
        ent_copy(root, ea.init + slot);
        // an agent crossing's beta is 0.0 (calculateBetaForCase, entangle_utils.cpp:1713-1719): a state that says otherwise was not
        // made by the reference's rules — flagged (nep_batch_check), since the propagation does not look at such betas
        for (int i = 0; i < root->n_alpha && i < NEP_FE_ENT_CAP; i++) if (root->id[i] <= N && root->beta[i] != 0.0 && ps.flags) atomicOr(ps.flags, NEP_FLAG_ENT_BETA);
      } else { long* z = (long*)root; for (int i = 0; i < (int)(sizeof(nep_fe_ent_state) / 8); i++) z[i] = 0; }
    }
    if (tid < MB) b_valid[tid] = 1;
  }
  int my_entangled = 0, my_overflow = 0, my_big = 0;
#ifdef NEP_PROFILE_PHASES
  if (ps.dbg && tid < 32) ps.dbg[(long)slot * 32 + tid] = 0;
#endif
  if (tid < 32) s_i[tid] = 0;     // [0] shortlist n, [1] winners n, [2] GJK work list n, [4] children, [5] feasible, [6] collision free, [7] goal occupied
  for (int k = tid; k < kFeVis; k += 256) v_key[k] = kFeEmpty;
  if (tid < NEP_FE_MAX_SAMPLES) fe_lattice_fill(sp, fc, s_lat, tid);
  const FeLattice lat{s_lat, s_lat + NEP_FE_MAX_SAMPLES, s_lat + 2 * NEP_FE_MAX_SAMPLES, s_lat + 3 * NEP_FE_MAX_SAMPLES};
  if (tid < 6) {                  // the root is the one-node beam of depth 0
    const double v = tid == 0 ? st->pos[0] : tid == 1 ? st->pos[1] : tid == 2 ? st->vel[0] : tid == 3 ? st->vel[1] : tid == 4 ? st->accel[0] : st->accel[1];
    b_end[tid] = v;               // parity 0
    if (tid == 0) b_g[0] = 0.0;
  }
  __syncthreads();
  // goal_occupied_ (setUp :210-226)
  {
    Pts4 G; const double r = 0.5;
    G.x[0] = gx + r; G.y[0] = gy + r; G.x[1] = gx + r; G.y[1] = gy - r; G.x[2] = gx - r; G.y[2] = gy + r; G.x[3] = gx - r; G.y[3] = gy - r;
    for (int j = tid; j < N; j += 256) {
      if (j == own) continue;
      const HullRef hr = hull_ref(ps, sp.n_hull, scene, j);
      const long h = hr.e * sp.num_pol + (D - 1);
      const int nv = blk(ps.hull_nv, hr.boff)[h];
      if (nv > 0 && gjk_collision(nv, blk(ps.hull_xy, hr.boff) + h * kHullV * 2, G)) s_i[7] = 1;
    }
  }
  int status = NEP_FE_NO_SOLUTION, best_depth = 0, best_rank = -1, nb_prev = 1, depth;
  int my_children = 0, my_feasible = 0, my_free = 0;
#ifdef NEP_PROFILE_PHASES
  long long tph[8] = {0, 0, 0, 0, 0, 0, 0, 0}; long long tlast = clock64();
  long long tent[4] = {0, 0, 0, 0};      // ENT, this thread: base squares, state copy, propagation, mask fill
#define FE_ENT_T0() const long long te0_ = clock64()
#define FE_ENT_T(k) tent[k] += clock64() - te0_
#define FE_TICK(k) do { const long long t_ = clock64(); tph[k] += t_ - tlast; tlast = t_; } while (0)
#else
#define FE_TICK(k) do { } while (0)
#define FE_ENT_T0() do { } while (0)
#define FE_ENT_T(k) do { } while (0)
#endif
  const double tau = sp.T_span;
  // box of a node's children: the control points are monotone in the jerk, so the four corner children bound them all (a
  // superset of every child's own box, feasible or not).  Made by whoever installs the node (the root: thread 0, here), so that a
  // depth starts with the shortlist at once: no serial phase on beam_width threads and no barrier in front of it.
  auto children_box = [&](const double* pe, int r) {
    double lo[2], hi[2];
#pragma unroll
    for (int ax = 0; ax < 2; ax++) {
      double l = __builtin_huge_val(), h = -__builtin_huge_val();
#pragma unroll
      for (int e = 0; e < 2; e++) {
        const double P[4] = {lat.j6[e ? ns - 1 : 0], pe[4 + ax] / 2, pe[2 + ax], pe[ax]};
        double Q[4];
        fe_pos_cps(P, tau, Q);
#pragma unroll
        for (int k = 0; k < 4; k++) { l = fmin(l, Q[k]); h = fmax(h, Q[k]); }      // (v_min / v_max_f64: no NaNs here, and a box's zero may have either sign)
      }
      lo[ax] = l; hi[ax] = h;
    }
    p_box[4 * r] = lo[0]; p_box[4 * r + 1] = hi[0]; p_box[4 * r + 2] = lo[1]; p_box[4 * r + 3] = hi[1];
  };
  if (tid == 0) children_box(b_end, 0);
  for (int k = tid; k < kFeDd; k += 256) d_slot[k] = -1;
  if constexpr (ENT) { for (int k = tid; k < MB * (2 * MW + SW); k += 256) m_ent[k] = 0u; }
  __syncthreads();
  for (depth = 1; depth <= D; depth++) {
    const int cur = depth & 1, prv = cur ^ 1;
    const int idx = (depth > D ? D : depth) - 1;
    FE_TICK(0);
    FE_TICK(1);
    // ---- shortlist: obstacles of this interval whose box meets some parent's box.  G lanes share an obstacle's pass over the
    //      parents (G = 2 at 64 agents + 20 statics: this loop over up to 64 parents on a third of the threads was a quarter of a
    //      search), each parent's box is one 32-byte LDS read, the four comparisons are combined without branches, and the
    //      lanes of a group exchange their verdicts inside the wave ----
    {
      int G = 1; while (2 * G * (N + S) <= 256 && G < 8) G *= 2;
      const int part = tid & (G - 1);
      for (int j0 = 0; j0 < N + S; j0 += 256 / G) {
        const int j = j0 + tid / G;
        const bool mine = j < N + S && j != own;
        double x0 = 0, x1 = 0, y0 = 0, y1 = 0;
        if (mine) { const double* bxj = ps.fe_box + (((long)scene * (N + S) + j) * sp.num_pol + idx) * 4; x0 = bxj[0]; x1 = bxj[1]; y0 = bxj[2]; y1 = bxj[3]; }      // (fe_box_kernel)
        bool near = false;
        for (int q = part; q < nb_prev; q += G) {
          const double2 pa = ((const double2*)p_box)[2 * q], pb = ((const double2*)p_box)[2 * q + 1];
          near |= (x1 >= pa.x) & (pa.y >= x0) & (y1 >= pb.x) & (pb.y >= y0);
        }
        int nr = (mine && near) ? 1 : 0;
        for (int o = 1; o < G; o <<= 1) nr |= __shfl_xor(nr, o);
        if (nr && part == 0 && mine) {      // (rare: now its vertices are worth fetching)
          int nv; const double* V;
          if (j < N) { const HullRef hr = hull_ref(ps, sp.n_hull, scene, j); const long h = hr.e * sp.num_pol + idx; nv = blk(ps.hull_nv, hr.boff)[h]; V = blk(ps.hull_xy, hr.boff) + h * kHullV * 2; }
          else { const long js = (long)scene * sp.static_stride + (j - N); nv = ps.static_nv[js]; V = ps.static_xy + js * kHullV * 2; }
          double2 vv[kHullV];
#pragma unroll
          for (int i = 0; i < kHullV; i++) vv[i] = ((const double2*)V)[i];
          const int o = atomicAdd(&s_i[0], 1); o_aabb[4 * o] = x0; o_aabb[4 * o + 1] = x1; o_aabb[4 * o + 2] = y0; o_aabb[4 * o + 3] = y1; o_nv[o] = nv; o_id[o] = j;
          if (o < kFeObsLds) {
#pragma unroll
            for (int i = 0; i < kHullV; i++) ((double2*)(o_V + o * kHullV * 2))[i] = vv[i];
          }
        }
      }
    }
    if constexpr (ENT) {
      // ---- who can matter to a parent's children (ent_agent_may_cross, ent_device.h): every sample of every child lies in the
      //      parent's children box (control-point box of the corner children).  Base squares: an agent farther than the cull
      //      distance from the whole box is farther from any first control point.  A clear bit is a proof; a set bit only means
      //      "run the reference's test".  255 agents x 3 steps per child -> a few dozen. ----
      FE_ENT_T0();
      const double kPad = 1e-9;
      const double safe_dist = (sp.T_span * sp.v_max) * 2;
      // (an obstacle per thread, the parents in the inner loop: its packed record is fetched once and then read from the L1 for
      // every parent, instead of once per (parent, obstacle) pair from wherever it was)
      for (int j = tid; j < N + S; j += 256) {
        if (j == own) continue;
        const double pbx = j < N ? ps.pb[2 * j] : 0.0, pby = j < N ? ps.pb[2 * j + 1] : 0.0;
        // the part of the proof that does not involve the box (the agent's moving tether segment sweeping over OUR base, per step):
        // once per obstacle — and kept for the children's crossing tests, which would evaluate it once per child and step
        unsigned fb_j = 0u;
        if (j < N) { fb_j = ec.ns <= 8 ? ent_agent_fbits_pk(ec, j, idx) : 0xffu; f_bits[j] = (unsigned char)fb_j; }
        // (three samples per interval, a tether without bend points besides its base: what the box test reads of the record — the
        // base and the four samples — is fetched once here, not once per parent)
        // (up to four bend points — the bench's config-5 tethers have two to four: every value of the record the test reads is in
        // registers before the loop over the parents; read inside it, one bend point at a time, the proof was a chain of round trips
        // per (obstacle, parent) and 16 % of a search)
        int nreg = 0; bool absent = false; Ev2 bq0{0, 0}, bq1{0, 0}, bq2{0, 0}, bq3{0, 0}, sm0{0, 0}, sm1{0, 0}, sm2{0, 0}, sm3{0, 0};
        if (j < N && !fb_j && ec.ns == 3) {
          const double* r = ent_rec(ec, j, idx);
          const int2 hd = *(const int2*)r;
          absent = !hd.x;                                   // (no trajectory: nothing to cross — its base square stays)
          if (hd.x && hd.y >= 1 && hd.y <= 4) {
            nreg = hd.y;
            const double2 b = *(const double2*)(r + kEntPkBend), b1 = *(const double2*)(r + kEntPkBend + 2), b2 = *(const double2*)(r + kEntPkBend + 4), b3 = *(const double2*)(r + kEntPkBend + 6);
            const double2 a0 = *(const double2*)(r + kEntPkHead), a1 = *(const double2*)(r + kEntPkHead + 2), a2 = *(const double2*)(r + kEntPkHead + 4), a3 = *(const double2*)(r + kEntPkHead + 6);
            bq0 = Ev2{b.x, b.y}; bq1 = Ev2{b1.x, b1.y}; bq2 = Ev2{b2.x, b2.y}; bq3 = Ev2{b3.x, b3.y};
            sm0 = Ev2{a0.x, a0.y}; sm1 = Ev2{a1.x, a1.y}; sm2 = Ev2{a2.x, a2.y}; sm3 = Ev2{a3.x, a3.y};
          }
        }
        const Ev2 bk = nreg <= 1 ? bq0 : nreg == 2 ? bq1 : nreg == 3 ? bq2 : bq3;      // the last bend point: where the moving segment starts
        for (int q = 0; q < nb_prev; q++) {
          const EntBox bx{p_box[4 * q] - kPad, p_box[4 * q + 1] + kPad, p_box[4 * q + 2] - kPad, p_box[4 * q + 3] + kPad};
          if (j < N) {
            {   // collidesWithBases2d's distance cull, from the box
              const double dx = fmax(fmax(bx.x0 - pbx, pbx - bx.x1), 0.0), dy = fmax(fmax(bx.y0 - pby, pby - bx.y1), 0.0);
              if (sqrt(dx * dx + dy * dy) <= safe_dist + 1e-6) atomicOr(&m_base[q * MW + (j >> 5)], 1u << (j & 31));
            }
            bool may;
            if (fb_j) may = true;
            else if (absent) may = false;
            else if (nreg) {      // ent_agent_may_cross_pk for nb <= 4, from registers
              const int s0 = ent_side(bx, sm0, bk);
              may = (s0 == 0) | (ent_side(bx, sm1, bk) != s0) | (ent_side(bx, sm2, bk) != s0) | (ent_side(bx, sm3, bk) != s0);
              if (nreg > 1) may |= ent_side(bx, bq1, bq0) == 0;
              if (nreg > 2) may |= ent_side(bx, bq2, bq1) == 0;
              if (nreg > 3) may |= ent_side(bx, bq3, bq2) == 0;
            } else may = ent_agent_may_cross_pk(ec, bx, j, idx);
            if (may) atomicOr(&m_ent[q * MW + (j >> 5)], 1u << (j & 31));
          } else {
            const int sj = j - N;
            if (ent_static_may_cross(ec, bx, sj)) atomicOr(&m_stat[q * SW + (sj >> 5)], 1u << (sj & 31));
          }
        }
      }
      FE_ENT_T(3);
    }
    __syncthreads();
    FE_TICK(2);
    const int n_obs = s_i[0];
    // ---- children: one per thread.  Pass 1 settles everything but the GJK tests: a child whose box meets
    //      shortlisted obstacles goes on a work list (its obstacle mask parked in s_vox), so that pass 2
    //      spreads the expensive tests over all threads instead of leaving them with the lanes that
    //      happened to draw crowded children ----
    const int n_c = nb_prev * NC;
    if (tid == 0) s_i[1] = 0;                              // (the winners' counter: last read in the previous depth's rank phase)
    // ENT: settling a collision-free child is split in two.  Part one (here, by whoever examined the child) is the base-square test;
    // a survivor is marked and, after the barrier, listed (p_list, in r_id's storage: free until the winners are compacted).  Part two — copy
    // of the parent's entangle state, entanglesWithOtherAgents, the voxel — runs after a barrier over that DENSE list, one
    // survivor per thread: the propagation is two orders of magnitude dearer than anything else a child costs and only a fifth of
    // the children reach it, so examined in place the threads that drew two or three survivors kept the others waiting (pass 1 was
    // 65 % of a config-5 search, its critical path three propagations per depth instead of one).
    unsigned short* p_list = r_id;      // (the GJK work list has been consumed when the survivors are listed; the winners are compacted after the propagation)
    const int n_lists_a = (int)((sizeof(double) * (4 * (size_t)(N + S) + kFeObsLds * kHullV * 2 + (rf_alias ? 0 : kFeCap))) / kEntLdsBytes);      // crossing lists the borrowed storage holds (o_aabb, o_V, r_f when it has storage of its own)
    const int n_lists_b = (int)((sizeof(int) * (size_t)kFeDd) / kEntLdsBytes), n_lists_c = (int)((sizeof(double) * (size_t)kFeCap) / kEntLdsBytes);      // ... the voxel table; s_f, and s_vox as many again
    auto settle_voxel = [&](int id, FeChild& ch, unsigned iz) {
      const long long vox = ENT ? (long long)(((unsigned long long)(unsigned short)ch.vx << 48) | ((unsigned long long)(unsigned short)ch.vy << 32) | iz)
                                : (((long long)ch.vx << 32) | (unsigned int)ch.vy);
      bool seen = false;
      for (unsigned h = fe_hash(vox) & (kFeVis - 1);; h = (h + 1) & (kFeVis - 1)) { const unsigned long long k = v_key[h]; if (k == (unsigned long long)vox) { seen = true; break; } if (k == kFeEmpty) break; }
      if constexpr (ENT) { if (!seen) { ea.st_f[(long)slot * kFeCap + id] = ch.f; ea.st_vox[(long)slot * kFeCap + id] = vox; } }      // (parked in global memory: s_f and s_vox are lent to the crossing lists during the propagation pass and filled after it)
      else { if (!seen) { s_f[id] = ch.f; s_vox[id] = vox; } }
      s_state[id] = seen ? 0 : 1;
    };
    auto settle = [&](int id, FeChild& ch) {       // collision free: closed voxel?  else alive
      if constexpr (ENT) {
        {   // collidesWithBases2d: the other agents' 0.7 m base squares within 2 T v_max of the first control point
          FE_ENT_T0();
          const double radius = 0.7, safe_dist = (sp.T_span * sp.v_max) * 2;
          Pts4 Bq;
#pragma unroll
          for (int i = 0; i < 4; i++) { Bq.x[i] = ch.Qx[i]; Bq.y[i] = ch.Qy[i]; }
          const unsigned* mb = m_base + (id / NC) * MW;
          for (int j = 0; j < N; j++) {
            if (!((mb[j >> 5] >> (j & 31)) & 1u)) { j |= 31 * !mb[j >> 5]; continue; }      // (farther than the cull distance from every child of this parent)
            if (j == own) continue;
            const double pbx = ps.pb[2 * j], pby = ps.pb[2 * j + 1];
            const double d1 = sqrt((ch.Qx[0] - pbx) * (ch.Qx[0] - pbx) + (ch.Qy[0] - pby) * (ch.Qy[0] - pby));
            if (d1 > safe_dist) continue;
            const double sq[8] = {pbx + radius, pby + radius, pbx + radius, pby - radius, pbx - radius, pby - radius, pbx - radius, pby + radius};
            if (gjk_collision(4, sq, Bq)) { s_state[id] = 0; return; }
          }
          FE_ENT_T(0);
        }
        my_free++;
        s_state[id] = 4;                                        // (awaiting its propagation: listed in id order after the barrier)
      } else {
        my_free++;
        settle_voxel(id, ch, 0u);
      }
    };
    // (fast instantiation: the lists of new crossings of the round's children were made by cross_round, below — hdr[rank * ns + j - 1] in
    // s_f's storage, the words in s_vox's — and this pass is the list surgery alone, ent_propagate_pre)
    unsigned* x_hdr = (unsigned*)s_f; unsigned* x_pool = (unsigned*)s_vox;
    const int x_pool_cap = (int)(sizeof(long long) * (size_t)kFeCap / sizeof(unsigned)), x_hdr_cap = (int)(sizeof(double) * (size_t)kFeCap / sizeof(unsigned));
    auto propagate = [&](int id, int rank_in_round) {      // ENT, part two (see above)
      const int pr_ = id / NC, cc = id % NC;
      FeChild ch;
      fe_child_again<true>(sp, fc, lat, b_end + (prv * MB + pr_) * 6, b_g[prv * MB + pr_], cc / ns, cc % ns, gx, gy, ch);
      EntLds L;
      {
        typedef __attribute__((address_space(3))) unsigned char* lds_bytes;
        // (threads beyond the borrowed storage's n_lists_a lists keep theirs in the depth's voxel table, which is dead — all -1 — until the
        // voxel pass after the propagation and is cleared again before it: 163 instead of 132 survivors per round at config 5, and a
        // depth's ~140 survivors are one round instead of two)
        // (and then in s_f and s_vox, whose contents — f and voxel of the children that survive this pass — are parked in global memory
        // until the pass is over: 259 lists at config 5, every depth one round)
        // (fast instantiation: s_f and s_vox hold the round's headers and pool instead — n_merge stops at n_lists_a + n_lists_b there)
        const lds_bytes base = (lds_bytes)(unsigned)(size_t)(tid < n_lists_a ? ent_lists + tid * kEntLdsBytes
                                                            : tid < n_lists_a + n_lists_b ? (unsigned char*)d_slot + (tid - n_lists_a) * kEntLdsBytes
                                                            : tid < n_lists_a + n_lists_b + n_lists_c ? (unsigned char*)s_f + (tid - n_lists_a - n_lists_b) * kEntLdsBytes
                                                            : (unsigned char*)s_vox + (tid - n_lists_a - n_lists_b - n_lists_c) * kEntLdsBytes);      // (the low 32 bits of a generic LDS address are the LDS offset)
        L.id = (ent_lds_short)base; L.cs = (ent_lds_char)(base + 2 * NEP_FE_ENT_CAP); L.bend = (ent_lds_char)(base + 3 * NEP_FE_ENT_CAP); L.beta = my_work->beta;
        L.cap = ea.fast_cap; L.bend_cap = ea.fast_bend;
        if constexpr (BIG) {
          if (ea.big_lds_off) {      // lists of up to kEntBigLdsCap entries in LDS of their own (one workgroup per CU: there is room), their betas in global memory
            const lds_bytes bb = (lds_bytes)(unsigned)(size_t)((unsigned char*)fe_smem + ea.big_lds_off + tid * kEntBigLdsBytes);
            L.id = (ent_lds_short)bb; L.cs = (ent_lds_char)(bb + 2 * kEntBigLdsCap); L.bend = (ent_lds_char)(bb + 3 * kEntBigLdsCap);
            L.beta = ea.big_beta + ((long)blockIdx.x * 256 + tid) * kEntBigLdsCap;
            L.cap = ea.big.cap < kEntBigLdsCap ? ea.big.cap : kEntBigLdsCap; L.bend_cap = kEntBigLdsBend;
          }
        }
      }
      double arc = 0.0;
      int rc = 2;
      ec.m_agent = m_ent + pr_ * MW; ec.m_static = m_stat + pr_ * SW;
      // the fixed record's path — unless the parent is a big record already (n_alpha < 0, ent_device.h) or holds more than this
      // handle's fast path takes
      bool loaded;
      {
        FE_ENT_T0();
        const nep_fe_ent_state* par = ent_node(depth - 1, depth == 1 ? 0 : pr_);
        if constexpr (BIG) { const int pn = par->n_alpha; loaded = pn < 0 ? (ea.big_lds_off != 0 && ent_lds_load_big(L, ent_big_view(ea.big, -pn - 1), N)) : ent_lds_load(L, par, N); }
        else loaded = ent_lds_load(L, par, N);
        FE_ENT_T(1);
      }
      if (loaded) {
        if constexpr (BIG) {
          unsigned add_tail[kEntAddCap - EntAdd::reg];
          { FE_ENT_T0(); rc = ent_propagate<EntAdd>(ec, &L, EntAdd::Store{add_tail, ea.fast_add}, ch.cx, ch.cy, Ev2{ch.e[0], ch.e[1]}, depth, arc, true, 1); FE_ENT_T(2); }
        } else {
          FE_ENT_T0(); rc = ent_propagate_pre(ec, &L, x_hdr + rank_in_round * ec.ns, x_pool, ea.xpool ? ea.xpool + (long)slot * ea.xpool_stride + rank_in_round * ec.ns * kEntAddCap : nullptr, ea.fast_add, ch.cx, ch.cy, Ev2{ch.e[0], ch.e[1]}, arc, true, 1); FE_ENT_T(2);
        }
      }
      // a capacity of the fixed record (its list, a step's new crossings, the bend points) or a big parent: the child is left for the
      // big-record pass below (rare: the flag keeps that pass out of every other depth's way)
      if (__builtin_expect(rc >= 2, 0)) {
        if constexpr (BIG) { s_state[id] = 5; s_i[11] = 1; }
        else { my_entangled++; my_overflow |= 1 << (rc - 2); s_state[id] = 0; }      // (pruned for now: the search is listed for the big-record instantiation)
        return;
      }
      if (rc) { my_entangled++; s_state[id] = 0; return; }
      if (BIG && (L.n_alpha > ea.fast_cap || L.n_alpha > NEP_FE_ENT_CAP || L.n_bend > ea.fast_bend || L.n_bend > NEP_MAX_BEND)) {
        // more than the fixed record holds (the LDS lists of this instantiation do): into a big record
        const int k = ea.big.base ? atomicAdd(ea.big.count, 1) : ea.big.n_rec;
        my_big++;
        if (k >= ea.big.n_rec) { my_entangled++; my_overflow |= 8; s_state[id] = 0; return; }
        ent_lds_store_big(ea.big, k, L, N);
        nep_fe_ent_state* sv = ea.saved + ((long)slot * kFeCap + id);
        sv->n_alpha = -(k + 1); sv->n_bend = 0;
      } else
      ent_lds_store(ea.saved + ((long)slot * kFeCap + id), L, N);
      ea.saved_arc[(long)slot * kFeCap + id] = arc;      // (for the install, should this child win its voxel and a rank)
      ch.g = b_g[prv * MB + pr_] + arc;
      ch.f = ch.g + fc.bias * ((ch.dist + 0.3 * (double)L.n_alpha) + 1.0 * (double)L.n_bend);
      settle_voxel(id, ch, ent_iz(&L));
    };
    auto propagate_big = [&](int id) {      // ENT: the same child in a big record, which has the reference's own bound and no other (ent_device.h)
      const int pr_ = id / NC, cc = id % NC;
      FeChild ch;
      fe_child_again<true>(sp, fc, lat, b_end + (prv * MB + pr_) * 6, b_g[prv * MB + pr_], cc / ns, cc % ns, gx, gy, ch);
      ec.m_agent = m_ent + pr_ * MW; ec.m_static = m_stat + pr_ * SW;
      const EntBigOut bo = ent_big_child(ec, ea.big, ent_node(depth - 1, depth == 1 ? 0 : pr_), ch.cx, ch.cy, Ev2{ch.e[0], ch.e[1]}, depth, true, 1);
      my_big++;
      if (bo.rc) { my_entangled++; if (bo.rc >= 2) my_overflow |= 8; s_state[id] = 0; return; }
      nep_fe_ent_state* sv = ea.saved + ((long)slot * kFeCap + id);
      sv->n_alpha = -(bo.k + 1); sv->n_bend = 0; ea.saved_arc[(long)slot * kFeCap + id] = bo.arc;
      ch.g = b_g[prv * MB + pr_] + bo.arc;
      ch.f = ch.g + fc.bias * ((ch.dist + 0.3 * (double)bo.n_alpha) + 1.0 * (double)bo.n_bend);
      settle_voxel(id, ch, bo.iz);
    };
    auto obstacle_V = [&](int o) -> const double* {
      if (o < kFeObsLds) return o_V + o * kHullV * 2;
      const int j = o_id[o];
      if (j < N) { const HullRef hr = hull_ref(ps, sp.n_hull, scene, j); return blk(ps.hull_xy, hr.boff) + (hr.e * sp.num_pol + idx) * kHullV * 2; }
      return ps.static_xy + ((long)scene * sp.static_stride + (j - N)) * kHullV * 2;
    };
    for (int id = tid; id < n_c; id += 256) {
      const int pr = id / NC, cc = id % NC;
      const double* pe = b_end + (prv * MB + pr) * 6;
      const double pg = b_g[prv * MB + pr];
      FeChild ch;
      my_children++;
      s_state[id] = 0;
      if (!fe_child(sp, fc, lat, pe, pg, depth == 1, cc / ns, cc % ns, gx, gy, bx, by, ch)) continue;
      my_feasible++;
      double qx0 = ch.Qx[0], qx1 = ch.Qx[0], qy0 = ch.Qy[0], qy1 = ch.Qy[0];
#pragma unroll
      for (int i = 1; i < 4; i++) { qx0 = fmin(qx0, ch.Qx[i]); qx1 = fmax(qx1, ch.Qx[i]); qy0 = fmin(qy0, ch.Qy[i]); qy1 = fmax(qy1, ch.Qy[i]); }
      unsigned long long cand_mask = 0;
      bool hit = false;
      for (int o = 0; o < n_obs; o++) {
        const bool ov = !(o_aabb[4 * o + 1] < qx0 || qx1 < o_aabb[4 * o] || o_aabb[4 * o + 3] < qy0 || qy1 < o_aabb[4 * o + 2]);
        if (!ov) continue;
        if (o < 64) cand_mask |= 1ull << o;
        else if (!hit) {                                  // (shortlists beyond 64: tested on the spot)
          Pts4 B;
#pragma unroll
          for (int i = 0; i < 4; i++) { B.x[i] = ch.Qx[i]; B.y[i] = ch.Qy[i]; }
          hit = gjk_collision(o_nv[o], obstacle_V(o), B);
        }
      }
      if (hit) continue;
      if (cand_mask == 0) { settle(id, ch); continue; }
      s_vox[id] = (long long)cand_mask; s_state[id] = 3;
      r_id[atomicAdd(&s_i[2], 1)] = (unsigned short)id;    // (r_id is free until the winners are compacted)
    }
    FE_TICK(7);
    __syncthreads();
    const int n_work = s_i[2];
#ifdef NEP_PROFILE_PHASES
    if (tid == 0 && ps.dbg) { ps.dbg[(long)slot * 32 + 9] += n_work; ps.dbg[(long)slot * 32 + 10] += n_obs; ps.dbg[(long)slot * 32 + 11] += n_c; }
#endif
    for (int w = tid; w < n_work; w += 256) {
      const int id = r_id[w];
      const int pr = id / NC, cc = id % NC;
      unsigned long long cand_mask = (unsigned long long)s_vox[id];
      FeChild ch;
      fe_child_again<true>(sp, fc, lat, b_end + (prv * MB + pr) * 6, b_g[prv * MB + pr], cc / ns, cc % ns, gx, gy, ch);
      Pts4 B;
#pragma unroll
      for (int i = 0; i < 4; i++) { B.x[i] = ch.Qx[i]; B.y[i] = ch.Qy[i]; }
      bool hit = false;
      while (cand_mask && !hit) {
        const int o = __ffsll((long long)cand_mask) - 1; cand_mask &= cand_mask - 1;
        if (o < kFeObsLds) { typedef __attribute__((address_space(3))) const double* lcd; hit = gjk_collision(o_nv[o], (lcd)(unsigned)(unsigned long long)(o_V + o * kHullV * 2), B); }
        else hit = gjk_collision(o_nv[o], obstacle_V(o), B);
      }
      if (hit) s_state[id] = 0; else settle(id, ch);
    }
    __syncthreads();
    FE_TICK(3);
    if constexpr (ENT) {
      // the survivors in id order (a block-wide prefix sum over the marks): neighbouring threads then propagate children of the SAME
      // parent — the same starting list, the same candidate agents, nearly the same crossings — so a wave's lanes follow one path
      // through the list surgery instead of 64, and their record loads fall on the same lines
      {
        const int per = (n_c + 255) >> 8;
        int cnt = 0;
        for (int k = 0; k < per; k++) { const int id = tid * per + k; cnt += (id < n_c && s_state[id] == 4) ? 1 : 0; }
        int incl = cnt;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const int y = __shfl_up(incl, o); if ((tid & 63) >= o) incl += y; }
        if ((tid & 63) == 63) s_i[20 + (tid >> 6)] = incl;
        __syncthreads();
        int off = incl - cnt;
        for (int w = 0; w < (tid >> 6); w++) off += s_i[20 + w];
        for (int k = 0; k < per; k++) { const int id = tid * per + k; if (id < n_c && s_state[id] == 4) p_list[off++] = (unsigned short)id; }
        if (tid == 255) s_i[3] = off;
        __syncthreads();
      }
      const int n_prop = s_i[3];
      int n_merge = BIG ? n_lists_a + n_lists_b + 2 * n_lists_c : n_lists_a + n_lists_b;      // threads whose lists fit the borrowed LDS (o_aabb, o_V and, when it has storage of its own, r_f; the voxel table; big-record instantiation: s_f and s_vox as well)
      if (n_merge > 256) n_merge = 256;
      if constexpr (BIG) { if (ea.big_lds_off) n_merge = 256; }      // (lists of their own)
      if constexpr (!BIG) { if (n_merge * ec.ns > x_hdr_cap) n_merge = x_hdr_cap / ec.ns; }
      { const int rounds = (n_prop + n_merge - 1) / n_merge; if (rounds > 1) n_merge = (n_prop + rounds - 1) / rounds; }      // (even rounds: 178 survivors are 89 + 89, not 131 + 47)
      // ---- cross_round (fast instantiation): the new crossings of every (child, sampled step) of a round, found in a dense pass of its
      //      own — see ent_cross_step, ent_device.h.  The packed records of the depth's candidate agents (the union of the parents'
      //      masks) are staged in the storage the crossing lists use AFTER this pass (o_aabb, o_V); one thread per (child, step) walks
      //      its parent's candidates in index order out of LDS and leaves the step's list in the pool. ----
      const int x_stage_cap = (ea.packed && ea.pk_stride > 0) ? (int)((sizeof(double) * (4 * (size_t)(N + S) + kFeObsLds * kHullV * 2)) / (sizeof(double) * (size_t)ea.pk_stride)) : 0;
      unsigned* x_U = (unsigned*)(((size_t)(f_bits + N) + 3) & ~(size_t)3);      // [MW] union of the parents' agent masks; [MW] popcounts before each word (storage of their own behind f_bits)
      unsigned short* x_ids = (unsigned short*)o_id;           // [n_stage] the staged agents, ascending (the shortlist's ids are dead by now)
      double* x_rec = o_aabb;                                  // [n_stage][pk_stride]
      auto cross_round = [&](int b0, int nr, bool first) {
        if (first) {      // (once per depth: who is staged)
          if (tid < MW) { unsigned u = 0u; for (int q = 0; q < nb_prev; q++) u |= m_ent[q * MW + tid]; x_U[tid] = u; }
          __syncthreads();
          if (tid == 0) {
            int acc = 0; for (int w = 0; w < MW; w++) { x_U[MW + w] = (unsigned)acc; acc += __popc(x_U[w]); }
            const int lim = 2 * (N + S);                           // (ids that fit o_id's storage)
            int n_st = acc < x_stage_cap ? acc : x_stage_cap; if (n_st > lim) n_st = lim;
            s_i[13] = n_st;
          }
          __syncthreads();
          const int n_st = s_i[13];
          if (tid < MW) { unsigned u = x_U[tid]; int k = (int)x_U[MW + tid]; while (u && k < n_st) { const int b = __ffs(u) - 1; u &= u - 1u; x_ids[k++] = (unsigned short)((tid << 5) + b); } }
          __syncthreads();
        }
        const int n_st = s_i[13];
        const int itv0 = depth > D ? D - 1 : depth - 1;
        // (every round: the previous round's crossing lists were written over the records)
        for (int e = tid; e < n_st * ea.pk_stride; e += 256) { const int sl = e / ea.pk_stride, d_ = e - sl * ea.pk_stride; x_rec[e] = ent_rec(ec, x_ids[sl], itv0)[d_]; }
        if (tid == 0) s_i[12] = 0;                                 // the pool's fill
        __syncthreads();
#ifdef NEP_PROFILE_PHASES
        const long long t_staged_ = clock64();
#endif
        typedef __attribute__((address_space(3))) const unsigned* x_lcu;
        const x_lcu x_U_l = (x_lcu)(unsigned)(unsigned long long)x_U;      // (x_U's address went through an integer: the compiler no longer knows it is LDS)
        const unsigned x_rec_lds = (unsigned)(unsigned long long)x_rec;      // (a generic pointer into LDS: its low half is the LDS address)
        auto rec_of = [&](int i, int itv) -> int {      // LDS byte offset of agent i's staged record, -1: read the packed record where it lies
          const int w = i >> 5;
          const int sl = (int)x_U_l[MW + w] + __popc(x_U_l[w] & ((1u << (i & 31)) - 1u));
          return (sl < n_st && itv == itv0) ? (int)(x_rec_lds + (unsigned)sl * (unsigned)(ea.pk_stride * 8)) : -1;
        };
        for (int q = tid; q < nr * ec.ns; q += 256) {
          const int rk = q / ec.ns, j = q - rk * ec.ns + 1;
          const int id = p_list[b0 + rk];
          const int pr_ = id / NC, cc = id % NC;
          FeChild ch;
          fe_child_again<false>(sp, fc, lat, b_end + (prv * MB + pr_) * 6, b_g[prv * MB + pr_], cc / ns, cc % ns, gx, gy, ch);      // (the polynomial and its end point: no control points here)
          const Ev2 end{ch.e[0], ch.e[1]};
          const Ev2 pk = ent_step_point(ec, ch.cx, ch.cy, end, j - 1), pk1 = ent_step_point(ec, ch.cx, ch.cy, end, j);
          ec.m_agent = m_ent + pr_ * MW; ec.m_static = m_stat + pr_ * SW;
          unsigned add_tail[kEntAddCap - EntAdd::reg];
          EntAdd add; add.attach(EntAdd::Store{add_tail, ea.fast_add}); add.clear(); add.r0 = add.r1 = add.r2 = add.r3 = 0u;
          ent_cross_step(add, ec, pk, pk1, depth, j, rec_of);
          unsigned h = kEntHdrOvf;
          if (!add.overflow) {
            const int off = add.n > 0 ? atomicAdd(&s_i[12], add.n) : 0;
            if (off + add.n <= x_pool_cap) {
              for (int e = 0; e < add.n; e++) x_pool[off + e] = add.get(e);
              h = (unsigned)off | ((unsigned)add.n << 16);
            } else if (ea.xpool && (q + 1) * kEntAddCap <= ea.xpool_stride) {      // the LDS pool is full: this pair's block in global memory
              unsigned* g = ea.xpool + (long)slot * ea.xpool_stride + q * kEntAddCap;
              for (int e = 0; e < add.n; e++) g[e] = add.get(e);
              h = kEntHdrGlobal | ((unsigned)add.n << 16);
            }
          }
          x_hdr[q] = h;
        }
#ifdef NEP_PROFILE_PHASES
        tent[1] += clock64() - t_staged_;      // (profiling builds: this thread's own (child, step) pairs, without the barrier — in the state copy's slot)
#endif
        __syncthreads();
      };
      for (int b0 = 0; b0 < n_prop; b0 += n_merge) {             // that many survivors at a time, one per thread
        if constexpr (!BIG) { FE_ENT_T0(); const int nr = n_prop - b0 < n_merge ? n_prop - b0 : n_merge; cross_round(b0, nr, b0 == 0); FE_ENT_T(0); }      // (profiling builds: counted with the base squares, tent[0])
        if (tid < n_merge && b0 + tid < n_prop) propagate(p_list[b0 + tid], tid);
        __syncthreads();
      }
      for (int k = tid; k < kFeDd; k += 256) d_slot[k] = -1;     // (lent to the lists above)
      __syncthreads();                                           // (nobody reads a list any more)
      if constexpr (BIG) {
        if (s_i[11] != 0) {                                      // children the fixed record could not carry
          for (int w = tid; w < n_prop; w += 256) { const int id = p_list[w]; if (s_state[id] == 5) propagate_big(id); }
          __syncthreads();
        }
      }
      for (int id = tid; id < n_c; id += 256) if (s_state[id] == 1) { s_f[id] = ea.st_f[(long)slot * kFeCap + id]; s_vox[id] = ea.st_vox[(long)slot * kFeCap + id]; }      // (back from where settle_voxel parked them)
      __syncthreads();
      FE_TICK(1);
    }
    // ---- one node per voxel: the best (f, id) claims the voxel's slot; whoever is displaced or beaten is out ----
    for (int id = tid; id < n_c; id += 256) {
      if (s_state[id] != 1) continue;
      const long long vi = s_vox[id]; const double fi = s_f[id];
      unsigned h = fe_hash(vi) & (kFeDd - 1);
      for (;;) {
        int k = atomicCAS(&d_slot[h], -1, id);
        if (k == -1) break;                                           // took an empty slot
        if (s_vox[k] != vi) { h = (h + 1) & (kFeDd - 1); continue; }  // another voxel's slot
        const double fk = s_f[k];
        if (fk < fi || (fk == fi && k < id)) { s_state[id] = 2; break; }   // the holder is better
        if (atomicCAS(&d_slot[h], k, id) == k) { s_state[k] = 2; break; }  // displaced the holder
      }
    }
    __syncthreads();
    FE_TICK(4);
    for (int id = tid; id < n_c; id += 256) if (s_state[id] == 1) { const int o = atomicAdd(&s_i[1], 1); r_f[o] = s_f[id]; r_id[o] = (unsigned short)id; }
    __syncthreads();
    FE_TICK(5);
    // ---- the beam: rank among the voxel winners (dense arrays) ----
    const int n_b = s_i[1];
    const int nb = n_b < W ? n_b : W;
    for (int a = tid; a < n_b; a += 256) {
      const int i = r_id[a];
      const double fi = r_f[a];
      // rank = the winners with a smaller f, ties by the lower id: the ids are only looked at when there is a tie (rare)
      int rank = 0, same = 0;
#pragma unroll 4
      for (int b = 0; b < n_b; b++) { const double fk = r_f[b]; rank += fk < fi ? 1 : 0; same += fk == fi ? 1 : 0; }
      if (same > 1) { for (int b = 0; b < n_b; b++) rank += (r_f[b] == fi && (int)r_id[b] < i) ? 1 : 0; }
      if (rank < W) {   // recompute the child (same arithmetic) and install it
        const int pr = i / NC, cc = i % NC;
        const double* pe = b_end + (prv * MB + pr) * 6;
        const double pg = b_g[prv * MB + pr];
        FeChild ch;
        fe_child_again<false>(sp, fc, lat, pe, pg, cc / ns, cc % ns, gx, gy, ch);
        if constexpr (ENT) {   // the same propagation again, this time into the node's own record
          // the state and the arc length this child arrived with were kept by whoever examined it (a second propagation here, by
          // up to beam_width threads while the others wait, was a quarter of the search)
          const nep_fe_ent_state* nd = ea.saved + ((long)slot * kFeCap + i);
          const double arc = ea.saved_arc[(long)slot * kFeCap + i];
          ch.g = pg + arc;
          int nA = nd->n_alpha, nB = nd->n_bend; bool valid;
          if (BIG && nA < 0) {      // a big record (ent_device.h): the node is its marker
            nep_fe_ent_state* dn = ent_node(depth, rank); dn->n_alpha = nA; dn->n_bend = 0;
            const EntBig B = ent_big_view(ea.big, -nA - 1);
            nA = B.n_alpha; nB = B.n_bend; valid = ent_valid_endpoint(&B, N);
          } else { ent_copy(ent_node(depth, rank), nd); valid = ent_valid_endpoint(nd, N); }
          ch.f = ch.g + fc.bias * ((ch.dist + 0.3 * (double)nA) + 1.0 * (double)nB);
          b_valid[rank] = valid ? 1 : 0;
        }
#pragma unroll
        for (int q = 0; q < 6; q++) b_end[(cur * MB + rank) * 6 + q] = ch.e[q];
        children_box(ch.e, rank);                         // (what the next depth's shortlist tests the obstacles against)
        b_g[cur * MB + rank] = ch.g; b_dist[rank] = ch.dist; b_f[rank] = ch.f;
        p_parent[depth * MB + rank] = (signed char)(depth == 1 ? -1 : pr);
        p_comb[depth * MB + rank] = (signed char)cc;
        const unsigned long long vox = (unsigned long long)s_vox[i];      // close the voxel to later depths
        for (unsigned h = fe_hash((long long)vox) & (kFeVis - 1);; h = (h + 1) & (kFeVis - 1)) { const unsigned long long o = atomicCAS(&v_key[h], kFeEmpty, vox); if (o == kFeEmpty || o == vox) break; }
      }
    }
    // (the next depth's counters and voxel table: nobody reads them any more in this one)
    if (tid == 0) { s_i[0] = 0; s_i[2] = 0; s_i[3] = 0; s_i[11] = 0; }
    for (int k = tid; k < kFeDd; k += 256) d_slot[k] = -1;
    __syncthreads();
    if constexpr (ENT) { for (int k = tid; k < MB * (2 * MW + SW); k += 256) m_ent[k] = 0u; __syncthreads(); }      // (read by the installs above, filled again by the next depth)
    FE_TICK(6);
    if (nb == 0) { status = depth == 1 ? NEP_FE_NO_SOLUTION : NEP_FE_EMPTY; break; }
    nb_prev = nb;
    // (first rank in order that may end a plan / that is inside the goal radius: one LDS read per lane and a ballot in every wave — the
    // beam holds at most 64 nodes — instead of every thread walking the ranks, which was 8 % of a depth)
    const int rl = tid & 63;
    const bool r_in = rl < nb, r_ok = r_in && (!ENT || b_valid[rl] != 0);
    if constexpr (ENT) { const unsigned long long mv = __ballot(r_ok); if (mv) { best_depth = depth; best_rank = __builtin_ctzll(mv); } }
    else { best_depth = depth; best_rank = 0; }
    const unsigned long long mr = __ballot(r_ok && b_dist[rl] < fc.goal_size);
    const int reached = mr ? __builtin_ctzll(mr) : -1;   // first in rank order (every wave finds the same: uniform)
    if (reached >= 0) { status = NEP_FE_GOAL_REACHED; best_depth = depth; best_rank = reached; break; }
    if (depth == D) { status = NEP_FE_DEPTH_REACHED; break; }
  }
  atomicAdd(&s_i[4], my_children); atomicAdd(&s_i[5], my_feasible); atomicAdd(&s_i[6], my_free);
  if constexpr (ENT) { atomicAdd(&s_i[8], my_entangled); if (my_overflow) atomicOr(&s_i[9], my_overflow); if (my_big) atomicAdd(&s_i[10], my_big); }
  __syncthreads();
#ifdef NEP_PROFILE_PHASES
  if (ps.dbg && tid == 0) { for (int k = 0; k < 8; k++) ps.dbg[(long)slot * 32 + k] = tph[k]; ps.dbg[(long)slot * 32 + 8] = depth; for (int k = 0; k < 4; k++) ps.dbg[(long)slot * 32 + 12 + k] = tent[k]; }
#endif
  if (tid == 0) {
    if constexpr (!BIG) {   // this search's device time: a statistic, and the next launch's ordering key
      const double us_ = (double)((long long)wall_clock64() - t_wg0) * sp.us_per_tick;
      if (ps.fe_us) ps.fe_us[slot] = (float)us_;
      // (the key remembers: the maximum of this search's bin and the previous key less fe_key_decay — a slot whose searches alternate
      // between short and long ones, e.g. no first primitive in one round and a full search in the next, is started with the long
      // ones; started last, a long search sets the kernel's end.  Simulated on the measured times of the config-5 chain, 768
      // workgroup slots: 17.5 ms with the previous round's time as the key, 16.8 with the maximum over the rounds so far, 16.0
      // with a perfect predictor, 15.5 = sum / 768)
      if (ps.fe_order_key) { const double k_ = us_ * (ENT ? 1.0 / 256.0 : 1.0 / 8.0); const int kn = k_ > 63.0 ? 63 : (int)k_, ko = ps.fe_order_key[slot] - sp.fe_key_decay; ps.fe_order_key[slot] = (sp.fe_key_decay > 0 && ko > kn) ? ko : kn; }
      if constexpr (ENT) {   // a child outgrew the fixed record: the search is run again by the big-record instantiation (launch_frontend)
        if (s_i[9] != 0 && ea.redo_list) { const int k_ = atomicAdd(ea.redo_count, 1); if (k_ < ea.redo_cap) ea.redo_list[k_] = slot; }      // (beyond the list: the search stays flagged)
      }
    }
    nep_guess* g = guess_out + slot;
    for (int e = 0; e < 3 * NEP_MAX_POL * 4; e++) (&g->coeff[0][0][0])[e] = 0.0;
    g->t_start = st->t_start; g->K = best_rank >= 0 ? best_depth : 0; g->n_alpha = 0;
    if (best_rank >= 0) {
      double pe[6]; double pg = 0.0;
      pe[0] = st->pos[0]; pe[1] = st->pos[1]; pe[2] = st->vel[0]; pe[3] = st->vel[1]; pe[4] = st->accel[0]; pe[5] = st->accel[1];
      signed char path[NEP_MAX_POL];
      int r = best_rank;
      for (int d = best_depth; d >= 1; d--) { path[d - 1] = p_comb[d * MB + r]; r = p_parent[d * MB + r]; }
      for (int d = 1; d <= best_depth; d++) {   // replay the path from the root with the same arithmetic
        FeChild ch;
        fe_child(sp, fc, lat, pe, pg, d == 1, path[d - 1] / ns, path[d - 1] % ns, gx, gy, bx, by, ch);
        for (int k = 0; k < 4; k++) { g->coeff[0][d - 1][k] = ch.cx[k]; g->coeff[1][d - 1][k] = ch.cy[k]; }
        for (int q = 0; q < 6; q++) pe[q] = ch.e[q];
        pg = ch.g;
      }
      if (fc.pad_hold && best_depth < D) {   // hold the end point for the rest of the horizon
        for (int d = best_depth + 1; d <= D; d++) { g->coeff[0][d - 1][3] = pe[0]; g->coeff[1][d - 1][3] = pe[1]; }
        g->K = D;
      }
      fe_initial_z(st->pos[2], st->vel[2], st->accel[2], st->goal[2], sp.T_span, D, sp.v_max, sp.a_max, g->coeff[2]);   // coeffs_z_ (:540)
      for (int d = g->K + 1; d <= D; d++) for (int k = 0; k < 4; k++) g->coeff[2][d - 1][k] = 0.0;
    }
    if (ENT && (s_i[9] & 8) != 0 && ps.flags) atomicOr(ps.flags, NEP_FLAG_ENT_POOL);      // the pool of big records ran out: WHICH children were pruned depends on the claim order — sticky, nep_batch_check reports it
    if (res_out) {
      nep_fe_result* o = res_out + slot;
      o->status = status; o->K = best_rank >= 0 ? best_depth : 0; o->depth = depth > D ? D : depth;
      o->n_children = s_i[4]; o->n_feasible = s_i[5]; o->n_collision_free = s_i[6]; o->goal_occupied = s_i[7]; o->_pad = 0;
      o->cost = best_rank >= 0 ? b_f[best_rank] : 0.0; o->dist_to_goal = best_rank >= 0 ? b_dist[best_rank] : 0.0;
      o->n_entangled = ENT ? s_i[8] : 0; o->ent_overflow = ENT ? (s_i[9] != 0 ? 1 : 0) : 0; o->_pad = ENT ? (s_i[9] | (s_i[10] << 8)) : 0;      // (_pad: bit 3 = the big-record pool ran out; from bit 8 up: children carried in big records)
    }
    if constexpr (ENT) {   // ranks of the path's nodes, for the case rows below
      int r = best_rank;
      for (int d = best_depth; d >= 1; d--) { s_i[16 + d] = r; r = p_parent[d * MB + r]; }
      s_i[16] = 0; s_i[15] = best_rank >= 0 ? best_depth : -1; s_i[14] = best_rank >= 0 ? guess_out[slot].K : 0;
    }
  }
  if constexpr (ENT) {
    // case id per (knot, agent): the state at the START of segment i is the path's node of depth i (solver_gurobi_poly.cpp:624-631)
    __syncthreads();
    if (ea.case_out) {
      const int bd = s_i[15], Kg = s_i[14];
      for (int e = tid; e < NEP_MAX_POL * N; e += 256) {
        const int i = e / N, j = e % N;
        int cid = 0;
        if (bd >= 0 && i < Kg) {
          const int dd = i < bd ? i : bd;
          const nep_fe_ent_state* sn = ent_node(dd, dd == 0 ? 0 : s_i[16 + dd]);
          if (BIG && sn->n_alpha < 0) { const EntBig B = ent_big_view(ea.big, -sn->n_alpha - 1); if (ent_count(B.id, B.n_alpha, j + 1) == 1) for (int a = 0; a < B.n_alpha; a++) if (B.id[a] == j + 1) cid = B.cs[a]; }
          else if (ent_count(sn->id, sn->n_alpha, j + 1) == 1) for (int a = 0; a < sn->n_alpha; a++) if (sn->id[a] == j + 1) cid = sn->cs[a];
        }
        ea.case_out[((long)slot * NEP_MAX_POL + i) * N + j] = cid;
      }
    }
  }
}

// Diagnostic (bench.py's "how hard are these problems"): inequality rows of the spline QP (solver_gurobi_poly.cpp:433-489) whose
// slack at the returned trajectory is below tol, per slot: out[slot] = (box rows, line rows).  One wave per slot; the lines are
// the ones the separator left in the buckets (near ones from the front, parked ones from the back; an LP the spatial presolve
// skipped has no line — it is farther than the radius from where the solution was verified to be).
__global__ __launch_bounds__(64) void active_rows_kernel(SceneParams sp, ProblemSet ps, double tol, int* __restrict__ out) {
  const int slot = blockIdx.x, lane = threadIdx.x;
  const nep_solution* sol = ps.solution + slot;
  const int K = sol->K;
  __shared__ double sQ[2][NEP_MAX_POL][4];
  int nb = 0, nl = 0;
  if (lane < 3 * NEP_MAX_POL) {
    const int ax = lane / NEP_MAX_POL, i = lane % NEP_MAX_POL;
    if (i < K) {
      const double* P = sol->coeff[ax][i];
      double Q[4], V[3];
      fe_pos_cps(P, sp.T_span, Q); fe_vel_cps(P, sp.T_span, V);
      for (int k = 0; k < 4; k++) { nb += (Q[k] > sp.maxs[ax] - tol) + (Q[k] < sp.mins[ax] + tol); if (ax < 2) sQ[ax][i][k] = Q[k]; }
      for (int k = 0; k < 3; k++) nb += fabs(V[k]) > sp.v_max - tol;
      nb += fabs(6 * sp.T_span * P[0] + 2 * P[1]) > sp.a_max - tol;
    }
  }
  __syncthreads();
  for (int i = 0; i < K && i < NEP_MAX_POL; i++) {
    const long o = (long)slot * NEP_MAX_POL + i;
    const int cn = line_count(ps.line_cnt[o]), cf = ps.line_far ? ps.line_far[o] : 0;
    const double* bucket = ps.line_nd + o * sp.lines_cap * 3;
    for (int c = lane; c < cn + cf; c += 64) {
      const long q = c < cn ? (long)c : (long)sp.lines_cap - 1 - (c - cn);
      const double n1 = bucket[3 * q], n2 = bucket[3 * q + 1], d = bucket[3 * q + 2];
      if (n1 == 0.0 && n2 == 0.0 && d == 0.0) continue;      // (LP without a separating line: no rows)
      for (int k = 0; k < 4; k++) nl += (n1 * sQ[0][i][k] + n2 * sQ[1][i][k] + d - 1.0) > -tol;
    }
  }
  for (int o = 32; o; o >>= 1) { nb += __shfl_xor(nb, o); nl += __shfl_xor(nl, o); }
  if (lane == 0) { out[2 * slot] = nb; out[2 * slot + 1] = nl; }
}
void launch_active_rows(int n_slots, const SceneParams& sp, const ProblemSet& ps, double tol, int* out, hipStream_t st) {
  if (n_slots > 0) hipLaunchKernelGGL(active_rows_kernel, dim3(n_slots), dim3(64), 0, st, sp, ps, tol, out);
}

// Stand-alone batched gjk::collision (tests / nep_gjk_batch): one lane per (polygon, four points) problem.
__global__ void gjk_explicit_kernel(int n_prob, const int* __restrict__ a_off, const double* __restrict__ a_xy, const double* __restrict__ b_xy, int* __restrict__ hit) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= n_prob) return;
  Pts4 B;
  for (int i = 0; i < 4; i++) { B.x[i] = b_xy[(p * 4 + i) * 2]; B.y[i] = b_xy[(p * 4 + i) * 2 + 1]; }
  hit[p] = gjk_collision(a_off[p + 1] - a_off[p], a_xy + 2 * (long)a_off[p], B) ? 1 : 0;
}
void launch_gjk_explicit(int n_prob, const int* a_off, const double* a_xy, const double* b_xy, int* hit, hipStream_t st) {
  if (n_prob <= 0) return;
  hipLaunchKernelGGL(gjk_explicit_kernel, dim3((n_prob + 63) / 64), dim3(64), 0, st, n_prob, a_off, a_xy, b_xy, hit);
}

size_t frontend_children_cap(const nep_fe_cfg& fc, int num_pol) { return (size_t)fe_sizes(fc.beam_width, fc.num_samples, num_pol).cap; }
// survivors per round of the entangle front end's propagation pass (fast instantiation: the crossing lists that fit the borrowed LDS,
// as the kernel counts them) x sampled steps x kEntAddCap words: the global fallback of cross_round's LDS pool
size_t frontend_ent_xpool_words(const SceneParams& sp, const nep_fe_cfg& fc, int ent_ns) {
  const size_t NS = (size_t)sp.num_agents + sp.n_static;
  const FeSizes z = fe_sizes(fc.beam_width, fc.num_samples, sp.num_pol);
  const bool rf_alias = 4 * NS + kFeObsLds * kHullV * 2 >= (size_t)z.cap;
  size_t n = (sizeof(double) * (4 * NS + kFeObsLds * kHullV * 2 + (rf_alias ? 0 : (size_t)z.cap))) / kEntLdsBytes + (sizeof(int) * (size_t)z.dd) / kEntLdsBytes;
  if (n > 256) n = 256;
  return n * (size_t)(ent_ns > 0 ? ent_ns : 1) * kEntAddCap;
}
size_t frontend_lds_bytes(const SceneParams& sp, const nep_fe_cfg& fc, bool ent) {
  const size_t NS = (size_t)sp.num_agents + sp.n_static;
  const FeSizes z = fe_sizes(fc.beam_width, fc.num_samples, sp.num_pol);
  const size_t MB = (size_t)z.mb;
  const bool rf_alias = 4 * NS + kFeObsLds * kHullV * 2 >= (size_t)z.cap;      // (as in the kernel's carve)
  size_t b = sizeof(double) * ((size_t)z.cap * (rf_alias ? 1 : 2) + 2 * MB * 6 + 2 * MB + 2 * MB + 4 * MB + 4 * NS + kFeObsLds * kHullV * 2 + 4 * NEP_FE_MAX_SAMPLES)
           + sizeof(long long) * ((size_t)z.cap + z.vis) + sizeof(int) * (z.dd + 2 * NS + 32)
           + sizeof(unsigned short) * z.cap + z.cap + 2 * (NEP_MAX_POL + 1) * MB + MB;
  if (ent) b = ((b + 3) & ~(size_t)3) + sizeof(unsigned) * MB * (2 * (size_t)((sp.num_agents + 31) >> 5) + (size_t)((sp.n_static + 31) >> 5)) + (((size_t)sp.num_agents + 3) & ~(size_t)3) + sizeof(unsigned) * 2 * (size_t)((sp.num_agents + 31) >> 5) + (NEP_FE_ENT_WGS == 1 ? 256 * sizeof(nep_fe_ent_state) + 8 : 0);      // (f_bits, then cross_round's union mask and its prefix counts)
  return (b + 15) & ~(size_t)15;
}

void launch_frontend(int n_slots, const SceneParams& sp, const ProblemSet& ps_in, const nep_fe_cfg& fc, const nep_fe_start* starts,
                     nep_guess* guess_out, nep_fe_result* res_out, const FeEntArgs* ea, hipStream_t st, int* order_buf, bool have_history) {
  if (n_slots <= 0) return;
  ProblemSet ps = ps_in;
  ps.fe_order = nullptr;
  const int xcd_mode = g_debug.fe_xcd;      // (A/B, nep_debug_set_global_option "fe_xcd": 0 = the launch order without the XCD placement)
  if (order_buf && n_slots > 1024 && n_slots % 8 == 0 && xcd_mode) {      // a few whole scenes per XCD, the longest expected searches first within each (order_xcd_kernel)
    launch_order_xcd(n_slots, (ps.fe_order_key && have_history) ? ps.fe_order_key : nullptr, order_buf, st);
    ps.fe_order = order_buf;
  } else if (ps.fe_order_key && order_buf && have_history && n_slots > 1024) {      // (more than one wave of workgroups)
    launch_qp_order(n_slots, ps.fe_order_key, order_buf, st);
    ps.fe_order = order_buf;
  }
  const bool ent = ea != nullptr;
  const size_t lds = frontend_lds_bytes(sp, fc, ent);
  // a search whose LDS leaves room for four workgroups on a CU (160 KB / 4) runs the instantiation bounded to 128 registers
  // (11 spilled); larger ones — more obstacles, a wider beam — the one bounded for three
  const bool four = !ent && lds <= (size_t)40 * 1024 && !g_debug.fe_three;      // (A/B, nep_debug_set_global_option "fe_three": keep the three-workgroup instantiation)
  static DynLdsAttr attr[3];
  const void* fn = ent ? (const void*)frontend_kernel<true, NEP_FE_ENT_WGS> : four ? (const void*)frontend_kernel<false, 4> : (const void*)frontend_kernel<false, NEP_FE_WAVES>;
  (void)attr[ent ? 0 : four ? 1 : 2].ensure(fn, lds);
  FeEntArgs none{};
  launch_boxes(n_slots / (sp.n_local > 0 ? sp.n_local : 1), sp, ps, st);
  if (ent && ea->packed) {
    const int n_scenes = n_slots / (sp.n_local > 0 ? sp.n_local : 1);
    const long np_ = (long)n_scenes * sp.num_agents * sp.num_pol;
    hipLaunchKernelGGL(ent_pack_kernel, dim3((unsigned)((np_ + 255) / 256)), dim3(256), 0, st, sp, ps, *ea, n_scenes);
  }
  if (ent) {
    hipLaunchKernelGGL((frontend_kernel<true, NEP_FE_ENT_WGS>), dim3(n_slots), dim3(256), lds, st, sp, ps, fc, starts, guess_out, res_out, *ea);
    if (ea->redo_list && ea->big.base && ea->redo_cap > 0) {      // the searches listed by that launch, again, with big records (nearly always none: the workgroups return at once)
      static DynLdsAttr attr_big;
      FeEntArgs eb = *ea;
      size_t lds_b = lds;
      if (eb.big_beta && lds + 256 * (size_t)kEntBigLdsBytes <= (size_t)160 * 1024) { eb.big_lds_off = (int)lds; lds_b = lds + 256 * (size_t)kEntBigLdsBytes; } else eb.big_lds_off = 0;
      (void)attr_big.ensure((const void*)frontend_kernel<true, 1, true>, lds_b);
      hipLaunchKernelGGL((frontend_kernel<true, 1, true>), dim3(ea->redo_cap < n_slots ? ea->redo_cap : n_slots), dim3(256), lds_b, st, sp, ps, fc, starts, guess_out, res_out, eb);
    }
  }
  else if (four) hipLaunchKernelGGL((frontend_kernel<false, 4>), dim3(n_slots), dim3(256), lds, st, sp, ps, fc, starts, guess_out, res_out, none);
  else hipLaunchKernelGGL((frontend_kernel<false, NEP_FE_WAVES>), dim3(n_slots), dim3(256), lds, st, sp, ps, fc, starts, guess_out, res_out, none);
}

// Neptune::SamplePointsOfIntervals (neptune.cpp:500-565) for every committed trajectory of every scene:
// sampled[scene][j][interval][0..ns][2] on the round's grid, present[scene][j] = the trajectory exists (trajs_ holds it)
__global__ void ent_sample_kernel(const nep_traj_rec* __restrict__ recs, int n_scenes, int N, const double* __restrict__ ts0, long ts_scene_stride,
                                  int num_pol, int ns, double T_span, double* __restrict__ sampled, int* __restrict__ present, int* __restrict__ zero_this) {
  if (zero_this && blockIdx.x == 0 && threadIdx.x == 0) *zero_this = 0;      // (the re-check's pool of big records starts empty)
  const long per = (long)num_pol * (ns + 1);
  const long total = (long)n_scenes * N * per;
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
    const int col = (int)(e % (ns + 1)); long r = e / (ns + 1);
    const int i = (int)(r % num_pol); r /= num_pol;
    const int j = (int)(r % N); const int scene = (int)(r / N);
    const nep_traj_rec* rec = recs + (long)scene * N + j;
    const bool ok = rec->valid && rec->is_agent && rec->pwp.n_seg >= 1;
    if (i == 0 && col == 0) present[(long)scene * N + j] = ok ? 1 : 0;
    double ox = 0.0, oy = 0.0;
    if (ok) {
      const double t_start = *(const double*)((const char*)ts0 + (long)scene * ts_scene_stride), t_end = t_start + num_pol * T_span;
      const double deltaT = (t_end - t_start) / (1.0 * num_pol);
      const double ts = t_start + deltaT * i + deltaT / ns * col;
      const int n = rec->pwp.n_seg;
      int low = 0;                                 // std::upper_bound: first knot > ts
      while (low <= n && !(rec->pwp.times[low] > ts)) low++;
      int seg; double te;
      if (low <= n) {
        seg = low - 1;
        if (seg < 0) seg = 0; else if (seg > n - 1) seg = n - 1;
        te = ts - rec->pwp.times[seg];
        if (te < 0) te = 0; else if (te > deltaT) te = deltaT;
      } else { seg = n - 1; te = rec->pwp.times[n] - rec->pwp.times[n - 1]; }
      const double t3 = te * te * te, t2 = te * te;
      const double* cxp = rec->pwp.coeff[0][seg]; const double* cyp = rec->pwp.coeff[1][seg];
      ox = ((cxp[0] * t3 + cxp[1] * t2) + cxp[2] * te) + cxp[3] * 1.0; oy = ((cyp[0] * t3 + cyp[1] * t2) + cyp[2] * te) + cyp[3] * 1.0;
    }
    sampled[e * 2] = ox; sampled[e * 2 + 1] = oy;
  }
}
void launch_ent_sample(const nep_traj_rec* recs, int n_scenes, int N, const double* ts0, long ts_scene_stride, int num_pol, int ns, double T_span,
                       double* sampled, int* present, hipStream_t st, int* zero_this) {
  const long total = (long)n_scenes * N * num_pol * (ns + 1);
  if (total <= 0) return;
  int blocks = (int)((total + 255) / 256); if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(ent_sample_kernel, dim3(blocks), dim3(256), 0, st, recs, n_scenes, N, ts0, ts_scene_stride, num_pol, ns, T_span, sampled, present, zero_this);
}

// KinodynamicSearch::entangleCheckGivenPwp for the first interval of every new trajectory (neptune.cpp:746-754): one thread
// per (scene, agent); entangles[scene][a] = 1 turns the trajectory down in the safety pass.
__global__ __launch_bounds__(64) void ent_check_kernel(SceneParams sp, ProblemSet ps, FeEntArgs ea, const nep_traj_rec* __restrict__ fresh, int n_scenes, double cable, int* __restrict__ entangles) {
  // one wave per trajectory: the lanes first prove, 64 obstacles at a time, who cannot add a crossing anywhere along the sampled
  // interval (ent_agent_may_cross against the box of the samples), then lane 0 walks the reference's test over the rest
  extern __shared__ __attribute__((aligned(16))) unsigned m_ec[];
  const int N = sp.num_agents, S = sp.n_static, lane = threadIdx.x;
  unsigned* m_agent = m_ec; unsigned* m_static = m_ec + ((N + 63) >> 6) * 2;
  const int ec_lds_words = 2 * (((N + 63) >> 6) + ((S + 63) >> 6) + 1);      // (the masks: launch_ent_check sizes the same; an even number of words, so that the record behind them is 8-byte aligned)
  const long idx = blockIdx.x;
  const int scene = (int)(idx / N), a = (int)(idx % N);
  const nep_traj_rec* r = fresh + idx;
  if (!(r->valid && r->pwp.n_seg >= 1)) { if (lane == 0) entangles[idx] = 0; return; }
  EntCtx ec;
  ec.N = N; ec.S = S; ec.own = a; ec.num_pol = sp.num_pol; ec.ns = ea.ns; ec.T_span = sp.T_span; ec.cable = cable;
  ec.pb = ps.pb; ec.srep = ea.srep + (long)scene * sp.static_stride * 4; ec.slong = ea.slong + (long)scene * sp.static_stride * 2;
  ec.sampled = ea.sampled; ec.present = ea.present;
  ec.ps = &ps; ec.scene = scene; ec.n_hull = sp.n_hull;
  ec.packed = ea.packed; ec.pk_stride = ea.pk_stride;      // (one record per (agent, interval), launch_ent_check: every visit one round trip instead of a chain through the hull tables)
  const double* cx0 = r->pwp.coeff[0][0]; const double* cy0 = r->pwp.coeff[1][0];
  const double T = sp.T_span;
  const Ev2 end{((cx0[0] * (T * T * T) + cx0[1] * (T * T)) + cx0[2] * T) + cx0[3] * 1.0, ((cy0[0] * (T * T * T) + cy0[1] * (T * T)) + cy0[2] * T) + cy0[3] * 1.0};
  EntBox bx{fmin(cx0[3], end.x), fmax(cx0[3], end.x), fmin(cy0[3], end.y), fmax(cy0[3], end.y)};      // the samples, exactly as ent_propagate takes them
  for (int j = 1; j < ea.ns; j++) {
    const double t = sp.T_span * j / ea.ns, t3 = t * t * t, t2 = t * t;
    const double x = ((cx0[0] * t3 + cx0[1] * t2) + cx0[2] * t) + cx0[3] * 1.0, y = ((cy0[0] * t3 + cy0[1] * t2) + cy0[2] * t) + cy0[3] * 1.0;
    bx.x0 = fmin(bx.x0, x); bx.x1 = fmax(bx.x1, x); bx.y0 = fmin(bx.y0, y); bx.y1 = fmax(bx.y1, y);
  }
  bx.x0 -= 1e-9; bx.x1 += 1e-9; bx.y0 -= 1e-9; bx.y1 += 1e-9;
  for (int j0 = 0; j0 < N; j0 += 64) {
    const int j = j0 + lane;
    const bool may = j < N && j != a && (ec.packed ? (ent_agent_fbits_pk(ec, j, 0) != 0u || ent_agent_may_cross_pk(ec, bx, j, 0)) : ent_agent_may_cross(ec, bx, j, 0));
    const unsigned long long bal = __ballot(may);
    if (lane == 0) { m_agent[j0 >> 5] = (unsigned)bal; if (j0 + 32 < N) m_agent[(j0 >> 5) + 1] = (unsigned)(bal >> 32); }
  }
  for (int s0 = 0; s0 < S; s0 += 64) {
    const int s_ = s0 + lane;
    const bool may = s_ < S && ent_static_may_cross(ec, bx, s_);
    const unsigned long long bal = __ballot(may);
    if (lane == 0) { m_static[s0 >> 5] = (unsigned)bal; if (s0 + 32 < S) m_static[(s0 >> 5) + 1] = (unsigned)(bal >> 32); }
  }
  __syncthreads();
  if (lane == 0) {
    ec.m_agent = m_agent; ec.m_static = m_static;
    // (the working record in LDS: lane 0's list surgery is a chain of dependent reads and writes of this record — in global memory
    // every one of them was a round trip)
    nep_fe_ent_state* wk = (nep_fe_ent_state*)(m_ec + ec_lds_words);
    if (ea.init) ent_copy(wk, ea.init + idx); else { long* z = (long*)wk; for (int i = 0; i < (int)(sizeof(nep_fe_ent_state) / 8); i++) z[i] = 0; }
    double arc = 0.0;
    unsigned add_tail[kEntAddCap - EntAdd::reg];
    int rc = ent_propagate<EntAdd>(ec, wk, EntAdd::Store{add_tail, ea.fast_add}, cx0, cy0, end, 1, arc, false, 3);      // (fast_add: 32 unless a test shrinks it)
    if (__builtin_expect(rc >= 2, 0)) {
      // a capacity of the fixed record, not the reference's verdict: the same interval in a big record, bounded by the reference's
      // rule only (three times the search's bound here).  An exhausted pool turns the trajectory down and raises NEP_FLAG_ENT_POOL.
      if (ea.init) ent_copy(wk, ea.init + idx); else { long* z = (long*)wk; for (int i = 0; i < (int)(sizeof(nep_fe_ent_state) / 8); i++) z[i] = 0; }
      const EntBigOut bo = ent_big_child(ec, ea.big_check, wk, cx0, cy0, end, 1, false, 3);
      rc = bo.rc;
      if (rc >= 2 && ps.flags) atomicOr(ps.flags, NEP_FLAG_ENT_POOL);
    }
    entangles[idx] = rc != 0 ? 1 : 0;
  }
}
void launch_ent_check(const SceneParams& sp, const ProblemSet& ps, const FeEntArgs& ea, const nep_traj_rec* fresh, int n_scenes, double cable, int* entangles, hipStream_t st) {
  const long total = (long)n_scenes * sp.num_agents;
  if (total <= 0) return;
  const size_t lds = sizeof(unsigned) * 2 * (size_t)(((sp.num_agents + 63) >> 6) + ((sp.n_static + 63) >> 6) + 1) + sizeof(nep_fe_ent_state);
  if (ea.packed && ea.ns <= 8) {      // the packed records of the NEW trajectories (the re-check reads interval 0 only; the kernel packs them all: 10 us)
    const long np_ = (long)n_scenes * sp.num_agents * sp.num_pol;
    hipLaunchKernelGGL(ent_pack_kernel, dim3((unsigned)((np_ + 255) / 256)), dim3(256), 0, st, sp, ps, ea, n_scenes);
  }
  hipLaunchKernelGGL(ent_check_kernel, dim3((int)total), dim3(64), lds, st, sp, ps, ea, fresh, n_scenes, cable, entangles);
}

// Point A of the next round for every slot (include/neptune_frontend.h: nep_batch_next_starts): the agent's committed
// trajectory evaluated at t = previous t_start + dt — in a bulk-synchronous loop every agent replans from the same clock,
// so "deltaT states ahead on the plan" (neptune.cpp:1366-1399) is one evaluation of the committed polynomial; the
// polynomial is evaluated the way generatePwpOut samples it (solver_gurobi_poly.cpp:921-929), beyond its last knot the
// vehicle rests at the end point.  With alt != null an agent that has arrived (within r_switch of its goal, slower than
// 0.05 m/s) swaps its goal with alt[slot]: fleets that keep flying (bench.py's `moving` leg).
__global__ void next_starts_kernel(const nep_traj_rec* __restrict__ recs, int slots, int N, int first_local, int n_local, double dt,
                                   nep_fe_start* __restrict__ starts, double* __restrict__ alt, double r_switch) {
  const int slot = blockIdx.x * blockDim.x + threadIdx.x;
  if (slot >= slots) return;
  const int scene = slot / n_local, a = first_local + (slot - scene * n_local);
  const nep_traj_rec* r = recs + (long)scene * N + a;
  nep_fe_start* s = starts + slot;
  const double t = s->t_start + dt;
  s->t_start = t;
  const int n = r->pwp.n_seg;
  if (r->valid && n >= 1) {
    int i = 0;
    for (int k = 1; k < n && k < NEP_TRAJ_MAX_SEG; k++) if (t >= r->pwp.times[k]) i = k;
    const bool past = t >= r->pwp.times[n];
    double u = t - r->pwp.times[i];
    if (u < 0.0) u = 0.0;
    if (past) u = r->pwp.times[n] - r->pwp.times[n - 1];
    for (int ax = 0; ax < 3; ax++) {
      const double* c = r->pwp.coeff[ax][i];
      s->pos[ax] = ((c[0] * (u * u * u) + c[1] * (u * u)) + c[2] * u) + c[3];
      s->vel[ax] = past ? 0.0 : (c[0] * (3 * u * u) + c[1] * (2 * u)) + c[2];
      s->accel[ax] = past ? 0.0 : c[0] * (6 * u) + c[1] * 2;
    }
  }
  if (alt) {
    const double dx = s->pos[0] - s->goal[0], dy = s->pos[1] - s->goal[1];
    if (sqrt(dx * dx + dy * dy) < r_switch && sqrt(s->vel[0] * s->vel[0] + s->vel[1] * s->vel[1]) < 0.05) {
      for (int ax = 0; ax < 3; ax++) { const double g = s->goal[ax]; s->goal[ax] = alt[(long)slot * 3 + ax]; alt[(long)slot * 3 + ax] = g; }
    }
  }
}
void launch_next_starts(const nep_traj_rec* recs, int n_scenes, int N, int first_local, int n_local, double dt, nep_fe_start* starts,
                        double* alt, double r_switch, hipStream_t st) {
  const int slots = n_scenes * n_local;
  if (slots <= 0) return;
  hipLaunchKernelGGL(next_starts_kernel, dim3((slots + 255) / 256), dim3(256), 0, st, recs, slots, N, first_local, n_local, dt, starts, alt, r_switch);
}

}  // namespace nep
