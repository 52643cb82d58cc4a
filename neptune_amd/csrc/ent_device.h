// ent_device.h — entanglement state of a front-end search node on the device (enable_entangle_check).
//
// KinodynamicSearch::entanglesWithOtherAgents (reference neptune/src/kinodynamic_search.cpp:707-895) with
// eu::entangleHSigToAddAgentInd (entangle_utils.cpp:1129-1228), entangleHSigToAddStatic (:1231-1277), addAlphaBetaToList
// + breakcondition (:1402-1534, :1608-1647), updateBendPts (:1536-1604), getBendPt2d / calculateBetaForCase /
// getTetherLength (:1649-1743), getIz / power_int (kinodynamic_search.cpp:2006-2031).  One thread carries one node: the
// crossing list lives in a fixed-size record (nep_fe_ent_state, NEP_FE_ENT_CAP entries) in global memory; active_cases[i]
// is the number of list entries of agent i and is derived.  Written in the association order of the CPU checker under
// oracle/ (this header is only included by geom_kernels.hip, which is built -ffp-contract=off): betas match bit for bit.
#pragma once
#include <hip/hip_runtime.h>

#include "nep_device.h"

namespace nep {

struct EntCtx {
  int N, S, own, num_pol, ns;
  double T_span, cable;
  const double* pb;          // [N][2]
  const double* srep;        // [S][2][2] of this scene
  const double* slong;       // [S][2]
  const double* sampled;     // [scenes][N][num_pol][ns+1][2] (scene 0's / block 0's base: indexed through hull_ref like the hulls)
  const int* present;        // [scenes][N]
  const ProblemSet* ps; int scene, n_hull;      // bend points come with the hull data (hull_ref)
  // Optional (front end): bit i set = agent i / static i MAY add a crossing for the step at hand; a clear bit is a proof that it
  // cannot (frontend_kernel's per-parent masks).  nullptr: everybody is examined.
  const unsigned* m_agent = nullptr; const unsigned* m_static = nullptr;
  // Optional (front end): what the entangle check reads of agent j in interval i, packed into one record (ent_pack_kernel):
  // [present, bend count | 8 bend points | ns + 1 samples], pk_stride doubles apart — one round trip instead of a chain of four
  const double* packed = nullptr; int pk_stride = 0;
  // Optional (front end, with `packed`): per agent, bit jj - 1 = the agent's moving tether segment sweeps over OUR base in sampled step jj
  // (f1 f2 < 0 in ent_cross_agent) — the same for every child of every parent of a depth, so it is evaluated once per depth
  // (ent_agent_fbits_pk, next to the masks) instead of once per child, obstacle and step
  const unsigned char* f_bits = nullptr;
  long long* prof = nullptr;      // (NEP_PROFILE_PHASES builds: seven per-search accumulators, see scripts/fe_ent_phases.py)
};
struct Ev2 { double x, y; };
constexpr int kEntAddCap = 32;
// The crossings one sampled step adds, in the reference's order (at most kEntAddCap; more: flagged).  An entry is one word —
// id | case << 16 | nb << 24 (nb: the crossed agent's bend-point count, known where the crossing is found; the merge needs it) —
// and the first four entries are plain members that stay in registers (a step adds two or three crossings).  The tail is a
// SEPARATE array behind a pointer: an array member indexed by a variable keeps the whole struct in scratch memory (the compiler
// splits a local struct into registers only if every access to it has a constant offset) — every push was a 32-byte load and
// store of the "register" entries, and every read in the list surgery a round trip.  (Eight register entries instead of four:
// the same time — the kernel is short of registers as it is.)  An entry the merge cancels is flagged in
// `gone` and stays where it is: no shifting, and the ids of the step's crossings remain readable after the merge.
struct EntAdd {
  static constexpr int cap = kEntAddCap, reg = 4;
  static constexpr bool exact = false;      // (a full list is a capacity: the caller retries on the global-memory form below)
  unsigned r0, r1, r2, r3; unsigned* rest; int n, overflow; unsigned gone; int lim;      // rest: kEntAddCap - reg words of the caller's; lim <= cap: entries accepted
  struct Store { unsigned* rest; int lim; };      // (what outlives a sampled step: the list itself is made anew for every step, so that nothing of it is live across steps)
  __device__ __forceinline__ void attach(const Store& s) { rest = s.rest; lim = s.lim; }
  __device__ __forceinline__ unsigned get(int i) const {
    if (__builtin_expect(i >= reg, 0)) return rest[i - reg];
    const unsigned a = (i & 1) ? r1 : r0, b = (i & 1) ? r3 : r2;
    return (i & 2) ? b : a;
  }
  __device__ __forceinline__ void set(int i, unsigned v) {
    if (__builtin_expect(i >= reg, 0)) { rest[i - reg] = v; return; }
    r0 = i == 0 ? v : r0; r1 = i == 1 ? v : r1; r2 = i == 2 ? v : r2; r3 = i == 3 ? v : r3;
  }
  __device__ __forceinline__ void clear() { n = 0; overflow = 0; gone = 0u; }
  __device__ __forceinline__ bool alive(int i) const { return !((gone >> i) & 1u); }
  __device__ __forceinline__ void kill(int i) { gone |= 1u << i; }
  __device__ __forceinline__ int n_alive() const { return n - __popc(gone); }
  __device__ __forceinline__ int id(int i) const { return (int)(get(i) & 0xffffu); }
  __device__ __forceinline__ int cs(int i) const { return (int)((get(i) >> 16) & 0xffu); }
  __device__ __forceinline__ int nb(int i) const { return (int)(get(i) >> 24); }
};
// The same list without a capacity of its own: words and cancellation bits in global memory (a big record's tail, below), room for
// num_agents + statics + 2 entries.  That is exact for the reference's rule: entanglesWithOtherAgents prunes the child when the
// node's list plus the step's new crossings exceed num_agents + statics (kinodynamic_search.cpp:850-854), an agent's own pushes are
// the only ones its base-addition pop removes (entangle_utils.cpp:1129-1228: the last TWO entries, same id), so a step that fills
// this list has at least lim - 1 > num_agents + statics crossings when the agent loop ends — pruned whatever the list holds.
struct EntAddBig {
  static constexpr bool exact = true;
  unsigned* w; unsigned* gone_w; int n, overflow, n_gone, lim;
  struct Store { unsigned* w; unsigned* gone_w; int lim; };
  __device__ __forceinline__ void attach(const Store& s) { w = s.w; gone_w = s.gone_w; lim = s.lim; }
  __device__ __forceinline__ unsigned get(int i) const { return w[i]; }
  __device__ __forceinline__ void set(int i, unsigned v) { w[i] = v; }
  __device__ __forceinline__ void clear() { n = 0; overflow = 0; n_gone = 0; for (int i = 0; i < (lim + 31) >> 5; i++) gone_w[i] = 0u; }
  __device__ __forceinline__ bool alive(int i) const { return !((gone_w[i >> 5] >> (i & 31)) & 1u); }
  __device__ __forceinline__ void kill(int i) { gone_w[i >> 5] |= 1u << (i & 31); n_gone++; }
  __device__ __forceinline__ int n_alive() const { return n - n_gone; }
  __device__ __forceinline__ int id(int i) const { return (int)(get(i) & 0xffffu); }
  __device__ __forceinline__ int cs(int i) const { return (int)((get(i) >> 16) & 0xffu); }
  __device__ __forceinline__ int nb(int i) const { return (int)(get(i) >> 24); }
};
constexpr int kEntPkHead = 2 + 2 * kBend;      // doubles of a packed record before the samples: [present, bend count | pad] [8 bend points] (everything 16-byte aligned)
constexpr int kEntPkBend = 2;

// (pb and srep are device buffers: read through global-typed pointers.  As members of a struct their address space is unknown to the
// compiler, and a FLAT load counts against the LDS counter as well — every wait for an LDS read then also waits for these)
typedef __attribute__((address_space(1))) const double* ent_gcd;
__device__ __forceinline__ Ev2 ent_pb(const EntCtx& c, int j) { const ent_gcd q = (ent_gcd)c.pb; return Ev2{q[2 * j], q[2 * j + 1]}; }
__device__ __forceinline__ Ev2 ent_srep(const EntCtx& c, int s, int col) { const ent_gcd q = (ent_gcd)c.srep; return Ev2{q[(s * 2 + col) * 2], q[(s * 2 + col) * 2 + 1]}; }
// (hr: hull_ref of agent i — the samples and the presence flags travel with the hull data, in one array or in all-gathered blocks)
__device__ __forceinline__ Ev2 ent_sampled(const EntCtx& c, const HullRef& hr, int interval, int col) {
  const double* q = blk(c.sampled, hr.boff) + ((hr.e * c.num_pol + interval) * (c.ns + 1) + col) * 2; return Ev2{q[0], q[1]};
}
__device__ __forceinline__ double ent_wedge(Ev2 a, Ev2 b, Ev2 cc) { return (b.x - a.x) * (cc.y - a.y) - (cc.x - a.x) * (b.y - a.y); }
__device__ __forceinline__ double ent_wedge2(Ev2 a, Ev2 b, Ev2 cc, Ev2& ab, Ev2& ac) { ab.x = b.x - a.x; ab.y = b.y - a.y; ac.x = cc.x - a.x; ac.y = cc.y - a.y; return ab.x * ac.y - ac.x * ab.y; }
__device__ __forceinline__ double ent_ratio(Ev2 u, Ev2 v) { return fabs(u.y * v.y) > fabs(u.x * v.x) ? u.y / v.y : u.x / v.x; }
__device__ __forceinline__ double ent_dist(Ev2 a, Ev2 b) { return sqrt((a.x - b.x) * (a.x - b.x) + (a.y - b.y) * (a.y - b.y)); }
template <class ADD> __device__ __forceinline__ void ent_push(ADD& a, int id, int cs, int nb = 0) { if (a.n < a.lim) { a.set(a.n, (unsigned)(id & 0xffff) | ((unsigned)(cs & 0xff) << 16) | ((unsigned)(nb & 0xff) << 24)); a.n++; } else a.overflow = 1; }

// ---- proofs that an obstacle adds no crossing for ANY sampled step inside a box (frontend_kernel's per-parent masks,
// ent_check_kernel's per-trajectory mask).  A step p_k -> p_k1 adds a crossing with a tether segment only if the two points lie
// on strictly opposite sides of the segment's line (ent_cross_agent: c1 c2 < 0), and the wedge is affine in the point: if all
// four corners of a box that holds every sample are on one side by a margin far above the wedge's rounding (1e-6 m^2 against
// < 1e-10), so is every sample, in exact and in floating-point arithmetic.  For the last tether segment, whose far end moves
// with the agent, the same side of all ns + 1 sampled lines.  The one test that does not involve the step (the agent's tether
// sweeping over OUR base) is evaluated as is.  false is a proof; true only means "run the reference's test".
struct EntBox { double x0, x1, y0, y1; };
__device__ __forceinline__ int ent_side(const EntBox& q, Ev2 b, Ev2 cc) {      // +1 / -1: every corner strictly on that side of the line (as ent_wedge(p, b, cc) sees it); 0: undecided
  const double kEps = 1e-6;
  const double w0 = ent_wedge(Ev2{q.x0, q.y0}, b, cc), w1 = ent_wedge(Ev2{q.x1, q.y0}, b, cc), w2 = ent_wedge(Ev2{q.x0, q.y1}, b, cc), w3 = ent_wedge(Ev2{q.x1, q.y1}, b, cc);
  const double lo = fmin(fmin(w0, w1), fmin(w2, w3)), hi = fmax(fmax(w0, w1), fmax(w2, w3));
  return lo > kEps ? 1 : (hi < -kEps ? -1 : 0);
}
__device__ bool ent_agent_may_cross(const EntCtx& c, const EntBox& q, int j, int interval) {      // (j != own)
  const HullRef hr = hull_ref(*c.ps, c.n_hull, c.scene, j);
  if (!blk(c.present, hr.boff)[hr.e]) return false;
  const int nbj = blk(c.ps->bend_n, hr.boff)[hr.e];
  const double* bp = blk(c.ps->bend_xy, hr.boff) + hr.e * kBend * 2;
  const Ev2 pb_self = ent_pb(c, c.own);
  bool maybe = false;
  for (int k = 0; k + 1 < nbj; k++) maybe |= ent_side(q, Ev2{bp[2 * (k + 1)], bp[2 * (k + 1) + 1]}, Ev2{bp[2 * k], bp[2 * k + 1]}) == 0;
  if (nbj >= 1) {
    const Ev2 bk{bp[2 * (nbj - 1)], bp[2 * (nbj - 1) + 1]};
    Ev2 pik = ent_sampled(c, hr, interval, 0);
    const int s0 = ent_side(q, pik, bk);
    maybe |= s0 == 0;
    for (int jj = 1; jj <= c.ns; jj++) {
      const Ev2 pik1 = ent_sampled(c, hr, interval, jj);
      maybe |= ent_side(q, pik1, bk) != s0;
      const double f1 = ent_wedge(pb_self, pik, bk), f2 = ent_wedge(pb_self, pik1, bk);
      maybe |= f1 * f2 < 0;
      pik = pik1;
    }
  }
  return maybe;
}
// the same proof from the packed record (front end): every load is independent of the others
__device__ __forceinline__ const double* ent_rec(const EntCtx& c, int j, int interval) { return c.packed + (((long)c.scene * c.N + j) * c.num_pol + interval) * c.pk_stride; }
__device__ __forceinline__ unsigned ent_agent_fbits_pk(const EntCtx& c, int j, int interval) {      // (j != own; ns <= 8)
  const double* r = ent_rec(c, j, interval);
  const int2 hd = *(const int2*)r;
  if (!hd.x || hd.y < 1) return 0u;
  const double* bp = r + kEntPkBend; const double* sm = r + kEntPkHead;
  const Ev2 pb_self = ent_pb(c, c.own);
  const Ev2 bk{bp[2 * (hd.y - 1)], bp[2 * (hd.y - 1) + 1]};
  Ev2 pik{sm[0], sm[1]};
  unsigned fb = 0u;
  for (int jj = 1; jj <= c.ns; jj++) {
    const Ev2 pik1{sm[2 * jj], sm[2 * jj + 1]};
    const double f1 = ent_wedge(pb_self, pik, bk), f2 = ent_wedge(pb_self, pik1, bk);
    fb |= (f1 * f2 < 0 ? 1u : 0u) << (jj - 1);
    pik = pik1;
  }
  return fb;
}
// (the box part of the proof: the caller has looked at ent_agent_fbits_pk — any bit set means "maybe" for every box)
__device__ bool ent_agent_may_cross_pk(const EntCtx& c, const EntBox& q, int j, int interval) {      // (j != own)
  const double* r = ent_rec(c, j, interval);
  const int2 hd = *(const int2*)r;
  if (!hd.x) return false;
  const int nbj = hd.y;
  const double* bp = r + kEntPkBend; const double* sm = r + kEntPkHead;
  bool maybe = false;
  for (int k = 0; k + 1 < nbj; k++) maybe |= ent_side(q, Ev2{bp[2 * (k + 1)], bp[2 * (k + 1) + 1]}, Ev2{bp[2 * k], bp[2 * k + 1]}) == 0;
  if (nbj >= 1) {
    const Ev2 bk{bp[2 * (nbj - 1)], bp[2 * (nbj - 1) + 1]};
    const int s0 = ent_side(q, Ev2{sm[0], sm[1]}, bk);
    maybe |= s0 == 0;
    for (int jj = 1; jj <= c.ns; jj++) maybe |= ent_side(q, Ev2{sm[2 * jj], sm[2 * jj + 1]}, bk) != s0;
  }
  return maybe;
}
__device__ __forceinline__ bool ent_static_may_cross(const EntCtx& c, const EntBox& q, int s) { return ent_side(q, ent_srep(c, s, 1), ent_srep(c, s, 0)) == 0; }

// (f_known: -1 = evaluate the base-sweep test here; 0 / 1 = its outcome, from EntCtx::f_bits)
// (b0: bend point 0 in registers, where the caller has fetched it with the record's header — most tethers have no other, and a load
// issued here waits for everything the caller has requested ahead, i.e. for the NEXT obstacle's record)
template <class ADD, class BP = const double*> __device__ __forceinline__ void ent_cross_agent(ADD& add, Ev2 pk, Ev2 pk1, Ev2 pik, Ev2 pik1, Ev2 pb_self, int nb, BP bp, int agent_id, int f_known = -1, bool have_b0 = false, Ev2 b0 = Ev2{0, 0}, Ev2 b1 = Ev2{0, 0}) {
  bool base_addition = false;
  // (have_b0: bend points 0 AND 1 are in registers, fetched by the caller with the record's header; a segment's far end is kept for
  // the next segment's near end.  Read where they are used, a tether of two to four bend points — the bench's config-5 tethers — cost
  // two to four round trips per visit; now none, one or two)
  Ev2 bk = b0;
  if (!have_b0 && nb > 0) { bk.x = bp[0]; bk.y = bp[1]; }
  for (int k = 0; k < nb; k++) {
    const bool last = k == nb - 1;
    Ev2 u, v, bn = bk; double c1, c2;
    if (!last) { if (have_b0 && k == 0) bn = b1; else { bn.x = bp[2 * (k + 1)]; bn.y = bp[2 * (k + 1) + 1]; } c1 = ent_wedge2(pk, bn, bk, u, v); c2 = ent_wedge(pk1, bn, bk); }
    else { c1 = ent_wedge2(pk, pik, bk, u, v); c2 = ent_wedge(pk1, pik1, bk); }
    if (last) {
      Ev2 ub, vb; bool sweeps;
      if (f_known < 0) { const double f1 = ent_wedge2(pb_self, pik, bk, ub, vb), f2 = ent_wedge(pb_self, pik1, bk); sweeps = f1 * f2 < 0; }
      else { sweeps = f_known != 0; ub.x = pik.x - pb_self.x; ub.y = pik.y - pb_self.y; vb.x = bk.x - pb_self.x; vb.y = bk.y - pb_self.y; }
      if (sweeps) {
        const double a = ent_ratio(ub, vb);
        if (a < 0) { }
        else if (a < 1) ent_push(add, agent_id, 1, nb);
        else if (k == 0) ent_push(add, agent_id, 0, nb);
        base_addition = true;
      }
    }
    if (c1 * c2 < 0) {
      const double a = ent_ratio(u, v);
      if (a < 0) ent_push(add, agent_id, k + 2, nb);
      else if (a < 1 && last) ent_push(add, agent_id, 1, nb);
      else if (a >= 1 && k == 0) ent_push(add, agent_id, 0, nb);
    }
    bk = bn;
  }
  if (base_addition && add.n >= 2 && ((add.get(add.n - 1) ^ add.get(add.n - 2)) & 0xffffffu) == 0u) add.n -= 2;      // (same id, same case)
}
template <class ADD> __device__ __forceinline__ void ent_cross_static(ADD& add, Ev2 pk, Ev2 pk1, const EntCtx& c) {
  if (c.m_static) {
    // (front end: the candidates of this parent in index order, the NEXT one's representative requested before the current one's
    // wedges are evaluated — as for the agents, ent_propagate)
    const int SWn = (c.S + 31) >> 5;
    int mw = -1; unsigned cw = 0u;      // (the mask word at hand in a register, as for the agents)
    auto next_cand = [&]() -> int {
      while (cw == 0u) { if (++mw >= SWn) return c.S; cw = c.m_static[mw]; }
      const int i = (mw << 5) + __ffs(cw) - 1; cw &= cw - 1u;
      if (i >= c.S) { cw = 0u; mw = SWn; return c.S; }
      return i;
    };
    int s = next_cand();
    const ent_gcd r = (ent_gcd)c.srep + (long)(s < c.S ? s : 0) * 4;
    double2 q0{r[0], r[1]}, q1{r[2], r[3]};
    while (s < c.S) {
      const int sn = next_cand();
      const ent_gcd rn = (ent_gcd)c.srep + (long)(sn < c.S ? sn : 0) * 4;
      const double2 n0{rn[0], rn[1]}, n1{rn[2], rn[3]};
      const Ev2 pik{q1.x, q1.y}, pbi{q0.x, q0.y};
      Ev2 u, v;
      const double c1 = ent_wedge2(pk, pik, pbi, u, v), c2 = ent_wedge(pk1, pik, pbi);
      if (c1 * c2 < 0) {
        const double a = ent_ratio(u, v);
        if (a < 0) { }
        else if (a < 1) ent_push(add, c.N + s + 1, 1);
        else ent_push(add, c.N + s + 1, 0);
      }
      s = sn; q0 = n0; q1 = n1;
    }
    return;
  }
  for (int s = 0; s < c.S; s++) {
    const Ev2 pik = ent_srep(c, s, 1), pbi = ent_srep(c, s, 0);
    Ev2 u, v;
    const double c1 = ent_wedge2(pk, pik, pbi, u, v), c2 = ent_wedge(pk1, pik, pbi);
    if (c1 * c2 < 0) {
      const double a = ent_ratio(u, v);
      if (a < 0) { }
      else if (a < 1) ent_push(add, c.N + s + 1, 1);
      else ent_push(add, c.N + s + 1, 0);
    }
  }
}
__device__ __forceinline__ Ev2 ent_anchor(int id, int cs, const EntCtx& c) { return id <= c.N ? ent_pb(c, id - 1) : ent_srep(c, id - c.N - 1, cs); }
template <class ST> __device__ Ev2 ent_cur_bend(const ST* st, Ev2 pb_self, const EntCtx& c) {
  if (st->n_bend == 0) return pb_self;
  const int b = st->bend[st->n_bend - 1];
  const int id = st->id[b], cs = st->cs[b];
  if (id <= c.N && id >= 1) return ent_pb(c, id - 1);
  if (id > c.N) return ent_srep(c, id - c.N - 1, cs);
  return Ev2{0, 0};
}
__device__ __forceinline__ double ent_beta(int id, int cs, Ev2 pk, Ev2 bp, const EntCtx& c) { return id <= c.N ? 0.0 : ent_wedge(pk, ent_srep(c, id - c.N - 1, cs), bp); }
// breakcondition (entangle_utils.cpp:1608-1647), where ent_merge's walk over the list ends after an entry that did not cancel: a
// crossing of an agent's tether segment (case >= 2) is looked for down to the last bend point; of an agent otherwise, in the whole
// list; of an obstacle, down to the last bend point or the first obstacle entry, whichever comes first.
// The beta of an AGENT crossing is 0.0 by the reference's own rule (calculateBetaForCase, entangle_utils.cpp:1713-1719: only static
// obstacles carry one).  The LDS-resident view (EntLds, below) keeps its betas in global memory and relies on that: it neither
// reads nor rewrites the beta of an agent entry — reads are round trips the list surgery would wait for, and nearly every entry is
// an agent's.  nep_fe_ent_state itself is handled as written (every beta moved and compared as stored).
struct EntLds; struct EntBig;
template <class ST> struct ent_lazy_beta { static constexpr bool v = false; };
template <> struct ent_lazy_beta<EntLds> { static constexpr bool v = true; };
template <> struct ent_lazy_beta<EntBig> { static constexpr bool v = true; };
// capacities of a state's form and the type of its bend indices (the fixed record and its LDS view: NEP_FE_ENT_CAP entries at most,
// NEP_MAX_BEND bend points, byte indices; the big record: as many as the reference's rule can ever leave on a list, 16-bit indices)
template <class ST> struct ent_tr {
  typedef signed char bend_t;
  static __device__ __forceinline__ int cap(const ST*) { return NEP_FE_ENT_CAP; }
  static __device__ __forceinline__ int bend_cap(const ST*) { return NEP_MAX_BEND; }
};
// A 64-bit signature of the ids on a list (bit id mod 64): the cancellation scan of a new crossing and the per-agent counts walk the
// whole list looking for entries of ONE id, and nearly always there is none (a crossing with somebody not crossed before is the
// common case) — a clear bit proves that without the walk.  Kept by the LDS view only (set on append, never cleared: a stale bit
// costs a walk, not a wrong answer).
template <class ST> __device__ __forceinline__ bool ent_sig_may_have(const ST*, int) { return true; }
template <class ST> __device__ __forceinline__ void ent_sig_add(ST*, int) { }
template <class ST> __device__ __forceinline__ unsigned long long ent_sig_of(const ST*) { return ~0ull; }
template <class ST> __device__ void ent_erase(ST* st, int j, int N) {
  constexpr bool lazy = ent_lazy_beta<ST>::v;
  for (int k = j; k + 1 < st->n_alpha; k++) {
    const int id_new = st->id[k + 1];
    if (!lazy) st->beta[k] = st->beta[k + 1];
    else if (id_new > N) st->beta[k] = st->beta[k + 1];      // (an agent entry's beta is never looked at: nothing to move)
    st->id[k] = (short)id_new; st->cs[k] = st->cs[k + 1];
  }
  st->n_alpha--;
}
__device__ int ent_bend_n(const EntCtx& c, int j) { const HullRef hr = hull_ref(*c.ps, c.n_hull, c.scene, j); return blk(c.ps->bend_n, hr.boff)[hr.e]; }
#ifdef NEP_PROFILE_PHASES
#define ENT_MT(k) do { if (mt) { const long long t_ = clock64(); mt[k] += t_ - *ml; *ml = t_; } } while (0)
#else
#define ENT_MT(k) do { } while (0)
#endif
template <class ST, class ADD> __device__ __forceinline__ bool ent_merge(ADD& add, ST* st, Ev2 pk, Ev2 pb_self, const EntCtx& c, long long* mt = nullptr, long long* ml = nullptr) {
  bool again = true;
  while (again) {
    again = false;
    const int b = st->n_bend ? st->bend[st->n_bend - 1] : -1;
    for (int i = 0; i < add.n && !again; i++) {
      if (!add.alive(i)) continue;
      const unsigned t_w = add.get(i);
      const int t_id = (int)(t_w & 0xffffu), t_cs = (int)((t_w >> 16) & 0xffu);
      const bool agent = t_id <= c.N;
      const int t_nb = agent ? (int)(t_w >> 24) : 0;
      if (!ent_sig_may_have(st, t_id)) continue;      // (every match needs an entry of the same id: there is none)
      // (the three ways an entry cancels against the new crossing and the rule that ends the walk — breakcondition, above — as flag
      // arithmetic: evaluated with short-circuit branches, an iteration of this walk was a dozen jumps)
      const bool t_deep = agent & (t_cs >= t_nb + 1), t_bend = agent & (t_cs >= 2);
      const bool stop_at_bend = !agent | t_bend, stop_at_static = !agent;
      // (the walk is a chain of list reads: the entry below is requested before this one's tests)
      int j = st->n_alpha - 1, l_id = 0, l_cs = 0;
      if (j >= 0) { l_id = st->id[j]; l_cs = st->cs[j]; }
      for (; j >= 0; j--) {
        int n_id = 0, n_cs = 0;
        if (j > 0) { n_id = st->id[j - 1]; n_cs = st->cs[j - 1]; }
        const int dcs = t_cs - l_cs;
        const bool match = (l_id == t_id) & ((dcs == 0) | (t_deep & (dcs < 0)) | (t_bend & (l_cs >= 2) & ((dcs == 1) | (dcs == -1)) & (j > b)));
        if (match) {
          add.kill(i);
          ENT_MT(1);
          ent_erase(st, j, c.N);      // (the id signature keeps the erased entry's bit: a set bit only ever costs a walk)
          if (j == b) {
            st->n_bend--;
            const Ev2 bp = ent_cur_bend(st, pb_self, c);
            for (int k = j; k < st->n_alpha; k++) if (!ent_lazy_beta<ST>::v || st->id[k] > c.N) st->beta[k] = ent_beta(st->id[k], st->cs[k], pk, bp, c);
          } else if (j < b) {
            st->bend[st->n_bend - 1] = (typename ent_tr<ST>::bend_t)(b - 1);
            for (int k = st->n_bend - 2; k >= 0; k--) { if (st->bend[k] > j) st->bend[k] -= 1; else break; }
          }
          again = true;
          ENT_MT(2);
          break;
        }
        if ((stop_at_bend & (j <= b)) | (stop_at_static & (l_id > c.N))) break;
        l_id = n_id; l_cs = n_cs;
      }
    }
  }
  ENT_MT(1);
  const int n_new = add.n_alive();
  if (n_new == 0) return false;
  if (st->n_alpha + n_new > ent_tr<ST>::cap(st)) return true;
  const Ev2 bp = ent_cur_bend(st, pb_self, c);
  for (int i = 0; i < add.n; i++) {
    if (!add.alive(i)) continue;
    st->id[st->n_alpha] = (short)add.id(i); st->cs[st->n_alpha] = (signed char)add.cs(i); ent_sig_add(st, add.id(i));
    if (!ent_lazy_beta<ST>::v || add.id(i) > c.N) st->beta[st->n_alpha] = ent_beta(add.id(i), add.cs(i), pk, bp, c);
    st->n_alpha++;
  }
  ENT_MT(3);
  return false;
}
template <class ST> __device__ bool ent_update_bends(ST* st, Ev2 pk1, Ev2 pb_self, const EntCtx& c) {
  const Ev2 bp = ent_cur_bend(st, pb_self, c);
  int idx_new = -1;
  const int start = st->n_bend ? st->bend[st->n_bend - 1] : -1;
  for (int i = start + 1; i < st->n_alpha; i++) {
    if (ent_lazy_beta<ST>::v && st->id[i] <= c.N) continue;      // (0.0 times the stored 0.0)
    const double beta = ent_beta(st->id[i], st->cs[i], pk1, bp, c); if (beta * st->beta[i] < -1e-7) idx_new = i;
  }
  if (idx_new > -1) {
    if (st->n_bend >= ent_tr<ST>::bend_cap(st)) return true;
    st->bend[st->n_bend++] = (typename ent_tr<ST>::bend_t)idx_new;
    const Ev2 nb = ent_anchor(st->id[idx_new], st->cs[idx_new], c);
    for (int i = idx_new + 1; i < st->n_alpha; i++) if (!ent_lazy_beta<ST>::v || st->id[i] > c.N) st->beta[i] = ent_beta(st->id[i], st->cs[i], pk1, nb, c);
    return false;
  }
  while (st->n_bend) {
    const Ev2 prev = st->n_bend == 1 ? pb_self : ent_anchor(st->id[st->bend[st->n_bend - 2]], st->cs[st->bend[st->n_bend - 2]], c);
    const int bi = st->bend[st->n_bend - 1];
    const double beta = ent_beta(st->id[bi], st->cs[bi], pk1, prev, c);
    const double stored = (ent_lazy_beta<ST>::v && st->id[bi] <= c.N) ? 0.0 : st->beta[bi];
    if (beta * stored > 1e-7) {
      for (int k = bi + 1; k < st->n_alpha; k++) if (!ent_lazy_beta<ST>::v || st->id[k] > c.N) st->beta[k] = ent_beta(st->id[k], st->cs[k], pk1, prev, c);
      st->n_bend--;
    } else break;
  }
  return false;
}
template <class ST> __device__ double ent_tether(const ST* st, Ev2 from, Ev2 pk1, const EntCtx& c) {
  double len = 0.0;
  for (int q = 0; q < st->n_bend; q++) {
    const int b = st->bend[q]; const int id = st->id[b], cs = st->cs[b];
    Ev2 bp; double comp;
    if (id <= c.N) { bp = ent_pb(c, id - 1); comp = 0.0; } else { bp = ent_srep(c, id - c.N - 1, cs); comp = c.slong[(id - c.N - 1) * 2 + cs]; }
    len += ent_dist(bp, from) + 2 * comp;
    from = bp;
  }
  return len + ent_dist(pk1, from);
}
template <class P> __device__ __forceinline__ int ent_count(P ids, int n, int id) { int k = 0; for (int i = 0; i < n; i++) k += ids[i] == id; return k; }

// 0: fine; 1: the reference's function returns true (prune); 2, 3, 4: a capacity exceeded (pruned, flagged; which one: see the returns)
// (ADD, store: the form of the list of a step's new crossings and its storage — EntAdd with its tail, or EntAddBig on a big record)
template <class ADD, class ST> __device__ int ent_propagate(const EntCtx& c, ST* st, const typename ADD::Store& store, const double* cxo, const double* cyo, Ev2 end, int index, double& arc, bool check_tether, int cap_mult) {
  const int ns = c.ns;
  const Ev2 pb_self = ent_pb(c, c.own);
  Ev2 pk{cxo[3], cyo[3]}, pk1 = pk;
#ifdef NEP_PROFILE_PHASES
  long long pt[5] = {0, 0, 0, 0, 0}; long long pl = clock64(); int p_add = 0, p_chg = 0, p_visit = 0; long long mt[5] = {0, 0, 0, 0, 0}, ml = 0;
#define ENT_PT(k) do { const long long t_ = clock64(); pt[k] += t_ - pl; pl = t_; } while (0)
#else
#define ENT_PT(k) do { } while (0)
#endif
  for (int j = 1; j <= ns; j++) {
    ADD add; add.attach(store); add.clear();
    ENT_PT(4);
    if (j < ns) {
      const double t = c.T_span * j / ns;
      const double t3 = t * t * t, t2 = t * t;
      pk1.x = ((cxo[0] * t3 + cxo[1] * t2) + cxo[2] * t) + cxo[3] * 1.0; pk1.y = ((cyo[0] * t3 + cyo[1] * t2) + cyo[2] * t) + cyo[3] * 1.0;
    } else pk1 = end;
    arc += ent_dist(pk1, pk);
    if (c.packed) {
      // front end: the candidates of this parent in index order (set bits of its mask), one packed record per (agent, interval);
      // the NEXT candidate's header and samples are requested before the current one is worked on (the loop is bound by that
      // round trip, not by the four wedges of a crossing test)
      const int itv = index > c.num_pol ? c.num_pol - 1 : index - 1;
      const int jl = index > c.num_pol ? ns : j - 1, jr = index > c.num_pol ? ns : j;
      const int MWn = (c.N + 31) >> 5;
      // (the mask word at hand is kept in a register and consumed bit by bit: a read of the LDS word per candidate sat on the path
      // from one candidate's index to the request for the next one's record)
      int mw = -1; unsigned cw = 0u;
      auto next_cand = [&]() -> int {
        while (cw == 0u) { if (++mw >= MWn) return c.N; cw = c.m_agent ? c.m_agent[mw] : ~0u; }
        const int i = (mw << 5) + __ffs(cw) - 1; cw &= cw - 1u;
        if (i >= c.N) { cw = 0u; mw = MWn; return c.N; }
        return i;
      };
      int i = next_cand();
      const double* r = ent_rec(c, i < c.N ? i : 0, itv);
      int2 hd = *(const int2*)r; double2 sa = *(const double2*)(r + kEntPkHead + 2 * jl), sb = *(const double2*)(r + kEntPkHead + 2 * jr), b0 = *(const double2*)(r + kEntPkBend), b1 = *(const double2*)(r + kEntPkBend + 2);
      while (i < c.N) {
        const int in = next_cand();
        const double* rn = ent_rec(c, in < c.N ? in : 0, itv);
        const int fk = !c.f_bits ? -1 : (index > c.num_pol ? 0 : (int)((c.f_bits[i] >> (j - 1)) & 1u));      // (holding at the end: pik = pik1, f1 f2 = f1^2)
        const int2 hdn = *(const int2*)rn; const double2 san = *(const double2*)(rn + kEntPkHead + 2 * jl), sbn = *(const double2*)(rn + kEntPkHead + 2 * jr), b0n = *(const double2*)(rn + kEntPkBend), b1n = *(const double2*)(rn + kEntPkBend + 2);
        if (i != c.own && hd.x) ent_cross_agent(add, pk, pk1, Ev2{sa.x, sa.y}, Ev2{sb.x, sb.y}, pb_self, hd.y, r + kEntPkBend, i + 1, fk, true, Ev2{b0.x, b0.y}, Ev2{b1.x, b1.y});
#ifdef NEP_PROFILE_PHASES
        p_visit++;
#endif
        i = in; r = rn; hd = hdn; sa = san; sb = sbn; b0 = b0n; b1 = b1n;
      }
    } else
    for (int i = 0; i < c.N; i++) {
      if (c.m_agent && !((c.m_agent[i >> 5] >> (i & 31)) & 1u)) { i |= 31 * !c.m_agent[i >> 5]; continue; }      // (an empty word is skipped whole)
      if (i == c.own) continue;
      const HullRef hr = hull_ref(*c.ps, c.n_hull, c.scene, i);
      if (!blk(c.present, hr.boff)[hr.e]) continue;
      Ev2 pik, pik1;
      if (index > c.num_pol) { pik = ent_sampled(c, hr, c.num_pol - 1, ns); pik1 = pik; }
      else { pik = ent_sampled(c, hr, index - 1, j - 1); pik1 = ent_sampled(c, hr, index - 1, j); }
      ent_cross_agent(add, pk, pk1, pik, pik1, pb_self, blk(c.ps->bend_n, hr.boff)[hr.e], blk(c.ps->bend_xy, hr.boff) + hr.e * kBend * 2, i + 1);
    }
    ENT_PT(0);
    ent_cross_static(add, pk, pk1, c);
    ENT_PT(1);
#ifdef NEP_PROFILE_PHASES
    p_add += add.n > 0;
#endif
    if (add.overflow) return ADD::exact ? 1 : 3;      // (EntAddBig: see there; 2: the list's capacity, 3: more than kEntAddCap new crossings in one step, 4: more than NEP_MAX_BEND bend points)
    if (st->n_alpha + add.n > (c.N + c.S) * cap_mult) return 1;
    if (add.n > 0) {     // (no crossing in this step: the list, and with it every count below, is what it was)
      // entanglesWithOtherAgents compares, for every agent in the new list, its number of entries before and after the merge
      // (:870-887: a second entry where there was at most one, or one more where there were two, prunes the child).  The merge
      // only ever touches entries of the agents in `add`, so theirs are the only counts that can differ: the old ones are
      // taken before the merge, for those agents only — no copy of the old list, no pass over every pair of entries.
      // With m matches of an agent in the merge (each removes one entry of the new list and one of the old), its count moves from
      // od to od - m + rem, rem = init - m its entries left in the new list: by d = 2 rem - init, known without looking at the old
      // list at all.  d <= 0 never prunes; d >= 2 always does; d = 1 prunes iff the agent had an entry before (od >= 1, i.e. two
      // entries now) — and the list's id signature, taken BEFORE the merge, answers that with "no" for the common case of an agent
      // crossed for the first time.  (Entries of one id are adjacent in both lists: the loops over obstacles run in index order.)
      const unsigned long long sig_before = ent_sig_of(st);
#ifdef NEP_PROFILE_PHASES
      ml = pl; { const long long t_ = clock64(); mt[0] += t_ - ml; ml = t_; }
      if (ent_merge(add, st, pk, pb_self, c, mt, &ml)) return 2;
#else
      if (ent_merge(add, st, pk, pb_self, c)) return 2;
#endif
      for (int e = 0; e < add.n;) {
        const int id_ = add.id(e);
        int init = 0, rem = 0;
        do { rem += add.alive(e) ? 1 : 0; init++; e++; } while (e < add.n && add.id(e) == id_);
        if (id_ > c.N) continue;
        const int d = 2 * rem - init;
        if (d <= 0) continue;
        if (d >= 2) return 1;
        if (!((sig_before >> (id_ & 63)) & 1ull)) continue;
        if (ent_count(st->id, st->n_alpha, id_) >= 2) return 1;
      }
#ifdef NEP_PROFILE_PHASES
      { const long long t_ = clock64(); mt[4] += t_ - ml; }
#endif
    }
    ENT_PT(2);
#ifdef NEP_PROFILE_PHASES
    const int nb_before_ = st->n_bend, lb_before_ = st->n_bend ? st->bend[st->n_bend - 1] : -1;
#endif
    if (ent_update_bends(st, pk1, pb_self, c)) return 4;
#ifdef NEP_PROFILE_PHASES
    p_chg |= (add.n_alive() > 0) | (st->n_bend != nb_before_) | ((st->n_bend ? st->bend[st->n_bend - 1] : -1) != lb_before_);
#endif
    ENT_PT(3);
    pk = pk1;
  }
  const bool too_long = check_tether && ent_tether(st, pb_self, pk1, c) > c.cable;
  ENT_PT(4);
#ifdef NEP_PROFILE_PHASES
  if (c.prof) { for (int k = 0; k < 5; k++) atomicAdd((unsigned long long*)c.prof + k, (unsigned long long)pt[k]); atomicAdd((unsigned long long*)c.prof + 5, 1ull); atomicAdd((unsigned long long*)c.prof + 6, (unsigned long long)p_add);
    atomicAdd((unsigned long long*)c.prof + 7, (unsigned long long)(p_chg != 0)); atomicMax((unsigned long long*)c.prof + 8, (unsigned long long)st->n_alpha); atomicAdd((unsigned long long*)c.prof + 9, (unsigned long long)st->n_alpha);
    for (int k = 0; k < 5; k++) atomicAdd((unsigned long long*)c.prof + 10 + k, (unsigned long long)mt[k]); atomicAdd((unsigned long long*)c.prof + 15, (unsigned long long)p_visit); }
#endif
  if (too_long) return 1;
  return 0;
}
// ---- The front end's fast path, in two passes (round 5).  Which crossings a sampled step adds does not depend on the entangle state:
// it is a function of the child's polynomial, the other agents' tethers and the static representatives alone.  So the kernel finds
// them in a DENSE pass of its own — one thread per (child, sampled step), the candidates' packed records staged in LDS, the lists of
// new crossings left in an LDS pool (ent_cross_step) — and the per-child pass that follows (ent_propagate_pre) is the list surgery
// alone: merge, counts, bend points, tether length, in the reference's order, on exactly the lists ent_propagate would have made
// (same ent_cross_agent / ent_cross_static, same operands, same order: bit-identical states).  Before, every child walked its
// candidates' records in global memory inside the surgery loop: 17 visits of 4 000 - 8 000 cycles per propagation, the lanes of a wave
// on different parents' candidate lists (65 % of a propagation).
// A (child, step)'s list in the pool: header word = offset | n << 16 | overflow << 24 (kEntHdrNone: not made).
constexpr unsigned kEntHdrOvf = 1u << 24, kEntHdrGlobal = 1u << 25;      // (global: the words are in the pair's block of FeEntArgs::xpool, the LDS pool was full)
template <class REC> __device__ __forceinline__ void ent_cross_step(EntAdd& add, const EntCtx& c, Ev2 pk, Ev2 pk1, int index, int j, REC rec_of) {
  const int ns = c.ns;
  const Ev2 pb_self = ent_pb(c, c.own);
  const int itv = index > c.num_pol ? c.num_pol - 1 : index - 1;
  const int jl = index > c.num_pol ? ns : j - 1, jr = index > c.num_pol ? ns : j;
  const int MWn = (c.N + 31) >> 5;
  int mw = -1; unsigned cw = 0u;
  auto next_cand = [&]() -> int {
    while (cw == 0u) { if (++mw >= MWn) return c.N; cw = c.m_agent ? c.m_agent[mw] : ~0u; }
    const int i = (mw << 5) + __ffs(cw) - 1; cw &= cw - 1u;
    if (i >= c.N) { cw = 0u; mw = MWn; return c.N; }
    return i;
  };
#ifdef NEP_PROFILE_PHASES
  long long xt0_ = clock64(); int xv_ = 0;
#endif
  for (int i = next_cand(); i < c.N; i = next_cand()) {
#ifdef NEP_PROFILE_PHASES
    xv_++;
#endif
    if (i == c.own) continue;
    const int fk = !c.f_bits ? -1 : (index > c.num_pol ? 0 : (int)((c.f_bits[i] >> (j - 1)) & 1u));
    // (a staged record is read through an LDS-typed pointer — ds_read: through a generic pointer the same reads are FLAT loads, which
    // take the vector-memory path even when the address is LDS; the agent loop is two dependent round trips of them per visit)
    const int lo = rec_of(i, itv);      // byte offset of the record's LDS copy, or -1: not staged
    if (lo >= 0) {
      typedef __attribute__((address_space(3))) const double* lcd; typedef __attribute__((address_space(3))) const int* lci;
      const lcd r = (lcd)(unsigned)lo;
      if (!((lci)r)[0]) continue;
      const int nbv = ((lci)r)[1];
      const Ev2 sa{r[kEntPkHead + 2 * jl], r[kEntPkHead + 2 * jl + 1]}, sb{r[kEntPkHead + 2 * jr], r[kEntPkHead + 2 * jr + 1]}, b0{r[kEntPkBend], r[kEntPkBend + 1]}, b1{r[kEntPkBend + 2], r[kEntPkBend + 3]};
      ent_cross_agent(add, pk, pk1, sa, sb, pb_self, nbv, r + kEntPkBend, i + 1, fk, true, b0, b1);
      continue;
    }
    const double* r = ent_rec(c, i, itv);
    const int2 hd = *(const int2*)r;
    if (!hd.x) continue;
    const double2 sa = *(const double2*)(r + kEntPkHead + 2 * jl), sb = *(const double2*)(r + kEntPkHead + 2 * jr), b0 = *(const double2*)(r + kEntPkBend), b1 = *(const double2*)(r + kEntPkBend + 2);
    ent_cross_agent(add, pk, pk1, Ev2{sa.x, sa.y}, Ev2{sb.x, sb.y}, pb_self, hd.y, r + kEntPkBend, i + 1, fk, true, Ev2{b0.x, b0.y}, Ev2{b1.x, b1.y});
  }
#ifdef NEP_PROFILE_PHASES
  const long long xt1_ = clock64();
#endif
  ent_cross_static(add, pk, pk1, c);
#ifdef NEP_PROFILE_PHASES
  if (c.prof) { const long long xt2_ = clock64(); atomicAdd((unsigned long long*)c.prof + 0, (unsigned long long)(xt1_ - xt0_)); atomicAdd((unsigned long long*)c.prof + 1, (unsigned long long)(xt2_ - xt1_));
    atomicAdd((unsigned long long*)c.prof + 5, 1ull); atomicAdd((unsigned long long*)c.prof + 15, (unsigned long long)xv_); atomicAdd((unsigned long long*)c.prof + 6, (unsigned long long)(add.n > 0)); }
#endif
}
// the point of a child's polynomial at sampled step j (0: its start), as ent_propagate evaluates it
__device__ __forceinline__ Ev2 ent_step_point(const EntCtx& c, const double* cxo, const double* cyo, Ev2 end, int j) {
  if (j == 0) return Ev2{cxo[3], cyo[3]};
  if (j >= c.ns) return end;
  const double t = c.T_span * j / c.ns;
  const double t3 = t * t * t, t2 = t * t;
  return Ev2{((cxo[0] * t3 + cxo[1] * t2) + cxo[2] * t) + cxo[3] * 1.0, ((cyo[0] * t3 + cyo[1] * t2) + cyo[2] * t) + cyo[3] * 1.0};
}
// the list surgery of ent_propagate on lists made beforehand: hdr[j - 1] = the header word of step j's list, pool = the words' base
// (a generic pointer into LDS), gblocks = the child's blocks in global memory (lists the LDS pool had no room for).  Same return codes as ent_propagate.
template <class ST> __device__ int ent_propagate_pre(const EntCtx& c, ST* st, const unsigned* hdr, const unsigned* pool, const unsigned* gblocks, int add_lim, const double* cxo, const double* cyo, Ev2 end, double& arc, bool check_tether, int cap_mult) {
  const int ns = c.ns;
  const Ev2 pb_self = ent_pb(c, c.own);
  Ev2 pk{cxo[3], cyo[3]}, pk1 = pk;
  for (int j = 1; j <= ns; j++) {
    pk1 = ent_step_point(c, cxo, cyo, end, j);
    arc += ent_dist(pk1, pk);
    // (hdr and pool are LDS: read through LDS-typed pointers — ds_read instead of flat loads, see ent_cross_step)
    typedef __attribute__((address_space(3))) const unsigned* lcu;
    const unsigned h = ((lcu)(unsigned)(unsigned long long)hdr)[j - 1];
    if (h & kEntHdrOvf) return 3;
    EntAdd add; add.clear(); add.lim = add_lim;
    add.n = (int)((h >> 16) & 0xffu);
    const unsigned* w;
    if (h & kEntHdrGlobal) {
      w = gblocks + (j - 1) * kEntAddCap;      // (gblocks: this child's blocks, one per step)
      add.r0 = add.n > 0 ? w[0] : 0u; add.r1 = add.n > 1 ? w[1] : 0u; add.r2 = add.n > 2 ? w[2] : 0u; add.r3 = add.n > 3 ? w[3] : 0u;
    } else {
      w = pool + (h & 0xffffu);
      const lcu wl = (lcu)(unsigned)(unsigned long long)w;
      add.r0 = add.n > 0 ? wl[0] : 0u; add.r1 = add.n > 1 ? wl[1] : 0u; add.r2 = add.n > 2 ? wl[2] : 0u; add.r3 = add.n > 3 ? wl[3] : 0u;
    }
    add.rest = const_cast<unsigned*>(w) + EntAdd::reg;      // (read only from here on: the merge flags cancelled entries in `gone`)
    if (st->n_alpha + add.n > (c.N + c.S) * cap_mult) return 1;
    if (add.n > 0) {
      const unsigned long long sig_before = ent_sig_of(st);
      if (ent_merge(add, st, pk, pb_self, c)) return 2;
      for (int e = 0; e < add.n;) {      // (the before / after counts of the agents the step touched: see ent_propagate)
        const int id_ = add.id(e);
        int init = 0, rem = 0;
        do { rem += add.alive(e) ? 1 : 0; init++; e++; } while (e < add.n && add.id(e) == id_);
        if (id_ > c.N) continue;
        const int d = 2 * rem - init;
        if (d <= 0) continue;
        if (d >= 2) return 1;
        if (!((sig_before >> (id_ & 63)) & 1ull)) continue;
        if (ent_count(st->id, st->n_alpha, id_) >= 2) return 1;
      }
    }
    if (ent_update_bends(st, pk1, pb_self, c)) return 4;
    pk = pk1;
  }
  if (check_tether && ent_tether(st, pb_self, pk1, c) > c.cable) return 1;
  return 0;
}

// A search node's state while its child is being merged (front end, phase two): the crossing list's ids and cases and the bend
// indices in LDS (the list surgery scans and shifts them: a chain of dependent reads), the betas — written when an entry is
// added or re-anchored, read once per step — in the thread's global working record.  Same member names as nep_fe_ent_state: the
// surgery code is a template over the two.
typedef __attribute__((address_space(3))) short* ent_lds_short;
typedef __attribute__((address_space(3))) signed char* ent_lds_char;
struct EntLds { int n_alpha, n_bend; ent_lds_short id; ent_lds_char cs; double* beta; ent_lds_char bend; unsigned long long sig; int cap, bend_cap; };      // cap <= NEP_FE_ENT_CAP, bend_cap <= NEP_MAX_BEND: what the view accepts (the handle's fast-path limits)
template <> struct ent_tr<EntLds> {
  typedef signed char bend_t;
  static __device__ __forceinline__ int cap(const EntLds* st) { return st->cap; }
  static __device__ __forceinline__ int bend_cap(const EntLds* st) { return st->bend_cap; }
};
template <> __device__ __forceinline__ bool ent_sig_may_have<EntLds>(const EntLds* st, int id) { return (st->sig >> (id & 63)) & 1ull; }
template <> __device__ __forceinline__ void ent_sig_add<EntLds>(EntLds* st, int id) { st->sig |= 1ull << (id & 63); }
template <> __device__ __forceinline__ unsigned long long ent_sig_of<EntLds>(const EntLds* st) { return st->sig; }
constexpr int kEntLdsBytes = ((NEP_FE_ENT_CAP * 3 + NEP_MAX_BEND + 3) & ~3) | 4;      // per thread; an odd number of dwords, so that the threads' lists fall into different banks
// (false: the record is a big record's marker, or holds more than the view takes — the caller goes to the big form)
__device__ __forceinline__ bool ent_lds_load(EntLds& L, const nep_fe_ent_state* __restrict__ src, int N) {
  L.n_alpha = src->n_alpha; L.n_bend = src->n_bend;
  if (__builtin_expect((L.n_alpha < 0) | (L.n_alpha > L.cap) | (L.n_bend > L.bend_cap), 0)) return false;
  unsigned long long g = 0ull;
  for (int i = 0; i < L.n_alpha; i++) { const int id_ = src->id[i]; L.id[i] = (short)id_; L.cs[i] = src->cs[i]; g |= 1ull << (id_ & 63); if (id_ > N) L.beta[i] = src->beta[i]; }      // (betas of statics only: see ent_lazy_beta)
  L.sig = g;
  for (int i = 0; i < L.n_bend; i++) L.bend[i] = src->bend[i];
  return true;
}
__device__ __forceinline__ void ent_lds_store(nep_fe_ent_state* __restrict__ dst, const EntLds& L, int N) {
  dst->n_alpha = L.n_alpha; dst->n_bend = L.n_bend;
  for (int i = 0; i < L.n_alpha; i++) { dst->id[i] = L.id[i]; dst->cs[i] = L.cs[i]; }
  for (int i = 0; i < L.n_bend; i++) dst->bend[i] = L.bend[i];
  for (int i = 0; i < L.n_alpha; i++) dst->beta[i] = L.id[i] > N ? L.beta[i] : 0.0;
}

// ---- A state without a capacity of its own: the big record.  A child whose step adds more crossings than EntAdd holds, whose list
// outgrows NEP_FE_ENT_CAP or whose tether bends more than NEP_MAX_BEND times — and every child of such a node — is carried in a record
// of the launch's pool in global memory, sized by the reference's own bound: entanglesWithOtherAgents prunes a child whose list plus
// new crossings exceed num_agents + statics (kinodynamic_search.cpp:850-854), so no list — and no set of bend points, which are list
// entries — is ever longer than that.  The fixed record that stands for such a node (saved[], nodes[]) holds n_alpha = -(k + 1), k
// the pool index; a record is written once and never changed (its children copy it).  Layout of record k (rec_bytes apart):
//   [n_alpha, n_bend | beta[cap] | id[cap] (16 bit) | bend[cap] (16 bit) | cs[cap] | pad | add words[add_lim] | add cancel bits]
struct EntBig { int n_alpha, n_bend; short* id; signed char* cs; double* beta; short* bend; int cap; unsigned long long sig; };      // (sig: as EntLds'; betas of agent entries neither read nor written, as there)
template <> __device__ __forceinline__ bool ent_sig_may_have<EntBig>(const EntBig* st, int id) { return (st->sig >> (id & 63)) & 1ull; }
template <> __device__ __forceinline__ void ent_sig_add<EntBig>(EntBig* st, int id) { st->sig |= 1ull << (id & 63); }
template <> __device__ __forceinline__ unsigned long long ent_sig_of<EntBig>(const EntBig* st) { return st->sig; }
template <> struct ent_tr<EntBig> {
  typedef short bend_t;
  static __device__ __forceinline__ int cap(const EntBig* st) { return st->cap; }
  static __device__ __forceinline__ int bend_cap(const EntBig* st) { return st->cap; }
};
__device__ __forceinline__ EntBig ent_big_view(const EntBigPool& P, int k) {
  unsigned char* r = P.base + (size_t)k * P.rec_bytes;
  EntBig B; B.cap = P.cap; B.sig = ~0ull; B.n_alpha = ((const int*)r)[0]; B.n_bend = ((const int*)r)[1];
  B.beta = (double*)(r + 8); B.id = (short*)(r + 8 + 8 * P.cap); B.bend = (short*)(r + 8 + 10 * P.cap); B.cs = (signed char*)(r + 8 + 12 * P.cap);
  return B;
}
__device__ __forceinline__ EntAddBig::Store ent_big_add(const EntBigPool& P, int k) {
  unsigned char* r = P.base + (size_t)k * P.rec_bytes + ((8 + 13 * P.cap + 3) & ~3);
  return EntAddBig::Store{(unsigned*)r, (unsigned*)(r + 4 * P.add_lim), P.add_lim};
}
__device__ __forceinline__ void ent_big_close(const EntBigPool& P, int k, const EntBig& B) { int* h = (int*)(P.base + (size_t)k * P.rec_bytes); h[0] = B.n_alpha; h[1] = B.n_bend; }
// One child in the big form (ent_big_child, at the end of this header; used by frontend_kernel<true, 1, true> — the instantiation that
// re-runs the searches the plain one lists — and by ent_check_kernel): claim a record, copy the parent's state into it (fixed or big),
// propagate.  rc = ent_propagate's verdict (0 / 1), or 5 = the pool is exhausted (pruned and flagged: nep_fe_result.ent_overflow,
// NEP_FLAG_ENT_POOL).  It is kept out of the search every slot runs: compiled into it — inlined as a pass of its own behind a flag, or
// as a called function — it cost that search's fixed-record path a third of its speed (geom_kernels.hip, frontend_kernel).
struct EntBigOut { int rc, k, n_alpha, n_bend; unsigned iz; double arc; };

template <class ST> __device__ unsigned ent_iz(const ST* st) {
  unsigned iz = 0;
  for (int i = 0; i < st->n_alpha; i++) {
    unsigned base = (unsigned)st->id[i], ex = (unsigned)st->cs[i], r;
    if (ex == 0) r = 1; else if (base < 2) r = base;
    else { r = 1; for (unsigned term = base;; term = term * term) { if (ex % 2 != 0) r *= term; ex /= 2; if (ex == 0) break; } }
    iz += (unsigned)(i + 1) * r;
  }
  return iz;
}
template <class ST> __device__ bool ent_valid_endpoint(const ST* st, int N) {
  for (int a = 0; a < st->n_alpha; a++) if (st->id[a] <= N && ent_count(st->id, st->n_alpha, st->id[a]) > 1) return false;
  return true;
}
__device__ void ent_copy(nep_fe_ent_state* dst, const nep_fe_ent_state* src) {
  const long* s = (const long*)src; long* d = (long*)dst;
#pragma unroll
  for (int i = 0; i < (int)(sizeof(nep_fe_ent_state) / 8); i++) d[i] = s[i];
}

// the same from / into a big record (the big-record instantiation keeps lists of up to kEntBigLdsCap entries in LDS: a list in a
// big record's global memory makes every step of the surgery a round trip of its own — a search that runs on big records took
// 3.4 ms against 0.8 for one that does not)
constexpr int kEntBigLdsCap = 120, kEntBigLdsBend = 16;
constexpr int kEntBigLdsBytes = ((kEntBigLdsCap * 3 + kEntBigLdsBend + 3) & ~3) | 4;      // per thread; an odd number of dwords
__device__ __forceinline__ bool ent_lds_load_big(EntLds& L, const EntBig& Q, int N) {
  L.n_alpha = Q.n_alpha; L.n_bend = Q.n_bend;
  if ((L.n_alpha > L.cap) | (L.n_bend > L.bend_cap)) return false;
  unsigned long long g = 0ull;
  for (int i = 0; i < L.n_alpha; i++) { const int id_ = Q.id[i]; L.id[i] = (short)id_; L.cs[i] = Q.cs[i]; g |= 1ull << (id_ & 63); if (id_ > N) L.beta[i] = Q.beta[i]; }
  L.sig = g;
  for (int i = 0; i < L.n_bend; i++) L.bend[i] = (signed char)Q.bend[i];
  return true;
}
__device__ __forceinline__ void ent_lds_store_big(const EntBigPool& P, int k, const EntLds& L, int N) {
  EntBig B = ent_big_view(P, k);
  for (int i = 0; i < L.n_alpha; i++) { const int id_ = L.id[i]; B.id[i] = (short)id_; B.cs[i] = L.cs[i]; if (id_ > N) B.beta[i] = L.beta[i]; }
  for (int i = 0; i < L.n_bend; i++) B.bend[i] = L.bend[i];
  B.n_alpha = L.n_alpha; B.n_bend = L.n_bend;
  ent_big_close(P, k, B);
}
__device__ __forceinline__ EntBigOut ent_big_child(const EntCtx& c, const EntBigPool& P, const nep_fe_ent_state* par, const double* cxo, const double* cyo, Ev2 end, int index, bool check_tether, int cap_mult) {
  EntBigOut o; o.rc = 5; o.k = -1; o.n_alpha = 0; o.n_bend = 0; o.iz = 0u; o.arc = 0.0;
  if (!P.base) return o;
  const int k = atomicAdd(P.count, 1);
  if (k >= P.n_rec) return o;
  EntBig B = ent_big_view(P, k);
  if (par->n_alpha < 0) {
    const EntBig Q = ent_big_view(P, -par->n_alpha - 1);
    B.n_alpha = Q.n_alpha; B.n_bend = Q.n_bend;
    unsigned long long g = 0ull;
    for (int i = 0; i < Q.n_alpha; i++) { const int id_ = Q.id[i]; B.id[i] = (short)id_; B.cs[i] = Q.cs[i]; g |= 1ull << (id_ & 63); if (id_ > c.N) B.beta[i] = Q.beta[i]; }
    B.sig = g;
    for (int i = 0; i < Q.n_bend; i++) B.bend[i] = Q.bend[i];
  } else {
    B.n_alpha = par->n_alpha; B.n_bend = par->n_bend;
    unsigned long long g = 0ull;
    for (int i = 0; i < par->n_alpha; i++) { const int id_ = par->id[i]; B.id[i] = (short)id_; B.cs[i] = par->cs[i]; g |= 1ull << (id_ & 63); if (id_ > c.N) B.beta[i] = par->beta[i]; }
    B.sig = g;
    for (int i = 0; i < par->n_bend; i++) B.bend[i] = par->bend[i];
  }
  double arc = 0.0;
  o.rc = ent_propagate<EntAddBig>(c, &B, ent_big_add(P, k), cxo, cyo, end, index, arc, check_tether, cap_mult);
  ent_big_close(P, k, B);
  o.arc = arc; o.k = k; o.n_alpha = B.n_alpha; o.n_bend = B.n_bend; o.iz = ent_iz(&B);
  return o;
}

}  // namespace nep
