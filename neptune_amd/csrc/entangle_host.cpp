// entangle_host.cpp — tether entanglement-state propagation (include/neptune_entangle.h).
// Host memory only; branchy list surgery that stays on the CPU in the reference as well.
//
// Follows (behaviour, not code): neptune/src/entangle_utils.cpp:16-28 (wedge), :1129-1228 and
// :1231-1277 (crossings to add), :1402-1534 + :1608-1647 (cancellation / append), :1536-1604 (bend
// points), :1649-1743 (bend point lookup, beta, tether length); neptune/src/kinodynamic_search.cpp:
// 105-128 (sample times), :707-895 (per-node update), :582-603 (state chain of a path);
// neptune/src/neptune.cpp:500-565 (sampling the other agents).
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../../include/neptune_entangle.h"

namespace {

struct V2 { double x, y; };
struct A2 { int id, cs; };                       // (agent or static id, case)
inline bool same(const A2& a, const A2& b) { return a.id == b.id && a.cs == b.cs; }

struct State {                                   // eu::ent_state
  std::vector<A2> alphas;
  std::vector<double> betas;
  std::vector<int> bend;                         // bendPointsIdx
  std::vector<int> active;                       // active_cases
};

struct Ctx {
  const nep_ent_cfg* c;
  const nep_ent_inputs* in;
  int N, S;
  V2 pb(int j) const { return V2{c->pb[2 * j], c->pb[2 * j + 1]}; }
  V2 srep(int s, int col) const { return V2{c->static_rep[(s * 2 + col) * 2], c->static_rep[(s * 2 + col) * 2 + 1]}; }
  int nbend(int j) const { return in->bend_off[j + 1] - in->bend_off[j]; }
  V2 bendpt(int j, int k) const { const double* p = in->bend_xy + 2 * (in->bend_off[j] + k); return V2{p[0], p[1]}; }
  V2 sampled(int i, int interval, int col) const {
    const double* p = in->sampled + (((size_t)i * c->num_pol + interval) * (c->num_samples + 1) + col) * 2;
    return V2{p[0], p[1]};
  }
};

// > 0: c is anti-clockwise of b seen from a
inline double wedge(const V2& a, const V2& b, const V2& c) { return (b.x - a.x) * (c.y - a.y) - (c.x - a.x) * (b.y - a.y); }
inline double wedge(const V2& a, const V2& b, const V2& c, V2& ab, V2& ac) {
  ab = V2{b.x - a.x, b.y - a.y};
  ac = V2{c.x - a.x, c.y - a.y};
  return ab.x * ac.y - ac.x * ab.y;
}
// where along (bend -> other end) the crossing falls: the ratio of the larger components
inline double cross_ratio(const V2& u, const V2& v) {
  return std::fabs(u.y * v.y) > std::fabs(u.x * v.x) ? u.y / v.y : u.x / v.x;
}
inline double dist(const V2& a, const V2& b) { return std::sqrt((a.x - b.x) * (a.x - b.x) + (a.y - b.y) * (a.y - b.y)); }

// crossings of the step pk -> pk1 with agent i's tether polyline base .. bend points .. agent
void crossings_agent(std::vector<A2>& add, const V2& pk, const V2& pk1, const V2& pik, const V2& pik1, const V2& pb_self,
                     const Ctx& cx, int i, int agent_id) {
  const int nb = cx.nbend(i);
  bool base_addition = false;
  for (int k = 0; k < nb; k++) {
    const bool last = k == nb - 1;
    const V2 bk = cx.bendpt(i, k);
    V2 u, v;                                       // (far end - pk), (bend k - pk)
    double c1, c2;
    if (!last) { const V2 bn = cx.bendpt(i, k + 1); c1 = wedge(pk, bn, bk, u, v); c2 = wedge(pk1, bn, bk); }
    else { c1 = wedge(pk, pik, bk, u, v); c2 = wedge(pk1, pik1, bk); }
    if (last) {  // the other agent's last tether piece sweeping over OUR base
      V2 ub, vb;
      const double f1 = wedge(pb_self, pik, bk, ub, vb);
      const double f2 = wedge(pb_self, pik1, bk);
      if (f1 * f2 < 0) {
        const double a = cross_ratio(ub, vb);
        if (a < 0) { /* between agent i and its bend point: nothing recorded */ }
        else if (a < 1) add.push_back(A2{agent_id, 1});
        else if (k == 0) add.push_back(A2{agent_id, 0});
        base_addition = true;
      }
    }
    if (c1 * c2 < 0) {
      const double a = cross_ratio(u, v);
      if (a < 0) add.push_back(A2{agent_id, k + 2});
      else if (a < 1 && last) add.push_back(A2{agent_id, 1});
      else if (a >= 1 && k == 0) add.push_back(A2{agent_id, 0});
    }
  }
  if (base_addition && add.size() >= 2 && same(add[add.size() - 1], add[add.size() - 2])) add.resize(add.size() - 2);
}

void crossings_static(std::vector<A2>& add, const V2& pk, const V2& pk1, const Ctx& cx) {
  for (int s = 0; s < cx.S; s++) {
    const V2 pik = cx.srep(s, 1), pbi = cx.srep(s, 0);
    V2 u, v;
    const double c1 = wedge(pk, pik, pbi, u, v), c2 = wedge(pk1, pik, pbi);
    if (c1 * c2 < 0) {
      const double a = cross_ratio(u, v);
      if (a < 0) { /* not recorded for static obstacles */ }
      else if (a < 1) add.push_back(A2{cx.N + s + 1, 1});
      else add.push_back(A2{cx.N + s + 1, 0});
    }
  }
}

V2 anchor_of(const A2& bend_id, const Ctx& cx) {
  if (bend_id.id <= cx.N) return cx.pb(bend_id.id - 1);               // only the base of a mobile agent can be a bend
  return cx.srep(bend_id.id - cx.N - 1, bend_id.cs);
}
V2 current_bend_point(const State& st, const V2& pb_self, const Ctx& cx) {
  if (st.bend.empty()) return pb_self;
  const A2 b = st.alphas[st.bend.back()];
  if (b.id <= cx.N && b.id >= 1) return cx.pb(b.id - 1);
  if (b.id > cx.N) return cx.srep(b.id - cx.N - 1, b.cs);
  return V2{0, 0};
}
double beta_for(const A2& a, const V2& pk, const V2& bp, const Ctx& cx) {
  if (a.id <= cx.N) return 0.0;
  return wedge(pk, cx.srep(a.id - cx.N - 1, a.cs), bp);
}

// may the scan for a cancelling partner pass over list entry j?  true = stop here
bool scan_stops(const A2& to_add, const A2& in_list, int j, int last_bend, const Ctx& cx) {
  if (to_add.id <= cx.N && to_add.cs >= 2) return j <= last_bend;
  if (to_add.id <= cx.N) return false;
  return in_list.id > cx.N || j <= last_bend;
}

void merge_crossings(std::vector<A2>& add, State& st, const V2& pk, const V2& pb_self, const Ctx& cx) {
  bool again = true;
  while (again) {
    again = false;
    const int b = st.bend.empty() ? -1 : st.bend.back();
    for (size_t i = 0; i < add.size() && !again; i++) {
      const A2 t = add[i];
      for (int j = (int)st.alphas.size() - 1; j >= 0; j--) {
        const A2 l = st.alphas[j];
        const bool agent = t.id <= cx.N;
        const bool match =
            same(l, t) ||
            (agent && l.id == t.id && (size_t)t.cs >= (size_t)cx.nbend(t.id - 1) + 1 && t.cs < l.cs) ||
            (agent && l.id == t.id && l.cs >= 2 && t.cs >= 2 && std::abs(t.cs - l.cs) == 1 && j > b);
        if (match) {
          st.active[t.id - 1] -= 1;
          add.erase(add.begin() + i);
          st.alphas.erase(st.alphas.begin() + j);
          st.betas.erase(st.betas.begin() + j);
          if (j == b) {
            st.bend.pop_back();
            const V2 bp = current_bend_point(st, pb_self, cx);
            for (size_t k = j; k < st.alphas.size(); k++) st.betas[k] = beta_for(st.alphas[k], pk, bp, cx);
          } else if (j < b) {
            st.bend.back() = b - 1;
            for (int k = (int)st.bend.size() - 2; k >= 0; k--) {
              if (st.bend[k] > j) st.bend[k] -= 1; else break;
            }
          }
          again = true;
          break;
        }
        if (scan_stops(t, l, j, b, cx)) break;
      }
    }
  }
  if (add.empty()) return;
  const V2 bp = current_bend_point(st, pb_self, cx);
  for (const A2& t : add) {
    st.alphas.push_back(t);
    st.active[t.id - 1] += 1;
    st.betas.push_back(beta_for(t, pk, bp, cx));
  }
}

void update_bend_points(State& st, const V2& pk1, const V2& pb_self, const Ctx& cx) {
  const V2 bp = current_bend_point(st, pb_self, cx);
  int idx_new = -1;
  const int start = st.bend.empty() ? -1 : st.bend.back();
  for (int i = start + 1; i < (int)st.alphas.size(); i++) {
    const double beta = beta_for(st.alphas[i], pk1, bp, cx);
    if (beta * st.betas[i] < -1e-7) idx_new = i;
  }
  if (idx_new > -1) {
    st.bend.push_back(idx_new);
    const V2 nb = anchor_of(st.alphas[idx_new], cx);
    for (int i = idx_new + 1; i < (int)st.alphas.size(); i++) st.betas[i] = beta_for(st.alphas[i], pk1, nb, cx);
    return;
  }
  while (!st.bend.empty()) {
    const V2 prev = st.bend.size() == 1 ? pb_self : anchor_of(st.alphas[st.bend[st.bend.size() - 2]], cx);
    const int bi = st.bend.back();
    const double beta = beta_for(st.alphas[bi], pk1, prev, cx);
    if (beta * st.betas[bi] > 1e-7) {
      for (int k = bi + 1; k < (int)st.alphas.size(); k++) st.betas[k] = beta_for(st.alphas[k], pk1, prev, cx);
      st.bend.pop_back();
    } else break;
  }
}

double tether_length(const State& st, V2 from, const V2& pk1, const Ctx& cx) {
  double len = 0.0;
  for (int bi : st.bend) {
    const A2 b = st.alphas[bi];
    V2 bp; double comp;
    if (b.id <= cx.N) { bp = cx.pb(b.id - 1); comp = 0.0; }
    else { bp = cx.srep(b.id - cx.N - 1, b.cs); comp = cx.c->static_longest[(b.id - cx.N - 1) * 2 + b.cs]; }
    len += dist(bp, from) + 2 * comp;
    from = bp;
  }
  return len + dist(pk1, from);
}

// KinodynamicSearch::entanglesWithOtherAgents for one segment
bool propagate_segment(const Ctx& cx, State& st, const double cxo[4], const double cyo[4], const V2& end, int index, double& arc) {
  const nep_ent_cfg& c = *cx.c;
  const int ns = c.num_samples;
  const V2 pb_self = cx.pb(c.id - 1);
  V2 pk{cxo[3], cyo[3]}, pk1 = pk;
  std::vector<int> old = st.active;
  for (int j = 1; j <= ns; j++) {
    std::vector<A2> add;
    if (j < ns) {
      const double t = c.T_span * j / ns;
      const double t3 = t * t * t, t2 = t * t;
      // Eigen's 4-term row * column product accumulates left to right
      pk1 = V2{((cxo[0] * t3 + cxo[1] * t2) + cxo[2] * t) + cxo[3] * 1.0, ((cyo[0] * t3 + cyo[1] * t2) + cyo[2] * t) + cyo[3] * 1.0};
    } else pk1 = end;
    arc += dist(pk1, pk);
    for (int i = 0; i < cx.N; i++) {
      if (i == c.id - 1) continue;
      if (!cx.in->present[i]) continue;
      V2 pik, pik1;
      if (index > c.num_pol) { pik = cx.sampled(i, c.num_pol - 1, ns); pik1 = pik; }
      else { pik = cx.sampled(i, index - 1, j - 1); pik1 = cx.sampled(i, index - 1, j); }
      crossings_agent(add, pk, pk1, pik, pik1, pb_self, cx, i, i + 1);
    }
    crossings_static(add, pk, pk1, cx);
    if ((int)(st.alphas.size() + add.size()) > cx.N + cx.S) return true;
    merge_crossings(add, st, pk, pb_self, cx);
    for (int i = 0; i < cx.N; i++) {
      if (old[i] < 2 && st.active[i] >= 2) return true;
      if (old[i] >= 2 && st.active[i] > old[i]) return true;
    }
    update_bend_points(st, pk1, pb_self, cx);
    old = st.active;
    pk = pk1;
  }
  return tether_length(st, pb_self, pk1, cx) > c.cable_length;
}

bool load_state(const nep_ent_state* s, State& st) {
  if (!s || s->n_alpha < 0 || s->n_bend < 0 || s->n_alpha > s->cap || s->n_bend > s->cap || s->n_active < 0) return false;
  if ((s->n_alpha && (!s->alphas || !s->betas)) || (s->n_bend && !s->bend_idx) || (s->n_active && !s->active_cases)) return false;
  st.alphas.resize(s->n_alpha); st.betas.resize(s->n_alpha); st.bend.resize(s->n_bend); st.active.resize(s->n_active);
  for (int i = 0; i < s->n_alpha; i++) { st.alphas[i] = A2{s->alphas[2 * i], s->alphas[2 * i + 1]}; st.betas[i] = s->betas[i]; }
  for (int i = 0; i < s->n_bend; i++) st.bend[i] = s->bend_idx[i];
  for (int i = 0; i < s->n_active; i++) st.active[i] = s->active_cases[i];
  return true;
}
int store_state(const State& st, nep_ent_state* s) {
  if ((int)st.alphas.size() > s->cap || (int)st.bend.size() > s->cap || (int)st.active.size() != s->n_active) return NEP_E_CAP;
  s->n_alpha = (int)st.alphas.size(); s->n_bend = (int)st.bend.size();
  for (int i = 0; i < s->n_alpha; i++) { s->alphas[2 * i] = st.alphas[i].id; s->alphas[2 * i + 1] = st.alphas[i].cs; s->betas[i] = st.betas[i]; }
  for (int i = 0; i < s->n_bend; i++) s->bend_idx[i] = st.bend[i];
  for (int i = 0; i < s->n_active; i++) s->active_cases[i] = st.active[i];
  return NEP_OK;
}
bool cfg_ok(const nep_ent_cfg* c, const nep_ent_inputs* in) {
  if (!c || !in || !c->pb || c->num_agents < 1 || c->id < 1 || c->id > c->num_agents || c->num_pol < 1 || c->num_samples < 1) return false;
  if (c->n_static < 0 || (c->n_static && (!c->static_rep || !c->static_longest))) return false;
  return in->sampled && in->present && in->bend_off && (in->bend_xy || in->bend_off[c->num_agents] == 0);
}

}  // namespace

extern "C" {

int nep_ent_sample_points(const nep_pwp* traj, double t_start, double t_end, int32_t num_pol, int32_t num_samples, double* out) {
  if (!traj || !out || num_pol < 1 || num_samples < 1 || traj->n_seg < 1 || traj->n_seg > NEP_TRAJ_MAX_SEG) return NEP_E_ARG;
  const int n = traj->n_seg;
  const double deltaT = (t_end - t_start) / (1.0 * num_pol);
  for (int i = 0; i < num_pol; i++) {
    for (int j = 0; j <= num_samples; j++) {
      const double ts = t_start + deltaT * i + deltaT / num_samples * j;
      int low = 0;                                 // std::upper_bound: first knot > ts
      while (low <= n && !(traj->times[low] > ts)) low++;
      int seg; double te;
      if (low <= n) {
        seg = low - 1;
        if (seg < 0) seg = 0; else if (seg > n - 1) seg = n - 1;
        te = ts - traj->times[seg];
        if (te < 0) te = 0; else if (te > deltaT) te = deltaT;
      } else {                                     // past the last knot: end of the last interval
        seg = n - 1;
        te = traj->times[n] - traj->times[n - 1];
      }
      const double t3 = te * te * te, t2 = te * te;
      double* o = out + ((size_t)i * (num_samples + 1) + j) * 2;
      for (int ax = 0; ax < 2; ax++) { const double* c = traj->coeff[ax][seg]; o[ax] = ((c[0] * t3 + c[1] * t2) + c[2] * te) + c[3] * 1.0; }
    }
  }
  return NEP_OK;
}

int nep_ent_propagate_segment(const nep_ent_cfg* cfg, const nep_ent_inputs* in, nep_ent_state* state, const double coeff_x[4],
                              const double coeff_y[4], const double end_xy[2], int32_t index, double* arc_length) {
  if (!cfg_ok(cfg, in) || !coeff_x || !coeff_y || !end_xy || index < 1) return NEP_E_ARG;
  State st;
  if (!load_state(state, st) || state->n_active < cfg->num_agents + cfg->n_static) return NEP_E_ARG;
  Ctx cx{cfg, in, cfg->num_agents, cfg->n_static};
  double arc = arc_length ? *arc_length : 0.0;
  const bool hit = propagate_segment(cx, st, coeff_x, coeff_y, V2{end_xy[0], end_xy[1]}, index, arc);
  if (arc_length) *arc_length = arc;
  if (int e = store_state(st, state)) return e;
  return hit ? 1 : 0;
}

int nep_ent_propagate_guess(const nep_ent_cfg* cfg, const nep_ent_inputs* in, const nep_ent_state* init, const nep_guess* guess,
                            int32_t alpha_cap, int32_t* alpha_off, int32_t* alphas, int32_t* active_cases, int32_t* entangled_at,
                            nep_ent_state* final_state) {
  if (!cfg_ok(cfg, in) || !guess || !alpha_off || !alphas || !active_cases || guess->K < 1 || guess->K > NEP_MAX_POL) return NEP_E_ARG;
  State st;
  if (!load_state(init, st) || init->n_active < cfg->num_agents + cfg->n_static) return NEP_E_ARG;
  Ctx cx{cfg, in, cfg->num_agents, cfg->n_static};
  const int K = guess->K, na = init->n_active;
  const double T = cfg->T_span;
  int off = 0, hit_at = 0;
  auto emit = [&](int knot) -> bool {
    alpha_off[knot] = off;
    if (off + (int)st.alphas.size() > alpha_cap) return false;
    for (const A2& a : st.alphas) { alphas[2 * off] = a.id; alphas[2 * off + 1] = a.cs; off++; }
    for (int i = 0; i < na; i++) active_cases[(size_t)knot * na + i] = st.active[i];
    return true;
  };
  if (!emit(0)) return NEP_E_CAP;
  for (int s = 1; s <= K; s++) {
    if (!hit_at) {
      const double* cxo = guess->coeff[0][s - 1]; const double* cyo = guess->coeff[1][s - 1];
      // the node's end state: the polynomial at T (kinodynamic_search.cpp:1079-1086 integrates the same cubic)
      V2 end;
      if (s < K) end = V2{guess->coeff[0][s][3], guess->coeff[1][s][3]};
      else end = V2{((cxo[0] * (T * T * T) + cxo[1] * (T * T)) + cxo[2] * T) + cxo[3], ((cyo[0] * (T * T * T) + cyo[1] * (T * T)) + cyo[2] * T) + cyo[3]};
      State next = st;
      double arc = 0.0;
      if (propagate_segment(cx, next, cxo, cyo, end, s, arc)) hit_at = s; else st = next;
    }
    if (!emit(s)) return NEP_E_CAP;
  }
  alpha_off[K + 1] = off;
  if (entangled_at) *entangled_at = hit_at;
  if (final_state) { if (int e = store_state(st, final_state)) return e; }
  return NEP_OK;
}

int nep_ent_case_ids(int32_t n_states, int32_t n_active, const int32_t* alpha_off, const int32_t* alphas, const int32_t* active_cases,
                     int32_t num_agents, int32_t* case_id) {
  if (n_states < 0 || !alpha_off || !active_cases || !case_id || num_agents < 1 || n_active < num_agents) return NEP_E_ARG;
  std::memset(case_id, 0, sizeof(int32_t) * NEP_MAX_POL * (size_t)num_agents);
  for (int i = 0; i < n_states && i < NEP_MAX_POL; i++)
    for (int j = 0; j < num_agents; j++) {
      if (active_cases[(size_t)i * n_active + j] != 1) continue;
      int cid = 0;                                  // the last matching alpha wins (solver_gurobi_poly.cpp:626-630)
      for (int a = alpha_off[i]; a < alpha_off[i + 1]; a++) if (alphas[2 * a] == j + 1) cid = alphas[2 * a + 1];
      case_id[(size_t)i * num_agents + j] = cid;
    }
  return NEP_OK;
}

}  // extern "C"
