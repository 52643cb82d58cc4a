// plan_host.cpp — ROS-free host side of the replan loop (include/neptune_plan.h): trajectory
// composition, the DynTraj wire format and the committed-plan deque.  Host memory only.
//
// Follows (behaviour, not code): neptune/src/utils.cpp:180-261,318-402 (compose, msg <-> pwp),
// mader_msgs/msg/{DynTraj,PieceWisePolTraj,CoeffPoly3}.msg, neptune/src/neptune_ros.cpp:379-480
// (publish / receive), neptune/include/mader_types.hpp:674-738 (plan deque),
// neptune/src/neptune.cpp:860-891,1366-1425,1661-1720 (goal pop, point A, splice, deltaT).
#include <cmath>
#include <cstring>
#include <deque>
#include <new>

#include "../../include/neptune_plan.h"

namespace {

bool pwp_ok(const nep_pwp* p) { return p && p->n_seg >= 0 && p->n_seg <= NEP_TRAJ_MAX_SEG; }

// appends knot `tk` and the interval [previous knot, tk] = src interval `seg`
bool push_interval(nep_pwp* out, double tk, const nep_pwp* src, int seg) {
  if (out->n_seg >= NEP_TRAJ_MAX_SEG) return false;
  int k = out->n_seg++;
  out->times[k + 1] = tk;
  for (int ax = 0; ax < 3; ++ax) std::memcpy(out->coeff[ax][k], src->coeff[ax][seg], 4 * sizeof(double));
  return true;
}

// ---- little-endian ROS1 primitives ----------------------------------------------------------
struct Writer {
  uint8_t* p;
  size_t cap, n;
  bool count_only;
  void raw(const void* src, size_t len) {
    if (!count_only && n + len <= cap) std::memcpy(p + n, src, len);
    n += len;
  }
  void u32(uint32_t v) { raw(&v, 4); }
  void i32(int32_t v) { raw(&v, 4); }
  void u8(uint8_t v) { raw(&v, 1); }
  void f32(float v) { raw(&v, 4); }
  void f64(double v) { raw(&v, 8); }
};

struct Reader {
  const uint8_t* p;
  size_t len, n;
  bool bad;
  bool take(void* dst, size_t k) {
    if (bad || len - n < k) { bad = true; return false; }
    if (dst) std::memcpy(dst, p + n, k);
    n += k;
    return true;
  }
  uint32_t u32() { uint32_t v = 0; take(&v, 4); return v; }
  int32_t i32() { int32_t v = 0; take(&v, 4); return v; }
  uint8_t u8() { uint8_t v = 0; take(&v, 1); return v; }
  float f32() { float v = 0; take(&v, 4); return v; }
  double f64() { double v = 0; take(&v, 8); return v; }
  void skip(size_t k) { take(nullptr, k); }
};

int64_t write_dyntraj(const nep_traj_rec* r, const nep_wire_header* h, Writer& w) {
  if (!r || !pwp_ok(&r->pwp) || r->n_bend < 0 || r->n_bend > NEP_MAX_BEND) return NEP_E_ARG;
  const char* frame = (h && h->frame_id) ? h->frame_id : "";
  uint32_t flen = (uint32_t)std::strlen(frame);
  w.u32(h ? h->seq : 0u);
  w.u32(h ? h->stamp_sec : 0u);
  w.u32(h ? h->stamp_nsec : 0u);
  w.u32(flen);
  w.raw(frame, flen);
  w.u32(3);                                   // function: three empty strings
  for (int i = 0; i < 3; ++i) w.u32(0);
  w.u32(3);                                   // bbox: float32[3]
  for (int i = 0; i < 3; ++i) w.f32((float)r->bbox[i]);
  for (int i = 0; i < 3; ++i) w.f64(r->pos[i]);
  w.i32(r->id);
  w.u8(r->is_agent ? 1 : 0);
  w.u32((uint32_t)r->n_bend);
  for (int i = 0; i < r->n_bend; ++i) { w.f64(r->bend[i][0]); w.f64(r->bend[i][1]); w.f64(0.0); }
  int n = r->pwp.n_seg;
  w.u32(n > 0 ? (uint32_t)(n + 1) : 0u);      // an empty pwp has no knots (default-constructed)
  for (int i = 0; n > 0 && i <= n; ++i) w.f64(r->pwp.times[i]);
  for (int ax = 0; ax < 3; ++ax) {
    w.u32((uint32_t)n);
    for (int i = 0; i < n; ++i)
      for (int c = 0; c < 4; ++c) w.f64(r->pwp.coeff[ax][i][c]);
  }
  return (int64_t)w.n;
}

}  // namespace

extern "C" {

int nep_pwp_compose(double t, double /*dc*/, nep_pwp* p1, nep_pwp* p2, nep_pwp* out) {
  if (!pwp_ok(p1) || !pwp_ok(p2) || !out || p1->n_seg < 1 || p2->n_seg < 1) return NEP_E_ARG;
  const int n1 = p1->n_seg, n2 = p2->n_seg;
  // the three in-place knot adjustments of the reference, in its order
  if (t > p1->times[n1] && t < p2->times[0]) p2->times[0] = t;
  if (p1->times[n1] < p2->times[0]) p2->times[0] = p1->times[n1];
  if (t < p1->times[0]) p1->times[0] = t;

  if (std::fabs(t - p2->times[0]) < 1e-5) {
    if (out != p2) *out = *p2;
    return NEP_OK;
  }
  nep_pwp res;
  std::memset(&res, 0, sizeof(res));
  if (p1->times[n1] < p2->times[0] || t > p2->times[n2] || t < p1->times[0]) {
    *out = res;                               // the reference's empty "dummy"
    return NEP_OK;
  }
  res.times[0] = t;
  bool ok = true;
  for (int i = 0; i <= n1 && ok; ++i)         // i >= 1 whenever the test holds: times[0] <= t here
    if (p1->times[i] > t && p1->times[i] < p2->times[0]) ok = push_interval(&res, p1->times[i], p1, i - 1);
  for (int i = 0; i <= n2 && ok; ++i) {
    if (!(p2->times[i] > t)) continue;
    ok = (i == 0) ? push_interval(&res, p2->times[0], p1, n1 - 1)
                  : push_interval(&res, p2->times[i], p2, i - 1);
  }
  if (!ok) return NEP_E_CAP;
  *out = res;
  return NEP_OK;
}

// Composition that describes the flown path exactly (an extension; see neptune_plan.h): every interval
// carries the coefficients of the source interval that actually covers it, re-based to its own knot.
static void rebase(const double c[4], double s, double o[4]) {      // q(w) = p(w + s)
  o[0] = c[0];
  o[1] = 3 * c[0] * s + c[1];
  o[2] = (3 * c[0] * s + 2 * c[1]) * s + c[2];
  o[3] = ((c[0] * s + c[1]) * s + c[2]) * s + c[3];
}
// p restricted to [t0, t1] appended to res (t0 < t1); beyond p's last knot the end point is held
static bool append_span(nep_pwp* res, const nep_pwp* p, double t0, double t1) {
  const int n = p->n_seg;
  double a = t0;
  while (a < t1) {
    int k = 0;
    while (k < n && p->times[k + 1] <= a) k++;
    if (res->n_seg >= NEP_TRAJ_MAX_SEG) return false;
    const int o = res->n_seg++;
    double b;
    if (k >= n) {                                   // past the end: hold the final point
      b = t1;
      const double T = p->times[n] - p->times[n - 1];
      for (int ax = 0; ax < 3; ax++) {
        double e[4]; rebase(p->coeff[ax][n - 1], T, e);
        res->coeff[ax][o][0] = res->coeff[ax][o][1] = res->coeff[ax][o][2] = 0.0; res->coeff[ax][o][3] = e[3];
      }
    } else {
      b = p->times[k + 1] < t1 ? p->times[k + 1] : t1;
      const double s = a - p->times[k] > 0 ? a - p->times[k] : 0.0;    // (a before p's first knot: p's start is extended backwards)
      for (int ax = 0; ax < 3; ax++) rebase(p->coeff[ax][k], a - p->times[k] < 0 ? a - p->times[k] : s, res->coeff[ax][o]);
    }
    res->times[o + 1] = b;
    a = b;
  }
  return true;
}

int nep_pwp_compose_exact(double t, const nep_pwp* p1, const nep_pwp* p2, nep_pwp* out) {
  if (!pwp_ok(p1) || !pwp_ok(p2) || !out || p1->n_seg < 1 || p2->n_seg < 1) return NEP_E_ARG;
  nep_pwp res;
  std::memset(&res, 0, sizeof(res));
  const double t2 = p2->times[0];
  if (t < t2) {                                       // the old trajectory until the new one takes over
    res.times[0] = t;
    if (!append_span(&res, p1, t, t2)) return NEP_E_CAP;
    for (int i = 0; i < p2->n_seg; i++) {
      if (res.n_seg >= NEP_TRAJ_MAX_SEG) return NEP_E_CAP;
      const int o = res.n_seg++;
      res.times[o + 1] = p2->times[i + 1];
      for (int ax = 0; ax < 3; ax++) std::memcpy(res.coeff[ax][o], p2->coeff[ax][i], 4 * sizeof(double));
    }
  } else {                                            // the new trajectory has already started: its part from t on
    res.times[0] = t;
    if (!append_span(&res, p2, t, p2->times[p2->n_seg] > t ? p2->times[p2->n_seg] : t + 1.0)) return NEP_E_CAP;
  }
  *out = res;
  return NEP_OK;
}

int64_t nep_dyntraj_wire_size(const nep_traj_rec* rec, const nep_wire_header* hdr) {
  Writer w{nullptr, 0, 0, true};
  return write_dyntraj(rec, hdr, w);
}

int64_t nep_dyntraj_encode(const nep_traj_rec* rec, const nep_wire_header* hdr, uint8_t* buf, size_t cap) {
  if (!buf) return NEP_E_ARG;
  Writer w{buf, cap, 0, false};
  int64_t n = write_dyntraj(rec, hdr, w);
  if (n < 0) return n;
  return (size_t)n <= cap ? n : (int64_t)NEP_E_CAP;
}

int64_t nep_dyntraj_decode(const uint8_t* buf, size_t len, nep_traj_rec* rec, nep_wire_header* hdr_out) {
  if (!buf || !rec) return NEP_E_ARG;
  Reader r{buf, len, 0, false};
  nep_traj_rec out;
  std::memset(&out, 0, sizeof(out));
  uint32_t seq = r.u32(), sec = r.u32(), nsec = r.u32();
  r.skip(r.u32());                            // frame_id
  uint32_t nfun = r.u32();
  for (uint32_t i = 0; i < nfun && !r.bad; ++i) r.skip(r.u32());
  uint32_t nb = r.u32();
  for (uint32_t i = 0; i < nb && !r.bad; ++i) {
    float v = r.f32();
    if (i < 3) out.bbox[i] = (double)v;
  }
  for (int i = 0; i < 3; ++i) out.pos[i] = r.f64();
  out.id = r.i32();
  out.is_agent = r.u8() ? 1 : 0;
  uint32_t nbend = r.u32();
  if (r.bad) return NEP_E_ARG;
  if (nbend > (uint32_t)NEP_MAX_BEND) return NEP_E_CAP;
  out.n_bend = (int32_t)nbend;
  for (uint32_t i = 0; i < nbend; ++i) { out.bend[i][0] = r.f64(); out.bend[i][1] = r.f64(); r.f64(); }
  uint32_t nt = r.u32();
  if (r.bad) return NEP_E_ARG;
  if (nt > (uint32_t)NEP_TRAJ_MAX_SEG + 1) return NEP_E_CAP;
  for (uint32_t i = 0; i < nt; ++i) out.pwp.times[i] = r.f64();
  uint32_t ncoef[3] = {0, 0, 0};
  for (int ax = 0; ax < 3; ++ax) {
    ncoef[ax] = r.u32();
    if (r.bad) return NEP_E_ARG;
    if (ncoef[ax] > (uint32_t)NEP_TRAJ_MAX_SEG) return NEP_E_CAP;
    for (uint32_t i = 0; i < ncoef[ax]; ++i)
      for (int c = 0; c < 4; ++c) out.pwp.coeff[ax][i][c] = r.f64();
  }
  if (r.bad) return NEP_E_ARG;
  if (ncoef[0] != ncoef[1] || ncoef[0] != ncoef[2]) return NEP_E_ARG;   // reference aborts here
  if (ncoef[0] > 0 && nt != ncoef[0] + 1) return NEP_E_ARG;
  out.pwp.n_seg = (int32_t)ncoef[0];
  out.valid = 1;
  *rec = out;
  if (hdr_out) {
    hdr_out->seq = seq;
    hdr_out->stamp_sec = sec;
    hdr_out->stamp_nsec = nsec;
    hdr_out->_pad = 0;
    hdr_out->frame_id = nullptr;
  }
  return (int64_t)r.n;
}

}  // extern "C"

// ---------------------------------------------------------------------------------------------
// committed plan
// ---------------------------------------------------------------------------------------------
struct PlanState { double v[12]; };

struct nep_plan {
  nep_plan_cfg cfg;
  std::deque<PlanState> q;
  int deltaT;
};

namespace {
// mu::saturate(int&, const int, const int): the call sites pass doubles, which C++ truncates
// to int because deltaT_ is an int lvalue (utils.cpp:744-754, neptune.cpp:1374).
void saturate_int(int& v, double lo, double hi) {
  int ilo = (int)lo, ihi = (int)hi;
  if (v < ilo) v = ilo;
  else if (v > ihi) v = ihi;
}
void saturate_dbl(double& v, double lo, double hi) {
  if (v < lo) v = lo;
  else if (v > hi) v = hi;
}
}  // namespace

extern "C" {

nep_plan_t* nep_plan_create(const nep_plan_cfg* cfg) {
  if (!cfg || !(cfg->dc > 0.0)) return nullptr;
  nep_plan* p = new (std::nothrow) nep_plan;
  if (!p) return nullptr;
  p->cfg = *cfg;
  p->deltaT = cfg->deltaT0;
  return p;
}

void nep_plan_destroy(nep_plan_t* p) { delete p; }

int nep_plan_reset(nep_plan_t* p, const double state[12]) {
  if (!p || !state) return NEP_E_ARG;
  p->q.clear();
  PlanState s;
  std::memcpy(s.v, state, sizeof(s.v));
  p->q.push_back(s);
  return NEP_OK;
}

int32_t nep_plan_size(const nep_plan_t* p) { return p ? (int32_t)p->q.size() : NEP_E_ARG; }

int nep_plan_get(const nep_plan_t* p, int32_t i, double state_out[12]) {
  if (!p || !state_out || i < 0 || i >= (int32_t)p->q.size()) return NEP_E_ARG;
  std::memcpy(state_out, p->q[(size_t)i].v, sizeof(PlanState));
  return NEP_OK;
}

int nep_plan_next_goal(nep_plan_t* p, double goal_out[12], int32_t* last_point) {
  if (!p || !goal_out) return NEP_E_ARG;
  if (p->q.empty()) return NEP_E_STATE;
  std::memcpy(goal_out, p->q.front().v, sizeof(PlanState));
  int last = 1;
  if (p->q.size() > 1) {
    p->q.pop_front();
    last = 0;
  }
  if (last_point) *last_point = last;
  return NEP_OK;
}

int nep_plan_select_a(nep_plan_t* p, const double state_pos[3], double time_now, nep_point_a* out) {
  if (!p || !state_pos || !out) return NEP_E_ARG;
  if (p->q.empty()) return NEP_E_STATE;
  const nep_plan_cfg& c = p->cfg;
  const int size = (int)p->q.size();
  saturate_int(p->deltaT, c.lower_bound_runtime / c.dc, c.upper_bound_runtime / c.dc);
  int future_index = size - p->deltaT;
  int k_end = future_index > 0 ? future_index : 0;
  if ((double)size < std::ceil(c.T_span / c.dc)) k_end = 0;
  int k_index = size - 1 - k_end;
  std::memcpy(out->A, p->q[(size_t)k_index].v, sizeof(PlanState));
  if (future_index < 0)
    for (int i = 3; i < 9; ++i) out->A[i] = 0.0;
  const double* head = p->q.front().v;
  double dx = head[0] - state_pos[0], dy = head[1] - state_pos[1], dz = head[2] - state_pos[2];
  if (std::sqrt(dx * dx + dy * dy + dz * dz) > 1.0)
    for (int i = 0; i < 3; ++i) out->A[i] = state_pos[i];
  double rs = (k_end != 0) ? k_index * c.dc - c.runtime_opt : c.upper_bound_runtime;
  saturate_dbl(rs, c.lower_bound_runtime - c.runtime_opt, c.upper_bound_runtime - c.runtime_opt);
  out->k_index = k_index;
  out->k_index_end = k_end;
  out->runtime_search = rs;
  out->t_start = k_index * c.dc + time_now;
  return NEP_OK;
}

int nep_plan_splice(nep_plan_t* p, int32_t k_index_end, const double* traj_out, int32_t n_states) {
  if (!p || k_index_end < 0 || n_states < 0 || (n_states > 0 && !traj_out)) return NEP_E_ARG;
  int size = (int)p->q.size();
  if (size - 1 - k_index_end < 0) return NEP_E_STATE;
  p->q.erase(p->q.end() - k_index_end - 1, p->q.end());
  for (int i = 0; i < n_states; ++i) {
    PlanState s;
    std::memcpy(s.v, traj_out + 12 * (size_t)i, sizeof(s.v));
    p->q.push_back(s);
  }
  return NEP_OK;
}

int nep_plan_update_delta(nep_plan_t* p, double elapsed_ms) {
  if (!p) return NEP_E_ARG;
  int states_last_replan = (int)std::ceil(elapsed_ms / (p->cfg.dc * 1000.0));
  double d = p->cfg.factor_alpha * states_last_replan;
  p->deltaT = (int)(d > 1.0 ? d : 1.0);
  return NEP_OK;
}

int32_t nep_plan_delta(const nep_plan_t* p) { return p ? p->deltaT : NEP_E_ARG; }

}  // extern "C"
