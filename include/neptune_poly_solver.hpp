// neptune_poly_solver.hpp — C++ host side of the drop-in, above the C ABI (neptune_backend.h).
//
// (1) neptune_amd::PolySolver: the public methods of the reference's `class PolySolverGurobi`
//     (reference neptune/include/solver_gurobi_poly.hpp:28-49) with the same names, argument
//     order, call sequence and failure behaviour, on std:: containers (no Eigen needed).
// (2) `class PolySolverGurobi` itself, with the reference's exact Eigen/mt::/eu:: signatures, is
//     compiled only inside the reference tree (when <Eigen/Dense> and mader_types.hpp are on the
//     include path); it forwards to (1).  With it, reference neptune/src/neptune.cpp:102-107 and
//     :1514-1527 compile unchanged and link against libneptune_backend.so instead of Gurobi/GLPK.
//
// Header-only; link with -lneptune_backend.
#pragma once
#include <array>
#include <cmath>
#include <stdexcept>
#include <string>
#include <vector>

#include "neptune_backend.h"

namespace neptune_amd {

using Vec2 = std::array<double, 2>;
using Vec4 = std::array<double, 4>;
using Polygon = std::vector<Vec2>;                 // mt::Polygon_Std (2 x V), vertex list
using HullsOfCurve = std::vector<Polygon>;         // mt::ConvexHullsOfCurve_Std2d   [interval]
using HullsOfCurves = std::vector<HullsOfCurve>;   // mt::ConvexHullsOfCurves_Std2d  [obstacle][interval]

struct PieceWisePol {                              // mt::PieceWisePol (mader_types.hpp:462-548)
  std::vector<double> times;
  std::vector<Vec4> coeff_x, coeff_y, coeff_z;
  void clear() { times.clear(); coeff_x.clear(); coeff_y.clear(); coeff_z.clear(); }
};
struct State {                                     // mt::state (mader_types.hpp:35-41)
  std::array<double, 3> pos{}, vel{}, accel{}, jerk{};
};
struct EntState {                                  // eu::ent_state (entangle_utils.hpp:23-29)
  std::vector<std::array<int, 2>> alphas;          // (agent_id, case_no)
  std::vector<double> betas;
  std::vector<int> bendPointsIdx;
  std::vector<int> active_cases;
};

class PolySolver {
 public:
  // PolySolverGurobi::PolySolverGurobi (solver_gurobi_poly.cpp:25-27)
  PolySolver(int num_pol, int deg_pol, int id, double T_span, const std::vector<Vec2>& pb, double weight_term,
             double rad_term, bool use_linear_constraints)
      : num_pol_(num_pol), T_span_(T_span), num_agents_((int)pb.size()) {
    std::vector<double> flat;
    for (auto& b : pb) { flat.push_back(b[0]); flat.push_back(b[1]); }
    nep_backend_cfg cfg{};
    cfg.num_pol = num_pol; cfg.deg_pol = deg_pol; cfg.id = id; cfg.num_agents = num_agents_;
    cfg.T_span = T_span; cfg.weight_term = weight_term; cfg.rad_term = rad_term;
    cfg.use_linear_constraints = use_linear_constraints ? 1 : 0; cfg.pb = flat.data();
    h_ = nep_backend_create(&cfg);
    // the reference aborts on an unsupported basis (solver_gurobi_poly.cpp:81-85); a constructor
    // cannot return a status, so misconfiguration throws here, before any replan
    if (!h_) throw std::runtime_error(std::string("nep_backend_create: ") + nep_last_error());
  }
  ~PolySolver() { nep_backend_destroy(h_); }
  PolySolver(const PolySolver&) = delete;
  PolySolver& operator=(const PolySolver&) = delete;

  void setMaxRuntime(double runtime) { check(nep_backend_set_max_runtime(h_, runtime)); }
  void setMaxValues(double x_min, double x_max, double y_min, double y_max, double z_min, double z_max, double v_max,
                    double a_max, double j_max) {
    check(nep_backend_set_max_values(h_, x_min, x_max, y_min, y_max, z_min, z_max, v_max, a_max, j_max));
  }
  void setTetherLength(double tetherLength) { check(nep_backend_set_tether_length(h_, tetherLength)); }
  void setStaticObstVert(const std::vector<Polygon>& convexHullOfStaticObs) {
    std::vector<int32_t> off; std::vector<double> xy;
    csr(convexHullOfStaticObs, off, xy);
    check(nep_backend_set_static_obst_vert(h_, (int32_t)convexHullOfStaticObs.size(), off.data(), xy.data()));
  }
  void setInitTrajectory(const PieceWisePol& pwp_init) {
    nep_pwp p{};
    p.n_seg = (int32_t)pwp_init.coeff_x.size();
    if (p.n_seg > NEP_TRAJ_MAX_SEG || (int)pwp_init.times.size() != p.n_seg + 1) throw std::invalid_argument("setInitTrajectory: bad sizes");
    for (int i = 0; i <= p.n_seg; i++) p.times[i] = pwp_init.times[i];
    for (int i = 0; i < p.n_seg; i++)
      for (int j = 0; j < 4; j++) { p.coeff[0][i][j] = pwp_init.coeff_x[i][j]; p.coeff[1][i][j] = pwp_init.coeff_y[i][j]; p.coeff[2][i][j] = pwp_init.coeff_z[i][j]; }
    check(nep_backend_set_init_trajectory(h_, &p));
  }
  void setHulls(const HullsOfCurves& hulls) {
    std::vector<Polygon> flat;
    for (auto& obs : hulls) for (int i = 0; i < num_pol_; i++) flat.push_back(i < (int)obs.size() ? obs[i] : Polygon());
    std::vector<int32_t> off; std::vector<double> xy;
    csr(flat, off, xy);
    check(nep_backend_set_hulls(h_, (int32_t)hulls.size(), off.data(), xy.data()));
  }
  void setHullsNoInflation(const HullsOfCurves& hulls) {
    std::vector<Polygon> flat;
    for (auto& obs : hulls) for (int i = 0; i < num_pol_; i++) flat.push_back(i < (int)obs.size() ? obs[i] : Polygon());
    std::vector<int32_t> off; std::vector<double> xy;
    csr(flat, off, xy);
    check(nep_backend_set_hulls_no_inflation(h_, (int32_t)hulls.size(), off.data(), xy.data()));
  }
  void setBetasVector(const std::vector<std::vector<std::array<double, 3>>>&) {}  // dead in the reference (:290-305)
  void setEntStateVector(const std::vector<EntState>& entStateVec, const std::vector<std::vector<Vec2>>& bendPtsForAgents) {
    std::vector<int32_t> aoff{0}, alphas, active, boff{0};
    int n_active = 0;
    for (auto& e : entStateVec) n_active = std::max(n_active, (int)e.active_cases.size());
    for (auto& e : entStateVec) {
      for (auto& a : e.alphas) { alphas.push_back(a[0]); alphas.push_back(a[1]); }
      aoff.push_back((int32_t)alphas.size() / 2);
      for (int j = 0; j < n_active; j++) active.push_back(j < (int)e.active_cases.size() ? e.active_cases[j] : 0);
    }
    std::vector<double> bxy;
    for (auto& b : bendPtsForAgents) { for (auto& pnt : b) { bxy.push_back(pnt[0]); bxy.push_back(pnt[1]); } boff.push_back((int32_t)bxy.size() / 2); }
    if (alphas.empty()) alphas.push_back(0);
    if (bxy.empty()) bxy.push_back(0.0);
    nep_ent_view v{};
    v.n_states = (int32_t)entStateVec.size(); v.n_active = n_active; v.alpha_off = aoff.data(); v.alphas = alphas.data();
    v.active_cases = active.data(); v.bend_off = boff.data(); v.bend_xy = bxy.data();
    check(nep_backend_set_ent_state_vector(h_, &v));
  }
  // optimize (:804-887): false <=> both solves failed; objective_value untouched in that case
  bool optimize(double& objective_value) {
    int st = nep_backend_optimize(h_, &objective_value);
    check(st);
    return st != NEP_FAILED;
  }
  // generatePwpOut (:889-936)
  void generatePwpOut(PieceWisePol& pwp_out, std::vector<State>& traj_out, double t_start, double dc) {
    nep_pwp p{};
    int cap = (int)std::ceil(num_pol_ * T_span_ / dc) + 3;
    std::vector<double> st((size_t)cap * NEP_STATE_DOUBLES);
    int32_t n = 0;
    check(nep_backend_generate_pwp_out(h_, t_start, dc, &p, st.data(), cap, &n));
    pwp_out.clear();
    for (int i = 0; i <= p.n_seg; i++) pwp_out.times.push_back(p.times[i]);
    for (int i = 0; i < p.n_seg; i++) {
      pwp_out.coeff_x.push_back({p.coeff[0][i][0], p.coeff[0][i][1], p.coeff[0][i][2], p.coeff[0][i][3]});
      pwp_out.coeff_y.push_back({p.coeff[1][i][0], p.coeff[1][i][1], p.coeff[1][i][2], p.coeff[1][i][3]});
      pwp_out.coeff_z.push_back({p.coeff[2][i][0], p.coeff[2][i][1], p.coeff[2][i][2], p.coeff[2][i][3]});
    }
    traj_out.clear();
    for (int s = 0; s < n; s++) {
      State x;
      for (int a = 0; a < 3; a++) { x.pos[a] = st[s * 12 + a]; x.vel[a] = st[s * 12 + 3 + a]; x.accel[a] = st[s * 12 + 6 + a]; x.jerk[a] = st[s * 12 + 9 + a]; }
      traj_out.push_back(x);
    }
  }
  nep_stats stats() const { nep_stats s{}; nep_backend_get_stats(h_, &s); return s; }
  nep_backend_t* handle() { return h_; }

 private:
  static void csr(const std::vector<Polygon>& polys, std::vector<int32_t>& off, std::vector<double>& xy) {
    off.assign(1, 0); xy.clear();
    for (auto& pl : polys) { for (auto& v : pl) { xy.push_back(v[0]); xy.push_back(v[1]); } off.push_back((int32_t)xy.size() / 2); }
    if (xy.empty()) xy.push_back(0.0);
  }
  // the reference lets solver exceptions (e.g. a Gurobi licence error) propagate and kill the
  // node; API misuse / HIP errors do the same here
  static void check(int rc) { if (rc < 0) throw std::runtime_error(std::string("neptune backend: ") + nep_last_error()); }
  nep_backend_t* h_ = nullptr;
  int num_pol_; double T_span_; int num_agents_;
};

}  // namespace neptune_amd

// -------------------------------------------------------------------------------------------------
// Exact-signature drop-in for the reference tree (needs Eigen + the reference's own type headers).
// -------------------------------------------------------------------------------------------------
#if defined(NEPTUNE_AMD_REFERENCE_SHIM)
#if !__has_include(<Eigen/Dense>)
#error "NEPTUNE_AMD_REFERENCE_SHIM needs Eigen (<Eigen/Dense>) and the reference's mader_types.hpp / entangle_utils.hpp on the include path"
#endif
#include <Eigen/Dense>
#include "entangle_utils.hpp"   // eu::ent_state
#include "mader_types.hpp"      // mt::PieceWisePol, mt::state, mt::Polygon_Std, ...

class PolySolverGurobi {
 public:
  PolySolverGurobi(int num_pol, int deg_pol, int id, double T_span, std::vector<Eigen::Vector2d> pb, double weight_term,
                   double rad_term, bool use_linear_constraints)
      : s_(num_pol, deg_pol, id, T_span, to_vec2(pb), weight_term, rad_term, use_linear_constraints) {}
  bool optimize(double& objective_value) { return s_.optimize(objective_value); }
  void setMaxRuntime(double runtime) { s_.setMaxRuntime(runtime); }
  void setMaxValues(double x_min, double x_max, double y_min, double y_max, double z_min, double z_max, double v_max,
                    double a_max, double j_max) { s_.setMaxValues(x_min, x_max, y_min, y_max, z_min, z_max, v_max, a_max, j_max); }
  void setInitTrajectory(mt::PieceWisePol pwp_init) {
    neptune_amd::PieceWisePol p; p.times = pwp_init.times;
    for (size_t i = 0; i < pwp_init.coeff_x.size(); i++) {
      p.coeff_x.push_back({pwp_init.coeff_x[i](0), pwp_init.coeff_x[i](1), pwp_init.coeff_x[i](2), pwp_init.coeff_x[i](3)});
      p.coeff_y.push_back({pwp_init.coeff_y[i](0), pwp_init.coeff_y[i](1), pwp_init.coeff_y[i](2), pwp_init.coeff_y[i](3)});
      p.coeff_z.push_back({pwp_init.coeff_z[i](0), pwp_init.coeff_z[i](1), pwp_init.coeff_z[i](2), pwp_init.coeff_z[i](3)});
    }
    s_.setInitTrajectory(p);
  }
  void setHulls(mt::ConvexHullsOfCurves_Std2d& hulls) { s_.setHulls(to_hulls(hulls)); }
  void setHullsNoInflation(mt::ConvexHullsOfCurves_Std2d& hulls) { s_.setHullsNoInflation(to_hulls(hulls)); }
  void setBetasVector(std::vector<std::vector<Eigen::Vector3d>>&) {}
  void setTetherLength(double tetherLength) { s_.setTetherLength(tetherLength); }
  void setStaticObstVert(std::vector<mt::Polygon_Std>& convexHullOfStaticObs) {
    std::vector<neptune_amd::Polygon> v; for (auto& m : convexHullOfStaticObs) v.push_back(to_poly(m)); s_.setStaticObstVert(v);
  }
  void setEntStateVector(std::vector<eu::ent_state>& entStateVec, std::vector<std::vector<Eigen::Vector2d>>& bendPtsForAgents) {
    std::vector<neptune_amd::EntState> e;
    for (auto& x : entStateVec) { neptune_amd::EntState y; for (auto& a : x.alphas) y.alphas.push_back({a(0), a(1)}); y.betas = x.betas; y.bendPointsIdx = x.bendPointsIdx; y.active_cases = x.active_cases; e.push_back(y); }
    std::vector<std::vector<neptune_amd::Vec2>> b; for (auto& v : bendPtsForAgents) b.push_back(to_vec2(v));
    s_.setEntStateVector(e, b);
  }
  void generatePwpOut(mt::PieceWisePol& pwp_out, std::vector<mt::state>& traj_out, double t_start, double dc) {
    neptune_amd::PieceWisePol p; std::vector<neptune_amd::State> st;
    s_.generatePwpOut(p, st, t_start, dc);
    pwp_out.clear(); pwp_out.times = p.times;
    for (size_t i = 0; i < p.coeff_x.size(); i++) {
      pwp_out.coeff_x.push_back(Eigen::Vector4d(p.coeff_x[i][0], p.coeff_x[i][1], p.coeff_x[i][2], p.coeff_x[i][3]));
      pwp_out.coeff_y.push_back(Eigen::Vector4d(p.coeff_y[i][0], p.coeff_y[i][1], p.coeff_y[i][2], p.coeff_y[i][3]));
      pwp_out.coeff_z.push_back(Eigen::Vector4d(p.coeff_z[i][0], p.coeff_z[i][1], p.coeff_z[i][2], p.coeff_z[i][3]));
    }
    traj_out.clear();
    for (auto& x : st) { mt::state s; s.setPos(x.pos[0], x.pos[1], x.pos[2]); s.setVel(x.vel[0], x.vel[1], x.vel[2]); s.setAccel(x.accel[0], x.accel[1], x.accel[2]); s.setJerk(x.jerk[0], x.jerk[1], x.jerk[2]); traj_out.push_back(s); }
  }

 private:
  static std::vector<neptune_amd::Vec2> to_vec2(const std::vector<Eigen::Vector2d>& v) { std::vector<neptune_amd::Vec2> o; for (auto& x : v) o.push_back({x(0), x(1)}); return o; }
  static neptune_amd::Polygon to_poly(const mt::Polygon_Std& m) { neptune_amd::Polygon o; for (int c = 0; c < m.cols(); c++) o.push_back({m(0, c), m(1, c)}); return o; }
  static neptune_amd::HullsOfCurves to_hulls(const mt::ConvexHullsOfCurves_Std2d& h) {
    neptune_amd::HullsOfCurves o; for (auto& obs : h) { neptune_amd::HullsOfCurve c; for (auto& m : obs) c.push_back(to_poly(m)); o.push_back(c); } return o;
  }
  neptune_amd::PolySolver s_;
};
#endif
