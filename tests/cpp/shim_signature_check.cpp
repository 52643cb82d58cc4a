// Compiles, links and (on a GPU box) runs include/neptune_poly_solver.hpp's exact-signature `class PolySolverGurobi` against the
// stand-in type declarations of tests/cpp/ref_types_min/ (NOT Eigen, NOT the reference's headers: a syntax + signature check,
// see the README there).  The signatures asserted below are the reference's (neptune/include/solver_gurobi_poly.hpp:28-49); the
// calls are made in the order Neptune makes them (neptune/src/neptune.cpp:102-107, 663, 1514-1527).
#define NEPTUNE_AMD_REFERENCE_SHIM 1
#include <cstdio>
#include <type_traits>

#include "neptune_poly_solver.hpp"

template <typename Want, typename Got> constexpr bool same(Got) { return std::is_same<Want, Got>::value; }
static_assert(std::is_constructible<PolySolverGurobi, int, int, int, double, std::vector<Eigen::Vector2d>, double, double, bool>::value, "constructor");
static_assert(same<bool (PolySolverGurobi::*)(double&)>(&PolySolverGurobi::optimize), "optimize");
static_assert(same<void (PolySolverGurobi::*)(double)>(&PolySolverGurobi::setMaxRuntime), "setMaxRuntime");
static_assert(same<void (PolySolverGurobi::*)(double, double, double, double, double, double, double, double, double)>(&PolySolverGurobi::setMaxValues), "setMaxValues");
static_assert(same<void (PolySolverGurobi::*)(mt::PieceWisePol)>(&PolySolverGurobi::setInitTrajectory), "setInitTrajectory");
static_assert(same<void (PolySolverGurobi::*)(mt::ConvexHullsOfCurves_Std2d&)>(&PolySolverGurobi::setHulls), "setHulls");
static_assert(same<void (PolySolverGurobi::*)(mt::ConvexHullsOfCurves_Std2d&)>(&PolySolverGurobi::setHullsNoInflation), "setHullsNoInflation");
static_assert(same<void (PolySolverGurobi::*)(std::vector<std::vector<Eigen::Vector3d>>&)>(&PolySolverGurobi::setBetasVector), "setBetasVector");
static_assert(same<void (PolySolverGurobi::*)(double)>(&PolySolverGurobi::setTetherLength), "setTetherLength");
static_assert(same<void (PolySolverGurobi::*)(std::vector<mt::Polygon_Std>&)>(&PolySolverGurobi::setStaticObstVert), "setStaticObstVert");
static_assert(same<void (PolySolverGurobi::*)(std::vector<eu::ent_state>&, std::vector<std::vector<Eigen::Vector2d>>&)>(&PolySolverGurobi::setEntStateVector), "setEntStateVector");
static_assert(same<void (PolySolverGurobi::*)(mt::PieceWisePol&, std::vector<mt::state>&, double, double)>(&PolySolverGurobi::generatePwpOut), "generatePwpOut");

static mt::Polygon_Std square(double cx, double cy, double half) {
  mt::Polygon_Std m(2, 4);
  const double dx[4] = {-1, 1, 1, -1}, dy[4] = {-1, -1, 1, 1};
  for (int v = 0; v < 4; v++) { m(0, v) = cx + half * dx[v]; m(1, v) = cy + half * dy[v]; }
  return m;
}

int main() {
  const int K = 4; const double T = 0.5;
  try {
    std::vector<Eigen::Vector2d> pb = {Eigen::Vector2d(-8.0, 0.0), Eigen::Vector2d(8.0, 0.0)};
    PolySolverGurobi s(8, 3, 1, T, pb, 1000.0, 0.5, true);                       // neptune.cpp:102-103
    s.setMaxValues(-12, 12, -12, 12, -0.2, 5.1, 2.0, 3.0, 5.0);                  // :104-105
    s.setMaxRuntime(0.05); s.setTetherLength(30.0);                              // :106-107
    std::vector<mt::Polygon_Std> statics = {square(0.0, 6.0, 1.65)};
    s.setStaticObstVert(statics);                                                // :663
    mt::PieceWisePol init;                                                       // a straight line flown at 1 m/s, at rest at neither end
    for (int i = 0; i <= K; i++) init.times.push_back(i * T);
    for (int i = 0; i < K; i++) {
      init.coeff_x.push_back(Eigen::Vector4d(0, 0, 1.0, -6.0 + i * T)); init.coeff_y.push_back(Eigen::Vector4d(0, 0, 0, 0.0));
      init.coeff_z.push_back(Eigen::Vector4d(0, 0, 0, 1.0));
    }
    mt::ConvexHullsOfCurves_Std2d hulls(1), hulls0(2);
    for (int i = 0; i < 8; i++) { hulls[0].push_back(square(4.0, -5.0, 1.2)); hulls0[1].push_back(square(4.0, -5.0, 0.05)); }
    std::vector<eu::ent_state> ent(K + 1); for (auto& e : ent) e.active_cases.assign(3, 0);
    std::vector<std::vector<Eigen::Vector2d>> bends = {{pb[0]}, {pb[1]}};
    std::vector<std::vector<Eigen::Vector3d>> betas;
    s.setInitTrajectory(init); s.setHulls(hulls); s.setHullsNoInflation(hulls0); s.setBetasVector(betas); s.setEntStateVector(ent, bends);   // :1514-1518
    double objective = -1.0;
    const bool ok = s.optimize(objective);                                       // :1519
    mt::PieceWisePol out; std::vector<mt::state> traj;
    s.generatePwpOut(out, traj, 3.0, 0.05);                                      // :1527
    std::printf("optimize -> %d objective %.6f segments %zu states %zu t0 %.2f x(0) %.3f\n", ok ? 1 : 0, objective, out.coeff_x.size(), traj.size(),
                out.times.empty() ? -1.0 : out.times[0], traj.empty() ? 0.0 : traj[0].pos(0));
    return (ok && out.coeff_x.size() == (size_t)K && traj.size() >= (size_t)(K * T / 0.05) && traj.size() <= (size_t)(K * T / 0.05) + 1 && out.times[0] == 3.0) ? 0 : 2;
  } catch (const std::exception& e) {
    std::printf("no solve: %s\n", e.what());                                     // (a box without a GPU: the back end has no CPU path)
    return 3;
  }
}
