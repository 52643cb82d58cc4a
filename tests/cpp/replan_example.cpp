// The reference's call sequence (neptune/src/neptune.cpp:102-107, 1514-1527) on the C++ host class.
// Reads one problem from stdin (text), prints the result; tests/test_gpu_parity.py compares it
// with the oracle.  Input: K T weight  x_min x_max y_min y_max z_min z_max v_max a_max
//                          then 3*K*4 coefficients (axis-major), then n_lines and n_lines*(seg n1 n2 d).
#include <cstdio>
#include <iostream>

#include "neptune_poly_solver.hpp"
#include "neptune_backend_debug.h"      // (the test hook below: lines as input)

int main() {
  int K; double T, w, b[8];
  std::cin >> K >> T >> w;
  for (double& x : b) std::cin >> x;
  neptune_amd::PieceWisePol init;
  for (int i = 0; i <= K; i++) init.times.push_back(i * T);
  std::vector<neptune_amd::Vec4>* co[3] = {&init.coeff_x, &init.coeff_y, &init.coeff_z};
  for (int ax = 0; ax < 3; ax++) for (int i = 0; i < K; i++) { neptune_amd::Vec4 v; for (double& x : v) std::cin >> x; co[ax]->push_back(v); }
  int nl; std::cin >> nl;
  std::vector<int32_t> seg(nl); std::vector<double> nd(3 * nl);
  for (int l = 0; l < nl; l++) std::cin >> seg[l] >> nd[3 * l] >> nd[3 * l + 1] >> nd[3 * l + 2];

  std::vector<neptune_amd::Vec2> pb = {{1e6, 1e6}};
  neptune_amd::PolySolver solver(8, 3, 1, T, pb, w, 0.5, true);                 // neptune.cpp:102-103
  solver.setMaxValues(b[0], b[1], b[2], b[3], b[4], b[5], b[6], b[7], 5.0);     // :104-105
  solver.setMaxRuntime(0.05);                                                    // :106
  solver.setTetherLength(40.0);                                                  // :107
  std::vector<neptune_amd::Polygon> statics;
  solver.setStaticObstVert(statics);                                             // :663
  solver.setInitTrajectory(init);                                                // :1514
  neptune_amd::HullsOfCurves hulls, hulls0(1);
  solver.setHulls(hulls);                                                        // :1515
  solver.setHullsNoInflation(hulls0);                                            // :1516
  nep_backend_debug_set_lines(solver.handle(), nl, seg.data(), nd.data());      // test hook: lines as input
  double objective = -12345.0;
  bool ok = solver.optimize(objective);                                          // :1519
  neptune_amd::PieceWisePol out; std::vector<neptune_amd::State> traj;
  solver.generatePwpOut(out, traj, 3.25, 0.05);                                  // :1527
  std::printf("%d %.17g %zu %.17g\n", ok ? 1 : 0, objective, traj.size(), out.times[0]);
  for (auto* c : {&out.coeff_x, &out.coeff_y, &out.coeff_z}) for (auto& v : *c) std::printf("%.17g %.17g %.17g %.17g\n", v[0], v[1], v[2], v[3]);
  return 0;
}
