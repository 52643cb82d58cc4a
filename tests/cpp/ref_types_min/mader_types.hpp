// Stand-in for the reference's mader_types.hpp: ONLY the names include/neptune_poly_solver.hpp's exact-signature class touches,
// with the member names and types the reference gives them (neptune/include/mader_types.hpp:21-30, 35-124, 462-548).  See README.md.
#pragma once
#include <vector>

#include <Eigen/Dense>

namespace mt {
typedef Eigen::Matrix<double, 2, Eigen::Dynamic> Polygon_Std;                // one polygon: 2 x V vertices
typedef std::vector<Polygon_Std> ConvexHullsOfCurve_Std2d;                    // per planning interval
typedef std::vector<ConvexHullsOfCurve_Std2d> ConvexHullsOfCurves_Std2d;      // per obstacle

struct state {
  Eigen::Vector3d pos, vel, accel, jerk;
  void setPos(const double x, const double y, const double z) { pos = Eigen::Vector3d(x, y, z); }
  void setVel(const double x, const double y, const double z) { vel = Eigen::Vector3d(x, y, z); }
  void setAccel(const double x, const double y, const double z) { accel = Eigen::Vector3d(x, y, z); }
  void setJerk(const double x, const double y, const double z) { jerk = Eigen::Vector3d(x, y, z); }
};

struct PieceWisePol {
  std::vector<double> times;                                                   // n + 1 knots
  std::vector<Eigen::Matrix<double, 4, 1>> coeff_x, coeff_y, coeff_z;          // [a b c d] per interval
  void clear() { times.clear(); coeff_x.clear(); coeff_y.clear(); coeff_z.clear(); }
};
}  // namespace mt
