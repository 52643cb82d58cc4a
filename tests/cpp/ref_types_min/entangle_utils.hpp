// Stand-in for the reference's entangle_utils.hpp: eu::ent_state only (neptune/include/entangle_utils.hpp:23-29).  See README.md.
#pragma once
#include <vector>

#include <Eigen/Dense>

namespace eu {
struct ent_state {
  std::vector<Eigen::Vector2i> alphas;      // (agent_id, case_no)
  std::vector<double> betas;
  std::vector<int> bendPointsIdx;
  std::vector<int> active_cases;
};
}  // namespace eu
