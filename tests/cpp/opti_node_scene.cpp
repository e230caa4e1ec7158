// opti_node_scene.cpp — drives the C++ shim exactly the way the reference's
// only built executable drives GradTrajOptimizer (src/opti_node.cpp:58-106 of
// EpicOne1/grad_traj_optimization): 40x40x5 m map @0.2, two walls of obstacle
// points, 11 waypoints, one optimizeTrajectory(OPT_SECOND_STEP).
// Prints one JSON object; tests/test_cpp_shim.py checks it against the oracle.
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "grad_traj_optimizer.hpp"

using namespace gtop_amd;

static void print_vec(const char *name, const std::vector<double> &v, bool comma = true) {
  std::printf("\"%s\": [", name);
  for (size_t i = 0; i < v.size(); ++i) std::printf("%s%.17g", i ? ", " : "", v[i]);
  std::printf("]%s\n", comma ? "," : "");
}

int main(int argc, char **argv) {
  GradTrajOptimizer::Config cfg;   // launch/opti_node.launch:3-28 defaults
  cfg.max_evals = argc > 1 ? std::atoi(argv[1]) : 60;
  cfg.time_limit_2 = 5.0;          // evaluation-capped so the run is reproducible
  cfg.optimize_on_device = argc > 2 ? std::atoi(argv[2]) : 0;   // 1: the whole optimisation in one launch
  GradTrajOptimizer grad_traj_opt(cfg);
  if (!grad_traj_opt.ok()) {
    std::fprintf(stderr, "GradTrajOptimizer: %s\n", grad_traj_opt.lastError());
    return 2;
  }
  grad_traj_opt.initSDFMap(Vec3{40, 40, 5}, Vec3{-40 / 2, -40 / 2, 0.0}, 0.2);

  std::vector<Vec3> obss;
  for (double x = 0.05; x <= 3.0; x += 0.2)
    for (double y = 2.05; y <= 2.7; y += 0.2)
      for (double z = 0.05; z <= 5.0; z += 0.2) obss.push_back(Vec3{x, y, z});
  for (double x = 0.05; x >= -3.0; x -= 0.2)
    for (double y = -2.05; y >= -2.7; y -= 0.2)
      for (double z = 0.05; z <= 5.0; z += 0.2) obss.push_back(Vec3{x, y, z});
  grad_traj_opt.updateSDFMap(obss);

  std::vector<Vec3> init_path = {{0, -5, 2}, {1, -4, 2}, {1, -3, 2}, {1, -2, 2}, {1, -1, 2}, {0, 0, 2},
                                 {-1, 1, 2}, {-1, 2, 2}, {-1, 3, 2}, {-1, 4, 2}, {0, 5, 2}};
  grad_traj_opt.setPath(init_path);
  if (!grad_traj_opt.ok()) {
    std::fprintf(stderr, "setup: %s\n", grad_traj_opt.lastError());
    return 3;
  }

  // the callback at the initial point, through the reference's costFunc signature
  std::vector<double> x0 = grad_traj_opt.freeDerivatives(), g0;
  const double c0 = GradTrajOptimizer::costFunc(x0, g0, &grad_traj_opt);

  Matrix coeff0, coeff;
  std::vector<double> time_sgm;
  grad_traj_opt.getCoefficient(coeff0);
  const auto t_opt0 = std::chrono::steady_clock::now();
  grad_traj_opt.optimizeTrajectory(OPT_SECOND_STEP);
  const double opt_seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_opt0).count();
  grad_traj_opt.getCoefficient(coeff);
  grad_traj_opt.getSegmentTime(time_sgm);

  std::vector<double> x1 = grad_traj_opt.freeDerivatives(), g1;
  const double c1 = GradTrajOptimizer::costFunc(x1, g1, &grad_traj_opt);
  std::vector<double> curve_c, curve_t;
  grad_traj_opt.getCostCurve(curve_c, curve_t);

  std::printf("{\n\"n_obstacle_points\": %zu,\n\"evals\": %d,\n\"cost0\": %.17g,\n\"cost1\": %.17g,\n\"optimize_seconds\": %.6g,\n",
              obss.size(), grad_traj_opt.iterations(), c0, c1, opt_seconds);
  print_vec("x0", x0);
  print_vec("grad0", g0);
  print_vec("x1", x1);
  print_vec("grad1", g1);
  print_vec("segment_time", time_sgm);
  print_vec("coeff0", coeff0.a);
  print_vec("coeff1", coeff.a);
  print_vec("cost_curve", curve_c, false);
  std::printf("}\n");
  return 0;
}
