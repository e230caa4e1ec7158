// scene_runner.cpp — drives the C++ shim (csrc/grad_traj_optimizer.hpp) through the public methods a planner
// node calls, in the order such a node calls them: initSDFMap, updateSDFMap, setPath (or setKinoPath), the cost
// function at the start point, optimizeTrajectory(OPT_SECOND_STEP), getCoefficient / getSegmentTime / getCostCurve.
// The scene itself is DATA: a text file written by tests/scenes.py (map, obstacle points, waypoints, optionally the
// waypoint velocities / accelerations / segment times of a kinodynamic front end).  Prints one JSON object that
// tests/test_cpp_shim.py and tests/test_config0.py check against the oracle.
//
//   gtop_scene_runner <scene.txt> [max_evals = 60] [optimize_on_device = 0]
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <string>
#include <vector>

#include "grad_traj_optimizer.hpp"

using namespace gtop_amd;

namespace {

struct Scene {
  Vec3 map_size{}, origin{};
  double resolution = 0.0;
  std::vector<Vec3> obstacles, waypoints, vel, acc;
  std::vector<double> seg_time;   // with vel/acc: the kinodynamic form (setKinoPath)
};

bool read_points(std::istream &in, std::vector<Vec3> &out) {
  size_t count = 0;
  if (!(in >> count)) return false;
  out.resize(count);
  for (Vec3 &p : out)
    if (!(in >> p[0] >> p[1] >> p[2])) return false;
  return true;
}

bool read_scene(const char *file, Scene &sc) {
  std::ifstream in(file);
  std::string key;
  while (in >> key) {
    bool good = true;
    if (key == "map_size") good = bool(in >> sc.map_size[0] >> sc.map_size[1] >> sc.map_size[2]);
    else if (key == "origin") good = bool(in >> sc.origin[0] >> sc.origin[1] >> sc.origin[2]);
    else if (key == "resolution") good = bool(in >> sc.resolution);
    else if (key == "obstacles") good = read_points(in, sc.obstacles);
    else if (key == "waypoints") good = read_points(in, sc.waypoints);
    else if (key == "velocities") good = read_points(in, sc.vel);
    else if (key == "accelerations") good = read_points(in, sc.acc);
    else if (key == "segment_times") {
      size_t count = 0;
      good = bool(in >> count);
      sc.seg_time.resize(good ? count : 0);
      for (double &t : sc.seg_time) good = good && bool(in >> t);
    } else good = false;
    if (!good) {
      std::fprintf(stderr, "scene file: bad entry '%s'\n", key.c_str());
      return false;
    }
  }
  return sc.resolution > 0.0 && sc.waypoints.size() >= 3;
}

Matrix rows_of(const std::vector<Vec3> &pts) {
  Matrix mat((int)pts.size(), 3);
  for (int i = 0; i < mat.rows; ++i)
    for (int a = 0; a < 3; ++a) mat(i, a) = pts[i][a];
  return mat;
}

void print_vec(const char *name, const std::vector<double> &v, bool comma = true) {
  std::printf("\"%s\": [", name);
  for (size_t i = 0; i < v.size(); ++i) std::printf("%s%.17g", i ? ", " : "", v[i]);
  std::printf("]%s\n", comma ? "," : "");
}

}  // namespace

int main(int argc, char **argv) {
  if (argc < 2) {
    std::fprintf(stderr, "usage: %s <scene.txt> [max_evals] [optimize_on_device]\n", argv[0]);
    return 1;
  }
  Scene sc;
  if (!read_scene(argv[1], sc)) return 1;

  GradTrajOptimizer::Config cfg;   // defaults = launch/opti_node.launch:3-28 of the reference
  cfg.max_evals = argc > 2 ? std::atoi(argv[2]) : 60;
  cfg.time_limit_2 = 5.0;          // evaluation-capped so the run is reproducible
  cfg.optimize_on_device = argc > 3 ? std::atoi(argv[3]) : 0;   // 1: the whole optimisation in one launch
  GradTrajOptimizer planner(cfg);
  if (!planner.ok()) {
    std::fprintf(stderr, "GradTrajOptimizer: %s\n", planner.lastError());
    return 2;
  }
  planner.initSDFMap(sc.map_size, sc.origin, sc.resolution);
  planner.updateSDFMap(sc.obstacles);
  const bool kino = !sc.seg_time.empty();
  if (kino) planner.setKinoPath(rows_of(sc.waypoints), rows_of(sc.vel), rows_of(sc.acc), sc.seg_time);
  else planner.setPath(sc.waypoints);
  if (!planner.ok()) {
    std::fprintf(stderr, "setup: %s\n", planner.lastError());
    return 3;
  }

  // the callback at the start point, through the reference's costFunc signature
  std::vector<double> x_start = planner.freeDerivatives(), grad_start;
  const double cost_start = GradTrajOptimizer::costFunc(x_start, grad_start, &planner);

  Matrix coeff_start, coeff_end;
  std::vector<double> seg_time;
  planner.getCoefficient(coeff_start);
  const auto clock0 = std::chrono::steady_clock::now();
  planner.optimizeTrajectory(OPT_SECOND_STEP);
  const double opt_seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - clock0).count();
  planner.getCoefficient(coeff_end);
  planner.getSegmentTime(seg_time);

  std::vector<double> x_end = planner.freeDerivatives(), grad_end;
  const double cost_end = GradTrajOptimizer::costFunc(x_end, grad_end, &planner);
  std::vector<double> curve_cost, curve_time;
  planner.getCostCurve(curve_cost, curve_time);

  std::printf("{\n\"n_obstacle_points\": %zu,\n\"evals\": %d,\n\"cost0\": %.17g,\n\"cost1\": %.17g,\n\"optimize_seconds\": %.6g,\n",
              sc.obstacles.size(), planner.iterations(), cost_start, cost_end, opt_seconds);
  print_vec("x0", x_start);
  print_vec("grad0", grad_start);
  print_vec("x1", x_end);
  print_vec("grad1", grad_end);
  print_vec("segment_time", seg_time);
  print_vec("coeff0", coeff_start.a);
  print_vec("coeff1", coeff_end.a);
  print_vec("cost_curve", curve_cost, false);
  std::printf("}\n");
  return 0;
}
