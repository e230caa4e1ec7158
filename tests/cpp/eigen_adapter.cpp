// eigen_adapter.cpp — the call sequence of the reference's planner node (src/opti_node.cpp:58-106: construct, initSDFMap,
// updateSDFMap, setPath, optimizeTrajectory(OPT_SECOND_STEP), getCoefficient, getSegmentTime) written against the
// GLOBAL `GradTrajOptimizer` with Eigen signatures of include/grad_traj_optimization/grad_traj_optimizer.h — the
// adapter a caller of the reference would compile against.  Built with tests/cpp/eigen_double/ on the include path (a
// test double of the Eigen types; the image has no Eigen), so what it checks is the adapter's own syntax and that it
// delegates: its output must equal gtop_scene_runner's on the same scene file bit for bit (tests/test_cpp_shim.py).
//
//   gtop_eigen_adapter <scene.txt> [max_evals = 60]
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <string>
#include <vector>

#include "grad_traj_optimization/grad_traj_optimizer.h"

using namespace std;   // (the reference's header says so too, grad_traj_optimizer.h:18)

static bool read_points(istream &in, vector<Eigen::Vector3d> &out) {
  size_t count = 0;
  if (!(in >> count)) return false;
  out.resize(count);
  for (auto &p : out)
    if (!(in >> p(0) >> p(1) >> p(2))) return false;
  return true;
}

static void print_vec(const char *name, const vector<double> &v, bool comma = true) {
  printf("\"%s\": [", name);
  for (size_t i = 0; i < v.size(); ++i) printf("%s%.17g", i ? ", " : "", v[i]);
  printf("]%s\n", comma ? "," : "");
}

int main(int argc, char **argv) {
  if (argc < 2) return 1;
  Eigen::Vector3d map_size, origin;
  double resolution = 0.0;
  vector<Eigen::Vector3d> obss, way_points, vel, acc;
  vector<double> seg_time_in;
  ifstream in(argv[1]);
  string key;
  while (in >> key) {
    if (key == "map_size") in >> map_size(0) >> map_size(1) >> map_size(2);
    else if (key == "origin") in >> origin(0) >> origin(1) >> origin(2);
    else if (key == "resolution") in >> resolution;
    else if (key == "obstacles") read_points(in, obss);
    else if (key == "waypoints") read_points(in, way_points);
    else if (key == "velocities") read_points(in, vel);
    else if (key == "accelerations") read_points(in, acc);
    else if (key == "segment_times") {
      size_t count = 0;
      in >> count;
      seg_time_in.resize(count);
      for (double &t : seg_time_in) in >> t;
    } else return 1;
  }

  GradTrajOptimizer::Config cfg;
  cfg.max_evals = argc > 2 ? atoi(argv[2]) : 60;
  cfg.time_limit_2 = 5.0;
  GradTrajOptimizer grad_traj_opt(cfg);                     // opti_node.cpp:58 (there: the ROS-parameter constructor)
  if (!grad_traj_opt.impl().ok()) {
    fprintf(stderr, "GradTrajOptimizer: %s\n", grad_traj_opt.impl().lastError());
    return 2;
  }
  grad_traj_opt.initSDFMap(map_size, origin, resolution);   // :61-64
  grad_traj_opt.updateSDFMap(obss);                         // :85
  if (!seg_time_in.empty()) {                               // the kinodynamic front end (compare2.cpp:236)
    const int np = (int)way_points.size();
    Eigen::MatrixXd Pos(np, 3), Vel(np, 3), Acc(np, 3);
    Eigen::VectorXd Time((int)seg_time_in.size());
    for (int i = 0; i < np; ++i)
      for (int a = 0; a < 3; ++a) { Pos(i, a) = way_points[i](a); Vel(i, a) = vel[i](a); Acc(i, a) = acc[i](a); }
    for (int i = 0; i < (int)seg_time_in.size(); ++i) Time(i) = seg_time_in[i];
    grad_traj_opt.setKinoPath(Pos, Vel, Acc, Time);
  } else {
    grad_traj_opt.setPath(way_points);                      // :99
  }
  Eigen::MatrixXd coeff;
  Eigen::VectorXd seg_time;
  grad_traj_opt.optimizeTrajectory(OPT_SECOND_STEP);        // :101
  grad_traj_opt.getCoefficient(coeff);                      // :102
  grad_traj_opt.getSegmentTime(seg_time);                   // :103
  vector<double> cost, time;
  grad_traj_opt.getCostCurve(cost, time);

  vector<double> c1((size_t)coeff.rows() * coeff.cols()), st((size_t)seg_time.rows());
  for (int i = 0; i < (int)coeff.rows(); ++i)
    for (int j = 0; j < (int)coeff.cols(); ++j) c1[(size_t)i * coeff.cols() + j] = coeff(i, j);
  for (int i = 0; i < (int)seg_time.rows(); ++i) st[i] = seg_time(i);
  printf("{\n\"evals\": %d,\n", grad_traj_opt.impl().iterations());
  print_vec("segment_time", st);
  print_vec("coeff1", c1);
  print_vec("cost_curve", cost, false);
  printf("}\n");
  return 0;
}
