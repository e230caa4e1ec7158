// batch_devices.cpp — many trajectories through the C++ host side on several devices (GradTrajBatch over gtop_group,
// include/gtop.h): the same scene file format as scene_runner.cpp, the waypoint list replicated B times with a small
// deterministic offset per copy, optimised once on the devices listed on the command line and once with one
// GradTrajOptimizer per trajectory (optimize_on_device: the single-problem form of the same loop).  Prints one JSON
// object that tests/test_gpu_group.py checks.
//
//   gtop_batch_devices <scene.txt> <B> <max_evals> <device> [<device> ...] [ragged] [fp32]
// ragged: every third copy loses one interior waypoint, every third two — three segment counts in one batch.
// fp32: the optimizer's evaluations in fp32 (Config::optimizer_fp32), in the batch and in the single-problem objects.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <string>
#include <vector>

#include "grad_traj_optimizer.hpp"

using namespace gtop_amd;

static bool read_points(std::istream &in, std::vector<Vec3> &out) {
  size_t count = 0;
  if (!(in >> count)) return false;
  out.resize(count);
  for (Vec3 &p : out)
    if (!(in >> p[0] >> p[1] >> p[2])) return false;
  return true;
}

int main(int argc, char **argv) {
  if (argc < 5) {
    std::fprintf(stderr, "usage: %s <scene.txt> <B> <max_evals> <device> [<device> ...]\n", argv[0]);
    return 1;
  }
  Vec3 map_size{}, origin{};
  double resolution = 0.0;
  std::vector<Vec3> obstacles, waypoints;
  {
    std::ifstream in(argv[1]);
    std::string key;
    while (in >> key) {
      if (key == "map_size") in >> map_size[0] >> map_size[1] >> map_size[2];
      else if (key == "origin") in >> origin[0] >> origin[1] >> origin[2];
      else if (key == "resolution") in >> resolution;
      else if (key == "obstacles") read_points(in, obstacles);
      else if (key == "waypoints") read_points(in, waypoints);
      else return 1;
    }
  }
  const int B = std::atoi(argv[2]);
  std::vector<int> devices;
  bool ragged = false, fp32 = false;
  for (int i = 4; i < argc; ++i) {
    if (std::string(argv[i]) == "ragged") ragged = true;
    else if (std::string(argv[i]) == "fp32") fp32 = true;
    else devices.push_back(std::atoi(argv[i]));
  }
  GradTrajOptimizer::Config cfg;
  cfg.max_evals = std::atoi(argv[3]);
  cfg.time_limit_2 = 30.0;   // evaluation-capped: reproducible
  cfg.optimizer_fp32 = fp32;  // (the single-problem objects below run the same loop: same bits)
  std::vector<std::vector<Vec3>> lists(B, waypoints);
  for (int b = 0; b < B; ++b)
    for (size_t i = 1; i + 1 < waypoints.size(); ++i) lists[b][i][0] += 0.01 * (b % 17) - 0.05 * ((b / 17) % 3);

  if (ragged && waypoints.size() >= 8)
    for (int b = 0; b < B; ++b) {
      if (b % 3 >= 1) lists[b].erase(lists[b].begin() + 5);
      if (b % 3 == 2) lists[b].erase(lists[b].begin() + 2);
    }
  GradTrajBatch batch(devices, cfg);
  if (!batch.ok()) { std::fprintf(stderr, "GradTrajBatch: %s\n", batch.lastError()); return 2; }
  batch.initSDFMap(map_size, origin, resolution);
  batch.updateSDFMap(obstacles);
  batch.setPaths(lists);
  batch.optimizeTrajectories(OPT_SECOND_STEP);
  if (!batch.ok()) { std::fprintf(stderr, "GradTrajBatch: %s\n", batch.lastError()); return 3; }

  // the same problems one object at a time
  double max_rel = 0.0, max_coeff = 0.0;
  cfg.optimize_on_device = 1;
  cfg.device = devices[0];
  const int check = B < 5 ? B : 5;
  for (int k = 0; k < check; ++k) {
    const int b = (int)((long long)k * (B - 1) / (check > 1 ? check - 1 : 1));
    GradTrajOptimizer one(cfg);
    one.initSDFMap(map_size, origin, resolution);
    one.updateSDFMap(obstacles);
    one.setPath(lists[b]);
    one.optimizeTrajectory(OPT_SECOND_STEP);
    std::vector<double> x = one.freeDerivatives(), g;
    const double c = GradTrajOptimizer::costFunc(x, g, &one);
    max_rel = std::fmax(max_rel, std::fabs(c - batch.costs()[b]) / std::fabs(c));
    Matrix ca, cb;
    one.getCoefficient(ca);
    batch.getCoefficient(b, cb);
    if (ca.a.size() != cb.a.size() || batch.segments(b) != (int)lists[b].size() - 1) return 4;
    for (size_t i = 0; i < ca.a.size(); ++i) max_coeff = std::fmax(max_coeff, std::fabs(ca.a[i] - cb.a[i]));
  }
  int min_ev = 1 << 30, max_ev = 0;
  for (int e : batch.evaluations()) { min_ev = e < min_ev ? e : min_ev; max_ev = e > max_ev ? e : max_ev; }
  int min_m = 1 << 30, max_m = 0;
  for (int b = 0; b < B; ++b) { const int m = batch.segments(b); min_m = m < min_m ? m : min_m; max_m = m > max_m ? m : max_m; }
  std::printf("{\"B\": %d, \"devices\": %d, \"gather\": \"%s\", \"max_rel_cost_diff\": %.3g, \"max_coeff_diff\": %.3g, "
              "\"min_evals\": %d, \"max_evals\": %d, \"min_segments\": %d, \"max_segments\": %d, \"cost0\": %.17g}\n",
              batch.size(), batch.devices(), batch.gatherBackend(), max_rel, max_coeff, min_ev, max_ev, min_m, max_m,
              batch.costs()[0]);
  return 0;
}
