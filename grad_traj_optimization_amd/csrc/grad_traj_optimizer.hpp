// grad_traj_optimizer.hpp — C++ host shim with the public interface of the
// reference's GradTrajOptimizer
// (include/grad_traj_optimization/grad_traj_optimizer.h:20-39, :127-130 of
// EpicOne1/grad_traj_optimization), implemented on the C-ABI of include/gtop.h.
//
// Same method names, argument meaning and (absence of) error behaviour:
// optimizeTrajectory always returns true (src/grad_traj_optimizer.cpp:242),
// nothing throws.  Differences forced by the image (no Eigen, no ROS, no
// NLopt headers): Eigen::Vector3d -> gtop_amd::Vec3, Eigen::MatrixXd ->
// gtop_amd::Matrix (row-major), the ROS parameter server -> the Config struct
// (same 21 names, src/grad_traj_optimizer.cpp:5-32), nlopt::opt -> the CCSA-MMA
// driver in mma.hpp (NLopt algorithm 24 = LD_MMA is what opti_node.launch
// selects; its algorithm is restated from Svanberg 2002 / NLopt 2.5.0's
// published mma.c, see mma.hpp).
#ifndef GTOP_AMD_GRAD_TRAJ_OPTIMIZER_HPP_
#define GTOP_AMD_GRAD_TRAJ_OPTIMIZER_HPP_

#include <array>
#include <string>
#include <vector>

#include "gtop.h"

#define OPT_INITIAL_TRY 0
#define OPT_FIRST_STEP 1
#define OPT_SECOND_STEP 2

namespace gtop_amd {

using Vec3 = std::array<double, 3>;

// Minimal dense row-major matrix (stands in for Eigen::MatrixXd at the API).
struct Matrix {
  int rows = 0, cols = 0;
  std::vector<double> a;
  Matrix() = default;
  Matrix(int r, int c) : rows(r), cols(c), a((size_t)r * c, 0.0) {}
  void resize(int r, int c) { rows = r; cols = c; a.assign((size_t)r * c, 0.0); }
  double &operator()(int r, int c) { return a[(size_t)r * cols + c]; }
  double operator()(int r, int c) const { return a[(size_t)r * cols + c]; }
};

class GradTrajOptimizer {
 public:
  // The parameters the reference's ctor reads from the ROS parameter server
  // under /traj_opti_node1/* (src/grad_traj_optimizer.cpp:5-32); defaults are
  // launch/opti_node.launch:3-28.
  struct Config {
    int alg = 24;                 // NLopt id; 24 = LD_MMA
    double time_limit_1 = 0.4, time_limit_2 = 0.1;
    double dt = 0.2;              // read but unused by the cost (:8, :351)
    double ws = 1.0, wc = 5.0;
    double alpha = 10.0, r = 0.5, d0 = 0.8;
    double alpha_v = 0.0, r_v = 1.5, v0 = 2.5;
    double alpha_a = 0.0, r_a = 1.5, a0 = 3.5;
    double bos = 3.0, vos = 8.0, aos = 10.0;
    double mean_v = 1.8, mean_a = 0.0, init_time = 0.3;
    // not in the reference: which GPU, whether to run the dyn-feasibility
    // block the reference has commented out (:383-407), and an evaluation cap
    // for reproducible runs (0 = stop on maxtime only, as the reference does).
    int device = 0;
    int enable_dyn = 0;
    int max_evals = 0;
    // 0: the optimizer (csrc/mma.hpp standing in for NLopt's LD_MMA) runs on the host and calls the cost function
    // once per evaluation, as the reference does (:195) — iter_num, total_time and the cost curve are kept per call.
    // 1: the whole optimisation is ONE launch of the batched device optimizer (gtop_optimize_batch_ex, same
    // algorithm, same stop rules); only the evaluation count is reported, the cost curve stays empty.
    int optimize_on_device = 0;
    // the device optimizer's evaluations in fp32 (gtop_set_optimizer_precision; its state and results stay fp64):
    // for large batches through GradTrajBatch; needs the device optimizer (optimize_on_device, or GradTrajBatch)
    int optimizer_fp32 = 0;
  };

  GradTrajOptimizer();
  explicit GradTrajOptimizer(const Config &cfg);
  ~GradTrajOptimizer();
  GradTrajOptimizer(const GradTrajOptimizer &) = delete;
  GradTrajOptimizer &operator=(const GradTrajOptimizer &) = delete;

  void setPath(const std::vector<Vec3> &way_points);                      // :67-110
  void setKinoPath(const Matrix &Pos, const Matrix &Vel, const Matrix &Acc,
                   const std::vector<double> &Time);                      // :35-65
  bool optimizeTrajectory(int step);                                      // :128-243
  void getCoefficient(Matrix &coeff);                                     // :245-247
  void getSegmentTime(std::vector<double> &seg_time);                     // :249-251
  void initSDFMap(Vec3 map_size_3d, Vec3 origin, double resolution);      // :112-115
  void updateSDFMap(const std::vector<Vec3> &obs);                        // :117-126
  void getCostCurve(std::vector<double> &cost, std::vector<double> &time);  // header :127-130

  // NLopt-format cost function (:554-562).  Private in the reference; public
  // here so that an external NLopt (or a test) can take its address.
  static double costFunc(const std::vector<double> &x, std::vector<double> &grad, void *func_data);

  // extras (not in the reference)
  bool ok() const { return ctx_ != nullptr && last_status_ == GTOP_OK; }
  const char *lastError() const;
  gtop_ctx *context() { return ctx_; }
  const std::vector<double> &freeDerivatives() const { return dp_; }      // Dp, axis-major
  int iterations() const { return last_evals_; }

 private:
  void setupProblem(const std::vector<double> &path_flat, int npts, const std::vector<double> &seg_time,
                    const std::vector<double> &Dx, const std::vector<double> &Dy, const std::vector<double> &Dz);
  void pushParams();
  void coefficientsFromDerivatives(const std::vector<double> &dp);        // :253-279

  Config cfg_;
  gtop_ctx *ctx_ = nullptr;
  int last_status_ = GTOP_OK;
  std::string create_error_;

  int m_ = 0, num_dp_ = 0;
  int step_ = 1;
  std::vector<double> path_;          // npts x 3
  std::vector<double> segment_time_;  // m
  std::vector<double> df_;            // 3 x 6
  std::vector<double> dp_;            // 3 x num_dp, axis-major == NLopt x
  Matrix coeff_;                      // m x 18
  int last_evals_ = 0;
};

// ---------------------------------------------------------------------------
// GradTrajBatch — the same public steps for MANY trajectories at once, on one or several GPUs (not in the reference,
// which optimises one trajectory per object: a planner that holds N candidate paths runs N objects one after the other).
// initSDFMap / updateSDFMap / setPaths / optimizeTrajectories / getCoefficient mirror GradTrajOptimizer's methods with a
// batch index; underneath is a gtop_group (include/gtop.h): the distance field replicated on every listed device, the
// batch in contiguous slices, the whole optimisation of a slice ONE launch on its device.
// ---------------------------------------------------------------------------
class GradTrajBatch {
 public:
  // devices: HIP ordinals, one slice of the batch each (an ordinal may repeat); cfg as for GradTrajOptimizer
  explicit GradTrajBatch(const std::vector<int> &devices, const GradTrajOptimizer::Config &cfg = GradTrajOptimizer::Config());
  ~GradTrajBatch();
  GradTrajBatch(const GradTrajBatch &) = delete;
  GradTrajBatch &operator=(const GradTrajBatch &) = delete;

  void initSDFMap(Vec3 map_size_3d, Vec3 origin, double resolution);
  void updateSDFMap(const std::vector<Vec3> &obs);
  // B waypoint lists of m_b + 1 >= 3 points each — the lengths may differ (candidate paths of a planner seldom have the
  // same number of waypoints): segment times, Df and the straight-line start as GradTrajOptimizer::setPath makes
  // them (src/grad_traj_optimizer.cpp:67-110); trajectories of equal segment count form one device problem
  void setPaths(const std::vector<std::vector<Vec3>> &way_points);
  // every trajectory's LD_MMA run (:128-243), all at once; stop rules: cfg.max_evals and the step's time limit
  bool optimizeTrajectories(int step);
  void getCoefficient(int b, Matrix &coeff) const;               // trajectory b, m_b x 18
  int segments(int b) const { return (b >= 0 && b < B_) ? m_of_[b] : 0; }
  const std::vector<double> &costs() const { return min_cost_; }  // the minimum each trajectory reached
  const std::vector<int> &evaluations() const { return nevals_; }
  int size() const { return B_; }
  int devices() const;
  const char *gatherBackend() const;                              // "rccl" / "copy" (gtop_group_gather_backend)
  bool ok() const { return grp_ != nullptr && last_status_ == GTOP_OK; }
  const char *lastError() const;

 private:
  GradTrajOptimizer::Config cfg_;
  gtop_group *grp_ = nullptr;
  int last_status_ = GTOP_OK;
  int B_ = 0;
  // per trajectory: segment count and where its rows start in the flat arrays (waypoints x 3, times, free variables)
  std::vector<int> m_of_;
  std::vector<size_t> path_at_, T_at_, x_at_;
  std::vector<double> path_, T_, Df_, x_, min_cost_;
  std::vector<int> nevals_;
};

}  // namespace gtop_amd

#endif  // GTOP_AMD_GRAD_TRAJ_OPTIMIZER_HPP_
