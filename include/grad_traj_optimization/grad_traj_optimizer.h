/*
 * grad_traj_optimization/grad_traj_optimizer.h — drop-in for the reference header of the same path
 * (include/grad_traj_optimization/grad_traj_optimizer.h:20-39, :127-130 of EpicOne1/grad_traj_optimization):
 * a GLOBAL `class GradTrajOptimizer` with the reference's Eigen signatures, so that a caller written against the
 * reference — src/opti_node.cpp:58-106: GradTrajOptimizer grad_traj_opt; initSDFMap; updateSDFMap; setPath;
 * optimizeTrajectory(OPT_SECOND_STEP); getCoefficient; getSegmentTime — compiles against this repo unchanged, with
 * this directory's parent on the include path in place of the reference's and libgtop_hip.so on the link line.
 *
 * It is a thin adapter: every method converts its Eigen arguments and delegates to gtop_amd::GradTrajOptimizer
 * (csrc/grad_traj_optimizer.hpp, the C++ host shim over the C-ABI of include/gtop.h).  Header-only; needs Eigen3.
 *
 * STATUS IN THIS REPOSITORY: the build image has no Eigen, so this header is not compiled against the real library
 * here; tests/test_cpp_shim.py compiles it against a small test double of the Eigen types it names
 * (tests/cpp/eigen_double/) and runs the opti_node call sequence through it — a check of this adapter's own syntax
 * and delegation, nothing more.  With ROS present (<ros/ros.h>) the constructor reads the same 21 parameters from the
 * same global names as the reference's (src/grad_traj_optimizer.cpp:5-32); without it the defaults of
 * launch/opti_node.launch:3-28 apply, and the Config constructor sets them explicitly.
 */
#ifndef _GRAD_TRAJ_OPTIMIZER_H_
#define _GRAD_TRAJ_OPTIMIZER_H_

#if !defined(__has_include)
#error "this adapter needs __has_include (C++17)"
#elif !__has_include(<Eigen/Eigen>)
#error "grad_traj_optimization/grad_traj_optimizer.h needs Eigen3 (<Eigen/Eigen>); without Eigen use csrc/grad_traj_optimizer.hpp (gtop_amd::GradTrajOptimizer)"
#else

#include <Eigen/Eigen>
#if __has_include(<ros/ros.h>)
#include <ros/ros.h>
#define GTOP_ADAPTER_HAVE_ROS 1
#endif

#include <vector>

#include "grad_traj_optimizer.hpp"   /* csrc/: gtop_amd::GradTrajOptimizer; also defines OPT_INITIAL_TRY / _FIRST_STEP / _SECOND_STEP */

#define GDTB getDistanceToBoundary   /* (grad_traj_optimizer.h:13; unused there too) */

class GradTrajOptimizer {
 public:
  using Config = gtop_amd::GradTrajOptimizer::Config;

  /* src/grad_traj_optimizer.cpp:3-33 */
  GradTrajOptimizer() : impl_(rosConfig()) {}
  /* not in the reference: parameters without a ROS master */
  explicit GradTrajOptimizer(const Config &cfg) : impl_(cfg) {}

  /* :67-110 */
  void setPath(const std::vector<Eigen::Vector3d> &way_points) {
    std::vector<gtop_amd::Vec3> wp(way_points.size());
    for (size_t i = 0; i < way_points.size(); ++i) wp[i] = {way_points[i](0), way_points[i](1), way_points[i](2)};
    impl_.setPath(wp);
  }

  /* :35-65 — Pos / Vel / Acc: (segments + 1) x 3, Time: segments */
  void setKinoPath(Eigen::MatrixXd &Pos, Eigen::MatrixXd &Vel, Eigen::MatrixXd &Acc, Eigen::VectorXd &Time) {
    gtop_amd::Matrix P = toMatrix(Pos), V = toMatrix(Vel), A = toMatrix(Acc);
    std::vector<double> T((size_t)Time.rows());
    for (int i = 0; i < (int)Time.rows(); ++i) T[i] = Time(i);
    impl_.setKinoPath(P, V, A, T);
  }

  /* :128-243 — always true, as the reference (:242) */
  bool optimizeTrajectory(int step) { return impl_.optimizeTrajectory(step); }

  /* :245-247 — segments x 18 */
  void getCoefficient(Eigen::MatrixXd &coeff) {
    gtop_amd::Matrix c;
    impl_.getCoefficient(c);
    coeff.resize(c.rows, c.cols);
    for (int i = 0; i < c.rows; ++i)
      for (int j = 0; j < c.cols; ++j) coeff(i, j) = c(i, j);
  }

  /* :249-251 */
  void getSegmentTime(Eigen::VectorXd &seg_time) {
    std::vector<double> t;
    impl_.getSegmentTime(t);
    seg_time.resize((int)t.size());
    for (int i = 0; i < (int)t.size(); ++i) seg_time(i) = t[i];
  }

  /* :112-115 */
  void initSDFMap(Eigen::Vector3d map_size_3d, Eigen::Vector3d origin, double resolution) {
    impl_.initSDFMap({map_size_3d(0), map_size_3d(1), map_size_3d(2)}, {origin(0), origin(1), origin(2)}, resolution);
  }

  /* :117-126 */
  void updateSDFMap(std::vector<Eigen::Vector3d> obs) {
    std::vector<gtop_amd::Vec3> o(obs.size());
    for (size_t i = 0; i < obs.size(); ++i) o[i] = {obs[i](0), obs[i](1), obs[i](2)};
    impl_.updateSDFMap(o);
  }

  /* grad_traj_optimizer.h:127-130 */
  void getCostCurve(std::vector<double> &cost, std::vector<double> &time) { impl_.getCostCurve(cost, time); }

  /* not in the reference: the object underneath (status, last error, evaluation count, the gtop_ctx) */
  gtop_amd::GradTrajOptimizer &impl() { return impl_; }

 private:
  static gtop_amd::Matrix toMatrix(const Eigen::MatrixXd &M) {
    gtop_amd::Matrix r((int)M.rows(), (int)M.cols());
    for (int i = 0; i < (int)M.rows(); ++i)
      for (int j = 0; j < (int)M.cols(); ++j) r(i, j) = M(i, j);
    return r;
  }

  static Config rosConfig() {
    Config c;   /* defaults: launch/opti_node.launch:3-28 */
#ifdef GTOP_ADAPTER_HAVE_ROS
    /* the reference's names, global and hard-coded (src/grad_traj_optimizer.cpp:5-32) */
    ros::param::get("/traj_opti_node1/alg", c.alg);
    ros::param::get("/traj_opti_node1/time_limit_1", c.time_limit_1);
    ros::param::get("/traj_opti_node1/time_limit_2", c.time_limit_2);
    ros::param::get("/traj_opti_node1/dt", c.dt);
    ros::param::get("/traj_opti_node1/ws", c.ws);
    ros::param::get("/traj_opti_node1/wc", c.wc);
    ros::param::get("/traj_opti_node1/alpha", c.alpha);
    ros::param::get("/traj_opti_node1/r", c.r);
    ros::param::get("/traj_opti_node1/d0", c.d0);
    ros::param::get("/traj_opti_node1/alpha_v", c.alpha_v);
    ros::param::get("/traj_opti_node1/r_v", c.r_v);
    ros::param::get("/traj_opti_node1/v0", c.v0);
    ros::param::get("/traj_opti_node1/alpha_a", c.alpha_a);
    ros::param::get("/traj_opti_node1/r_a", c.r_a);
    ros::param::get("/traj_opti_node1/a0", c.a0);
    ros::param::get("/traj_opti_node1/bos", c.bos);
    ros::param::get("/traj_opti_node1/vos", c.vos);
    ros::param::get("/traj_opti_node1/aos", c.aos);
    ros::param::get("/traj_opti_node1/mean_v", c.mean_v);
    ros::param::get("/traj_opti_node1/mean_a", c.mean_a);
    ros::param::get("/traj_opti_node1/init_time", c.init_time);
#endif
    return c;
  }

  gtop_amd::GradTrajOptimizer impl_;
};

#endif /* Eigen present */
#endif /* _GRAD_TRAJ_OPTIMIZER_H_ */
