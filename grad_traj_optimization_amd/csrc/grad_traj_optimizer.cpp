// grad_traj_optimizer.cpp — host shim over the C-ABI (see the header).
// file:line citations are into EpicOne1/grad_traj_optimization.
#include "grad_traj_optimizer.hpp"

#include <algorithm>
#include <cmath>
#include <cstdio>

#include "mma.hpp"

namespace gtop_amd {

namespace {

// Coefficients of one quintic segment from its boundary derivatives
// [p0, pT, v0, vT, a0, aT] and duration T: the closed form of A_s^-1 d
// (A_s: src/qp_generator.cpp:185-195; the reference forms A.inverse(), :334-336, :390).
void quintic_from_boundary(const double d[6], double T, double c[6]) {
  const double p0 = d[0], pT = d[1], v0 = d[2], vT = d[3], a0 = d[4], aT = d[5];
  const double T2 = T * T, iT = 1.0 / T, iT3 = iT * iT * iT;
  const double P = pT - p0 - v0 * T - 0.5 * a0 * T2;
  const double V = (vT - v0 - a0 * T) * T;
  const double A = (aT - a0) * T2;
  c[0] = p0;
  c[1] = v0;
  c[2] = 0.5 * a0;
  c[3] = (10 * P - 4 * V + 0.5 * A) * iT3;
  c[4] = (-15 * P + 7 * V - A) * (iT3 * iT);
  c[5] = (6 * P - 3 * V + 0.5 * A) * (iT3 * iT * iT);
}

double nlopt_trampoline(unsigned n, const double *x, double *grad, void *data) {
  return gtop_cost_nlopt(n, x, grad, data);
}

}  // namespace

GradTrajOptimizer::GradTrajOptimizer() : GradTrajOptimizer(Config()) {}

GradTrajOptimizer::GradTrajOptimizer(const Config &cfg) : cfg_(cfg) {
  last_status_ = gtop_create(&ctx_, cfg_.device);
  if (last_status_ != GTOP_OK) {
    create_error_ = gtop_last_error(nullptr);
    ctx_ = nullptr;
    return;
  }
  pushParams();
}

GradTrajOptimizer::~GradTrajOptimizer() {
  if (ctx_) gtop_destroy(ctx_);
}

const char *GradTrajOptimizer::lastError() const {
  return ctx_ ? gtop_last_error(ctx_) : create_error_.c_str();
}

void GradTrajOptimizer::pushParams() {
  if (!ctx_) return;
  gtop_params p;
  p.ws = cfg_.ws; p.wc = cfg_.wc;
  p.alpha = cfg_.alpha; p.r = cfg_.r; p.d0 = cfg_.d0;
  p.alpha_v = cfg_.alpha_v; p.r_v = cfg_.r_v; p.v0 = cfg_.v0;
  p.alpha_a = cfg_.alpha_a; p.r_a = cfg_.r_a; p.a0 = cfg_.a0;
  p.step = step_;
  p.enable_dyn = cfg_.enable_dyn;
  last_status_ = gtop_set_params(ctx_, &p);
}

void GradTrajOptimizer::initSDFMap(Vec3 map_size_3d, Vec3 origin, double resolution) {
  if (!ctx_) return;
  last_status_ = gtop_init_sdf_map(ctx_, map_size_3d.data(), origin.data(), resolution);
}

void GradTrajOptimizer::updateSDFMap(const std::vector<Vec3> &obs) {
  if (!ctx_) return;
  // std::array<double,3> is layout-compatible with xyz triples
  last_status_ = gtop_update_sdf_map(ctx_, obs.empty() ? nullptr : obs[0].data(), (int)obs.size());
}

// Common tail of setPath / setKinoPath: Df, Dp (getInitialD,
// src/qp_generator.cpp:407-451), initial coefficients (Px = A^-1 Dx,
// :334-336 / :134-136), then the problem goes to the device.
void GradTrajOptimizer::setupProblem(const std::vector<double> &path_flat, int npts,
                                     const std::vector<double> &seg_time, const std::vector<double> &Dx,
                                     const std::vector<double> &Dy, const std::vector<double> &Dz) {
  path_ = path_flat;
  segment_time_ = seg_time;
  m_ = npts - 1;
  num_dp_ = 3 * m_ - 3;
  const std::vector<double> *D[3] = {&Dx, &Dy, &Dz};
  df_.assign(18, 0.0);
  dp_.assign((size_t)3 * num_dp_, 0.0);
  coeff_.resize(m_, 18);
  for (int a = 0; a < 3; ++a) {
    const std::vector<double> &d = *D[a];
    df_[a * 6 + 0] = d[0];                  // :418-423
    df_[a * 6 + 3] = d[d.size() - 5];
    df_[a * 6 + 1] = 0.0;                   // startVel / startAcc (:425-431): file-scope globals that
    df_[a * 6 + 2] = 0.0;                   // every caller leaves at zero (setPath passes 0, :83-85)
    for (int k = 1; k < m_; k++)            // :433-439
      for (int i = 0; i < 3; i++) dp_[(size_t)a * num_dp_ + (k - 1) * 3 + i] = d[(k - 1) * 6 + 2 * i + 1];
    for (int s = 0; s < m_; ++s) {
      double c[6];
      quintic_from_boundary(&d[s * 6], segment_time_[s], c);
      for (int j = 0; j < 6; ++j) coeff_(s, 6 * a + j) = c[j];
    }
  }
  if (!ctx_) return;
  last_status_ = gtop_set_problem(ctx_, 1, m_, segment_time_.data(), m_, df_.data());
}

void GradTrajOptimizer::setPath(const std::vector<Vec3> &way_points) {
  const int npts = (int)way_points.size();
  if (npts < 3) {   // m >= 2: StackOptiDep indexes out of bounds otherwise (src/qp_generator.cpp:365-378)
    last_status_ = GTOP_ERR_INVALID;
    return;
  }
  const int m = npts - 1;
  std::vector<double> path((size_t)npts * 3);
  for (int i = 0; i < npts; ++i)
    for (int a = 0; a < 3; ++a) path[i * 3 + a] = way_points[i][a];
  // :73-81 — `i == segment_time.size()` never holds, so only segment 0 gets init_time
  std::vector<double> T(m);
  for (int i = 0; i < m; ++i) {
    const double dx = path[i * 3] - path[(i + 1) * 3], dy = path[i * 3 + 1] - path[(i + 1) * 3 + 1],
                 dz = path[i * 3 + 2] - path[(i + 1) * 3 + 2];
    const double len = std::sqrt(dx * dx + dy * dy + dz * dz);
    if (i == 0 || i == m) T[i] = len / cfg_.mean_v + cfg_.init_time;
    else T[i] = len / cfg_.mean_v;
  }
  // PolyQPGeneration(type = 2), src/qp_generator.cpp:199-221: positions at both
  // ends of each segment, start vel = acc = 0 (:83-85), everything else 0.
  std::vector<double> D[3];
  for (int a = 0; a < 3; ++a) {
    D[a].assign((size_t)6 * m, 0.0);
    for (int k = 1; k < m + 1; k++) {
      D[a][(k - 1) * 6] = path[(k - 1) * 3 + a];
      D[a][(k - 1) * 6 + 1] = path[k * 3 + a];
    }
  }
  setupProblem(path, npts, T, D[0], D[1], D[2]);
}

void GradTrajOptimizer::setKinoPath(const Matrix &Pos, const Matrix &Vel, const Matrix &Acc,
                                    const std::vector<double> &Time) {
  const int m = (int)Time.size();
  if (m < 2 || Pos.rows != m + 1 || Vel.rows != m + 1 || Acc.rows != m + 1 || Pos.cols != 3 ||
      Vel.cols != 3 || Acc.cols != 3) {
    last_status_ = GTOP_ERR_INVALID;
    return;
  }
  std::vector<double> path((size_t)(m + 1) * 3);
  for (int i = 0; i < m + 1; ++i)
    for (int a = 0; a < 3; ++a) path[i * 3 + a] = Pos(i, a);
  // PolyKinoGeneration, src/qp_generator.cpp:63-86
  std::vector<double> D[3];
  for (int a = 0; a < 3; ++a) {
    D[a].assign((size_t)6 * m, 0.0);
    for (int k = 0; k < m; k++) {
      D[a][k * 6] = Pos(k, a);
      D[a][k * 6 + 1] = Pos(k + 1, a);
      D[a][k * 6 + 2] = Vel(k, a);
      D[a][k * 6 + 3] = Vel(k + 1, a);
      D[a][k * 6 + 4] = Acc(k, a);
      D[a][k * 6 + 5] = Acc(k + 1, a);
    }
  }
  setupProblem(path, m + 1, Time, D[0], D[1], D[2]);
}

// getCoefficientFromDerivative, :253-279: coe = L d, which per segment is
// A_s^-1 applied to the derivatives of its two waypoints.
void GradTrajOptimizer::coefficientsFromDerivatives(const std::vector<double> &dp) {
  coeff_.resize(m_, 18);
  for (int a = 0; a < 3; ++a) {
    auto wp = [&](int j, int der) -> double {   // derivative `der` of waypoint j on axis a
      if (j == 0) return df_[a * 6 + der];
      if (j == m_) return df_[a * 6 + 3 + der];
      return dp[(size_t)a * num_dp_ + 3 * (j - 1) + der];
    };
    for (int s = 0; s < m_; ++s) {
      const double d[6] = {wp(s, 0), wp(s + 1, 0), wp(s, 1), wp(s + 1, 1), wp(s, 2), wp(s + 1, 2)};
      double c[6];
      quintic_from_boundary(d, segment_time_[s], c);
      for (int j = 0; j < 6; ++j) coeff_(s, 6 * a + j) = c[j];
    }
  }
}

bool GradTrajOptimizer::optimizeTrajectory(int step) {
  if (step != 0 && step != 1 && step != 2) {   // :129-131 (the reference prints and carries on)
    std::printf("step number error, step should be 0, 1 or 2\n");
  }
  if (!ctx_ || m_ < 2) return true;   // :242 — always true
  step_ = step;
  if (step == 0 || step == 1 || step == 2) pushParams();

  const unsigned n = 3u * (unsigned)num_dp_;
  MmaOptions opt;
  if (step == OPT_FIRST_STEP) opt.maxtime = cfg_.time_limit_1;        // :144-148
  else if (step == OPT_SECOND_STEP) opt.maxtime = cfg_.time_limit_2;
  opt.maxeval = cfg_.max_evals;

  // bounds, :151-179
  std::vector<double> lb(n), ub(n);
  for (int i = 0; i < num_dp_; ++i) {
    for (int a = 0; a < 3; ++a) {
      const size_t j = (size_t)i + (size_t)a * num_dp_;
      if (i % 3 == 0) {
        lb[j] = path_[(i / 3 + 1) * 3 + a] - cfg_.bos;
        ub[j] = path_[(i / 3 + 1) * 3 + a] + cfg_.bos;
      } else if (i % 3 == 1) {
        lb[j] = -cfg_.vos;
        ub[j] = cfg_.vos;
      } else {
        lb[j] = -cfg_.aos;
        ub[j] = cfg_.aos;
      }
    }
  }
  std::vector<double> x = dp_;   // :182-187
  gtop_clear_cost_curve(ctx_);   // :192-194
  if (cfg_.optimize_on_device) {
    // (maxtime alone bounds the reference's run: an evaluation cap far past what it can reach stands in for "none")
    const gtop_stop stop = {opt.maxeval > 0 ? opt.maxeval : (1 << 24), 0.0, 0.0, opt.maxtime};
    double minf = 0.0;
    int32_t nev = 0, code = 0;
    gtop_set_optimizer_precision(ctx_, cfg_.optimizer_fp32 ? GTOP_F32 : GTOP_F64);
    last_status_ = gtop_optimize_batch_ex(ctx_, 1, x.data(), lb.data(), ub.data(), &stop, &minf, &nev, &code);
    last_evals_ = nev;
  } else {
    MmaResult r = mma_minimize(n, nlopt_trampoline, ctx_, lb.data(), ub.data(), x.data(), opt);
    last_evals_ = r.nevals;
  }
  dp_ = x;                        // :202-207
  coefficientsFromDerivatives(x);  // :230
  if (step == 1 || step == 2) {   // :233-240
    int64_t it = 0;
    double tt = 0;
    gtop_get_stats(ctx_, &it, &tt);
    std::printf("total time:%g\niterative num:%lld\n", tt, (long long)it);
    if (step == 2) gtop_reset_stats(ctx_);
  }
  return true;
}

void GradTrajOptimizer::getCoefficient(Matrix &coeff) { coeff = coeff_; }

void GradTrajOptimizer::getSegmentTime(std::vector<double> &seg_time) { seg_time = segment_time_; }

void GradTrajOptimizer::getCostCurve(std::vector<double> &cost, std::vector<double> &time) {
  cost.clear();
  time.clear();
  if (!ctx_) return;
  int count = 0;
  gtop_get_cost_curve(ctx_, nullptr, nullptr, 0, &count);
  cost.resize(count);
  time.resize(count);
  if (count) gtop_get_cost_curve(ctx_, cost.data(), time.data(), count, &count);
}

// :554-562
double GradTrajOptimizer::costFunc(const std::vector<double> &x, std::vector<double> &grad, void *func_data) {
  GradTrajOptimizer *gtop = reinterpret_cast<GradTrajOptimizer *>(func_data);
  grad.resize(x.size());   // the reference resizes grad itself (:426)
  return gtop_cost_nlopt((unsigned)x.size(), x.data(), grad.data(), gtop->ctx_);
}

// ---- GradTrajBatch ----------------------------------------------------------------------------------------------

GradTrajBatch::GradTrajBatch(const std::vector<int> &devices, const GradTrajOptimizer::Config &cfg) : cfg_(cfg) {
  last_status_ = gtop_group_create(&grp_, devices.data(), (int)devices.size());
  if (last_status_ != GTOP_OK) grp_ = nullptr;
}

GradTrajBatch::~GradTrajBatch() {
  if (grp_) gtop_group_destroy(grp_);
}

int GradTrajBatch::devices() const { return grp_ ? gtop_group_size(grp_) : 0; }
const char *GradTrajBatch::gatherBackend() const { return grp_ ? gtop_group_gather_backend(grp_) : ""; }
const char *GradTrajBatch::lastError() const { return grp_ ? gtop_group_last_error(grp_) : "gtop_group_create failed"; }

void GradTrajBatch::initSDFMap(Vec3 map_size_3d, Vec3 origin, double resolution) {
  if (grp_) last_status_ = gtop_group_init_sdf_map(grp_, map_size_3d.data(), origin.data(), resolution);
}

void GradTrajBatch::updateSDFMap(const std::vector<Vec3> &obs) {
  if (grp_) last_status_ = gtop_group_update_sdf_map(grp_, obs.empty() ? nullptr : obs[0].data(), (int)obs.size());
}

void GradTrajBatch::setPaths(const std::vector<std::vector<Vec3>> &way_points) {
  const int B = (int)way_points.size();
  bool valid = B > 0;
  for (const auto &w : way_points) valid = valid && w.size() >= 3;
  if (!grp_ || !valid) {
    last_status_ = GTOP_ERR_INVALID;
    return;
  }
  B_ = B;
  m_of_.assign(B, 0);
  path_at_.assign(B, 0);
  T_at_.assign(B, 0);
  x_at_.assign(B, 0);
  size_t np = 0, nt = 0, nx = 0;
  for (int b = 0; b < B; ++b) {
    const int m = (int)way_points[b].size() - 1;
    m_of_[b] = m;
    path_at_[b] = np; T_at_[b] = nt; x_at_[b] = nx;
    np += (size_t)(m + 1) * 3; nt += (size_t)m; nx += 9 * (size_t)(m - 1);
  }
  path_.assign(np, 0.0);
  T_.assign(nt, 0.0);
  Df_.assign((size_t)B * 18, 0.0);
  x_.assign(nx, 0.0);
  for (int b = 0; b < B; ++b) {
    const int m = m_of_[b], num_dp = 3 * m - 3;
    double *p = &path_[path_at_[b]], *T = &T_[T_at_[b]], *x = &x_[x_at_[b]];
    for (int i = 0; i <= m; ++i)
      for (int a = 0; a < 3; ++a) p[i * 3 + a] = way_points[b][i][a];
    for (int i = 0; i < m; ++i) {   // :73-81 (only segment 0 gets init_time: the `i == size()` clause never holds)
      const double dx = p[i * 3] - p[(i + 1) * 3], dy = p[i * 3 + 1] - p[(i + 1) * 3 + 1], dz = p[i * 3 + 2] - p[(i + 1) * 3 + 2];
      T[i] = std::sqrt(dx * dx + dy * dy + dz * dz) / cfg_.mean_v + (i == 0 ? cfg_.init_time : 0.0);
    }
    for (int a = 0; a < 3; ++a) {   // getInitialD (src/qp_generator.cpp:407-451): positions only, zero velocity / acceleration
      Df_[(size_t)b * 18 + a * 6 + 0] = p[a];
      Df_[(size_t)b * 18 + a * 6 + 3] = p[m * 3 + a];
      for (int k = 1; k < m; ++k) x[(size_t)a * num_dp + (k - 1) * 3] = p[k * 3 + a];
    }
  }
  min_cost_.assign(B, 0.0);
  nevals_.assign(B, 0);
  last_status_ = GTOP_OK;
}

bool GradTrajBatch::optimizeTrajectories(int step) {
  if (!grp_ || B_ == 0) return true;   // (:242 — always true)
  gtop_params p;
  p.ws = cfg_.ws; p.wc = cfg_.wc;
  p.alpha = cfg_.alpha; p.r = cfg_.r; p.d0 = cfg_.d0;
  p.alpha_v = cfg_.alpha_v; p.r_v = cfg_.r_v; p.v0 = cfg_.v0;
  p.alpha_a = cfg_.alpha_a; p.r_a = cfg_.r_a; p.a0 = cfg_.a0;
  p.step = step;
  p.enable_dyn = cfg_.enable_dyn;
  if ((last_status_ = gtop_group_set_params(grp_, &p)) != GTOP_OK) return true;
  for (int i = 0; i < gtop_group_size(grp_); ++i)
    gtop_set_optimizer_precision(gtop_group_context(grp_, i), cfg_.optimizer_fp32 ? GTOP_F32 : GTOP_F64);
  const double maxtime = step == OPT_FIRST_STEP ? cfg_.time_limit_1 : (step == OPT_SECOND_STEP ? cfg_.time_limit_2 : 0.0);
  const gtop_stop stop = {cfg_.max_evals > 0 ? cfg_.max_evals : (1 << 24), 0.0, 0.0, maxtime};   // :144-148
  // one device problem per distinct segment count, in rising order; each is the whole optimisation of its
  // trajectories in one launch per device (the wall-clock limit is each problem's own, as it is each object's in the
  // reference)
  std::vector<int> order(B_);
  for (int b = 0; b < B_; ++b) order[b] = b;
  std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return m_of_[a] < m_of_[b]; });
  std::vector<double> T, Df, x, path, lb, ub, cost;
  std::vector<int32_t> nev, code;
  for (int lo = 0; lo < B_;) {
    const int m = m_of_[order[lo]];
    int hi = lo;
    while (hi < B_ && m_of_[order[hi]] == m) ++hi;
    const int nb = hi - lo;
    const size_t n = 9 * (size_t)(m - 1), np = (size_t)(m + 1) * 3;
    T.resize((size_t)nb * m); Df.resize((size_t)nb * 18); x.resize(nb * n); path.resize(nb * np);
    lb.resize(nb * n); ub.resize(nb * n); cost.resize(nb); nev.resize(nb); code.resize(nb);
    for (int k = 0; k < nb; ++k) {
      const int b = order[lo + k];
      std::copy_n(&T_[T_at_[b]], m, &T[(size_t)k * m]);
      std::copy_n(&Df_[(size_t)b * 18], 18, &Df[(size_t)k * 18]);
      std::copy_n(&x_[x_at_[b]], n, &x[k * n]);
      std::copy_n(&path_[path_at_[b]], np, &path[k * np]);
    }
    gtop_default_bounds(nb, m, path.data(), cfg_.bos, cfg_.vos, cfg_.aos, lb.data(), ub.data());   // :151-179
    if ((last_status_ = gtop_group_set_problem(grp_, nb, m, T.data(), m, Df.data())) != GTOP_OK) return true;
    last_status_ = gtop_group_optimize_batch_ex(grp_, nb, x.data(), lb.data(), ub.data(), &stop, cost.data(), nev.data(),
                                                code.data());
    if (last_status_ != GTOP_OK) return true;
    for (int k = 0; k < nb; ++k) {
      const int b = order[lo + k];
      std::copy_n(&x[k * n], n, &x_[x_at_[b]]);
      min_cost_[b] = cost[k];
      nevals_[b] = nev[k];
    }
    lo = hi;
  }
  return true;
}

void GradTrajBatch::getCoefficient(int b, Matrix &coeff) const {
  if (b < 0 || b >= B_) {
    coeff.resize(0, 18);
    return;
  }
  const int m = m_of_[b], num_dp = 3 * m - 3;
  coeff.resize(m, 18);
  const double *df = &Df_[(size_t)b * 18], *x = &x_[x_at_[b]], *T = &T_[T_at_[b]];
  for (int a = 0; a < 3; ++a) {
    auto wp = [&](int j, int der) -> double {
      if (j == 0) return df[a * 6 + der];
      if (j == m) return df[a * 6 + 3 + der];
      return x[(size_t)a * num_dp + 3 * (j - 1) + der];
    };
    for (int s = 0; s < m; ++s) {
      const double d[6] = {wp(s, 0), wp(s + 1, 0), wp(s, 1), wp(s + 1, 1), wp(s, 2), wp(s + 1, 2)};
      double c[6];
      quintic_from_boundary(d, T[s], c);
      for (int j = 0; j < 6; ++j) coeff(s, 6 * a + j) = c[j];
    }
  }
}

}  // namespace gtop_amd
