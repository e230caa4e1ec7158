// mma.hpp — bound-constrained CCSA-MMA driver (host side).
//
// The reference hands its callback to NLopt algorithm 24, LD_MMA
// (src/grad_traj_optimizer.cpp:137-140, launch/opti_node.launch:3), with box
// bounds only (:151-179) and a wall-clock stop (:144-148).  NLopt is a
// third-party dependency that is absent from this image (no headers; the
// binary vendored in the reference's lib/ is not loaded), so this file
// restates the published algorithm it implements: K. Svanberg, "A class of
// globally convergent optimization methods based on conservative convex
// separable approximations", SIAM J. Optim. 12 (2002), in the MMA form of
// NLopt 2.5.0 (src/algs/mma/mma.c) specialised to zero nonlinear constraints,
// where the dual problem is empty and every inner iteration is the closed-form
// separable minimiser.  PARITY UNPINNED for the optimizer trajectory (NLopt
// cannot be run here); the callback it drives is parity-checked separately.
#ifndef GTOP_AMD_MMA_HPP_
#define GTOP_AMD_MMA_HPP_

#include <chrono>
#include <cmath>
#include <cstring>
#include <vector>

namespace gtop_amd {

// nlopt_func
typedef double (*mma_objective)(unsigned n, const double *x, double *grad, void *data);

struct MmaOptions {
  double maxtime = 0.0;    // seconds, 0 = none  (set_maxtime, :144-148)
  int maxeval = 0;         // 0 = none
  double ftol_rel = 0.0, xtol_rel = 0.0;
};

enum MmaCode {             // nlopt_result values
  MMA_FAILURE = -1,
  MMA_SUCCESS = 1,
  MMA_FTOL_REACHED = 3,
  MMA_XTOL_REACHED = 4,
  MMA_MAXEVAL_REACHED = 5,
  MMA_MAXTIME_REACHED = 6
};

struct MmaResult {
  int code = MMA_FAILURE;
  double minf = HUGE_VAL;
  int nevals = 0;
};

// NLopt's relstop (stop.c) without the absolute tolerance the reference never sets
inline bool mma_relstop(double vold, double vnew, double reltol) {
  if (std::isinf(vold)) return false;
  return std::fabs(vnew - vold) < reltol * (std::fabs(vnew) + std::fabs(vold)) * 0.5 || (reltol > 0 && vnew == vold);
}

// One separable step from base point x with gradient dfdx, asymptote widths
// sigma and conservativeness rho: fills xcur and returns the approximant's
// value g(xcur) and w(xcur) = 0.5 * sum dx^2 / (sigma^2 - dx^2).
inline void mma_separable_step(unsigned n, const double *x, const double *dfdx, const double *sigma,
                               double rho, const double *lb, const double *ub, double fval,
                               double *xcur, double *gval, double *wval) {
  double g = fval, w = 0.0;
  for (unsigned j = 0; j < n; ++j) {
    if (sigma[j] == 0.0) {   // lb == ub
      xcur[j] = x[j];
      continue;
    }
    const double sigma2 = sigma[j] * sigma[j];
    double u = dfdx[j];
    const double v = std::fabs(dfdx[j]) * sigma[j] + 0.5 * rho;
    u *= sigma2;
    // root of u dx^2 + 2 v sigma^2 dx + u sigma^2 = 0 with |dx| <= sigma
    double dx = (u / v) / (-1.0 - std::sqrt(std::fabs(1.0 - (u / (v * sigma[j])) * (u / (v * sigma[j])))));
    xcur[j] = x[j] + dx;
    if (xcur[j] > ub[j]) xcur[j] = ub[j];
    else if (xcur[j] < lb[j]) xcur[j] = lb[j];
    if (xcur[j] > x[j] + 0.9 * sigma[j]) xcur[j] = x[j] + 0.9 * sigma[j];
    else if (xcur[j] < x[j] - 0.9 * sigma[j]) xcur[j] = x[j] - 0.9 * sigma[j];
    dx = xcur[j] - x[j];
    const double dx2 = dx * dx;
    const double denominv = 1.0 / (sigma2 - dx2);
    g += (dfdx[j] * (sigma2 * dx) + (std::fabs(dfdx[j]) * sigma[j] + 0.5 * rho) * dx2) * denominv;
    w += 0.5 * dx2 * denominv;
  }
  *gval = g;
  *wval = w;
}

inline MmaResult mma_minimize(unsigned n, mma_objective f, void *f_data, const double *lb,
                              const double *ub, double *x, const MmaOptions &opt) {
  MmaResult res;
  const auto t_start = std::chrono::steady_clock::now();
  auto elapsed = [&]() {
    return std::chrono::duration<double>(std::chrono::steady_clock::now() - t_start).count();
  };
  auto stop_now = [&]() -> int {
    if (opt.maxeval > 0 && res.nevals >= opt.maxeval) return MMA_MAXEVAL_REACHED;
    if (opt.maxtime > 0 && elapsed() >= opt.maxtime) return MMA_MAXTIME_REACHED;
    return 0;
  };
  std::vector<double> sigma(n), dfdx(n), dfdx_cur(n), xcur(n), xprev(n), xprevprev(n);
  for (unsigned j = 0; j < n; ++j) {
    if (std::isinf(ub[j]) || std::isinf(lb[j])) sigma[j] = 1.0;
    else sigma[j] = 0.5 * (ub[j] - lb[j]);
    if (x[j] < lb[j]) x[j] = lb[j];       // nlopt clamps the start into the box
    if (x[j] > ub[j]) x[j] = ub[j];
  }
  double rho = 1.0;
  double fcur = f(n, x, dfdx.data(), f_data);
  res.nevals++;
  res.minf = fcur;
  std::memcpy(xcur.data(), x, sizeof(double) * n);
  // A non-finite value is not an exit, as in NLopt's mma.c (and in the device loop, gtop_device_common.h): +inf from an
  // exp that overflowed on a long step is a rejected step — rho grows and the next one is shorter; NaN is rejected
  // with rho unchanged, so the same point is evaluated until a limit ends the run.  Only a NaN with NO limit set
  // could never end: that alone is reported as a failure.
  auto hopeless = [&](double f) { return std::isnan(f) && opt.maxeval <= 0 && opt.maxtime <= 0; };
  if (hopeless(fcur)) { res.code = MMA_FAILURE; return res; }
  int k = 0, ret = MMA_SUCCESS;
  const double kRhoMin = 1e-5;

  while (true) {   // outer iterations
    const double fprev = fcur;
    if (int s = stop_now()) { ret = s; break; }
    if (++k > 1) xprevprev = xprev;
    xprev = xcur;

    while (true) {   // inner iterations: make the approximation conservative
      double gval, wval;
      mma_separable_step(n, x, dfdx.data(), sigma.data(), rho, lb, ub, res.minf, xcur.data(), &gval, &wval);
      fcur = f(n, xcur.data(), dfdx_cur.data(), f_data);
      res.nevals++;
      if (hopeless(fcur)) { res.code = MMA_FAILURE; return res; }
      const bool inner_done = gval >= fcur;
      if (fcur < res.minf) {   // accept: new base point
        res.minf = fcur;
        std::memcpy(x, xcur.data(), sizeof(double) * n);
        dfdx = dfdx_cur;
      }
      // NLopt's order (mma.c): the evaluation and time limits are looked at right after every evaluation, BEFORE the
      // inner loop may end — an evaluation that is the last one allowed reports MAXEVAL even where it also completes
      // an outer iteration that meets ftol / xtol
      if (int s = stop_now()) { ret = s; break; }
      if (inner_done) break;
      if (fcur > gval) rho = std::fmin(10 * rho, 1.1 * (rho + (fcur - gval) / wval));
    }
    if (ret != MMA_SUCCESS) break;

    // nlopt_stop_ftol / nlopt_stop_x (stop.c, relstop): |new - old| < tol * (|new| + |old|) / 2, or new == old with a
    // tolerance set (catches new == old == 0); x is tested after f and its verdict stands when both hold
    if (mma_relstop(fprev, fcur, opt.ftol_rel)) ret = MMA_FTOL_REACHED;
    if (opt.xtol_rel > 0) {
      bool all = true;
      for (unsigned j = 0; j < n && all; ++j) all = mma_relstop(xprev[j], xcur[j], opt.xtol_rel);
      if (all) ret = MMA_XTOL_REACHED;
    }
    if (ret != MMA_SUCCESS) break;

    // asymptote and rho update for outer iteration k+1
    rho = std::fmax(0.1 * rho, kRhoMin);
    if (k > 1) {
      for (unsigned j = 0; j < n; ++j) {
        const double dx2 = (xcur[j] - xprev[j]) * (xprev[j] - xprevprev[j]);
        const double gam = dx2 < 0 ? 0.7 : (dx2 > 0 ? 1.2 : 1.0);
        sigma[j] *= gam;
        if (!std::isinf(ub[j]) && !std::isinf(lb[j])) {
          sigma[j] = std::fmin(sigma[j], 10 * (ub[j] - lb[j]));
          sigma[j] = std::fmax(sigma[j], 0.01 * (ub[j] - lb[j]));
        }
      }
    }
  }
  res.code = ret;
  return res;
}

}  // namespace gtop_amd

#endif  // GTOP_AMD_MMA_HPP_
