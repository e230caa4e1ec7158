// gtop_mma.hip — lock-step batched CCSA-MMA update on gfx950.
//
// Replaces, for B independent problems at once, the optimizer loop that the
// reference runs one problem at a time through NLopt LD_MMA
// (src/grad_traj_optimizer.cpp:137-195 of EpicOne1/grad_traj_optimization:
// nlopt::opt(algorithm(24)), set_lower/upper_bounds, optimize()).  With box
// bounds only, the MMA dual problem is empty and an inner iteration is the
// closed-form separable minimiser, so one optimizer iteration per trajectory is
// elementwise work over its n variables plus two sums — one wavefront per
// trajectory, between two launches of the cost/gradient kernel.
// Algorithm: Svanberg 2002 (CCSA) as published in NLopt 2.5.0's mma.c; the
// host restatement (same arithmetic, one problem) is csrc/mma.hpp.
//
// HBM-bound: per iteration and trajectory it streams 7 vectors of n doubles.

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "gtop_kernels.h"

namespace {

__device__ __forceinline__ double wsum(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

// One separable step for this lane's variables: fills xcur, returns the
// lane-partial sums of g(xcur) - f and w(xcur).   (mma.hpp: mma_separable_step)
__device__ __forceinline__ void separable_step(int n, int lane, const double *x, const double *dfdx,
                                               const double *sigma, double rho, const double *lb,
                                               const double *ub, double *xcur, double &g, double &w) {
  g = 0.0;
  w = 0.0;
  for (int j = lane; j < n; j += 64) {
    const double sg = sigma[j], xj = x[j];
    if (sg == 0.0) {
      xcur[j] = xj;
      continue;
    }
    const double df = dfdx[j];
    const double sigma2 = sg * sg;
    const double u = df * sigma2;
    const double v = fabs(df) * sg + 0.5 * rho;
    const double q = u / (v * sg);
    double dx = (u / v) / (-1.0 - sqrt(fabs(1.0 - q * q)));
    double xc = xj + dx;
    xc = xc > ub[j] ? ub[j] : (xc < lb[j] ? lb[j] : xc);
    const double hi = xj + 0.9 * sg, lo = xj - 0.9 * sg;
    xc = xc > hi ? hi : (xc < lo ? lo : xc);
    xcur[j] = xc;
    dx = xc - xj;
    const double dx2 = dx * dx;
    const double denominv = 1.0 / (sigma2 - dx2);
    g += (df * (sigma2 * dx) + (fabs(df) * sg + 0.5 * rho) * dx2) * denominv;
    w += 0.5 * dx2 * denominv;
  }
}

// state[b]: 0 = first evaluation pending, 1 = inside an inner loop
__global__ void __launch_bounds__(64)
mma_update_kernel(GtopMmaState st, int B, int n, const double *__restrict__ fcur_all,
                  const double *__restrict__ gcur_all) {
  const int b = blockIdx.x;
  if (b >= B) return;
  const int lane = threadIdx.x;
  const size_t o = (size_t)b * n;
  double *x = st.x + o, *xcur = st.xcur + o, *xprev = st.xprev + o, *xprevprev = st.xprevprev + o;
  double *dfdx = st.dfdx + o, *sigma = st.sigma + o;
  const double *lb = st.lb + o, *ub = st.ub + o, *gcur = gcur_all + o;
  const double fcur = fcur_all[b];
  double rho = st.rho[b], minf = st.minf[b];
  int k = st.k[b];
  const int state = st.state[b];
  bool new_outer;

  if (state == 0) {
    // f(x0): base point = start (mma.hpp: first evaluation)
    minf = fcur;
    for (int j = lane; j < n; j += 64) {
      x[j] = xcur[j];
      dfdx[j] = gcur[j];
    }
    new_outer = true;
  } else {
    const double gval = st.gval[b], wval = st.wval[b];
    const bool inner_done = gval >= fcur;
    if (fcur < minf) {   // accept: new base point
      minf = fcur;
      for (int j = lane; j < n; j += 64) {
        x[j] = xcur[j];
        dfdx[j] = gcur[j];
      }
    }
    if (inner_done) {
      // end of the outer iteration: relax rho, adapt the asymptotes
      rho = fmax(0.1 * rho, 1e-5);
      if (k > 1) {
        for (int j = lane; j < n; j += 64) {
          const double dx2 = (xcur[j] - xprev[j]) * (xprev[j] - xprevprev[j]);
          const double gam = dx2 < 0 ? 0.7 : (dx2 > 0 ? 1.2 : 1.0);
          double s = sigma[j] * gam;
          const double range = ub[j] - lb[j];
          if (!isinf(ub[j]) && !isinf(lb[j])) {
            s = fmin(s, 10 * range);
            s = fmax(s, 0.01 * range);
          }
          sigma[j] = s;
        }
      }
      new_outer = true;
    } else {
      if (fcur > gval) rho = fmin(10 * rho, 1.1 * (rho + (fcur - gval) / wval));
      new_outer = false;
    }
  }
  if (new_outer) {
    ++k;
    for (int j = lane; j < n; j += 64) {
      if (k > 1) xprevprev[j] = xprev[j];
      xprev[j] = xcur[j];
    }
  }
  // every lane's writes above are to its own j; the step below reads x/dfdx/sigma
  // at the same j only, so no cross-lane hazard
  double g, w;
  separable_step(n, lane, x, dfdx, sigma, rho, lb, ub, xcur, g, w);
  g = wsum(g);
  w = wsum(w);
  if (lane == 0) {
    st.gval[b] = minf + g;
    st.wval[b] = w;
    st.rho[b] = rho;
    st.minf[b] = minf;
    st.k[b] = k;
    st.state[b] = 1;
  }
}

__global__ void __launch_bounds__(256)
mma_init_kernel(GtopMmaState st, int B, int n, const double *__restrict__ x0) {
  const size_t total = (size_t)B * n;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const double lo = st.lb[i], hi = st.ub[i];
    st.sigma[i] = (isinf(lo) || isinf(hi)) ? 1.0 : 0.5 * (hi - lo);
    double v = x0[i];
    v = v < lo ? lo : (v > hi ? hi : v);   // nlopt clamps the start into the box
    st.xcur[i] = v;
    st.x[i] = v;
    st.xprev[i] = v;
    st.xprevprev[i] = v;
    st.dfdx[i] = 0.0;
  }
  for (size_t b = (size_t)blockIdx.x * blockDim.x + threadIdx.x; b < (size_t)B; b += (size_t)gridDim.x * blockDim.x) {
    st.rho[b] = 1.0;
    st.minf[b] = 0.0;
    st.gval[b] = 0.0;
    st.wval[b] = 0.0;
    st.k[b] = 0;
    st.state[b] = 0;
  }
}

}  // namespace

hipError_t gtop_launch_mma_init(const GtopMmaState &st, int B, int n, const double *x0, hipStream_t stream) {
  if (B <= 0) return hipSuccess;
  hipLaunchKernelGGL(mma_init_kernel, dim3(1024), dim3(256), 0, stream, st, B, n, x0);
  return hipGetLastError();
}

hipError_t gtop_launch_mma_update(const GtopMmaState &st, int B, int n, const double *fcur, const double *gcur,
                                  hipStream_t stream) {
  if (B <= 0) return hipSuccess;
  hipLaunchKernelGGL(mma_update_kernel, dim3(B), dim3(64), 0, stream, st, B, n, fcur, gcur);
  return hipGetLastError();
}
