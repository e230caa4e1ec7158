// gtop_mma.hip — batched CCSA-MMA update on gfx950 (stand-alone launch form).
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

#include "gtop_device_common.h"
#include "gtop_kernels.h"

namespace {

__global__ void __launch_bounds__(64)
mma_update_kernel(GtopMmaState st, int B, int n, const double *__restrict__ fcur_all,
                  const double *__restrict__ gcur_all) {
  const int b = blockIdx.x;
  if (b >= B) return;
  gtop_mma_update_trajectory(st, b, n, threadIdx.x, fcur_all[b], gcur_all + (size_t)b * n);
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
    st.fprev[b] = 0.0;
    st.k[b] = 0;
    st.state[b] = 0;
    st.nevals[b] = 0;
  }
}

// what the caller sees of the stop state: nlopt_result-style code (a trajectory still running when the loop ended
// has used up its evaluations: MAXEVAL_REACHED) and the evaluations it consumed
__global__ void __launch_bounds__(256)
mma_finish_kernel(GtopMmaState st, int B, int *__restrict__ code, int *__restrict__ nevals) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  if (code) code[b] = st.state[b] >= 3 ? st.state[b] : GTOP_MMA_MAXEVAL_REACHED;
  if (nevals) nevals[b] = st.nevals[b];
}

}  // namespace

hipError_t gtop_launch_mma_finish(const GtopMmaState &st, int B, int *code, int *nevals, hipStream_t stream) {
  if (B <= 0 || (!code && !nevals)) return hipSuccess;
  hipLaunchKernelGGL(mma_finish_kernel, dim3((B + 255) / 256), dim3(256), 0, stream, st, B, code, nevals);
  return hipGetLastError();
}

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
