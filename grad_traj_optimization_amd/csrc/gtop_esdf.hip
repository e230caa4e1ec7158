// gtop_esdf.hip — Euclidean distance field construction on gfx950.
//
// Replaces SDFMap::resetBuffer / setOccupancy / updateESDF3d
// (src/sdf_map.cpp:26-53, :80-99, :310-368 of EpicOne1/grad_traj_optimization):
// three 1-D lower-envelope (Felzenszwalb–Huttenlocher) sweeps, z then y then x,
// then dist = min(res*sqrt(val), previous).  Integer/byte work plus exactly
// rounded fp64 (+, -, /, sqrt), so the result is bit-identical to the CPU
// restatement.  HBM-bound: one lane per grid line; in the y and x sweeps
// neighbouring lanes walk neighbouring z columns, so every step of the sweep
// is a coalesced row; the per-line envelope scratch is laid out [k][line] for
// the same reason.

#include <hip/hip_runtime.h>
#include <float.h>
#include <stdint.h>

#include "gtop_kernels.h"

namespace {

__global__ void __launch_bounds__(256)
esdf_reset_kernel(uint8_t *__restrict__ occ, double *__restrict__ dist, size_t nvox) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (; i < nvox; i += stride) {
    occ[i] = 0;
    dist[i] = 10000.0;   // sdf_map.cpp:22, :51
  }
}

__global__ void __launch_bounds__(256)
esdf_mark_kernel(const GtopGrid g, const double *__restrict__ pts, int npts,
                 uint8_t *__restrict__ occ) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= npts) return;
  const double p[3] = {pts[3 * i], pts[3 * i + 1], pts[3 * i + 2]};
  // isInMap, sdf_map.cpp:55-69
  for (int k = 0; k < 3; ++k)
    if (p[k] < g.min_range[k] + 1e-4 || p[k] > g.max_range[k] - 1e-4) return;
  // posToIndex, sdf_map.cpp:71-74
  const int ix = (int)floor((p[0] - g.origin[0]) * g.res_inv);
  const int iy = (int)floor((p[1] - g.origin[1]) * g.res_inv);
  const int iz = (int)floor((p[2] - g.origin[2]) * g.res_inv);
  if (ix < 0 || iy < 0 || iz < 0 || ix >= g.nx || iy >= g.ny || iz >= g.nz) return;  // memory safety only
  occ[((size_t)ix * g.ny + iy) * g.nz + iz] = 1;   // sdf_map.cpp:97-98
}

// One lane per line.  PASS 0: z sweep reading occupancy; 1: y sweep; 2: x sweep
// with the final min(res*sqrt(.), old) (sdf_map.cpp:355-361).
// fillESDF, sdf_map.cpp:266-308, with start = 0, end = n-1.
template <int PASS>
__global__ void __launch_bounds__(256)
esdf_sweep_kernel(const GtopGrid g, const uint8_t *__restrict__ occ,
                  const double *__restrict__ fin, double *__restrict__ fout,
                  int *__restrict__ vws, double *__restrict__ zws,
                  size_t line0, size_t nlines_chunk, size_t ws_lines) {
  const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= nlines_chunk) return;
  const size_t line = line0 + t;
  const size_t nyz = (size_t)g.ny * g.nz;
  size_t base, stride;
  int n;
  if (PASS == 0) {          // line = x*ny + y
    base = line * g.nz; stride = 1; n = g.nz;
  } else if (PASS == 1) {   // line = x*nz + z
    const size_t x = line / g.nz, z = line - x * g.nz;
    base = x * nyz + z; stride = g.nz; n = g.ny;
  } else {                  // line = y*nz + z
    base = line; stride = nyz; n = g.nx;
  }
  auto f = [&](int q) -> double {
    if (PASS == 0) return occ[base + (size_t)q * stride] == 1 ? 0.0 : DBL_MAX;  // :314-318
    return fin[base + (size_t)q * stride];
  };
  // scratch: v[k], z[k] at [k*ws_lines + t]
  auto V = [&](int k) -> int & { return vws[(size_t)k * ws_lines + t]; };
  auto Z = [&](int k) -> double & { return zws[(size_t)k * ws_lines + t]; };

  int k = 0;
  V(0) = 0;
  Z(0) = -DBL_MAX;
  Z(1) = DBL_MAX;
  int vk = 0;            // v[k] kept in a register
  double fvk = f(0);     // f(v[k])
  for (int q = 1; q < n; q++) {
    const double fq = f(q);
    double s;
    k++;
    do {
      k--;
      vk = V(k);
      fvk = f(vk);
      s = ((fq + q * q) - (fvk + vk * vk)) / (2 * q - 2 * vk);
    } while (s <= Z(k));
    k++;
    V(k) = q;
    Z(k) = s;
    Z(k + 1) = DBL_MAX;
  }
  k = 0;
  vk = V(0);
  fvk = f(vk);
  double znext = Z(1);
  for (int q = 0; q < n; q++) {
    while (znext < q) {
      k++;
      vk = V(k);
      fvk = f(vk);
      znext = Z(k + 1);
    }
    const double val = (q - vk) * (q - vk) + fvk;
    const size_t o = base + (size_t)q * stride;
    if (PASS == 2) {
      const double d = g.res * sqrt(val);
      const double old = fout[o];
      fout[o] = d < old ? d : old;
    } else {
      fout[o] = val;
    }
  }
}

}  // namespace

hipError_t gtop_launch_esdf_reset(uint8_t *occ, double *dist, size_t nvox, hipStream_t stream) {
  hipLaunchKernelGGL(esdf_reset_kernel, dim3(2048), dim3(256), 0, stream, occ, dist, nvox);
  return hipGetLastError();
}

hipError_t gtop_launch_esdf_mark(const GtopGrid &g, const double *pts, int npts, uint8_t *occ,
                                 hipStream_t stream) {
  if (npts <= 0) return hipSuccess;
  hipLaunchKernelGGL(esdf_mark_kernel, dim3((npts + 255) / 256), dim3(256), 0, stream, g, pts, npts, occ);
  return hipGetLastError();
}

size_t gtop_esdf_ws_lines(const GtopGrid &g) {
  size_t l0 = (size_t)g.nx * g.ny, l1 = (size_t)g.nx * g.nz, l2 = (size_t)g.ny * g.nz;
  size_t mx = l0 > l1 ? l0 : l1;
  mx = mx > l2 ? mx : l2;
  const size_t cap = 65536;   // lines processed per launch
  return mx < cap ? mx : cap;
}

hipError_t gtop_launch_esdf_build(const GtopGrid &g, const uint8_t *occ, double *tmp1, double *tmp2,
                                  double *dist, int *vws, double *zws, size_t ws_lines,
                                  hipStream_t stream) {
  const size_t lines[3] = {(size_t)g.nx * g.ny, (size_t)g.nx * g.nz, (size_t)g.ny * g.nz};
  for (int pass = 0; pass < 3; ++pass) {
    for (size_t l0 = 0; l0 < lines[pass]; l0 += ws_lines) {
      const size_t nl = lines[pass] - l0 < ws_lines ? lines[pass] - l0 : ws_lines;
      const dim3 grid((unsigned)((nl + 255) / 256)), block(256);
      if (pass == 0)
        hipLaunchKernelGGL(esdf_sweep_kernel<0>, grid, block, 0, stream, g, occ, (const double *)nullptr,
                           tmp1, vws, zws, l0, nl, ws_lines);
      else if (pass == 1)
        hipLaunchKernelGGL(esdf_sweep_kernel<1>, grid, block, 0, stream, g, occ, (const double *)tmp1,
                           tmp2, vws, zws, l0, nl, ws_lines);
      else
        hipLaunchKernelGGL(esdf_sweep_kernel<2>, grid, block, 0, stream, g, occ, (const double *)tmp2,
                           dist, vws, zws, l0, nl, ws_lines);
      hipError_t e = hipGetLastError();
      if (e != hipSuccess) return e;
    }
  }
  return hipSuccess;
}
