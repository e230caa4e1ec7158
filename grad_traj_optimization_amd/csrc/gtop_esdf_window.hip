// gtop_esdf_window.hip — the reference's LOCAL map update on the device.
//
// compare2.cpp:147-152 / compare22.cpp:145-156 of the reference update a map the way an online planner does: only the
// box of space the sensor has just seen —
//     sdf_map.resetBuffer(min_pos, max_pos);           src/sdf_map.cpp:28-53   occupancy 0, distance 10000 in the box
//     sdf_map.setOccupancy(p) for every point;         :80-99
//     [sdf_map.setUpdateRange(min_pos, max_pos);]      :244-264                the same box as voxel indices
//     sdf_map.updateESDF3d();                          :310-368                the three sweeps over min_vec .. max_vec
// — with these consequences, all kept: the sweeps see only the window's part of every line (an obstacle outside the
// box casts no distance into it), distances outside the box keep their values, and inside it the result is
// min(res*sqrt(val), previous) with previous = 10000 after the reset (:355-361).
//
// Each sweep computes, per line segment, out(q) = min over v in the segment of (q - v)^2 + in(v) — the minimum the
// reference's lower-envelope pass (fillESDF, :266-308) finds; on this data every quantity is an exact integer, so the
// kernels take that minimum directly in int32 with an INF sentinel (see gtop_esdf.hip, whose whole-grid kernels this
// file's full-window case hands over to).  Because the sweeps never look outside the window, the update IS the
// whole-grid transform of the window taken alone: for a window of at least 12 x 12 x 3 voxels gtop_capi.cpp keeps its
// occupancy as a compact grid beside the map's (window_reset_compact_kernel, window_mark_kernel), runs gtop_esdf.hip's
// builder on it and writes the result back (window_scatter_kernel).  The plain kernels below — one lane per voxel of the window, lanes along z: the
// nearest occupied voxel of the column by an outward walk, then two outward scans with the exact cut-off
// d^2 >= best — serve the slivers (10x the time per voxel: no packed 16-bit scans, candidate lists or slab skipping).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "gtop_kernels.h"

namespace {

constexpr int kInf = 0x3fffffff;   // "no obstacle on this line segment" (as gtop_esdf.hip)

struct Window {
  int lo[3], hi[3];   // inclusive voxel indices
};

__device__ __forceinline__ bool window_voxel(const GtopGrid &g, const Window &w, int &x, int &y, int &z, size_t &idx) {
  const int wz = w.hi[2] - w.lo[2] + 1, wy = w.hi[1] - w.lo[1] + 1, wx = w.hi[0] - w.lo[0] + 1;
  const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (size_t)wx * wy * wz) return false;
  z = w.lo[2] + (int)(t % (size_t)wz);
  const size_t r = t / (size_t)wz;
  y = w.lo[1] + (int)(r % (size_t)wy);
  x = w.lo[0] + (int)(r / (size_t)wy);
  idx = ((size_t)x * g.ny + y) * g.nz + z;
  return true;
}

// resetBuffer(min_pos, max_pos), sdf_map.cpp:46-52
__global__ void __launch_bounds__(256)
window_reset_kernel(const GtopGrid g, const Window w, uint8_t *__restrict__ occ, double *__restrict__ dist) {
  int x, y, z;
  size_t idx;
  if (!window_voxel(g, w, x, y, z, idx)) return;
  occ[idx] = 0;
  dist[idx] = 10000.0;
}

// z sweep (:311-326) over z in [lo2, hi2]: squared distance to the nearest occupied voxel of the column segment
__global__ void __launch_bounds__(256)
window_z_kernel(const GtopGrid g, const Window w, const uint8_t *__restrict__ occ, int *__restrict__ out) {
  int x, y, z;
  size_t idx;
  if (!window_voxel(g, w, x, y, z, idx)) return;
  int best = kInf;
  for (int d = 0;; ++d) {
    const bool below = z - d >= w.lo[2], above = z + d <= w.hi[2];
    if (!below && !above) break;
    if ((below && occ[idx - d] == 1) || (above && occ[idx + d] == 1)) {
      best = d * d;
      break;
    }
  }
  out[idx] = best;
}

// y sweep (:328-345, STRIDE = nz, AXIS = 1) and x sweep (:347-363, STRIDE = ny nz, AXIS = 0; FINAL: res*sqrt, min with
// the previous distance): out(q) = min_v (q - v)^2 + in(v) over the window's segment of the line, by an outward scan
// from q with the exact cut-off d^2 >= best (every in(v) >= 0)
template <int AXIS, bool FINAL>
__global__ void __launch_bounds__(256)
window_scan_kernel(const GtopGrid g, const Window w, const int *__restrict__ in, int *__restrict__ out,
                   double *__restrict__ dist) {
  int c[3];
  size_t idx;
  if (!window_voxel(g, w, c[0], c[1], c[2], idx)) return;
  const size_t stride = AXIS == 1 ? (size_t)g.nz : (size_t)g.ny * g.nz;
  const int q = c[AXIS], lo = w.lo[AXIS], hi = w.hi[AXIS];
  int best = in[idx];
  for (int d = 1;; ++d) {
    const int d2 = d * d;
    if (d2 >= best) break;
    const bool below = q - d >= lo, above = q + d <= hi;
    if (!below && !above) break;
    if (below) {
      const int v = in[idx - (size_t)d * stride];
      if (v < kInf) best = min(best, d2 + v);   // (d <= 2^15, v < kInf: below 2^31)
    }
    if (above) {
      const int v = in[idx + (size_t)d * stride];
      if (v < kInf) best = min(best, d2 + v);
    }
  }
  if constexpr (FINAL) {
    // :355-361: min(res*sqrt(val), previous); a line segment without obstacles has val = DBL_MAX there: previous stays
    if (best < kInf) {
      const double r = g.res * sqrt((double)best);
      const double old = dist[idx];
      dist[idx] = r < old ? r : old;
    }
  } else {
    out[idx] = best;
  }
}

// The window's occupancy as a compact grid of its own, and the compact grid's distances back into the window: a
// windowed update IS the whole-grid transform of the window taken alone (its sweeps never look outside it), so windows
// that are large enough go through gtop_esdf.hip's optimised builder on the compact copy.
// Its reset and marking in one pass each (no gather pass over the window): the window's occupancy is cleared in the map AND in
// the compact copy (the distances need no reset there: the scatter rewrites every voxel of the window), and a point is
// marked in the map wherever it falls (setOccupancy, sdf_map.cpp:80-99) and in the compact copy when it falls inside.
__global__ void __launch_bounds__(256)
window_reset_compact_kernel(const GtopGrid g, const Window w, uint8_t *__restrict__ occ, uint8_t *__restrict__ sub) {
  int x, y, z;
  size_t idx;
  if (!window_voxel(g, w, x, y, z, idx)) return;
  occ[idx] = 0;
  sub[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = 0;
}

__global__ void __launch_bounds__(256)
window_mark_kernel(const GtopGrid g, const Window w, const double *__restrict__ pts, int npts, uint8_t *__restrict__ occ,
                   uint8_t *__restrict__ sub) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= npts) return;
  const double p[3] = {pts[3 * i], pts[3 * i + 1], pts[3 * i + 2]};
  for (int k = 0; k < 3; ++k)   // isInMap, sdf_map.cpp:55-69
    if (p[k] < g.min_range[k] + 1e-4 || p[k] > g.max_range[k] - 1e-4) return;
  const int ix = (int)floor((p[0] - g.origin[0]) * g.res_inv);   // posToIndex, sdf_map.cpp:71-74
  const int iy = (int)floor((p[1] - g.origin[1]) * g.res_inv);
  const int iz = (int)floor((p[2] - g.origin[2]) * g.res_inv);
  if (ix < 0 || iy < 0 || iz < 0 || ix >= g.nx || iy >= g.ny || iz >= g.nz) return;  // memory safety only
  occ[((size_t)ix * g.ny + iy) * g.nz + iz] = 1;   // sdf_map.cpp:97-98
  if (ix >= w.lo[0] && ix <= w.hi[0] && iy >= w.lo[1] && iy <= w.hi[1] && iz >= w.lo[2] && iz <= w.hi[2]) {
    const int wz = w.hi[2] - w.lo[2] + 1, wy = w.hi[1] - w.lo[1] + 1;
    sub[((size_t)(ix - w.lo[0]) * wy + (iy - w.lo[1])) * wz + (iz - w.lo[2])] = 1;
  }
}

__global__ void __launch_bounds__(256)
window_scatter_kernel(const GtopGrid g, const Window w, const double *__restrict__ sub, double *__restrict__ dist) {
  int x, y, z;
  size_t idx;
  if (!window_voxel(g, w, x, y, z, idx)) return;
  // :355-361: min(res*sqrt(val), previous), previous = 10000 after the reset; the compact build's values are <= 10000
  dist[idx] = sub[(size_t)blockIdx.x * blockDim.x + threadIdx.x];
}

}  // namespace

// reset + mark of the compact path (see the kernels): occupancy of the map and of the compact copy `sub`, no gather
hipError_t gtop_launch_esdf_window_reset_mark_compact(const GtopGrid &g, const int lo[3], const int hi[3], const double *pts,
                                                      int npts, uint8_t *occ, uint8_t *sub, hipStream_t stream) {
  Window w{{lo[0], lo[1], lo[2]}, {hi[0], hi[1], hi[2]}};
  const long long n = (long long)(hi[0] - lo[0] + 1) * (hi[1] - lo[1] + 1) * (hi[2] - lo[2] + 1);
  hipLaunchKernelGGL(window_reset_compact_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, g, w, occ, sub);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess || npts <= 0) return e;
  hipLaunchKernelGGL(window_mark_kernel, dim3((unsigned)((npts + 255) / 256)), dim3(256), 0, stream, g, w, pts, npts, occ, sub);
  return hipGetLastError();
}

hipError_t gtop_launch_esdf_window_scatter(const GtopGrid &g, const int lo[3], const int hi[3], const double *sub,
                                           double *dist, hipStream_t stream) {
  Window w{{lo[0], lo[1], lo[2]}, {hi[0], hi[1], hi[2]}};
  const long long n = (long long)(hi[0] - lo[0] + 1) * (hi[1] - lo[1] + 1) * (hi[2] - lo[2] + 1);
  hipLaunchKernelGGL(window_scatter_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, g, w, sub, dist);
  return hipGetLastError();
}

// lo / hi: the window as resetBuffer / setUpdateRange compute it (inclusive voxel indices, inside the grid).
// reset: resetBuffer(min, max).  Then the caller marks its points (gtop_launch_esdf_mark) and calls build.
hipError_t gtop_launch_esdf_window_reset(const GtopGrid &g, const int lo[3], const int hi[3], uint8_t *occ, double *dist,
                                         hipStream_t stream) {
  Window w{{lo[0], lo[1], lo[2]}, {hi[0], hi[1], hi[2]}};
  const long long n = (long long)(hi[0] - lo[0] + 1) * (hi[1] - lo[1] + 1) * (hi[2] - lo[2] + 1);
  if (hi[0] < lo[0] || hi[1] < lo[1] || hi[2] < lo[2]) return hipSuccess;
  hipLaunchKernelGGL(window_reset_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, g, w, occ, dist);
  return hipGetLastError();
}

hipError_t gtop_launch_esdf_window_build(const GtopGrid &g, const int lo[3], const int hi[3], const uint8_t *occ, int *tmp1,
                                         int *tmp2, double *dist, hipStream_t stream) {
  if (hi[0] < lo[0] || hi[1] < lo[1] || hi[2] < lo[2]) return hipSuccess;
  Window w{{lo[0], lo[1], lo[2]}, {hi[0], hi[1], hi[2]}};
  const long long n = (long long)(hi[0] - lo[0] + 1) * (hi[1] - lo[1] + 1) * (hi[2] - lo[2] + 1);
  const dim3 grid((unsigned)((n + 255) / 256)), block(256);
  hipLaunchKernelGGL(window_z_kernel, grid, block, 0, stream, g, w, occ, tmp1);
  hipLaunchKernelGGL((window_scan_kernel<1, false>), grid, block, 0, stream, g, w, (const int *)tmp1, tmp2, dist);
  hipLaunchKernelGGL((window_scan_kernel<0, true>), grid, block, 0, stream, g, w, (const int *)tmp2, (int *)nullptr, dist);
  return hipGetLastError();
}
