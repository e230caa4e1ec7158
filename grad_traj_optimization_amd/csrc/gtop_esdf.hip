// gtop_esdf.hip — Euclidean distance field construction on gfx950.
//
// Replaces SDFMap::resetBuffer / setOccupancy / updateESDF3d
// (src/sdf_map.cpp:26-53, :80-99, :310-368 of EpicOne1/grad_traj_optimization).
// The reference runs three 1-D lower-envelope (Felzenszwalb–Huttenlocher)
// sweeps, z then y then x, over doubles with DBL_MAX as "no obstacle", then
// dist = min(res*sqrt(val), previous).  Each sweep computes, per line,
//     out(q) = min_v ( (q - v)^2 + in(v) )
// and on this data every quantity is an exact integer (squared voxel
// distances; the envelope's intersection abscissae are ratios of small
// integers, so their rounding cannot flip a comparison), i.e. the sweep output
// IS that exact minimum.  The kernels below compute the same minimum directly,
// one lane per voxel, in int32 with an INF sentinel:
//   z sweep : the input is the occupancy itself, so out(q) = (distance to the
//             nearest occupied voxel of the column)^2 — found with wave
//             ballots (one 64-bit mask per 64 voxels of the column) and
//             clz/ctz, no search loop;
//   y sweep : after the z sweep a voxel of line (x, ., z) is finite exactly when its
//             (x,y) column holds an obstacle, whatever z — so every row x has ONE
//             sorted list of candidate columns (built by esdf_rows_kernel with
//             ballots/popcounts).  A voxel walks that list outward from its own y,
//             four candidates per round trip, with the exact cut-off d^2 >= best;
//             obstacle-free stretches cost nothing;
//   x sweep : outward scan v = q, q±1, q±2, ... with the same cut-off (in(v) >= 0),
//             neighbouring lanes on neighbouring z so every step is a coalesced
//             row, 4 steps per round trip — the scan is latency-bound, not
//             bandwidth-bound;
//             it also applies the final res*sqrt(.) (exactly rounded fp64, as the
//             reference's) and writes the fp32 copy used by the GTOP_F32 path.
// Result: bit-identical to the CPU restatement and to scipy's exact EDT
// (tests), HBM/latency-bound integer work, no scratch workspace.  (An LDS-tiled
// variant of the y/x scans was measured 4-6x SLOWER: 3 200 long-running
// wavefronts instead of 125 000 short ones; profiles/r1/esdf_kernels.txt.)

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "gtop_kernels.h"

namespace {

constexpr int kInf = 0x3fffffff;   // "no obstacle on this line so far"; d^2 + kInf stays below 2^31 for lines <= 2^15

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

// z sweep (sdf_map.cpp:311-326): one wavefront per (x,y) column.
constexpr int kMaxChunks = 64;   // columns up to 4096 voxels

__global__ void __launch_bounds__(256)
esdf_z_kernel(const GtopGrid g, const uint8_t *__restrict__ occ, int *__restrict__ out,
              uint8_t *__restrict__ colany) {
  __shared__ unsigned long long masks[4][kMaxChunks];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const size_t ncol = (size_t)g.nx * g.ny;
  const int nz = g.nz, nchunk = (nz + 63) >> 6;
  for (size_t col = (size_t)blockIdx.x * 4 + w; col < ncol; col += (size_t)gridDim.x * 4) {
    const uint8_t *c = occ + col * nz;
    unsigned long long any = 0ull;
    for (int k = 0; k < nchunk; ++k) {
      const int z = k * 64 + lane;
      const bool o = (z < nz) && (c[z] == 1);
      const unsigned long long mk = __ballot(o);
      if (lane == 0) masks[w][k] = mk;
      any |= mk;
    }
    if (lane == 0) colany[col] = any != 0ull;   // the column holds an obstacle: finite for the y sweep
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    for (int ko = 0; ko < nchunk; ++ko) {
      const int z = ko * 64 + lane;
      int best = kInf;   // distance in voxels to the nearest occupied voxel of the column
      for (int k = 0; k < nchunk; ++k) {
        const unsigned long long mk = masks[w][k];
        if (mk == 0) continue;   // wave-uniform
        int d;
        if (k < ko) {
          d = z - (k * 64 + 63 - __clzll((long long)mk));        // its highest occupied voxel
        } else if (k > ko) {
          d = k * 64 + (__ffsll((long long)mk) - 1) - z;          // its lowest occupied voxel
        } else {
          const unsigned long long below = mk & (lane == 63 ? ~0ull : ((2ull << lane) - 1ull));   // bits <= lane
          const unsigned long long above = mk & (~0ull << lane);                                     // bits >= lane
          const int d1 = below ? lane - (63 - __clzll((long long)below)) : kInf;
          const int d2 = above ? (__ffsll((long long)above) - 1) - lane : kInf;
          d = d1 < d2 ? d1 : d2;
        }
        best = d < best ? d : best;
      }
      if (z < nz) out[col * nz + z] = best >= kInf ? kInf : best * best;
    }
    __builtin_amdgcn_wave_barrier();
  }
}

// Candidate columns of every row x for the y sweep: cols[x][0..cnt[x]) = the y of the
// columns that hold an obstacle, ascending; rank[x][y] = number of them below y
// (= index of the first candidate at or above y).  One wavefront per row.
__global__ void __launch_bounds__(64)
esdf_rows_kernel(const GtopGrid g, const uint8_t *__restrict__ colany, int *__restrict__ cols,
                 int *__restrict__ rank, int *__restrict__ cnt) {
  const int lane = threadIdx.x, ny = g.ny;
  for (int x = blockIdx.x; x < g.nx; x += gridDim.x) {
    int base = 0;
    for (int y0 = 0; y0 < ny; y0 += 64) {
      const int y = y0 + lane;
      const bool f = (y < ny) && colany[(size_t)x * ny + y];
      const unsigned long long mk = __ballot(f);
      const int pos = base + __popcll(mk & ((1ull << lane) - 1ull));
      if (y < ny) rank[(size_t)x * ny + y] = pos;
      if (f) cols[(size_t)x * ny + pos] = y;
      base += __popcll(mk);
    }
    if (lane == 0) cnt[x] = base;
  }
}

// y sweep (sdf_map.cpp:328-346): out(x,y,z) = min over candidate columns v of (y-v)^2 + in(x,v,z).
__global__ void __launch_bounds__(256)
esdf_y_kernel(const GtopGrid g, const int *__restrict__ fin, int *__restrict__ fout, const int *__restrict__ cols,
              const int *__restrict__ rank, const int *__restrict__ cnt) {
  constexpr int U = 4;   // candidates per round trip and side
  const size_t nvox = (size_t)g.nx * g.ny * g.nz;
  const size_t nyz = (size_t)g.ny * g.nz;
  const int ny = g.ny, nz = g.nz;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < nvox; i += (size_t)gridDim.x * blockDim.x) {
    const int x = (int)(i / nyz);
    const size_t r = i - (size_t)x * nyz;
    const int q = (int)(r / nz);
    const int *line = fin + (i - (size_t)q * nz);   // (x, 0, z)
    const int *cx = cols + (size_t)x * ny;
    const int c = cnt[x];
    int best = fin[i];
    const int k0 = rank[(size_t)x * ny + q];        // first candidate at or above q
    // below q: candidates k0-1, k0-2, ... (descending y, ascending distance)
    for (int k = k0 - 1; k >= 0; k -= U) {
      const int d0 = q - cx[k];
      if (d0 * d0 >= best) break;   // in(v) >= 0: nothing farther can win
      int v[U], f[U];
#pragma unroll
      for (int u = 0; u < U; ++u) v[u] = (k - u >= 0) ? cx[k - u] : -1;
#pragma unroll
      for (int u = 0; u < U; ++u) f[u] = (v[u] >= 0) ? line[(size_t)v[u] * nz] : kInf;
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int d = q - v[u];
        const int cand = (v[u] >= 0) ? d * d + f[u] : kInf;   // d <= 2^15, f <= kInf: below 2^31
        best = cand < best ? cand : best;
      }
    }
    // above q (the voxel's own column, if it is a candidate, is `best` already)
    for (int k = k0 + ((k0 < c && cx[k0] == q) ? 1 : 0); k < c; k += U) {
      const int d0 = cx[k] - q;
      if (d0 * d0 >= best) break;
      int v[U], f[U];
#pragma unroll
      for (int u = 0; u < U; ++u) v[u] = (k + u < c) ? cx[k + u] : -1;
#pragma unroll
      for (int u = 0; u < U; ++u) f[u] = (v[u] >= 0) ? line[(size_t)v[u] * nz] : kInf;
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int d = v[u] - q;
        const int cand = (v[u] >= 0) ? d * d + f[u] : kInf;
        best = cand < best ? cand : best;
      }
    }
    fout[i] = best > kInf ? kInf : best;
  }
}

// x sweep (sdf_map.cpp:348-364): one lane per voxel, lanes along z.
// out(q) = min_v ((q-v)^2 + in(v)), scanning outward; then dist = min(res*sqrt(out), previous).
__global__ void __launch_bounds__(256)
esdf_x_kernel(const GtopGrid g, const int *__restrict__ fin, double *__restrict__ dist, float *__restrict__ dist32) {
  constexpr int kScanBatch = 4;   // steps per round trip (4 beats 8 and 16 at 400^3; profiles/r1/esdf_kernels.txt)
  const size_t nvox = (size_t)g.nx * g.ny * g.nz;
  const size_t nyz = (size_t)g.ny * g.nz;
  const int n = g.nx;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < nvox; i += (size_t)gridDim.x * blockDim.x) {
    const int q = (int)(i / nyz);
    const int *line = fin + (i - (size_t)q * nyz);
    int best = line[(size_t)q * nyz];
    const int reach = q > n - 1 - q ? q : n - 1 - q;
    // The loads of a batch are independent and issue together; entries past the exact
    // cut-off d^2 >= best cannot win (in(v) >= 0), so reading a few of them changes nothing.
    for (int d0 = 1; d0 <= reach; d0 += kScanBatch) {
      if (d0 * d0 >= best) break;
      int lo[kScanBatch], hi[kScanBatch];
#pragma unroll
      for (int u = 0; u < kScanBatch; ++u) {
        const int d = d0 + u;
        lo[u] = (q - d >= 0) ? line[(size_t)(q - d) * nyz] : kInf;
        hi[u] = (q + d < n) ? line[(size_t)(q + d) * nyz] : kInf;
      }
#pragma unroll
      for (int u = 0; u < kScanBatch; ++u) {
        const int d = d0 + u;
        const int f = lo[u] < hi[u] ? lo[u] : hi[u];
        const int c = d * d + f;   // < 2^31: d <= 2^15, f <= kInf
        best = c < best ? c : best;
      }
    }
    // sdf_map.cpp:355-361: min(res*sqrt(val), previous) with previous = 10000 after the
    // reset; a line without obstacles carries DBL_MAX there, i.e. keeps the 10000
    double dv = 10000.0;
    if (best < kInf) {
      const double e = g.res * sqrt((double)best);
      dv = e < dv ? e : dv;
    }
    dist[i] = dv;
    dist32[i] = (float)dv;
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

bool gtop_esdf_supported(const GtopGrid &g) {
  return g.nz <= 64 * kMaxChunks && g.nx <= 32768 && g.ny <= 32768;
}

size_t gtop_esdf_rows_ints(const GtopGrid &g) {
  const size_t ncol = (size_t)g.nx * g.ny;
  return 2 * ncol + (size_t)g.nx + (ncol + 3) / 4;   // cols, rank, cnt, colany (bytes)
}

hipError_t gtop_launch_esdf_build(const GtopGrid &g, const uint8_t *occ, int *tmp1, int *tmp2, int *rows,
                                  double *dist, float *dist32, hipStream_t stream) {
  const size_t ncol = (size_t)g.nx * g.ny, nvox = ncol * g.nz;
  int *cols = rows, *rank = rows + ncol, *cnt = rows + 2 * ncol;
  uint8_t *colany = reinterpret_cast<uint8_t *>(rows + 2 * ncol + g.nx);
  const unsigned zblocks = (unsigned)((ncol + 3) / 4 < 65536 ? (ncol + 3) / 4 : 65536);
  hipLaunchKernelGGL(esdf_z_kernel, dim3(zblocks), dim3(256), 0, stream, g, occ, tmp1, colany);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(esdf_rows_kernel, dim3(g.nx < 65536 ? g.nx : 65536), dim3(64), 0, stream, g,
                     (const uint8_t *)colany, cols, rank, cnt);
  e = hipGetLastError();
  if (e != hipSuccess) return e;
  const unsigned vblocks = (unsigned)((nvox + 255) / 256 < (1u << 20) ? (nvox + 255) / 256 : (1u << 20));
  hipLaunchKernelGGL(esdf_y_kernel, dim3(vblocks), dim3(256), 0, stream, g, (const int *)tmp1, tmp2,
                     (const int *)cols, (const int *)rank, (const int *)cnt);
  e = hipGetLastError();
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(esdf_x_kernel, dim3(vblocks), dim3(256), 0, stream, g, (const int *)tmp2, dist, dist32);
  return hipGetLastError();
}
