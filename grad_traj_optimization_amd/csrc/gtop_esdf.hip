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
//   x sweep : outward scan v = q-1, q+1, q-2, ... with the same cut-off (in(v) >= 0);
//             it also applies the final res*sqrt(.) (exactly rounded fp64, as the
//             reference's).  The fp32 copy used by the GTOP_F32 path is made by the
//             caller on first use (a third of the sweep's writes).
// What bounds the scans, and what this file does about it (measured, profiles/r2/esdf_kernels.txt):
//   * the texture-address unit takes 16 cycles per wave64 load whatever its width, so a lane owns 4 voxels
//     adjacent in z and every load is 16 bytes (y sweep 46 -> 22 us at 200^3);
//   * the x sweep re-read every row ~40 times (once per slab within reach): a lane now owns 4 consecutive
//     slabs as well and one outward pass serves all four (106 -> 65 us), and the yz plane is partitioned over
//     the 8 XCDs so that those re-reads hit one L2 instead of crossing the fabric (400^3: 1.44 -> 0.96 ms
//     before the blocking; 0.45 ms with it);
//   * what is left in the x sweep is its integer min-plus arithmetic (3 instructions per two candidates).
// Result: bit-identical to the CPU restatement and to scipy's exact EDT (tests), no scratch workspace.
// Rejected with measurements: an LDS-tiled variant of the scans (4-6x slower, round 1); the reference's own
// lower-envelope algorithm with one lane per line and the stack in LDS (155 us for the x sweep at 200^3, 2.8 ms
// at 400^3: a line's pops diverge across the 64 lanes and 4 B x line length of LDS per lane leaves one
// wavefront per CU at 400^3).

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
    if (dist) dist[i] = 10000.0;   // sdf_map.cpp:22, :51
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

// The same for columns of up to 256 voxels (NCH <= 4 chunks of 64): the ballots stay in scalar registers, what the
// other chunks contribute to a chunk (their highest occupied voxel below it, their lowest above it) is scalar
// arithmetic done once per column, and a lane only searches its own chunk's mask (21 -> 18 us at 200^3; what is
// left is one dependent round trip per wavefront generation — storing the distance as a byte instead of its square
// as an int made this sweep 2 us faster and the y sweep 2.4 us slower).
template <int NCH>
__global__ void __launch_bounds__(256)
esdf_z_small_kernel(const GtopGrid g, const uint8_t *__restrict__ occ, int *__restrict__ out,
                    uint8_t *__restrict__ colany) {
  constexpr int kFar = 1 << 20;   // "no occupied voxel on that side": farther than any column is long
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const size_t ncol = (size_t)g.nx * g.ny;
  const int nz = g.nz;
  for (size_t col = (size_t)blockIdx.x * 4 + w; col < ncol; col += (size_t)gridDim.x * 4) {
    const uint8_t *c = occ + col * nz;
    unsigned long long mk[NCH];
    int hi[NCH], lo[NCH];   // highest / lowest occupied voxel of each chunk (scalar)
    unsigned long long any = 0ull;
#pragma unroll
    for (int k = 0; k < NCH; ++k) {
      const int z = k * 64 + lane;
      mk[k] = __ballot((z < nz) && (c[z] == 1));
      any |= mk[k];
      hi[k] = mk[k] ? k * 64 + 63 - __clzll((long long)mk[k]) : -kFar;
      lo[k] = mk[k] ? k * 64 + (__ffsll((long long)mk[k]) - 1) : kFar;
    }
    if (lane == 0) colany[col] = any != 0ull;   // the column holds an obstacle: finite for the y sweep
#pragma unroll
    for (int ko = 0; ko < NCH; ++ko) {
      int below = -kFar, above = kFar;   // nearest occupied voxel in the chunks under / over this one
#pragma unroll
      for (int k = 0; k < NCH; ++k) {
        if (k < ko) below = max(below, hi[k]);
        if (k > ko) above = min(above, lo[k]);
      }
      const int z = ko * 64 + lane;
      const unsigned long long mb = mk[ko] & (lane == 63 ? ~0ull : ((2ull << lane) - 1ull));   // bits <= lane
      const unsigned long long ma = mk[ko] & (~0ull << lane);                                     // bits >= lane
      const int d1 = mb ? lane - (63 - __clzll((long long)mb)) : kFar;
      const int d2 = ma ? (__ffsll((long long)ma) - 1) - lane : kFar;
      const int best = min(min(d1, d2), min(z - below, above - z));
      if (z < nz) out[col * nz + z] = best >= (kFar >> 1) ? kInf : best * best;
    }
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

// The y and x scans below are bound by the texture-address unit: a wave64 load instruction occupies it for 16
// cycles whether each lane fetches 4 bytes or 16 (measured: the one-voxel-per-lane x scan at 200^3 ran 20 000
// loads per CU in 122 us = 16 cycles each).  So a lane owns V = 4 voxels adjacent in z (the fastest axis) and every
// load is a 16-byte one: the same rows in a quarter of the instructions.  The four voxels share the scan's radius
// (the widest of theirs; the extra candidates a voxel sees cannot win).  V = 1 serves grids whose nz is not a
// multiple of 4.
template <int V> struct IntV;
template <> struct IntV<1> { int v[1]; };
template <> struct __attribute__((aligned(16))) IntV<4> { int v[4]; };

template <int V>
__device__ __forceinline__ IntV<V> load_v(const int *p) { return *reinterpret_cast<const IntV<V> *>(p); }

// y sweep (sdf_map.cpp:328-346): out(x,y,z) = min over candidate columns v of (y-v)^2 + in(x,v,z).
// 32-bit index arithmetic throughout (nvox < 2^31; ny, nz < 2^15 so that v*nz is a 24-bit product).
template <int V>
__global__ void __launch_bounds__(256)
esdf_y_kernel(const GtopGrid g, const int *__restrict__ fin, int *__restrict__ fout, const int *__restrict__ cols,
              const int *__restrict__ rank, const int *__restrict__ cnt) {
  constexpr int U = 4;   // candidates per round trip and side
  const int nyz = g.ny * g.nz;
  const int ny = g.ny, nz = g.nz;
  // Workgroups are dealt round-robin over the 8 XCDs, each with its own L2: slab x (whose voxels only read
  // slab x) goes to XCD x mod 8, so a slab is fetched into ONE L2 instead of all eight.
  // grid = 8 * ceil(nx/8) * bps workgroups, bps = ceil(nyz/V/256).
  const int bps = (nyz / V + 255) >> 8;
  const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
  const int x = xcd + 8 * (j / bps);
  const int r = ((j % bps) * 256 + (int)threadIdx.x) * V;
  if (x >= g.nx || r >= nyz) return;
  const int i = x * nyz + r;
  const int q = r / nz;                   // (nz % V == 0: the V voxels share q)
  const int *line = fin + (i - q * nz);   // (x, 0, z)
  const int *cx = cols + x * ny;
  const int c = cnt[x];
  IntV<V> best = load_v<V>(fin + i);
  int worst = best.v[0];
#pragma unroll
  for (int e = 1; e < V; ++e) worst = max(worst, best.v[e]);
  const int k0 = rank[x * ny + q];        // first candidate at or above q
  // below q: candidates k0-1, k0-2, ... (descending y, ascending distance).  Indices are clamped to the
  // list's first entry instead of masked: a re-read candidate cannot beat itself.
  for (int k = k0 - 1; k >= 0; k -= U) {
    const int d0 = q - cx[k];
    if (__mul24(d0, d0) >= worst) break;   // in(v) >= 0: nothing farther can win
    int v[U];
    IntV<V> f[U];
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] = cx[max(k - u, 0)];
#pragma unroll
    for (int u = 0; u < U; ++u) f[u] = load_v<V>(line + __mul24(v[u], nz));
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int d = q - v[u], d2 = __mul24(d, d);
#pragma unroll
      for (int e = 0; e < V; ++e) best.v[e] = min(best.v[e], d2 + f[u].v[e]);   // d <= 2^15, f <= kInf: below 2^31
    }
    worst = best.v[0];
#pragma unroll
    for (int e = 1; e < V; ++e) worst = max(worst, best.v[e]);
  }
  // above q (the voxel's own column, if it is a candidate, is `best` already)
  for (int k = k0 + ((k0 < c && cx[k0] == q) ? 1 : 0); k < c; k += U) {
    const int d0 = cx[k] - q;
    if (__mul24(d0, d0) >= worst) break;
    int v[U];
    IntV<V> f[U];
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] = cx[min(k + u, c - 1)];
#pragma unroll
    for (int u = 0; u < U; ++u) f[u] = load_v<V>(line + __mul24(v[u], nz));
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int d = v[u] - q, d2 = __mul24(d, d);
#pragma unroll
      for (int e = 0; e < V; ++e) best.v[e] = min(best.v[e], d2 + f[u].v[e]);
    }
    worst = best.v[0];
#pragma unroll
    for (int e = 1; e < V; ++e) worst = max(worst, best.v[e]);
  }
#pragma unroll
  for (int e = 0; e < V; ++e) best.v[e] = best.v[e] > kInf ? kInf : best.v[e];
  *reinterpret_cast<IntV<V> *>(fout + i) = best;
}

// x sweep (sdf_map.cpp:348-364): out(q) = min_v ((q-v)^2 + in(v)), scanning outward; then
// dist = min(res*sqrt(out), previous).  A lane owns a block of V voxels along z times kXB = 4 consecutive slabs
// along x: the rows the four slabs' scans need overlap almost entirely, so one pass outward from the block
// (rows q0-d and q0+3+d, distance d+e resp. d+3-e to the block's e-th slab) serves all of them — a quarter of the
// loads of four separate scans, which is what bounded this kernel (L2 bandwidth: every row was re-read by the
// ~2 x 20 slabs around it).  Element indices advance by +-nyz per step and are CLAMPED to the line's ends instead
// of masked: past an end the lane re-reads the end row with a larger d, an over-estimate of a candidate it has
// already seen, which can never win: exact.  The scan stops when (d+1)^2 >= the worst of the block's minima.
#ifndef GTOP_ESDF_XB
#define GTOP_ESDF_XB 4
#endif
constexpr int kXB = GTOP_ESDF_XB;

template <int V>
__global__ void __launch_bounds__(256)
esdf_x_kernel(const GtopGrid g, const int *__restrict__ fin, double *__restrict__ dist, float *__restrict__ dist32) {
  constexpr int kScanBatch = 4;   // steps per round trip
  const int nyz = g.ny * g.nz;
  const int n = g.nx;
  // XCD-aware order: the yz plane is cut in 8 parts and XCD c scans part c of every slab block, block after block —
  // the rows a lane reads are then shared, in one L2, with the lanes of the neighbouring slab blocks that run at
  // the same time (dealt linearly, every XCD walked every slab: 1.44 ms -> 0.96 ms at 400^3 with this order).
  // grid = 8 * ceil(n/kXB) * bpp workgroups, bpp = workgroups per part = ceil(ceil(nyz/V/8)/256).
  const int nl = nyz / V;
  const int bpp = (((nl + 7) >> 3) + 255) >> 8, part = bpp << 8;
  const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
  const int q0 = (j / bpp) * kXB;
  const int fl = xcd * part + (j % bpp) * 256 + (int)threadIdx.x;
  if (fl >= nl || fl >= (xcd + 1) * part) return;
  const int first = fl * V;                 // the line (y,z) = its voxel in slab 0
  const int last = first + (n - 1) * nyz;   // the line's end voxels: first, last
  // the block's own rows (slabs past the end of the line shadow the last one; they are not stored)
  int row[kXB];
  IntV<V> best[kXB];
#pragma unroll
  for (int e = 0; e < kXB; ++e) {
    row[e] = first + min(q0 + e, n - 1) * nyz;
    best[e] = load_v<V>(fin + row[e]);
  }
  {
    IntV<V> own[kXB];
#pragma unroll
    for (int e = 0; e < kXB; ++e) own[e] = best[e];
#pragma unroll
    for (int e = 0; e < kXB; ++e)
#pragma unroll
      for (int o = 0; o < kXB; ++o)
        if (o != e)
#pragma unroll
          for (int v = 0; v < V; ++v) best[e].v[v] = min(best[e].v[v], (e - o) * (e - o) + own[o].v[v]);
  }
  auto worst_of = [&]() {
    int w = 0;
#pragma unroll
    for (int e = 0; e < kXB; ++e)
#pragma unroll
      for (int v = 0; v < V; ++v) w = max(w, best[e].v[v]);
    return w;
  };
  int worst = worst_of();
  const int reach = max(max(q0, n - kXB - q0), 0);
  int lo = row[0], hi = row[kXB - 1], d = 0;
  // The loads of a batch are independent and issue together; entries past the exact
  // cut-off cannot win (in(v) >= 0), so reading a few of them changes nothing.
  while (d < reach) {
    if (__mul24(d + 1, d + 1) >= worst) break;
    IntV<V> flo[kScanBatch], fhi[kScanBatch];
#pragma unroll
    for (int u = 0; u < kScanBatch; ++u) {
      lo = max(lo - nyz, first);
      hi = min(hi + nyz, last);
      flo[u] = load_v<V>(fin + lo);
      fhi[u] = load_v<V>(fin + hi);
    }
#pragma unroll
    for (int u = 0; u < kScanBatch; ++u) {
      ++d;
#pragma unroll
      for (int e = 0; e < kXB; ++e) {
        const int dl = d + e, dr = d + (kXB - 1 - e);
        const int dl2 = __mul24(dl, dl), dr2 = __mul24(dr, dr);   // (24-bit multiplies are full rate, 32-bit ones a
                                                                  // quarter; d <= 2^15, f <= kInf: sums below 2^31)
#pragma unroll
        for (int v = 0; v < V; ++v)
          best[e].v[v] = min(best[e].v[v], min(dl2 + flo[u].v[v], dr2 + fhi[u].v[v]));
      }
    }
    worst = worst_of();
  }
  // sdf_map.cpp:355-361: min(res*sqrt(val), previous) with previous = 10000 after the
  // reset; a line without obstacles carries DBL_MAX there, i.e. keeps the 10000
#pragma unroll
  for (int e = 0; e < kXB; ++e) {
    if (q0 + e >= n) break;
#pragma unroll
    for (int v = 0; v < V; ++v) {
      double dv = 10000.0;
      if (best[e].v[v] < kInf) {
        const double r = g.res * sqrt((double)best[e].v[v]);
        dv = r < dv ? r : dv;
      }
      dist[row[e] + v] = dv;
      if (dist32) dist32[row[e] + v] = (float)dv;   // (wave-uniform; the caller may make the fp32 copy later)
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

bool gtop_esdf_supported(const GtopGrid &g) {
  return g.nz <= 64 * kMaxChunks && g.nx <= 32768 && g.ny <= 32768;
}

size_t gtop_esdf_rows_ints(const GtopGrid &g) {
  const size_t ncol = (size_t)g.nx * g.ny;
  return 2 * ncol + (size_t)g.nx + (ncol + 3) / 4;   // cols, rank, cnt, colany (bytes)
}

hipError_t gtop_launch_esdf_build(const GtopGrid &g, const uint8_t *occ, int *tmp1, int *tmp2, int *rows,
                                  double *dist, float *dist32, hipStream_t stream) {
  const size_t ncol = (size_t)g.nx * g.ny;
  int *cols = rows, *rank = rows + ncol, *cnt = rows + 2 * ncol;
  uint8_t *colany = reinterpret_cast<uint8_t *>(rows + 2 * ncol + g.nx);
  const unsigned zblocks = (unsigned)((ncol + 3) / 4 < 65536 ? (ncol + 3) / 4 : 65536);
  switch ((g.nz + 63) >> 6) {
    case 1: hipLaunchKernelGGL(esdf_z_small_kernel<1>, dim3(zblocks), dim3(256), 0, stream, g, occ, tmp1, colany); break;
    case 2: hipLaunchKernelGGL(esdf_z_small_kernel<2>, dim3(zblocks), dim3(256), 0, stream, g, occ, tmp1, colany); break;
    case 3: hipLaunchKernelGGL(esdf_z_small_kernel<3>, dim3(zblocks), dim3(256), 0, stream, g, occ, tmp1, colany); break;
    case 4: hipLaunchKernelGGL(esdf_z_small_kernel<4>, dim3(zblocks), dim3(256), 0, stream, g, occ, tmp1, colany); break;
    default: hipLaunchKernelGGL(esdf_z_kernel, dim3(zblocks), dim3(256), 0, stream, g, occ, tmp1, colany);
  }
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(esdf_rows_kernel, dim3(g.nx < 65536 ? g.nx : 65536), dim3(64), 0, stream, g,
                     (const uint8_t *)colany, cols, rank, cnt);
  e = hipGetLastError();
  if (e != hipSuccess) return e;
  const int nyz = g.ny * g.nz;
#ifndef GTOP_ESDF_VEC
#define GTOP_ESDF_VEC 4
#endif
  const int V = (GTOP_ESDF_VEC == 4 && g.nz % 4 == 0) ? 4 : 1;   // voxels per lane (16-byte loads need nz % 4 == 0)
  const int nl = nyz / V;
  const unsigned yblocks = 8u * (unsigned)((g.nx + 7) / 8) * (unsigned)((nl + 255) / 256);
  const unsigned xblocks = 8u * (unsigned)((g.nx + kXB - 1) / kXB) * (unsigned)((((nl + 7) >> 3) + 255) >> 8);
  if (V == 4)
    hipLaunchKernelGGL(esdf_y_kernel<4>, dim3(yblocks), dim3(256), 0, stream, g, (const int *)tmp1, tmp2,
                       (const int *)cols, (const int *)rank, (const int *)cnt);
  else
    hipLaunchKernelGGL(esdf_y_kernel<1>, dim3(yblocks), dim3(256), 0, stream, g, (const int *)tmp1, tmp2,
                       (const int *)cols, (const int *)rank, (const int *)cnt);
  e = hipGetLastError();
  if (e != hipSuccess) return e;
  if (V == 4)
    hipLaunchKernelGGL(esdf_x_kernel<4>, dim3(xblocks), dim3(256), 0, stream, g, (const int *)tmp2, dist, dist32);
  else
    hipLaunchKernelGGL(esdf_x_kernel<1>, dim3(xblocks), dim3(256), 0, stream, g, (const int *)tmp2, dist, dist32);
  return hipGetLastError();
}
