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
//             sorted list of candidate columns (ballots/popcounts over the z sweep's
//             per-column flags: by the y sweep's workgroup itself, in LDS, or by
//             esdf_rows_kernel for ny > 2048).  A voxel walks that list outward from its own y,
//             four candidates per round trip, with the exact cut-off d^2 >= best;
//             obstacle-free stretches cost nothing;
//   x sweep : outward scan v = q-1, q+1, q-2, ... with the same cut-off (in(v) >= 0);
//             it also applies the final res*sqrt(.) (exactly rounded fp64, as the
//             reference's).  (Round 4: the GTOP_F32 path reads fp32 corner records, gtop_records.hip; dist32 is NULL.)  The fp32 copy used by the GTOP_F32 path was made by the
//             caller on first use (a third of the sweep's writes).
// What bounds the scans, and what this file does about it (measured, profiles/r2/esdf_kernels.txt):
//   * the texture-address unit takes 16 cycles per wave64 load whatever its width, so a lane owns 4 voxels
//     adjacent in z and every load is 16 bytes (y sweep 46 -> 22 us at 200^3);
//   * the x sweep re-read every row ~40 times (once per slab within reach): a lane now owns 4 consecutive
//     slabs as well and one outward pass serves all four (106 -> 65 us), and the yz plane is partitioned over
//     the 8 XCDs so that those re-reads hit one L2 instead of crossing the fabric (400^3: 1.44 -> 0.96 ms
//     before the blocking; 0.45 ms with it);
//   * the x sweep's min-plus arithmetic runs on packed 16-bit values where they fit (esdf_x16_kernel: two voxels
//     per v_pk_add_u16 / v_pk_min_u16, 8 voxels per 16-byte load; the 32-bit scan is its exact fallback);
//   * a SIMD's wavefronts share its issue slots, so the x sweep ends when the SIMD with the longest scans does:
//     the plane is dealt to the XCDs in 64-lane chunks (not one contiguous eighth each) so that each gets an
//     even sample of the map (65 -> 56 us at 200^3; per-wavefront timeline: tools/esdf_stamps.py).
// Result: bit-identical to the CPU restatement and to scipy's exact EDT (tests).
// Rejected with measurements: an LDS-tiled variant of the scans (4-6x slower, round 1); the reference's own
// lower-envelope algorithm with one lane per line and the stack in LDS (155 us for the x sweep at 200^3, 2.8 ms
// at 400^3: a line's pops diverge across the 64 lanes and 4 B x line length of LDS per lane leaves one
// wavefront per CU at 400^3).

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "gtop_kernels.h"

namespace {

constexpr int kInf = 0x3fffffff;   // "no obstacle on this line so far"; d^2 + kInf stays below 2^31 for lines <= 2^15

// resetBuffer (sdf_map.cpp:26-53).  16 occupancy bytes per lane and store (hipMalloc'd: 256-byte aligned); the
// last nvox % 16 bytes one at a time.
__global__ void __launch_bounds__(256)
esdf_reset_kernel(uint8_t *__restrict__ occ, double *__restrict__ dist, size_t nvox) {
  const size_t tid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  const size_t nq = nvox >> 4;
  uint4 *occ16 = reinterpret_cast<uint4 *>(occ);
  for (size_t i = tid; i < nq; i += stride) occ16[i] = make_uint4(0u, 0u, 0u, 0u);
  for (size_t i = (nq << 4) + tid; i < nvox; i += stride) occ[i] = 0;
  if (dist)
    for (size_t i = tid; i < nvox; i += stride) dist[i] = 10000.0;   // sdf_map.cpp:22, :51
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
esdf_z_kernel(const GtopGrid g, const uint8_t *__restrict__ occ, int *__restrict__ out, uint16_t *__restrict__ out16,
              uint8_t *__restrict__ colany, int *__restrict__ n_empty_slabs) {
  if (blockIdx.x == 0 && threadIdx.x == 0) *n_empty_slabs = 0;   // counted by the y sweep / esdf_rows_kernel, read by the x sweep
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
      if (z < nz) {
        out[col * nz + z] = best >= kInf ? kInf : best * best;
        if (out16) out16[col * nz + z] = (uint16_t)(best >= 256 ? 0xFFFF : best * best);   // min(value, 0xFFFF): esdf_y16_kernel
      }
    }
    __builtin_amdgcn_wave_barrier();
  }
}

// The same for columns of up to 512 voxels (NCH <= 8 chunks of 64): the ballots stay in scalar registers, what the
// other chunks contribute to a chunk (their highest occupied voxel below it, their lowest above it) is scalar
// arithmetic done once per column, and a lane only searches its own chunk's mask (21 -> 18 us at 200^3).  What is
// left is instruction issue — 160 VALU + 81 SALU per column, 39 columns per SIMD — not latency: four columns per
// trip with all their loads issued first made it slower (20.6 us); storing the distance as a byte instead of its
// square as an int made this sweep 2 us faster and the y sweep 2.4 us slower.
template <int NCH>
__global__ void __launch_bounds__(256)
esdf_z_small_kernel(const GtopGrid g, const uint8_t *__restrict__ occ, int *__restrict__ out, uint16_t *__restrict__ out16,
                    uint8_t *__restrict__ colany, int *__restrict__ n_empty_slabs) {
  if (blockIdx.x == 0 && threadIdx.x == 0) *n_empty_slabs = 0;   // counted by the y sweep / esdf_rows_kernel, read by the x sweep
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
    if (any == 0ull) {   // (wave-uniform) most columns of a map are empty: nothing to search
      // ... and, for the packed y sweep (out16 set), nothing to store: esdf_y16_kernel reads only columns that hold an
      // obstacle (its candidates; a voxel's own column by the same flag), so an empty column's 6 bytes per voxel —
      // most of this sweep's stores — are never looked at
      if (!out16) {
#pragma unroll
        for (int ko = 0; ko < NCH; ++ko) {
          const int z = ko * 64 + lane;
          if (z < nz) out[col * nz + z] = kInf;
        }
      }
      continue;
    }
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
      if (z < nz) {
        out[col * nz + z] = best >= (kFar >> 1) ? kInf : best * best;
        if (out16) out16[col * nz + z] = (uint16_t)(best >= 256 ? 0xFFFF : best * best);   // min(value, 0xFFFF)
      }
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
    if (lane == 0) {
      cnt[x] = base;
      if (base == 0) atomicAdd(cnt + g.nx, 1);
    }
  }
}

// The y and x scans below are bound by the texture-address unit: a wave64 load instruction occupies it for 16
// cycles whether each lane fetches 4 bytes or 16 (measured: the one-voxel-per-lane x scan at 200^3 ran 20 000
// loads per CU in 122 us = 16 cycles each).  So a lane owns V = 4 voxels adjacent in z (the fastest axis) and every
// load is a 16-byte one: the same rows in a quarter of the instructions.  The four voxels share the scan's radius
// (the widest of theirs; the extra candidates a voxel sees cannot win).  V = 1 serves grids whose nz is not a
// multiple of 4.
#ifdef GTOP_ESDF_STAMPS
// tuning aid (tools/esdf_stamps.py): per wavefront of the x sweep, start / end on the 100 MHz wall clock, scan
// steps and placement.  Not part of the product build.
__device__ unsigned long long g_esdf_stamps[4 * 65536];
#ifdef GTOP_ESDF_STAMP_Y
constexpr bool getenv_stamp_y = true;
#else
constexpr bool getenv_stamp_y = false;
#endif
__device__ __forceinline__ void esdf_stamp(unsigned long long t0, int steps) {
  const unsigned long long t1 = wall_clock64();
  if ((threadIdx.x & 63) == 0) {
    const unsigned w = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    if (w < 65536) {
      g_esdf_stamps[4 * w + 0] = t0;
      g_esdf_stamps[4 * w + 1] = t1;
      g_esdf_stamps[4 * w + 2] = (unsigned long long)steps;
      g_esdf_stamps[4 * w + 3] = (unsigned long long)__builtin_amdgcn_s_getreg((15 << 11) | 4 /* HW_ID, bits 15:0 */) |
                                 ((unsigned long long)__smid() << 32);
    }
  }
}
#endif

template <int V> struct IntV;
template <> struct IntV<1> { int v[1]; };
template <> struct __attribute__((aligned(16))) IntV<4> { int v[4]; };

template <int V>
__device__ __forceinline__ IntV<V> load_v(const int *p) { return *reinterpret_cast<const IntV<V> *>(p); }

// eight voxels as packed 16-bit squares, saturated at 0xFFFF (the packed sweeps: esdf_y16_kernel, esdf_x16_kernel)
typedef unsigned short gtop_u16x2 __attribute__((ext_vector_type(2)));
struct __attribute__((aligned(16))) PkV { gtop_u16x2 p[4]; };

// y sweep (sdf_map.cpp:328-346): out(x,y,z) = min over candidate columns v of (y-v)^2 + in(x,v,z).
// 32-bit index arithmetic throughout (nvox < 2^31; ny, nz < 2^15 so that v*nz is a 24-bit product).
// LOCAL: the workgroup builds its slab's candidate list itself, in LDS, from the z sweep's per-column flags (ny <=
// kYLocalMax) — no esdf_rows_kernel launch between the sweeps (5 us of a 110 us build at 200^3), and the list is
// then read from LDS instead of global memory.
constexpr int kYLocalMax = 2048;

template <int V, bool LOCAL>
__global__ void __launch_bounds__(256)
esdf_y_kernel(const GtopGrid g, const int *__restrict__ fin, int *__restrict__ fout, uint16_t *__restrict__ fout16,
              const int *__restrict__ cols, const int *__restrict__ rank, const int *__restrict__ cnt,
              const uint8_t *__restrict__ colany, int *__restrict__ cnt_out) {
  constexpr int U = 4;   // candidates per round trip and side
  const int nyz = g.ny * g.nz;
  const int ny = g.ny, nz = g.nz;
  // Workgroups are dealt round-robin over the 8 XCDs, each with its own L2: slab x (whose voxels only read
  // slab x) goes to XCD x mod 8, so a slab is fetched into ONE L2 instead of all eight.
  // grid = 8 * ceil(nx/8) * bps workgroups, bps = ceil(nyz/V/256).
#ifdef GTOP_ESDF_STAMPS
  const unsigned long long t0_stamp = wall_clock64();
  int trips = 0;
#endif
  const int bps = (nyz / V + 255) >> 8;
  const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
  const int x = xcd + 8 * (j / bps);
  const int r = ((j % bps) * 256 + (int)threadIdx.x) * V;
  __shared__ unsigned short s_cols[LOCAL ? kYLocalMax : 1];
  __shared__ unsigned long long s_mask[LOCAL ? kYLocalMax / 64 : 1];
  __shared__ int s_pref[LOCAL ? kYLocalMax / 64 + 1 : 1];
  if constexpr (LOCAL) {
    if (x >= g.nx) return;   // (workgroup-uniform)
    if (threadIdx.x < 64) {  // one wavefront: ballots over the slab's column flags, 64 columns at a time
      const int lane = threadIdx.x;
      const uint8_t *ca = colany + x * ny;
      constexpr int kPre = 4;   // flag loads in flight
      int base = 0;
      for (int y0 = 0; y0 < ny; y0 += 64 * kPre) {
        bool f[kPre];
#pragma unroll
        for (int u = 0; u < kPre; ++u) {
          const int y = y0 + 64 * u + lane;
          f[u] = (y < ny) && ca[y];
        }
#pragma unroll
        for (int u = 0; u < kPre; ++u) {
          const int y = y0 + 64 * u + lane;
          if (y0 + 64 * u >= ny) break;
          const unsigned long long mk = __ballot(f[u]);
          if (f[u]) s_cols[base + __popcll(mk & ((1ull << lane) - 1ull))] = (unsigned short)y;
          if (lane == 0) {
            s_mask[(y0 >> 6) + u] = mk;
            s_pref[(y0 >> 6) + u] = base;
          }
          base += __popcll(mk);
        }
      }
      if (lane == 0) {
        s_pref[(ny + 63) >> 6] = base;
        if (j % bps == 0) {   // the slab's number of obstacle columns (0: the x sweep jumps over its rows)
          cnt_out[x] = base;
          if (base == 0) atomicAdd(cnt_out + g.nx, 1);
        }
      }
    }
    __syncthreads();
    if (r >= nyz) return;
  } else {
    if (x >= g.nx || r >= nyz) return;
  }
  const int i = x * nyz + r;
  const int q = r / nz;                   // (nz % V == 0: the V voxels share q)
  const int *line = fin + (i - q * nz);   // (x, 0, z)
  // the slab's candidate columns, ascending; c of them; k0 = the first at or above q
  auto cx = [&](int k) -> int {
    if constexpr (LOCAL) return s_cols[k];
    else return cols[x * ny + k];
  };
  int c, k0;
  if constexpr (LOCAL) {
    c = s_pref[(ny + 63) >> 6];
    k0 = s_pref[q >> 6] + __popcll(s_mask[q >> 6] & ((1ull << (q & 63)) - 1ull));
  } else {
    c = cnt[x];
    k0 = rank[x * ny + q];
  }
  IntV<V> best = load_v<V>(fin + i);
  int worst = best.v[0];
#pragma unroll
  for (int e = 1; e < V; ++e) worst = max(worst, best.v[e]);
  // below q: candidates k0-1, k0-2, ... (descending y, ascending distance).  Indices are clamped to the
  // list's first entry instead of masked: a re-read candidate cannot beat itself.
  for (int k = k0 - 1; k >= 0; k -= U) {
    const int d0 = q - cx(k);
    if (__mul24(d0, d0) >= worst) break;   // in(v) >= 0: nothing farther can win
#ifdef GTOP_ESDF_STAMPS
    ++trips;
#endif
    int v[U];
    IntV<V> f[U];
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] = cx(max(k - u, 0));
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
  for (int k = k0 + ((k0 < c && cx(k0) == q) ? 1 : 0); k < c; k += U) {
    const int d0 = cx(k) - q;
    if (__mul24(d0, d0) >= worst) break;
#ifdef GTOP_ESDF_STAMPS
    ++trips;
#endif
    int v[U];
    IntV<V> f[U];
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] = cx(min(k + u, c - 1));
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
  if constexpr (V == 4) {
    if (fout16) {   // the x sweep's packed 16-bit form: min(value, 0xFFFF) (wave-uniform branch)
      uint2 pk;
      pk.x = (unsigned)min(best.v[0], 0xFFFF) | ((unsigned)min(best.v[1], 0xFFFF) << 16);
      pk.y = (unsigned)min(best.v[2], 0xFFFF) | ((unsigned)min(best.v[3], 0xFFFF) << 16);
      *reinterpret_cast<uint2 *>(fout16 + i) = pk;
    }
  }
#ifdef GTOP_ESDF_STAMPS
  if (getenv_stamp_y) {
    int tmax = trips;
    for (int off = 32; off > 0; off >>= 1) tmax = max(tmax, __shfl_xor(tmax, off));
    esdf_stamp(t0_stamp, tmax);
  }
#endif
}


// The y sweep on packed 16-bit values (round 3): the scan is bound by the texture-address unit — a wave64 load costs
// it 16 cycles whatever its width — so the z sweep leaves min(value, 0xFFFF) as 16-bit words beside its int32 output
// and a lane here owns EIGHT voxels adjacent in z: a 16-byte load brings 8 voxels instead of 4, and the min-plus step
// on two of them is one saturating v_pk_add_u16 + one v_pk_min_u16.  As in esdf_x16_kernel the packed scan returns
// min(exact, 0xFFFF) for every voxel whatever the inputs (a saturated candidate, or a d^2 past 16 bits, can never
// beat a minimum below 0xFFFF; unsaturated values are exact); a wavefront that ends with a saturated minimum (more than
// 255 voxels to the nearest obstacle within the slab) redoes its voxels with the 32-bit scan on the int32 data.
// Same candidate lists, same XCD placement, same outputs (int32 + 16-bit) as esdf_y_kernel<4, LOCAL>; nz % 8 == 0.
template <bool LOCAL>
__global__ void __launch_bounds__(256)
esdf_y16_kernel(const GtopGrid g, const uint16_t *__restrict__ fin16, const int *__restrict__ fin, int *__restrict__ fout,
                uint16_t *__restrict__ fout16, const int *__restrict__ cols, const int *__restrict__ rank,
                const int *__restrict__ cnt, const uint8_t *__restrict__ colany, int *__restrict__ cnt_out) {
  constexpr int U = 4, V = 8;   // candidates per round trip and side; voxels per lane
  const int nyz = g.ny * g.nz;
  const int ny = g.ny, nz = g.nz;
  const int bps = (nyz / V + 255) >> 8;
  const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
  const int x = xcd + 8 * (j / bps);
  const int r_raw = ((j % bps) * 256 + (int)threadIdx.x) * V;
  __shared__ unsigned short s_cols[LOCAL ? kYLocalMax : 1];
  __shared__ unsigned long long s_mask[LOCAL ? kYLocalMax / 64 : 1];
  __shared__ int s_pref[LOCAL ? kYLocalMax / 64 + 1 : 1];
  if (x >= g.nx) return;   // (workgroup-uniform)
  if constexpr (LOCAL) {
    if (threadIdx.x < 64) {  // one wavefront: ballots over the slab's column flags (as esdf_y_kernel)
      const int lane = threadIdx.x;
      const uint8_t *ca = colany + x * ny;
      constexpr int kPre = 4;
      int base = 0;
      for (int y0 = 0; y0 < ny; y0 += 64 * kPre) {
        bool f[kPre];
#pragma unroll
        for (int u = 0; u < kPre; ++u) {
          const int y = y0 + 64 * u + lane;
          f[u] = (y < ny) && ca[y];
        }
#pragma unroll
        for (int u = 0; u < kPre; ++u) {
          const int y = y0 + 64 * u + lane;
          if (y0 + 64 * u >= ny) break;
          const unsigned long long mk = __ballot(f[u]);
          if (f[u]) s_cols[base + __popcll(mk & ((1ull << lane) - 1ull))] = (unsigned short)y;
          if (lane == 0) {
            s_mask[(y0 >> 6) + u] = mk;
            s_pref[(y0 >> 6) + u] = base;
          }
          base += __popcll(mk);
        }
      }
      if (lane == 0) {
        s_pref[(ny + 63) >> 6] = base;
        if (j % bps == 0) {
          cnt_out[x] = base;
          if (base == 0) atomicAdd(cnt_out + g.nx, 1);
        }
      }
    }
    __syncthreads();
  }
  // whole wavefronts only (the fallback below is decided per wavefront): lanes past the slab's end shadow its last
  // lane with work and store nothing
  const int wave_first = r_raw - ((int)threadIdx.x & 63) * V;
  if (wave_first >= nyz) return;
  const bool has_work = r_raw < nyz;
  const int r = has_work ? r_raw : nyz - V;
  const int i = x * nyz + r;
  const int q = r / nz;                       // (nz % 8 == 0: the 8 voxels share q)
  const int lbase = i - q * nz;               // (x, 0, z)
  auto cx = [&](int k) -> int {
    if constexpr (LOCAL) return s_cols[k];
    else return cols[x * ny + k];
  };
  int c, k0;
  if constexpr (LOCAL) {
    c = s_pref[(ny + 63) >> 6];
    k0 = s_pref[q >> 6] + __popcll(s_mask[q >> 6] & ((1ull << (q & 63)) - 1ull));
  } else {
    c = cnt[x];
    k0 = rank[x * ny + q];
  }
  const int ka = k0 + ((k0 < c && cx(k0) == q) ? 1 : 0);   // the first candidate above q that is not q's own column
  if (c == 0) {   // (workgroup-uniform) a slab without obstacles: nothing but "no obstacle"
    if (has_work) {
      IntV<4> inf;
      inf.v[0] = inf.v[1] = inf.v[2] = inf.v[3] = kInf;
      *reinterpret_cast<IntV<4> *>(fout + i) = inf;
      *reinterpret_cast<IntV<4> *>(fout + i + 4) = inf;
      if (fout16) *reinterpret_cast<uint4 *>(fout16 + i) = make_uint4(~0u, ~0u, ~0u, ~0u);
    }
    return;
  }
  auto splat = [](int d2) {
    gtop_u16x2 s2;
    s2.x = (unsigned short)d2;
    s2.y = (unsigned short)d2;
    return s2;
  };
  // the voxels' own column: read only if it holds an obstacle (the z sweep stores nothing for the others)
  const bool own = k0 < c && cx(k0) == q;
  PkV best;
  best.p[0] = best.p[1] = best.p[2] = best.p[3] = splat(0xFFFF);
  if (own) best = *reinterpret_cast<const PkV *>(fin16 + i);
  auto worst_of = [&]() {
    const gtop_u16x2 w = __builtin_elementwise_max(__builtin_elementwise_max(best.p[0], best.p[1]),
                                                   __builtin_elementwise_max(best.p[2], best.p[3]));
    return max((int)w.x, (int)w.y);
  };
  int worst = worst_of();
  // below q: candidates k0-1, k0-2, ... (ascending distance); indices clamped to the list's ends instead of masked
  for (int k = k0 - 1; k >= 0; k -= U) {
    const int d0 = q - cx(k);
    if (__mul24(d0, d0) >= worst) break;   // in(v) >= 0: nothing farther can win (d0 >= 256: d0^2 > 0xFFFF >= worst)
    int v[U];
    PkV f[U];
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] = cx(max(k - u, 0));
#pragma unroll
    for (int u = 0; u < U; ++u) f[u] = *reinterpret_cast<const PkV *>(fin16 + lbase + __mul24(v[u], nz));
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int d = q - v[u];
      const gtop_u16x2 dd = splat(min(__mul24(d, d), 0xFFFF));
#pragma unroll
      for (int p = 0; p < 4; ++p) best.p[p] = __builtin_elementwise_min(best.p[p], __builtin_elementwise_add_sat(f[u].p[p], dd));
    }
    worst = worst_of();
  }
  for (int k = ka; k < c; k += U) {
    const int d0 = cx(k) - q;
    if (__mul24(d0, d0) >= worst) break;
    int v[U];
    PkV f[U];
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] = cx(min(k + u, c - 1));
#pragma unroll
    for (int u = 0; u < U; ++u) f[u] = *reinterpret_cast<const PkV *>(fin16 + lbase + __mul24(v[u], nz));
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int d = v[u] - q;
      const gtop_u16x2 dd = splat(min(__mul24(d, d), 0xFFFF));
#pragma unroll
      for (int p = 0; p < 4; ++p) best.p[p] = __builtin_elementwise_min(best.p[p], __builtin_elementwise_add_sat(f[u].p[p], dd));
    }
    worst = worst_of();
  }
  if (__any(worst == 0xFFFF)) {
    // (wave-uniform) a minimum at or past 2^16 - 1 somewhere in the wavefront: the exact 32-bit scan of esdf_y_kernel
    // for its voxels, four at a time
#pragma unroll 1
    for (int h = 0; h < 2; ++h) {
      const int ih = i + 4 * h;
      const int *line = fin + lbase + 4 * h;
      IntV<4> b32;
      b32.v[0] = b32.v[1] = b32.v[2] = b32.v[3] = kInf;
      if (own) b32 = load_v<4>(fin + ih);
      int w32 = max(max(b32.v[0], b32.v[1]), max(b32.v[2], b32.v[3]));
      for (int k = k0 - 1; k >= 0; --k) {
        const int vv = cx(k), d = q - vv, d2 = __mul24(d, d);
        if (d2 >= w32) break;
        const IntV<4> f = load_v<4>(line + __mul24(vv, nz));
#pragma unroll
        for (int e = 0; e < 4; ++e) b32.v[e] = min(b32.v[e], d2 + f.v[e]);
        w32 = max(max(b32.v[0], b32.v[1]), max(b32.v[2], b32.v[3]));
      }
      for (int k = ka; k < c; ++k) {
        const int vv = cx(k), d = vv - q, d2 = __mul24(d, d);
        if (d2 >= w32) break;
        const IntV<4> f = load_v<4>(line + __mul24(vv, nz));
#pragma unroll
        for (int e = 0; e < 4; ++e) b32.v[e] = min(b32.v[e], d2 + f.v[e]);
        w32 = max(max(b32.v[0], b32.v[1]), max(b32.v[2], b32.v[3]));
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) b32.v[e] = b32.v[e] > kInf ? kInf : b32.v[e];
      if (has_work) {
        *reinterpret_cast<IntV<4> *>(fout + ih) = b32;
        if (fout16) {
          uint2 pk;
          pk.x = (unsigned)min(b32.v[0], 0xFFFF) | ((unsigned)min(b32.v[1], 0xFFFF) << 16);
          pk.y = (unsigned)min(b32.v[2], 0xFFFF) | ((unsigned)min(b32.v[3], 0xFFFF) << 16);
          *reinterpret_cast<uint2 *>(fout16 + ih) = pk;
        }
      }
    }
    return;
  }
  // every value is below 0xFFFF: the 16-bit words ARE the exact minima, and the only copy stored — the x sweep reads
  // the int32 output only where the 16-bit one says 0xFFFF (esdf_x_scan_block), i.e. in the wavefronts above
  if (has_work) *reinterpret_cast<PkV *>(fout16 + i) = best;
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

// Slabs without any obstacle hold nothing but "no obstacle" after the y sweep: a candidate from such a row can never
// lower a minimum (INF + d^2 >= INF).  The x scans therefore JUMP over runs of empty slabs: before a batch, if the
// next slab on the left and the next on the right both start a run of empty slabs, d advances by the shorter run
// with no loads and no arithmetic (rows inside a batch are still processed unconditionally: an INF row is harmless,
// and the dense path stays as it was).  Exact, and what makes sparse maps cheap — a few obstacles in a 200^3 map:
// 189 of 200 slabs empty: x sweep 103 -> 25 us.  near[0][x] / near[1][x] = the nearest slab with obstacles at or left / at or right
// of x (-1 / n when there is none), built by the workgroup from cnt[] (y sweep / esdf_rows_kernel) with ballots;
// lines of more than kSlabMax slabs scan every row.
constexpr int kSlabMax = 2048;
struct EsdfSlabRuns {
  short near[2][kSlabMax];
  unsigned long long mask[kSlabMax / 64];
  int any_empty;   // 0: every slab holds obstacles — the scans then never ask
};
__device__ __forceinline__ bool esdf_stage_slab_runs(EsdfSlabRuns *sr, const int *__restrict__ cnt, int n) {
  // (building the tables costs a workgroup ~2 us: only where at least a quarter of the slabs is empty — cnt[n] counts
  // them — i.e. where the jumps pay; 200^3 at 2 % occupancy has 31 empty slabs and gains nothing, 400^3 at 4 % has a
  // handful and 16 000 workgroups)
  if (n <= kSlabMax && cnt[n] * 4 >= n) {
    const int lane = threadIdx.x & 63, nw = blockDim.x >> 6, w = threadIdx.x >> 6, nch = (n + 63) >> 6;
    if (threadIdx.x == 0) sr->any_empty = 0;
    __syncthreads();
    for (int c = w; c < nch; c += nw) {
      const int x = c * 64 + lane;
      const unsigned long long m = __ballot(x < n && cnt[x] > 0);
      const unsigned long long full = (c == nch - 1 && (n & 63)) ? ((1ull << (n & 63)) - 1ull) : ~0ull;
      if (lane == 0) {
        sr->mask[c] = m;
        if (m != full) sr->any_empty = 1;
      }
    }
    __syncthreads();
    if (!sr->any_empty) return false;   // (workgroup-uniform) nothing to jump over
    for (int x = threadIdx.x; x < n; x += blockDim.x) {
      const int c = x >> 6, b = x & 63;
      int left = -1, right = n;
      unsigned long long m = sr->mask[c] & (b == 63 ? ~0ull : ((2ull << b) - 1ull));   // bits <= b
      for (int cc = c;; --cc) {
        if (m) { left = cc * 64 + 63 - __clzll((long long)m); break; }
        if (cc == 0) break;
        m = sr->mask[cc - 1];
      }
      m = sr->mask[c] & (~0ull << b);                                                    // bits >= b
      for (int cc = c;; ++cc) {
        if (m) { right = cc * 64 + __ffsll((long long)m) - 1; break; }
        if (cc == nch - 1) break;
        m = sr->mask[cc + 1];
      }
      sr->near[0][x] = (short)left;
      sr->near[1][x] = (short)right;
    }
    __syncthreads();
    return true;
  }
  return false;
}
// steps the scan may jump when its next rows are slabs xl - 1 (left) and xh + 1 (right); 0 = none (wave-uniform)
__device__ __forceinline__ int esdf_empty_run(const EsdfSlabRuns *sr, int n, int xl, int xh) {
  if (!sr) return 0;   // (uniform) lines too long for the tables, or no empty slab at all
  const int a = xl - 1, b = xh + 1;
  const int runl = a < 0 ? 0x7fff : a - (int)sr->near[0][a];        // empty slabs from a leftwards (past the end: all)
  const int runr = b >= n ? 0x7fff : (int)sr->near[1][b] - b;
  return __builtin_amdgcn_readfirstlane(min(runl, runr));
}

// the scan of one lane's block: V voxels from `first` (slab 0) times the kXB slabs from q0
template <int V>
__device__ __forceinline__ void esdf_x_scan_block(const GtopGrid &g, const int *__restrict__ fin, double *__restrict__ dist,
                                                  float *__restrict__ dist32, const int first, const int q0,
                                                  const EsdfSlabRuns *sr, const uint16_t *__restrict__ f16 = nullptr) {
  constexpr int kScanBatch = 4;   // steps per round trip
  // f16 set (the packed x sweep's exact fallback, V = 4): the packed y sweep stores its int32 output only where the
  // 16-bit copy is saturated (whole wavefronts of it), so a row is read from the 16-bit copy — exact wherever it is
  // below 0xFFFF — and from the int32 output only where it is not
  auto load_row = [&](int idx) -> IntV<V> {
    if constexpr (V == 4) {
      if (f16) {
        const uint2 pk = *reinterpret_cast<const uint2 *>(f16 + idx);
        IntV<4> r;
        r.v[0] = (int)(pk.x & 0xFFFFu); r.v[1] = (int)(pk.x >> 16);
        r.v[2] = (int)(pk.y & 0xFFFFu); r.v[3] = (int)(pk.y >> 16);
        if (max(max(r.v[0], r.v[1]), max(r.v[2], r.v[3])) == 0xFFFF) r = load_v<4>(fin + idx);
        return r;
      }
    }
    return load_v<V>(fin + idx);
  };
  const int nyz = g.ny * g.nz;
  const int n = g.nx;
  const int last = first + (n - 1) * nyz;   // the line's end voxels: first, last
  // the block's own rows (slabs past the end of the line shadow the last one; they are not stored)
  int row[kXB];
  IntV<V> best[kXB];
#pragma unroll
  for (int e = 0; e < kXB; ++e) {
    row[e] = first + min(q0 + e, n - 1) * nyz;
    best[e] = load_row(row[e]);
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
  int xl = q0, xh = min(q0 + kXB - 1, n - 1);   // the slabs of lo / hi
  int run_ahead = esdf_empty_run(sr, n, xl, xh);
  // The loads of a batch are independent and issue together; entries past the exact
  // cut-off cannot win (in(v) >= 0), so reading a few of them changes nothing.
  while (d < reach) {
    if (__mul24(d + 1, d + 1) >= worst) break;
    {
      const int skip = min(run_ahead, reach - d);
      if (skip > 0) {   // (wave-uniform) nothing but empty slabs for `skip` steps on both sides
        d += skip;
        xl = max(xl - skip, 0);
        xh = min(xh + skip, n - 1);
        lo = first + xl * nyz;
        hi = first + xh * nyz;
        run_ahead = esdf_empty_run(sr, n, xl, xh);
        continue;
      }
    }
    xl = max(xl - kScanBatch, 0);
    xh = min(xh + kScanBatch, n - 1);
    run_ahead = esdf_empty_run(sr, n, xl, xh);   // for the NEXT trip: its LDS reads ride behind this batch's loads
    IntV<V> flo[kScanBatch], fhi[kScanBatch];
#pragma unroll
    for (int u = 0; u < kScanBatch; ++u) {
      lo = max(lo - nyz, first);
      hi = min(hi + nyz, last);
      flo[u] = load_row(lo);
      fhi[u] = load_row(hi);
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


// XCD-aware order of the x sweep's work: a lane of the yz plane belongs to ONE XCD for every slab block, block
// after block — the rows a lane reads are then shared, in one L2, with the lanes of the neighbouring slab blocks
// that run at the same time (dealt linearly, every XCD walked every slab: 1.44 ms -> 0.96 ms at 400^3 with this
// order).  grid = 8 * ceil(n/kXB) * bpp workgroups, bpp = workgroups per XCD and slab block, nl = lanes per slab
// block.  Returns false for a lane without work.
template <int BLOCK = 256, bool SHADOW = false>
__device__ __forceinline__ bool esdf_x_lane(const int nl, int *fl, int *q0) {
#ifdef GTOP_ESDF_X_PARTS   // round 2's first form: XCD c owns the c-th eighth of the plane (one contiguous part)
  const int part = (nl + 7) >> 3, bpp = (part + BLOCK - 1) / BLOCK;
  const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
  const int li = (j % bpp) * BLOCK + (int)threadIdx.x;
  *q0 = (j / bpp) * kXB;
  *fl = xcd * part + li;
  return li < part && *fl < nl;
#else
  // 64-lane chunks of the plane dealt round-robin over the XCDs: the same lanes of every slab block still meet in
  // one L2, and every XCD gets an even sample of the map (with one contiguous eighth each, the XCD that owned the
  // most open space finished 10 us after the others at 200^3)
  const int cpx = (((nl + 63) >> 6) + 7) >> 3, bpp = (cpx * 64 + BLOCK - 1) / BLOCK;   // chunks, workgroups per XCD
  const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
  const int li = (j % bpp) * BLOCK + (int)threadIdx.x;
  *q0 = (j / bpp) * kXB;
  *fl = (((li >> 6) << 3) + xcd) * 64 + (li & 63);
  if (SHADOW) {   // a wavefront with any work keeps all its lanes: those past the end shadow the last lane with work
    const int wave_first = *fl - (li & 63);
    if ((li >> 6) >= cpx || wave_first >= nl) return false;
    if (*fl >= nl) *fl = nl - 1;
    return true;
  }
  return (li >> 6) < cpx && *fl < nl;
#endif
}

template <int V>
__global__ void __launch_bounds__(256)
esdf_x_kernel(const GtopGrid g, const int *__restrict__ fin, double *__restrict__ dist, float *__restrict__ dist32,
              const int *__restrict__ cnt) {
  __shared__ EsdfSlabRuns s_runs;
  const EsdfSlabRuns *sr = esdf_stage_slab_runs(&s_runs, cnt, g.nx) ? &s_runs : nullptr;
  int fl, q0;
#ifdef GTOP_ESDF_STAMPS
  const unsigned long long t0 = wall_clock64();
#endif
  if (!esdf_x_lane(g.ny * g.nz / V, &fl, &q0)) return;
  esdf_x_scan_block<V>(g, fin, dist, dist32, fl * V, q0, sr);
#ifdef GTOP_ESDF_STAMPS
  if (!getenv_stamp_y) esdf_stamp(t0, -1);
#endif
}

// The x sweep on packed 16-bit values.  Squared distances below 2^16 (255 voxels: 51 m at the reference's 0.2 m)
// fit 16 bits, and the min-plus step on two voxels is then ONE v_pk_add_u16 (saturating) + ONE v_pk_min_u16 —
// a third of the 32-bit form's instructions per candidate — and a 16-byte load brings 8 voxels.  The y sweep
// leaves min(value, 0xFFFF) beside its int32 output; a lane owns 8 voxels along z times kXB slabs.  The packed scan
// returns min(exact, 0xFFFF) for every voxel whatever the inputs: a saturated candidate (>= 0xFFFF, d^2 clamped
// likewise) can never beat a minimum below 0xFFFF and unsaturated ones are exact.  A wavefront that ends with a
// saturated minimum (more than 255 voxels of free space, a line without obstacles) takes the 32-bit scan above for
// its voxels instead.

// sqrt of an integer below 2^16, exactly rounded, without the library routine's rescaling of tiny / huge arguments and
// its special-case selects (22 instructions): the hardware's reciprocal-square-root estimate and the same Goldschmidt
// refinement (two residual corrections), 12 instructions.  Bit for bit the library's result on every value the packed
// sweep can produce (tests/test_gpu_parity.py: a 256^3 map with one obstacle in its corner holds them all).
#ifndef GTOP_ESDF_LEAN_SQRT
#define GTOP_ESDF_LEAN_SQRT 1
#endif
__device__ __forceinline__ double esdf_sqrt_u16(int n) {
#if GTOP_ESDF_LEAN_SQRT
  const double x = (double)n;
  const double y = __builtin_amdgcn_rsq(x);   // (n = 0: inf, selected away below)
  double g = x * y, h = 0.5 * y;
  const double r = fma(-h, g, 0.5);
  g = fma(g, r, g);
  h = fma(h, r, h);
  g = fma(fma(-g, g, x), h, g);
  g = fma(fma(-g, g, x), h, g);
  return n == 0 ? 0.0 : g;
#else
  return sqrt((double)n);
#endif
}

#ifndef GTOP_ESDF_X16_BLOCK
#define GTOP_ESDF_X16_BLOCK 128
#endif
#ifndef GTOP_ESDF_X16_BATCH
#define GTOP_ESDF_X16_BATCH 4
#endif
constexpr int kX16Block = GTOP_ESDF_X16_BLOCK;

__global__ void __launch_bounds__(kX16Block)
#ifdef GTOP_ESDF_X16_WPE
__attribute__((amdgpu_waves_per_eu(GTOP_ESDF_X16_WPE)))
#endif
esdf_x16_kernel(const GtopGrid g, const uint16_t *__restrict__ f16, const int *__restrict__ fin,
                double *__restrict__ dist, float *__restrict__ dist32, const int *__restrict__ cnt) {
  __shared__ EsdfSlabRuns s_runs;
  const EsdfSlabRuns *sr = esdf_stage_slab_runs(&s_runs, cnt, g.nx) ? &s_runs : nullptr;
  constexpr int kScanBatch = GTOP_ESDF_X16_BATCH;
  const int nyz = g.ny * g.nz;
  const int n = g.nx;
  int fl, q0;
#ifdef GTOP_ESDF_STAMPS
  const unsigned long long t0 = wall_clock64();
#endif
  if (!esdf_x_lane<kX16Block, true>(nyz >> 3, &fl, &q0)) return;   // (whole wavefronts only: see the epilogue)
  // first voxel of the wavefront's chunk of the plane: lane 0 is never a shadow
  const int wave_base = __builtin_amdgcn_readfirstlane(fl) << 3;
  const int first = fl << 3;
  const int last = first + (n - 1) * nyz;
  int row[kXB];
  PkV best[kXB];
#pragma unroll
  for (int e = 0; e < kXB; ++e) {
    row[e] = first + min(q0 + e, n - 1) * nyz;
    best[e] = *reinterpret_cast<const PkV *>(f16 + row[e]);
  }
  auto worst_of = [&]() {
    gtop_u16x2 w = best[0].p[0];
#pragma unroll
    for (int e = 0; e < kXB; ++e)
#pragma unroll
      for (int p = 0; p < 4; ++p) w = __builtin_elementwise_max(w, best[e].p[p]);
    return max((int)w.x, (int)w.y);
  };
  auto splat = [](int d) {   // d^2, clamped, in both halves (d is wave-uniform: scalar unit)
    const unsigned d2 = (unsigned)min(d * d, 0xFFFF);
    gtop_u16x2 r;
    r.x = (unsigned short)d2;
    r.y = (unsigned short)d2;
    return r;
  };
  {
    PkV own[kXB];
#pragma unroll
    for (int e = 0; e < kXB; ++e) own[e] = best[e];
#pragma unroll
    for (int e = 0; e < kXB; ++e)
#pragma unroll
      for (int o = 0; o < kXB; ++o)
        if (o != e) {
          const gtop_u16x2 dd = splat(e - o);
#pragma unroll
          for (int p = 0; p < 4; ++p)
            best[e].p[p] = __builtin_elementwise_min(best[e].p[p], __builtin_elementwise_add_sat(own[o].p[p], dd));
        }
  }
  int worst = worst_of();
  const int reach = max(max(q0, n - kXB - q0), 0);
  int lo = row[0], hi = row[kXB - 1], d = 0;
  int xl = q0, xh = min(q0 + kXB - 1, n - 1);   // the slabs of lo / hi
  int run_ahead = esdf_empty_run(sr, n, xl, xh);
  while (d < reach) {
    if (__mul24(d + 1, d + 1) >= worst) break;
    {
      const int skip = min(run_ahead, reach - d);
      if (skip > 0) {   // (wave-uniform) nothing but empty slabs for `skip` steps on both sides: see EsdfSlabRuns
        d += skip;
        xl = max(xl - skip, 0);
        xh = min(xh + skip, n - 1);
        lo = first + xl * nyz;
        hi = first + xh * nyz;
        run_ahead = esdf_empty_run(sr, n, xl, xh);
        continue;
      }
    }
    xl = max(xl - kScanBatch, 0);
    xh = min(xh + kScanBatch, n - 1);
    run_ahead = esdf_empty_run(sr, n, xl, xh);   // for the NEXT trip: its LDS reads ride behind this batch's loads
    PkV flo[kScanBatch], fhi[kScanBatch];
#pragma unroll
    for (int u = 0; u < kScanBatch; ++u) {
      lo = max(lo - nyz, first);
      hi = min(hi + nyz, last);
      flo[u] = *reinterpret_cast<const PkV *>(f16 + lo);
      fhi[u] = *reinterpret_cast<const PkV *>(f16 + hi);
    }
#pragma unroll
    for (int u = 0; u < kScanBatch; ++u) {
      ++d;
#pragma unroll
      for (int e = 0; e < kXB; ++e) {
        const gtop_u16x2 sl = splat(d + e), sr = splat(d + (kXB - 1 - e));
#pragma unroll
        for (int p = 0; p < 4; ++p)
          best[e].p[p] = __builtin_elementwise_min(
              best[e].p[p], __builtin_elementwise_min(__builtin_elementwise_add_sat(flo[u].p[p], sl),
                                                      __builtin_elementwise_add_sat(fhi[u].p[p], sr)));
      }
    }
    worst = worst_of();
  }
  if (__any(worst == 0xFFFF)) {   // (wave-uniform) a minimum at or past 2^16 - 1: the exact 32-bit scan instead
    esdf_x_scan_block<4>(g, fin, dist, dist32, first, q0, sr, f16);
    esdf_x_scan_block<4>(g, fin, dist, dist32, first + 4, q0, sr, f16);
    return;
  }
#ifdef GTOP_ESDF_STAMPS
  const int steps_done = d;
#endif
  // every value is below 0xFFFF here: res*sqrt(.) <= 256*res, below the 10000 of sdf_map.cpp:355-361 for any
  // resolution under 39 m — the min with 10000 is kept for the letter of it
  // A lane owns 8 consecutive voxels = 64 bytes of a row: stored from there, one store instruction would touch 64
  // lines a quarter each.  The packed results are transposed through LDS first (lane s writes its 4 words at
  // 4s .. 4s+3; lane l then reads word 64k + l, k = 0 .. 3: the voxel pair 128k + 2l of the wavefront's 512), so
  // that every store instruction writes 1 KB of whole lines (the vector-memory path was 79 % busy at 400^3, nearly
  // half of its requests these partial lines).
  __shared__ unsigned int s_tr[kX16Block / 64][256];
  const int wv = (int)threadIdx.x >> 6, ln = (int)threadIdx.x & 63;
#pragma unroll
  for (int e = 0; e < kXB; ++e) {
    if (q0 + e >= n) break;
    __builtin_amdgcn_wave_barrier();
    *reinterpret_cast<uint4 *>(&s_tr[wv][4 * ln]) = *reinterpret_cast<const uint4 *>(&best[e]);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    const int rowbase = wave_base + min(q0 + e, n - 1) * nyz;   // first voxel of the wavefront's 512 in this row
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const unsigned pk = s_tr[wv][64 * k + ln];
      const int v = 128 * k + 2 * ln;                            // voxel pair (v, v + 1) of the wavefront's chunk
      if (wave_base + v >= nyz) continue;                        // past the end of the plane (last chunk only)
      const int nn[2] = {(int)(pk & 0xFFFFu), (int)(pk >> 16)};
      double2 dv;
      {
        const double r0 = g.res * esdf_sqrt_u16(nn[0]), r1 = g.res * esdf_sqrt_u16(nn[1]);
        dv.x = r0 < 10000.0 ? r0 : 10000.0;
        dv.y = r1 < 10000.0 ? r1 : 10000.0;
      }
      *reinterpret_cast<double2 *>(dist + rowbase + v) = dv;
      if (dist32) *reinterpret_cast<float2 *>(dist32 + rowbase + v) = make_float2((float)dv.x, (float)dv.y);
    }
  }
#ifdef GTOP_ESDF_STAMPS
  if (!getenv_stamp_y) esdf_stamp(t0, steps_done);
#endif
}

}  // namespace

#ifdef GTOP_ESDF_STAMPS
extern "C" int gtop_debug_esdf_stamps(unsigned long long *out, size_t n) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_esdf_stamps), n * sizeof(unsigned long long));
}
#endif

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
  const size_t nvox = ncol * (size_t)g.nz;
  // cols, rank, cnt (+ the count of empty slabs), colany (bytes), padding to 16 bytes, the y sweep's 16-bit output, the
  // z sweep's 16-bit output (each nvox 16-bit words, rounded up to 16 bytes)
  return ((2 * ncol + (size_t)g.nx + 1 + (ncol + 3) / 4 + 3) & ~(size_t)3) + 2 * (((nvox + 1) / 2 + 3) & ~(size_t)3);   // (cnt: nx + 1)
}

hipError_t gtop_launch_esdf_build(const GtopGrid &g, const uint8_t *occ, int *tmp1, int *tmp2, int *rows,
                                  double *dist, float *dist32, hipStream_t stream) {
  const size_t ncol = (size_t)g.nx * g.ny;
  int *cols = rows, *rank = rows + ncol, *cnt = rows + 2 * ncol;
  uint8_t *colany = reinterpret_cast<uint8_t *>(rows + 2 * ncol + g.nx + 1);   // cnt[nx] = number of empty slabs
  uint16_t *f16 = reinterpret_cast<uint16_t *>(rows + ((2 * ncol + (size_t)g.nx + 1 + (ncol + 3) / 4 + 3) & ~(size_t)3));
  const size_t nvox_all = ncol * (size_t)g.nz;
  uint16_t *z16_buf = f16 + 2 * (((nvox_all + 1) / 2 + 3) & ~(size_t)3);   // behind the y sweep's 16-bit output
#ifndef GTOP_ESDF_Y16
#define GTOP_ESDF_Y16 1
#endif
#ifndef GTOP_ESDF_X16
#define GTOP_ESDF_X16 1
#endif
#ifndef GTOP_ESDF_VEC
#define GTOP_ESDF_VEC 4
#endif
  // the packed 16-bit y sweep (8 voxels per lane): where the packed x sweep runs and a lane's 8 voxels share a y
  const bool y16k = GTOP_ESDF_Y16 && GTOP_ESDF_X16 && GTOP_ESDF_VEC == 4 && g.nz % 8 == 0;
  uint16_t *z16 = y16k ? z16_buf : (uint16_t *)nullptr;
  const unsigned zblocks = (unsigned)((ncol + 3) / 4 < 65536 ? (ncol + 3) / 4 : 65536);
  switch ((g.nz + 63) >> 6) {
    case 1: hipLaunchKernelGGL(esdf_z_small_kernel<1>, dim3(zblocks), dim3(256), 0, stream, g, occ, tmp1, z16, colany, cnt + g.nx); break;
    case 2: hipLaunchKernelGGL(esdf_z_small_kernel<2>, dim3(zblocks), dim3(256), 0, stream, g, occ, tmp1, z16, colany, cnt + g.nx); break;
    case 3: hipLaunchKernelGGL(esdf_z_small_kernel<3>, dim3(zblocks), dim3(256), 0, stream, g, occ, tmp1, z16, colany, cnt + g.nx); break;
    case 4: hipLaunchKernelGGL(esdf_z_small_kernel<4>, dim3(zblocks), dim3(256), 0, stream, g, occ, tmp1, z16, colany, cnt + g.nx); break;
    case 5: hipLaunchKernelGGL(esdf_z_small_kernel<5>, dim3(zblocks), dim3(256), 0, stream, g, occ, tmp1, z16, colany, cnt + g.nx); break;
    case 6: hipLaunchKernelGGL(esdf_z_small_kernel<6>, dim3(zblocks), dim3(256), 0, stream, g, occ, tmp1, z16, colany, cnt + g.nx); break;
    case 7: hipLaunchKernelGGL(esdf_z_small_kernel<7>, dim3(zblocks), dim3(256), 0, stream, g, occ, tmp1, z16, colany, cnt + g.nx); break;
    case 8: hipLaunchKernelGGL(esdf_z_small_kernel<8>, dim3(zblocks), dim3(256), 0, stream, g, occ, tmp1, z16, colany, cnt + g.nx); break;
    default: hipLaunchKernelGGL(esdf_z_kernel, dim3(zblocks), dim3(256), 0, stream, g, occ, tmp1, z16, colany, cnt + g.nx);
  }
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return e;
#ifndef GTOP_ESDF_YLOCAL
#define GTOP_ESDF_YLOCAL 1
#endif
  const bool ylocal = GTOP_ESDF_YLOCAL && g.ny <= kYLocalMax;   // the y sweep lists its slab's candidates itself
  if (!ylocal) {
    hipLaunchKernelGGL(esdf_rows_kernel, dim3(g.nx < 65536 ? g.nx : 65536), dim3(64), 0, stream, g,
                       (const uint8_t *)colany, cols, rank, cnt);
    e = hipGetLastError();
    if (e != hipSuccess) return e;
  }
  const int nyz = g.ny * g.nz;
  const int V = (GTOP_ESDF_VEC == 4 && g.nz % 4 == 0) ? 4 : 1;   // voxels per lane (16-byte loads need nz % 4 == 0)
  const int nl = nyz / V;
  const unsigned yblocks = 8u * (unsigned)((g.nx + 7) / 8) * (unsigned)((nl + 255) / 256);
  auto x_blocks = [&](int lanes, int block) {   // esdf_x_lane's grid: 8 XCDs x slab blocks x workgroups per XCD
#ifdef GTOP_ESDF_X_PARTS
    const int per_xcd = (lanes + 7) >> 3;
#else
    const int per_xcd = ((((lanes + 63) >> 6) + 7) >> 3) * 64;
#endif
    return 8u * (unsigned)((g.nx + kXB - 1) / kXB) * (unsigned)((per_xcd + block - 1) / block);
  };
  const unsigned xblocks = x_blocks(nl, 256);
  const bool x16 = GTOP_ESDF_X16 && V == 4 && nyz % 8 == 0;   // the packed 16-bit x sweep (8 voxels per lane)
  uint16_t *y16 = x16 ? f16 : (uint16_t *)nullptr;
#define GTOP_Y_LAUNCH(VV, LL)                                                                                       \
  hipLaunchKernelGGL((esdf_y_kernel<VV, LL>), dim3(yblocks), dim3(256), 0, stream, g, (const int *)tmp1, tmp2, y16, \
                     (const int *)cols, (const int *)rank, (const int *)cnt, (const uint8_t *)colany, cnt)
  if (y16k) {
    const unsigned y16blocks = 8u * (unsigned)((g.nx + 7) / 8) * (unsigned)((nyz / 8 + 255) / 256);
    if (ylocal)
      hipLaunchKernelGGL(esdf_y16_kernel<true>, dim3(y16blocks), dim3(256), 0, stream, g, (const uint16_t *)z16, (const int *)tmp1,
                         tmp2, y16, (const int *)cols, (const int *)rank, (const int *)cnt, (const uint8_t *)colany, cnt);
    else
      hipLaunchKernelGGL(esdf_y16_kernel<false>, dim3(y16blocks), dim3(256), 0, stream, g, (const uint16_t *)z16, (const int *)tmp1,
                         tmp2, y16, (const int *)cols, (const int *)rank, (const int *)cnt, (const uint8_t *)colany, cnt);
  } else if (V == 4) {
    if (ylocal) GTOP_Y_LAUNCH(4, true);
    else GTOP_Y_LAUNCH(4, false);
  } else {
    if (ylocal) GTOP_Y_LAUNCH(1, true);
    else GTOP_Y_LAUNCH(1, false);
  }
#undef GTOP_Y_LAUNCH
  e = hipGetLastError();
  if (e != hipSuccess) return e;
  if (x16) {
    hipLaunchKernelGGL(esdf_x16_kernel, dim3(x_blocks(nyz >> 3, kX16Block)), dim3(kX16Block), 0, stream, g, (const uint16_t *)f16,
                       (const int *)tmp2, dist, dist32, (const int *)cnt);
  } else if (V == 4)
    hipLaunchKernelGGL(esdf_x_kernel<4>, dim3(xblocks), dim3(256), 0, stream, g, (const int *)tmp2, dist, dist32,
                       (const int *)cnt);
  else
    hipLaunchKernelGGL(esdf_x_kernel<1>, dim3(xblocks), dim3(256), 0, stream, g, (const int *)tmp2, dist, dist32,
                       (const int *)cnt);
  return hipGetLastError();
}
