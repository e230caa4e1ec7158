// gtop_records.hip — the resident, gather-friendly copy of the distance field: CORNER RECORDS.
//
// The boundary keeps the reference's layout, `distance_buffer[x*ny*nz + y*nz + z]` (src/sdf_map.cpp:172-173): that is
// what gtop_set_sdf / gtop_get_sdf exchange and what the ESDF builder writes.  But getDistWithGradTrilinear
// (src/sdf_map.cpp:185-242) reads the 8 corners of a cell, and in that layout they are four (z, z+1) pairs in four
// different 128-byte lines: a lookup touches 4.25 lines to use 64 bytes (round 3, configs[4]: 1.69 M line requests
// per launch, the memory system's line rate the bound; the f4 query kernel 4.24 lines per query).  So the lookups read
// a second, derived copy laid out for them (the "texture-style" gather of the north_star):
//
//   record (cx, cy, cz), cx = ix + 1 in 0 .. nx, cy = iy + 1 in 0 .. ny, cz = level + 1 in 0 .. nz + 1
//     = [ D(x0,y0,z), D(x0,y1,z), D(x1,y0,z), D(x1,y1,z) ]
//       x0 = clamp(ix), x1 = clamp(ix + 1), y0 = clamp(iy), y1 = clamp(iy + 1), z = clamp(level)
//       (the per-axis index clamp of getDistance(int,int,int), src/sdf_map.cpp:166-174; ix, iy = -1 .. n-1 are the
//        base indices an in-map position can have, :201-204; level = -1 .. nz)
//   records are z-fastest: index (cx (ny+1) + cy)(nz+2) + cz.
//
// The 8 corners of base index (ix, iy, iz) are then the two CONSECUTIVE records of levels iz and iz + 1: 64
// contiguous bytes in fp64, 32 in fp32 — half a line (a lookup straddles two lines one time in four: 1.25 lines per
// lookup), fetched with one address and immediate offsets, the border clamps already applied.  Cost: 4x the field's
// bytes (200^3: 262 MB fp64, 400^3: 2.1 GB — 288 GB of HBM) and one streaming pass per map update (below).  Neighbouring
// cells along z share three quarters of their bytes, so a trajectory's consecutive samples still reuse lines.
//
// The builder: one lane per HALF record, lanes along z (the source rows are read coalesced; every store instruction
// of a wavefront writes one contiguous kilobyte of whole lines), record slab cx on XCD cx mod 8 so that the two
// source slabs a record slab reads stay in that XCD's L2.  A window form rebuilds only the records a
// changed voxel box touches (gtop_update_sdf_map_window).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

#include "gtop_device_common.h"
#include "gtop_kernels.h"

namespace {

typedef float gtop_rec_f4 __attribute__((ext_vector_type(4)));

template <typename D> struct Half2;   // half a record: the two y-corners of one x column
template <> struct Half2<double> { double lo, hi; } __attribute__((aligned(16)));
template <> struct Half2<float> { float lo, hi; } __attribute__((aligned(8)));

// records cx in [cx0, cx1], cy in [cy0, cy1], cz in [cz0, cz1] (inclusive); the whole field: 0..nx, 0..ny, 0..nz+1.
// A lane owns HALF a record — the pair (D(x_h, y0, z), D(x_h, y1, z)) of x column h of record (cx, cy, cz) — for kTY
// consecutive cy: lanes 2j and 2j+1 are the two halves of one record, consecutive lane pairs consecutive cz, so every
// store instruction of a wavefront writes one contiguous kilobyte (fp64; 512 bytes fp32) of whole lines (a lane per
// whole record made each 16-byte store instruction touch every line of a 2 KB span half: 3 TB/s of writes), and the
// row it loaded as the upper corner of one record is the lower corner of the next: one 8-byte load per 16 bytes stored.
// rec32 != nullptr: the fp32 records of the same field in the same pass (one read of the field for both).
constexpr int kTY = 8;   // (4 .. 32 rows per lane measured: 4 and 8 alike, 16 and 32 slower)

template <typename S, typename D>
__global__ void __launch_bounds__(256)
records_kernel(const S *__restrict__ field, D *__restrict__ rec, float *__restrict__ rec32, int nx, int ny, int nz, int cx0,
               int cx1, int cy0, int cy1, int cz0, int cz1, int blocks_per_slab) {
  const int xcd = (int)blockIdx.x & 7, q = (int)blockIdx.x >> 3;
  const int slab = q / blocks_per_slab, within = q - slab * blocks_per_slab;
  // slab cx on XCD cx mod 8: the first slab of the window that lives on this XCD, then every eighth
  const int first = cx0 + ((xcd - cx0) & 7);
  const int cx = first + 8 * slab;
  if (cx > cx1) return;
  const int wz2 = 2 * (cz1 - cz0 + 1), wy = cy1 - cy0 + 1, tiles_y = (wy + kTY - 1) / kTY;
  const int flat = within * 256 + (int)threadIdx.x;
  if (flat >= tiles_y * wz2) return;
  const int ty = flat / wz2, c = flat - ty * wz2, h = c & 1, cz = cz0 + (c >> 1);
  const int cya = cy0 + ty * kTY, cyb = min(cya + kTY - 1, cy1);
  const int x = h ? min(cx, nx - 1) : min(max(cx - 1, 0), nx - 1);
  const int z = min(max(cz - 1, 0), nz - 1);
  const S *col = field + (size_t)x * ny * nz + z;
  S lo = col[(size_t)min(max(cya - 1, 0), ny - 1) * nz];
  size_t hi_at = 2 * (((size_t)cx * (ny + 1) + cya) * (nz + 2) + cz) + h;   // in half records
  const size_t row = 2 * (size_t)(nz + 2);                                  // half records per cy
  // non-temporal stores throughout: the records are not read again by this kernel, and there are four bytes of them for
  // every byte of the field it reads (whole-map rebuild 179 -> 167 us at 200^3, 1 400 -> 1 333 us at 400^3, one box)
  auto store_half = [&](size_t at, S a, S b) {
    __builtin_nontemporal_store((D)a, &reinterpret_cast<Half2<D> *>(rec)[at].lo);
    __builtin_nontemporal_store((D)b, &reinterpret_cast<Half2<D> *>(rec)[at].hi);
  };
  if constexpr (std::is_same<D, double>::value) {
    if (rec32) {
      // Both precisions.  An fp32 half record is 8 bytes: stored per lane, an instruction carries 512 bytes.  Two rows
      // at a time instead: the lane pair (2j, 2j+1) — the two halves of one record — trade halves (one DPP quad swap),
      // lane 2j stores the WHOLE fp32 record of row cy, lane 2j+1 that of row cy + 1: 16 bytes per lane, every lane
      // busy, one contiguous kilobyte per instruction as for the fp64 records.
      for (int cy = cya; cy <= cyb; cy += 2, hi_at += 2 * row) {
        const bool two = cy + 1 <= cyb;       // (the same for both lanes of a pair: they share their tile)
        const S hiA = col[(size_t)min(cy, ny - 1) * nz];
        const S hiB = two ? col[(size_t)min(cy + 1, ny - 1) * nz] : hiA;
        store_half(hi_at, lo, hiA);
        if (two) store_half(hi_at + row, hiA, hiB);
        const float s0 = h ? (float)lo : (float)hiA, s1 = h ? (float)hiA : (float)hiB;     // what the partner needs
        const float r0 = gtop_dpp_move<0xb1>(s0), r1 = gtop_dpp_move<0xb1>(s1);             // quad_perm [1,0,3,2]
        if (two) {
          const gtop_rec_f4 out = h ? (gtop_rec_f4){r0, r1, (float)hiA, (float)hiB} : (gtop_rec_f4){(float)lo, (float)hiA, r0, r1};
          __builtin_nontemporal_store(out, reinterpret_cast<gtop_rec_f4 *>(rec32) + ((hi_at - h) / 2 + (h ? row / 2 : 0)));
        } else {                              // a last single row: its halves as 8-byte stores
          __builtin_nontemporal_store((float)lo, &reinterpret_cast<Half2<float> *>(rec32)[hi_at].lo);
          __builtin_nontemporal_store((float)hiA, &reinterpret_cast<Half2<float> *>(rec32)[hi_at].hi);
        }
        lo = hiB;
      }
      return;
    }
  }
  for (int cy = cya; cy <= cyb; ++cy, hi_at += row) {
    const S hi = col[(size_t)min(cy, ny - 1) * nz];
    store_half(hi_at, lo, hi);
    lo = hi;
  }
}

template <typename S, typename D>
hipError_t launch_records(const GtopGrid &g, const S *field, D *rec, float *rec32, const int lo[3], const int hi[3],
                          hipStream_t s) {
  const int wx = hi[0] - lo[0] + 1, wy = hi[1] - lo[1] + 1, wz = hi[2] - lo[2] + 1;
  if (wx <= 0 || wy <= 0 || wz <= 0) return hipSuccess;
  const long long per_slab = ((long long)((wy + kTY - 1) / kTY) * 2 * wz + 255) / 256;
  const long long slabs_per_xcd = (wx + 7) / 8;
  const long long grid = 8 * slabs_per_xcd * per_slab;
  if (grid > 0x7fffffffLL) return hipErrorInvalidValue;
  hipLaunchKernelGGL((records_kernel<S, D>), dim3((unsigned)grid), dim3(256), 0, s, field, rec, rec32, g.nx, g.ny, g.nz,
                     lo[0], hi[0], lo[1], hi[1], lo[2], hi[2], (int)per_slab);
  return hipGetLastError();
}

}  // namespace

size_t gtop_record_count(const GtopGrid &g) { return (size_t)(g.nx + 1) * (g.ny + 1) * (g.nz + 2); }

// Records of the voxel box [vlo, vhi] (inclusive voxel indices; NULL = the whole field): every record that holds one
// of its voxels — voxel x sits in records cx = x and x + 1 (and, clamped, in the border records beyond the grid's
// first / last voxel), likewise y; level z in cz = z + 1 (and in the padding levels at the z ends).
template <typename S, typename D>
hipError_t gtop_launch_build_records(const GtopGrid &g, const S *field, D *rec, float *rec32, const int *vlo,
                                     const int *vhi, hipStream_t stream) {
  const int n[3] = {g.nx, g.ny, g.nz};
  int lo[3], hi[3];
  for (int k = 0; k < 3; ++k) {
    const int a = vlo ? vlo[k] : 0, b = vhi ? vhi[k] : n[k] - 1;
    if (k < 2) {
      lo[k] = a <= 0 ? 0 : a;              // voxel a is x1 of record a and x0 of record a + 1; voxel 0 also fills record 0
      hi[k] = b >= n[k] - 1 ? n[k] : b + 1;
    } else {
      lo[k] = a <= 0 ? 0 : a + 1;          // level z is record z + 1; level 0 also fills the padding record 0
      hi[k] = b >= n[k] - 1 ? n[k] + 1 : b + 1;
    }
  }
  return launch_records<S, D>(g, field, rec, rec32, lo, hi, stream);
}

template hipError_t gtop_launch_build_records<double, double>(const GtopGrid &, const double *, double *, float *,
                                                              const int *, const int *, hipStream_t);
template hipError_t gtop_launch_build_records<double, float>(const GtopGrid &, const double *, float *, float *,
                                                             const int *, const int *, hipStream_t);
template hipError_t gtop_launch_build_records<float, float>(const GtopGrid &, const float *, float *, float *,
                                                            const int *, const int *, hipStream_t);
