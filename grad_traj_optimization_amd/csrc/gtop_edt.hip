// gtop_edt.hip — batched distance queries against the static field plus moving
// boxes, on gfx950.
//
// Replaces EDTEnvironment::evaluateEDTWithGrad / distToBox / minDistToAllBox
// (src/edt_environment.cpp:26-122 of EpicOne1/grad_traj_optimization; SURVEY §8f
// row f4: the other consumer of the trilinear stencil): value and gradient by
// trilinear interpolation over the 8 corner voxels, each corner's value being
// min(static distance, distance to the nearest box at the query's time); a
// negative time means "static only" (:91-94).  A box is {p0, vel, scale}, its
// centre at time t the constant-velocity prediction p0 + vel t
// (obj_predictor.h:57-66).  That file is outside the reference's build and uses
// an SDFMap API the in-tree class lacks, so the interpolation data are the
// in-tree ones (sdf_map.cpp:201-219: base index, diff, per-axis clamped corner
// loads — with time < 0 the result IS getDistWithGradTrilinear), a corner's
// position is the centre of its voxel, and a query outside the map returns -1
// with a zero gradient (sdf_map.cpp:187).
//
// One lane per query; the boxes (a few dozen at most) sit in LDS and are walked
// by every lane in step, so their reads are broadcasts.  The N x 3 query
// positions and gradients cross HBM as whole rows of the workgroup (768
// consecutive doubles, transposed through LDS at an odd stride), not as
// stride-3 accesses.  fp64.  Gather-bound: the 8 corners are 64 contiguous bytes of
// the corner records + 64 B of query/result per lane; algorithmic bytes per query
// 8*8 + 4*8 + 4*8 = 128.
// COARSE = EDTEnvironment::evaluateCoarseEDT (src/edt_environment.cpp:124-136):
// the distance of the voxel holding the position (SDFMap::getDistance(pos),
// src/sdf_map.cpp:155-164), min'ed with the box distance from the position
// itself; no interpolation, no gradient.

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "gtop_kernels.h"

namespace {

constexpr int kBoxChunk = 128;   // boxes staged in LDS per pass

template <bool COARSE>
__global__ void __launch_bounds__(256)
edt_query_kernel(const GtopGrid g, const double *__restrict__ field, const double *__restrict__ rec, int nbox,
                 const double *__restrict__ box_p0,
                 const double *__restrict__ box_vel, const double *__restrict__ box_scale, int N,
                 const double *__restrict__ pos, const double *__restrict__ time, double *__restrict__ dist,
                 double *__restrict__ grad) {
  __shared__ double bx[kBoxChunk][9];   // p0, vel, scale
  __shared__ double xyz[3 * 256];       // the workgroup's positions, later its gradients, as they lie in HBM
  const int tid = threadIdx.x;
  const size_t row0 = (size_t)blockIdx.x * 256;   // first query of the workgroup
  const int i = (int)row0 + tid;
  const bool live = i < N;
  const size_t n3 = 3 * (size_t)N;
  for (int k = 0; k < 3; ++k) {
    const size_t e = 3 * row0 + tid + 256 * k;
    xyz[tid + 256 * k] = e < n3 ? pos[e] : 0.0;
  }
  __syncthreads();
  double p[3] = {xyz[3 * tid], xyz[3 * tid + 1], xyz[3 * tid + 2]};
  const double t = live ? time[i] : -1.0;
  // isInMap, sdf_map.cpp:55-69
  bool out = false;
  for (int k = 0; k < 3; ++k) out |= (p[k] < g.min_range[k] + 1e-4) | (p[k] > g.max_range[k] - 1e-4);
  const bool dyn = live & (COARSE || !out) & (t >= 0.0);
  int idx[3];
  double diff[3] = {0, 0, 0};
  double values[2][2][2];
  double coarse = -1.0;   // COARSE: SDFMap::getDistance(pos)
  if constexpr (COARSE) {
    // posToIndex, sdf_map.cpp:71-74 (clamped for memory safety only: an in-map position indexes inside the grid)
    for (int k = 0; k < 3; ++k) idx[k] = (int)floor((p[k] - g.origin[k]) * g.res_inv);
    const int cx = min(max(idx[0], 0), g.nx - 1), cy = min(max(idx[1], 0), g.ny - 1), cz = min(max(idx[2], 0), g.nz - 1);
    const double v = field[((size_t)cx * g.ny + cy) * g.nz + cz];
    coarse = out ? -1.0 : v;
  } else {
    // base index and diff, sdf_map.cpp:201-209
    for (int k = 0; k < 3; ++k) {
      const double pm = p[k] - 0.5 * g.res;
      idx[k] = (int)floor((pm - g.origin[k]) * g.res_inv);
      diff[k] = (p[k] - ((idx[k] + 0.5) * g.res + g.origin[k])) * g.res_inv;
    }
    // The 8 corner loads of sdf_map.cpp:211-219, each index clamped per axis (getDistance(int,int,int), :166-174),
    // from the CORNER RECORDS (gtop_records.hip): the two consecutive records of levels iz and iz + 1 hold all eight,
    // clamps applied — 64 contiguous bytes, four 16-byte loads at one address (round 3 read four (z, z+1) pairs from
    // four lines: 4.24 lines of 128 bytes per query, 16 bytes used of each; now 1.25).
    {
      typedef double d2 __attribute__((ext_vector_type(2)));
      const int cx = min(max(idx[0], -1), g.nx - 1) + 1, cy = min(max(idx[1], -1), g.ny - 1) + 1;
      const int cz = min(max(idx[2], -1), g.nz - 1) + 1;
      const d2 *r = reinterpret_cast<const d2 *>(rec + 4 * (((size_t)cx * (g.ny + 1) + cy) * (g.nz + 2) + cz));
      const d2 q0 = r[0], q1 = r[1], q2 = r[2], q3 = r[3];
      values[0][0][0] = q0.x; values[0][1][0] = q0.y; values[1][0][0] = q1.x; values[1][1][0] = q1.y;
      values[0][0][1] = q2.x; values[0][1][1] = q2.y; values[1][0][1] = q3.x; values[1][1][1] = q3.y;
    }
  }
  // min over the boxes (edt_environment.cpp:26-73): at the 8 corner centres (:96-98), or at the position (:131)
  double dbox = 10000000.0;   // :64
  double vmax = 0.0;          // the largest of the 8 corner values so far
  if constexpr (!COARSE) {
#pragma unroll
    for (int x = 0; x < 2; ++x)
#pragma unroll
      for (int y = 0; y < 2; ++y)
#pragma unroll
        for (int z = 0; z < 2; ++z) vmax = fmax(vmax, values[x][y][z]);
  }
  for (int b0 = 0; b0 < nbox; b0 += kBoxChunk) {
    const int nb = min(kBoxChunk, nbox - b0);
    __syncthreads();
    for (int q = tid; q < nb * 9; q += blockDim.x) {
      const int b = q / 9, f = q - 9 * b;
      const double *src = f < 3 ? box_p0 : (f < 6 ? box_vel : box_scale);
      bx[b][f] = src[3 * (size_t)(b0 + b) + (f % 3)];
    }
    __syncthreads();
    if (dyn) {
      for (int b = 0; b < nb; ++b) {
        double bmin[3], bmax[3];
        for (int k = 0; k < 3; ++k) {
          const double c = bx[b][k] + bx[b][3 + k] * t;
          bmax[k] = c + 0.5 * bx[b][6 + k];
          bmin[k] = c - 0.5 * bx[b][6 + k];
        }
        if constexpr (COARSE) {
          double d2 = 0.0;
          for (int k = 0; k < 3; ++k) {
            const double dk = (p[k] >= bmin[k] && p[k] <= bmax[k]) ? 0.0 : fmin(fabs(p[k] - bmin[k]), fabs(p[k] - bmax[k]));
            d2 += dk * dk;
          }
          const double d = sqrt(d2);   // dist.norm()
          dbox = d < dbox ? d : dbox;
        } else {
          // per axis and corner offset: 0 inside the slab, else the distance to its nearer face (:36-40)
          double d1[3][2];
          for (int k = 0; k < 3; ++k)
            for (int o = 0; o < 2; ++o) {
              const double pt = (idx[k] + o + 0.5) * g.res + g.origin[k];
              d1[k][o] = (pt >= bmin[k] && pt <= bmax[k]) ? 0.0 : fmin(fabs(pt - bmin[k]), fabs(pt - bmax[k]));
            }
          // The corner nearest to the box takes, per axis, the smaller of the two offsets' distances, and its
          // distance is the same floating-point expression as in the loop below; every other corner's is no
          // smaller (sums of non-negative terms and sqrt round monotonically).  A box that does not undercut the
          // LARGEST of the 8 current values there cannot change any of them: skipped, bit for bit the same result —
          // with a few dozen boxes in a map most are far from a query (2^20 queries, 32 boxes: 323 -> 211 us; a
          // wavefront still pays for a box any of its 64 queries is near).
          const double near2 = fmin(d1[0][0], d1[0][1]) * fmin(d1[0][0], d1[0][1]) +
                               fmin(d1[1][0], d1[1][1]) * fmin(d1[1][0], d1[1][1]) +
                               fmin(d1[2][0], d1[2][1]) * fmin(d1[2][0], d1[2][1]);
          if (sqrt(near2) < vmax) {
            vmax = 0.0;
#pragma unroll
            for (int x = 0; x < 2; ++x)
#pragma unroll
              for (int y = 0; y < 2; ++y)
#pragma unroll
                for (int z = 0; z < 2; ++z) {
                  const double d2 = sqrt(d1[0][x] * d1[0][x] + d1[1][y] * d1[1][y] + d1[2][z] * d1[2][z]);   // dist.norm()
                  values[x][y][z] = d2 < values[x][y][z] ? d2 : values[x][y][z];
                  vmax = fmax(vmax, values[x][y][z]);
                }
          }
        }
      }
    }
  }
  if constexpr (COARSE) {
    if (live) dist[i] = (t < 0.0) ? coarse : (coarse < dbox ? coarse : dbox);   // :125-135
    return;
  }
  // trilinear value and gradient, edt_environment.cpp:104-121 (= sdf_map.cpp:221-239)
  const double v00 = (1 - diff[0]) * values[0][0][0] + diff[0] * values[1][0][0];
  const double v01 = (1 - diff[0]) * values[0][0][1] + diff[0] * values[1][0][1];
  const double v10 = (1 - diff[0]) * values[0][1][0] + diff[0] * values[1][1][0];
  const double v11 = (1 - diff[0]) * values[0][1][1] + diff[0] * values[1][1][1];
  const double v0 = (1 - diff[1]) * v00 + diff[1] * v10;
  const double v1 = (1 - diff[1]) * v01 + diff[1] * v11;
  const double d = (1 - diff[2]) * v0 + diff[2] * v1;
  double gx = (1 - diff[2]) * (1 - diff[1]) * (values[1][0][0] - values[0][0][0]);
  gx += (1 - diff[2]) * diff[1] * (values[1][1][0] - values[0][1][0]);
  gx += diff[2] * (1 - diff[1]) * (values[1][0][1] - values[0][0][1]);
  gx += diff[2] * diff[1] * (values[1][1][1] - values[0][1][1]);
  const double gy = ((1 - diff[2]) * (v10 - v00) + diff[2] * (v11 - v01)) * g.res_inv;
  const double gz = (v1 - v0) * g.res_inv;
  if (live) dist[i] = out ? -1.0 : d;
  __syncthreads();   // every lane has read its position: the tile now carries the gradients out
  xyz[3 * tid] = out ? 0.0 : gx * g.res_inv;
  xyz[3 * tid + 1] = out ? 0.0 : gy;
  xyz[3 * tid + 2] = out ? 0.0 : gz;
  __syncthreads();
  for (int k = 0; k < 3; ++k) {
    const size_t e = 3 * row0 + tid + 256 * k;
    if (e < n3) grad[e] = xyz[tid + 256 * k];
  }
}

}  // namespace

hipError_t gtop_launch_edt_query(const GtopGrid &g, const double *field, const double *rec, int nbox, const double *box_p0,
                                 const double *box_vel, const double *box_scale, int N, const double *pos,
                                 const double *time, double *dist, double *grad, hipStream_t stream) {
  if (N <= 0) return hipSuccess;
  if (grad)
    hipLaunchKernelGGL(edt_query_kernel<false>, dim3((N + 255) / 256), dim3(256), 0, stream, g, field, rec, nbox, box_p0,
                       box_vel, box_scale, N, pos, time, dist, grad);
  else   // evaluateCoarseEDT: no interpolation, no gradient
    hipLaunchKernelGGL(edt_query_kernel<true>, dim3((N + 255) / 256), dim3(256), 0, stream, g, field, rec, nbox, box_p0,
                       box_vel, box_scale, N, pos, time, dist, grad);
  return hipGetLastError();
}
