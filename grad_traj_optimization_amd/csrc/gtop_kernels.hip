// gtop_kernels.hip — hand-written gfx950 (CDNA4) kernels for the batched
// cost/gradient callback of GTOP.  No MFMA: this is a stencil/gather path.
//
// What one launch computes, per trajectory b (reference lines are file:line
// into EpicOne1/grad_traj_optimization):
//   cost_b, grad_b = GradTrajOptimizer::getCostAndGradient(x_b)
//                    (src/grad_traj_optimizer.cpp:281-432)
// with the distance query SDFMap::getDistWithGradTrilinear
// (src/sdf_map.cpp:185-242) inlined.
//
// Formulation.  The reference multiplies dense L (6m x 3m+3)
// and R ((3m+3)^2) that it built once from segment_time
// (src/qp_generator.cpp:357-405).  A is block diagonal, so L's row-block s is
// A_s^-1 (quintic Hermite, closed form) scattered onto the columns of
// waypoints s and s+1, and d'Rd = sum_s c_s' Q_s c_s.  The kernel therefore
// takes (x, Df, T) and works per segment.  ONE kernel family serves every
// launch, gtop_eval_wave_kernel: a wavefront owns one or two
// whole trajectories — or as many as fit —, a segment is sampled by LPS = 30/SPL
// adjacent lanes with SPL samples each (SPL = 3: ten lanes, up to 6 segments;
// SPL = 6: five lanes, up to 12 segments at a time; SPL = 10 / 30: three lanes /
// one lane, 21 / 64 segment slots shared by 21/m / 64/m trajectories), and the
// evaluation is one dependent chain:
//   * every lane of a segment forms the segment's 18 polynomial coefficients
//     c_{s,k} = A_s^-1 d_{s,k} in registers;
//   * per sample: position/velocity (float round trip), trilinear field lookup
//     with analytic gradient, exp penalty; each sample adds
//     w1_k*[t^j] + w2_k*[j t^(j-1)] to the lane's 18-entry coefficient-space
//     gradient, which lane 0 of the segment STARTED at the jerk term ws*2Qc;
//   * after its samples the lane applies A_s^-T (linear, commutes with the
//     sums); the lanes write to an LDS tile, and each free variable = end of
//     segment w-1 + start of segment w, +1e-5, is summed from it; the scalar
//     cost is summed the same way, +1e-3;
//   * with the optimizer state as template argument the same wavefront then
//     runs the CCSA-MMA update and evaluates again (the whole loop, one launch).
// All structural zeros the reference multiplies through are skipped; nothing
// else is approximated.

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

#include "gtop_device_common.h"
#include "gtop_kernels.h"

namespace {

constexpr int kSamples = 30;     // src/grad_traj_optimizer.cpp:351
constexpr int kRedVals = 19;     // 18 gradient entries + 1 cost per sample
// row stride of the LDS tile: the busy lanes of a wave (LPS*SPW of 64), made odd
constexpr int red_stride(int spl) {
  const int lps = kSamples / spl, busy = lps * (64 / lps);
  return busy | 1;
}

// Diagnostic build (-DGTOP_STAMPS): s_memtime at the phase boundaries of lane 0
// of wave 0 of the first 4096 workgroups, into a buffer of its own that nothing
// else reads.  Never defined in the shipped library.
#ifdef GTOP_STAMPS
__device__ unsigned long long g_gtop_stamps[4096][16];
// -DGTOP_STAMPS=2: only the wavefront's first and last stamp (0 and 11) — two s_memtime instead of twelve, so that the
// wavefront's lifetime is (nearly) the uninstrumented one; slots 12 / 13 then hold the constant-rate wall clock
// (wall_clock64, 100 MHz) at the same two points: lifetimes in seconds, and the shader clock's rate from the two.
#if GTOP_STAMPS == 2
#define GTOP_STAMP_WANTED(i) ((i) == 0 || (i) == 11)
#else
#define GTOP_STAMP_WANTED(i) true
#endif
#define GTOP_STAMP(i)                                                                          \
  do {                                                                                         \
    if (GTOP_STAMP_WANTED(i)) {                                                                \
      unsigned long long t_;                                                                   \
      __builtin_amdgcn_sched_barrier(0);                                                       \
      asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");              \
      __builtin_amdgcn_sched_barrier(0);                                                       \
      if (threadIdx.x == 0 && blockIdx.x < 4096) g_gtop_stamps[blockIdx.x][i] = t_;            \
      if (((i) == 0 || (i) == 11) && threadIdx.x == 0 && blockIdx.x < 4096)                    \
        g_gtop_stamps[blockIdx.x][(i) == 0 ? 12 : 13] = wall_clock64();                        \
    }                                                                                          \
  } while (0)
// where the wavefront runs (HW_ID: wave/simd/cu/sh/se; XCC_ID), into stamp slots 14 and 15
#define GTOP_STAMP_HWID()                                                                      \
  do {                                                                                         \
    unsigned hw_, xcc_;                                                                        \
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)\n\ts_getreg_b32 %1, hwreg(HW_REG_XCC_ID)" \
                 : "=s"(hw_), "=s"(xcc_));                                                      \
    if (threadIdx.x == 0 && blockIdx.x < 4096) {                                               \
      g_gtop_stamps[blockIdx.x][14] = hw_;                                                     \
      g_gtop_stamps[blockIdx.x][15] = xcc_;                                                    \
    }                                                                                          \
  } while (0)
#else
#define GTOP_STAMP(i)
#define GTOP_STAMP_HWID()
#endif

// A phase boundary the compiler holds: the scheduling barrier alone only binds the machine scheduler, and by then
// instruction selection has already placed the (side-effect-free) loads and arithmetic wherever it liked — in the
// shipped build both barriers of the sample loop had ended up next to each other in front of the input wait and the
// first sample's corner loads were waited for 20 instructions after their issue.  The empty asm with a memory
// clobber orders the loads at the IR level; the barriers then keep the machine scheduler from undoing it.
#ifndef GTOP_NO_PHASE_FENCE
#define GTOP_PHASE_FENCE()                  \
  do {                                      \
    __builtin_amdgcn_sched_barrier(0);      \
    asm volatile("" ::: "memory");          \
    __builtin_amdgcn_sched_barrier(0);      \
  } while (0)
#else
#define GTOP_PHASE_FENCE() __builtin_amdgcn_sched_barrier(0)
#endif

// Diagnostic (-DGTOP_MARKS): comment markers in the ISA at region boundaries of gtop_eval_wave_kernel (pinned with
// scheduling barriers) so that instructions can be counted per region with tools/isa_regions.py.
#ifdef GTOP_MARKS
#define GTOP_MARK(n)                                \
  do {                                              \
    __builtin_amdgcn_sched_barrier(0);              \
    asm volatile("; GTOP_MARK " #n ::: "memory");   \
    __builtin_amdgcn_sched_barrier(0);              \
  } while (0)
#else
#define GTOP_MARK(n)
#endif

template <typename R> struct Pair { R x, y; } __attribute__((packed));
template <typename R> constexpr bool kIsF32 = false;
template <> constexpr bool kIsF32<float> = true;

// exp for the per-sample penalty (src/grad_traj_optimizer.cpp:509,:514): one
// range reduction x = k ln2 + r, |r| <= ln2/2, a degree-9 polynomial in Horner
// form and ldexp — about a third of the instructions of the library routine.
// The polynomial is 1 + r + r^2/2 + r^3 q(r) with q the degree-6 minimax fit
// (relative error of the whole: 2.7e-14 on the interval, Lawson iteration in
// 50-digit arithmetic, checked against exp in double Horner evaluation; the
// degree-11 Taylor polynomial it replaces: 6e-15, four instructions more per
// sample) — the three low coefficients stay the inline operands 0.5, 1, 1.  |x| beyond the fp64 exponent range
// saturates to 0 / inf through v_cvt_i32_f64 (saturating) and v_ldexp_f64;
// NaN propagates through p.
// The constants live in a struct so that the latency variant can pin them
// in VGPRs (ExpConsts::pin): 24 literal dwords less to hold in SGPRs, which that body
// otherwise spills to VGPR lanes and re-materialises with s_mov pairs.
struct ExpConsts {
  double inv_ln2 = 1.4426950408889634074, ln2_hi = -6.93147180369123816490e-01, ln2_lo = -1.90821492927058770002e-10;
  static constexpr int kN = 7;
  double c[kN] = {2.7452117538893314e-06,    // r^9  (1/9!  = 2.7557e-06)
                  2.4872720083484436e-05,    // r^8  (1/8!  = 2.4802e-05)
                  1.9841623829444966e-04,    // r^7
                  1.3888830888367963e-03,    // r^6
                  8.3333330482387603e-03,    // r^5
                  4.1666666813179924e-02,    // r^4
                  1.6666666667306995e-01};   // r^3
  __device__ __forceinline__ void pin() {
    asm volatile("" : "+v"(inv_ln2), "+v"(ln2_hi), "+v"(ln2_lo));
#pragma unroll
    for (int i = 0; i < kN; ++i) asm volatile("" : "+v"(c[i]));
  }
};
__device__ __forceinline__ double penalty_exp(double x, const ExpConsts &K) {
  const double k = rint(x * K.inv_ln2);            // x / ln2
  double r = fma(k, K.ln2_hi, x);
  r = fma(k, K.ln2_lo, r);
  double p = K.c[0];
#pragma unroll
  for (int i = 1; i < ExpConsts::kN; ++i) p = fma(p, r, K.c[i]);
  p = fma(p, r, 0.5);
  p = fma(p, r, 1.0);
  p = fma(p, r, 1.0);
  return ldexp(p, (int)k);   // v_cvt_i32_f64: saturating, NaN -> 0
}
__device__ __forceinline__ float penalty_exp(float x, const ExpConsts &) { return expf(x); }

// 1/x: hardware estimate + two Newton steps (fp64), full-precision divide (fp32)
__device__ __forceinline__ double fast_rcp(double x) {
  double y = __builtin_amdgcn_rcp(x);
  y = fma(fma(-x, y, 1.0), y, y);
  y = fma(fma(-x, y, 1.0), y, y);
  return y;
}
__device__ __forceinline__ float fast_rcp(float x) { return 1.0f / x; }
// 1/x to ~1e-15: hardware estimate + one Newton step (used where the result only scales a gradient term)
__device__ __forceinline__ double quick_rcp(double x) {
  const double y = __builtin_amdgcn_rcp(x);
  return fma(fma(-x, y, 1.0), y, y);
}
__device__ __forceinline__ float quick_rcp(float x) { return 1.0f / x; }
template <typename R> __device__ __forceinline__ R gfma(R a, R b, R c);
template <> __device__ __forceinline__ double gfma<double>(double a, double b, double c) { return fma(a, b, c); }
template <> __device__ __forceinline__ float gfma<float>(float a, float b, float c) { return fmaf(a, b, c); }

// sqrt of a squared speed (src/grad_traj_optimizer.cpp:358).  fp64: the same
// v_rsq_f64 + coupled Newton refinement the library routine uses, without its
// exponent rescaling for arguments below 2^-767: the argument is clamped to
// 1e-200 instead, which changes nothing that survives the "+ 1e-5" of :358.
__device__ __forceinline__ double speed_sqrt(double s) {
  s = fmax(s, 1e-200);
  const double y = __builtin_amdgcn_rsq(s);
  double g = s * y, h = 0.5 * y;
  const double r = fma(-h, g, 0.5);
  g = fma(g, r, g);
  h = fma(h, r, h);
  g = fma(fma(-g, g, s), h, g);
  g = fma(fma(-g, g, s), h, g);
  return g;
}
__device__ __forceinline__ float speed_sqrt(float s) { return sqrtf(s); }
template <typename R> __device__ __forceinline__ R gfloor(R v);
template <> __device__ __forceinline__ double gfloor<double>(double v) { return floor(v); }
template <typename R> __device__ __forceinline__ R gabs(R v);
template <> __device__ __forceinline__ double gabs<double>(double v) { return fabs(v); }
template <> __device__ __forceinline__ float gabs<float>(float v) { return fabsf(v); }

// The reference stores pos/vel in `float` locals and widens them again
// (src/grad_traj_optimizer.cpp:457-465, :477-485).
__device__ __forceinline__ double round_through_float(double v) { return (double)(float)v; }
__device__ __forceinline__ float round_through_float(float v) { return v; }

// SDFMap::getDistWithGradTrilinear, src/sdf_map.cpp:185-242.
// Out of map (src/sdf_map.cpp:55-69, :187): dist = -1; the reference leaves
// grad uninitialised there, this build defines it as 0.
// Branch-free: the record index is clamped, so the loads are always in bounds
// and the out-of-map case is a final select.
//
// The field the lookups read is the resident CORNER-RECORD copy (gtop_records.hip), not the
// boundary's z-fastest buffer: record (cx, cy, cz), cx = ix + 1 in 0 .. nx, cy = iy + 1 in 0 .. ny, cz = level + 1 in
// 0 .. nz + 1, holds the four (x, y) corners of base index (ix, iy) at z level `level`, every index clamped per axis as
// getDistance(int,int,int) clamps it (src/sdf_map.cpp:166-174) —
//     [ D(x0,y0,z), D(x0,y1,z), D(x1,y0,z), D(x1,y1,z) ],  x0 = clamp(ix), x1 = clamp(ix+1), ..., z = clamp(level)
// — records z-fastest, so the 8 corners of a lookup (:211-219) are the TWO CONSECUTIVE records of levels iz and
// iz + 1: 64 contiguous bytes in fp64 (four 16-byte loads at one address + 0/16/32/48), 32 in fp32 (two), half a
// 128-byte line, where the z-fastest field cost four lines of which 16 bytes each were used.  The border clamps
// live in the records: no border selects, no weight clamp, no z-gradient select here — at a z border both levels hold
// the same voxels, so the value is v0 + dz*0 and the z-gradient 0 exactly as the reference computes them.
// The base index of an in-map position is -1 .. n-1 per axis (:201-204); the clamp keeps an out-of-map position's
// loads inside the buffer.
// WIDE = false (the host checks (nx+1)(ny+1) < 2^23, nz+2 < 2^23, records below 4 GiB): 24-bit multiply-adds (full
// rate; v_mul_lo_u32 is not) and a uniform base + 32-bit byte offset per lane.  WIDE = true: 64-bit indices.
// clamp(v, -1, hi) as one v_med3_i32 (the compiler forms med3 only between constants; min(max()) is two instructions,
// three times per lookup).  Not volatile: free to move and to be eliminated like any arithmetic.  Used by the latency
// variant's hand-issued lookups only: in the 168-VGPR bodies the asm's operand constraints cost registers the
// allocator does not have, there the clamp stays min(max()).
__device__ __forceinline__ int clamp_index_med3(int v, int hi) {
#ifdef GTOP_NO_MED3
  return min(max(v, -1), hi);
#else
  int r;
  asm("v_med3_i32 %0, %1, -1, %2" : "=v"(r) : "v"(v), "s"(hi));
  return r;
#endif
}
__device__ __forceinline__ int clamp_index(int v, int hi) { return min(max(v, -1), hi); }

// index of the lookup's first record, less K0 = ((ny+1) + 1)(nz+2) + 1 (the "+1" of every axis, folded into one
// constant the caller adds): (cx (ny+1) + cy)(nz+2) + cz with cx, cy, cz the clamped BASE indices, -1 .. n-1
template <bool MED3>
__device__ __forceinline__ int record_index(int ix, int iy, int iz, int nx, int ny, int nz) {
  const int cx = MED3 ? clamp_index_med3(ix, nx - 1) : clamp_index(ix, nx - 1);
  const int cy = MED3 ? clamp_index_med3(iy, ny - 1) : clamp_index(iy, ny - 1);
  const int cz = MED3 ? clamp_index_med3(iz, nz - 1) : clamp_index(iz, nz - 1);
  return __mul24(__mul24(cx, ny + 1) + cy, nz + 2) + cz;
}

// The query is split in two so that a caller can put several lookups in flight
// before consuming the first: sdf_issue does the index arithmetic and issues the
// loads, sdf_blend is the trilinear arithmetic on the loaded corners.
template <typename R> struct SdfTap {
  R v[8];         // the two records: v000, v010, v100, v110 (level iz), v001, v011, v101, v111 (level iz + 1); v[x][y][z]
  R dx, dy, dz;   // interpolation weights (sdf_map.cpp:206-209)
};

typedef double gtop_d2 __attribute__((ext_vector_type(2)));
typedef float gtop_f4 __attribute__((ext_vector_type(4)));

template <typename R, bool WIDE>
__device__ __forceinline__ void record_loads(const GtopKernelArgs<R> &a, int ix, int iy, int iz, R (&v)[8]) {
  const int nx = a.nx, ny = a.ny, nz = a.nz;
  const char *p;
  if constexpr (!WIDE) {
    const int k0 = (ny + 2) * (nz + 2) + 1;
    const uint32_t off = (uint32_t)(record_index<false>(ix, iy, iz, nx, ny, nz) + k0) * (uint32_t)(4 * sizeof(R));
    p = reinterpret_cast<const char *>(a.sdf) + off;
  } else {
    const size_t idx = ((size_t)(clamp_index(ix, nx - 1) + 1) * (size_t)(ny + 1) + (size_t)(clamp_index(iy, ny - 1) + 1)) *
                           (size_t)(nz + 2) + (size_t)(clamp_index(iz, nz - 1) + 1);
    p = reinterpret_cast<const char *>(a.sdf + 4 * idx);
  }
  if constexpr (sizeof(R) == 8) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const gtop_d2 t = *reinterpret_cast<const gtop_d2 *>(p + 16 * q);
      v[2 * q] = t.x;
      v[2 * q + 1] = t.y;
    }
  } else {
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const gtop_f4 t = *reinterpret_cast<const gtop_f4 *>(p + 16 * q);
      v[4 * q] = t.x; v[4 * q + 1] = t.y; v[4 * q + 2] = t.z; v[4 * q + 3] = t.w;
    }
  }
}

// isInMap (sdf_map.cpp:55-69) of one position, exactly as the reference tests it
template <typename R>
__device__ __forceinline__ bool out_of_map(const R (&lo)[3], const R (&hi)[3], R px, R py, R pz) {
  return (px < lo[0]) | (py < lo[1]) | (pz < lo[2]) | (px > hi[0]) | (py > hi[1]) | (pz > hi[2]);
}


// isInMap's box (sdf_map.cpp:55-69, margins included)
template <typename R> struct MapBox {
  R lo[3], hi[3];
};
// posToIndex's constants (sdf_map.cpp:71-74, :201-204): origin, res/2, 1/res — in DOUBLE whatever the kernel's
// arithmetic type.  Which cell a sample reads is decided as the reference decides it: the position, a `float` value
// (:457-465), widened, and floor((pos - res/2 - origin) / res) in double.  The interpolant is continuous across cell
// faces, its gradient is not, so a cell chosen in fp32 arithmetic (round 3's fp32 bodies) gave a sample within fp32's
// rounding of a face — one in 1e5 — the neighbouring cell's gradient: a row's gradient off by that sample's weight.
struct IndexBox {
  double org[3], half, rinv;
};

template <typename R, bool WIDE>
__device__ __forceinline__ SdfTap<R> sdf_issue(const GtopKernelArgs<R> &a, const IndexBox &box, double px, double py, double pz) {
  SdfTap<R> tp;
  const double rinv = box.rinv, half = box.half;
  // posToIndex(pos - 0.5 res)  (:201-204 -> :71-74)
  const double tx = (px - half) - box.org[0], ty = (py - half) - box.org[1], tz = (pz - half) - box.org[2];
  const double ux = tx * rinv, uy = ty * rinv, uz = tz * rinv;
  const double fx = floor(ux), fy = floor(uy), fz = floor(uz);
  record_loads<R, WIDE>(a, (int)fx, (int)fy, (int)fz, tp.v);
  // indexToPos (:76-78) and diff (:209): (pos - centre(idx)) / res is the fractional
  // part of u (equal up to a few ulp of u, ~1e-14 of a voxel).  Written as the fused form the compiler
  // contracts `u - floor(u)` to where it can: every body, however it is scheduled, takes the same bits.
  tp.dx = (R)fma(tx, rinv, -fx);
  tp.dy = (R)fma(ty, rinv, -fy);
  tp.dz = (R)fma(tz, rinv, -fz);
  return tp;
}

// The same with the four 16-byte loads issued by hand (fp64, 32-bit offsets): the lone-wavefront body wants all of a
// lane's corner loads in flight BEFORE the arithmetic that does not need them, and the compiler — free to sink
// side-effect-free loads of a read-only noalias field, and keen to, at 232 VGPRs — put each sample's loads right in
// front of their use (round 2: the first sample's loads were waited for 20 instructions after their issue, whatever
// scheduling barriers said).  A volatile asm keeps its place among the phase fences; the compiler does not count
// these loads, so the caller waits for them itself (gtop_wait_pairs) before it reads `raw`
// (tools/kernel_resources.py check_asm_loads walks the ISA: nothing names their registers before that wait).
template <int BYTE_OFF>
__device__ __forceinline__ gtop_d2 asm_load_pair(const void *base, uint32_t byte_off) {
  gtop_d2 v;
  asm volatile("global_load_dwordx4 %0, %1, %2 offset:%3" : "=v"(v) : "v"(byte_off), "s"(base), "n"(BYTE_OFF));
  return v;
}

// s_waitcnt vmcnt(LEFT) on the hand-issued loads (in issue order, LEFT of them may still be in flight); `raw` passes
// through, so that nothing reads it before the wait, and `after0/1` are values that must be complete first.
template <int LEFT>
__device__ __forceinline__ void gtop_wait_pairs(gtop_d2 (&raw)[4], double after0, double after1) {
  asm volatile("s_waitcnt vmcnt(%6)"
               : "+v"(raw[0]), "+v"(raw[1]), "+v"(raw[2]), "+v"(raw[3])
               : "v"(after0), "v"(after1), "n"(LEFT));
}

__device__ __forceinline__ SdfTap<double> sdf_issue_asm(const GtopKernelArgs<double> &a, const IndexBox &box,
                                                        double px, double py, double pz, gtop_d2 (&raw)[4]) {
  typedef double R;
  SdfTap<R> tp;
  const R rinv = box.rinv, half = box.half;
  const R tx = (px - half) - box.org[0], ty = (py - half) - box.org[1], tz = (pz - half) - box.org[2];
  const R ux = tx * rinv, uy = ty * rinv, uz = tz * rinv;
  const R fx = gfloor(ux), fy = gfloor(uy), fz = gfloor(uz);
  tp.dx = gfma(tx, rinv, -fx);
  tp.dy = gfma(ty, rinv, -fy);
  tp.dz = gfma(tz, rinv, -fz);
  // record_loads<double, false>, loads by hand: one address, four immediate offsets
  const int nx = a.nx, ny = a.ny, nz = a.nz;
  const int k0 = (ny + 2) * (nz + 2) + 1;
  const uint32_t off = (uint32_t)(record_index<true>((int)fx, (int)fy, (int)fz, nx, ny, nz) + k0) * 32u;
  raw[0] = asm_load_pair<0>(a.sdf, off);
  raw[1] = asm_load_pair<16>(a.sdf, off);
  raw[2] = asm_load_pair<32>(a.sdf, off);
  raw[3] = asm_load_pair<48>(a.sdf, off);
  return tp;
}

// Trilinear value and gradient (src/sdf_map.cpp:211-241) in difference form:
// every interpolation is a + w (b - a), and the corner differences it needs are
// the ones the gradient is made of, so nothing is computed twice.  The gradient
// comes back UNSCALED — in distance per voxel; the caller folds 1/resolution
// (:231-239) into the weight that multiplies it.
template <typename R>
__device__ __forceinline__ R sdf_blend(const SdfTap<R> &tp, R &gx, R &gy, R &gz) {
  const R dx = tp.dx, dy = tp.dy, dz = tp.dz;
  // values[x][y][z]
  const R v000 = tp.v[0], v010 = tp.v[1], v001 = tp.v[4], v011 = tp.v[5];
  const R d00 = tp.v[2] - v000, d01 = tp.v[6] - v001;   // x-differences of the four (y,z) edges
  const R d10 = tp.v[3] - v010, d11 = tp.v[7] - v011;
  const R v00 = gfma(dx, d00, v000), v01 = gfma(dx, d01, v001);   // :221-224
  const R v10 = gfma(dx, d10, v010), v11 = gfma(dx, d11, v011);
  const R e0 = v10 - v00, e1 = v11 - v01;                          // y-differences
  const R v0 = gfma(dy, e0, v00), v1 = gfma(dy, e1, v01);          // :226-227
  const R dd = v1 - v0;                                            // z-difference (:231)
  const R dist = gfma(dz, dd, v0);                                 // :229
  gy = gfma(dz, e1 - e0, e0);                                      // :232-233
  const R h0 = gfma(dy, d10 - d00, d00), h1 = gfma(dy, d11 - d01, d01);
  gx = gfma(dz, h1 - h0, h0);                                      // :234-239
  gz = dd;   // (at a z border both levels hold the same voxels: 0, as :231 gives it)
  return dist;   // (out of the map: the caller overrides value and gradient, sdf_map.cpp:187)
}

// sum of N consecutive values as a balanced tree (depth log2 N instead of an
// N-long dependent chain)
template <typename R, int N>
__device__ __forceinline__ R tree_sum(const R *p) {
  if constexpr (N == 1) return p[0];
  else if constexpr (N == 2) return p[0] + p[1];
  else return tree_sum<R, N / 2>(p) + tree_sum<R, N - N / 2>(p + N / 2);
}

// ---------------------------------------------------------------------------
// Packed-fp32 sample path.  On gfx950 a wave64 VALU instruction occupies its
// SIMD for 4 cycles whether it is fp64, scalar fp32 or PACKED fp32
// (v_pk_fma_f32 & co.: two fp32 per lane) — measured, tools/ubench/pk_rate.hip.
// So the fp32 path evaluates TWO samples of a lane at once in float2 registers:
// the arithmetic (polynomials, trilinear blend, weights, the 18 accumulators)
// issues as v_pk_* and costs half; only index/clamp/load/select and the
// transcendental ops stay per component.
// ---------------------------------------------------------------------------
typedef float f2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ f2 splat(float v) { return (f2){v, v}; }
__device__ __forceinline__ f2 pk_fma(f2 a, f2 b, f2 c) { return __builtin_elementwise_fma(a, b, c); }

// two SDFMap::getDistWithGradTrilinear queries (src/sdf_map.cpp:185-242) at the positions pA, pB (`float` values, as
// the reference holds them); the cell and the weights from double arithmetic on the widened position (IndexBox)
template <bool WIDE>
__device__ __forceinline__ f2 sdf_query_pair(const GtopKernelArgs<float> &a, const IndexBox &box, const float (&pA)[3],
                                             const float (&pB)[3], f2 &gx, f2 &gy, f2 &gz) {
  int idx[2][3];
  float w[2][3];
#pragma unroll
  for (int c = 0; c < 2; ++c)
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const double t = ((double)(c ? pB[k] : pA[k]) - box.half) - box.org[k];   // posToIndex(pos - 0.5 res), :201-204
      const double f = floor(t * box.rinv);
      idx[c][k] = (int)f;
      w[c][k] = (float)fma(t, box.rinv, -f);                                     // diff, :206-209
    }
  const f2 dx = {w[0][0], w[1][0]}, dy = {w[0][1], w[1][1]}, dz = {w[0][2], w[1][2]};
  // the two records of each sample (record_loads): border clamps are in the records, dz is the plain fraction
  float c8[2][8];
#pragma unroll
  for (int c = 0; c < 2; ++c) record_loads<float, WIDE>(a, idx[c][0], idx[c][1], idx[c][2], c8[c]);
  const f2 v000 = {c8[0][0], c8[1][0]}, v010 = {c8[0][1], c8[1][1]};
  const f2 v100 = {c8[0][2], c8[1][2]}, v110 = {c8[0][3], c8[1][3]};
  const f2 v001 = {c8[0][4], c8[1][4]}, v011 = {c8[0][5], c8[1][5]};
  const f2 v101 = {c8[0][6], c8[1][6]}, v111 = {c8[0][7], c8[1][7]};
  // difference form, gradient unscaled (per voxel): see sdf_blend
  const f2 d00 = v100 - v000, d01 = v101 - v001, d10 = v110 - v010, d11 = v111 - v011;
  const f2 v00 = pk_fma(dx, d00, v000), v01 = pk_fma(dx, d01, v001);   // :221-224
  const f2 v10 = pk_fma(dx, d10, v010), v11 = pk_fma(dx, d11, v011);
  const f2 e0 = v10 - v00, e1 = v11 - v01;
  const f2 v0 = pk_fma(dy, e0, v00), v1 = pk_fma(dy, e1, v01);         // :226-227
  const f2 dd = v1 - v0;                                               // :231
  f2 dist = pk_fma(dz, dd, v0);                                        // :229
  gy = pk_fma(dz, e1 - e0, e0);                                        // :232-233
  const f2 h0 = pk_fma(dy, d10 - d00, d00), h1 = pk_fma(dy, d11 - d01, d01);
  gx = pk_fma(dz, h1 - h0, h0);                                        // :234-239
  gz = dd;   // (0 at a z border: both levels hold the same voxels)
  return dist;
}

// Two samples (tA, tB) of one segment: everything phase 2 does per sample
// (src/grad_traj_optimizer.cpp:353-381), accumulated component-wise into
// acc2[19]; the caller adds the two components after its loop.
// The POSITIONS are evaluated as the reference evaluates them — the polynomial in double (qd: the coefficients in
// double; the sample times in double), rounded to `float` (:457-465) — and the cell they fall in is found in double
// (sdf_query_pair), so the fp32 path reads the same cells and decides out-of-map the same way as the fp64 path and
// the reference on the same inputs.  Everything else — velocity, blend, penalty, accumulation — is packed fp32.
template <bool DYN, bool WIDE>
__device__ __forceinline__ void sample_pair_f32(const GtopKernelArgs<float> &a, const IndexBox &box, const double (&qd)[3][6],
                                                const float *cq, double tA, double tB, bool liveA, bool liveB, float wdt,
                                                float dt, f2 (&acc2)[kRedVals]) {
  float pA[3], pB[3];
  {
    const double a2 = tA * tA, a3 = a2 * tA, a4 = a2 * a2, a5 = a4 * tA;
    const double b2 = tB * tB, b3 = b2 * tB, b4 = b2 * b2, b5 = b4 * tB;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      pA[k] = (float)(qd[k][0] + qd[k][1] * tA + qd[k][2] * a2 + qd[k][3] * a3 + qd[k][4] * a4 + qd[k][5] * a5);
      pB[k] = (float)(qd[k][0] + qd[k][1] * tB + qd[k][2] * b2 + qd[k][3] * b3 + qd[k][4] * b4 + qd[k][5] * b5);
    }
  }
  const f2 t = {(float)tA, (float)tB};
  const f2 t2 = t * t, t3 = t2 * t, t4 = t2 * t2, t5 = t4 * t;
  const f2 d2 = splat(2.0f) * t, d3 = splat(3.0f) * t2, d4 = splat(4.0f) * t3, d5 = splat(5.0f) * t4;   // d/dt of the powers
  f2 vel[3], acc3[3];
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const float *q = cq + 6 * k;
    vel[k] = splat(q[1]) + splat(q[2]) * d2 + splat(q[3]) * d3 + splat(q[4]) * d4 + splat(q[5]) * d5;
    if (DYN) acc3[k] = splat(2.0f * q[2]) + splat(6.0f * q[3]) * t + splat(12.0f * q[4]) * t2 + splat(20.0f * q[5]) * t3;
  }
  const f2 v2 = vel[0] * vel[0] + vel[1] * vel[1] + vel[2] * vel[2];
  const f2 vn = (f2){__builtin_amdgcn_sqrtf(v2.x), __builtin_amdgcn_sqrtf(v2.y)} + splat(1e-5f);   // :358
  const f2 ivn = {__builtin_amdgcn_rcpf(vn.x), __builtin_amdgcn_rcpf(vn.y)};
  f2 g3[3];
  f2 dist = sdf_query_pair<WIDE>(a, box, pA, pB, g3[0], g3[1], g3[2]);   // :363
  // isInMap (sdf_map.cpp:55-69) on the float position, against the bounds rounded INTO the box (lo_f / hi_f: for a float
  // p, p < lo <=> p < lo_f — the reference's double comparison, decided exactly); out of the map: dist = -1, grad := 0
  // (sdf_map.cpp:187).  Straight-line selects: with three wavefronts per SIMD they are cheaper than a
  // rarely taken branch (measured: 21.9 against 22.9 us)
  const bool outA = out_of_map(a.lo_f, a.hi_f, pA[0], pA[1], pA[2]);
  const bool outB = out_of_map(a.lo_f, a.hi_f, pB[0], pB[1], pB[2]);
  if (outA) dist.x = -1.0f;
  if (outB) dist.y = -1.0f;
  const f2 arg = (splat(a.d0) - dist) * splat(a.inv_r);
  f2 e = {__expf(arg.x), __expf(arg.y)};          // exp(-(d - d0)/r)
  if (!liveA) e.x = 0.0f;                          // past the loop bound of :353 / idle lane
  if (!liveB) e.y = 0.0f;
  const f2 cd = splat(a.alpha) * e;                // :509
  const f2 gd = splat(-a.alpha_over_r) * e;        // :514
  f2 csum = splat(wdt) * (cd * vn);                // :373
  f2 f1 = splat(wdt * a.res_inv) * (gd * cd * vn);   // 1/res: g3 is per voxel
  const f2 f2_ = splat(wdt) * (cd * ivn);
  if (outA) f1.x = 0.0f;
  if (outB) f1.y = 0.0f;
  f2 w1[3], w2[3], w3[3];
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    w1[k] = f1 * g3[k];
    w2[k] = f2_ * vel[k];
    w3[k] = splat(0.0f);
  }
  if constexpr (DYN) {
    // the commented-out block :383-407 with the formulas of :517-535 (see the fp64 path of gtop_eval_wave_kernel): per
    // axis cv = alpha_v exp((|v| - v0)/r_v), ca likewise; in the gradient cv, ca are the LAST axis's, no sign(v)
    f2 ev[3], ea[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const f2 xv = ((f2){fabsf(vel[k].x), fabsf(vel[k].y)} - splat(a.v0)) * splat(a.inv_r_v);
      const f2 xa = ((f2){fabsf(acc3[k].x), fabsf(acc3[k].y)} - splat(a.a0)) * splat(a.inv_r_a);
      ev[k] = (f2){__expf(xv.x), __expf(xv.y)};
      ea[k] = (f2){__expf(xa.x), __expf(xa.y)};
    }
    const f2 sdt = {liveA ? dt : 0.0f, liveB ? dt : 0.0f};   // every term of the block carries dt; 0 past the loop bound
    csum += (splat(a.alpha_v) * ((ev[0] + ev[1]) + ev[2]) + splat(a.alpha_a) * ((ea[0] + ea[1]) + ea[2])) * vn * sdt;
    const f2 clast = (splat(a.alpha_v) * ev[2] + splat(a.alpha_a) * ea[2]) * ivn;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      w2[k] += (splat(a.gv_scale) * ev[k] * vn + clast * vel[k]) * sdt;
      w3[k] = (splat(a.ga_scale) * ea[k] * vn) * sdt;
    }
  }
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    f2 *ak = acc2 + 6 * k;
    // two fused multiply-adds per entry
    ak[0] += w1[k];
    ak[1] = pk_fma(w1[k], t, ak[1] + w2[k]);
    ak[2] = pk_fma(w1[k], t2, pk_fma(w2[k], d2, ak[2]));
    ak[3] = pk_fma(w1[k], t3, pk_fma(w2[k], d3, ak[3]));
    ak[4] = pk_fma(w1[k], t4, pk_fma(w2[k], d4, ak[4]));
    ak[5] = pk_fma(w1[k], t5, pk_fma(w2[k], d5, ak[5]));
    if (DYN) {
      ak[2] += w3[k] * splat(2.0f);
      ak[3] += w3[k] * splat(6.0f) * t;
      ak[4] += w3[k] * splat(12.0f) * t2;
      ak[5] += w3[k] * splat(20.0f) * t3;
    }
  }
  acc2[18] += csum;
}

// ---------------------------------------------------------------------------
// gtop_eval_wave_kernel — one wavefront = NT whole trajectories, nothing shared
// between wavefronts, the whole evaluation as ONE dependent chain without a
// workgroup barrier.  Built for the latency regime (a batch of about one
// wavefront per SIMD: the bench default, 1 024 x 20 control points), where a
// lone wavefront pays four cycles per instruction of any kind and every round
// trip is exposed, so the design removes phases rather than overlapping them:
//   * no phase 1: every lane of a segment loads the segment's seven inputs per
//     axis itself (the 10 lanes of a segment read the same addresses: one
//     request) and forms the 18 polynomial coefficients in registers — no LDS
//     staging, no barrier, no per-sample LDS reads;
//   * the jerk term rides in the accumulators: its gradient is linear in the
//     coefficients, so lane 0 of each segment STARTS its 18 coefficient-space
//     accumulators at ws*2Qc (and its cost accumulator at ws*c'Qc) instead of
//     zero; the A_s^-T applied after the samples (linear) takes it to
//     derivative space together with the collision gradient;
//   * one LDS round trip: the lanes write their 18 values to a tile, and the
//     lane that owns free variable i sums the 2*LPS entries that make it up
//     (end of segment w-1 + start of segment w) straight from the tile, adds
//     1e-5 and stores; the scalar cost is a DPP wavefront sum of the lanes'
//     cost accumulators, no LDS at all.
// Against the oracle: <= 1e-12 (only the order of the final sums differs from the reference's loop).
// Host-checked: NT*m <= 64/LPS, one workgroup per group of NT trajectories.
// ---------------------------------------------------------------------------
// COLLI = false is the |wc| < 1e-4 case (:346, no collision term), decided by the launcher: the kernel body is
// then one basic block, in which the order the phases are written in can be held (scheduling barriers).
//
// Kernel arguments: what the first loads need (three input pointers, B, m, the time stride) and the field
// descriptor come as leading scalar arguments, which the build preloads into SGPRs at wavefront launch
// (-amdgpu-kernarg-preload-count, csrc/Makefile): measured with s_memtime stamps, a lone wavefront otherwise
// waits ~1 100 cycles for its kernel-argument fetch before it can even request its inputs.  The rest of the
// arguments (`a`; its x/Df/T/sdf/B/m/t_stride/nx/ny/nz fields are not read) arrive while the inputs do.
// The fp64 constants of the closed forms that are neither inline operands (0.5, 1, 2, 4) nor VOP2 literals: as
// literals each costs an s_mov pair in front of its use (43 of them, every one a 4-cycle issue slot of a lone
// wavefront); as kernel arguments they arrive in SGPRs with the rest of the argument block.
template <typename R>
struct GtopWaveConsts {
  R q36 = 36, q72 = 72, q120 = 120, q192 = 192, q360 = 360, q720 = 720;   // jerk Hessian, src/qp_generator.cpp:226-234
  R k10 = 10, k15 = 15, k7 = 7, k6 = 6, k3 = 3, k8 = 8, k1p5 = 1.5, k5 = 5;   // A_s^-1 / A_s^-T / d/dt of the powers
  R eps = 1e-5;                                                           // :358
};

// The fp32 kernels form the per-lane SET-UP — the 18 polynomial coefficients and the jerk term — in double (scalar fp32
// and fp64 cost a SIMD the same four cycles per wave64 instruction; only PACKED fp32 is cheaper, and the set-up is
// not packed): the coefficients are what the sample positions are evaluated from, which the reference does in double,
// and the jerk term's 1/T^5 was where the fp32 path's arithmetic error sat.  Its constants, in double:
template <typename R> struct GtopSetupConsts {};                                  // (fp64 kernels: K itself)
template <> struct GtopSetupConsts<float> : GtopWaveConsts<double> {};
__device__ __forceinline__ const GtopWaveConsts<double> &gtop_setup_consts(const GtopWaveConsts<double> &K, const GtopSetupConsts<double> &) { return K; }
__device__ __forceinline__ const GtopWaveConsts<double> &gtop_setup_consts(const GtopWaveConsts<float> &, const GtopSetupConsts<float> &KD) { return KD; }

// c = A_s^-1 d for the three axes (closed form; rows of A_s: src/qp_generator.cpp:185-195)
template <typename C, typename R>
__device__ __forceinline__ void gtop_form_coefficients(C (&q)[3][6], C T, const GtopWaveConsts<C> &K, const R (&w0)[3][3],
                                                       const R (&w1)[3][3]) {
  const C T2 = T * T;
  const C iT = fast_rcp(T), iT3 = iT * iT * iT, iT4 = iT3 * iT, iT5 = iT4 * iT;
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const C p0 = (C)w0[k][0], v0 = (C)w0[k][1], a0 = (C)w0[k][2];
    const C pT = (C)w1[k][0], vT = (C)w1[k][1], aT = (C)w1[k][2];
    const C P = pT - p0 - v0 * T - (C)0.5 * a0 * T2;
    const C V = (vT - v0 - a0 * T) * T;
    const C A = (aT - a0) * T2;
    q[k][0] = p0; q[k][1] = v0; q[k][2] = (C)0.5 * a0;
    q[k][3] = (K.k10 * P - (C)4 * V + (C)0.5 * A) * iT3;
    q[k][4] = (K.k7 * V - K.k15 * P - A) * iT4;
    q[k][5] = (K.k6 * P - K.k3 * V + (C)0.5 * A) * iT5;
  }
}

// The jerk term of one segment (Jerk Hessian Q_s: src/qp_generator.cpp:226-234, i,j in {3,4,5}): g[k][i-3] = wj 2 (Qc)_i,
// the share of ws*(2Rfp'df + 2Rpp dp) (:330-336) in coefficient space, and cost = wj c'Qc, the share of d'Rd (:326-327)
template <typename C>
__device__ __forceinline__ C gtop_jerk_term(const C (&q)[3][6], C T, C wj, const GtopWaveConsts<C> &K, C (&g)[3][3]) {
  const C T2 = T * T, T3 = T2 * T, T4 = T2 * T2, T5 = T4 * T;
  const C Q33 = K.q36 * T, Q34 = K.q72 * T2, Q35 = K.q120 * T3, Q44 = K.q192 * T3, Q45 = K.q360 * T4, Q55 = K.q720 * T5;
  const C wj2 = wj + wj;
  C jc = (C)0;
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const C c3 = q[k][3], c4 = q[k][4], c5 = q[k][5];
    const C q3 = Q33 * c3 + Q34 * c4 + Q35 * c5;
    const C q4 = Q34 * c3 + Q44 * c4 + Q45 * c5;
    const C q5 = Q35 * c3 + Q45 * c4 + Q55 * c5;
    jc += c3 * q3 + c4 * q4 + c5 * q5;
    g[k][0] = wj2 * q3; g[k][1] = wj2 * q4; g[k][2] = wj2 * q5;
  }
  return wj * jc;
}

// MINW = wavefronts per SIMD the register budget must leave room for: 2 in the latency regime (constants pinned
// in VGPRs, 232 of them), 3 for batches that can fill a third (no pins; 168-VGPR budget).
// MM = GtopMmaState (fp64, NT = 1): the batched CCSA-MMA driver's loop around the evaluation, all st.iters
// evaluations of the trajectory in this one launch — evaluate at st.xcur, update (gtop_mma_update_trajectory: accept /
// reject, asymptotes, stop rules, next trial point), evaluate again; cost and gradient never leave the chip.  One
// wavefront owns the trajectory, so the loop needs no barrier at all.  MM = GtopNoMma: a plain evaluation.
struct GtopNoMma {};
// DYN compiles in the velocity / acceleration penalties the reference has commented out
// (src/grad_traj_optimizer.cpp:383-407, formulas :517-535; they sit inside the collision sample loop, so DYN needs
// COLLI; the launcher picks DYN only for enable_dyn at step 2, as the block's own `step == 2` test would).
// LONG serves trajectories of more than 12 segments with the same wavefront: the segments go through the body 12 at a
// time (five lanes per segment), every chunk leaves its 18 derivative-space rows in ITS columns of an LDS tile of
// 5 m columns, and the free variables are gathered from the whole tile after the last chunk; the cost is a plain
// wavefront sum of the lanes' accumulators.  Up to 227 segments (160 KB of LDS; 118 in the optimizer loop).
// Register budget (wavefronts per SIMD) of a variant: MINW, except that the optimizer loop on six samples per lane
// and the DYN bodies keep the one-sample-at-a-time structure of MINW = 3 on the two-wavefront budget (the update and
// the extra penalty terms need the room).
// (64-bit field indices — fields past 4 GiB — spill the 168-VGPR fp64 bodies at six samples per lane: two-wavefront budget)
// (the same for the fp64 body that walks more than 12 segments in chunks: 19 spilled registers at 168)
template <typename R, bool WIDE, typename MM, int SPL, int MINW, bool DYN, bool LONG>
constexpr int gtop_wave_budget() {
  return ((!std::is_same<MM, GtopNoMma>::value && SPL == 6) || DYN || ((WIDE || LONG) && sizeof(R) == 8 && SPL >= 6) ||
          (sizeof(R) == 4 && SPL >= 6))   // (packed fp32 with the double coefficients of its exact positions: 190 VGPRs)
             ? 2
             : MINW;
}

// NW = 2: ONE trajectory of 7 .. 12 segments over TWO wavefronts at ten lanes per segment (a 128-thread workgroup;
// wavefront w holds segments 6w .. 6w + 5; the tile is shared and one workgroup barrier sits in front of the
// gather) — for batches too small to fill the chip with one wavefront per trajectory, where six samples per lane
// on one wavefront are simply the longer chain (B = 1, 10 segments — the NLopt callback on the reference's own scene).
template <typename R, bool WIDE, int SPL, int NT, bool COLLI, int MINW, typename MM = GtopNoMma, bool DYN = false,
          bool LONG = false, int NW = 1>
__global__ void __launch_bounds__(64 * NW) __attribute__((amdgpu_waves_per_eu(gtop_wave_budget<R, WIDE, MM, SPL, MINW, DYN, LONG>())))
gtop_eval_wave_kernel(const R *__restrict__ arg_x, const R *__restrict__ arg_Df, const R *__restrict__ arg_T,
                      const R *__restrict__ arg_sdf, int arg_B, int arg_m, int arg_t_stride, int arg_nx, int arg_ny,
                      int arg_nz, const GtopKernelArgs<R> arg_rest, const GtopWaveConsts<R> K, const MM st,
                      const GtopSetupConsts<R> KD) {
  constexpr bool MMA = !std::is_same<MM, GtopNoMma>::value;
  static_assert(!MMA || NT == 1 || (SPL == 6 && !LONG), "the optimizer loop: one trajectory per wavefront, or two at five lanes per segment");
  // the optimizer's state, bounds, Df and T are fp64 whatever R is: with R = float only the evaluation runs in fp32
  // (its inputs converted as they are read from LDS, its cost and gradient widened for the update)
  using In = typename std::conditional<MMA, double, R>::type;
  GtopKernelArgs<R> a = arg_rest;
  a.x = arg_x; a.Df = arg_Df; a.T = arg_T; a.sdf = arg_sdf;
  a.B = arg_B; a.m = arg_m; a.t_stride = arg_t_stride;
  a.nx = arg_nx; a.ny = arg_ny; a.nz = arg_nz;
  constexpr int LPS = kSamples / SPL;   // lanes per segment
  constexpr int SPW = 64 / LPS;         // segment slots per wavefront
  constexpr int kStride = NW == 2 ? ((2 * LPS * SPW) | 1) : red_stride(SPL);   // (two wavefronts: 120 busy lanes)
  constexpr int kMVc = (SPL == 6 && NT == 1) ? 128 : 64;   // rows of the optimizer loop's LDS vectors: n <= 45 resp. 99 variables
  static_assert(SPL == 3 || SPL == 6 || SPL == 10 || SPL == 30, "10 or 5 lanes per segment; three or one with as many trajectories per wavefront as fit");
  // MANY (SPL = 30): ONE lane per segment — a lane walks all 30 samples of its segment, so the per-lane set-up
  // (coefficients, jerk term, A^-T) is paid once per segment instead of five or ten times and no sum over a segment's
  // lanes is left — and as many whole trajectories per wavefront as fit: nt = 64 / m (10 of 6 segments, 5 of 12).  The
  // epilogue is a list of nt n + nt tasks dealt over the lanes: a free variable = two tile entries, a trajectory's cost
  // = its m entries of row 18 added in segment order.  For batches that put several such wavefronts on every SIMD.
  // (SPL = 10: the same with three lanes per segment — 21 segment slots: 3 trajectories of up to 7 segments)
  constexpr bool MANY = SPL == 30 || SPL == 10;
  static_assert(!MANY || (NT == 1 && !LONG && !MMA && NW == 1 && MINW >= 3), "one lane per segment: plain evaluation, one sample (pair) at a time");
  static_assert(SPL == 3 || MINW >= 3, "six samples per lane: one (pair) at a time only");
  static_assert(NT == 1 || NT == 2, "one or two trajectories per wavefront");
  static_assert(!LONG || (SPL == 6 && NT == 1), "more than 12 segments: five lanes per segment, one trajectory");
  static_assert(!DYN || (COLLI && MINW >= 3), "the velocity/acceleration block lives in the sample loop, one sample at a time");
  static_assert(NW == 1 || (NW == 2 && SPL == 3 && NT == 1 && !LONG && !MMA), "two wavefronts per trajectory: plain evaluation at ten lanes per segment");
  extern __shared__ __align__(16) unsigned char smem_raw[];
  R *tile = reinterpret_cast<R *>(smem_raw);   // [19][kStride] (+ [kRounds*64] gradient for the optimizer update)
  GTOP_STAMP(0);
  GTOP_STAMP_HWID();
  const int lane = NW == 2 ? (int)(threadIdx.x & 63u) : (int)threadIdx.x;
  const int wave = NW == 2 ? (int)(threadIdx.x >> 6) : 0;   // which half of the trajectory's segments
  const int m = a.m, ndp = 3 * m - 3, n = 3 * ndp;
  if constexpr (LONG) __builtin_assume(m > SPW);
  else __builtin_assume(m >= 2 && NT * m <= NW * SPW);
  __builtin_assume(lane >= 0 && lane < 64);
  const int tstride = LONG ? ((LPS * m) | 1) : kStride;    // row stride of the tile (LONG: 5 m columns, made odd)
  const int kMV = LONG ? ((n + 63) & ~63) : kMVc;          // rows of the optimizer loop's LDS vectors
  const int nchunks = LONG ? (m + SPW - 1) / SPW : 1;
  constexpr int kTileRows = LONG ? 18 : kRedVals;          // (LONG sums the cost in registers)
  GTOP_STAMP(1);

  // XCD-aware order (workgroup id mod 8 = XCD): XCD x gets the x-th contiguous eighth of the batch
  const int nt = MANY ? SPW / m : NT;   // trajectories per wavefront
  const int ngroups = (a.B + nt - 1) / nt;
  const int per_xcd = (ngroups + 7) >> 3;
  // The grid is 8*per_xcd workgroups; the up to 7 beyond the batch take no early exit (a branch here would
  // split the kernel-argument loads into two dependent round trips): they shadow the last group with every
  // lane idle and every store predicated off.
  const int grp_raw = ((int)blockIdx.x & 7) * per_xcd + ((int)blockIdx.x >> 3);
  const bool grp_ok = grp_raw < ngroups;
  const int grp = grp_ok ? grp_raw : ngroups - 1;
  const int b0 = grp * nt;

  const int slot_w = lane / LPS, li = lane - slot_w * LPS;   // segment slot within the wavefront, lane within the segment
  const int slot = wave * SPW + slot_w;
  int tl = 0, s = slot;
  if constexpr (MANY) {
    // slot / m by a 16-bit reciprocal (exact: slot < 64, and slot/m is never within 2^-10 below an integer)
    tl = (int)(((unsigned)slot * (unsigned)((65536 + m - 1) / m)) >> 16);
    s -= tl * m;
  } else if constexpr (NT == 2) {
    tl = s >= m;
    s -= tl * m;
  }
  bool seg_ok = grp_ok & (slot_w < SPW) & (slot < nt * m) & (b0 + tl < a.B);   // (LONG: set per chunk, below)
  if (!seg_ok) { tl = 0; s = 0; }   // idle lanes shadow the first segment: finite data, results never read

  unsigned long long t_launch = 0ull;
  int npass = 1;
  if constexpr (MMA) {
    t_launch = wall_clock64();
    npass = st.iters;
  }
  // The optimizer loop keeps its trajectory's state on the chip between passes: the eight n-vectors in LDS (behind
  // the tile and the gradient rows; n <= 45 < 64, lane j owns entry j) and the scalars in registers; global memory
  // sees it once, at the end.  With the state in global memory every pass paid five dependent round trips for it — 7 us per pass,
  // of which the evaluation is 2.5.
  // NT = 2 (up to 6 segments each, large batches): the wavefront keeps BOTH its trajectories' states — two blocks of
  // kState doubles — and runs the update twice per pass, once per trajectory, every lane serving the trajectory in turn;
  // a trajectory that has stopped is still evaluated (its lanes cannot leave) but no longer updated.
  [[maybe_unused]] double *mv = nullptr;   // per trajectory [8][kMV]: x, xcur, xprev, xprevprev, dfdx, sigma, lb, ub; then Df, T
  [[maybe_unused]] const int kState = LONG ? 0 : 8 * kMVc + 32;   // doubles per trajectory (LONG: one trajectory)
  [[maybe_unused]] GtopMmaVecs mvecs[NT] = {};
  [[maybe_unused]] GtopMmaScalars msc[NT] = {};
  [[maybe_unused]] bool mma_live[NT] = {};
  if constexpr (MMA) {
    mv = reinterpret_cast<double *>(tile) + kTileRows * tstride + (LONG ? kMV : 128);
    // (the evaluation reads its inputs from here too — the trial point, and Df and T staged once behind the
    // vectors — so a pass has no global load but the distance-field corners, and no global store at all)
#pragma unroll
    for (int t = 0; t < NT; ++t) {
    double *mvt = mv + t * kState;
    const int bt = b0 + t;
    mvecs[t] = GtopMmaVecs{mvt, mvt + kMV, mvt + 2 * kMV, mvt + 3 * kMV, mvt + 4 * kMV, mvt + 5 * kMV, mvt + 6 * kMV,
                           mvt + 7 * kMV, mvt + kMV};
    mma_live[t] = grp_ok & (bt < a.B);
    if (mma_live[t]) {
      const size_t o = (size_t)bt * n;
      const double *Df64 = reinterpret_cast<const double *>(a.Df), *T64 = reinterpret_cast<const double *>(a.T);
      if (lane < 18) mvt[8 * kMV + lane] = Df64[(size_t)bt * 18 + lane];
      if constexpr (LONG) {
        for (int j = lane; j < m; j += 64) mvt[8 * kMV + 18 + j] = T64[(size_t)bt * a.t_stride + j];
      } else {
        if (lane < m) mvt[8 * kMV + 18 + lane] = T64[(size_t)bt * a.t_stride + lane];
      }
      if (st.x0_init) {   // (uniform) a fresh problem: mma_init_kernel's arithmetic, straight into LDS
        msc[t] = GtopMmaScalars{1.0, 0.0, 0.0, 0.0, 0.0, 0, 0, 0};
        for (int j = lane; j < n; j += 64) {
          const double lo = st.lb[o + j], hi = st.ub[o + j];
          double v = st.x0_init[o + j];
          v = v < lo ? lo : (v > hi ? hi : v);   // nlopt clamps the start into the box
          mvt[j] = v; mvt[kMV + j] = v; mvt[2 * kMV + j] = v; mvt[3 * kMV + j] = v;
          mvt[4 * kMV + j] = 0.0;
          mvt[5 * kMV + j] = (isinf(lo) || isinf(hi)) ? 1.0 : 0.5 * (hi - lo);
          mvt[6 * kMV + j] = lo;
          mvt[7 * kMV + j] = hi;
        }
      } else {
      msc[t] = gtop_mma_load_scalars(st, bt);
      for (int j = lane; j < n; j += 64) {
        mvt[j] = st.x[o + j];
        mvt[kMV + j] = st.xcur[o + j];
        mvt[2 * kMV + j] = st.xprev[o + j];
        mvt[3 * kMV + j] = st.xprevprev[o + j];
        mvt[4 * kMV + j] = st.dfdx[o + j];
        mvt[5 * kMV + j] = st.sigma[o + j];
        mvt[6 * kMV + j] = st.lb[o + j];
        mvt[7 * kMV + j] = st.ub[o + j];
      }
      }
    }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");   // staged by some lanes, read by others of this wavefront
    __builtin_amdgcn_wave_barrier();
  }
  for (int pass = 0; pass < npass; ++pass) {
  const R *xsrc = a.x;
  if constexpr (MMA) {
    // stop rules (mma.hpp:35-39; set_maxtime, :144-148), wave-uniform: a trajectory that has stopped (ftol / xtol in the
    // update) leaves the loop; past the wall-clock limit a running one stops where it is, after at least one evaluation
    bool running = false;
#pragma unroll
    for (int t = 0; t < NT; ++t) running |= mma_live[t] && msc[t].state < 3;
    if (!running) break;
    if (st.max_ticks > 0 && pass > 0 && (long long)(wall_clock64() - t_launch) > st.max_ticks) {
#pragma unroll
      for (int t = 0; t < NT; ++t)
        if (mma_live[t] && msc[t].state < 3) msc[t].state = GTOP_MMA_MAXTIME_REACHED;
      break;
    }
  }
  [[maybe_unused]] R cost_run = (R)0;   // LONG: this lane's share of the cost over its chunks
  constexpr int kRounds = (LONG || MANY) ? 1 : (NT * 9 * (NW * SPW / NT - 1) + 64 * NW - 1) / (64 * NW);
  const int tid = NW == 2 ? (int)threadIdx.x : lane;
  int offA[kRounds], offB[kRounds];     // (filled below, while the inputs are on their way)
  bool okq[kRounds];
  bool cost_lane = false;
  for (int ch = 0; ch < nchunks; ++ch) {
  if constexpr (LONG) {
    s = ch * SPW + slot_w;
    seg_ok = grp_ok & (slot_w < SPW) & (s < m);
    if (!seg_ok) s = 0;
  }
  // ---- inputs: two waypoints' (p, v, a) per axis and T_s, straight from HBM/L2 (the optimizer loop: from LDS) ----
  // derivative vector layout (src/qp_generator.cpp:363-387): start | end | waypoint 1 | ... | waypoint m-1
  const In *xb, *dfb, *Tb;
  if constexpr (MMA) {
    const double *mvl = mv + tl * kState;   // this lane's trajectory
    xb = mvl + kMV;
    dfb = mvl + 8 * kMV;
    Tb = mvl + 8 * kMV + 18;
  } else {
    xb = xsrc + (size_t)b0 * n + tl * n;      // this lane's trajectory (b0: wave-uniform)
    dfb = a.Df + (size_t)b0 * 18 + tl * 18;
    Tb = a.T + (size_t)b0 * a.t_stride + tl * a.t_stride;
  }
  const R T = (R)Tb[s];
  // axis 0: the (p, v, a) triple at the segment's start and at its end; the other axes are one per-lane stride
  // further (6 within Df, 3m-3 within x: :182-187)
  const bool first = s == 0, last = s + 1 == m;
  const In *p0 = first ? dfb : xb + 3 * (s - 1);
  const In *p1 = last ? dfb + 3 : xb + 3 * s;
  const int st0 = first ? 6 : ndp, st1 = last ? 6 : ndp;
  R w0[3][3], w1[3][3];
#pragma unroll
  for (int k = 0; k < 3; ++k) {
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      w0[k][i] = (R)p0[k * st0 + i];
      w1[k][i] = (R)p1[k * st1 + i];
    }
  }

  // Where this lane's free variable(s) will be summed from at the very end (index arithmetic done here, while
  // the inputs are on their way): free variable = end of segment wpt-1 (entry 2 der + 1) + start of segment
  // wpt (entry 2 der)  (:425-432); tile[v][lane] holds entry v of lane's segment.
#pragma unroll
  for (int r = 0; r < ((LONG || MANY) ? 0 : kRounds); ++r) {
    const int qi = tid + 64 * NW * r;
    int tq = 0, i = qi;
    if constexpr (NT == 2) {
      tq = i >= n;
      i -= tq * n;
    }
    okq[r] = grp_ok & (qi < NT * n) & (b0 + tq < a.B);
    const int axis = (i >= ndp) + (i >= 2 * ndp), c = i - axis * ndp;
    const int wpt = c / 3 + 1, der = c - 3 * (wpt - 1);   // interior waypoint 1..m-1
    const int rowB = axis * 6 + 2 * der;
    const int sA = tq * m + wpt - 1;
    offA[r] = okq[r] ? (rowB + 1) * kStride + sA * LPS : 0;
    offB[r] = okq[r] ? rowB * kStride + (sA + 1) * LPS : 0;
  }
  // The scalar cost takes the same road: row 18 of the tile holds the lanes' cost accumulators, and lanes 48 ..
  // 48 + NT*m - 1 — idle in the LAST round of the sum above (at most 45, 35 or 26 of its lanes carry free variables) —
  // each sum one segment's entries with the very instructions the free variables use; DPP row shifts then add the
  // segment sums of a trajectory.  (A 64-lane DPP sum of the accumulators was 46 instructions of a lone wavefront's
  // issue time.)
  // NT = 2: trajectory t's segments sit in lanes 48 + 8t .. 48 + 8t + m - 1, so that both trajectories' sums associate
  // the same way (a trajectory's result must not depend on its place in the pair).
  const int cs = lane - 48;
  const int ct = NT == 2 ? cs >> 3 : 0, csi = NT == 2 ? cs & 7 : cs;   // trajectory, segment
  cost_lane = !LONG & !MANY & (wave == NW - 1) & (cs >= 0) & (csi < m);   // (two wavefronts: the second one's lanes 48 ..)
  if (cost_lane) offA[kRounds - 1] = 18 * kStride + (ct * m + csi) * LPS;
  const R ws = a.ws;   // the launcher has applied :412-415 (step 1 -> ws = 0): `step` is not read here
  const R wc = a.wc;
  ExpConsts expk;
  R pen_d0 = a.d0, pen_inv_r = a.inv_r, pen_alpha = a.alpha, pen_gd = -a.alpha_over_r;   // (:507-515)
  MapBox<R> mapbox = {{a.lo[0], a.lo[1], a.lo[2]}, {a.hi[0], a.hi[1], a.hi[2]}};
  // the cell lookup's constants, in double (fp32 kernels: the grid's own doubles, not their fp32 roundings)
  IndexBox ibox;
  if constexpr (kIsF32<R>) ibox = IndexBox{{a.idx_origin[0], a.idx_origin[1], a.idx_origin[2]}, a.idx_half, a.idx_rinv};
  else ibox = IndexBox{{(double)a.origin[0], (double)a.origin[1], (double)a.origin[2]}, 0.5 * (double)a.res, (double)a.res_inv};
  if constexpr (COLLI && !kIsF32<R> && MINW <= 2 && !MMA) {   // (the optimizer loop has no registers to spare)
    // the exp constants only: with the map box pinned as well (12 more VGPRs) the body spills two registers since
    // the hand-issued loads hold all 12 corner pairs at once — 4.45 against 4.23 us
    expk.pin();
  }
#if defined(GTOP_STAMPS) && GTOP_STAMPS != 2   // (the first / last stamp build adds no waits)
  GTOP_STAMP(2);   // inputs requested
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  GTOP_STAMP(3);   // inputs landed
#endif
  // ---- coefficients c = A_s^-1 d (closed form; rows of A_s: src/qp_generator.cpp:185-195) ----
  const R T2 = T * T;   // (for the A_s^-T at the end; the coefficients and the jerk term form their own powers)
  const R iT = fast_rcp(T), iT3 = iT * iT * iT, iT4 = iT3 * iT, iT5 = iT4 * iT;
  // :351, dt = T/30.  The quotient proper (a dozen instructions) is only needed where the sample COUNT hangs on
  // the accumulated sample time (tiny T, below); everywhere else T * (1/30) is the same to an ulp.
  const R dt = T * (R)(1.0 / 30.0);
  const R wdt = wc * dt;
  R q[3][6];
  [[maybe_unused]] double qd[kIsF32<R> ? 3 : 1][6];   // fp32 kernels: the coefficients in double (positions; the jerk term)
  const GtopWaveConsts<double> &KC = gtop_setup_consts(K, KD);
  if constexpr (kIsF32<R>) {
    gtop_form_coefficients(qd, (double)T, KC, w0, w1);
#pragma unroll
    for (int k = 0; k < 3; ++k)
#pragma unroll
      for (int c = 0; c < 6; ++c) q[k][c] = (R)qd[k][c];
  } else {
    gtop_form_coefficients(q, T, K, w0, w1);
  }
  // The jerk term, as the START value of the accumulators (a lambda: it is placed where the distance-field
  // loads are in flight, see below).  Jerk Hessian Q_s: src/qp_generator.cpp:226-234, i,j in {3,4,5}.
  R acc[kRedVals];
  auto jerk_init = [&]() {
    // lane 0 of a segment carries the segment's jerk term into the sums
    const R wj = (seg_ok & (li == 0)) ? ws : (R)0;
    R g[3][3], jcost;
    if constexpr (kIsF32<R>) {   // in double, from the double coefficients (see GtopSetupConsts)
      double gdd[3][3];
      jcost = (R)gtop_jerk_term(qd, (double)T, (double)wj, KC, gdd);
#pragma unroll
      for (int k = 0; k < 3; ++k)
#pragma unroll
        for (int i = 0; i < 3; ++i) g[k][i] = (R)gdd[k][i];
    } else {
      jcost = gtop_jerk_term(q, T, wj, K, g);
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      // (entries 0..2 of the coefficient-space gradient are zero)
      acc[6 * k + 0] = (R)0; acc[6 * k + 1] = (R)0; acc[6 * k + 2] = (R)0;
      acc[6 * k + 3] = g[k][0]; acc[6 * k + 4] = g[k][1]; acc[6 * k + 5] = g[k][2];
    }
    acc[18] = jcost;
  };

  // ---- collision samples (:345-409); sample index = li + j*LPS ----
  if constexpr (COLLI) {
    // Sample times (:353): t_i = 1e-3 + i*dt; segments with T < 0.0301 (where the
    // sample COUNT depends on the accumulated value) replay the reference's addition chain.
    // Every term of a sample carries the factor alpha * wc * dt (cd of :509 times the weights of :373/:417);
    // a sample past the loop bound of :353 contributes nothing, i.e. has that factor zero.  With T >= 0.0301
    // all 30 samples are inside the bound, so only the replay path ever has to clear it.
    // Sample TIMES are formed in double whatever R is (the reference's are doubles, and the fp32 kernels evaluate the
    // positions from them in double: see sample_pair_f32); R = double: TT is R, nothing changes.
    using TT = double;
    const TT Tt = (TT)T;
    TT dtt;
    if constexpr (kIsF32<R>) dtt = Tt * (1.0 / 30.0);
    else dtt = (TT)dt;
    const bool tiny_T = T < (R)0.0301;
    const bool any_tiny = __ballot(tiny_T) != 0ull;   // wave-uniform, rare
    // The quotient proper (:351) is only needed on that replay path; the empty asm keeps the dozen instructions of
    // the division inside the rare branch (the compiler otherwise computes it up front for everybody).
    auto tiny_dt = [&]() {
      TT Tq = Tt;
      asm volatile("" : "+v"(Tq));
      return Tq / (TT)30.0;
    };
    const R aw_all = pen_alpha * wdt;
    // (sdt: the sample's own dt, zero past the loop bound — the factor every DYN term carries; unused otherwise)
    auto sample_time = [&](int j, TT &t, R &awj, R &sdt) {
      t = (TT)(li + j * LPS) * dtt + (TT)1e-3;
      awj = aw_all;
      sdt = dt;
      if (any_tiny) {
        if (tiny_T) {
          const TT dtq = tiny_dt();
          t = (TT)1e-3;
          for (int i = 0; i < li + j * LPS; ++i) t += dtq;
          awj = (t < Tt) ? pen_alpha * (wc * (R)dtq) : (R)0;
          sdt = (t < Tt) ? (R)dtq : (R)0;
          // a sample past the loop bound (:353) is not evaluated by the reference; here it is, with weight 0 — at the
          // first sample's time rather than on the extrapolated polynomial, where an exp could overflow into 0 * inf
          t = (t < Tt) ? t : (TT)1e-3;
        }
      }
    };
    constexpr int NTS = (MINW <= 2) ? SPL : 1;   // latency regime: all sample times before the first load
    TT ts[NTS];
    R aw[NTS];
    [[maybe_unused]] R sdts[NTS];
    if constexpr (MINW <= 2) {
#pragma unroll
      for (int j = 0; j < SPL; ++j) {
        ts[j] = (TT)(li + j * LPS) * dtt + (TT)1e-3;
        aw[j] = aw_all;
        sdts[j] = dt;
      }
      if (any_tiny) {   // ONE wave-uniform branch for all of the lane's samples: the addition chain runs on from one to the next
        if (tiny_T) {
          const TT dtq = tiny_dt();
          TT t = (TT)1e-3;
          int i = 0;
#pragma unroll
          for (int j = 0; j < SPL; ++j) {
            for (; i < li + j * LPS; ++i) t += dtq;
            ts[j] = (t < Tt) ? t : (TT)1e-3;   // (past the loop bound: weight 0, evaluated at the first sample's time; see sample_time)
            aw[j] = (t < Tt) ? pen_alpha * (wc * (R)dtq) : (R)0;
            sdts[j] = (t < Tt) ? (R)dtq : (R)0;
          }
        }
      }
    }
    // The samples of a lane go through two stages, CH at a time.  Latency regime (MINW = 2): CH = SPL, all 12 corner
    // loads of the lane in flight at once, the jerk term and the speeds computed behind them, the order pinned by
    // scheduling barriers.  Throughput regime (MINW = 3): one sample at a time — other wavefronts cover the loads,
    // and only one sample's corners are live (the 168-VGPR budget of a third wavefront).
    constexpr int CH = (MINW <= 2) ? SPL : 1;
    constexpr int kUnrollJ = (SPL <= 3 && MINW <= 2) ? SPL : 1;   // one sample at a time (MINW = 3): a loop (code size; unrolled, the three-sample body spills at 168 VGPRs)
#ifndef GTOP_ASM_LOADS
#define GTOP_ASM_LOADS 1
#endif
    // hand-issued corner loads (sdf_issue_asm): the lone-wavefront fp64 body with 32-bit field offsets
    constexpr bool ASMLD = GTOP_ASM_LOADS && !kIsF32<R> && !WIDE && MINW <= 2 && SPL == 3 && !MMA;   // (MMA: no registers left)
    if constexpr (kIsF32<R> && SPL % 2 == 0) {
      // packed fp32 (see sample_pair_f32): samples jj and jj+1 of this lane together in float2 registers
      jerk_init();
      float cq[18];
#pragma unroll
      for (int k = 0; k < 3; ++k)
#pragma unroll
        for (int c = 0; c < 6; ++c) cq[6 * k + c] = (float)q[k][c];
      f2 acc2[kRedVals];
#pragma unroll
      for (int v = 0; v < kRedVals; ++v) acc2[v] = (f2){(float)acc[v], 0.0f};
#pragma unroll 1
      for (int jj = 0; jj < SPL; jj += 2) {
        TT tA, tB;
        R awA, awB, sdA, sdB;
        sample_time(jj, tA, awA, sdA);
        sample_time(jj + 1, tB, awB, sdB);
        // (live = inside the loop bound of :353; without DYN every term carries alpha, so alpha*wc*dt != 0 says the same)
        sample_pair_f32<DYN, WIDE>(reinterpret_cast<const GtopKernelArgs<float> &>(a), ibox, qd, cq, tA, tB,
                                   DYN ? sdA != (R)0 : awA != (R)0, DYN ? sdB != (R)0 : awB != (R)0, (float)wdt,
                                   (float)dt, acc2);
      }
#pragma unroll
      for (int v = 0; v < kRedVals; ++v) acc[v] = (R)(acc2[v].x + acc2[v].y);
    } else {
    if constexpr (MINW > 2) jerk_init();
#pragma unroll kUnrollJ
    for (int j0 = 0; j0 < SPL; j0 += CH) {
      // stage A: positions, index arithmetic, corner loads
      R vels[CH][3];
      [[maybe_unused]] R accs[CH][3];
      SdfTap<R> taps[CH];
      gtop_d2 raw[ASMLD ? CH : 1][4];
      if constexpr (MINW > 2) sample_time(j0, ts[0], aw[0], sdts[0]);
      // The position of a sample (:457-465, sums in the reference's order) goes through `float` (the reference's
      // local), so isInMap's double comparisons (sdf_map.cpp:55-69) are decided exactly by float comparisons against
      // the bounds rounded INTO the box (a.lo_f = the smallest float >= lo, a.hi_f = the largest <= hi: for a float p,
      // p < lo <=> p < lo_f).  The common path only asks whether ANY of the lane's CH samples is outside — min / max
      // over the samples per axis, six compares — and a wave-uniform, rarely taken branch in stage B does the rest.
      // (Only the latency variant: with other wavefronts on the SIMD the straight-line selects are cheaper than the
      // branch — measured, B = 16 384 fp64: 34.4 us with the selects, 37.6 with the branch.)
      constexpr bool kRareOut = CH == SPL;
      // one sample at a time (the throughput bodies): the sample's four corner loads first, everything that does not
      // need them — velocity, speed, its reciprocal — while they are in flight
#ifndef GTOP_LOADS_FIRST
#define GTOP_LOADS_FIRST 1
#endif
      constexpr bool kLoadsFirst = GTOP_LOADS_FIRST && CH == 1 && !LONG;   // (the chunked body, on the two-wavefront budget: 3 % slower with it, measured)
      [[maybe_unused]] float pmin[3], pmax[3];
      [[maybe_unused]] bool outs[CH];
      // (the position polynomial in double — fp32 kernels: from the double coefficients and the double sample time)
      auto position = [&](int k, TT t, TT t2, TT t3, TT t4, TT t5) {
        if constexpr (kIsF32<R>)
          return (float)(qd[k][0] + qd[k][1] * t + qd[k][2] * t2 + qd[k][3] * t3 + qd[k][4] * t4 + qd[k][5] * t5);
        else
          return (float)(q[k][0] + q[k][1] * t + q[k][2] * t2 + q[k][3] * t3 + q[k][4] * t4 + q[k][5] * t5);
      };
#pragma unroll
      for (int c = 0; c < CH; ++c) {
        const TT tt = ts[MINW <= 2 ? j0 + c : 0];
        const TT u2 = tt * tt, u3 = u2 * tt, u4 = u2 * u2, u5 = u4 * tt;
        const R t = (R)tt;
        const R t2 = t * t, t3 = t2 * t, t4 = t2 * t2;   // (R = double: the same values as u2 .. u4)
        const R d2 = (R)2 * t, d3 = K.k3 * t2, d4 = (R)4 * t3, d5 = K.k5 * t4;   // d/dt of the powers
        double pos[3];
        [[maybe_unused]] float posf[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
          // :457-465 / :477-485 (sums in the reference's order), then the float round trip
          const float pf = position(k, tt, u2, u3, u4, u5);
          pos[k] = (double)pf;
          posf[k] = pf;
          if constexpr (kRareOut) {
            pmin[k] = c == 0 ? pf : fminf(pmin[k], pf);
            pmax[k] = c == 0 ? pf : fmaxf(pmax[k], pf);
          }
          if constexpr (!kLoadsFirst) {
            vels[c][k] = round_through_float(q[k][1] + q[k][2] * d2 + q[k][3] * d3 + q[k][4] * d4 + q[k][5] * d5);
            if constexpr (DYN)   // getAccelerationFromCoeff, :491-505 (through `float` like the other two)
              accs[c][k] = round_through_float((R)2 * q[k][2] + K.k6 * q[k][3] * t + (R)12 * q[k][4] * t2 + (R)20 * q[k][5] * t3);
          }
        }
        if constexpr (ASMLD) taps[c] = sdf_issue_asm(a, ibox, pos[0], pos[1], pos[2], raw[c]);
        else taps[c] = sdf_issue<R, WIDE>(a, ibox, pos[0], pos[1], pos[2]);   // :363
        if constexpr (!kRareOut) {
          if constexpr (kIsF32<R>) outs[c] = out_of_map(a.lo_f, a.hi_f, posf[0], posf[1], posf[2]);   // (exact: see above)
          else outs[c] = out_of_map(mapbox.lo, mapbox.hi, (R)pos[0], (R)pos[1], (R)pos[2]);
        }
      }
      bool any_out = false;
      if constexpr (kRareOut) {
        const bool lane_out = (pmin[0] < a.lo_f[0]) | (pmin[1] < a.lo_f[1]) | (pmin[2] < a.lo_f[2]) |
                              (pmax[0] > a.hi_f[0]) | (pmax[1] > a.hi_f[1]) | (pmax[2] > a.hi_f[2]);
        any_out = __ballot(lane_out) != 0ull;   // wave-uniform, rare
      }
      if (j0 == 0) GTOP_STAMP(4);   // corner loads issued
      if constexpr (CH == SPL || kLoadsFirst) GTOP_PHASE_FENCE();   // every corner load is issued above this line ...
      if constexpr (kLoadsFirst) {
        // ... one sample at a time: the velocity (and the speeds below) behind the loads.  Without the fence the
        // compiler, short of registers, loaded one record, waited, and only then loaded the other into the same
        // registers: two exposed round trips per sample.
        const R t = (R)ts[0];
        const R t2 = t * t, t3 = t2 * t, t4 = t2 * t2;
        const R d2 = (R)2 * t, d3 = K.k3 * t2, d4 = (R)4 * t3, d5 = K.k5 * t4;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
          vels[0][k] = round_through_float(q[k][1] + q[k][2] * d2 + q[k][3] * d3 + q[k][4] * d4 + q[k][5] * d5);
          if constexpr (DYN)
            accs[0][k] = round_through_float((R)2 * q[k][2] + K.k6 * q[k][3] * t + (R)12 * q[k][4] * t2 + (R)20 * q[k][5] * t3);
        }
      }
      // ... and what does not need them runs while they are in flight: the jerk term and the speeds
      if constexpr (MINW <= 2) jerk_init();
      R vns[CH], ivns[CH];
#pragma unroll
      for (int c = 0; c < CH; ++c) {
        const R *vel = vels[c];
        vns[c] = speed_sqrt(vel[0] * vel[0] + vel[1] * vel[1] + vel[2] * vel[2]) + K.eps;   // :358
        ivns[c] = quick_rcp(vns[c]);
      }
#if defined(GTOP_STAMPS) && GTOP_STAMPS != 2
      if (j0 == 0) {
        asm volatile("" ::"v"(vns[0]), "v"(ivns[CH - 1]), "v"(acc[18]), "v"(acc[3]));
        GTOP_STAMP(5);   // in-flight work done
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        GTOP_STAMP(6);   // corner loads landed
      }
#endif
      if constexpr (CH == SPL) GTOP_PHASE_FENCE();
      // stage B: trilinear blend, penalty, accumulation
#pragma unroll
      for (int c = 0; c < CH; ++c) {
        if constexpr (ASMLD) {
          // sample c's four loads have landed once at most 4*(CH-1-c) of the later ones are in flight; the jerk
          // term, the speeds and the previous sample's accumulation come first
          // (acc[3], acc[17]: operands of explicit fmas — an extra use of a bare product such as acc[18] = wj*jc would
          // change whether the compiler contracts it with the next addition, and with it the last bit against the
          // variants of this kernel that load the ordinary way)
          if (CH - 1 - c == 2) gtop_wait_pairs<8>(raw[c], acc[3], acc[17]);
          else if (CH - 1 - c == 1) gtop_wait_pairs<4>(raw[c], acc[3], acc[17]);
          else gtop_wait_pairs<0>(raw[c], acc[3], acc[17]);
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            taps[c].v[2 * q] = raw[c][q].x;
            taps[c].v[2 * q + 1] = raw[c][q].y;
          }
        }
        const TT tt = ts[MINW <= 2 ? j0 + c : 0];
        const R t = (R)tt;
        const R *vel = vels[c];
        const R t2 = t * t, t3 = t2 * t, t4 = t2 * t2, t5 = t4 * t;
        const R vn = vns[c], ivn = ivns[c];
        R g3[3];
        R dist = sdf_blend(taps[c], g3[0], g3[1], g3[2]);   // g3 per voxel, not per metre
        if constexpr (!kRareOut) {   // dist = -1 (sdf_map.cpp:187); grad := 0 below, through its weight f1
          dist = outs[c] ? (R)-1 : dist;
        } else if (any_out) {   // (rare) which of the samples it is: the reference's own test on the sample's position
          const TT u2 = tt * tt, u3 = u2 * tt, u4 = u2 * u2, u5 = u4 * tt;
          const float pfx = position(0, tt, u2, u3, u4, u5), pfy = position(1, tt, u2, u3, u4, u5),
                      pfz = position(2, tt, u2, u3, u4, u5);
          bool is_out;
          if constexpr (kIsF32<R>) is_out = out_of_map(a.lo_f, a.hi_f, pfx, pfy, pfz);   // (exact: see stage A)
          else is_out = out_of_map(mapbox.lo, mapbox.hi, (R)pfx, (R)pfy, (R)pfz);
          if (is_out) {   // dist = -1, grad := 0 (sdf_map.cpp:187)
            dist = (R)-1;
            g3[0] = g3[1] = g3[2] = (R)0;
          }
        }
        const R e = penalty_exp((pen_d0 - dist) * pen_inv_r, expk);   // exp(-(d - d0)/r)
        const R cdw = aw[MINW <= 2 ? j0 + c : 0] * e;   // wc*dt * cd, cd of :509 (idle lanes: shadow data, never read)
        const R cv = cdw * vn;
        acc[18] = gfma(cdw, vn, acc[18]);   // += cv: :373, weighted as in :417-418 (fusions are spelled out: -ffp-contract=on)
        // g_colli.row(k) += (gd*grad(k)*cd*vn * T*Ldp + cd*(vel(k)/vn) * T*V*Ldp) * dt   (:376-381); gd of :514
        R f1 = ((pen_gd * a.res_inv) * e) * cv;   // (out of the map, rare-branch form: g3 = 0)
        if constexpr (!kRareOut) f1 = outs[c] ? (R)0 : f1;   // grad := 0
        const R f2 = cdw * ivn;
        const R d2 = (R)2 * t, d3 = K.k3 * t2, d4 = (R)4 * t3, d5 = K.k5 * t4;
        [[maybe_unused]] R dw2[3], dw3[3];
        if constexpr (DYN) {
          // The block commented out at :383-407 with the formulas of :517-535: per axis cv = alpha_v exp((|v| - v0)/r_v),
          // ca likewise on the acceleration; cost += (cv + ca) |v| dt per axis (wv = wa = 1, :412); in the gradient the
          // velocity row gets gv |v| + (cv + ca) v/|v| and the acceleration row ga |v|, where cv, ca are the values
          // the cost loop LEFT BEHIND — the last axis's — and there is no sign(v) factor: both as written.
          const R sdt = sdts[MINW <= 2 ? j0 + c : 0];   // this sample's dt; 0 past the loop bound of :353
          const R *acc3 = accs[c];
          R ev[3], ea[3];
#pragma unroll
          for (int k = 0; k < 3; ++k) {
            ev[k] = penalty_exp((gabs(vel[k]) - a.v0) * a.inv_r_v, expk);
            ea[k] = penalty_exp((gabs(acc3[k]) - a.a0) * a.inv_r_a, expk);
          }
          const R csum_dyn = a.alpha_v * ((ev[0] + ev[1]) + ev[2]) + a.alpha_a * ((ea[0] + ea[1]) + ea[2]);
          acc[18] = gfma(csum_dyn * vn, sdt, acc[18]);
          const R clast = (a.alpha_v * ev[2] + a.alpha_a * ea[2]) * ivn;
#pragma unroll
          for (int k = 0; k < 3; ++k) {
            dw2[k] = (a.gv_scale * ev[k] * vn + clast * vel[k]) * sdt;
            dw3[k] = (a.ga_scale * ea[k] * vn) * sdt;          // on T*V*V = [0, 0, 2, 6t, 12t^2, 20t^3]
          }
        }
#pragma unroll
        for (int k = 0; k < 3; ++k) {
          const R w1k = f1 * g3[k], w2k = f2 * vel[k];
          R *ak = acc + 6 * k;
          ak[0] = gfma(f1, g3[k], ak[0]);
          ak[1] = gfma(w1k, t, gfma(f2, vel[k], ak[1]));
          ak[2] = gfma(w1k, t2, gfma(w2k, d2, ak[2]));
          ak[3] = gfma(w1k, t3, gfma(w2k, d3, ak[3]));
          ak[4] = gfma(w1k, t4, gfma(w2k, d4, ak[4]));
          ak[5] = gfma(w1k, t5, gfma(w2k, d5, ak[5]));
          if constexpr (DYN) {
            ak[1] += dw2[k];
            ak[2] = gfma(dw2[k], d2, gfma(dw3[k], (R)2, ak[2]));
            ak[3] = gfma(dw2[k], d3, gfma(dw3[k] * K.k6, t, ak[3]));
            ak[4] = gfma(dw2[k], d4, gfma(dw3[k] * (R)12, t2, ak[4]));
            ak[5] = gfma(dw2[k], d5, gfma(dw3[k] * (R)20, t3, ak[5]));
          }
        }
      }
      if constexpr (CH != SPL) __builtin_amdgcn_sched_barrier(0);   // keep the samples apart: one sample's corners live
    }
    }
  } else {
    (void)wdt; (void)pen_d0; (void)pen_inv_r; (void)pen_alpha; (void)pen_gd;
    jerk_init();
  }
#if defined(GTOP_STAMPS) && GTOP_STAMPS != 2
  asm volatile("" ::"v"(acc[0]), "v"(acc[5]), "v"(acc[11]), "v"(acc[17]), "v"(acc[18]));
  GTOP_STAMP(7);   // stage B done
#endif
  // ---- A_s^-T on the lane's 18 accumulators: coefficient space -> [p0,pT,v0,vT,a0,aT] per axis ----
  R jT3 = iT3, jT4 = iT4, jT5 = iT5;
  if constexpr (MINW > 2) {   // (three registers fewer to carry across the samples)
    R iTb = iT;
    asm volatile("" : "+v"(iTb));
    jT3 = iTb * iTb * iTb; jT4 = jT3 * iTb; jT5 = jT4 * iTb;
  }
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    R *g = acc + 6 * k;
    const R H3 = g[3] * jT3, H4 = g[4] * jT4, H5 = g[5] * jT5;
    const R ap = K.k10 * H3 - K.k15 * H4 + K.k6 * H5;
    const R o0 = g[0] - ap;
    const R o2 = g[1] + T * (K.k8 * H4 - K.k6 * H3 - K.k3 * H5);
    const R o3 = T * (K.k7 * H4 - (R)4 * H3 - K.k3 * H5);
    const R o4 = (R)0.5 * g[2] + T2 * (K.k1p5 * H4 - K.k1p5 * H3 - (R)0.5 * H5);
    const R o5 = T2 * ((R)0.5 * H3 - H4 + (R)0.5 * H5);
    g[0] = o0; g[1] = ap; g[2] = o2; g[3] = o3; g[4] = o4; g[5] = o5;
  }
  // ---- the one LDS round trip: tile[v][lane], then each free variable (and each segment's cost) sums its entries ----
  if constexpr (LONG) {
    if (seg_ok) {   // this chunk's columns of the 5 m-column tile
      const int col = s * LPS + li;
#pragma unroll
      for (int v = 0; v < 18; ++v) tile[v * tstride + col] = acc[v];
      cost_run += acc[18];
    }
  } else {
  if (lane < LPS * SPW) {
#pragma unroll
    for (int v = 0; v < kRedVals; ++v) tile[v * kStride + wave * (LPS * SPW) + lane] = acc[v];   // (columns of idle slots are never read)
  }
  }
  GTOP_STAMP(8);   // A^-T + tile writes issued
  GTOP_STAMP(9);
  }   // chunk
  if constexpr (NW == 2) {
    __syncthreads();   // the other wavefront's half of the tile
  } else {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");   // writers and readers are this wavefront's own lanes,
    __builtin_amdgcn_wave_barrier();                          // whose LDS operations execute in order
  }
  R csum_seg = (R)0;
  R *gl = tile + kTileRows * tstride;   // [kRounds*64] (LONG: [kMV]): the gradient for the optimizer update (MMA only)
  if constexpr (LONG) {
    // every free variable = end of segment wpt-1 + start of segment wpt (:425-432), straight from the whole tile
    for (int qi = lane; qi < n; qi += 64) {
      const int axis = (qi >= ndp) + (qi >= 2 * ndp), c = qi - axis * ndp;
      const int wpt = c / 3 + 1, der = c - 3 * (wpt - 1);   // interior waypoint 1..m-1
      const int rowB = axis * 6 + 2 * der;
      const R sa = tree_sum<R, LPS>(tile + (rowB + 1) * tstride + (wpt - 1) * LPS),
              sb = tree_sum<R, LPS>(tile + rowB * tstride + wpt * LPS);
      if constexpr (MMA) gl[qi] = (sa + sb) + K.eps;
      else if (grp_ok) a.grad[(size_t)b0 * n + qi] = (sa + sb) + K.eps;
    }
    const R ctot = gtop_wave_sum(cost_run) + (R)1e-3;   // (:417-418; every lane gets the sum)
    if constexpr (MMA) {
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");   // gl is complete (this wavefront's own LDS writes)
      __builtin_amdgcn_wave_barrier();
      gtop_mma_update_core(st, mvecs[0], msc[0], n, lane, (double)ctot, static_cast<const R *>(gl));
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");   // the tile is rewritten by the next evaluation
      __builtin_amdgcn_wave_barrier();
    } else {
      if (grp_ok & (lane == 0)) a.cost[b0] = ctot;
    }
  } else if constexpr (MANY) {
    // the wavefront's nt n free variables, then its nt costs, dealt over the lanes 64 at a time; trajectory tq's
    // variable i sits at b0 n + qi with qi = tq n + i: the rows of a wavefront are consecutive, the stores coalesced
    const int ntn = nt * n, ntasks = ntn + nt;
    const unsigned inv_n = (unsigned)(((1u << 24) + (unsigned)n - 1u) / (unsigned)n);   // qi / n exactly: qi < 1 024, n <= 567
    for (int r0 = 0; r0 < ntasks; r0 += 64) {
      const int qi = r0 + lane;
      if (qi < ntn) {
        const int tq = (int)(((unsigned)qi * inv_n) >> 24);
        const int i = qi - tq * n;
        const int axis = (i >= ndp) + (i >= 2 * ndp), c = i - axis * ndp;
        const int wpt = c / 3 + 1, der = c - 3 * (wpt - 1);   // interior waypoint 1..m-1
        const int rowB = axis * 6 + 2 * der, sA = tq * m + wpt - 1;
        // end of segment wpt-1 (entry 2 der + 1) + start of segment wpt (entry 2 der)  (:425-432)
        const R gq = (tree_sum<R, LPS>(tile + (rowB + 1) * kStride + sA * LPS) +
                      tree_sum<R, LPS>(tile + rowB * kStride + (sA + 1) * LPS)) + K.eps;
        if (grp_ok & (b0 + tq < a.B)) a.grad[(size_t)b0 * n + qi] = gq;
      } else if (qi < ntasks) {
        const int t = qi - ntn;
        const R *row = tile + 18 * kStride + t * m * LPS;
        R csum = tree_sum<R, LPS>(row);
        for (int sg = 1; sg < m; ++sg) csum += tree_sum<R, LPS>(row + sg * LPS);   // in segment order, whatever the trajectory's place in the wavefront
        if (grp_ok & (b0 + t < a.B)) a.cost[b0 + t] = csum + (R)1e-3;   // (:417-418)
      }
    }
  } else {
#pragma unroll
  for (int r = 0; r < kRounds; ++r) {
    const R sa = tree_sum<R, LPS>(tile + offA[r]), sb = tree_sum<R, LPS>(tile + offB[r]);
    R gq = (sa + sb) + K.eps;
    asm volatile("" : "+v"(gq));   // both reads of the round in front of the store's branch: one LDS round trip, not two
    if (r == kRounds - 1 && cost_lane) csum_seg = sa;
    if constexpr (MMA) {
      if (okq[r]) gl[lane + 64 * r] = gq;   // consumed below; nothing leaves the chip
    } else {
      if (okq[r]) a.grad[(size_t)b0 * n + tid + 64 * NW * r] = gq;
    }
  }
  // ---- cost (:417-418): every term is already weighted; lanes 48.. hold the segment sums ----
  {
    R cpart = cost_lane ? csum_seg : (R)0;
    cpart += gtop_dpp_move<0x111>(cpart);   // row_shr:1
    cpart += gtop_dpp_move<0x112>(cpart);   // row_shr:2
    cpart += gtop_dpp_move<0x114>(cpart);   // row_shr:4  -> lane 55 holds lanes 48..55, lane 63 lanes 56..63
    if constexpr (NT == 2 && MMA) {
      // f(xcur) of each trajectory to every lane (lane 55: the first's, lane 63: the second's), then the two updates
      const unsigned long long u = __builtin_bit_cast(unsigned long long, (double)(cpart + (R)1e-3));
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");   // gl is complete (this wavefront's own LDS writes)
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const unsigned lo32 = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)u, 55 + 8 * t);
        const unsigned hi32 = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(u >> 32), 55 + 8 * t);
        const double fcur = __builtin_bit_cast(double, ((unsigned long long)hi32 << 32) | lo32);
        if (mma_live[t]) gtop_mma_update_core(st, mvecs[t], msc[t], n, lane, fcur, static_cast<const R *>(gl) + t * n);
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");   // the tile is rewritten by the next evaluation
      __builtin_amdgcn_wave_barrier();
    } else if constexpr (NT == 2) {
      if (grp_ok & (lane == 55)) a.cost[b0] = cpart + (R)1e-3;
      if (grp_ok & (lane == 63) & (b0 + 1 < a.B)) a.cost[b0 + 1] = cpart + (R)1e-3;
    } else if constexpr (NW * SPW > 8 && !MMA) {
      cpart += gtop_dpp_move<0x118>(cpart);   // row_shr:8 -> lane 63 holds lanes 48..63 (up to 12 segments)
      if (grp_ok & (wave == NW - 1) & (lane == 63)) a.cost[b0] = cpart + (R)1e-3;
    } else if constexpr (MMA) {
      // f(xcur) to every lane, then this wavefront's optimizer step for its trajectory
      constexpr int kCostLane = SPW > 8 ? 63 : 55;
      if constexpr (SPW > 8) cpart += gtop_dpp_move<0x118>(cpart);   // row_shr:8 (up to 12 segments)
      const unsigned long long u = __builtin_bit_cast(unsigned long long, (double)(cpart + (R)1e-3));
      const unsigned lo32 = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)u, kCostLane);
      const unsigned hi32 = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(u >> 32), kCostLane);
      const double fcur = __builtin_bit_cast(double, ((unsigned long long)hi32 << 32) | lo32);
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");   // gl is complete (this wavefront's own LDS writes)
      __builtin_amdgcn_wave_barrier();
      gtop_mma_update_core(st, mvecs[0], msc[0], n, lane, fcur, static_cast<const R *>(gl));
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");   // the tile is rewritten by the next evaluation
      __builtin_amdgcn_wave_barrier();
    } else {
      if (grp_ok & (lane == 55)) a.cost[b0] = cpart + (R)1e-3;
    }
  }
  }   // !LONG
  GTOP_STAMP(10);   // gradient stored (issued)
#ifdef GTOP_STAMPS
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  GTOP_STAMP(11);   // stores acknowledged
#endif
  }   // pass
  if constexpr (MMA) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int t = 0; t < NT; ++t) {
    if (mma_live[t]) {   // the state goes home
      const double *mvt = mv + t * kState;
      const int bt = b0 + t;
      const size_t o = (size_t)bt * n;
      for (int j = lane; j < n; j += 64) {
        st.x[o + j] = mvt[j];
        st.xcur[o + j] = mvt[kMV + j];
        st.xprev[o + j] = mvt[2 * kMV + j];
        st.xprevprev[o + j] = mvt[3 * kMV + j];
        st.dfdx[o + j] = mvt[4 * kMV + j];
        st.sigma[o + j] = mvt[5 * kMV + j];
      }
      if (lane == 0) gtop_mma_store_scalars(st, bt, msc[t]);
      // the results, where the caller wants them (otherwise: copies + mma_finish_kernel after this launch)
      if (st.out_x)
        for (int j = lane; j < n; j += 64) st.out_x[o + j] = mvt[j];
      if (lane == 0) {
        if (st.out_minf) st.out_minf[bt] = msc[t].minf;
        if (st.out_code) st.out_code[bt] = msc[t].state >= 3 ? msc[t].state : GTOP_MMA_MAXEVAL_REACHED;
        if (st.out_nevals) st.out_nevals[bt] = msc[t].nevals;
      }
    }
    }
  }
}

}  // namespace

// WIDE = false needs 24-bit (signed) multiplicands and corner records below 4 GiB (record_loads)
bool gtop_field_is_narrow(int nx, int ny, int nz, size_t elem) {
  const unsigned long long nrec = (unsigned long long)(nx + 1) * (ny + 1) * (nz + 2);
  return (unsigned long long)(nx + 1) * (ny + 1) < (1ull << 23) && nz + 2 < (1 << 23) && nrec * 4 * elem < (1ull << 32);
}

#ifndef GTOP_TWO_PER_WAVE_F64_FROM
#define GTOP_TWO_PER_WAVE_F64_FROM 4096
#endif
#ifndef GTOP_TWO_PER_WAVE_F32_FROM
#define GTOP_TWO_PER_WAVE_F32_FROM 2048
#endif
// The optimizer loop with two trajectories per wavefront (both states in LDS, the update once per trajectory) needs the
// two-wavefront register budget, where the plain two-per-wavefront body runs three wavefronts per SIMD.  Measured per
// pass of the loop (one box, us, one | two per wavefront): fp32 evaluations B = 2 048 5.0 | 5.2, 3 072 9.0 | 7.5,
// 4 096 11.2 | 7.6, 8 192 19.2 | 14.5, 16 384 35.8 | 28.0 — from 3 072; fp64 4 096 9.6 | 9.8, 8 192 18.4 | 18.5,
// 16 384 36.2 | 35.3: no gain, the fp64 loop stays at one per wavefront unless six samples per lane are pinned.
#ifndef GTOP_OPT_TWO_PER_WAVE_F32_FROM
#define GTOP_OPT_TWO_PER_WAVE_F32_FROM 3072
#endif
#ifndef GTOP_OPT_TWO_PER_WAVE_F64_FROM
#define GTOP_OPT_TWO_PER_WAVE_F64_FROM (1 << 30)
#endif
// One lane per segment (SPL = 30, as many trajectories per wavefront as fit) has a quarter fewer instructions per
// trajectory than five lanes per segment — and four times the distinct 128-byte lines per load instruction (the lanes
// of an instruction are 64 different segments; five lanes of a segment share a line or two), which is what the vector
// L1 counts.  Measured (one box, us, launch rule | one lane per segment): fp64 B = 16 384 29.6 | 34.1, 65 536 113.9 |
// 130.2, 131 072 240 | 259 — never; fp32 (half the loads per sample) 16 384 23.4 | 26.1, 32 768 47.2 | 51.0, 65 536
// 85.8 | 80.1, 131 072 166 | 135 (-19 %); 12 segments fp32 32 768 82.5 | 79.2.  So: fp32 only, from B x m = 393 216.
#ifndef GTOP_THREE_LANES_SHORT_FROM   // three lanes per segment: 2 .. 5 segments from this batch, 7 .. 10 from that
#define GTOP_THREE_LANES_SHORT_FROM 8192
#endif
#ifndef GTOP_THREE_LANES_MID_FROM
#define GTOP_THREE_LANES_MID_FROM 4096
#endif
#ifndef GTOP_ONE_LANE_F32_FROM_SEGMENTS
#define GTOP_ONE_LANE_F32_FROM_SEGMENTS (65536LL * 6)
#endif
#ifndef GTOP_TWO_WAVES_UP_TO
#define GTOP_TWO_WAVES_UP_TO 1024   // trajectories of 7 .. 12 segments: two wavefronts each up to this batch (2 048 wavefronts)
#endif
#ifndef GTOP_WAVE_MINW3_FROM
#define GTOP_WAVE_MINW3_FROM 3072   // batches that put a third wavefront on a SIMD (1 024 SIMDs)
#endif

// LDS of one workgroup (= one wavefront): the tile, the optimizer's gradient rows and — for the optimizer loop — its
// state (eight vectors, Df, T).  Mirrors the pointer arithmetic of gtop_eval_wave_kernel.
static size_t wave_lds_bytes(const GtopEvalPlan &p, int m, size_t elem, bool mma) {
  if (!p.is_long) {
    const int kmv = (p.spl == 6 && p.nt == 1) ? 128 : 64;
    const int stride = p.nw == 2 ? 121 : red_stride(p.spl);
    return (size_t)(kRedVals * stride + 128 + (mma ? p.nt * (8 * kmv + 32) : 0)) * elem;
  }
  const int n = 9 * (m - 1), kmv = (n + 63) & ~63, tstride = ((kSamples / 6) * m) | 1;
  return ((size_t)18 * tstride + (mma ? (size_t)9 * kmv + 18 + m + 8 : 0)) * elem;
}

// The launch rule (measured).  Up to 6 segments: ten lanes per segment, one wavefront per
// trajectory, up to 12 288 trajectories in fp64 and 8 192 in fp32, where two trajectories per wavefront at five lanes
// per segment take over (fp32: packed sample pairs).  7 .. 12 segments: five lanes per segment, one
// trajectory per wavefront.  Past 12: the same wavefront walks the segments 12 at a time (LONG).  The optimizer loop
// follows the same rule with its own switch points (elem: the precision of its evaluations).  pinned_spl = 3 or 6 overrides the lanes-per-segment choice where it can
// be honoured (3: up to 6 segments; 30 = one lane per segment: up to 12 segments, plain evaluations; by itself the rule
// takes it for fp32 batches of 65 536 six-segment trajectories and more).
bool gtop_eval_plan(int B, int m, size_t elem, int pinned_spl, bool for_optimizer, GtopEvalPlan *plan) {
  if (m < 2 || (pinned_spl != 0 && pinned_spl != 3 && pinned_spl != 6 && pinned_spl != 10 && pinned_spl != 30)) return false;
  GtopEvalPlan p{};
  p.nw = 1;
  // one lane per segment (SPL = 30; the kernel's MANY): trajectories of up to 12 segments, 64 / m of them per wavefront,
  // plain evaluation only — for batches that put several such wavefronts on every SIMD
  const bool many_ok = m <= 64 && !for_optimizer;   // (a wavefront has 64 segment slots)
  if (pinned_spl == 30 && !many_ok) return false;
  // three lanes per segment (SPL = 10): 21 segment slots per wavefront, 21 / m whole trajectories of up to 10 segments.
  // Five lanes per segment hold 12 slots — two trajectories of up to 6 segments or one of up to 12 — and leave most of a
  // wavefront idle for every length but 6, 11 and 12 (busy lanes, five | three per segment: m = 2: 20 | 60, 3: 30 | 63,
  // 4: 40 | 60, 5: 50 | 60, 6: 60 | 54, 7: 35 | 63, 8: 40 | 48, 9: 45 | 54, 10: 50 | 60, 11: 55 | 33, 12: 60 | 36).
  // Measured (one box, us, launch rule of before | three lanes per segment), B = 16 384 fp64: m = 2 27.1 | 13.3, 3 27.5 |
  // 18.5, 4 28.1 | 24.2, 5 29.1 | 25.5; fp32: 3 22.6 | 14.9, 4 22.8 | 18.4, 5 23.1 | 19.2; B = 8 192 fp64: 7 26.7 | 18.6,
  // 8 27.0 | 23.4, 9 27.2 | 23.7, 10 27.7 | 24.2; fp32: 7 21.7 | 14.7, 8 21.9 | 17.9, 10 22.1 | 18.3; B = 4 096: m = 4
  // 9.1 | 9.5, 7 14.6 | 12.9, 10 15.1 | 13.8 (fp32 7: 11.2 | 9.6); B = 3 072, m = 8: 11.5 | 12.4; 2 048: alike.
  const bool three_ok = m <= 10 && !for_optimizer;
  if (pinned_spl == 10 && !three_ok) return false;
  const bool three_auto = pinned_spl == 0 && three_ok && m != 6 &&
                          B >= (m <= 5 ? GTOP_THREE_LANES_SHORT_FROM : GTOP_THREE_LANES_MID_FROM);
  if (three_ok && (pinned_spl == 10 || three_auto) && !(pinned_spl == 0 && many_ok && elem == 4 &&
                                                        (long long)B * m >= GTOP_ONE_LANE_F32_FROM_SEGMENTS)) {
    p.spl = 10;
    p.nt = 21 / m;
    p.is_long = false;
    *plan = p;
    return true;
  }
  // ... and past 12 segments (up to 64: a wavefront's segment slots) where the chunked body — 12 segments at a time at
  // five lanes per segment — ends on a mostly idle chunk: 64 / m trajectories per wavefront against ceil(m / 12) chunks
  // per trajectory.  Measured (one box, us, chunked | one lane per segment, fp64): B = 8 192 m = 13 56.0 | 34.7, 17 58.4 |
  // 47.7, 22 58.8 | 61.8, 24 60.3 | 64.2, 25 83.4 | 70.5, 32 110 | 87.7, 33 110 | 109, 36 111 | 112; B = 4 096 m = 13
  // 29.8 | 24.0, 17 31.5 | 32.4, 32 60.8 | 38.9, 40 90.3 | 61.1, 48 107 | 68.5, 64 192 | 87.6; B = 2 048 m = 13 15.5 | 19.6,
  // 17 16.6 | 21.2, 32 39.4 | 25.5; fp32 B = 8 192 m = 13 43.9 | 23.6, 17 45.2 | 33.9, 24 46.1 | 43.0, 32 65.6 | 48.4, 36 66.1
  // | 78.1.  So: when (trajectories per wavefront) x (chunks) >= 5, or from four chunks, and the batch gives every SIMD
  // a wavefront.
  bool many_long = false;
  if (pinned_spl == 0 && many_ok && m > 12) {
    const int nt30 = 64 / m, chunks = (m + 11) / 12;
    many_long = (nt30 * chunks >= 5 || chunks >= 4) && B >= 1024 * nt30;
  }
  if (many_ok && (pinned_spl == 30 || many_long ||
                  (pinned_spl == 0 && m <= 12 && elem == 4 && (long long)B * m >= GTOP_ONE_LANE_F32_FROM_SEGMENTS))) {
    p.spl = 30;
    p.nt = 64 / m;
    p.is_long = false;
    if (wave_lds_bytes(p, m, elem, false) > 160u * 1024u) return false;
    *plan = p;
    return true;
  }
  // (two trajectories per wavefront at five lanes per segment amortise the per-lane set-up — coefficients, jerk term,
  // A^-T — over six samples instead of three: fewer instructions per trajectory, longer chains per wavefront; it wins
  // once the batch puts several wavefronts on every SIMD.  Round 4, corner records and the sample's loads issued
  // together (one box, us, ten lanes | five lanes per segment): fp64 B = 3 072 8.0 | 9.1, 4 096 10.4 | 9.7, 8 192 18.6 |
  // 16.9, 16 384 36.9 | 30.7 — from 4 096 (round 3: 12 288); fp32, packed pairs: 2 048 5.25 | 4.90, 4 096 8.3 | 7.4,
  // 16 384 27.0 | 23.5 — from 2 048 (round 3: 8 192))
  if (m <= 6) {
    const int from = for_optimizer ? (elem == 4 ? GTOP_OPT_TWO_PER_WAVE_F32_FROM : GTOP_OPT_TWO_PER_WAVE_F64_FROM)
                                   : (elem == 4 ? GTOP_TWO_PER_WAVE_F32_FROM : GTOP_TWO_PER_WAVE_F64_FROM);
    p.spl = pinned_spl ? pinned_spl : (B >= from ? 6 : 3);
  }
  else if (m <= 12 && !for_optimizer && pinned_spl != 6 &&
           (pinned_spl == 3 || B <= (elem == 4 ? GTOP_TWO_WAVES_UP_TO / 2 : GTOP_TWO_WAVES_UP_TO))) {
    // 7 .. 12 segments, a batch that leaves SIMDs idle with one wavefront per trajectory: two wavefronts per
    // trajectory at ten lanes per segment (measured on one box, 12 segments, fp64: B = 1 3.6 us against 5.8 on one
    // wavefront, 256: 4.1 / 6.2, 1 024: 6.2 / 7.0, 1 280: 9.1 / 9.4; fp32, whose one-wavefront body runs packed
    // pairs: 512: 4.9 / 7.1, 768: 5.1 / 4.9)
    p.spl = 3;
    p.nw = 2;
  } else if (pinned_spl == 3) return false;   // ten lanes per segment: six segments fill a wavefront, twelve fill two
  else p.spl = 6;
  p.is_long = m > 12;
  p.nt = (p.spl == 6 && 2 * m <= 12) ? 2 : 1;
  // (the optimizer's state and tile are fp64 whatever precision its evaluations run in)
  if (wave_lds_bytes(p, m, for_optimizer ? sizeof(double) : elem, for_optimizer) > 160u * 1024u) return false;   // ~200 segments
  *plan = p;
  return true;
}

namespace {

template <typename R, typename MM>
using WaveKernelFn = void (*)(const R *, const R *, const R *, const R *, int, int, int, int, int, int,
                              const GtopKernelArgs<R>, const GtopWaveConsts<R>, const MM, const GtopSetupConsts<R>);

// one geometry: the collision-free, the ordinary and the DYN instantiation (DYN: one sample at a time, MINW >= 3)
template <typename R, bool WIDE, int SPL, int NT, int MINW, typename MM, bool LONG, int NW = 1>
static WaveKernelFn<R, MM> pick_body(bool colli, bool dyn) {
  if (!colli) return gtop_eval_wave_kernel<R, WIDE, SPL, NT, false, MINW, MM, false, LONG, NW>;   // (:346: no sample loop, no DYN)
  if (dyn) return gtop_eval_wave_kernel<R, WIDE, SPL, NT, true, (MINW < 3 ? 3 : MINW), MM, true, LONG, NW>;
  return gtop_eval_wave_kernel<R, WIDE, SPL, NT, true, MINW, MM, false, LONG, NW>;
}

template <typename R, bool WIDE, typename MM>
static WaveKernelFn<R, MM> pick_geometry(const GtopEvalPlan &p, int B, bool colli, bool dyn) {
  constexpr bool MMA = !std::is_same<MM, GtopNoMma>::value;
  if (p.is_long) return pick_body<R, WIDE, 6, 1, 3, MM, true>(colli, dyn);
  if (p.spl == 30) {
    if constexpr (MMA) return nullptr;   // (the optimizer loop: one or two trajectories per wavefront)
    else return pick_body<R, WIDE, 30, 1, 3, MM, false>(colli, dyn);
  }
  if (p.spl == 10) {
    if constexpr (MMA) return nullptr;
    else return pick_body<R, WIDE, 10, 1, 3, MM, false>(colli, dyn);
  }
  if constexpr (!MMA) {
    if (p.nw == 2) return pick_body<R, WIDE, 3, 1, 2, MM, false, 2>(colli, dyn);   // (small batches only: the latency structure)
  }
  if (p.spl == 3) {
    // latency variant (every corner load of a lane in flight, 252 VGPRs) up to the batch that puts a third wavefront
    // on a SIMD; the optimizer loop at every size (its update's working set spills a 168-VGPR budget); 64-bit field
    // indices cost the 168-VGPR fp64 body 14 spilled registers: those stay on the two-wavefront budget too
    if constexpr (!MMA && !(WIDE && sizeof(R) == 8)) {
      if (B >= GTOP_WAVE_MINW3_FROM) return pick_body<R, WIDE, 3, 1, 3, MM, false>(colli, dyn);
    }
    return pick_body<R, WIDE, 3, 1, 2, MM, false>(colli, dyn);
  }
  if (p.nt == 2) {
    if constexpr (MMA) {   // (the loop never takes two trajectories per wavefront with DYN: gtop_optimize_device_ex)
      if (dyn) return nullptr;
      return colli ? gtop_eval_wave_kernel<R, WIDE, 6, 2, true, 3, MM, false, false, 1>
                   : gtop_eval_wave_kernel<R, WIDE, 6, 2, false, 3, MM, false, false, 1>;
    } else {
      return pick_body<R, WIDE, 6, 2, 3, MM, false>(colli, dyn);
    }
  }
  return pick_body<R, WIDE, 6, 1, 3, MM, false>(colli, dyn);
}

// one launch covers up to 2^25 wavefronts (grid x 64 threads stays below 2^32); a larger batch goes in slices
constexpr int kMaxGroupsPerLaunch = 1 << 25;

template <typename R, typename MM>
static hipError_t launch_wave(const GtopKernelArgs<R> &args, const MM &st, const GtopEvalPlan &plan, bool dyn,
                              hipStream_t stream) {
  constexpr bool MMA = !std::is_same<MM, GtopNoMma>::value;
  if (args.B <= 0) return hipSuccess;
  GtopKernelArgs<R> wa = args;
  if (wa.step == 1) wa.ws = (R)0;   // :412-415, applied here so that the kernel need not fetch `step`
  const bool colli = !((wa.wc < (R)0 ? -wa.wc : wa.wc) < (R)1e-4);   // :346
  dyn = dyn && wa.step == 2;        // the commented-out block's own test (:383)
  const bool wide = !gtop_field_is_narrow(wa.nx, wa.ny, wa.nz, sizeof(R));
  WaveKernelFn<R, MM> kern;
  if constexpr (MMA && sizeof(R) == 4) {   // (the optimizer loop with fp32 evaluations: no 64-bit-index bodies — a field past 4 GiB in fp32)
    if (wide) return hipErrorInvalidValue;
    kern = pick_geometry<R, false, MM>(plan, wa.B, colli, dyn);
  } else {
    kern = wide ? pick_geometry<R, true, MM>(plan, wa.B, colli, dyn) : pick_geometry<R, false, MM>(plan, wa.B, colli, dyn);
  }
  if (!kern) return hipErrorInvalidValue;
  const size_t smem = wave_lds_bytes(plan, wa.m, MMA ? sizeof(double) : sizeof(R), MMA);   // (the optimizer's state is fp64)
  if (smem > 160u * 1024u) return hipErrorInvalidValue;
  if (smem > 64u * 1024u) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                       (int)smem);
    if (e != hipSuccess) return e;
  }
  const int n = 9 * (wa.m - 1);
  const long long per_launch = (long long)kMaxGroupsPerLaunch * plan.nt;
  if (MMA && wa.B > per_launch) return hipErrorInvalidValue;   // (the optimizer's state for 2^25 trajectories is 90 GB)
  for (long long b0 = 0; b0 < args.B; b0 += per_launch) {
    GtopKernelArgs<R> s = wa;
    s.B = (int)((args.B - b0) < per_launch ? (args.B - b0) : per_launch);
    s.x = wa.x + (size_t)b0 * n;
    s.Df = wa.Df + (size_t)b0 * 18;
    s.T = wa.T + (size_t)b0 * wa.t_stride;
    s.cost = wa.cost ? wa.cost + b0 : nullptr;
    s.grad = wa.grad ? wa.grad + (size_t)b0 * n : nullptr;
    const int groups = (s.B + plan.nt - 1) / plan.nt;
    const int grid = 8 * ((groups + 7) / 8);   // the kernel deals its workgroups over 8 XCD-contiguous ranges
    hipLaunchKernelGGL(kern, dim3(grid), dim3(64 * plan.nw), smem, stream, s.x, s.Df, s.T, s.sdf, s.B, s.m, s.t_stride, s.nx, s.ny,
                       s.nz, s, GtopWaveConsts<R>{}, st, GtopSetupConsts<R>{});
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
  }
  return hipSuccess;
}

}  // namespace

template <typename R>
hipError_t gtop_launch_eval(const GtopKernelArgs<R> &args, const GtopEvalPlan &plan, bool dyn, hipStream_t stream) {
  return launch_wave<R, GtopNoMma>(args, GtopNoMma{}, plan, dyn, stream);
}

// the optimizer loop: st.iters evaluations at st.xcur, each followed by the CCSA-MMA update, in one launch (fp64)
hipError_t gtop_launch_eval_mma(const GtopKernelArgs<double> &args, const GtopMmaState &st, const GtopEvalPlan &plan,
                                bool dyn, hipStream_t stream) {
  if (plan.nw != 1 || plan.spl == 30 || plan.spl == 10 || (plan.nt != 1 && !(plan.nt == 2 && plan.spl == 6 && !plan.is_long))) return hipErrorInvalidValue;
  return launch_wave<double, GtopMmaState>(args, st, plan, dyn, stream);
}
// the same loop with the evaluations in fp32 on the fp32 field: args.Df / args.T still point at fp64 rows (the state,
// the bounds, the update and the results are fp64; see the kernel's `In`), args.x / cost / grad are not read
hipError_t gtop_launch_eval_mma(const GtopKernelArgs<float> &args, const GtopMmaState &st, const GtopEvalPlan &plan,
                                bool dyn, hipStream_t stream) {
  if (plan.nw != 1 || plan.spl == 30 || plan.spl == 10 || (plan.nt != 1 && !(plan.nt == 2 && plan.spl == 6 && !plan.is_long))) return hipErrorInvalidValue;
  return launch_wave<float, GtopMmaState>(args, st, plan, dyn, stream);
}

#ifdef GTOP_STAMPS
extern "C" int gtop_debug_read_stamps(unsigned long long *out /*4096*16*/) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_gtop_stamps), sizeof(unsigned long long) * 4096 * 16);
}
#endif

template hipError_t gtop_launch_eval<double>(const GtopKernelArgs<double> &, const GtopEvalPlan &, bool, hipStream_t);
template hipError_t gtop_launch_eval<float>(const GtopKernelArgs<float> &, const GtopEvalPlan &, bool, hipStream_t);

// ---------------------------------------------------------------------------
// Device-side clock stamp (gtop_device_clock_stamp): one lane reads the constant-rate wall clock (the counter the
// optimizer loop's maxtime rule uses) and folds it into minmax[0] = earliest, minmax[1] = latest stamp.  Captured as
// the first and the last node of a graph of evaluation launches, latest - earliest is the GPU's own time for the
// launches in between — no host clock, no profiler instrumentation in the measured interval.
// ---------------------------------------------------------------------------
namespace {
__global__ void __launch_bounds__(64) gtop_clock_stamp_kernel(unsigned long long *minmax) {
  if (threadIdx.x == 0) {
    const unsigned long long t = wall_clock64();
    atomicMin(minmax, t);
    atomicMax(minmax + 1, t);
  }
}
}  // namespace

hipError_t gtop_launch_clock_stamp(unsigned long long *minmax, hipStream_t stream) {
  hipLaunchKernelGGL(gtop_clock_stamp_kernel, dim3(1), dim3(64), 0, stream, minmax);
  return hipGetLastError();
}
