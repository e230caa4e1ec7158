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
// takes (x, Df, T) and works per segment (geometry: DESIGN.md §5.1 — a segment
// is sampled by LPS = 30/SPL adjacent lanes with SPL samples each, a wavefront
// holds 64/LPS segments of one or more trajectories):
//   phase 1  one lane per (segment, axis): c_{s,k} = A_s^-1 d_{s,k}, the jerk cost
//            c'Qc and the jerk gradient 2Qc taken to derivative space by A_s^-T
//   phase 2  per sample: position/velocity (float round trip), trilinear field
//            lookup with analytic gradient, exp penalty; each sample adds
//            w1_k*[t^j] + w2_k*[j t^(j-1)] to the lane's 18-entry
//            coefficient-space gradient; after its samples the lane applies
//            A_s^-T (linear, commutes with the sums) and the LPS lanes of a
//            segment are summed through a per-wavefront LDS tile
//   phase 4  each free variable = end of segment w-1 + start of segment w,
//            +1e-5; the scalar cost is a DPP wavefront reduction, +1e-3; with
//            MMA = true the workgroup then runs the optimizer update and
//            evaluates again (the whole CCSA-MMA loop in one launch).
// All structural zeros the reference multiplies through are skipped; nothing
// else is approximated.

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

#include "gtop_device_common.h"
#include "gtop_kernels.h"

namespace {

#ifndef GTOP_SAMPLE_UNROLL
#define GTOP_SAMPLE_UNROLL 1     // unroll factor of the per-lane sample loop (tuning knob)
#endif
#ifndef GTOP_MAX_THREADS
#define GTOP_MAX_THREADS 512
#endif
#ifndef GTOP_F64_MIN_WAVES
#define GTOP_F64_MIN_WAVES 2
#endif
#ifndef GTOP_F64_MIN_WAVES_ROLLED
#define GTOP_F64_MIN_WAVES_ROLLED 3
#endif
#ifndef GTOP_F32_MIN_WAVES
#define GTOP_F32_MIN_WAVES 3
#endif
#ifdef GTOP_WAVES_PER_EU         // register budget: 512 / GTOP_WAVES_PER_EU VGPRs per lane
#define GTOP_WAVES_PER_EU_ATTR __attribute__((amdgpu_waves_per_eu(GTOP_WAVES_PER_EU)))
#else
#define GTOP_WAVES_PER_EU_ATTR
#endif
constexpr int kSamples = 30;     // src/grad_traj_optimizer.cpp:351
constexpr int kRedVals = 19;     // 18 gradient entries + 1 cost per sample
// row stride of the transpose-reduction tile: the busy lanes of a wave (LPS*SPW of 64), made odd
constexpr int red_stride(int spl) {
  const int lps = kSamples / spl, busy = lps * (64 / lps);
  return busy | 1;
}
#ifndef GTOP_PIN_CONSTS
#define GTOP_PIN_CONSTS 2
#endif
#ifndef GTOP_PREISSUE
#define GTOP_PREISSUE 0
#endif
#ifndef GTOP_RED_CHUNK
#define GTOP_RED_CHUNK 19
#endif
constexpr int kRedChunkFull = GTOP_RED_CHUNK;   // values per transpose-reduction pass
// The specialised fp64 SPL = 6 bodies (one 40-control-point or two 20-control-point trajectories per
// wavefront) fit 128 VGPRs; reducing in two passes of 10 rows brings their LDS to 8.8 KB per workgroup,
// and 16 workgroups (4 wavefronts per SIMD) fit a CU.
constexpr int red_chunk(size_t elem, int spl, int tpbc) { return (elem == 8 && spl == 6 && tpbc > 0) ? 10 : kRedChunkFull; }

// Diagnostic build (-DGTOP_STAMPS): s_memtime at the phase boundaries of lane 0
// of wave 0 of the first 4096 workgroups, into a buffer of its own that nothing
// else reads.  Never defined in the shipped library.
#ifdef GTOP_STAMPS
__device__ unsigned long long g_gtop_stamps[4096][16];
#define GTOP_STAMP(i)                                                                          \
  do {                                                                                         \
    unsigned long long t_;                                                                     \
    __builtin_amdgcn_sched_barrier(0);                                                         \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");                  \
    __builtin_amdgcn_sched_barrier(0);                                                         \
    if (threadIdx.x == 0 && blockIdx.x < 4096) g_gtop_stamps[blockIdx.x][i] = t_;              \
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

template <typename R> __device__ __forceinline__ R gexp(R v);
template <> __device__ __forceinline__ double gexp<double>(double v) { return exp(v); }
template <> __device__ __forceinline__ float gexp<float>(float v) { return expf(v); }

// exp for the per-sample penalty (src/grad_traj_optimizer.cpp:509,:514): one
// range reduction x = k ln2 + r, |r| <= ln2/2, a degree-11 Taylor/Horner
// polynomial (truncation 6e-15 relative) and ldexp — about a third of the
// instructions of the library routine.  |x| beyond the fp64 exponent range
// saturates to 0 / inf through v_cvt_i32_f64 (saturating) and v_ldexp_f64;
// NaN propagates through p.
// The constants live in a struct so that the unrolled small-batch bodies can pin them
// in VGPRs (ExpConsts::pin): 24 literal dwords less to hold in SGPRs, which those bodies
// otherwise spill to VGPR lanes and re-materialise with s_mov pairs.
struct ExpConsts {
  double inv_ln2 = 1.4426950408889634074, ln2_hi = -6.93147180369123816490e-01, ln2_lo = -1.90821492927058770002e-10;
  double c[9] = {2.505210838544172e-08,    // 1/11!
                 2.755731922398589e-07,    // 1/10!
                 2.7557319223985893e-06,   // 1/9!
                 2.48015873015873e-05,     // 1/8!
                 1.984126984126984e-04,    // 1/7!
                 1.388888888888889e-03,    // 1/6!
                 8.333333333333333e-03,    // 1/5!
                 4.1666666666666664e-02,   // 1/4!
                 1.6666666666666666e-01};  // 1/3!
  __device__ __forceinline__ void pin() {
    asm volatile("" : "+v"(inv_ln2), "+v"(ln2_hi), "+v"(ln2_lo));
#pragma unroll
    for (int i = 0; i < 9; ++i) asm volatile("" : "+v"(c[i]));
  }
};
__device__ __forceinline__ double penalty_exp(double x, const ExpConsts &K) {
  const double k = rint(x * K.inv_ln2);            // x / ln2
  double r = fma(k, K.ln2_hi, x);
  r = fma(k, K.ln2_lo, r);
  double p = K.c[0];
#pragma unroll
  for (int i = 1; i < 9; ++i) p = fma(p, r, K.c[i]);
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
template <typename R> __device__ __forceinline__ R gmin(R a, R b);
template <> __device__ __forceinline__ double gmin<double>(double a, double b) { return fmin(a, b); }
template <> __device__ __forceinline__ float gmin<float>(float a, float b) { return fminf(a, b); }
template <typename R> __device__ __forceinline__ R gmax(R a, R b);
template <> __device__ __forceinline__ double gmax<double>(double a, double b) { return fmax(a, b); }
template <> __device__ __forceinline__ float gmax<float>(float a, float b) { return fmaxf(a, b); }
template <typename R> __device__ __forceinline__ R gfloor(R v);
template <> __device__ __forceinline__ double gfloor<double>(double v) { return floor(v); }
template <> __device__ __forceinline__ float gfloor<float>(float v) { return floorf(v); }
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
// Branch-free: the corner indices are clamped anyway (:166-174), so the loads
// are always in bounds and the out-of-map case is a final select.
// The four (x,y) corner columns of a lookup, each loaded as the pair
// (D[..][zb], D[..][zb+1]).  Per-axis clamp of the corner indices as in
// src/sdf_map.cpp:166-174: x0 = clamp(ix), x1 = clamp(ix+1), so x1 = x0 + 1
// exactly when 0 <= ix <= nx-2 and x1 = x0 otherwise (same for y) — the second
// column is the first plus a stride that is selected to zero at the borders.
// WIDE = false (the host checks nx*ny < 2^24, nz < 2^24, field < 4 GiB):
// 24-bit multiplies (full rate; v_mul_lo_u32 is not) and a uniform base +
// 32-bit byte offset per lane.  WIDE = true: 64-bit indices, any field.
template <typename R, bool WIDE>
__device__ __forceinline__ int corner_loads(const GtopKernelArgs<R> &a, int ix, int iy, int iz,
                                            Pair<R> &p00, Pair<R> &p01, Pair<R> &p10, Pair<R> &p11) {
  const int nx = a.nx, ny = a.ny, nz = a.nz;
  const int x0 = min(max(ix, 0), nx - 1);
  const int y0 = min(max(iy, 0), ny - 1);
  const int zb = min(max(iz, 0), nz - 2);
  const bool cx = (unsigned)ix < (unsigned)(nx - 1), cy = (unsigned)iy < (unsigned)(ny - 1);
  if constexpr (!WIDE) {
    const char *D = reinterpret_cast<const char *>(a.sdf);
    constexpr uint32_t esz = (uint32_t)sizeof(R);
    const uint32_t o00 = (__umul24(__umul24((uint32_t)x0, (uint32_t)ny) + (uint32_t)y0, (uint32_t)nz) + (uint32_t)zb) * esz;
    const uint32_t sy = cy ? (uint32_t)nz * esz : 0u;
    const uint32_t sx = cx ? (uint32_t)ny * (uint32_t)nz * esz : 0u;
    const uint32_t o10 = o00 + sx;
    p00 = *reinterpret_cast<const Pair<R> *>(D + o00);
    p01 = *reinterpret_cast<const Pair<R> *>(D + (o00 + sy));
    p10 = *reinterpret_cast<const Pair<R> *>(D + o10);
    p11 = *reinterpret_cast<const Pair<R> *>(D + (o10 + sy));
  } else {
    const R *D = a.sdf;
    const size_t i00 = ((size_t)x0 * ny + y0) * nz + zb;
    const size_t sy = cy ? (size_t)nz : 0, sx = cx ? (size_t)ny * nz : 0;
    p00 = *reinterpret_cast<const Pair<R> *>(D + i00);
    p01 = *reinterpret_cast<const Pair<R> *>(D + i00 + sy);
    p10 = *reinterpret_cast<const Pair<R> *>(D + i00 + sx);
    p11 = *reinterpret_cast<const Pair<R> *>(D + i00 + sx + sy);
  }
  return zb;
}

// The query is split in two so that a caller can put several lookups in flight
// before consuming the first: sdf_issue does the index arithmetic and issues the
// four pair loads, sdf_blend is the trilinear arithmetic on the loaded corners.
template <typename R> struct SdfTap {
  Pair<R> p00, p01, p10, p11;   // (D[x][y][zb], D[x][y][zb+1]) for the four (x,y) corners
  R dx, dy, dze;                // interpolation weights (dz already folded with the z-border clamp)
  bool out, zflat;              // outside the map; clamped at a z border (zero z-gradient)
};

// isInMap's box (sdf_map.cpp:55-69, margins included); a struct so that the unrolled
// small-batch bodies can keep it in VGPRs instead of 12 SGPRs (see ExpConsts)
template <typename R> struct MapBox {
  R lo[3], hi[3];
  R org[3], half, rinv;   // origin, res/2, 1/res of posToIndex (sdf_map.cpp:71-74, :201-204)
  __device__ __forceinline__ void pin() {
#pragma unroll
    for (int i = 0; i < 3; ++i) asm volatile("" : "+v"(lo[i]), "+v"(hi[i]));
  }
  __device__ __forceinline__ void pin_index() {
#pragma unroll
    for (int i = 0; i < 3; ++i) asm volatile("" : "+v"(org[i]));
    asm volatile("" : "+v"(half), "+v"(rinv));
  }
};

template <typename R, bool WIDE>
__device__ __forceinline__ SdfTap<R> sdf_issue(const GtopKernelArgs<R> &a, const MapBox<R> &box, R px, R py, R pz) {
  SdfTap<R> tp;
  tp.out = (px < box.lo[0]) | (py < box.lo[1]) | (pz < box.lo[2]) |
           (px > box.hi[0]) | (py > box.hi[1]) | (pz > box.hi[2]);
  const R rinv = box.rinv, half = box.half;
  // posToIndex(pos - 0.5 res)  (:201-204 -> :71-74)
  const R tx = (px - half) - box.org[0], ty = (py - half) - box.org[1], tz = (pz - half) - box.org[2];
  const R ux = tx * rinv, uy = ty * rinv, uz = tz * rinv;
  const R fx = gfloor(ux), fy = gfloor(uy);
  const int ix = (int)fx, iy = (int)fy, iz = (int)gfloor(uz);
  // indexToPos (:76-78) and diff (:209): (pos - centre(idx)) / res is the fractional
  // part of u (equal up to a few ulp of u, ~1e-14 of a voxel).  Written as the fused form the compiler
  // contracts `u - floor(u)` to where it can: every body, however it is scheduled, takes the same bits.
  tp.dx = gfma(tx, rinv, -fx);
  tp.dy = gfma(ty, rinv, -fy);

  // z is the fastest axis, so the two z-corners of each (x,y) column are one
  // 2-element load; the clamp at the z borders becomes a clamp of the weight.
  const int zb = corner_loads<R, WIDE>(a, ix, iy, iz, tp.p00, tp.p01, tp.p10, tp.p11);
  // At a z border both z-corners clamp to the same voxel (:166-174); with the
  // pair (D[zb], D[zb+1]) in hand that is dz := 0 (iz < 0) or dz := 1
  // (iz > nz-2) — i.e. uz - zb clamped to [0,1] — and a zero z-gradient.
  tp.dze = gmin(gmax(gfma(tz, rinv, -(R)zb), (R)0), (R)1);
  tp.zflat = iz != zb;
  return tp;
}

// The same with the four pair loads issued by hand (fp64, 32-bit offsets): the lone-wavefront body wants all of a
// lane's corner loads in flight BEFORE the arithmetic that does not need them, and the compiler — free to sink
// side-effect-free loads of a read-only noalias field, and keen to, at 232 VGPRs — put each sample's loads right in
// front of their use (round 2: the first sample's loads were waited for 20 instructions after their issue, whatever
// scheduling barriers said).  A volatile asm keeps its place among the phase fences; the compiler does not count
// these loads, so the caller waits for them itself (gtop_wait_pairs) before it reads `raw`.
typedef double gtop_d2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ gtop_d2 asm_load_pair(const void *base, uint32_t byte_off) {
  gtop_d2 v;
  asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(v) : "v"(byte_off), "s"(base));
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

__device__ __forceinline__ SdfTap<double> sdf_issue_asm(const GtopKernelArgs<double> &a, const MapBox<double> &box,
                                                        double px, double py, double pz, gtop_d2 (&raw)[4]) {
  typedef double R;
  SdfTap<R> tp;
  tp.out = (px < box.lo[0]) | (py < box.lo[1]) | (pz < box.lo[2]) |
           (px > box.hi[0]) | (py > box.hi[1]) | (pz > box.hi[2]);
  const R rinv = box.rinv, half = box.half;
  const R tx = (px - half) - box.org[0], ty = (py - half) - box.org[1], tz = (pz - half) - box.org[2];
  const R ux = tx * rinv, uy = ty * rinv, uz = tz * rinv;
  const R fx = gfloor(ux), fy = gfloor(uy);
  const int ix = (int)fx, iy = (int)fy, iz = (int)gfloor(uz);
  tp.dx = gfma(tx, rinv, -fx);
  tp.dy = gfma(ty, rinv, -fy);
  // corner_loads<double, false>, loads by hand
  const int nx = a.nx, ny = a.ny, nz = a.nz;
  const int x0 = min(max(ix, 0), nx - 1);
  const int y0 = min(max(iy, 0), ny - 1);
  const int zb = min(max(iz, 0), nz - 2);
  const bool cx = (unsigned)ix < (unsigned)(nx - 1), cy = (unsigned)iy < (unsigned)(ny - 1);
  const uint32_t o00 = (__umul24(__umul24((uint32_t)x0, (uint32_t)ny) + (uint32_t)y0, (uint32_t)nz) + (uint32_t)zb) * 8u;
  const uint32_t sy = cy ? (uint32_t)nz * 8u : 0u;
  const uint32_t sx = cx ? (uint32_t)ny * (uint32_t)nz * 8u : 0u;
  const uint32_t o10 = o00 + sx;
  raw[0] = asm_load_pair(a.sdf, o00);
  raw[1] = asm_load_pair(a.sdf, o00 + sy);
  raw[2] = asm_load_pair(a.sdf, o10);
  raw[3] = asm_load_pair(a.sdf, o10 + sy);
  tp.dze = gmin(gmax(gfma(tz, rinv, -(R)zb), (R)0), (R)1);
  tp.zflat = iz != zb;
  return tp;
}

// Trilinear value and gradient (src/sdf_map.cpp:211-241) in difference form:
// every interpolation is a + w (b - a), and the corner differences it needs are
// the ones the gradient is made of, so nothing is computed twice.  The gradient
// comes back UNSCALED — in distance per voxel; the caller folds 1/resolution
// (:231-239) into the weight that multiplies it.
template <typename R>
__device__ __forceinline__ R sdf_blend(const SdfTap<R> &tp, R &gx, R &gy, R &gz, bool &is_out) {
  const R dx = tp.dx, dy = tp.dy, dze = tp.dze;
  // values[x][y][z]
  const R v000 = tp.p00.x, v001 = tp.p00.y, v010 = tp.p01.x, v011 = tp.p01.y;
  const R d00 = tp.p10.x - v000, d01 = tp.p10.y - v001;   // x-differences of the four (y,z) edges
  const R d10 = tp.p11.x - v010, d11 = tp.p11.y - v011;
  const R v00 = gfma(dx, d00, v000), v01 = gfma(dx, d01, v001);   // :221-224
  const R v10 = gfma(dx, d10, v010), v11 = gfma(dx, d11, v011);
  const R e0 = v10 - v00, e1 = v11 - v01;                          // y-differences
  const R v0 = gfma(dy, e0, v00), v1 = gfma(dy, e1, v01);          // :226-227
  const R dd = v1 - v0;                                            // z-difference (:231)
  const R dist = gfma(dze, dd, v0);                                // :229
  gy = gfma(dze, e1 - e0, e0);                                     // :232-233
  const R h0 = gfma(dy, d10 - d00, d00), h1 = gfma(dy, d11 - d01, d01);
  gx = gfma(dze, h1 - h0, h0);                                     // :234-239
  gz = tp.zflat ? (R)0 : dd;
  is_out = tp.out;
  return tp.out ? (R)-1 : dist;   // the caller zeroes the gradient's weight when out (grad := 0)
}

template <typename R>
__device__ __forceinline__ R wave_sum(R v) { return gtop_wave_sum(v); }

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

// two SDFMap::getDistWithGradTrilinear queries (src/sdf_map.cpp:185-242)
template <bool WIDE>
__device__ __forceinline__ f2 sdf_query_pair(const GtopKernelArgs<float> &a, f2 px, f2 py, f2 pz,
                                             f2 &gx, f2 &gy, f2 &gz, bool &outA, bool &outB) {
  outA = (px.x < a.lo[0]) | (py.x < a.lo[1]) | (pz.x < a.lo[2]) | (px.x > a.hi[0]) | (py.x > a.hi[1]) | (pz.x > a.hi[2]);
  outB = (px.y < a.lo[0]) | (py.y < a.lo[1]) | (pz.y < a.lo[2]) | (px.y > a.hi[0]) | (py.y > a.hi[1]) | (pz.y > a.hi[2]);
  const f2 res = splat(a.res), rinv = splat(a.res_inv), half = splat(0.5f * a.res);
  const f2 ox = splat(a.origin[0]), oy = splat(a.origin[1]), oz = splat(a.origin[2]);
  f2 fx = ((px - half) - ox) * rinv, fy = ((py - half) - oy) * rinv, fz = ((pz - half) - oz) * rinv;
  fx = (f2){floorf(fx.x), floorf(fx.y)};
  fy = (f2){floorf(fy.x), floorf(fy.y)};
  fz = (f2){floorf(fz.x), floorf(fz.y)};
  const f2 h = splat(0.5f);
  const f2 dx = (px - ((fx + h) * res + ox)) * rinv;
  const f2 dy = (py - ((fy + h) * res + oy)) * rinv;
  f2 dz = (pz - ((fz + h) * res + oz)) * rinv;

  const int nz = a.nz;
  Pair<float> p00[2], p01[2], p10[2], p11[2];
  bool zflat[2];
#pragma unroll
  for (int c = 0; c < 2; ++c) {
    const int ix = (int)(c ? fx.y : fx.x), iy = (int)(c ? fy.y : fy.x), iz = (int)(c ? fz.y : fz.x);
    corner_loads<float, WIDE>(a, ix, iy, iz, p00[c], p01[c], p10[c], p11[c]);
    // z border (:166-174): both z-corners clamp to the same voxel.  With the
    // pair (D[zb], D[zb+1]) loaded, that is dz := 0 (iz = -1) or 1 (iz = nz-1)
    // and a zero z-gradient.
    const bool lo = iz < 0, hi = iz > nz - 2;
    zflat[c] = lo | hi;
    const float dzc = lo ? 0.0f : (hi ? 1.0f : (c ? dz.y : dz.x));
    if (c) dz.y = dzc; else dz.x = dzc;
  }
  const f2 v000 = {p00[0].x, p00[1].x}, v001 = {p00[0].y, p00[1].y};
  const f2 v010 = {p01[0].x, p01[1].x}, v011 = {p01[0].y, p01[1].y};
  const f2 v100 = {p10[0].x, p10[1].x}, v101 = {p10[0].y, p10[1].y};
  const f2 v110 = {p11[0].x, p11[1].x}, v111 = {p11[0].y, p11[1].y};
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
  gz = dd;
  if (zflat[0]) gz.x = 0.0f;
  if (zflat[1]) gz.y = 0.0f;
  return dist;
}

// Two samples (t.x, t.y) of one segment: everything phase 2 does per sample
// (src/grad_traj_optimizer.cpp:353-381), accumulated component-wise into
// acc2[19]; the caller adds the two components after its loop.
template <bool DYN, bool WIDE>
__device__ __forceinline__ void sample_pair_f32(const GtopKernelArgs<float> &a, const float *cq, f2 t,
                                                bool liveA, bool liveB, float wdt, float dt, f2 (&acc2)[kRedVals]) {
  const f2 t2 = t * t, t3 = t2 * t, t4 = t2 * t2, t5 = t4 * t;
  const f2 d2 = splat(2.0f) * t, d3 = splat(3.0f) * t2, d4 = splat(4.0f) * t3, d5 = splat(5.0f) * t4;   // d/dt of the powers
  f2 pos[3], vel[3], acc3[3];
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const float *q = cq + 6 * k;
    pos[k] = splat(q[0]) + splat(q[1]) * t + splat(q[2]) * t2 + splat(q[3]) * t3 + splat(q[4]) * t4 + splat(q[5]) * t5;
    vel[k] = splat(q[1]) + splat(q[2]) * d2 + splat(q[3]) * d3 + splat(q[4]) * d4 + splat(q[5]) * d5;
    if (DYN) acc3[k] = splat(2.0f * q[2]) + splat(6.0f * q[3]) * t + splat(12.0f * q[4]) * t2 + splat(20.0f * q[5]) * t3;
  }
  const f2 v2 = vel[0] * vel[0] + vel[1] * vel[1] + vel[2] * vel[2];
  const f2 vn = (f2){__builtin_amdgcn_sqrtf(v2.x), __builtin_amdgcn_sqrtf(v2.y)} + splat(1e-5f);   // :358
  const f2 ivn = {__builtin_amdgcn_rcpf(vn.x), __builtin_amdgcn_rcpf(vn.y)};
  f2 g3[3];
  bool outA, outB;
  f2 dist = sdf_query_pair<WIDE>(a, pos[0], pos[1], pos[2], g3[0], g3[1], g3[2], outA, outB);   // :363
  if (outA) dist.x = -1.0f;   // out of map (sdf_map.cpp:187): dist = -1, grad := 0
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
  if (DYN && a.step == 2) {   // the commented-out block :383-407 (see the scalar path)
    f2 cv = splat(0.0f), ca = splat(0.0f);
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const f2 av = (f2){fabsf(vel[k].x), fabsf(vel[k].y)}, aa = (f2){fabsf(acc3[k].x), fabsf(acc3[k].y)};
      const f2 xv = (av - splat(a.v0)) / splat(a.r_v), xa = (aa - splat(a.a0)) / splat(a.r_a);
      cv = splat(a.alpha_v) * (f2){expf(xv.x), expf(xv.y)};
      ca = splat(a.alpha_a) * (f2){expf(xa.x), expf(xa.y)};
      csum += (cv + ca) * vn * splat(dt);
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const f2 av = (f2){fabsf(vel[k].x), fabsf(vel[k].y)}, aa = (f2){fabsf(acc3[k].x), fabsf(acc3[k].y)};
      const f2 xv = (av - splat(a.v0)) / splat(a.r_v), xa = (aa - splat(a.a0)) / splat(a.r_a);
      const f2 gv = splat(a.alpha_v / a.r_v) * (f2){expf(xv.x), expf(xv.y)};
      const f2 ga = splat(a.alpha_a / a.r_a) * (f2){expf(xa.x), expf(xa.y)};
      w2[k] += (gv * vn + cv * (vel[k] * ivn) + ca * (vel[k] * ivn)) * splat(dt);
      w3[k] = (ga * vn) * splat(dt);
    }
    if (!liveA) { csum.x = 0.0f; w2[0].x = w2[1].x = w2[2].x = 0.0f; w3[0].x = w3[1].x = w3[2].x = 0.0f; }
    if (!liveB) { csum.y = 0.0f; w2[0].y = w2[1].y = w2[2].y = 0.0f; w3[0].y = w3[1].y = w3[2].y = 0.0f; }
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

// One workgroup owns TPB consecutive trajectories (grid-stride over groups of
// TPB), NW = blockDim/64 wavefronts.  Each polynomial segment is sampled by
// LPS = 30/SPL adjacent lanes, SPL samples per lane (sample index =
// lane-in-segment + j*LPS, so neighbouring lanes gather neighbouring voxels); a
// wavefront holds SPW = 64/LPS segments.  The workgroup's segments are
// numbered S = tl*m + s over its TPB trajectories ("virtual segments"), so the
// few-lane phases (1, 3, 4) and the reduction serve TPB trajectories per pass.
//   SPL = 1, NW = 3, TPB = 1 : a 20-control-point trajectory over 3 wavefronts
//                              (small, latency-bound batches)
//   SPL = 3, NW = 1, TPB = 1 : one wavefront per trajectory (60/64 lanes)
//   SPL = 15, NW = 1, TPB = 5: 2 lanes per segment, 5 trajectories per wavefront
// Register budget (waves per SIMD the compiler must leave room for).  fp64: the
// rolled sample loops (SPL >= 5, the large-batch geometries) fit 168 VGPRs = 3
// waves per SIMD without spilling; the unrolled small-batch bodies need ~210 and
// the optional velocity/acceleration block (DYN) and the optimizer epilogue (MMA) more.  fp32: 3 waves; the
// 128-VGPR budget of 4 waves spills.
template <typename R, int SPL, bool DYN, bool MMA, int TPBC> struct MinWaves {
  static constexpr int v = (!DYN && !MMA && SPL == 6 && TPBC > 0) ? 4
                           : (!DYN && !MMA && (SPL == 5 || SPL == 6)) ? GTOP_F64_MIN_WAVES_ROLLED : GTOP_F64_MIN_WAVES;
};
template <int SPL, bool DYN, bool MMA, int TPBC> struct MinWaves<float, SPL, DYN, MMA, TPBC> {
  static constexpr int v = DYN ? 2 : ((SPL == 6 && TPBC > 0 && !MMA) ? 4 : GTOP_F32_MIN_WAVES);
};

//
// MMA = true (fp64 only) appends the optimizer step: `a.x` is then the trial
// point st.xcur of the batched CCSA-MMA driver, and after cost and gradient of
// a trajectory are known the same workgroup runs its MMA update
// (gtop_mma_update_trajectory) and — st.iters times in all — evaluates again.
template <typename R, bool DYN, int SPL, bool MMA, bool WIDE, int TPBC>
__global__ void __launch_bounds__(GTOP_MAX_THREADS, (MinWaves<R, SPL, DYN, MMA, TPBC>::v)) GTOP_WAVES_PER_EU_ATTR
gtop_eval_kernel(const GtopKernelArgs<R> a, const GtopMmaState st) {
  constexpr int LPS = kSamples / SPL;        // lanes per segment
  constexpr int SPW = 64 / LPS;              // segments per wavefront
  constexpr int kRedStride = red_stride(SPL);
  constexpr int kRedChunk = red_chunk(sizeof(R), SPL, TPBC);
  // up to three samples per lane are unrolled outright (the small-batch geometry: one wavefront per SIMD,
  // the scheduler interleaves the samples); longer loops stay rolled to hold 2 waves per SIMD
  constexpr int kUnroll = (SPL <= 3 && !DYN) ? SPL : GTOP_SAMPLE_UNROLL;
  constexpr int CH = (GTOP_PREISSUE && SPL <= 3) ? SPL : 1;   // distance-field lookups in flight per lane
  static_assert(LPS * SPL == kSamples, "SPL must divide 30");
  extern __shared__ __align__(16) unsigned char smem_raw[];
  R *sm = reinterpret_cast<R *>(smem_raw);
  const int m = a.m, ndp = 3 * m - 3, n = 3 * ndp;
  // TPBC > 0: the launcher guarantees one wavefront per workgroup holding TPBC whole
  // trajectories (2 <= m, TPBC m <= SPW) and a workgroup per group.  With that known at
  // compile time the work-distribution loops of every phase collapse to straight-line code
  // (a lone wavefront pays four cycles per instruction of any kind) and the loop-invariant
  // lane predicates no longer overflow the SGPR file.  ONE = the single-trajectory case.
  constexpr bool FIXED = TPBC > 0, ONE = TPBC == 1;
  if constexpr (FIXED) __builtin_assume(m >= 2 && TPBC * m <= SPW);
  const int TPB = FIXED ? TPBC : a.tpb, MS = TPB * m;   // trajectories / virtual segments per workgroup
  const int tid = threadIdx.x, nthr = FIXED ? 64 : (int)blockDim.x;
  if constexpr (FIXED) __builtin_assume(tid >= 0 && tid < 64);
  const int lane = tid & 63, wave = FIXED ? 0 : tid >> 6, NW = FIXED ? 1 : nthr >> 6;
  const int slot = lane / LPS, li = lane - slot * LPS;   // segment slot in this wave, lane in segment

  R *Ts = sm;                // [MS]       segment_time
  R *coef = Ts + MS;         // [MS][3][6] polynomial coefficients (:253-279)
  R *Gs = coef + 18 * MS;    // [MS][3][6] jerk gradient, derivative space
  R *csm = Gs + 18 * MS;     // [MS][3]    jerk cost per (segment, axis)
  R *dts = csm + 3 * MS;     // [MS]       T_s / 30 (:351)
  R *ccol = Ts;              // [MS]       wc * collision (+dyn) cost per segment; takes the place of T_s,
                             //            which only the segment's own wavefront reads, before it writes this
  R *gseg = coef;            // [MS][3][6] total gradient per segment, derivative space [p0,pT,v0,vT,a0,aT];
                             //            written by a wavefront over ITS segments' coef after its sample loop
  R *red = dts + MS;         // [NW][kRedChunk][kRedStride] per-wave transpose-reduction tile
  R *myred = red + wave * (kRedChunk * kRedStride);

  const R ws = (a.step == 1) ? (R)0 : a.ws;  // :412-415
  ExpConsts expk;
  R pen_d0 = a.d0, pen_inv_r = a.inv_r, pen_alpha = a.alpha, pen_gd = -a.alpha_over_r;   // penalty parameters (:507-515)
  MapBox<R> mapbox = {{a.lo[0], a.lo[1], a.lo[2]}, {a.hi[0], a.hi[1], a.hi[2]},
                      {a.origin[0], a.origin[1], a.origin[2]}, (R)0.5 * a.res, a.res_inv};
  if constexpr (GTOP_PIN_CONSTS && !kIsF32<R> && SPL <= 3 && !MMA && !DYN) {   // (MMA/DYN bodies have no VGPRs to spare)
    expk.pin();
    if (GTOP_PIN_CONSTS > 1) mapbox.pin();
    if (GTOP_PIN_CONSTS > 2) mapbox.pin_index();
    if (GTOP_PIN_CONSTS > 3) asm volatile("" : "+v"(pen_d0), "+v"(pen_inv_r), "+v"(pen_alpha), "+v"(pen_gd));
  }
  const R wc = a.wc;
  const bool do_colli = !(gabs(wc) < (R)1e-4);  // :346
  const unsigned long long t_launch = MMA ? wall_clock64() : 0ull;   // for the wall-clock stop of the optimizer loop

  // XCD-aware order: workgroups are dealt round-robin over the 8 XCDs (blocks b
  // and b+8 share an L2), so give XCD x the x-th contiguous eighth of the
  // batch: with a spatially ordered batch each L2 then serves one region of
  // the distance field.  Pure speed; any order is correct.
  const int ngroups = (a.B + TPB - 1) / TPB;
  const int per_xcd = (ngroups + 7) >> 3;
  for (int vb = blockIdx.x; vb < 8 * per_xcd; vb += gridDim.x) {
    const int grp = (vb & 7) * per_xcd + (vb >> 3);
    if (grp >= ngroups) continue;      // block-uniform
    const int b0 = grp * TPB;
    // MMA: the trajectories of a group are independent of every other group, so the whole
    // optimizer loop of the group runs here — evaluate at xcur, update, evaluate again —
    // with st.iters evaluations per launch.  What one wavefront of the workgroup writes
    // (xcur and the MMA state) the next pass reads after a workgroup barrier; the CU's
    // vector L1 is shared by the workgroup, so that needs no cache maintenance.
    const int npass = MMA ? st.iters : 1;
    for (int pass = 0; pass < npass; ++pass) {
    GTOP_STAMP(0);
    GTOP_STAMP_HWID();
    const int ntraj = ONE ? 1 : min(TPB, a.B - b0);   // trajectories this pass
    if constexpr (MMA) {
      // Stop rules (mma.hpp:35-39; the reference's own is set_maxtime, :144-148): a group whose trajectories have
      // all stopped (ftol/xtol, gtop_mma_update_trajectory) leaves the loop instead of burning the remaining
      // evaluations; past the wall-clock limit the ones still running stop where they are (after at least one
      // evaluation).  Decided by one lane and shared through LDS: the branch must be workgroup-uniform.
      __shared__ int s_stop;
      if (tid == 0) {
        int running = 0;
        for (int tl = 0; tl < ntraj; ++tl)
          running += __hip_atomic_load(&st.state[b0 + tl], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < 3;
        int stop = running == 0;
        if (!stop && st.max_ticks > 0 && pass > 0 && (long long)(wall_clock64() - t_launch) > st.max_ticks) {
          for (int tl = 0; tl < ntraj; ++tl)
            if (__hip_atomic_load(&st.state[b0 + tl], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < 3)
              st.state[b0 + tl] = GTOP_MMA_MAXTIME_REACHED;
          stop = 1;
        }
        s_stop = stop;
      }
      __syncthreads();
      const int stop = s_stop;
      __syncthreads();   // s_stop is rewritten by the next pass
      if (stop) break;
    }
    const int nseg = ntraj * m;             // live virtual segments
    // ---- phase 1: per (segment, axis): coefficients, jerk cost and jerk gradient ----
    // Its 3*m*TPB lanes read their seven inputs (two waypoints' p,v,a and T_s)
    // straight from HBM/L2 — there is no staging pass: nothing else needs x or Df.
    for (int w = tid; w < 3 * nseg; w += nthr) {
      const int S = w / 3, k = w - 3 * S;
      int tl = 0, s = S;
      if constexpr (TPBC == 2) {
        tl = s >= m;
        s -= tl * m;
      } else if constexpr (!ONE) {
        while (s >= m) { s -= m; ++tl; }
      }
      const R *xk = a.x + (size_t)(b0 + tl) * n + k * ndp;    // free variables of axis k (:182-187)
      const R *df = a.Df + (size_t)(b0 + tl) * 18 + k * 6;    // [p,v,a]_start, [p,v,a]_end
      // derivative vector layout (src/qp_generator.cpp:363-387): start | end | waypoint 1 | ... | waypoint m-1
      const R *w0 = (s == 0) ? df : xk + 3 * (s - 1);
      const R *w1 = (s + 1 == m) ? df + 3 : xk + 3 * s;
      const R p0 = w0[0], v0 = w0[1], a0 = w0[2];
      const R pT = w1[0], vT = w1[1], aT = w1[2];
      const R T = a.T[(size_t)(b0 + tl) * a.t_stride + s];
      const R T2 = T * T, T3 = T2 * T, T4 = T2 * T2, T5 = T4 * T;
      const R iT = fast_rcp(T), iT3 = iT * iT * iT, iT4 = iT3 * iT, iT5 = iT4 * iT;
      // closed-form A_s^-1 (rows of A_s: src/qp_generator.cpp:185-195)
      const R P = pT - p0 - v0 * T - (R)0.5 * a0 * T2;
      const R V = (vT - v0 - a0 * T) * T;
      const R A = (aT - a0) * T2;
      const R c3 = ((R)10 * P - (R)4 * V + (R)0.5 * A) * iT3;
      const R c4 = ((R)-15 * P + (R)7 * V - A) * iT4;
      const R c5 = ((R)6 * P - (R)3 * V + (R)0.5 * A) * iT5;
      R *cf = coef + w * 6;
      cf[0] = p0; cf[1] = v0; cf[2] = (R)0.5 * a0; cf[3] = c3; cf[4] = c4; cf[5] = c5;
      // jerk Hessian Q_s (src/qp_generator.cpp:226-234): i,j in {3,4,5}
      const R q3 = (R)36 * T * c3 + (R)72 * T2 * c4 + (R)120 * T3 * c5;
      const R q4 = (R)72 * T2 * c3 + (R)192 * T3 * c4 + (R)360 * T4 * c5;
      const R q5 = (R)120 * T3 * c3 + (R)360 * T4 * c4 + (R)720 * T5 * c5;
      csm[w] = c3 * q3 + c4 * q4 + c5 * q5;   // c'Qc  == this (s,k)'s share of d'Rd (:326-327)
      // ws * 2Qc == share of ws*(2Rfp'df + 2Rpp dp) (:330-336), taken to derivative
      // space [p0,pT,v0,vT,a0,aT] by A_s^-T right away (its coefficient-space
      // entries 0..2 are zero)
      const R H3 = ws * (R)2 * q3 * iT3, H4 = ws * (R)2 * q4 * iT4, H5 = ws * (R)2 * q5 * iT5;
      const R ap = (R)10 * H3 - (R)15 * H4 + (R)6 * H5;
      R *g = Gs + w * 6;
      g[0] = -ap;
      g[1] = ap;
      g[2] = T * ((R)-6 * H3 + (R)8 * H4 - (R)3 * H5);
      g[3] = T * ((R)-4 * H3 + (R)7 * H4 - (R)3 * H5);
      g[4] = T2 * ((R)-1.5 * H3 + (R)1.5 * H4 - (R)0.5 * H5);
      g[5] = T2 * ((R)0.5 * H3 - H4 + (R)0.5 * H5);
      if (k == 0) {
        Ts[S] = T;
        dts[S] = T / (R)30.0;   // :351
        if (!do_colli) ccol[S] = (R)0;   // (aliases Ts: only when the sample phase will not read it)
      }
      if constexpr (FIXED) break;   // 3 TPBC m <= 3 SPW <= 64 work items: one trip
    }
    __syncthreads();
    GTOP_STAMP(1);
    GTOP_STAMP(2);

    // Sample times: the reference's `for (t = 1e-3; t < T; t += dt)` (:353)
    // accumulates t by repeated addition.  For T >= 0.0301 all 30 samples pass the
    // loop test whatever the rounding, and t_i = 1e-3 + i*dt differs from the
    // accumulated value by a few ulp (1e-15 relative, far inside the 1e-5 budget;
    // SURVEY A.4 Q8), so lanes form it with one fma.  Below that the sample COUNT
    // depends on the accumulated value (29 at T = 0.03, fewer for tinier T), so for
    // those segments each lane replays the addition chain.
    // ---- phase 2: collision samples (:345-409) ----
    if (do_colli) {
      for (int s0 = 0; s0 < nseg; s0 += SPW * NW) {   // block-uniform trip count
        const int S = s0 + wave * SPW + slot;
        const bool seg_ok = (slot < SPW) & (S < nseg);
        // idle lanes shadow a live segment (their own wavefront's first one when it has
        // any) so that they compute on finite data; their results are never read
        const int s_first = s0 + wave * SPW;
        const int sc = seg_ok ? S : (s_first < nseg ? s_first : 0);
        R acc[kRedVals];
#pragma unroll
        for (int v = 0; v < kRedVals; ++v) acc[v] = (R)0;
        const R Tseg = Ts[sc];
        const R dt = dts[sc];
        const R wdt = wc * dt;
        const bool tiny_T = Tseg < (R)0.0301;
        const bool any_tiny = __ballot(tiny_T) != 0ull;   // scalar: the replay below is skipped by a uniform branch
        int coff = sc * 18;
        if constexpr (kIsF32<R> && (SPL % 2 == 0)) {
          // packed fp32: samples jj and jj+1 of this lane together
          f2 acc2[kRedVals];
#pragma unroll
          for (int v = 0; v < kRedVals; ++v) acc2[v] = (f2){0.0f, 0.0f};
#pragma unroll GTOP_SAMPLE_UNROLL
          for (int jj = 0; jj < SPL; jj += 2) {
            asm volatile("" : "+v"(coff));
            const float *cq = reinterpret_cast<const float *>(coef) + coff;
            const int si = li + jj * LPS;
            f2 t = {(float)si * (float)dt + 1e-3f, (float)(si + LPS) * (float)dt + 1e-3f};
            if (any_tiny && tiny_T) {   // rare (the first test is wave-uniform): exact replay of `t += dt`
              float ta = 1e-3f;
              for (int i = 0; i < si; ++i) ta += (float)dt;
              float tb = ta;
              for (int i = 0; i < LPS; ++i) tb += (float)dt;
              t = (f2){ta, tb};
            }
            const bool liveA = seg_ok & (t.x < (float)Tseg), liveB = seg_ok & (t.y < (float)Tseg);
            sample_pair_f32<DYN, WIDE>(reinterpret_cast<const GtopKernelArgs<float> &>(a), cq, t, liveA, liveB,
                                 (float)wdt, (float)dt, acc2);
          }
#pragma unroll
          for (int v = 0; v < kRedVals; ++v) acc[v] = (R)(acc2[v].x + acc2[v].y);
        } else
#pragma unroll kUnroll
        for (int j0 = 0; j0 < SPL; j0 += CH) {
          // Stage A, CH samples: sample time, position/velocity polynomials, distance-field
          // index arithmetic and the corner loads.  With CH > 1 (the latency
          // regime: one wavefront per SIMD, nothing else to hide a miss behind) all
          // CH lookups are in flight before the first is consumed.
          R ts[CH], vels[CH][3], accs[CH][3];
          bool lives[CH];
          SdfTap<R> taps[CH];
#pragma unroll
          for (int c = 0; c < CH; ++c) {
            // the 18 coefficients are re-read from LDS for every sample (broadcast
            // reads) instead of living in 36 VGPRs across the loop; the empty asm
            // keeps the compiler from hoisting them back out.
            asm volatile("" : "+v"(coff));
            const R *cq = coef + coff;
            const int si = li + (j0 + c) * LPS;                           // sample index 0..29
            R t = (R)si * dt + (R)1e-3;
            if (any_tiny) {   // wave-uniform, rare: exact replay of `t += dt`
              if (tiny_T) {
                t = (R)1e-3;
                for (int i = 0; i < si; ++i) t += dt;
              }
            }
            ts[c] = t;
            lives[c] = seg_ok & (t < Tseg);   // the loop condition of :353
            const R t2 = t * t, t3 = t2 * t, t4 = t2 * t2, t5 = t4 * t;
            const R d2 = (R)2 * t, d3 = (R)3 * t2, d4 = (R)4 * t3, d5 = (R)5 * t4;   // d/dt of the powers
            R pos[3];
#pragma unroll
            for (int k = 0; k < 3; ++k) {
              const R *q = cq + 6 * k;
              // :457-465 / :477-485 (sums in the reference's order), then the float round trip
              pos[k] = round_through_float(q[0] + q[1] * t + q[2] * t2 + q[3] * t3 + q[4] * t4 + q[5] * t5);
              vels[c][k] = round_through_float(q[1] + q[2] * d2 + q[3] * d3 + q[4] * d4 + q[5] * d5);
              if (DYN)  // :497-502
                accs[c][k] = round_through_float((R)2 * q[2] + (R)6 * q[3] * t + (R)12 * q[4] * t2 + (R)20 * q[5] * t3);
            }
            taps[c] = sdf_issue<R, WIDE>(a, mapbox, pos[0], pos[1], pos[2]);   // :363
          }
          if constexpr (CH > 1) __builtin_amdgcn_sched_barrier(0);   // keep every load of stage A above stage B
#ifdef GTOP_STAMPS
          if (j0 == 0 && s0 == 0) {
            GTOP_STAMP(8);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            GTOP_STAMP(9);
          }
#endif
          // Stage B: trilinear blend, penalty, accumulation.
#pragma unroll
          for (int c = 0; c < CH; ++c) {
          const R t = ts[c];
          const bool live = lives[c];
          const R *vel = vels[c], *acc3 = accs[c];
          const R t2 = t * t, t3 = t2 * t, t4 = t2 * t2, t5 = t4 * t;
          const R vn = speed_sqrt(vel[0] * vel[0] + vel[1] * vel[1] + vel[2] * vel[2]) + (R)1e-5;  // :358
          const R ivn = quick_rcp(vn);
          R g3[3];
          bool is_out;
          const R dist = sdf_blend(taps[c], g3[0], g3[1], g3[2], is_out);   // g3 per voxel, not per metre
          // samples past the loop bound of :353 and idle lanes contribute nothing:
          // every term below carries a factor e
          const R e = live ? penalty_exp((pen_d0 - dist) * pen_inv_r, expk) : (R)0;   // exp(-(d - d0)/r)
          const R cd = pen_alpha * e;                  // :509
          const R gd = pen_gd * e;                     // :514
          R csum = wdt * (cd * vn);                    // :373, weighted as in :417-418
          // g_colli.row(k) += (gd*grad(k)*cd*vn * T*Ldp + cd*(vel(k)/vn) * T*V*Ldp) * dt   (:376-381)
          const R f1 = is_out ? (R)0 : (wdt * a.res_inv) * (gd * cd * vn), f2 = wdt * (cd * ivn);   // 1/res: g3's unit
          R w1[3], w2[3], w3[3];
#pragma unroll
          for (int k = 0; k < 3; ++k) {
            w1[k] = f1 * g3[k];
            w2[k] = f2 * vel[k];
            w3[k] = (R)0;
          }
          if (DYN && a.step == 2) {
            // the block commented out at :383-407, formulas :517-535.  cv/ca
            // in the gradient are the values left by the LAST axis of the
            // cost loop, and there is no sign(v) factor — both as written.
            R cv = (R)0, ca = (R)0;
#pragma unroll
            for (int k = 0; k < 3; ++k) {
              cv = a.alpha_v * gexp((gabs(vel[k]) - a.v0) / a.r_v);
              ca = a.alpha_a * gexp((gabs(acc3[k]) - a.a0) / a.r_a);
              csum += (cv + ca) * vn * dt;       // wv = wa = 1 (:412)
            }
#pragma unroll
            for (int k = 0; k < 3; ++k) {
              const R gv = (a.alpha_v / a.r_v) * gexp((gabs(vel[k]) - a.v0) / a.r_v);
              const R ga = (a.alpha_a / a.r_a) * gexp((gabs(acc3[k]) - a.a0) / a.r_a);
              w2[k] += (gv * vn + cv * (vel[k] * ivn) + ca * (vel[k] * ivn)) * dt;
              w3[k] = (ga * vn) * dt;            // on T*V*V
            }
          }
          if (DYN && !live) {
            csum = (R)0;
#pragma unroll
            for (int k = 0; k < 3; ++k) w2[k] = w3[k] = (R)0;
          }
          // T = [1,t,..,t^5] (:544-551); T*V = [0,1,2t,3t^2,4t^3,5t^4]; T*V*V = [0,0,2,6t,12t^2,20t^3]
          const R d2 = (R)2 * t, d3 = (R)3 * t2, d4 = (R)4 * t3, d5 = (R)5 * t4;
#pragma unroll
          for (int k = 0; k < 3; ++k) {
            R *ak = acc + 6 * k;
            // two fused multiply-adds per entry
            ak[0] += w1[k];
            ak[1] = gfma(w1[k], t, ak[1] + w2[k]);
            ak[2] = gfma(w1[k], t2, gfma(w2[k], d2, ak[2]));
            ak[3] = gfma(w1[k], t3, gfma(w2[k], d3, ak[3]));
            ak[4] = gfma(w1[k], t4, gfma(w2[k], d4, ak[4]));
            ak[5] = gfma(w1[k], t5, gfma(w2[k], d5, ak[5]));
            if (DYN) {
              ak[2] += w3[k] * (R)2;
              ak[3] += w3[k] * (R)6 * t;
              ak[4] += w3[k] * (R)12 * t2;
              ak[5] += w3[k] * (R)20 * t3;
            }
          }
          acc[18] += csum;
          }
#ifdef GTOP_STAMPS
          if (j0 == 0 && s0 == 0) {
            asm volatile("" ::"v"(acc[0]), "v"(acc[5]), "v"(acc[11]), "v"(acc[17]), "v"(acc[18]));
            GTOP_STAMP(10);
          }
#endif
        }
        // A_s^-T on this lane's 18 accumulators (coefficient space -> [p0,pT,v0,vT,a0,aT]
        // per axis), so that the reduction below already yields what the free
        // variables gather; the map is linear, so it commutes with the sums.
        {
          const R T = Tseg, T2 = T * T, iT = fast_rcp(T);
          const R iT3 = iT * iT * iT, iT4 = iT3 * iT, iT5 = iT4 * iT;
#pragma unroll
          for (int k = 0; k < 3; ++k) {
            R *g = acc + 6 * k;
            const R H3 = g[3] * iT3, H4 = g[4] * iT4, H5 = g[5] * iT5;
            const R ap = (R)10 * H3 - (R)15 * H4 + (R)6 * H5;
            const R o0 = g[0] - ap;
            const R o2 = g[1] + T * ((R)-6 * H3 + (R)8 * H4 - (R)3 * H5);
            const R o3 = T * ((R)-4 * H3 + (R)7 * H4 - (R)3 * H5);
            const R o4 = (R)0.5 * g[2] + T2 * ((R)-1.5 * H3 + (R)1.5 * H4 - (R)0.5 * H5);
            const R o5 = T2 * ((R)0.5 * H3 - H4 + (R)0.5 * H5);
            g[0] = o0; g[1] = ap; g[2] = o2; g[3] = o3; g[4] = o4; g[5] = o5;
          }
        }
#ifdef GTOP_STAMPS   // pin the sample arithmetic in front of the stamp
        asm volatile("" ::"v"(acc[0]), "v"(acc[5]), "v"(acc[11]), "v"(acc[17]), "v"(acc[18]));
#endif
        GTOP_STAMP(3);
        if (LPS == 1) {
          // one lane owns the whole segment: no cross-lane reduction
          if (seg_ok) {
#pragma unroll
            for (int v = 0; v < 18; ++v) gseg[S * 18 + v] = Gs[S * 18 + v] + acc[v];
            ccol[S] = acc[18];
          }
        } else {
          // transpose-reduce over the LPS lanes of each segment through LDS, kRedChunk
          // of the 19 values at a time (tile = kRedChunk rows of kRedStride elements)
#pragma unroll
          for (int c0 = 0; c0 < kRedVals; c0 += kRedChunk) {
            const int cn = (kRedVals - c0) < kRedChunk ? (kRedVals - c0) : kRedChunk;
#pragma unroll
            for (int v = 0; v < kRedChunk; ++v)
              if ((v < cn) & (lane < LPS * SPW)) myred[v * kRedStride + lane] = acc[c0 + v];
            // The tile belongs to this wavefront alone (writers and readers are its own
            // lanes), and a wavefront's LDS operations execute in order: no workgroup
            // barrier, only a compiler-level ordering point.
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
#ifdef GTOP_STAMPS
            if (s0 == 0 && c0 == 0) GTOP_STAMP(11);
#endif
            // each lane owns up to kRdr (segment slot, value) sums; all tile reads are
            // issued before any result is stored (one LDS round trip, not kRdr)
            constexpr int kRdr = (SPW * kRedChunk + 63) / 64;
            R sums[kRdr];
#pragma unroll
            for (int u = 0; u < kRdr; ++u) {
              const int r = lane + 64 * u;
              const int rs = r / cn, v = r - rs * cn;   // (segment slot, value in chunk)
              const int Sr = s0 + wave * SPW + rs;
              const bool ok = (r < SPW * cn) & (Sr < nseg);
              const R *col = myred + (ok ? v * kRedStride + rs * LPS : 0);
              const R jerk = (ok & (c0 + v < 18)) ? Gs[Sr * 18 + c0 + v] : (R)0;
              sums[u] = jerk + tree_sum<R, LPS>(col);
            }
#ifdef GTOP_STAMPS
            if (s0 == 0 && c0 == 0) { asm volatile("" ::"v"(sums[0])); GTOP_STAMP(12); }
#endif
#pragma unroll
            for (int u = 0; u < kRdr; ++u) {
              const int r = lane + 64 * u;
              const int rs = r / cn, v = r - rs * cn;
              const int Sr = s0 + wave * SPW + rs;
              if ((r < SPW * cn) & (Sr < nseg)) {
                if (c0 + v < 18) gseg[Sr * 18 + c0 + v] = sums[u];
                else ccol[Sr] = sums[u];
              }
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");   // tile is rewritten by the next chunk / pass
            __builtin_amdgcn_wave_barrier();
          }
        }
        if constexpr (FIXED) break;   // TPBC m <= SPW segments: one pass
      }
    } else {
      for (int q = tid; q < 18 * nseg; q += nthr) gseg[q] = Gs[q];   // |wc| < 1e-4: no collision term (:346)
    }
    __syncthreads();   // gseg / ccol of every wavefront are complete
    GTOP_STAMP(4);

    GTOP_STAMP(5);

    // ---- phase 4: gather to the free variables, +1e-5 (:425-432); cost (:417-418) ----
    {
      R *gb = a.grad + (size_t)b0 * n;
      R *gl = Gs;    // [TPB][n] gradient copy for the fused optimizer step (Gs is dead after phase 3)
      R *fc = dts;   // [TPB]    cost copy (dts is dead after phase 2)
      for (int q = tid; q < ntraj * n; q += nthr) {
        int tl = 0, i = q;
        if constexpr (TPBC == 2) {
          tl = i >= n;
          i -= tl * n;
        } else if constexpr (!ONE) {
          while (i >= n) { i -= n; ++tl; }
        }
        const int axis = (i >= ndp) + (i >= 2 * ndp), c = i - axis * ndp;
        const int wpt = c / 3 + 1, der = c - 3 * (wpt - 1);   // interior waypoint 1..m-1
        const R *gs = gseg + tl * m * 18;
        const R v = gs[((wpt - 1) * 3 + axis) * 6 + 2 * der + 1] +   // end of segment wpt-1
                    gs[(wpt * 3 + axis) * 6 + 2 * der];              // start of segment wpt
        if (MMA) gl[q] = v + (R)1e-5;   // consumed by the update below; nothing leaves the chip
        else gb[q] = v + (R)1e-5;
        if constexpr (ONE && 9 * (SPW - 1) <= 64) break;   // n = 9(m-1) <= 64 free variables: one trip
      }
#ifdef GTOP_STAMPS
      GTOP_STAMP(13);
#endif
      for (int tl = wave; tl < ntraj; tl += NW) {   // one wavefront reduction per trajectory
        R part = (R)0;
        if constexpr (ONE) {
          if (lane < 3 * m) part = ws * csm[lane];
          if (lane < m) part += ccol[lane];
        } else {
          for (int i = lane; i < 3 * m; i += 64) part += ws * csm[tl * 3 * m + i];
          for (int i = lane; i < m; i += 64) part += ccol[tl * m + i];
        }
        part = wave_sum(part);
        if (lane == 0) {
          if (MMA) fc[tl] = part + (R)1e-3;
          else a.cost[b0 + tl] = part + (R)1e-3;
        }
      }
      if constexpr (MMA) {
        __syncthreads();   // gl / fc complete
        for (int tl = wave; tl < ntraj; tl += NW)
          gtop_mma_update_trajectory(st, b0 + tl, n, lane, (double)fc[tl], reinterpret_cast<const double *>(gl) + tl * n);
      }
    }
    GTOP_STAMP(6);
    if constexpr (FIXED && !MMA) break;   // nothing follows: no group, no pass
    __syncthreads();   // LDS is reused by the next pass / the next group of this block
    }
    if constexpr (FIXED) break;   // the launcher gives every group its own workgroup
  }
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
// Same arithmetic as gtop_eval_kernel sample for sample; only the order of the
// final sums differs (measured against the oracle: <= 1e-12).
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

// MINW = wavefronts per SIMD the register budget must leave room for: 2 in the latency regime (constants pinned
// in VGPRs, 232 of them), 3 for batches that can fill a third (no pins; 168-VGPR budget).
// MM = GtopMmaState (fp64, NT = 1): the batched CCSA-MMA driver's loop around the evaluation, all st.iters
// evaluations of the trajectory in this one launch — evaluate at st.xcur, update (gtop_mma_update_trajectory: accept /
// reject, asymptotes, stop rules, next trial point), evaluate again; cost and gradient never leave the chip.  One
// wavefront owns the trajectory, so the loop needs no barrier at all.  MM = GtopNoMma: a plain evaluation.
struct GtopNoMma {};
// register budget (wavefronts per SIMD) of a variant: MINW, except that the optimizer loop on six samples per lane
// keeps the one-sample-at-a-time structure of MINW = 3 on the two-wavefront budget (its update needs the room)
template <typename MM, int SPL, int MINW>
constexpr int gtop_wave_budget() { return (!std::is_same<MM, GtopNoMma>::value && SPL == 6) ? 2 : MINW; }

template <typename R, bool WIDE, int SPL, int NT, bool COLLI, int MINW, typename MM = GtopNoMma>
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(gtop_wave_budget<MM, SPL, MINW>())))
gtop_eval_wave_kernel(const R *__restrict__ arg_x, const R *__restrict__ arg_Df, const R *__restrict__ arg_T,
                      const R *__restrict__ arg_sdf, int arg_B, int arg_m, int arg_t_stride, int arg_nx, int arg_ny,
                      int arg_nz, const GtopKernelArgs<R> arg_rest, const GtopWaveConsts<R> K, const MM st) {
  constexpr bool MMA = !std::is_same<MM, GtopNoMma>::value;
  static_assert(!MMA || (NT == 1 && sizeof(R) == 8), "the optimizer loop: fp64, one trajectory per wavefront");
  GtopKernelArgs<R> a = arg_rest;
  a.x = arg_x; a.Df = arg_Df; a.T = arg_T; a.sdf = arg_sdf;
  a.B = arg_B; a.m = arg_m; a.t_stride = arg_t_stride;
  a.nx = arg_nx; a.ny = arg_ny; a.nz = arg_nz;
  constexpr int LPS = kSamples / SPL;   // lanes per segment
  constexpr int SPW = 64 / LPS;         // segment slots per wavefront
  constexpr int kStride = red_stride(SPL);
  constexpr int kMV = SPL == 6 ? 128 : 64;   // rows of the optimizer loop's LDS vectors: n <= 45 resp. 99 variables
  static_assert(SPL == 3 || SPL == 6, "10 or 5 lanes per segment");
  static_assert(SPL == 3 || MINW >= 3, "six samples per lane: one (pair) at a time only");
  static_assert(!MMA || SPL == 3 || !WIDE, "the six-samples-per-lane optimizer loop: 32-bit field offsets");
  static_assert(NT == 1 || NT == 2, "one or two trajectories per wavefront");
  extern __shared__ __align__(16) unsigned char smem_raw[];
  R *tile = reinterpret_cast<R *>(smem_raw);   // [19][kStride] (+ [kRounds*64] gradient for the optimizer update)
  GTOP_STAMP(0);
  GTOP_STAMP_HWID();
  const int lane = threadIdx.x;
  const int m = a.m, ndp = 3 * m - 3, n = 3 * ndp;
  __builtin_assume(m >= 2 && NT * m <= SPW);
  __builtin_assume(lane >= 0 && lane < 64);
  GTOP_STAMP(1);

  // XCD-aware order (see gtop_eval_kernel): XCD x gets the x-th contiguous eighth of the batch
  const int ngroups = (a.B + NT - 1) / NT;
  const int per_xcd = (ngroups + 7) >> 3;
  // The grid is 8*per_xcd workgroups; the up to 7 beyond the batch take no early exit (a branch here would
  // split the kernel-argument loads into two dependent round trips): they shadow the last group with every
  // lane idle and every store predicated off.
  const int grp_raw = ((int)blockIdx.x & 7) * per_xcd + ((int)blockIdx.x >> 3);
  const bool grp_ok = grp_raw < ngroups;
  const int grp = grp_ok ? grp_raw : ngroups - 1;
  const int b0 = grp * NT;

  const int slot = lane / LPS, li = lane - slot * LPS;
  int tl = 0, s = slot;
  if constexpr (NT == 2) {
    tl = s >= m;
    s -= tl * m;
  }
  const bool seg_ok = grp_ok & (slot < NT * m) & (b0 + tl < a.B);
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
  [[maybe_unused]] double *mv = nullptr;   // [8][kMV]: x, xcur, xprev, xprevprev, dfdx, sigma, lb, ub; then Df, T
  [[maybe_unused]] GtopMmaVecs mvecs = {};
  [[maybe_unused]] GtopMmaScalars msc = {};
  [[maybe_unused]] bool mma_live = false;
  if constexpr (MMA) {
    mv = reinterpret_cast<double *>(tile) + kRedVals * kStride + 128;
    // (the evaluation reads its inputs from here too — the trial point, and Df and T staged once behind the
    // vectors — so a pass has no global load but the distance-field corners, and no global store at all)
    mvecs = GtopMmaVecs{mv, mv + kMV, mv + 2 * kMV, mv + 3 * kMV, mv + 4 * kMV, mv + 5 * kMV, mv + 6 * kMV, mv + 7 * kMV,
                        mv + kMV};
    mma_live = grp_ok;
    if (mma_live) {
      const size_t o = (size_t)b0 * n;
      if (lane < 18) mv[8 * kMV + lane] = a.Df[(size_t)b0 * 18 + lane];
      if (lane < m) mv[8 * kMV + 18 + lane] = a.T[(size_t)b0 * a.t_stride + lane];
      if (st.x0_init) {   // (uniform) a fresh problem: mma_init_kernel's arithmetic, straight into LDS
        msc = GtopMmaScalars{1.0, 0.0, 0.0, 0.0, 0.0, 0, 0, 0};
        for (int j = lane; j < n; j += 64) {
          const double lo = st.lb[o + j], hi = st.ub[o + j];
          double v = st.x0_init[o + j];
          v = v < lo ? lo : (v > hi ? hi : v);   // nlopt clamps the start into the box
          mv[j] = v; mv[kMV + j] = v; mv[2 * kMV + j] = v; mv[3 * kMV + j] = v;
          mv[4 * kMV + j] = 0.0;
          mv[5 * kMV + j] = (isinf(lo) || isinf(hi)) ? 1.0 : 0.5 * (hi - lo);
          mv[6 * kMV + j] = lo;
          mv[7 * kMV + j] = hi;
        }
      } else {
      msc = gtop_mma_load_scalars(st, b0);
      for (int j = lane; j < n; j += 64) {
        mv[j] = st.x[o + j];
        mv[kMV + j] = st.xcur[o + j];
        mv[2 * kMV + j] = st.xprev[o + j];
        mv[3 * kMV + j] = st.xprevprev[o + j];
        mv[4 * kMV + j] = st.dfdx[o + j];
        mv[5 * kMV + j] = st.sigma[o + j];
        mv[6 * kMV + j] = st.lb[o + j];
        mv[7 * kMV + j] = st.ub[o + j];
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
    if (!mma_live || msc.state >= 3) break;
    if (st.max_ticks > 0 && pass > 0 && (long long)(wall_clock64() - t_launch) > st.max_ticks) {
      msc.state = GTOP_MMA_MAXTIME_REACHED;
      break;
    }
  }
  // ---- inputs: two waypoints' (p, v, a) per axis and T_s, straight from HBM/L2 (the optimizer loop: from LDS) ----
  // derivative vector layout (src/qp_generator.cpp:363-387): start | end | waypoint 1 | ... | waypoint m-1
  const R *xb = xsrc + (size_t)b0 * n + tl * n;      // this lane's trajectory (b0: wave-uniform)
  const R *dfb = a.Df + (size_t)b0 * 18 + tl * 18;
  const R *Tb = a.T + (size_t)b0 * a.t_stride + tl * a.t_stride;
  if constexpr (MMA) {
    xb = reinterpret_cast<const R *>(mv + kMV);
    dfb = reinterpret_cast<const R *>(mv + 8 * kMV);
    Tb = reinterpret_cast<const R *>(mv + 8 * kMV + 18);
  }
  const R T = Tb[s];
  // axis 0: the (p, v, a) triple at the segment's start and at its end; the other axes are one per-lane stride
  // further (6 within Df, 3m-3 within x: :182-187)
  const bool first = s == 0, last = s + 1 == m;
  const R *p0 = first ? dfb : xb + 3 * (s - 1);
  const R *p1 = last ? dfb + 3 : xb + 3 * s;
  const int st0 = first ? 6 : ndp, st1 = last ? 6 : ndp;
  R w0[3][3], w1[3][3];
#pragma unroll
  for (int k = 0; k < 3; ++k) {
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      w0[k][i] = p0[k * st0 + i];
      w1[k][i] = p1[k * st1 + i];
    }
  }

  // Where this lane's free variable(s) will be summed from at the very end (index arithmetic done here, while
  // the inputs are on their way): free variable = end of segment wpt-1 (entry 2 der + 1) + start of segment
  // wpt (entry 2 der)  (:425-432); tile[v][lane] holds entry v of lane's segment.
  constexpr int kRounds = (NT * 9 * (SPW / NT - 1) + 63) / 64;
  int offA[kRounds], offB[kRounds];
  bool okq[kRounds];
#pragma unroll
  for (int r = 0; r < kRounds; ++r) {
    const int qi = lane + 64 * r;
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
  const bool cost_lane = (cs >= 0) & (csi < m);
  if (cost_lane) offA[kRounds - 1] = 18 * kStride + (ct * m + csi) * LPS;
  const R ws = a.ws;   // the launcher has applied :412-415 (step 1 -> ws = 0): `step` is not read here
  const R wc = a.wc;
  ExpConsts expk;
  R pen_d0 = a.d0, pen_inv_r = a.inv_r, pen_alpha = a.alpha, pen_gd = -a.alpha_over_r;   // (:507-515)
  MapBox<R> mapbox = {{a.lo[0], a.lo[1], a.lo[2]}, {a.hi[0], a.hi[1], a.hi[2]},
                      {a.origin[0], a.origin[1], a.origin[2]}, (R)0.5 * a.res, a.res_inv};
  if constexpr (COLLI && !kIsF32<R> && MINW <= 2 && !MMA) {   // (the optimizer loop has no registers to spare)
    // the exp constants only: with the map box pinned as well (12 more VGPRs) the body spills two registers since
    // the hand-issued loads hold all 12 corner pairs at once — 4.45 against 4.23 us
    expk.pin();
  }
#ifdef GTOP_STAMPS
  GTOP_STAMP(2);   // inputs requested
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  GTOP_STAMP(3);   // inputs landed
#endif
  // ---- coefficients c = A_s^-1 d (closed form; rows of A_s: src/qp_generator.cpp:185-195) ----
  const R T2 = T * T, T3 = T2 * T, T4 = T2 * T2, T5 = T4 * T;
  const R iT = fast_rcp(T), iT3 = iT * iT * iT, iT4 = iT3 * iT, iT5 = iT4 * iT;
  // :351, dt = T/30.  The quotient proper (a dozen instructions) is only needed where the sample COUNT hangs on
  // the accumulated sample time (tiny T, below); everywhere else T * (1/30) is the same to an ulp.
  const R dt = T * (R)(1.0 / 30.0);
  const R wdt = wc * dt;
  R q[3][6];
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const R p0 = w0[k][0], v0 = w0[k][1], a0 = w0[k][2];
    const R pT = w1[k][0], vT = w1[k][1], aT = w1[k][2];
    const R P = pT - p0 - v0 * T - (R)0.5 * a0 * T2;
    const R V = (vT - v0 - a0 * T) * T;
    const R A = (aT - a0) * T2;
    q[k][0] = p0; q[k][1] = v0; q[k][2] = (R)0.5 * a0;
    q[k][3] = (K.k10 * P - (R)4 * V + (R)0.5 * A) * iT3;
    q[k][4] = (K.k7 * V - K.k15 * P - A) * iT4;
    q[k][5] = (K.k6 * P - K.k3 * V + (R)0.5 * A) * iT5;
  }
  // The jerk term, as the START value of the accumulators (a lambda: it is placed where the distance-field
  // loads are in flight, see below).  Jerk Hessian Q_s: src/qp_generator.cpp:226-234, i,j in {3,4,5}.
  R acc[kRedVals];
  auto jerk_init = [&]() {
    const R Q33 = K.q36 * T, Q34 = K.q72 * T2, Q35 = K.q120 * T3, Q44 = K.q192 * T3, Q45 = K.q360 * T4,
            Q55 = K.q720 * T5;
    // lane 0 of a segment carries the segment's jerk term into the sums
    const R wj = (seg_ok & (li == 0)) ? ws : (R)0;
    const R wj2 = wj + wj;
    R jc = (R)0;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const R c3 = q[k][3], c4 = q[k][4], c5 = q[k][5];
      const R q3 = Q33 * c3 + Q34 * c4 + Q35 * c5;
      const R q4 = Q34 * c3 + Q44 * c4 + Q45 * c5;
      const R q5 = Q35 * c3 + Q45 * c4 + Q55 * c5;
      jc += c3 * q3 + c4 * q4 + c5 * q5;   // c'Qc: this (s,k)'s share of d'Rd (:326-327)
      // ws * 2Qc: share of ws*(2Rfp'df + 2Rpp dp) (:330-336) in coefficient space (entries 0..2 are zero)
      acc[6 * k + 0] = (R)0; acc[6 * k + 1] = (R)0; acc[6 * k + 2] = (R)0;
      acc[6 * k + 3] = wj2 * q3; acc[6 * k + 4] = wj2 * q4; acc[6 * k + 5] = wj2 * q5;
    }
    acc[18] = wj * jc;
  };

  // ---- collision samples (:345-409); sample index = li + j*LPS ----
  if constexpr (COLLI) {
    // Sample times (:353, see gtop_eval_kernel): t_i = 1e-3 + i*dt; segments with T < 0.0301 (where the
    // sample COUNT depends on the accumulated value) replay the reference's addition chain.
    // Every term of a sample carries the factor alpha * wc * dt (cd of :509 times the weights of :373/:417);
    // a sample past the loop bound of :353 contributes nothing, i.e. has that factor zero.  With T >= 0.0301
    // all 30 samples are inside the bound, so only the replay path ever has to clear it.
    // Sample times (:353, see gtop_eval_kernel): t_i = 1e-3 + i*dt; segments with T < 0.0301 (where the
    // sample COUNT depends on the accumulated value) replay the reference's addition chain.
    const bool tiny_T = T < (R)0.0301;
    const bool any_tiny = __ballot(tiny_T) != 0ull;   // wave-uniform, rare
    auto sample_time = [&](int j, R &t, R &awj) {
      t = (R)(li + j * LPS) * dt + (R)1e-3;
      awj = pen_alpha * wdt;
      if (any_tiny) {
        if (tiny_T) {
          const R dtq = T / (R)30.0;   // the quotient proper, :351
          t = (R)1e-3;
          for (int i = 0; i < li + j * LPS; ++i) t += dtq;
          awj = (t < T) ? pen_alpha * (wc * dtq) : (R)0;
        }
      }
    };
    constexpr int NTS = (MINW <= 2) ? SPL : 1;   // latency regime: all sample times before the first load
    R ts[NTS], aw[NTS];
    if constexpr (MINW <= 2) {
#pragma unroll
      for (int j = 0; j < SPL; ++j) sample_time(j, ts[j], aw[j]);
    }
    // The samples of a lane go through two stages, CH at a time.  Latency regime (MINW = 2): CH = SPL, all 12 corner
    // loads of the lane in flight at once, the jerk term and the speeds computed behind them, the order pinned by
    // scheduling barriers.  Throughput regime (MINW = 3): one sample at a time — other wavefronts cover the loads,
    // and only one sample's corners are live (the 168-VGPR budget of a third wavefront).
    constexpr int CH = (MINW <= 2) ? SPL : 1;
    constexpr int kUnrollJ = SPL <= 3 ? SPL : 1;   // six samples per lane stay a loop (code size)
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
        R tA, tB, awA, awB;
        sample_time(jj, tA, awA);
        sample_time(jj + 1, tB, awB);
        sample_pair_f32<false, WIDE>(reinterpret_cast<const GtopKernelArgs<float> &>(a), cq, (f2){(float)tA, (float)tB},
                                     awA != (R)0, awB != (R)0, (float)wdt, (float)dt, acc2);
      }
#pragma unroll
      for (int v = 0; v < kRedVals; ++v) acc[v] = (R)(acc2[v].x + acc2[v].y);
    } else {
    if constexpr (MINW > 2) jerk_init();
#pragma unroll kUnrollJ
    for (int j0 = 0; j0 < SPL; j0 += CH) {
      // stage A: positions, index arithmetic, corner loads
      R vels[CH][3];
      SdfTap<R> taps[CH];
      gtop_d2 raw[ASMLD ? CH : 1][4];
      if constexpr (MINW > 2) sample_time(j0, ts[0], aw[0]);
#pragma unroll
      for (int c = 0; c < CH; ++c) {
        const R t = ts[MINW <= 2 ? j0 + c : 0];
        const R t2 = t * t, t3 = t2 * t, t4 = t2 * t2, t5 = t4 * t;
        const R d2 = (R)2 * t, d3 = K.k3 * t2, d4 = (R)4 * t3, d5 = K.k5 * t4;   // d/dt of the powers
        R pos[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
          // :457-465 / :477-485 (sums in the reference's order), then the float round trip
          pos[k] = round_through_float(q[k][0] + q[k][1] * t + q[k][2] * t2 + q[k][3] * t3 + q[k][4] * t4 + q[k][5] * t5);
          vels[c][k] = round_through_float(q[k][1] + q[k][2] * d2 + q[k][3] * d3 + q[k][4] * d4 + q[k][5] * d5);
        }
        if constexpr (ASMLD) taps[c] = sdf_issue_asm(a, mapbox, pos[0], pos[1], pos[2], raw[c]);
        else taps[c] = sdf_issue<R, WIDE>(a, mapbox, pos[0], pos[1], pos[2]);   // :363
      }
      if (j0 == 0) GTOP_STAMP(4);   // corner loads issued
      if constexpr (CH == SPL) GTOP_PHASE_FENCE();   // every corner load is issued above this line ...
      // ... and what does not need them runs while they are in flight: the jerk term and the speeds
      if constexpr (MINW <= 2) jerk_init();
      R vns[CH], ivns[CH];
#pragma unroll
      for (int c = 0; c < CH; ++c) {
        const R *vel = vels[c];
        vns[c] = speed_sqrt(vel[0] * vel[0] + vel[1] * vel[1] + vel[2] * vel[2]) + K.eps;   // :358
        ivns[c] = quick_rcp(vns[c]);
      }
#ifdef GTOP_STAMPS
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
          taps[c].p00.x = raw[c][0].x; taps[c].p00.y = raw[c][0].y;
          taps[c].p01.x = raw[c][1].x; taps[c].p01.y = raw[c][1].y;
          taps[c].p10.x = raw[c][2].x; taps[c].p10.y = raw[c][2].y;
          taps[c].p11.x = raw[c][3].x; taps[c].p11.y = raw[c][3].y;
        }
        const R t = ts[MINW <= 2 ? j0 + c : 0];
        const R *vel = vels[c];
        const R t2 = t * t, t3 = t2 * t, t4 = t2 * t2, t5 = t4 * t;
        const R vn = vns[c], ivn = ivns[c];
        R g3[3];
        bool is_out;
        const R dist = sdf_blend(taps[c], g3[0], g3[1], g3[2], is_out);   // g3 per voxel, not per metre
        const R e = penalty_exp((pen_d0 - dist) * pen_inv_r, expk);   // exp(-(d - d0)/r)
        const R cdw = aw[MINW <= 2 ? j0 + c : 0] * e;   // wc*dt * cd, cd of :509 (idle lanes: shadow data, never read)
        const R cv = cdw * vn;
        acc[18] = gfma(cdw, vn, acc[18]);   // += cv: :373, weighted as in :417-418 (fusions are spelled out: -ffp-contract=on)
        // g_colli.row(k) += (gd*grad(k)*cd*vn * T*Ldp + cd*(vel(k)/vn) * T*V*Ldp) * dt   (:376-381); gd of :514
        const R f1 = is_out ? (R)0 : ((pen_gd * a.res_inv) * e) * cv, f2 = cdw * ivn;
        const R d2 = (R)2 * t, d3 = K.k3 * t2, d4 = (R)4 * t3, d5 = K.k5 * t4;
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
        }
      }
      if constexpr (CH != SPL) __builtin_amdgcn_sched_barrier(0);   // keep the samples apart: one sample's corners live
    }
    }
  } else {
    (void)wdt; (void)pen_d0; (void)pen_inv_r; (void)pen_alpha; (void)pen_gd;
    jerk_init();
  }
#ifdef GTOP_STAMPS
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
  if (lane < LPS * SPW) {
#pragma unroll
    for (int v = 0; v < kRedVals; ++v) tile[v * kStride + lane] = acc[v];   // (columns of idle slots are never read)
  }
  GTOP_STAMP(8);   // A^-T + tile writes issued
  GTOP_STAMP(9);
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");   // writers and readers are this wavefront's own lanes,
  __builtin_amdgcn_wave_barrier();                          // whose LDS operations execute in order
  R csum_seg = (R)0;
  R *gl = tile + kRedVals * kStride;   // [kRounds*64]: the gradient for the optimizer update (MMA only)
#pragma unroll
  for (int r = 0; r < kRounds; ++r) {
    const R sa = tree_sum<R, LPS>(tile + offA[r]), sb = tree_sum<R, LPS>(tile + offB[r]);
    if (r == kRounds - 1 && cost_lane) csum_seg = sa;
    if constexpr (MMA) {
      if (okq[r]) gl[lane + 64 * r] = (sa + sb) + K.eps;   // consumed below; nothing leaves the chip
    } else {
      if (okq[r]) a.grad[(size_t)b0 * n + lane + 64 * r] = (sa + sb) + K.eps;
    }
  }
  // ---- cost (:417-418): every term is already weighted; lanes 48.. hold the segment sums ----
  {
    R cpart = cost_lane ? csum_seg : (R)0;
    cpart += gtop_dpp_move<0x111>(cpart);   // row_shr:1
    cpart += gtop_dpp_move<0x112>(cpart);   // row_shr:2
    cpart += gtop_dpp_move<0x114>(cpart);   // row_shr:4  -> lane 55 holds lanes 48..55, lane 63 lanes 56..63
    if constexpr (NT == 2) {
      if (grp_ok & (lane == 55)) a.cost[b0] = cpart + (R)1e-3;
      if (grp_ok & (lane == 63) & (b0 + 1 < a.B)) a.cost[b0 + 1] = cpart + (R)1e-3;
    } else if constexpr (SPW > 8 && !MMA) {
      cpart += gtop_dpp_move<0x118>(cpart);   // row_shr:8 -> lane 63 holds lanes 48..63 (up to 12 segments)
      if (grp_ok & (lane == 63)) a.cost[b0] = cpart + (R)1e-3;
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
      gtop_mma_update_core(st, mvecs, msc, n, lane, fcur, reinterpret_cast<const double *>(gl));
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");   // the tile is rewritten by the next evaluation
      __builtin_amdgcn_wave_barrier();
    } else {
      if (grp_ok & (lane == 55)) a.cost[b0] = cpart + (R)1e-3;
    }
  }
  GTOP_STAMP(10);   // gradient stored (issued)
#ifdef GTOP_STAMPS
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  GTOP_STAMP(11);   // stores acknowledged
#endif
  }   // pass
  if constexpr (MMA) {
    if (mma_live) {   // the state goes home
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      const size_t o = (size_t)b0 * n;
      for (int j = lane; j < n; j += 64) {
        st.x[o + j] = mv[j];
        st.xcur[o + j] = mv[kMV + j];
        st.xprev[o + j] = mv[2 * kMV + j];
        st.xprevprev[o + j] = mv[3 * kMV + j];
        st.dfdx[o + j] = mv[4 * kMV + j];
        st.sigma[o + j] = mv[5 * kMV + j];
      }
      if (lane == 0) gtop_mma_store_scalars(st, b0, msc);
      // the results, where the caller wants them (otherwise: copies + mma_finish_kernel after this launch)
      if (st.out_x)
        for (int j = lane; j < n; j += 64) st.out_x[o + j] = mv[j];
      if (lane == 0) {
        if (st.out_minf) st.out_minf[b0] = msc.minf;
        if (st.out_code) st.out_code[b0] = msc.state >= 3 ? msc.state : GTOP_MMA_MAXEVAL_REACHED;
        if (st.out_nevals) st.out_nevals[b0] = msc.nevals;
      }
    }
  }
}

}  // namespace

size_t gtop_eval_smem_bytes(int m, int waves, int tpb, int spl, size_t elem, int red_rows) {
  const size_t MS = (size_t)tpb * m;
  const size_t elems = MS + 18 * MS * 2 + 3 * MS + MS + (size_t)waves * (red_rows > 0 ? red_rows : kRedChunkFull) * red_stride(spl);
  return elems * elem;
}

int gtop_eval_segments_per_wave(int spl) { return 64 / (kSamples / spl); }

// WIDE = false needs 24-bit row/column counts and a field below 4 GiB (corner_loads)
bool gtop_field_is_narrow(int nx, int ny, int nz, size_t elem) {
  const unsigned long long nvox = (unsigned long long)nx * ny * nz;
  return (unsigned long long)nx * ny < (1ull << 24) && nz < (1 << 24) && (nvox + 2) * elem < (1ull << 32);
}

#ifndef GTOP_WAVE_KERNEL
#define GTOP_WAVE_KERNEL 1
#endif
#define GTOP_WAVE_KERNEL_DEFAULT GTOP_WAVE_KERNEL
#ifndef GTOP_WAVE_SPL6
#define GTOP_WAVE_SPL6 1
#endif
#define GTOP_WAVE_SPL6_DEFAULT GTOP_WAVE_SPL6

// the specialised straight-line bodies: one wavefront = one whole trajectory (SPL 3 or 6) or two (SPL 6)
template <typename R>
static bool gtop_fixed_body_ok(const GtopKernelArgs<R> &args, int waves, int spl, int grid) {
  const int groups = (args.B + args.tpb - 1) / args.tpb;
  return (spl == 3 || spl == 6) && waves == 1 && args.m >= 2 &&
         args.tpb * args.m <= gtop_eval_segments_per_wave(spl) &&
         grid == 8 * ((groups + 7) / 8);   // (no grid-stride loop in those bodies)
}

template <typename R, bool DYN, bool MMA, bool WIDE>
static hipError_t launch_spl(const GtopKernelArgs<R> &args, const GtopMmaState &st, int waves, int spl, int grid,
                             size_t smem /* of the generic body */, hipStream_t stream, bool wave_ok) {
  void (*kern)(const GtopKernelArgs<R>, const GtopMmaState) = nullptr;
  const bool fixed_ok = gtop_fixed_body_ok(args, waves, spl, grid);
  // A field that outgrows the 256 MB Infinity Cache is served by HBM; a fourth wavefront per SIMD then only
  // adds L2 misses (measured at 400^3 fp64, m = 12: 47 us with the generic three-wavefront body, 52 us with
  // four, 50 us with the specialised body held at three), so the SPL = 6 specialisations are for resident fields.
  const unsigned long long field_bytes = (unsigned long long)args.nx * args.ny * args.nz * sizeof(R);
  const bool resident = field_bytes <= (256ull << 20);
  const bool one = fixed_ok && args.tpb == 1 && (spl == 3 || resident);
  const bool two = fixed_ok && args.tpb == 2 && spl == 6 && !MMA && resident;
  if (!MMA && spl == 6 && (one || two))   // their tile
    smem = gtop_eval_smem_bytes(args.m, waves, args.tpb, spl, sizeof(R), red_chunk(sizeof(R), 6, args.tpb));

#ifndef GTOP_WAVE_MINW3_FROM
#define GTOP_WAVE_MINW3_FROM 3072   // batches that put a third wavefront on a SIMD (1 024 SIMDs)
#endif

  if constexpr (GTOP_WAVE_KERNEL && !DYN) {
    // gtop_eval_wave_kernel: whole trajectories per wavefront, no workgroup barrier.  spl 3: one trajectory of up to
    // 6 segments (latency variant below GTOP_WAVE_MINW3_FROM trajectories) — with or without the optimizer loop;
    // spl 6: one trajectory of up to 12 segments, or two of up to 6 (fp32: packed sample pairs).
    // wave_ok = false (the optimizer's separate-update mode) keeps a plain evaluation on the body its fused modes
    // run: the wave kernel exactly where they run it (spl 3, one trajectory per wavefront).
    GtopKernelArgs<R> wa = args;
    if (wa.step == 1) wa.ws = (R)0;   // :412-415, applied here so that the kernel need not fetch `step`
    const bool colli = !((wa.wc < (R)0 ? -wa.wc : wa.wc) < (R)1e-4);   // :346
    // (64-bit field indices — fields past 4 GiB — cost the 168-VGPR fp64 body 14 spilled registers: those stay on
    // the two-wavefront budget)
    const bool three = args.B >= GTOP_WAVE_MINW3_FROM && !(WIDE && sizeof(R) == 8);
    // tile + the optimizer's gradient rows (+ its state: 8 vectors of 64, Df, T; gtop_eval_wave_kernel)
    const size_t wsmem = (kRedVals * red_stride(spl) + 128 + (MMA ? 8 * (spl == 6 ? 128 : 64) + 32 : 0)) * sizeof(R);
    if constexpr (MMA) {
      if constexpr (sizeof(R) == 8) {
        if (one && spl == 3) {
          // (two wavefronts per SIMD at every batch size: the update's working set spills a 168-VGPR budget)
          auto wk = colli ? gtop_eval_wave_kernel<R, WIDE, 3, 1, true, 2, GtopMmaState>
                          : gtop_eval_wave_kernel<R, WIDE, 3, 1, false, 2, GtopMmaState>;
          hipLaunchKernelGGL(wk, dim3(grid), dim3(64), wsmem, stream, wa.x, wa.Df, wa.T, wa.sdf, wa.B, wa.m, wa.t_stride,
                             wa.nx, wa.ny, wa.nz, wa, GtopWaveConsts<R>{}, st);
          return hipGetLastError();
        }
        if constexpr (!WIDE) {
          if (GTOP_WAVE_SPL6 && fixed_ok && spl == 6 && args.tpb == 1) {
            // 7 .. 12 segments: one trajectory per wavefront at five lanes per segment, one sample at a time
            auto wk = colli ? gtop_eval_wave_kernel<R, false, 6, 1, true, 3, GtopMmaState>
                            : gtop_eval_wave_kernel<R, false, 6, 1, false, 3, GtopMmaState>;
            hipLaunchKernelGGL(wk, dim3(grid), dim3(64), wsmem, stream, wa.x, wa.Df, wa.T, wa.sdf, wa.B, wa.m,
                               wa.t_stride, wa.nx, wa.ny, wa.nz, wa, GtopWaveConsts<R>{}, st);
            return hipGetLastError();
          }
        }
      }
    } else {
      constexpr int kW6 = 3;   // register budget of the spl 6 variants: wavefronts per SIMD (fp32 at 4 spills 25 VGPRs)
      void (*wk)(const R *, const R *, const R *, const R *, int, int, int, int, int, int, const GtopKernelArgs<R>,
                 const GtopWaveConsts<R>, const GtopNoMma) = nullptr;
      if (one && spl == 3) {
        wk = colli ? gtop_eval_wave_kernel<R, WIDE, 3, 1, true, 2> : gtop_eval_wave_kernel<R, WIDE, 3, 1, false, 2>;
        if (three) wk = colli ? gtop_eval_wave_kernel<R, WIDE, 3, 1, true, 3> : gtop_eval_wave_kernel<R, WIDE, 3, 1, false, 3>;
      } else if (GTOP_WAVE_SPL6 && (wave_ok || (sizeof(R) == 8 && !WIDE)) && fixed_ok && spl == 6 && args.tpb == 1) {
        // (wave_ok = false: exactly where the fused optimizer modes run their five-lanes-per-segment loop, above)
        wk = colli ? gtop_eval_wave_kernel<R, WIDE, 6, 1, true, kW6> : gtop_eval_wave_kernel<R, WIDE, 6, 1, false, kW6>;
      } else if (GTOP_WAVE_SPL6 && wave_ok && fixed_ok && spl == 6 && args.tpb == 2) {
        wk = colli ? gtop_eval_wave_kernel<R, WIDE, 6, 2, true, kW6> : gtop_eval_wave_kernel<R, WIDE, 6, 2, false, kW6>;
      }
      if (wk) {
        hipLaunchKernelGGL(wk, dim3(grid), dim3(64), wsmem, stream, wa.x, wa.Df, wa.T, wa.sdf, wa.B, wa.m, wa.t_stride,
                           wa.nx, wa.ny, wa.nz, wa, GtopWaveConsts<R>{}, GtopNoMma{});
        return hipGetLastError();
      }
    }
  }
  if constexpr (MMA) {   // the fused optimizer step is built for the geometries the auto rules pick
    switch (spl) {
      case 1: kern = gtop_eval_kernel<R, DYN, 1, true, WIDE, 0>; break;
      case 3: kern = one ? gtop_eval_kernel<R, DYN, 3, true, WIDE, 1> : gtop_eval_kernel<R, DYN, 3, true, WIDE, 0>; break;
      case 6: kern = one ? gtop_eval_kernel<R, DYN, 6, true, WIDE, 1> : gtop_eval_kernel<R, DYN, 6, true, WIDE, 0>; break;
      default: return hipErrorInvalidValue;
    }
  } else {
    switch (spl) {
      case 1: kern = gtop_eval_kernel<R, DYN, 1, false, WIDE, 0>; break;
      case 2: kern = gtop_eval_kernel<R, DYN, 2, false, WIDE, 0>; break;
      case 3: kern = one ? gtop_eval_kernel<R, DYN, 3, false, WIDE, 1> : gtop_eval_kernel<R, DYN, 3, false, WIDE, 0>; break;
      case 5: kern = gtop_eval_kernel<R, DYN, 5, false, WIDE, 0>; break;
      case 6: kern = one ? gtop_eval_kernel<R, DYN, 6, false, WIDE, 1>
                         : (two ? gtop_eval_kernel<R, DYN, 6, false, WIDE, 2> : gtop_eval_kernel<R, DYN, 6, false, WIDE, 0>);
              break;
      case 10: kern = gtop_eval_kernel<R, DYN, 10, false, WIDE, 0>; break;
      case 15: kern = gtop_eval_kernel<R, DYN, 15, false, WIDE, 0>; break;
      case 30: kern = gtop_eval_kernel<R, DYN, 30, false, WIDE, 0>; break;
      default: return hipErrorInvalidValue;
    }
  }
  if (smem > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    if (e != hipSuccess) return e;
  }
  hipLaunchKernelGGL(kern, dim3(grid), dim3(64 * waves), smem, stream, args, st);
  return hipGetLastError();
}

template <typename R, bool MMA>
static hipError_t launch_any(const GtopKernelArgs<R> &args, const GtopMmaState &st, int waves, int spl, bool dyn,
                             int max_blocks, hipStream_t stream, bool wave_ok = true) {
  if (args.B <= 0) return hipSuccess;
  const size_t smem = gtop_eval_smem_bytes(args.m, waves, args.tpb, spl, sizeof(R));
  const int groups = (args.B + args.tpb - 1) / args.tpb;
  const int vblocks = 8 * ((groups + 7) / 8);   // the kernel walks 8 XCD-contiguous ranges
  const int grid = vblocks < max_blocks ? vblocks : max_blocks;
  const bool wide = !gtop_field_is_narrow(args.nx, args.ny, args.nz, sizeof(R));
  if (wide)
    return dyn ? launch_spl<R, true, MMA, true>(args, st, waves, spl, grid, smem, stream, wave_ok)
               : launch_spl<R, false, MMA, true>(args, st, waves, spl, grid, smem, stream, wave_ok);
  return dyn ? launch_spl<R, true, MMA, false>(args, st, waves, spl, grid, smem, stream, wave_ok)
             : launch_spl<R, false, MMA, false>(args, st, waves, spl, grid, smem, stream, wave_ok);
}

template <typename R>
hipError_t gtop_launch_eval(const GtopKernelArgs<R> &args, int waves, int spl, bool dyn,
                            int max_blocks, hipStream_t stream, bool wave_kernel_ok) {
  const GtopMmaState none{};
  return launch_any<R, false>(args, none, waves, spl, dyn, max_blocks, stream, wave_kernel_ok);
}

bool gtop_eval_mma_is_wave_loop(const GtopKernelArgs<double> &args, int waves, int spl, bool dyn, int max_blocks) {
  if (args.B <= 0 || dyn || !GTOP_WAVE_KERNEL_DEFAULT) return false;
  const int groups = (args.B + args.tpb - 1) / args.tpb;
  const int vblocks = 8 * ((groups + 7) / 8);
  const int grid = vblocks < max_blocks ? vblocks : max_blocks;
  // launch_spl: `one && spl == 3`, or the six-samples-per-lane loop (32-bit field offsets only)
  if (!gtop_fixed_body_ok(args, waves, spl, grid) || args.tpb != 1) return false;
  return spl == 3 || (GTOP_WAVE_SPL6_DEFAULT && spl == 6 && gtop_field_is_narrow(args.nx, args.ny, args.nz, sizeof(double)));
}

// cost/gradient at st.xcur + the MMA update, one launch (fp64; spl 1, 3 or 6)
hipError_t gtop_launch_eval_mma(const GtopKernelArgs<double> &args, const GtopMmaState &st, int waves, int spl,
                                bool dyn, int max_blocks, hipStream_t stream) {
  return launch_any<double, true>(args, st, waves, spl, dyn, max_blocks, stream);
}

#ifdef GTOP_STAMPS
extern "C" int gtop_debug_read_stamps(unsigned long long *out /*4096*16*/) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_gtop_stamps), sizeof(unsigned long long) * 4096 * 16);
}
#endif

template hipError_t gtop_launch_eval<double>(const GtopKernelArgs<double> &, int, int, bool, int, hipStream_t, bool);
template hipError_t gtop_launch_eval<float>(const GtopKernelArgs<float> &, int, int, bool, int, hipStream_t, bool);

// ---------------------------------------------------------------------------
// fp64 -> fp32 copy of the distance field for the GTOP_F32 path
// ---------------------------------------------------------------------------
namespace {
__global__ void __launch_bounds__(256)
gtop_f64_to_f32_kernel(const double *__restrict__ src, float *__restrict__ dst, size_t nelem) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (; i < nelem; i += stride) dst[i] = (float)src[i];
}
}  // namespace

hipError_t gtop_launch_f64_to_f32(const double *src, float *dst, size_t nelem, hipStream_t stream) {
  if (nelem == 0) return hipSuccess;
  hipLaunchKernelGGL(gtop_f64_to_f32_kernel, dim3(2048), dim3(256), 0, stream, src, dst, nelem);
  return hipGetLastError();
}
