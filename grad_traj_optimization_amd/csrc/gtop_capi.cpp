// gtop_capi.cpp — implementation of the C-ABI in include/gtop.h on top of the
// gfx950 kernels.  Host-side only: owns device buffers, fills kernel
// arguments, launches.  There is deliberately no CPU code path: without a
// gfx950 device every entry point fails (GTOP_ERR_NO_DEVICE).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <vector>

#include "gtop.h"
#include "gtop_guard.h"
#include "gtop_kernels.h"

#define GTOP_ABI_VERSION 2   // round 4: gtop_update_sdf_map_window*, gtop_set_field_precisions, gtop_device_clock_*, gtop_group_gather_note, GTOP_ERR_INTERNAL

struct gtop_ctx {
  int device = 0;
  std::string err;
  hipStream_t stream = nullptr;   // used by the host-pointer entry points

  gtop_params prm{};
  bool have_params = false;

  GtopGrid grid{};
  bool have_grid = false;
  // The distance field.  sdf64 is the BOUNDARY copy, z fastest (src/sdf_map.cpp:172-173): what gtop_set_sdf uploads,
  // the ESDF builder writes, gtop_get_sdf returns and the coarse voxel query reads; owned, or borrowed from
  // gtop_set_sdf_device(GTOP_F64).  sdf32b is a borrowed fp32 field (gtop_set_sdf_device(GTOP_F32); no fp64 copy then).
  // What the lookups of every kernel read are the CORNER RECORDS derived from it (gtop_records.hip),
  // always owned: rec64 and rec32.  rec64 is rebuilt wherever the field changes.  The fp32 records of a rebuilt field:
  // gtop_update_sdf_map (host points, synchronous) defers them to the first fp32 evaluation unless one has been seen
  // on this context (`fp32_in_use`, sticky), because they are a third of the pass's writes;
  // gtop_update_sdf_map_device (asynchronous, capturable into a hipGraph) always builds them behind the fp64 ones, so
  // that a REPLAY of the captured rebuild — which never passes through this host code again — leaves both current.
  double *sdf64 = nullptr;
  const float *sdf32b = nullptr;
  bool own64 = false;
  size_t sdf_cap64 = 0;              // elements, for the owned buffer
  double *rec64 = nullptr;
  float *rec32 = nullptr;
  size_t rec_cap64 = 0, rec_cap32 = 0;   // records
  bool rec64_ok = false, rec32_ok = false;   // the records hold the current field
  bool rec32_stale = false;          // ... or must be rebuilt from sdf64 before the next fp32 use
  bool fp32_in_use = false;
  bool fp32_wanted = true;           // gtop_set_field_precisions: 0 = fp64 records only (fp32 evaluations refused)

  // ESDF construction workspace
  uint8_t *occ = nullptr;
  int *tmp1 = nullptr, *tmp2 = nullptr, *rows = nullptr;
  double *boxes = nullptr;   // moving boxes: p0 | vel | scale, nbox x 3 each
  size_t cap_boxes = 0;
  int nbox = 0;
  double *d_q = nullptr;     // host-API staging of gtop_edt_query: pos | time | dist | grad
  size_t cap_q = 0;
  double *pin = nullptr;     // pinned, device-visible host staging for small host-buffer evaluations: x | cost | grad
  double *pin_dev = nullptr; // its device address
  size_t cap_pin = 0;
  bool poll_completion = true;   // GTOP_POLL_COMPLETION=0: always wait through the stream (gtop_eval_batch)
  uint64_t poll_sentinel = 0;    // preset of the polled output slots (GTOP_POLL_SENTINEL=<hex> overrides: tests)
  double *d_pts = nullptr;
  size_t cap_occ = 0, cap_tmp1 = 0, cap_tmp2 = 0, cap_rows = 0, pts_cap = 0;
  uint8_t *win_occ = nullptr;    // gtop_update_sdf_map_window: the window's occupancy / distances as a compact grid
  double *win_dist = nullptr;
  size_t cap_win_occ = 0, cap_win_dist = 0;

  // problem set by gtop_set_problem
  int B = 0, m = 0, t_stride = 0;
  double *d_T = nullptr, *d_Df = nullptr, *d_x = nullptr, *d_cost = nullptr, *d_grad = nullptr;
  size_t cap_T = 0, cap_Df = 0, cap_x = 0, cap_grad = 0, cap_cost = 0;

  // batched optimizer workspace (gtop_optimize_*)
  double *mma_vec = nullptr;   // 6 x [B][n]
  double *mma_scal = nullptr;  // 4 x [B]
  int *mma_int = nullptr;      // 2 x [B]
  double *mma_f = nullptr, *mma_g = nullptr, *mma_lb = nullptr, *mma_ub = nullptr;
  int *mma_res = nullptr;      // nevals | code of gtop_optimize_batch_ex, 2 x [B]
  size_t cap_mma_res = 0;
  size_t cap_mma_vec = 0, cap_mma_scal = 0, cap_mma_int = 0, cap_mma_f = 0, cap_mma_g = 0, cap_mma_lb = 0,
         cap_mma_ub = 0;

  int spl = 0;     // samples per lane: 0 = auto, 3, 6, 10 or 30 (gtop_set_launch_geometry)
  int opt_dtype = GTOP_F64;   // gtop_set_optimizer_precision: the arithmetic of the evaluations inside the batched optimizer
  int fuse_mma = 2;         // optimizer: 0 separate update launch, 1 update fused into the evaluation kernel,
                            //            2 (default) the whole loop in one launch (tuning/debug knob)

  // bookkeeping of the callback (grad_traj_optimizer.cpp:284, :436, :439-447)
  int64_t iter_num = 0;
  double total_time = 0.0;
  std::vector<double> vec_cost, vec_time;
  std::chrono::steady_clock::time_point time_start = std::chrono::steady_clock::now();
};

namespace {

// error text of a failed gtop_create (there is no context to hold it yet)
thread_local std::string g_create_err;

int fail(gtop_ctx *c, int code, const std::string &msg) {
  if (c) c->err = msg;
  else g_create_err = msg;
  return code;
}

// what an exception caught at the boundary leaves behind (gtop_guard.h); must not throw itself
void note_exception(gtop_ctx *c, const char *what) noexcept {
  try {
    fail(c, GTOP_ERR_INTERNAL, std::string("exception caught at the C boundary: ") + what);
  } catch (...) {
  }
}
#define GTOP_CATCH_STATUS(c) GTOP_CATCH_WITH(note_exception, c, GTOP_ERR_INTERNAL)
#define GTOP_CATCH_HUGE(c) GTOP_CATCH_WITH(note_exception, c, HUGE_VAL)

#define HIPCHK(ctx, call)                                                              \
  do {                                                                                 \
    hipError_t e_ = (call);                                                            \
    if (e_ != hipSuccess)                                                              \
      return fail(ctx, GTOP_ERR_HIP, std::string(#call) + ": " + hipGetErrorString(e_)); \
  } while (0)

constexpr size_t kPollDoubles = 16384;                   // outputs per call the completion poll scans (B <= 356 at m = 6)
constexpr uint64_t kPollSentinel = 0x7ff8dead5eed0badull;   // a quiet NaN with a payload the hardware never generates
constexpr double kPollSeconds = 2e-3;
constexpr size_t kZeroCopyDoubles = 1u << 17;   // (measured: 2x faster at B = 1, 1.5x at B = 1024, on par at B = 4096 x 45)
  // host-buffer batches up to this many free variables skip the staged copies

template <typename T>
int ensure(gtop_ctx *c, T **p, size_t *cap, size_t need) {
  if (need <= *cap && *p) return GTOP_OK;
  if (*p) HIPCHK(c, hipFree(*p));
  *p = nullptr;
  *cap = 0;
  HIPCHK(c, hipMalloc(reinterpret_cast<void **>(p), need * sizeof(T)));
  *cap = need;
  return GTOP_OK;
}

int fill_grid(gtop_ctx *c, int nx, int ny, int nz, const double origin[3], const double *map_size,
              double res) {
  if (!origin || nx < 2 || ny < 2 || nz < 2 || !(res > 0.0))
    return fail(c, GTOP_ERR_INVALID, "SDF geometry: need origin, grid >= 2 per axis, resolution > 0");
  if ((double)nx * ny * nz >= 2147483648.0)
    return fail(c, GTOP_ERR_INVALID, "SDF geometry: nx*ny*nz must be < 2^31");
  GtopGrid &g = c->grid;
  g.nx = nx; g.ny = ny; g.nz = nz;
  g.res = res;
  g.res_inv = 1 / res;   // sdf_map.cpp:7
  const int gs[3] = {nx, ny, nz};
  for (int i = 0; i < 3; ++i) {
    g.origin[i] = origin[i];
    g.min_range[i] = origin[i];                                            // sdf_map.cpp:11
    g.max_range[i] = origin[i] + (map_size ? map_size[i] : gs[i] * res);   // sdf_map.cpp:12
  }
  c->have_grid = true;
  return GTOP_OK;
}

void release_sdf(gtop_ctx *c) {
  if (c->own64 && c->sdf64) (void)hipFree(c->sdf64);
  c->sdf64 = nullptr;
  c->sdf32b = nullptr;
  c->own64 = false;
  c->sdf_cap64 = 0;
  c->rec64_ok = c->rec32_ok = c->rec32_stale = false;   // (the record buffers stay: grow-only)
}

// room for the corner records of the current grid, both precisions (allocated up front: a captured map rebuild must
// not allocate, and the first fp32 evaluation may come from inside a capture)
int ensure_records(gtop_ctx *c) {
  const size_t nrec = gtop_record_count(c->grid);
  if (c->rec_cap64 < nrec) {
    if (c->rec64) (void)hipFree(c->rec64);
    c->rec64 = nullptr; c->rec_cap64 = 0;
    HIPCHK(c, hipMalloc(reinterpret_cast<void **>(&c->rec64), nrec * 4 * sizeof(double)));
    c->rec_cap64 = nrec;
  }
  if (c->rec_cap32 < nrec) {
    if (c->rec32) (void)hipFree(c->rec32);
    c->rec32 = nullptr; c->rec_cap32 = 0;
    HIPCHK(c, hipMalloc(reinterpret_cast<void **>(&c->rec32), nrec * 4 * sizeof(float)));
    c->rec_cap32 = nrec;
  }
  return GTOP_OK;
}

int own_sdf_buffers(gtop_ctx *c, size_t nvox) {
  if (!c->own64 || c->sdf_cap64 < nvox) {
    if (c->own64 && c->sdf64) (void)hipFree(c->sdf64);
    c->sdf64 = nullptr; c->own64 = false; c->sdf_cap64 = 0;
    HIPCHK(c, hipMalloc(reinterpret_cast<void **>(&c->sdf64), nvox * sizeof(double)));
    c->own64 = true; c->sdf_cap64 = nvox;
  }
  c->sdf32b = nullptr;
  return ensure_records(c);
}

// Corner records of the fp64 field on stream `s` (the whole field, or the voxel box [vlo, vhi]); the fp32 records too
// when they are wanted now, otherwise they are marked stale and the first fp32 use builds them.
int build_records_on_stream(gtop_ctx *c, hipStream_t s, bool fp32_now, const int *vlo = nullptr, const int *vhi = nullptr) {
  fp32_now = fp32_now && c->fp32_wanted;
  // (both precisions in one pass over the field when both are wanted)
  HIPCHK(c, (gtop_launch_build_records<double, double>(c->grid, c->sdf64, c->rec64, fp32_now ? c->rec32 : nullptr, vlo, vhi, s)));
  c->rec64_ok = true;
  if (fp32_now) {
    c->rec32_ok = true;
    c->rec32_stale = false;
  } else {
    c->rec32_ok = false;
    c->rec32_stale = true;
  }
  return GTOP_OK;
}

// the fp32 records, current, before an fp32 use enqueued on stream `s`
int fp32_records_ready(gtop_ctx *c, hipStream_t s) {
  if (!c->fp32_wanted)
    return fail(c, GTOP_ERR_STATE, "fp32 evaluation on a context whose fp32 records are switched off (gtop_set_field_precisions)");
  c->fp32_in_use = true;
  if (c->rec32_ok) return GTOP_OK;
  if (!c->rec32_stale || !c->sdf64) return fail(c, GTOP_ERR_STATE, "no fp32 distance field resident");
  // only a synchronous entry point (gtop_set_sdf, gtop_init_sdf_map, gtop_update_sdf_map) leaves the records stale,
  // and it has synchronised: the fp64 field is complete whatever stream `s` is
  HIPCHK(c, (gtop_launch_build_records<double, float>(c->grid, c->sdf64, c->rec32, nullptr, nullptr, nullptr, s)));
  c->rec32_ok = true;
  c->rec32_stale = false;
  return GTOP_OK;
}

template <typename R>
void fill_args(const gtop_ctx *c, GtopKernelArgs<R> &a) {
  const GtopGrid &g = c->grid;
  a.nx = g.nx; a.ny = g.ny; a.nz = g.nz;
  for (int i = 0; i < 3; ++i) {
    a.origin[i] = (R)g.origin[i];
    a.lo[i] = (R)g.min_range[i] + (R)1e-4;   // sdf_map.cpp:56-57
    a.hi[i] = (R)g.max_range[i] - (R)1e-4;   // sdf_map.cpp:62-63
    // the same bounds for positions that are float values (the reference keeps pos in `float` locals): the smallest
    // float >= lo and the largest <= hi decide `p < lo` / `p > hi` exactly for every float p
    float lf = (float)a.lo[i], hf = (float)a.hi[i];
    if ((double)lf < (double)a.lo[i]) lf = std::nextafterf(lf, INFINITY);
    if ((double)hf > (double)a.hi[i]) hf = std::nextafterf(hf, -INFINITY);
    a.lo_f[i] = lf;
    a.hi_f[i] = hf;
  }
  a.res = (R)g.res;
  a.res_inv = (R)g.res_inv;
  for (int i = 0; i < 3; ++i) a.idx_origin[i] = g.origin[i];
  a.idx_half = 0.5 * g.res;
  a.idx_rinv = g.res_inv;
  const gtop_params &p = c->prm;
  a.ws = (R)p.ws; a.wc = (R)p.wc; a.alpha = (R)p.alpha; a.d0 = (R)p.d0;
  a.inv_r = (R)1 / (R)p.r;
  a.alpha_over_r = (R)p.alpha / (R)p.r;
  a.alpha_v = (R)p.alpha_v; a.r_v = (R)p.r_v; a.v0 = (R)p.v0;
  a.alpha_a = (R)p.alpha_a; a.r_a = (R)p.r_a; a.a0 = (R)p.a0;
  a.inv_r_v = p.r_v != 0.0 ? (R)1 / (R)p.r_v : (R)0;   // (r_v, r_a are only read with enable_dyn, which requires them non-zero)
  a.inv_r_a = p.r_a != 0.0 ? (R)1 / (R)p.r_a : (R)0;
  a.gv_scale = (R)p.alpha_v * a.inv_r_v;
  a.ga_scale = (R)p.alpha_a * a.inv_r_a;
  a.step = p.step;
}

template <typename R>
int launch_eval(gtop_ctx *c, const R *sdf, int B, int m, const void *d_x, const void *d_Df,
                const void *d_T, int t_stride, void *d_cost, void *d_grad, hipStream_t stream,
                const GtopEvalPlan *optimizer_plan = nullptr) {
  GtopKernelArgs<R> a;
  fill_args(c, a);
  a.sdf = sdf;
  a.x = static_cast<const R *>(d_x);
  a.Df = static_cast<const R *>(d_Df);
  a.T = static_cast<const R *>(d_T);
  a.cost = static_cast<R *>(d_cost);
  a.grad = static_cast<R *>(d_grad);
  a.B = B; a.m = m; a.t_stride = t_stride;
  GtopEvalPlan plan;
  if (optimizer_plan) plan = *optimizer_plan;   // the geometry the optimizer's fused forms run: same bits
  else if (!gtop_eval_plan(B, m, sizeof(R), c->spl, false, &plan))
    return fail(c, GTOP_ERR_INVALID, "this many segments cannot be served (ten lanes per segment: up to 6 segments; "
                                     "one wavefront's LDS: 227)");
  HIPCHK(c, gtop_launch_eval<R>(a, plan, c->prm.enable_dyn != 0, stream));
  return GTOP_OK;
}

int check_eval_state(gtop_ctx *c) {
  if (!c->have_params) return fail(c, GTOP_ERR_STATE, "gtop_set_params has not been called");
  if (!c->have_grid) return fail(c, GTOP_ERR_STATE, "no distance field set");
  return GTOP_OK;
}

}  // namespace

extern "C" {

int gtop_abi_version(void) { return GTOP_ABI_VERSION; }

int gtop_create(gtop_ctx **out, int device) try {
  if (!out) return fail(nullptr, GTOP_ERR_INVALID, "gtop_create: out is NULL");
  *out = nullptr;
  int ndev = 0;
  hipError_t e = hipGetDeviceCount(&ndev);
  if (e != hipSuccess || ndev <= 0)
    return fail(nullptr, GTOP_ERR_NO_DEVICE,
                std::string("gtop_create: no HIP device (hipGetDeviceCount: ") + hipGetErrorString(e) + ")");
  if (device < 0 || device >= ndev) return fail(nullptr, GTOP_ERR_INVALID, "gtop_create: device ordinal out of range");
  hipDeviceProp_t prop;
  if ((e = hipGetDeviceProperties(&prop, device)) != hipSuccess)
    return fail(nullptr, GTOP_ERR_HIP, std::string("hipGetDeviceProperties: ") + hipGetErrorString(e));
  if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0)   // the code objects are gfx950 only
    return fail(nullptr, GTOP_ERR_NO_DEVICE, std::string("gtop_create: device is ") + prop.gcnArchName + ", need gfx950");
  if ((e = hipSetDevice(device)) != hipSuccess)
    return fail(nullptr, GTOP_ERR_HIP, std::string("hipSetDevice: ") + hipGetErrorString(e));
  gtop_ctx *c = new (std::nothrow) gtop_ctx();
  if (!c) return fail(nullptr, GTOP_ERR_INVALID, "gtop_create: out of memory");
  c->device = device;
  if (const char *pc = std::getenv("GTOP_POLL_COMPLETION")) c->poll_completion = std::atoi(pc) != 0;
  c->poll_sentinel = kPollSentinel;
  if (const char *ps = std::getenv("GTOP_POLL_SENTINEL")) c->poll_sentinel = std::strtoull(ps, nullptr, 16);
  if ((e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking)) != hipSuccess) {
    delete c;
    return fail(nullptr, GTOP_ERR_HIP, std::string("hipStreamCreateWithFlags: ") + hipGetErrorString(e));
  }
  *out = c;
  return GTOP_OK;
} GTOP_CATCH_STATUS(nullptr)

int gtop_destroy(gtop_ctx *c) try {
  if (!c) return GTOP_ERR_INVALID;
  (void)hipSetDevice(c->device);
  if (c->stream) (void)hipStreamSynchronize(c->stream);
  release_sdf(c);
  void *bufs[] = {c->win_occ, c->win_dist, c->rec64, c->rec32, c->occ, c->tmp1, c->tmp2, c->rows, c->boxes, c->d_q, c->d_pts,
                  c->d_T, c->d_Df, c->d_x, c->d_cost, c->d_grad,
                  c->mma_vec, c->mma_scal, c->mma_int, c->mma_f, c->mma_g, c->mma_lb, c->mma_ub, c->mma_res};
  for (void *p : bufs)
    if (p) (void)hipFree(p);
  if (c->pin) (void)hipHostFree(c->pin);
  if (c->stream) (void)hipStreamDestroy(c->stream);
  delete c;
  return GTOP_OK;
} GTOP_CATCH_STATUS(c)

const char *gtop_last_error(const gtop_ctx *c) { return c ? c->err.c_str() : g_create_err.c_str(); }

int gtop_set_params(gtop_ctx *c, const gtop_params *p) try {
  if (!c) return GTOP_ERR_INVALID;
  if (!p) return fail(c, GTOP_ERR_INVALID, "params is NULL");
  if (p->step < 0 || p->step > 2)   // grad_traj_optimizer.cpp:129-131
    return fail(c, GTOP_ERR_INVALID, "step number error, step should be 0, 1 or 2");
  if (p->r == 0.0) return fail(c, GTOP_ERR_INVALID, "r must be non-zero");
  if (p->enable_dyn && (p->r_v == 0.0 || p->r_a == 0.0))
    return fail(c, GTOP_ERR_INVALID, "r_v and r_a must be non-zero when enable_dyn is set");
  c->prm = *p;
  c->have_params = true;
  return GTOP_OK;
} GTOP_CATCH_STATUS(c)

int gtop_set_sdf(gtop_ctx *c, const double *dist_host, int nx, int ny, int nz,
                 const double origin[3], const double *map_size, double resolution) try {
  if (!c) return GTOP_ERR_INVALID;
  if (!dist_host) return fail(c, GTOP_ERR_INVALID, "dist_host is NULL");
  HIPCHK(c, hipSetDevice(c->device));
  int rc = fill_grid(c, nx, ny, nz, origin, map_size, resolution);
  if (rc) return rc;
  const size_t nvox = (size_t)nx * ny * nz;
  if ((rc = own_sdf_buffers(c, nvox))) { c->have_grid = false; return rc; }
  HIPCHK(c, hipMemcpyAsync(c->sdf64, dist_host, nvox * sizeof(double), hipMemcpyHostToDevice, c->stream));
  if ((rc = build_records_on_stream(c, c->stream, c->fp32_in_use))) return rc;   // the upload transform
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return GTOP_OK;
} GTOP_CATCH_STATUS(c)

int gtop_set_sdf_device(gtop_ctx *c, int dtype, const void *dist_dev, int nx, int ny, int nz,
                        const double origin[3], const double *map_size, double resolution) try {
  if (!c) return GTOP_ERR_INVALID;
  if (!dist_dev) return fail(c, GTOP_ERR_INVALID, "dist_dev is NULL");
  if (dtype != GTOP_F64 && dtype != GTOP_F32) return fail(c, GTOP_ERR_INVALID, "bad dtype");
  HIPCHK(c, hipSetDevice(c->device));
  int rc = fill_grid(c, nx, ny, nz, origin, map_size, resolution);
  if (rc) return rc;
  release_sdf(c);
  if ((rc = ensure_records(c))) { c->have_grid = false; return rc; }
  // The buffer is borrowed as the boundary copy (gtop_get_sdf and the coarse voxel query read it in place); the
  // corner records the lookups read are derived from it HERE — a caller that rewrites the buffer calls again.
  if (dtype == GTOP_F64) {
    c->sdf64 = const_cast<double *>(static_cast<const double *>(dist_dev));
    if ((rc = build_records_on_stream(c, c->stream, c->fp32_in_use))) return rc;
  } else {
    c->sdf32b = static_cast<const float *>(dist_dev);
    HIPCHK(c, (gtop_launch_build_records<float, float>(c->grid, c->sdf32b, c->rec32, nullptr, nullptr, nullptr, c->stream)));
    c->rec32_ok = true;
  }
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return GTOP_OK;
} GTOP_CATCH_STATUS(c)

int gtop_init_sdf_map(gtop_ctx *c, const double map_size[3], const double origin[3], double resolution) try {
  if (!c) return GTOP_ERR_INVALID;
  if (!map_size || !origin || !(resolution > 0.0))
    return fail(c, GTOP_ERR_INVALID, "initSDFMap: need map_size, origin, resolution > 0");
  HIPCHK(c, hipSetDevice(c->device));
  int gs[3];
  for (int i = 0; i < 3; ++i) gs[i] = (int)std::ceil(map_size[i] / resolution);   // sdf_map.cpp:9
  int rc = fill_grid(c, gs[0], gs[1], gs[2], origin, map_size, resolution);
  if (rc) return rc;
  const size_t nvox = (size_t)gs[0] * gs[1] * gs[2];
  if ((rc = own_sdf_buffers(c, nvox))) { c->have_grid = false; return rc; }
  if ((rc = ensure(c, &c->occ, &c->cap_occ, nvox))) return rc;
  // sdf_map.cpp:22-23: distance 10000, occupancy 0
  HIPCHK(c, gtop_launch_esdf_reset(c->occ, c->sdf64, nvox, c->stream));
  if ((rc = build_records_on_stream(c, c->stream, c->fp32_in_use))) return rc;
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return GTOP_OK;
} GTOP_CATCH_STATUS(c)

// the build proper: obstacle points already in HBM, launches on `s`, no synchronisation
static int update_sdf_map_on_stream(gtop_ctx *c, const double *d_pts, int npts, hipStream_t s, bool convert_now) {
  const GtopGrid &g = c->grid;
  const size_t nvox = (size_t)g.nx * g.ny * g.nz;
  int rc;
  // gtop_set_sdf may have moved the context to a larger grid since gtop_init_sdf_map sized the occupancy
  if ((rc = ensure(c, &c->occ, &c->cap_occ, nvox))) return rc;
  if ((rc = ensure(c, &c->tmp1, &c->cap_tmp1, nvox))) return rc;
  if ((rc = ensure(c, &c->tmp2, &c->cap_tmp2, nvox))) return rc;
  if ((rc = ensure(c, &c->rows, &c->cap_rows, gtop_esdf_rows_ints(g)))) return rc;
  if (!gtop_esdf_supported(g))
    return fail(c, GTOP_ERR_INVALID, "updateSDFMap: grid too large for the device builder (nz <= 4096, nx, ny <= 32768)");
  // resetBuffer (sdf_map.cpp:26-53): the occupancy; the distances need no reset of their own, the x sweep
  // writes every voxel (10000 where the line holds no obstacle, as the reset would have left it)
  HIPCHK(c, gtop_launch_esdf_reset(c->occ, nullptr, nvox, s));
  HIPCHK(c, gtop_launch_esdf_mark(g, d_pts, npts, c->occ, s));         // setOccupancy
  HIPCHK(c, gtop_launch_esdf_build(g, c->occ, c->tmp1, c->tmp2, c->rows, c->sdf64, nullptr, s));   // updateESDF3d
  // the corner records behind it, on the same stream (device-side: holds for graph replays too); the fp32 ones now
  // when they are wanted now, otherwise at the first fp32 evaluation (host-synchronous caller only, see gtop_ctx)
  return build_records_on_stream(c, s, convert_now || c->fp32_in_use);
}

int gtop_update_sdf_map(gtop_ctx *c, const double *pts, int npts) try {
  if (!c) return GTOP_ERR_INVALID;
  if (npts < 0 || (npts > 0 && !pts)) return fail(c, GTOP_ERR_INVALID, "bad obstacle list");
  if (!c->have_grid || !c->own64 || !c->occ)
    return fail(c, GTOP_ERR_STATE, "updateSDFMap: call gtop_init_sdf_map first");
  HIPCHK(c, hipSetDevice(c->device));
  int rc;
  if (npts > 0) {
    if ((rc = ensure(c, &c->d_pts, &c->pts_cap, (size_t)npts * 3))) return rc;
    HIPCHK(c, hipMemcpyAsync(c->d_pts, pts, (size_t)npts * 3 * sizeof(double), hipMemcpyHostToDevice, c->stream));
  }
  if ((rc = update_sdf_map_on_stream(c, c->d_pts, npts, c->stream, /*convert_now=*/false))) return rc;
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return GTOP_OK;
} GTOP_CATCH_STATUS(c)

int gtop_update_sdf_map_device(gtop_ctx *c, const void *d_pts, int npts, void *hip_stream) try {
  if (!c) return GTOP_ERR_INVALID;
  if (npts < 0 || (npts > 0 && !d_pts)) return fail(c, GTOP_ERR_INVALID, "bad obstacle list");
  if (!c->have_grid || !c->own64 || !c->occ)
    return fail(c, GTOP_ERR_STATE, "updateSDFMap: call gtop_init_sdf_map first");
  HIPCHK(c, hipSetDevice(c->device));
  return update_sdf_map_on_stream(c, static_cast<const double *>(d_pts), npts, static_cast<hipStream_t>(hip_stream),
                                  /*convert_now=*/true);
} GTOP_CATCH_STATUS(c)

// The window of (min_pos, max_pos) in voxel indices, as resetBuffer(min, max) and setUpdateRange compute it
// (sdf_map.cpp:28-45, :244-260): both positions clamped to [min_range, max_range], then posToIndex(min_pos) and
// posToIndex(max_pos - res/2).  Indices are clipped into the grid (memory safety only: they are inside already).
static void window_ids(const GtopGrid &g, const double min_pos[3], const double max_pos[3], int lo[3], int hi[3]) {
  const int n[3] = {g.nx, g.ny, g.nz};
  for (int i = 0; i < 3; ++i) {
    const double a = std::max(min_pos[i], g.min_range[i]), b = std::min(max_pos[i], g.max_range[i]);
    lo[i] = (int)std::floor((a - g.origin[i]) * g.res_inv);                      // posToIndex, :71-74
    hi[i] = (int)std::floor(((b - g.res / 2) - g.origin[i]) * g.res_inv);
    lo[i] = std::max(lo[i], 0);
    hi[i] = std::min(hi[i], n[i] - 1);
  }
}

// resetBuffer(min, max) + setOccupancy per point + setUpdateRange(min, max) + updateESDF3d, then the corner records of
// the voxels that changed; launches on `s`, no synchronisation
static int update_window_on_stream(gtop_ctx *c, const double min_pos[3], const double max_pos[3], const double *d_pts,
                                   int npts, hipStream_t s, bool convert_now) {
  const GtopGrid &g = c->grid;
  const size_t nvox = (size_t)g.nx * g.ny * g.nz;
  int lo[3], hi[3];
  window_ids(g, min_pos, max_pos, lo, hi);
  int rc;
  if ((rc = ensure(c, &c->occ, &c->cap_occ, nvox))) return rc;
  if ((rc = ensure(c, &c->tmp1, &c->cap_tmp1, nvox))) return rc;
  if ((rc = ensure(c, &c->tmp2, &c->cap_tmp2, nvox))) return rc;
  // (scratch for the compact path, sized for the whole grid whatever this window is: the first window update of a map
  // allocates, later ones of any size do not — capturable, and no implicit device synchronisation from a reallocation
  // between two enqueued updates)
  if ((rc = ensure(c, &c->rows, &c->cap_rows, gtop_esdf_rows_ints(g)))) return rc;
  if ((rc = ensure(c, &c->win_occ, &c->cap_win_occ, nvox))) return rc;
  if ((rc = ensure(c, &c->win_dist, &c->cap_win_dist, nvox))) return rc;
  const bool empty = hi[0] < lo[0] || hi[1] < lo[1] || hi[2] < lo[2];
  const bool whole = !empty && lo[0] == 0 && lo[1] == 0 && lo[2] == 0 && hi[0] == g.nx - 1 && hi[1] == g.ny - 1 && hi[2] == g.nz - 1;
  if (whole)   // the window is the map: the whole-grid builder (same results: every distance is 10000 after the reset)
    return update_sdf_map_on_stream(c, d_pts, npts, s, convert_now);
  // (the compact path — see below — resets and marks on its own)
  const int cwx = hi[0] - lo[0] + 1, cwy = hi[1] - lo[1] + 1, cwz = hi[2] - lo[2] + 1;
  GtopGrid csub = g;
  csub.nx = cwx; csub.ny = cwy; csub.nz = cwz;
  const bool compact = !empty && cwx >= 12 && cwy >= 12 && cwz >= 3 && gtop_esdf_supported(csub);
  if (!compact) {
    HIPCHK(c, gtop_launch_esdf_window_reset(g, lo, hi, c->occ, c->sdf64, s));
    HIPCHK(c, gtop_launch_esdf_mark(g, d_pts, npts, c->occ, s));   // (anywhere in the map: setOccupancy does not look at the window)
  }
  if (empty) return GTOP_OK;
  // The sweeps over the window see nothing outside it: the update IS the whole-grid transform of the window taken alone.
  // A window of at least 12 x 12 x 3 voxels therefore goes through the whole-grid builder (gtop_esdf.hip: packed 16-bit
  // scans, candidate lists, slab skipping) on a compact copy of its occupancy, and the result is written back into the
  // window — 10x less time per voxel than the plain window kernels, which serve the slivers.
  const int wx = hi[0] - lo[0] + 1, wy = hi[1] - lo[1] + 1, wz = hi[2] - lo[2] + 1;
  GtopGrid sub = g;
  sub.nx = wx; sub.ny = wy; sub.nz = wz;
  if (compact) {
    // reset + marking of the map's occupancy and of the compact copy in two kernels, no gather; the window's distances
    // need no reset: the scatter below rewrites every voxel of it
    (void)wx; (void)wy; (void)wz;
    HIPCHK(c, gtop_launch_esdf_window_reset_mark_compact(g, lo, hi, d_pts, npts, c->occ, c->win_occ, s));
    HIPCHK(c, gtop_launch_esdf_build(sub, c->win_occ, c->tmp1, c->tmp2, c->rows, c->win_dist, nullptr, s));
    HIPCHK(c, gtop_launch_esdf_window_scatter(g, lo, hi, c->win_dist, c->sdf64, s));
  } else {
    HIPCHK(c, gtop_launch_esdf_window_build(g, lo, hi, c->occ, c->tmp1, c->tmp2, c->sdf64, s));
  }
  // Only the records that hold a voxel of the window change.  fp32 records that are current stay current (their window
  // is rebuilt in the same pass); stale ones cannot be made current by a window: they are rebuilt whole where this
  // entry's rule says fp32 must follow (the capturable device entry, or a context that runs fp32 evaluations), and
  // stay stale — to be rebuilt at the first fp32 use — otherwise.
  if (c->rec32_ok) return build_records_on_stream(c, s, true, lo, hi);
  if ((rc = build_records_on_stream(c, s, false, lo, hi))) return rc;
  if ((convert_now || c->fp32_in_use) && c->fp32_wanted) return fp32_records_ready(c, s);
  return GTOP_OK;
}

int gtop_update_sdf_map_window(gtop_ctx *c, const double min_pos[3], const double max_pos[3], const double *pts,
                               int npts) try {
  if (!c) return GTOP_ERR_INVALID;
  if (!min_pos || !max_pos || npts < 0 || (npts > 0 && !pts)) return fail(c, GTOP_ERR_INVALID, "update window: bad arguments");
  if (!c->have_grid || !c->own64 || !c->occ || !c->rec64_ok)
    return fail(c, GTOP_ERR_STATE, "update window: call gtop_init_sdf_map first");
  HIPCHK(c, hipSetDevice(c->device));
  int rc;
  if (npts > 0) {
    if ((rc = ensure(c, &c->d_pts, &c->pts_cap, (size_t)npts * 3))) return rc;
    HIPCHK(c, hipMemcpyAsync(c->d_pts, pts, (size_t)npts * 3 * sizeof(double), hipMemcpyHostToDevice, c->stream));
  }
  if ((rc = update_window_on_stream(c, min_pos, max_pos, c->d_pts, npts, c->stream, /*convert_now=*/false))) return rc;
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return GTOP_OK;
} GTOP_CATCH_STATUS(c)

int gtop_update_sdf_map_window_device(gtop_ctx *c, const double min_pos[3], const double max_pos[3], const void *d_pts,
                                      int npts, void *hip_stream) try {
  if (!c) return GTOP_ERR_INVALID;
  if (!min_pos || !max_pos || npts < 0 || (npts > 0 && !d_pts)) return fail(c, GTOP_ERR_INVALID, "update window: bad arguments");
  if (!c->have_grid || !c->own64 || !c->occ || !c->rec64_ok)
    return fail(c, GTOP_ERR_STATE, "update window: call gtop_init_sdf_map first");
  HIPCHK(c, hipSetDevice(c->device));
  return update_window_on_stream(c, min_pos, max_pos, static_cast<const double *>(d_pts), npts,
                                 static_cast<hipStream_t>(hip_stream), /*convert_now=*/true);
} GTOP_CATCH_STATUS(c)

int gtop_get_sdf(gtop_ctx *c, double *dist_host, int grid_out[3]) try {
  if (!c) return GTOP_ERR_INVALID;
  if (!c->have_grid || !c->sdf64) return fail(c, GTOP_ERR_STATE, "no fp64 distance field resident");
  HIPCHK(c, hipSetDevice(c->device));
  const GtopGrid &g = c->grid;
  if (grid_out) { grid_out[0] = g.nx; grid_out[1] = g.ny; grid_out[2] = g.nz; }
  if (dist_host) {
    const size_t nvox = (size_t)g.nx * g.ny * g.nz;
    HIPCHK(c, hipMemcpyAsync(dist_host, c->sdf64, nvox * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
  }
  return GTOP_OK;
} GTOP_CATCH_STATUS(c)

int gtop_set_problem(gtop_ctx *c, int B, int m, const double *segment_time, int time_stride,
                     const double *Df) try {
  if (!c) return GTOP_ERR_INVALID;
  if (B < 1 || m < 2 || !segment_time || !Df || (time_stride != 0 && time_stride != m))
    return fail(c, GTOP_ERR_INVALID, "set_problem: need B >= 1, m >= 2, time_stride in {0, m}");
  const size_t nT = time_stride ? (size_t)B * m : (size_t)m;
  for (size_t i = 0; i < nT; ++i)
    if (!(segment_time[i] > 0.0)) return fail(c, GTOP_ERR_INVALID, "segment_time must be > 0");
  HIPCHK(c, hipSetDevice(c->device));
  const size_t n = 9 * (size_t)(m - 1);
  int rc;
  if ((rc = ensure(c, &c->d_T, &c->cap_T, nT))) return rc;
  if ((rc = ensure(c, &c->d_Df, &c->cap_Df, (size_t)B * 18))) return rc;
  if ((rc = ensure(c, &c->d_x, &c->cap_x, (size_t)B * n))) return rc;
  if ((rc = ensure(c, &c->d_grad, &c->cap_grad, (size_t)B * n))) return rc;
  if ((rc = ensure(c, &c->d_cost, &c->cap_cost, (size_t)B))) return rc;
  HIPCHK(c, hipMemcpyAsync(c->d_T, segment_time, nT * sizeof(double), hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipMemcpyAsync(c->d_Df, Df, (size_t)B * 18 * sizeof(double), hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  c->B = B; c->m = m; c->t_stride = time_stride;
  return GTOP_OK;
} GTOP_CATCH_STATUS(c)

int gtop_eval_batch(gtop_ctx *c, int B, const double *x, double *cost, double *grad) try {
  if (!c) return GTOP_ERR_INVALID;
  int rc = check_eval_state(c);
  if (rc) return rc;
  if (c->B == 0) return fail(c, GTOP_ERR_STATE, "gtop_set_problem has not been called");
  if (!c->rec64_ok) return fail(c, GTOP_ERR_STATE, "no fp64 distance field resident");
  if (B < 1 || B > c->B || !x || !cost || !grad)
    return fail(c, GTOP_ERR_INVALID, "eval_batch: 1 <= B <= problem batch, non-NULL buffers");
  HIPCHK(c, hipSetDevice(c->device));
  const size_t n = 9 * (size_t)(c->m - 1);
  if ((size_t)B * n <= kZeroCopyDoubles) {
    // Small batches (the NLopt callback is B = 1): three staged copies cost more than the
    // evaluation.  The kernel reads x from, and writes cost and gradient to, pinned host
    // memory it can address directly — one launch and one synchronisation.
    const size_t bn = (size_t)B * n, need = 2 * bn + (size_t)B;
    if (need > c->cap_pin) {
      if (c->pin) HIPCHK(c, hipHostFree(c->pin));
      c->pin = nullptr;
      c->cap_pin = 0;
      const size_t cap = need < 4096 ? 4096 : need;
      // coherent (fine-grained): the kernel's stores must reach host memory as they retire, not at the end of the
      // kernel — the completion poll below reads them while the kernel is still "running" for the runtime
      HIPCHK(c, hipHostMalloc(reinterpret_cast<void **>(&c->pin), cap * sizeof(double),
                              hipHostMallocMapped | hipHostMallocCoherent));
      c->cap_pin = cap;
      HIPCHK(c, hipHostGetDevicePointer(reinterpret_cast<void **>(&c->pin_dev), c->pin, 0));
    }
    double *dpin = c->pin_dev;
    std::memcpy(c->pin, x, bn * sizeof(double));
    // The serial caller's round trip (the NLopt callback, B = 1) is launch + 3.5 us of kernel + completion, and
    // most of the completion is the end-of-kernel protocol (cache release, completion signal, the runtime's wait).
    // Every output is stored exactly once, 8 bytes at a time, into coherent host memory: the slots are preset to a
    // NaN pattern no evaluation produces, and the call returns when none is left.  A kernel that does not finish
    // within kPollSeconds falls back to the stream synchronisation (which also reports a fault).
    const size_t nout = (size_t)B + bn;
    const bool poll = c->poll_completion && nout <= kPollDoubles;
    volatile uint64_t *out = reinterpret_cast<volatile uint64_t *>(c->pin + bn);
    if (poll)
      for (size_t i = 0; i < nout; ++i) out[i] = c->poll_sentinel;
    if ((rc = launch_eval<double>(c, c->rec64, B, c->m, dpin, c->d_Df, c->d_T, c->t_stride, dpin + bn,
                                  dpin + bn + B, c->stream)))
      return rc;
    bool done = false;
    if (poll) {
      const auto t0 = std::chrono::steady_clock::now();
      size_t i = 0;
      unsigned spins = 0;
      // A slot has landed when BOTH of its 32-bit halves differ from the sentinel's: an 8-byte store that reached
      // host memory as two dwords is then never taken half-written.  A genuine result that shares a half with the
      // sentinel (2^-32 per half) is merely never "seen": the call falls back to the stream wait, never returns a
      // wrong value.
      const uint32_t kLo = (uint32_t)c->poll_sentinel, kHi = (uint32_t)(c->poll_sentinel >> 32);
      while (i < nout) {
        const uint64_t v = out[i];
        if ((uint32_t)v != kLo && (uint32_t)(v >> 32) != kHi) { ++i; continue; }
        __builtin_ia32_pause();
        if ((++spins & 255u) == 0 &&
            std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > kPollSeconds)
          break;
      }
      done = i == nout;
      std::atomic_thread_fence(std::memory_order_acquire);
    }
    if (!done) HIPCHK(c, hipStreamSynchronize(c->stream));
    std::memcpy(cost, c->pin + bn, (size_t)B * sizeof(double));
    std::memcpy(grad, c->pin + bn + B, bn * sizeof(double));
    return GTOP_OK;
  }
  HIPCHK(c, hipMemcpyAsync(c->d_x, x, (size_t)B * n * sizeof(double), hipMemcpyHostToDevice, c->stream));
  if ((rc = launch_eval<double>(c, c->rec64, B, c->m, c->d_x, c->d_Df, c->d_T, c->t_stride, c->d_cost,
                                c->d_grad, c->stream)))
    return rc;
  HIPCHK(c, hipMemcpyAsync(cost, c->d_cost, (size_t)B * sizeof(double), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipMemcpyAsync(grad, c->d_grad, (size_t)B * n * sizeof(double), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return GTOP_OK;
} GTOP_CATCH_STATUS(c)

double gtop_cost_nlopt(unsigned n, const double *x, double *grad, void *vctx) try {
  gtop_ctx *c = static_cast<gtop_ctx *>(vctx);
  if (!c) return HUGE_VAL;
  const auto tb1 = std::chrono::steady_clock::now();
  c->iter_num++;   // grad_traj_optimizer.cpp:284
  if (c->B == 0 || n != 9u * (unsigned)(c->m - 1) || !x) {
    fail(c, GTOP_ERR_INVALID, "cost_nlopt: n does not match the problem (9(m-1)) or x is NULL");
    return HUGE_VAL;
  }
  double cost = HUGE_VAL;
  std::vector<double> gtmp;
  double *g = grad;
  if (!g) {   // the reference always computes the gradient (:426)
    gtmp.resize(n);
    g = gtmp.data();
  }
  if (gtop_eval_batch(c, 1, x, &cost, g) != GTOP_OK) return HUGE_VAL;
  const auto te1 = std::chrono::steady_clock::now();
  c->total_time += std::chrono::duration<double>(te1 - tb1).count();   // :436
  // best-so-far cost curve, :439-447
  c->vec_time.push_back(std::chrono::duration<double>(te1 - c->time_start).count());
  if (c->vec_cost.empty() || c->vec_cost.back() > cost)
    c->vec_cost.push_back(cost);
  else
    c->vec_cost.push_back(c->vec_cost.back());
  return cost;
} GTOP_CATCH_HUGE(static_cast<gtop_ctx *>(vctx))

int gtop_eval_device(gtop_ctx *c, int dtype, int B, int m, const void *d_x, const void *d_Df,
                     const void *d_T, int time_stride, void *d_cost, void *d_grad, void *hip_stream) try {
  if (!c) return GTOP_ERR_INVALID;
  int rc = check_eval_state(c);
  if (rc) return rc;
  if (B < 0 || m < 2 || (time_stride != 0 && time_stride != m))
    return fail(c, GTOP_ERR_INVALID, "eval_device: need B >= 0, m >= 2, time_stride in {0, m}");
  if (B == 0) return GTOP_OK;
  if (!d_x || !d_Df || !d_T || !d_cost || !d_grad) return fail(c, GTOP_ERR_INVALID, "eval_device: NULL buffer");
  HIPCHK(c, hipSetDevice(c->device));
  hipStream_t s = static_cast<hipStream_t>(hip_stream);
  if (dtype == GTOP_F64) {
    if (!c->rec64_ok) return fail(c, GTOP_ERR_STATE, "no fp64 distance field resident");
    return launch_eval<double>(c, c->rec64, B, m, d_x, d_Df, d_T, time_stride, d_cost, d_grad, s);
  } else if (dtype == GTOP_F32) {
    if ((rc = fp32_records_ready(c, s))) return rc;
    return launch_eval<float>(c, c->rec32, B, m, d_x, d_Df, d_T, time_stride, d_cost, d_grad, s);
  }
  return fail(c, GTOP_ERR_INVALID, "bad dtype");
} GTOP_CATCH_STATUS(c)

// ---- setup (f3) and post-processing (f4) ----
int gtop_setup_paths_device(gtop_ctx *c, int B, int m, const void *d_wp, double mean_v, double init_time,
                            void *d_T, void *d_Df, void *d_x0, void *hip_stream) try {
  if (!c) return GTOP_ERR_INVALID;
  if (B < 0 || m < 2 || !(mean_v > 0.0)) return fail(c, GTOP_ERR_INVALID, "setup_paths: need B >= 0, m >= 2, mean_v > 0");
  if (B == 0) return GTOP_OK;
  if (!d_wp || !d_T || !d_Df || !d_x0) return fail(c, GTOP_ERR_INVALID, "setup_paths: NULL buffer");
  HIPCHK(c, hipSetDevice(c->device));
  HIPCHK(c, gtop_launch_setup_paths(B, m, static_cast<const double *>(d_wp), mean_v, init_time,
                                    static_cast<double *>(d_T), static_cast<double *>(d_Df),
                                    static_cast<double *>(d_x0), static_cast<hipStream_t>(hip_stream)));
  return GTOP_OK;
} GTOP_CATCH_STATUS(c)

int gtop_set_paths(gtop_ctx *c, int B, int m, const double *waypoints, double mean_v, double init_time,
                   double *x0) try {
  if (!c) return GTOP_ERR_INVALID;
  if (B < 1 || m < 2 || !waypoints || !(mean_v > 0.0))
    return fail(c, GTOP_ERR_INVALID, "set_paths: need B >= 1, m >= 2 (3+ waypoints), mean_v > 0");
  // same rule as gtop_set_problem: every segment time must be > 0.  Coincident consecutive waypoints give
  // T_s = 0 (grad_traj_optimizer.cpp:73-81), a singular A_s in the reference and NaN here.
  for (int b = 0; b < B; ++b)
    for (int s = 0; s < m; ++s) {
      const double *p = waypoints + ((size_t)b * (m + 1) + s) * 3;
      const double dx = p[0] - p[3], dy = p[1] - p[4], dz = p[2] - p[5];
      const double T = std::sqrt(dx * dx + dy * dy + dz * dz) / mean_v + (s == 0 ? init_time : 0.0);
      if (!(T > 0.0)) return fail(c, GTOP_ERR_INVALID, "set_paths: coincident consecutive waypoints (segment time 0)");
    }
  HIPCHK(c, hipSetDevice(c->device));
  const size_t n = 9 * (size_t)(m - 1), nwp = (size_t)B * (m + 1) * 3;
  int rc;
  if ((rc = ensure(c, &c->d_T, &c->cap_T, (size_t)B * m))) return rc;
  if ((rc = ensure(c, &c->d_Df, &c->cap_Df, (size_t)B * 18))) return rc;
  if ((rc = ensure(c, &c->d_x, &c->cap_x, (size_t)B * n))) return rc;
  if ((rc = ensure(c, &c->d_grad, &c->cap_grad, (size_t)B * n))) return rc;
  if ((rc = ensure(c, &c->d_cost, &c->cap_cost, (size_t)B))) return rc;
  if ((rc = ensure(c, &c->d_pts, &c->pts_cap, nwp))) return rc;   // staging, shared with the obstacle list
  HIPCHK(c, hipMemcpyAsync(c->d_pts, waypoints, nwp * sizeof(double), hipMemcpyHostToDevice, c->stream));
  if ((rc = gtop_setup_paths_device(c, B, m, c->d_pts, mean_v, init_time, c->d_T, c->d_Df, c->d_x, c->stream)))
    return rc;
  if (x0) HIPCHK(c, hipMemcpyAsync(x0, c->d_x, (size_t)B * n * sizeof(double), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  c->B = B; c->m = m; c->t_stride = m;
  return GTOP_OK;
} GTOP_CATCH_STATUS(c)

int gtop_get_problem(gtop_ctx *c, double *segment_time, double *Df) try {
  if (!c) return GTOP_ERR_INVALID;
  if (c->B == 0) return fail(c, GTOP_ERR_STATE, "no problem set");
  HIPCHK(c, hipSetDevice(c->device));
  const size_t nT = c->t_stride ? (size_t)c->B * c->m : (size_t)c->m;
  if (segment_time)
    HIPCHK(c, hipMemcpyAsync(segment_time, c->d_T, nT * sizeof(double), hipMemcpyDeviceToHost, c->stream));
  if (Df) HIPCHK(c, hipMemcpyAsync(Df, c->d_Df, (size_t)c->B * 18 * sizeof(double), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return GTOP_OK;
} GTOP_CATCH_STATUS(c)

int gtop_coefficients_device(gtop_ctx *c, int B, int m, const void *d_x, const void *d_Df, const void *d_T,
                             int time_stride, void *d_coeff, void *hip_stream) try {
  if (!c) return GTOP_ERR_INVALID;
  if (B < 0 || m < 2 || (time_stride != 0 && time_stride != m))
    return fail(c, GTOP_ERR_INVALID, "coefficients: need B >= 0, m >= 2, time_stride in {0, m}");
  if (B == 0) return GTOP_OK;
  if (!d_x || !d_Df || !d_T || !d_coeff) return fail(c, GTOP_ERR_INVALID, "coefficients: NULL buffer");
  HIPCHK(c, hipSetDevice(c->device));
  HIPCHK(c, gtop_launch_coefficients(B, m, static_cast<const double *>(d_x), static_cast<const double *>(d_Df),
                                     static_cast<const double *>(d_T), time_stride, static_cast<double *>(d_coeff),
                                     static_cast<hipStream_t>(hip_stream)));
  return GTOP_OK;
} GTOP_CATCH_STATUS(c)

int gtop_sample_trajectories_device(gtop_ctx *c, int B, int m, const void *d_coeff, const void *d_T, int time_stride,
                                    double dt_sample, void *d_stats, void *d_samples, int max_samples,
                                    void *hip_stream) try {
  if (!c) return GTOP_ERR_INVALID;
  if (B < 0 || m < 1 || !(dt_sample > 0.0) || (time_stride != 0 && time_stride != m) || max_samples < 0)
    return fail(c, GTOP_ERR_INVALID, "eval_trajectories: need B >= 0, m >= 1, dt_sample > 0, time_stride in {0, m}");
  if (B == 0) return GTOP_OK;
  if (!d_coeff || !d_T || !d_stats) return fail(c, GTOP_ERR_INVALID, "eval_trajectories: NULL buffer");
  HIPCHK(c, hipSetDevice(c->device));
  HIPCHK(c, gtop_launch_eval_trajectories(B, m, static_cast<const double *>(d_coeff), static_cast<const double *>(d_T),
                                          time_stride, dt_sample, static_cast<double *>(d_stats),
                                          max_samples > 0 ? static_cast<double *>(d_samples) : nullptr, max_samples,
                                          static_cast<hipStream_t>(hip_stream)));
  return GTOP_OK;
} GTOP_CATCH_STATUS(c)

int gtop_eval_trajectories_device(gtop_ctx *c, int B, int m, const void *d_coeff, const void *d_T, int time_stride,
                                  double dt_sample, void *d_stats, void *hip_stream) try {
  return gtop_sample_trajectories_device(c, B, m, d_coeff, d_T, time_stride, dt_sample, d_stats, nullptr, 0, hip_stream);
} GTOP_CATCH_STATUS(c)

int gtop_trajectory_stats(gtop_ctx *c, int B, const double *x, double dt_sample, double *coeff, double *stats) try {
  return gtop_trajectory_samples(c, B, x, dt_sample, coeff, stats, nullptr, 0);
} GTOP_CATCH_STATUS(c)

int gtop_trajectory_samples(gtop_ctx *c, int B, const double *x, double dt_sample, double *coeff, double *stats,
                            double *samples, int max_samples) try {
  if (!c) return GTOP_ERR_INVALID;
  if (c->B == 0) return fail(c, GTOP_ERR_STATE, "gtop_set_problem / gtop_set_paths has not been called");
  if (B < 1 || B > c->B || !x || (!coeff && !stats && !samples) || max_samples < 0 || (samples && max_samples == 0))
    return fail(c, GTOP_ERR_INVALID, "trajectory_stats: 1 <= B <= problem batch, x and an output required");
  HIPCHK(c, hipSetDevice(c->device));
  const int m = c->m;
  const size_t n = 9 * (size_t)(m - 1), ncoef = (size_t)B * m * 18;
  int rc;
  if ((rc = ensure(c, &c->mma_g, &c->cap_mma_g, ncoef > (size_t)B * n ? ncoef : (size_t)B * n))) return rc;   // coefficient scratch
  if ((rc = ensure(c, &c->mma_f, &c->cap_mma_f, (size_t)B * GTOP_TRAJ_STATS))) return rc;
  HIPCHK(c, hipMemcpyAsync(c->d_x, x, (size_t)B * n * sizeof(double), hipMemcpyHostToDevice, c->stream));
  if ((rc = gtop_coefficients_device(c, B, m, c->d_x, c->d_Df, c->d_T, c->t_stride, c->mma_g, c->stream))) return rc;
  if (coeff) HIPCHK(c, hipMemcpyAsync(coeff, c->mma_g, ncoef * sizeof(double), hipMemcpyDeviceToHost, c->stream));
  if (stats || samples) {
    const size_t ns = samples ? (size_t)B * max_samples * 3 : 0;
    if (ns && (rc = ensure(c, &c->d_q, &c->cap_q, ns))) return rc;   // (the query staging buffer doubles as sample scratch)
    if (ns) HIPCHK(c, hipMemsetAsync(c->d_q, 0, ns * sizeof(double), c->stream));   // rows past a trajectory's count read 0
    if ((rc = gtop_sample_trajectories_device(c, B, m, c->mma_g, c->d_T, c->t_stride, dt_sample, c->mma_f,
                                              samples ? c->d_q : nullptr, samples ? max_samples : 0, c->stream)))
      return rc;
    if (stats)
      HIPCHK(c, hipMemcpyAsync(stats, c->mma_f, (size_t)B * GTOP_TRAJ_STATS * sizeof(double), hipMemcpyDeviceToHost,
                               c->stream));
    if (samples) HIPCHK(c, hipMemcpyAsync(samples, c->d_q, ns * sizeof(double), hipMemcpyDeviceToHost, c->stream));
  }
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return GTOP_OK;
} GTOP_CATCH_STATUS(c)

int gtop_set_moving_boxes(gtop_ctx *c, int nbox, const double *p0, const double *vel, const double *scale) try {
  if (!c) return GTOP_ERR_INVALID;
  if (nbox < 0 || (nbox > 0 && (!p0 || !vel || !scale))) return fail(c, GTOP_ERR_INVALID, "set_moving_boxes: bad box list");
  HIPCHK(c, hipSetDevice(c->device));
  c->nbox = 0;
  if (nbox == 0) return GTOP_OK;
  int rc;
  const size_t n3 = (size_t)nbox * 3;
  if ((rc = ensure(c, &c->boxes, &c->cap_boxes, 3 * n3))) return rc;
  HIPCHK(c, hipMemcpyAsync(c->boxes, p0, n3 * sizeof(double), hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipMemcpyAsync(c->boxes + n3, vel, n3 * sizeof(double), hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipMemcpyAsync(c->boxes + 2 * n3, scale, n3 * sizeof(double), hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));   // the host arrays may go away
  c->nbox = nbox;
  return GTOP_OK;
} GTOP_CATCH_STATUS(c)

int gtop_edt_query_device(gtop_ctx *c, int N, const void *d_pos, const void *d_time, void *d_dist, void *d_grad,
                          void *hip_stream) try {
  if (!c) return GTOP_ERR_INVALID;
  if (!c->have_grid || !c->rec64_ok) return fail(c, GTOP_ERR_STATE, "no fp64 distance field resident");
  if (N < 0) return fail(c, GTOP_ERR_INVALID, "edt_query: N < 0");
  if (N == 0) return GTOP_OK;
  if (!d_pos || !d_time || !d_dist || !d_grad) return fail(c, GTOP_ERR_INVALID, "edt_query: NULL buffer");
  HIPCHK(c, hipSetDevice(c->device));
  const size_t n3 = (size_t)c->nbox * 3;
  HIPCHK(c, gtop_launch_edt_query(c->grid, c->sdf64, c->rec64, c->nbox, c->boxes, c->boxes + n3, c->boxes + 2 * n3, N,
                                  static_cast<const double *>(d_pos), static_cast<const double *>(d_time),
                                  static_cast<double *>(d_dist), static_cast<double *>(d_grad),
                                  static_cast<hipStream_t>(hip_stream)));
  return GTOP_OK;
} GTOP_CATCH_STATUS(c)

int gtop_edt_query(gtop_ctx *c, int N, const double *pos, const double *time, double *dist, double *grad) try {
  if (!c) return GTOP_ERR_INVALID;
  if (N < 0 || (N > 0 && (!pos || !time || !dist || !grad))) return fail(c, GTOP_ERR_INVALID, "edt_query: bad arguments");
  if (N == 0) return GTOP_OK;
  HIPCHK(c, hipSetDevice(c->device));
  int rc;
  const size_t n = (size_t)N;
  if ((rc = ensure(c, &c->d_q, &c->cap_q, 8 * n))) return rc;
  double *dp = c->d_q, *dt = dp + 3 * n, *dd = dt + n, *dg = dd + n;
  HIPCHK(c, hipMemcpyAsync(dp, pos, 3 * n * sizeof(double), hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipMemcpyAsync(dt, time, n * sizeof(double), hipMemcpyHostToDevice, c->stream));
  if ((rc = gtop_edt_query_device(c, N, dp, dt, dd, dg, c->stream))) return rc;
  HIPCHK(c, hipMemcpyAsync(dist, dd, n * sizeof(double), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipMemcpyAsync(grad, dg, 3 * n * sizeof(double), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return GTOP_OK;
} GTOP_CATCH_STATUS(c)

// EDTEnvironment::evaluateCoarseEDT (src/edt_environment.cpp:124-136)
int gtop_edt_coarse_query_device(gtop_ctx *c, int N, const void *d_pos, const void *d_time, void *d_dist,
                                 void *hip_stream) try {
  if (!c) return GTOP_ERR_INVALID;
  if (!c->have_grid || !c->sdf64) return fail(c, GTOP_ERR_STATE, "no fp64 distance field resident");
  if (N < 0) return fail(c, GTOP_ERR_INVALID, "edt_coarse_query: N < 0");
  if (N == 0) return GTOP_OK;
  if (!d_pos || !d_time || !d_dist) return fail(c, GTOP_ERR_INVALID, "edt_coarse_query: NULL buffer");
  HIPCHK(c, hipSetDevice(c->device));
  const size_t n3 = (size_t)c->nbox * 3;
  HIPCHK(c, gtop_launch_edt_query(c->grid, c->sdf64, c->rec64, c->nbox, c->boxes, c->boxes + n3, c->boxes + 2 * n3, N,
                                  static_cast<const double *>(d_pos), static_cast<const double *>(d_time),
                                  static_cast<double *>(d_dist), nullptr, static_cast<hipStream_t>(hip_stream)));
  return GTOP_OK;
} GTOP_CATCH_STATUS(c)

int gtop_edt_coarse_query(gtop_ctx *c, int N, const double *pos, const double *time, double *dist) try {
  if (!c) return GTOP_ERR_INVALID;
  if (N < 0 || (N > 0 && (!pos || !time || !dist))) return fail(c, GTOP_ERR_INVALID, "edt_coarse_query: bad arguments");
  if (N == 0) return GTOP_OK;
  HIPCHK(c, hipSetDevice(c->device));
  int rc;
  const size_t n = (size_t)N;
  if ((rc = ensure(c, &c->d_q, &c->cap_q, 5 * n))) return rc;
  double *dp = c->d_q, *dt = dp + 3 * n, *dd = dt + n;
  HIPCHK(c, hipMemcpyAsync(dp, pos, 3 * n * sizeof(double), hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipMemcpyAsync(dt, time, n * sizeof(double), hipMemcpyHostToDevice, c->stream));
  if ((rc = gtop_edt_coarse_query_device(c, N, dp, dt, dd, c->stream))) return rc;
  HIPCHK(c, hipMemcpyAsync(dist, dd, n * sizeof(double), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return GTOP_OK;
} GTOP_CATCH_STATUS(c)

// Batched optimizer: max_evals rounds of {cost/gradient, MMA update} per trajectory on
// `stream` — one launch for the whole loop (fusion mode 2), one per round (1), or two
// per round (0); no host synchronisation inside.
int gtop_optimize_device_ex(gtop_ctx *c, int B, int m, void *d_x, const void *d_Df, const void *d_T,
                            int time_stride, const void *d_lb, const void *d_ub, const gtop_stop *stop, void *d_minf,
                            int32_t *d_nevals, int32_t *d_code, void *hip_stream) try {
  if (!c) return GTOP_ERR_INVALID;
  int rc = check_eval_state(c);
  if (rc) return rc;
  if (!stop || stop->max_evals < 1 || stop->ftol_rel < 0 || stop->xtol_rel < 0 || stop->maxtime < 0)
    return fail(c, GTOP_ERR_INVALID, "optimize: stop rules need max_evals >= 1 and non-negative tolerances / maxtime");
  const int max_evals = stop->max_evals;
  if (B < 0 || m < 2 || (time_stride != 0 && time_stride != m))
    return fail(c, GTOP_ERR_INVALID, "optimize_device: need B >= 0, m >= 2, time_stride in {0, m}");
  if (B == 0) return GTOP_OK;
  if (!d_x || !d_Df || !d_T || !d_lb || !d_ub) return fail(c, GTOP_ERR_INVALID, "optimize_device: NULL buffer");
  if (!c->rec64_ok) return fail(c, GTOP_ERR_STATE, "no fp64 distance field resident");
  HIPCHK(c, hipSetDevice(c->device));
  hipStream_t s = static_cast<hipStream_t>(hip_stream);
  const size_t n = 9 * (size_t)(m - 1), bn = (size_t)B * n;
  if ((rc = ensure(c, &c->mma_vec, &c->cap_mma_vec, 6 * bn))) return rc;
  if ((rc = ensure(c, &c->mma_scal, &c->cap_mma_scal, 5 * (size_t)B))) return rc;
  if ((rc = ensure(c, &c->mma_int, &c->cap_mma_int, 3 * (size_t)B))) return rc;
  if ((rc = ensure(c, &c->mma_f, &c->cap_mma_f, (size_t)B))) return rc;
  if ((rc = ensure(c, &c->mma_g, &c->cap_mma_g, bn))) return rc;
  GtopMmaState st;
  st.x = c->mma_vec; st.xcur = st.x + bn; st.xprev = st.xcur + bn; st.xprevprev = st.xprev + bn;
  st.dfdx = st.xprevprev + bn; st.sigma = st.dfdx + bn;
  st.lb = static_cast<const double *>(d_lb);
  st.ub = static_cast<const double *>(d_ub);
  st.rho = c->mma_scal; st.minf = st.rho + B; st.gval = st.minf + B; st.wval = st.gval + B; st.fprev = st.wval + B;
  st.k = c->mma_int; st.state = st.k + B; st.nevals = st.state + B;
  st.ftol_rel = stop->ftol_rel;
  st.xtol_rel = stop->xtol_rel;
  st.max_ticks = (long long)(stop->maxtime * 1e8);   // wall_clock64(): 100 MHz
  st.x0_init = nullptr;
  st.out_x = st.out_minf = nullptr;
  st.out_code = st.out_nevals = nullptr;
  // one geometry in every launch form; the whole loop in one launch by default (fusion mode 2)
  GtopEvalPlan plan;
  const bool f32 = c->opt_dtype == GTOP_F32;   // gtop_set_optimizer_precision: see below
  const size_t eval_elem = f32 ? sizeof(float) : sizeof(double);
  const int opt_spl = (c->spl == 30 || c->spl == 10) ? 0 : c->spl;   // (three lanes / one lane per segment: plain-evaluation geometries)
  bool planned = gtop_eval_plan(B, m, eval_elem, opt_spl, /*for_optimizer=*/true, &plan);
  // (two trajectories per wavefront with the velocity / acceleration block compiled in: the fp32 loop would spill —
  // enable_dyn keeps the loop at ten lanes per segment, one trajectory per wavefront)
  if (planned && plan.nt == 2 && c->prm.enable_dyn != 0) planned = gtop_eval_plan(B, m, eval_elem, 3, true, &plan);
  if (!planned)
    return fail(c, GTOP_ERR_INVALID, "optimize: this many segments cannot be served (ten lanes per segment: up to 6; "
                                     "one wavefront's LDS with the optimizer's state: 118)");
  const bool fused = c->fuse_mma != 0;
  const bool resident = c->fuse_mma == 2;   // one launch runs all max_evals evaluations of every trajectory
  st.iters = resident ? max_evals : 1;
  st.max_evals = max_evals;
  GtopKernelArgs<double> a;
  fill_args(c, a);
  a.sdf = c->rec64;
  a.x = st.xcur;
  a.Df = static_cast<const double *>(d_Df);
  a.T = static_cast<const double *>(d_T);
  a.cost = c->mma_f;
  a.grad = c->mma_g;
  a.B = B; a.m = m; a.t_stride = time_stride;
  const bool dyn = c->prm.enable_dyn != 0;
  // gtop_set_optimizer_precision(GTOP_F32): the same loop with its evaluations in fp32 on the fp32 field; the state,
  // the bounds, Df, T, the update and every result stay fp64 (the kernel converts as it reads its inputs from LDS)
  GtopKernelArgs<float> a32;
  if (f32) {
    if (!fused) return fail(c, GTOP_ERR_INVALID, "optimize: fp32 evaluations need a fused launch form (gtop_set_optimizer_fusion 1 or 2)");
    if ((rc = fp32_records_ready(c, s))) return rc;
    fill_args(c, a32);
    a32.sdf = c->rec32;
    a32.x = nullptr;   // (the loop reads its trial point from LDS)
    a32.Df = reinterpret_cast<const float *>(d_Df);   // fp64 rows: staged by the kernel as such
    a32.T = reinterpret_cast<const float *>(d_T);
    a32.cost = nullptr;
    a32.grad = nullptr;
    a32.B = B; a32.m = m; a32.t_stride = time_stride;
  }
  auto launch_loop = [&]() -> hipError_t {
    return f32 ? gtop_launch_eval_mma(a32, st, plan, dyn, s) : gtop_launch_eval_mma(a, st, plan, dyn, s);
  };
  // The whole optimisation as ONE launch: the loop initialises the state from d_x itself and writes the results where
  // they are wanted — no init kernel in front, no copies and no finish kernel behind (each a stream operation of its
  // own: 75 -> ~25 us of fixed cost per call).
  if (resident) {
    st.x0_init = static_cast<const double *>(d_x);
    // the kernel reads the start points row by row before it writes anything there, and a row is read and written
    // by the same wavefront: d_x can be the output as well
    st.out_x = static_cast<double *>(d_x);
    st.out_minf = static_cast<double *>(d_minf);
    st.out_code = d_code;
    st.out_nevals = d_nevals;
    HIPCHK(c, launch_loop());
    return GTOP_OK;
  }
  HIPCHK(c, gtop_launch_mma_init(st, B, (int)n, static_cast<const double *>(d_x), s));
  for (int it = 0; it < max_evals; ++it) {
    if (fused) {
      // one launch per iteration: the evaluation kernel runs the MMA update as its epilogue
      HIPCHK(c, launch_loop());
    } else {
      if ((rc = launch_eval<double>(c, c->rec64, B, m, st.xcur, d_Df, d_T, time_stride, c->mma_f, c->mma_g, s,
                                    &plan)))   // the geometry the fused modes run: same bits
        return rc;
      HIPCHK(c, gtop_launch_mma_update(st, B, (int)n, c->mma_f, c->mma_g, s));
    }
  }
  HIPCHK(c, hipMemcpyAsync(d_x, st.x, bn * sizeof(double), hipMemcpyDeviceToDevice, s));
  if (d_minf) HIPCHK(c, hipMemcpyAsync(d_minf, st.minf, (size_t)B * sizeof(double), hipMemcpyDeviceToDevice, s));
  HIPCHK(c, gtop_launch_mma_finish(st, B, d_code, d_nevals, s));
  return GTOP_OK;
} GTOP_CATCH_STATUS(c)

int gtop_optimize_device(gtop_ctx *c, int B, int m, void *d_x, const void *d_Df, const void *d_T,
                         int time_stride, const void *d_lb, const void *d_ub, int max_evals, void *d_minf,
                         void *hip_stream) try {
  const gtop_stop stop = {max_evals, 0.0, 0.0, 0.0};
  return gtop_optimize_device_ex(c, B, m, d_x, d_Df, d_T, time_stride, d_lb, d_ub, &stop, d_minf, nullptr, nullptr,
                                 hip_stream);
} GTOP_CATCH_STATUS(c)

int gtop_optimize_batch_ex(gtop_ctx *c, int B, double *x, const double *lb, const double *ub, const gtop_stop *stop,
                           double *min_cost, int32_t *nevals, int32_t *code) try {
  if (!c) return GTOP_ERR_INVALID;
  if (c->B == 0) return fail(c, GTOP_ERR_STATE, "gtop_set_problem has not been called");
  if (B < 1 || B > c->B || !x || !lb || !ub || !stop)
    return fail(c, GTOP_ERR_INVALID, "optimize_batch: 1 <= B <= problem batch, non-NULL buffers and stop rules");
  HIPCHK(c, hipSetDevice(c->device));
  const size_t n = 9 * (size_t)(c->m - 1), bn = (size_t)B * n;
  int rc;
  if ((rc = ensure(c, &c->mma_lb, &c->cap_mma_lb, bn))) return rc;
  if ((rc = ensure(c, &c->mma_ub, &c->cap_mma_ub, bn))) return rc;
  if ((rc = ensure(c, &c->mma_res, &c->cap_mma_res, 2 * (size_t)B))) return rc;
  HIPCHK(c, hipMemcpyAsync(c->d_x, x, bn * sizeof(double), hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipMemcpyAsync(c->mma_lb, lb, bn * sizeof(double), hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipMemcpyAsync(c->mma_ub, ub, bn * sizeof(double), hipMemcpyHostToDevice, c->stream));
  if ((rc = gtop_optimize_device_ex(c, B, c->m, c->d_x, c->d_Df, c->d_T, c->t_stride, c->mma_lb, c->mma_ub, stop,
                                    c->d_cost, c->mma_res, c->mma_res + B, c->stream)))
    return rc;
  HIPCHK(c, hipMemcpyAsync(x, c->d_x, bn * sizeof(double), hipMemcpyDeviceToHost, c->stream));
  if (min_cost)
    HIPCHK(c, hipMemcpyAsync(min_cost, c->d_cost, (size_t)B * sizeof(double), hipMemcpyDeviceToHost, c->stream));
  if (nevals)
    HIPCHK(c, hipMemcpyAsync(nevals, c->mma_res, (size_t)B * sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
  if (code)
    HIPCHK(c, hipMemcpyAsync(code, c->mma_res + B, (size_t)B * sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return GTOP_OK;
} GTOP_CATCH_STATUS(c)

int gtop_optimize_batch(gtop_ctx *c, int B, double *x, const double *lb, const double *ub, int max_evals,
                        double *min_cost) try {
  const gtop_stop stop = {max_evals, 0.0, 0.0, 0.0};
  return gtop_optimize_batch_ex(c, B, x, lb, ub, &stop, min_cost, nullptr, nullptr);
} GTOP_CATCH_STATUS(c)

// grad_traj_optimizer.cpp:151-179
int gtop_default_bounds(int B, int m, const double *path, double bos, double vos, double aos, double *lb,
                        double *ub) {
  if (B < 1 || m < 2 || !path || !lb || !ub) return GTOP_ERR_INVALID;
  const int num_dp = 3 * m - 3;
  const size_t n = 3 * (size_t)num_dp;
  for (int b = 0; b < B; ++b) {
    const double *p = path + (size_t)b * (m + 1) * 3;
    double *l = lb + (size_t)b * n, *u = ub + (size_t)b * n;
    for (int i = 0; i < num_dp; ++i)
      for (int a = 0; a < 3; ++a) {
        const size_t j = (size_t)i + (size_t)a * num_dp;
        if (i % 3 == 0) {
          l[j] = p[(i / 3 + 1) * 3 + a] - bos;
          u[j] = p[(i / 3 + 1) * 3 + a] + bos;
        } else if (i % 3 == 1) {
          l[j] = -vos;
          u[j] = vos;
        } else {
          l[j] = -aos;
          u[j] = aos;
        }
      }
  }
  return GTOP_OK;
}

int gtop_get_stats(const gtop_ctx *c, int64_t *iter_num, double *total_time) {
  if (!c) return GTOP_ERR_INVALID;
  if (iter_num) *iter_num = c->iter_num;
  if (total_time) *total_time = c->total_time;
  return GTOP_OK;
}

int gtop_reset_stats(gtop_ctx *c) try {
  if (!c) return GTOP_ERR_INVALID;
  c->iter_num = 0;
  c->total_time = 0.0;
  return GTOP_OK;
} GTOP_CATCH_STATUS(c)

int gtop_get_cost_curve(const gtop_ctx *c, double *cost, double *time, int cap, int *count) {
  if (!c) return GTOP_ERR_INVALID;
  const int nn = (int)c->vec_cost.size();
  if (count) *count = nn;
  const int k = cap < nn ? cap : nn;
  for (int i = 0; i < k; ++i) {
    if (cost) cost[i] = c->vec_cost[i];
    if (time) time[i] = c->vec_time[i];
  }
  return GTOP_OK;
}

int gtop_clear_cost_curve(gtop_ctx *c) try {
  if (!c) return GTOP_ERR_INVALID;
  c->vec_cost.clear();   // grad_traj_optimizer.cpp:192-194
  c->vec_time.clear();
  c->time_start = std::chrono::steady_clock::now();
  return GTOP_OK;
} GTOP_CATCH_STATUS(c)

// ---- measurement aid: the device's own clock around enqueued work ----
int gtop_device_clock_stamp(gtop_ctx *c, void *d_minmax, void *hip_stream) try {
  if (!c) return GTOP_ERR_INVALID;
  if (!d_minmax) return fail(c, GTOP_ERR_INVALID, "device_clock_stamp: NULL buffer");
  HIPCHK(c, hipSetDevice(c->device));
  HIPCHK(c, gtop_launch_clock_stamp(static_cast<unsigned long long *>(d_minmax), static_cast<hipStream_t>(hip_stream)));
  return GTOP_OK;
} GTOP_CATCH_STATUS(c)

// ---- SURVEY 8e's collective as point-to-point stores (gtop_push.hip) ----
int gtop_push_rows(gtop_ctx *c, const void *d_src, size_t bytes, void *const *d_dsts, int n_dsts, void *d_clock_minmax,
                   void *hip_stream) try {
  if (!c) return GTOP_ERR_INVALID;
  if (n_dsts < 0 || n_dsts > GTOP_PUSH_MAX_DSTS || (bytes > 0 && n_dsts > 0 && (!d_src || !d_dsts)))
    return fail(c, GTOP_ERR_INVALID, "push_rows: 0 .. 16 destinations, non-NULL buffers");
  if ((bytes == 0 || n_dsts == 0) && !d_clock_minmax) return GTOP_OK;
  GtopPushDsts dsts{};
  if (reinterpret_cast<uintptr_t>(d_src) & 15u) return fail(c, GTOP_ERR_INVALID, "push_rows: source not 16-byte aligned");
  for (int k = 0; k < n_dsts; ++k) {
    if (!d_dsts[k] || (reinterpret_cast<uintptr_t>(d_dsts[k]) & 15u))
      return fail(c, GTOP_ERR_INVALID, "push_rows: a destination is NULL or not 16-byte aligned");
    dsts.p[k] = d_dsts[k];
  }
  HIPCHK(c, hipSetDevice(c->device));
  HIPCHK(c, gtop_launch_push_rows(d_src, bytes, dsts, n_dsts, static_cast<unsigned long long *>(d_clock_minmax),
                                  static_cast<hipStream_t>(hip_stream)));
  return GTOP_OK;
} GTOP_CATCH_STATUS(c)

// Buffers another process can map: plain hipMalloc allocations (an IPC handle names a whole allocation: a
// sub-allocation of a caching allocator would not do) exported / opened with HIP's IPC calls.  A peer's buffer is
// opened with THIS context's device current and lazy peer access: the mapping is made for the device whose kernels
// will store into it (what RCCL's own point-to-point transport does).
int gtop_shared_alloc(gtop_ctx *c, size_t bytes, void **d_ptr, unsigned char handle[GTOP_IPC_HANDLE_BYTES]) try {
  if (!c) return GTOP_ERR_INVALID;
  static_assert(sizeof(hipIpcMemHandle_t) <= GTOP_IPC_HANDLE_BYTES, "handle size");
  if (!d_ptr || !handle || bytes == 0) return fail(c, GTOP_ERR_INVALID, "shared_alloc: bytes > 0, non-NULL outputs");
  *d_ptr = nullptr;
  HIPCHK(c, hipSetDevice(c->device));
  void *p = nullptr;
  HIPCHK(c, hipMalloc(&p, bytes));
  hipError_t e = hipMemset(p, 0, bytes);
  hipIpcMemHandle_t h;
  if (e == hipSuccess) e = hipIpcGetMemHandle(&h, p);
  if (e != hipSuccess) {
    (void)hipFree(p);
    HIPCHK(c, e);
  }
  std::memset(handle, 0, GTOP_IPC_HANDLE_BYTES);
  std::memcpy(handle, &h, sizeof(h));
  *d_ptr = p;
  return GTOP_OK;
} GTOP_CATCH_STATUS(c)

int gtop_shared_open(gtop_ctx *c, const unsigned char handle[GTOP_IPC_HANDLE_BYTES], int owner_device, void **d_ptr) try {
  if (!c) return GTOP_ERR_INVALID;
  if (!d_ptr || !handle) return fail(c, GTOP_ERR_INVALID, "shared_open: NULL argument");
  *d_ptr = nullptr;
  HIPCHK(c, hipSetDevice(c->device));
  // The owner's device, when the caller knows it: no mapping is made unless this device can reach that one (a kernel
  // storing through a mapping its GPU cannot reach faults the whole process), and peer access is switched on here
  // rather than left to the lazy flag alone.
  if (owner_device >= 0 && owner_device != c->device) {
    int ndev = 0, can = 0;
    HIPCHK(c, hipGetDeviceCount(&ndev));
    if (owner_device >= ndev) return fail(c, GTOP_ERR_INVALID, "shared_open: the owner's device ordinal is not visible to this process");
    HIPCHK(c, hipDeviceCanAccessPeer(&can, c->device, owner_device));
    if (!can) return fail(c, GTOP_ERR_STATE, "shared_open: this device has no peer access to the owner's device");
    const hipError_t pe = hipDeviceEnablePeerAccess(owner_device, 0);
    if (pe != hipSuccess && pe != hipErrorPeerAccessAlreadyEnabled) HIPCHK(c, pe);
    (void)hipGetLastError();   // (already enabled: not an error to keep)
  }
  hipIpcMemHandle_t h;
  std::memcpy(&h, handle, sizeof(h));
  void *p = nullptr;
  HIPCHK(c, hipIpcOpenMemHandle(&p, h, hipIpcMemLazyEnablePeerAccess));
  *d_ptr = p;
  return GTOP_OK;
} GTOP_CATCH_STATUS(c)

int gtop_shared_close(gtop_ctx *c, void *d_ptr) try {
  if (!c) return GTOP_ERR_INVALID;
  if (!d_ptr) return GTOP_OK;
  HIPCHK(c, hipSetDevice(c->device));
  HIPCHK(c, hipIpcCloseMemHandle(d_ptr));
  return GTOP_OK;
} GTOP_CATCH_STATUS(c)

int gtop_shared_free(gtop_ctx *c, void *d_ptr) try {
  if (!c) return GTOP_ERR_INVALID;
  if (!d_ptr) return GTOP_OK;
  HIPCHK(c, hipSetDevice(c->device));
  HIPCHK(c, hipFree(d_ptr));
  return GTOP_OK;
} GTOP_CATCH_STATUS(c)

int gtop_device_clock_hz(gtop_ctx *c, double *hz) try {
  if (!c) return GTOP_ERR_INVALID;
  if (!hz) return fail(c, GTOP_ERR_INVALID, "device_clock_hz: NULL");
  int khz = 0;
  HIPCHK(c, hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, c->device));
  if (khz <= 0) return fail(c, GTOP_ERR_HIP, "hipDeviceAttributeWallClockRate reports no wall clock");
  *hz = 1e3 * (double)khz;
  return GTOP_OK;
} GTOP_CATCH_STATUS(c)

int gtop_set_field_precisions(gtop_ctx *c, int keep_fp32) try {
  if (!c) return GTOP_ERR_INVALID;
  if (keep_fp32 != 0 && keep_fp32 != 1) return fail(c, GTOP_ERR_INVALID, "field precisions: 0 (fp64 records only) or 1 (fp32 records too)");
  c->fp32_wanted = keep_fp32 != 0;
  if (!c->fp32_wanted) {
    c->rec32_ok = false;
    c->rec32_stale = c->sdf64 != nullptr;   // (switched on again: rebuilt from the fp64 field at the first fp32 use)
  }
  return GTOP_OK;
} GTOP_CATCH_STATUS(c)

int gtop_set_optimizer_fusion(gtop_ctx *c, int fused) try {
  if (!c) return GTOP_ERR_INVALID;
  if (fused < 0 || fused > 2) return fail(c, GTOP_ERR_INVALID, "optimizer fusion mode is 0, 1 or 2");
  c->fuse_mma = fused;
  return GTOP_OK;
} GTOP_CATCH_STATUS(c)

int gtop_set_optimizer_precision(gtop_ctx *c, int dtype) try {
  if (!c) return GTOP_ERR_INVALID;
  if (dtype != GTOP_F64 && dtype != GTOP_F32) return fail(c, GTOP_ERR_INVALID, "optimizer precision is GTOP_F64 or GTOP_F32");
  c->opt_dtype = dtype;
  return GTOP_OK;
} GTOP_CATCH_STATUS(c)

int gtop_set_launch_geometry(gtop_ctx *c, int waves, int samples_per_lane) try {
  if (!c) return GTOP_ERR_INVALID;
  // one kernel family: a workgroup is one wavefront; the lanes-per-segment choice is what is left to pin
  if (waves != 0 && waves != 1) return fail(c, GTOP_ERR_INVALID, "waves per workgroup must be 0 (auto) or 1");
  const int s = samples_per_lane;
  if (s != 0 && s != 3 && s != 6 && s != 10 && s != 30)
    return fail(c, GTOP_ERR_INVALID, "samples per lane must be 0 (auto), 3 (ten lanes per segment, up to 6 segments), "
                                     "6 (five lanes per segment), 10 (three lanes per segment, up to 10 segments) or 30 "
                                     "(one lane per segment, up to 12 segments); 10 and 30 serve plain evaluations only: "
                                     "the optimizer loop keeps its own rule");
  c->spl = s;
  return GTOP_OK;
} GTOP_CATCH_STATUS(c)

}  // extern "C"
