// gtop_group.cpp — one batch over several GPUs from ONE host process (include/gtop.h, gtop_group_*).
//
// The reference has no multi-device code at all; its path is one NLopt instance per problem calling costFunc serially
// (src/grad_traj_optimizer.cpp:137-195, :554-562).  The batched callback shards trivially: trajectories
// are independent, the distance field is read-only.  A group owns one gtop_ctx and one HIP stream per device, keeps
// the field REPLICATED (built on every device from the same obstacle points), cuts the batch into contiguous slices of
// ceil(B / n) rows, launches every slice's evaluation on its own device without waiting in between, and collects the
// results either on the host (gtop_group_eval_batch) or on the devices: an all-gather of the costs (optionally the
// gradients) so that every device holds the whole batch's results (gtop_group_eval_resident).  There is no reduction
// anywhere, so results are bit-identical to the unsharded evaluation.
//
// The device-side all-gather uses RCCL (ncclAllGather inside a group call, one communicator per device, from
// librccl.so loaded on first use) when the group's devices are all different, and plain peer copies otherwise (RCCL
// refuses two ranks on one device; a group may list a device twice, e.g. to overlap two streams on one card or to
// test on a single-GPU box).  Pure host logic over the public C-ABI: a gtop_ctx is opaque here too.
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <new>
#include <set>
#include <string>
#include <vector>

#include "gtop.h"
#include "gtop_guard.h"

namespace {

struct Rccl {
  void *lib = nullptr;
  ncclResult_t (*CommInitAll)(ncclComm_t *, int, const int *) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*GroupStart)() = nullptr;
  ncclResult_t (*GroupEnd)() = nullptr;
  const char *(*GetErrorString)(ncclResult_t) = nullptr;
  bool load(std::string &err) {
    if (lib) return true;
    for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
      lib = dlopen(name, RTLD_NOW | RTLD_LOCAL);
      if (lib) break;
    }
    if (!lib) {
      err = std::string("librccl.so not found: ") + dlerror();
      return false;
    }
    CommInitAll = reinterpret_cast<decltype(CommInitAll)>(dlsym(lib, "ncclCommInitAll"));
    CommDestroy = reinterpret_cast<decltype(CommDestroy)>(dlsym(lib, "ncclCommDestroy"));
    AllGather = reinterpret_cast<decltype(AllGather)>(dlsym(lib, "ncclAllGather"));
    GroupStart = reinterpret_cast<decltype(GroupStart)>(dlsym(lib, "ncclGroupStart"));
    GroupEnd = reinterpret_cast<decltype(GroupEnd)>(dlsym(lib, "ncclGroupEnd"));
    GetErrorString = reinterpret_cast<decltype(GetErrorString)>(dlsym(lib, "ncclGetErrorString"));
    if (!CommInitAll || !CommDestroy || !AllGather || !GroupStart || !GroupEnd || !GetErrorString) {
      err = "librccl.so lacks an expected symbol";
      return false;
    }
    return true;
  }
};

struct Member {
  int device = 0;
  gtop_ctx *ctx = nullptr;
  hipStream_t stream = nullptr;
  hipEvent_t done = nullptr;     // this member's slice (and its outgoing copies) is complete
  hipEvent_t ready = nullptr;    // everything enqueued so far that may still read its gathered buffers has run
  double *pts = nullptr;         // obstacle points of the last map update
  size_t pts_cap = 0;            // (doubles)
  int first = 0, count = 0;      // its slice of the batch
  double *x = nullptr, *Df = nullptr, *T = nullptr, *cost = nullptr, *grad = nullptr;   // slice buffers, `per` rows
  double *cost_all = nullptr, *grad_all = nullptr;                                      // gathered: n * per rows
  double *lb = nullptr, *ub = nullptr;
  int32_t *nev = nullptr, *code = nullptr;
  // capacities (bytes) of the buffers above, in free_slices' order: a new problem reuses what is large enough, so a
  // caller that walks through problems of different sizes (GradTrajBatch: one per segment count) does not reallocate
  size_t cap[11] = {0};
  ncclComm_t comm = nullptr;
};

}  // namespace

struct gtop_group {
  std::vector<Member> mem;
  std::string err;
  int B = 0, m = 0, t_stride = 0, per = 0;   // per = rows per slice (the last may hold fewer)
  bool have_problem = false, use_rccl = false, x_resident = false;
  std::string gather_note;   // why the gather backend is what it is (gtop_group_gather_note)
  Rccl rccl;
};

namespace {

int gfail(gtop_group *g, int code, const std::string &msg) {
  if (g) g->err = msg;
  return code;
}

// error text of a failed gtop_group_create (there is no group to hold it): gtop_group_last_error(NULL) returns it
thread_local std::string g_group_create_err;

int gfail_create(int code, const std::string &msg) {
  g_group_create_err = msg;
  return code;
}

void note_exception(gtop_group *g, const char *what) noexcept {
  try {
    const std::string msg = std::string("exception caught at the C boundary: ") + what;
    if (g) g->err = msg;
    else g_group_create_err = msg;
  } catch (...) {
  }
}
#define GTOP_CATCH_STATUS(g) GTOP_CATCH_WITH(note_exception, static_cast<gtop_group *>(g), GTOP_ERR_INTERNAL)

#define GHIP(g, call)                                                                         \
  do {                                                                                        \
    hipError_t e_ = (call);                                                                   \
    if (e_ != hipSuccess) return gfail(g, GTOP_ERR_HIP, std::string(#call) + ": " + hipGetErrorString(e_)); \
  } while (0)
#define GCTX(g, mb, call)                                                                     \
  do {                                                                                        \
    int rc_ = (call);                                                                         \
    if (rc_ != GTOP_OK)                                                                       \
      return gfail(g, rc_, std::string(#call) + " on device " + std::to_string((mb).device) + ": " + gtop_last_error((mb).ctx)); \
  } while (0)

void free_slices(Member &mb) {
  (void)hipSetDevice(mb.device);
  for (void *p : {(void *)mb.x, (void *)mb.Df, (void *)mb.T, (void *)mb.cost, (void *)mb.grad, (void *)mb.cost_all,
                  (void *)mb.grad_all, (void *)mb.lb, (void *)mb.ub, (void *)mb.nev, (void *)mb.code})
    if (p) (void)hipFree(p);
  mb.x = mb.Df = mb.T = mb.cost = mb.grad = mb.cost_all = mb.grad_all = mb.lb = mb.ub = nullptr;
  mb.nev = mb.code = nullptr;
  for (size_t &c : mb.cap) c = 0;
}

// grow-only device buffer (the member's device is current)
template <typename P> hipError_t ensure(P *&p, size_t &cap, size_t bytes) {
  if (bytes <= cap && p) return hipSuccess;
  if (p) (void)hipFree(p);
  p = nullptr;
  cap = 0;
  const hipError_t e = hipMalloc(reinterpret_cast<void **>(&p), bytes ? bytes : 8);
  if (e == hipSuccess) cap = bytes ? bytes : 8;
  return e;
}

// results of every slice to every member.  RCCL: one all-gather per member inside a group call (the slices are
// padded to `per` rows, so the gathered layout [rank][per] IS the batch order).  Copies: each member sends its slice
// to every member's gathered buffer on its own stream, then every stream waits for every sender.
int all_gather(gtop_group *g, bool grads) {
  const int n = (int)g->mem.size();
  const size_t nvar = 9 * (size_t)(g->m - 1);
  if (g->use_rccl) {
    ncclResult_t r = g->rccl.GroupStart();
    for (int i = 0; i < n && r == ncclSuccess; ++i) {
      Member &mb = g->mem[i];
      r = g->rccl.AllGather(mb.cost, mb.cost_all, (size_t)g->per, ncclDouble, mb.comm, mb.stream);
      if (r == ncclSuccess && grads)
        r = g->rccl.AllGather(mb.grad, mb.grad_all, (size_t)g->per * nvar, ncclDouble, mb.comm, mb.stream);
    }
    const ncclResult_t r2 = g->rccl.GroupEnd();
    if (r != ncclSuccess || r2 != ncclSuccess)
      return gfail(g, GTOP_ERR_HIP, std::string("ncclAllGather: ") + g->rccl.GetErrorString(r != ncclSuccess ? r : r2));
    return GTOP_OK;
  }
  // a sender writes into the other members' gathered buffers: not before whatever those members had enqueued on
  // their own streams (a consumer of the previous gather, say) has run
  for (Member &mb : g->mem) {
    GHIP(g, hipSetDevice(mb.device));
    GHIP(g, hipEventRecord(mb.ready, mb.stream));
  }
  for (int i = 0; i < n; ++i) {
    Member &src = g->mem[i];
    GHIP(g, hipSetDevice(src.device));
    for (int j = 0; j < n; ++j)
      if (j != i) GHIP(g, hipStreamWaitEvent(src.stream, g->mem[j].ready, 0));
    for (int j = 0; j < n; ++j) {
      Member &dst = g->mem[j];
      GHIP(g, hipMemcpyPeerAsync(dst.cost_all + src.first, dst.device, src.cost, src.device,
                                 (size_t)src.count * sizeof(double), src.stream));
      if (grads)
        GHIP(g, hipMemcpyPeerAsync(dst.grad_all + (size_t)src.first * nvar, dst.device, src.grad, src.device,
                                   (size_t)src.count * nvar * sizeof(double), src.stream));
    }
    GHIP(g, hipEventRecord(src.done, src.stream));
  }
  for (int j = 0; j < n; ++j) {
    GHIP(g, hipSetDevice(g->mem[j].device));
    for (int i = 0; i < n; ++i)
      if (i != j) GHIP(g, hipStreamWaitEvent(g->mem[j].stream, g->mem[i].done, 0));
  }
  return GTOP_OK;
}

int sync_all(gtop_group *g) {
  for (Member &mb : g->mem) {
    GHIP(g, hipSetDevice(mb.device));
    GHIP(g, hipStreamSynchronize(mb.stream));
  }
  return GTOP_OK;
}

}  // namespace

extern "C" {

int gtop_group_create(gtop_group **out, const int *devices, int n_devices) try {
  if (!out) return gfail_create(GTOP_ERR_INVALID, "gtop_group_create: out is NULL");
  *out = nullptr;
  if (!devices || n_devices < 1 || n_devices > 64)
    return gfail_create(GTOP_ERR_INVALID, "gtop_group_create: need a device list of 1 .. 64 entries");
  gtop_group *g = new (std::nothrow) gtop_group();
  if (!g) return gfail_create(GTOP_ERR_INTERNAL, "gtop_group_create: out of memory");
  g->mem.resize(n_devices);
  std::set<int> distinct;
  for (int i = 0; i < n_devices; ++i) {
    Member &mb = g->mem[i];
    mb.device = devices[i];
    distinct.insert(devices[i]);
    int rc = gtop_create(&mb.ctx, mb.device);
    if (rc == GTOP_OK && hipSetDevice(mb.device) != hipSuccess) rc = GTOP_ERR_HIP;
    if (rc == GTOP_OK && hipStreamCreateWithFlags(&mb.stream, hipStreamNonBlocking) != hipSuccess) rc = GTOP_ERR_HIP;
    if (rc == GTOP_OK && hipEventCreateWithFlags(&mb.done, hipEventDisableTiming) != hipSuccess) rc = GTOP_ERR_HIP;
    if (rc == GTOP_OK && hipEventCreateWithFlags(&mb.ready, hipEventDisableTiming) != hipSuccess) rc = GTOP_ERR_HIP;
    if (rc != GTOP_OK) {
      const std::string why = std::string("gtop_group_create: member ") + std::to_string(i) + " (device " +
                              std::to_string(mb.device) + "): " +
                              (mb.ctx ? "stream / event creation failed" : gtop_last_error(nullptr));
      gtop_group_destroy(g);
      (void)hipGetLastError();   // (the runtime's error of the failed call must not surface at somebody's next launch)
      return gfail_create(rc, why);
    }
  }
  // peers: every pair of different devices that can reach each other directly (the copy path; RCCL finds its own way)
  for (int i = 0; i < n_devices; ++i)
    for (int j = 0; j < n_devices; ++j) {
      const int a = g->mem[i].device, b = g->mem[j].device;
      int can = 0;
      if (a != b && hipDeviceCanAccessPeer(&can, a, b) == hipSuccess && can) {
        (void)hipSetDevice(a);
        (void)hipDeviceEnablePeerAccess(b, 0);   // (already enabled: an error we do not care about)
        (void)hipGetLastError();
      }
    }
  // RCCL needs one rank per device; a device listed twice keeps the group on peer copies.  GTOP_GROUP_GATHER=copy
  // (environment) forces the copies, =rccl makes a missing librccl an error instead of a fallback.
  const char *want = std::getenv("GTOP_GROUP_GATHER");
  const bool force_copy = want && std::strcmp(want, "copy") == 0, force_rccl = want && std::strcmp(want, "rccl") == 0;
  if (!force_copy && (int)distinct.size() == n_devices) {
    std::string why;
    if (g->rccl.load(why)) {
      std::vector<ncclComm_t> comms(n_devices);
      const ncclResult_t r = g->rccl.CommInitAll(comms.data(), n_devices, devices);
      if (r == ncclSuccess) {
        for (int i = 0; i < n_devices; ++i) g->mem[i].comm = comms[i];
        g->use_rccl = true;
      } else {
        why = std::string("ncclCommInitAll: ") + g->rccl.GetErrorString(r);
      }
    }
    if (!g->use_rccl && force_rccl) {
      gtop_group_destroy(g);
      return gfail_create(GTOP_ERR_HIP, "gtop_group_create: GTOP_GROUP_GATHER=rccl, but " + why);
    }
    // a silent fallback hides a broken RCCL installation: the reason stays readable (gtop_group_gather_note)
    g->gather_note = g->use_rccl ? "rccl: one communicator per device (ncclCommInitAll)"
                                 : "copy: peer copies, because " + why;
  } else if (force_rccl) {
    gtop_group_destroy(g);
    // a device listed twice cannot have an RCCL communicator
    return gfail_create(GTOP_ERR_INVALID, "gtop_group_create: GTOP_GROUP_GATHER=rccl needs every listed device to be different");
  } else {
    g->gather_note = force_copy ? "copy: peer copies, forced by GTOP_GROUP_GATHER=copy"
                                : "copy: peer copies, because a device is listed more than once (RCCL wants one rank per device)";
  }
  *out = g;
  return GTOP_OK;
} GTOP_CATCH_STATUS(nullptr)

int gtop_group_destroy(gtop_group *g) try {
  if (!g) return GTOP_ERR_INVALID;
  for (Member &mb : g->mem) {
    if (!mb.stream) continue;
    (void)hipSetDevice(mb.device);
    (void)hipStreamSynchronize(mb.stream);
  }
  for (Member &mb : g->mem) {
    if (!mb.ctx) continue;   // (a member whose creation failed: its device ordinal may not even exist)
    if (mb.comm && g->rccl.CommDestroy) (void)g->rccl.CommDestroy(mb.comm);
    free_slices(mb);
    if (mb.pts) (void)hipFree(mb.pts);
    if (mb.done) (void)hipEventDestroy(mb.done);
    if (mb.ready) (void)hipEventDestroy(mb.ready);
    if (mb.stream) (void)hipStreamDestroy(mb.stream);
    if (mb.ctx) (void)gtop_destroy(mb.ctx);
  }
  delete g;
  return GTOP_OK;
} GTOP_CATCH_STATUS(g)

int gtop_group_size(const gtop_group *g) { return g ? (int)g->mem.size() : 0; }
gtop_ctx *gtop_group_context(gtop_group *g, int i) {
  return (g && i >= 0 && i < (int)g->mem.size()) ? g->mem[i].ctx : nullptr;
}
const char *gtop_group_last_error(const gtop_group *g) { return g ? g->err.c_str() : g_group_create_err.c_str(); }
const char *gtop_group_gather_backend(const gtop_group *g) { return (g && g->use_rccl) ? "rccl" : "copy"; }
const char *gtop_group_gather_note(const gtop_group *g) { return g ? g->gather_note.c_str() : ""; }

int gtop_group_set_params(gtop_group *g, const gtop_params *p) try {
  if (!g) return GTOP_ERR_INVALID;
  for (Member &mb : g->mem) GCTX(g, mb, gtop_set_params(mb.ctx, p));
  return GTOP_OK;
} GTOP_CATCH_STATUS(g)

int gtop_group_init_sdf_map(gtop_group *g, const double map_size[3], const double origin[3], double resolution) try {
  if (!g) return GTOP_ERR_INVALID;
  for (Member &mb : g->mem) GCTX(g, mb, gtop_init_sdf_map(mb.ctx, map_size, origin, resolution));
  return GTOP_OK;
} GTOP_CATCH_STATUS(g)

int gtop_group_update_sdf_map(gtop_group *g, const double *pts, int npts) try {
  if (!g) return GTOP_ERR_INVALID;
  if (npts < 0 || (npts > 0 && !pts)) return gfail(g, GTOP_ERR_INVALID, "bad obstacle list");
  // replicated: every device builds the field from the same points, all of them at the same time (the upload and the
  // build are enqueued on each member's stream before any is waited for)
  const size_t need = (size_t)npts * 3;
  for (Member &mb : g->mem) {
    GHIP(g, hipSetDevice(mb.device));
    if (need > mb.pts_cap) {
      if (mb.pts) GHIP(g, hipFree(mb.pts));
      mb.pts = nullptr;
      mb.pts_cap = 0;
      GHIP(g, hipMalloc(reinterpret_cast<void **>(&mb.pts), need * sizeof(double)));
      mb.pts_cap = need;
    }
    if (need) GHIP(g, hipMemcpyAsync(mb.pts, pts, need * sizeof(double), hipMemcpyHostToDevice, mb.stream));
    GCTX(g, mb, gtop_update_sdf_map_device(mb.ctx, mb.pts, npts, mb.stream));
  }
  return sync_all(g);
} GTOP_CATCH_STATUS(g)

// the reference's local update (gtop_update_sdf_map_window) on every member: the replicated fields stay identical
int gtop_group_update_sdf_map_window(gtop_group *g, const double min_pos[3], const double max_pos[3], const double *pts,
                                     int npts) try {
  if (!g) return GTOP_ERR_INVALID;
  if (!min_pos || !max_pos || npts < 0 || (npts > 0 && !pts)) return gfail(g, GTOP_ERR_INVALID, "bad window / obstacle list");
  const size_t need = (size_t)npts * 3;
  for (Member &mb : g->mem) {
    GHIP(g, hipSetDevice(mb.device));
    if (need > mb.pts_cap) {
      if (mb.pts) GHIP(g, hipFree(mb.pts));
      mb.pts = nullptr;
      mb.pts_cap = 0;
      GHIP(g, hipMalloc(reinterpret_cast<void **>(&mb.pts), need * sizeof(double)));
      mb.pts_cap = need;
    }
    if (need) GHIP(g, hipMemcpyAsync(mb.pts, pts, need * sizeof(double), hipMemcpyHostToDevice, mb.stream));
    GCTX(g, mb, gtop_update_sdf_map_window_device(mb.ctx, min_pos, max_pos, mb.pts, npts, mb.stream));
  }
  return sync_all(g);
} GTOP_CATCH_STATUS(g)

int gtop_group_set_sdf(gtop_group *g, const double *dist_host, int nx, int ny, int nz, const double origin[3],
                       const double *map_size, double resolution) try {
  if (!g) return GTOP_ERR_INVALID;
  for (Member &mb : g->mem) GCTX(g, mb, gtop_set_sdf(mb.ctx, dist_host, nx, ny, nz, origin, map_size, resolution));
  return GTOP_OK;
} GTOP_CATCH_STATUS(g)

int gtop_group_set_problem(gtop_group *g, int B, int m, const double *segment_time, int time_stride, const double *Df) try {
  if (!g) return GTOP_ERR_INVALID;
  if (B < 1 || m < 2 || !segment_time || !Df || (time_stride != 0 && time_stride != m))
    return gfail(g, GTOP_ERR_INVALID, "group set_problem: need B >= 1, m >= 2, time_stride in {0, m}");
  const int n = (int)g->mem.size();
  const size_t nvar = 9 * (size_t)(m - 1);
  const int per = (B + n - 1) / n;
  g->have_problem = false;
  g->x_resident = false;
  // nothing of the previous problem may still be running on the buffers about to be reused or replaced — a member's
  // own work, or another member's copies into its gathered rows
  if (int rc0 = sync_all(g)) return rc0;
  for (int i = 0; i < n; ++i) {
    Member &mb = g->mem[i];
    mb.first = std::min(B, i * per);
    mb.count = std::min(B, mb.first + per) - mb.first;
    GHIP(g, hipSetDevice(mb.device));
    // slices padded to `per` rows (RCCL's all-gather sends equal counts); the padding is never read as a result
    GHIP(g, ensure(mb.x, mb.cap[0], (size_t)per * nvar * sizeof(double)));
    GHIP(g, ensure(mb.Df, mb.cap[1], (size_t)per * 18 * sizeof(double)));
    GHIP(g, ensure(mb.T, mb.cap[2], (time_stride ? (size_t)per * m : (size_t)m) * sizeof(double)));
    GHIP(g, ensure(mb.cost, mb.cap[3], (size_t)per * sizeof(double)));
    GHIP(g, ensure(mb.grad, mb.cap[4], (size_t)per * nvar * sizeof(double)));
    GHIP(g, ensure(mb.cost_all, mb.cap[5], (size_t)n * per * sizeof(double)));
    GHIP(g, ensure(mb.grad_all, mb.cap[6], (size_t)n * per * nvar * sizeof(double)));
    GHIP(g, hipMemsetAsync(mb.cost, 0, (size_t)per * sizeof(double), mb.stream));
    GHIP(g, hipMemsetAsync(mb.grad, 0, (size_t)per * nvar * sizeof(double), mb.stream));
    if (mb.count > 0) {
      GHIP(g, hipMemcpyAsync(mb.Df, Df + (size_t)mb.first * 18, (size_t)mb.count * 18 * sizeof(double),
                             hipMemcpyHostToDevice, mb.stream));
      if (time_stride)
        GHIP(g, hipMemcpyAsync(mb.T, segment_time + (size_t)mb.first * m, (size_t)mb.count * m * sizeof(double),
                               hipMemcpyHostToDevice, mb.stream));
    }
    if (!time_stride)
      GHIP(g, hipMemcpyAsync(mb.T, segment_time, (size_t)m * sizeof(double), hipMemcpyHostToDevice, mb.stream));
  }
  int rc = sync_all(g);
  if (rc) return rc;
  g->B = B; g->m = m; g->t_stride = time_stride; g->per = per;
  g->have_problem = true;
  return GTOP_OK;
} GTOP_CATCH_STATUS(g)

int gtop_group_shard(const gtop_group *g, int i, int *first, int *count) {
  if (!g || !g->have_problem || i < 0 || i >= (int)g->mem.size()) return GTOP_ERR_INVALID;
  if (first) *first = g->mem[i].first;
  if (count) *count = g->mem[i].count;
  return GTOP_OK;
}

// every slice's evaluation enqueued on its own device before any is waited for
static int launch_slices(gtop_group *g) {
  for (Member &mb : g->mem) {
    if (mb.count == 0) continue;
    GCTX(g, mb, gtop_eval_device(mb.ctx, GTOP_F64, mb.count, g->m, mb.x, mb.Df, mb.T, g->t_stride, mb.cost, mb.grad,
                                 mb.stream));
  }
  return GTOP_OK;
}

int gtop_group_eval_batch(gtop_group *g, int B, const double *x, double *cost, double *grad) try {
  if (!g) return GTOP_ERR_INVALID;
  if (!g->have_problem) return gfail(g, GTOP_ERR_STATE, "gtop_group_set_problem has not been called");
  if (B != g->B || !x || !cost || !grad) return gfail(g, GTOP_ERR_INVALID, "group eval_batch: B must be the problem's batch");
  const size_t nvar = 9 * (size_t)(g->m - 1);
  for (Member &mb : g->mem) {
    if (mb.count == 0) continue;
    GHIP(g, hipSetDevice(mb.device));
    GHIP(g, hipMemcpyAsync(mb.x, x + (size_t)mb.first * nvar, (size_t)mb.count * nvar * sizeof(double),
                           hipMemcpyHostToDevice, mb.stream));
  }
  int rc = launch_slices(g);
  if (rc) return rc;
  for (Member &mb : g->mem) {
    if (mb.count == 0) continue;
    GHIP(g, hipSetDevice(mb.device));
    GHIP(g, hipMemcpyAsync(cost + mb.first, mb.cost, (size_t)mb.count * sizeof(double), hipMemcpyDeviceToHost, mb.stream));
    GHIP(g, hipMemcpyAsync(grad + (size_t)mb.first * nvar, mb.grad, (size_t)mb.count * nvar * sizeof(double),
                           hipMemcpyDeviceToHost, mb.stream));
  }
  g->x_resident = true;
  return sync_all(g);
} GTOP_CATCH_STATUS(g)

int gtop_group_upload_x(gtop_group *g, int B, const double *x) try {
  if (!g) return GTOP_ERR_INVALID;
  if (!g->have_problem) return gfail(g, GTOP_ERR_STATE, "gtop_group_set_problem has not been called");
  if (B != g->B || !x) return gfail(g, GTOP_ERR_INVALID, "group upload_x: B must be the problem's batch");
  const size_t nvar = 9 * (size_t)(g->m - 1);
  for (Member &mb : g->mem) {
    if (mb.count == 0) continue;
    GHIP(g, hipSetDevice(mb.device));
    GHIP(g, hipMemcpyAsync(mb.x, x + (size_t)mb.first * nvar, (size_t)mb.count * nvar * sizeof(double),
                           hipMemcpyHostToDevice, mb.stream));
  }
  g->x_resident = true;
  return sync_all(g);
} GTOP_CATCH_STATUS(g)

int gtop_group_eval_resident(gtop_group *g, int gather, int synchronize) try {
  if (!g) return GTOP_ERR_INVALID;
  if (!g->have_problem || !g->x_resident) return gfail(g, GTOP_ERR_STATE, "group eval_resident: set the problem and upload x first");
  if (gather < 0 || gather > 2) return gfail(g, GTOP_ERR_INVALID, "gather: 0 none, 1 costs, 2 costs and gradients");
  int rc = launch_slices(g);
  if (rc) return rc;
  if (gather && (rc = all_gather(g, gather == 2))) return rc;
  return synchronize ? sync_all(g) : GTOP_OK;
} GTOP_CATCH_STATUS(g)

int gtop_group_synchronize(gtop_group *g) { return g ? sync_all(g) : GTOP_ERR_INVALID; }

int gtop_group_read_gathered(gtop_group *g, int member, double *cost, double *grad) try {
  if (!g || !g->have_problem || member < 0 || member >= (int)g->mem.size()) return GTOP_ERR_INVALID;
  Member &mb = g->mem[member];
  const size_t nvar = 9 * (size_t)(g->m - 1);
  GHIP(g, hipSetDevice(mb.device));
  if (cost) GHIP(g, hipMemcpyAsync(cost, mb.cost_all, (size_t)g->B * sizeof(double), hipMemcpyDeviceToHost, mb.stream));
  if (grad)
    GHIP(g, hipMemcpyAsync(grad, mb.grad_all, (size_t)g->B * nvar * sizeof(double), hipMemcpyDeviceToHost, mb.stream));
  GHIP(g, hipStreamSynchronize(mb.stream));
  return GTOP_OK;
} GTOP_CATCH_STATUS(g)

int gtop_group_device_buffers(gtop_group *g, int member, void **d_x, void **d_cost, void **d_grad, void **d_cost_all,
                              void **d_grad_all, void **hip_stream) try {
  if (!g || !g->have_problem || member < 0 || member >= (int)g->mem.size()) return GTOP_ERR_INVALID;
  Member &mb = g->mem[member];
  if (d_x) *d_x = mb.x;
  if (d_cost) *d_cost = mb.cost;
  if (d_grad) *d_grad = mb.grad;
  if (d_cost_all) *d_cost_all = mb.cost_all;
  if (d_grad_all) *d_grad_all = mb.grad_all;
  if (hip_stream) *hip_stream = mb.stream;
  g->x_resident = true;   // the caller writes x where it lives
  return GTOP_OK;
} GTOP_CATCH_STATUS(g)

int gtop_group_optimize_batch_ex(gtop_group *g, int B, double *x, const double *lb, const double *ub,
                                 const gtop_stop *stop, double *min_cost, int32_t *nevals, int32_t *code) try {
  if (!g) return GTOP_ERR_INVALID;
  if (!g->have_problem) return gfail(g, GTOP_ERR_STATE, "gtop_group_set_problem has not been called");
  if (B != g->B || !x || !lb || !ub || !stop) return gfail(g, GTOP_ERR_INVALID, "group optimize: B must be the problem's batch");
  const size_t nvar = 9 * (size_t)(g->m - 1);
  for (Member &mb : g->mem) {
    if (mb.count == 0) continue;
    GHIP(g, hipSetDevice(mb.device));
    const size_t bytes = (size_t)mb.count * nvar * sizeof(double);
    GHIP(g, ensure(mb.lb, mb.cap[7], (size_t)g->per * nvar * sizeof(double)));
    GHIP(g, ensure(mb.ub, mb.cap[8], (size_t)g->per * nvar * sizeof(double)));
    GHIP(g, ensure(mb.nev, mb.cap[9], (size_t)g->per * sizeof(int32_t)));
    GHIP(g, ensure(mb.code, mb.cap[10], (size_t)g->per * sizeof(int32_t)));
    GHIP(g, hipMemcpyAsync(mb.x, x + (size_t)mb.first * nvar, bytes, hipMemcpyHostToDevice, mb.stream));
    GHIP(g, hipMemcpyAsync(mb.lb, lb + (size_t)mb.first * nvar, bytes, hipMemcpyHostToDevice, mb.stream));
    GHIP(g, hipMemcpyAsync(mb.ub, ub + (size_t)mb.first * nvar, bytes, hipMemcpyHostToDevice, mb.stream));
  }
  for (Member &mb : g->mem) {   // every device's whole optimisation is one launch; none waits for another
    if (mb.count == 0) continue;
    GCTX(g, mb, gtop_optimize_device_ex(mb.ctx, mb.count, g->m, mb.x, mb.Df, mb.T, g->t_stride, mb.lb, mb.ub, stop,
                                        mb.cost, mb.nev, mb.code, mb.stream));
  }
  for (Member &mb : g->mem) {
    if (mb.count == 0) continue;
    GHIP(g, hipSetDevice(mb.device));
    GHIP(g, hipMemcpyAsync(x + (size_t)mb.first * nvar, mb.x, (size_t)mb.count * nvar * sizeof(double),
                           hipMemcpyDeviceToHost, mb.stream));
    if (min_cost)
      GHIP(g, hipMemcpyAsync(min_cost + mb.first, mb.cost, (size_t)mb.count * sizeof(double), hipMemcpyDeviceToHost, mb.stream));
    if (nevals)
      GHIP(g, hipMemcpyAsync(nevals + mb.first, mb.nev, (size_t)mb.count * sizeof(int32_t), hipMemcpyDeviceToHost, mb.stream));
    if (code)
      GHIP(g, hipMemcpyAsync(code + mb.first, mb.code, (size_t)mb.count * sizeof(int32_t), hipMemcpyDeviceToHost, mb.stream));
  }
  g->x_resident = true;
  return sync_all(g);
} GTOP_CATCH_STATUS(g)

}  // extern "C"
