/*
 * gtop.h — C-ABI of the MI355X-native batched cost/gradient path of GTOP.
 *
 * This is the drop-in boundary.  Each entry point names the
 * reference interface it replaces as file:line into the reference tree
 * (EpicOne1/grad_traj_optimization).  Plain pointers and sizes only; no C++
 * or torch types; no exceptions cross this boundary (every entry point that can
 * allocate is a function-try-block, csrc/gtop_guard.h: whatever is thrown inside
 * comes back as GTOP_ERR_INTERNAL) — every function returns a gtop_status
 * (0 = OK) except gtop_cost_nlopt / gtop_cost_nlopt_shared, whose shape is fixed
 * by NLopt's `nlopt_func` (they return HUGE_VAL).
 *
 * Conventions
 *   m        number of polynomial segments (= #waypoints - 1), m >= 2
 *   n        free variables per trajectory = 9 (m - 1)
 *   x, grad  n values per trajectory, axis-major: x[i + axis*(3m-3)]
 *            (src/grad_traj_optimizer.cpp:182-187, :428-432)
 *   Df       3 x 6 per trajectory, row-major: per axis
 *            [p_start, v_start, a_start, p_end, v_end, a_end]
 *            (src/qp_generator.cpp:407-431)
 *   T        segment times, m per trajectory (src/grad_traj_optimizer.cpp:73-81)
 *   dist     voxel distance field, index x*ny*nz + y*nz + z
 *            (src/sdf_map.cpp:172-173)
 *
 * One gtop_ctx per host thread / HIP stream; a context is not thread-safe
 * (neither is the reference object it replaces: it mutates iter_num,
 * total_time and the cost curve on every call,
 * src/grad_traj_optimizer.cpp:284,436,439-447).
 */
#ifndef GTOP_H_
#define GTOP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct gtop_ctx gtop_ctx;

typedef enum {
  GTOP_OK = 0,
  GTOP_ERR_INVALID = 1,   /* bad argument (NULL, m < 2, size mismatch ...) */
  GTOP_ERR_HIP = 2,       /* a HIP runtime call failed; see gtop_last_error */
  GTOP_ERR_NO_DEVICE = 3, /* no gfx950 device visible */
  GTOP_ERR_STATE = 4,     /* call order: SDF / problem / params not set */
  GTOP_ERR_INTERNAL = 5   /* a C++ exception (e.g. std::bad_alloc) was caught at this
                             boundary; see gtop_last_error.  The object stays usable. */
} gtop_status;

typedef enum { GTOP_F64 = 0, GTOP_F32 = 1 } gtop_dtype;

/* The ROS parameters the callback reads (ctor,
 * src/grad_traj_optimizer.cpp:5-32) plus `step` (:132, :413-415).
 * enable_dyn switches on the velocity/acceleration penalty block that is
 * commented out in the reference (:383-407); 0 = behave as shipped. */
typedef struct {
  double ws, wc;           /* w_smooth, w_collision  */
  double alpha, r, d0;     /* distance penalty       */
  double alpha_v, r_v, v0; /* velocity penalty       */
  double alpha_a, r_a, a0; /* acceleration penalty   */
  int32_t step;            /* OPT_INITIAL_TRY 0 / OPT_FIRST_STEP 1 / OPT_SECOND_STEP 2 */
  int32_t enable_dyn;
} gtop_params;

/* ---- lifetime ------------------------------------------------------- */

/* Replaces GradTrajOptimizer::GradTrajOptimizer()
 * (src/grad_traj_optimizer.cpp:3-33).  `device` is the HIP device ordinal.
 * Fails with GTOP_ERR_NO_DEVICE when no GPU is present: there is no CPU
 * fallback behind this ABI. */
int gtop_create(gtop_ctx **out, int device);
int gtop_destroy(gtop_ctx *ctx);
/* Text of the last error on this context ("" if none); with ctx == NULL,
 * the text of the calling thread's last failed gtop_create. */
const char *gtop_last_error(const gtop_ctx *ctx);
/* Library/ABI version, for the loader to check (2 since round 4: the windowed map
 * update, gtop_set_field_precisions, gtop_device_clock_*, gtop_group_gather_note,
 * GTOP_ERR_INTERNAL; nothing of version 1 changed meaning). */
int gtop_abi_version(void);

/* ---- configuration -------------------------------------------------- */

/* Replaces the 21 ros::param::get calls of the ctor
 * (src/grad_traj_optimizer.cpp:5-32) and `this->step = step` (:132). */
int gtop_set_params(gtop_ctx *ctx, const gtop_params *p);

/* Replaces GradTrajOptimizer::initSDFMap + the distance_buffer that
 * updateSDFMap leaves behind (src/grad_traj_optimizer.cpp:112-126,
 * src/sdf_map.cpp:3-24).  `dist_host` holds nx*ny*nz doubles; it is uploaded
 * to HBM; the corner records every lookup reads (csrc/gtop_records.hip: the
 * 8 corners of a cell as 64 contiguous bytes, border clamps applied, 4x the
 * field's bytes) are derived from it on the device, fp32 ones at the first
 * fp32 use.
 * `map_size` may be NULL (then max_range = origin + grid*res); when given,
 * max_range = origin + map_size as src/sdf_map.cpp:12 has it.
 * Requires nx, ny, nz >= 2 and nx*ny*nz < 2^31. */
int gtop_set_sdf(gtop_ctx *ctx, const double *dist_host, int nx, int ny, int nz,
                 const double origin[3], const double *map_size,
                 double resolution);

/* Same, from a distance field already resident in HBM (dtype says which; the
 * caller has synchronised whatever wrote it).  The buffer is borrowed as the
 * boundary copy — gtop_get_sdf and the coarse voxel query read it in place, so
 * it must outlive its use — and the gather-friendly corner records the lookups
 * read (csrc/gtop_records.hip) are derived from it AT THIS CALL: after writing
 * to the buffer, call again.  A GTOP_F64 field serves evaluations of both
 * precisions, a GTOP_F32 field fp32 evaluations only. */
int gtop_set_sdf_device(gtop_ctx *ctx, int dtype, const void *dist_dev, int nx,
                        int ny, int nz, const double origin[3],
                        const double *map_size, double resolution);

/* Replaces GradTrajOptimizer::updateSDFMap (src/grad_traj_optimizer.cpp:117-126
 * -> src/sdf_map.cpp:26-53, :80-99, :310-368): reset, mark the voxel under
 * each of the `npts` obstacle points (xyz triples), rebuild the Euclidean
 * distance field on the device and keep it resident.  Call after an SDF
 * geometry has been set with gtop_init_sdf_map. */
int gtop_init_sdf_map(gtop_ctx *ctx, const double map_size[3],
                      const double origin[3], double resolution);
int gtop_update_sdf_map(gtop_ctx *ctx, const double *obstacle_pts, int npts);
/* The same with the obstacle points (npts x 3 doubles, xyz-contiguous) already in
 * HBM: launches on `hip_stream` and returns without synchronising — the build is
 * five kernels, 0.08 ms for a 200^3 map, against 0.6 ms of PCIe for the points of
 * the host form.  The corner records of BOTH precisions are rebuilt behind the
 * sweeps on the same stream, so the call can be captured into a hipGraph and
 * replayed with fp64 or fp32 evaluations behind it. */
int gtop_update_sdf_map_device(gtop_ctx *ctx, const void *d_obstacle_pts, int npts, void *hip_stream);
/* The reference's LOCAL map update — what compare2.cpp:147-152 does per sensor frame:
 *   sdf_map.resetBuffer(min_pos, max_pos)     src/sdf_map.cpp:28-53
 *   sdf_map.setOccupancy(p) for every point   :80-99
 *   [setUpdateRange(min_pos, max_pos)]        :244-264
 *   sdf_map.updateESDF3d()                    :310-368, its loops over min_vec .. max_vec
 * with its semantics: the box is clamped to the map and turned into voxel indices as
 * the reference does (posToIndex(min_pos) .. posToIndex(max_pos - res/2)); inside
 * it occupancy is cleared and distances reset to 10000; the points are marked
 * wherever in the map they fall; the three sweeps run over the box only and see
 * only the box's part of every line (an obstacle outside casts no distance into
 * it); distances outside the box keep their values, inside they become
 * min(res*sqrt(val), 10000).  Only the corner records of the box are rebuilt, so
 * the cost follows the box, not the map.  A box that covers the whole map takes
 * the whole-grid builder (same results).  Occupancy persists between calls, as in
 * the reference.  The _device form takes the points in HBM and only enqueues;
 * the first window update after gtop_init_sdf_map allocates scratch for any
 * window of that map (make it outside a stream capture), later ones allocate
 * nothing and can be captured. */
int gtop_update_sdf_map_window(gtop_ctx *ctx, const double min_pos[3], const double max_pos[3],
                               const double *obstacle_pts, int npts);
int gtop_update_sdf_map_window_device(gtop_ctx *ctx, const double min_pos[3], const double max_pos[3],
                                      const void *d_obstacle_pts, int npts, void *hip_stream);
/* Copy the resident fp64 distance field back to the host (nx*ny*nz doubles). */
int gtop_get_sdf(gtop_ctx *ctx, double *dist_host, int grid_out[3]);

/* Replaces the problem state setPath/setKinoPath leave in the object
 * (segment_time, Df; L and R follow from segment_time and are formed on the
 * device: src/grad_traj_optimizer.cpp:67-110, src/qp_generator.cpp:357-405).
 * B trajectories of m segments each; time_stride = m (per-trajectory times)
 * or 0 (one shared time vector).  Host pointers; uploaded to HBM. */
int gtop_set_problem(gtop_ctx *ctx, int B, int m, const double *segment_time,
                     int time_stride, const double *Df);

/* Replaces GradTrajOptimizer::setPath for B waypoint lists at once
 * (src/grad_traj_optimizer.cpp:67-110): segment times (:73-81: length/mean_v,
 * + init_time on the first segment only), Df and the straight-line initial Dp
 * (src/qp_generator.cpp:199-221, :407-451), computed on the device.
 * waypoints: B x (m+1) x 3.  gtop_set_paths makes the result the context's
 * problem (as gtop_set_problem would) and returns the start point x0 (B*n, may
 * be NULL); gtop_setup_paths_device writes device buffers and touches no state.
 * gtop_get_problem reads segment_time (B*m, or m when shared) and Df (B*18) back.
 * gtop_set_paths rejects coincident consecutive waypoints (a segment time of 0:
 * GTOP_ERR_INVALID, the rule gtop_set_problem applies); gtop_setup_paths_device
 * cannot look at device memory without a synchronisation and behaves as the
 * reference does there (T_s = 0, a singular A_s: NaN cost and gradient). */
int gtop_set_paths(gtop_ctx *ctx, int B, int m, const double *waypoints,
                   double mean_v, double init_time, double *x0);
int gtop_setup_paths_device(gtop_ctx *ctx, int B, int m, const void *d_waypoints,
                            double mean_v, double init_time, void *d_T,
                            void *d_Df, void *d_x0, void *hip_stream);
int gtop_get_problem(gtop_ctx *ctx, double *segment_time, double *Df);

/* ---- evaluation ----------------------------------------------------- */

/* Batched form of GradTrajOptimizer::costFunc / getCostAndGradient
 * (src/grad_traj_optimizer.cpp:554-562, :281-448) for the problem set by
 * gtop_set_problem: x, grad are B*n host doubles, cost B host doubles.
 * fp64 on the device; includes the PCIe copies.  Small batches (up to 16384
 * outputs: the NLopt callback, a rendezvous generation) are read from and written to pinned host
 * memory by the kernel itself, and the call returns when the last output has
 * landed there (each is stored once; the slots are preset to a NaN pattern no
 * evaluation produces).  GTOP_POLL_COMPLETION=0 in the environment at
 * gtop_create, or 2 ms without completion, wait through the stream instead. */
int gtop_eval_batch(gtop_ctx *ctx, int B, const double *x, double *cost,
                    double *grad);

/* Exactly NLopt's `nlopt_func` (what nlopt::opt::set_min_objective's
 * trampoline calls; src/grad_traj_optimizer.cpp:140, :554-562): evaluates
 * trajectory 0 of the current problem.  `grad` may be NULL (the reference
 * always fills it, :426; NLopt passes NULL to derivative-free algorithms).
 * Returns the cost; on error returns HUGE_VAL and records gtop_last_error.
 * Also keeps iter_num / total_time / the best-so-far cost curve
 * (:284, :436, :439-447) — read them with gtop_get_stats/gtop_get_cost_curve. */
double gtop_cost_nlopt(unsigned n, const double *x, double *grad, void *ctx);

/* ---- rendezvous: N serial callers share one launch ------------------- */
/* The reference runs one NLopt instance per problem and each calls the callback
 * serially (src/grad_traj_optimizer.cpp:137-195, :554-562).  One trajectory per
 * launch is launch-bound on a GPU, so this layer lets N host threads — each
 * running its own serial optimizer on its own trajectory — meet in ONE
 * gtop_eval_batch of the shared context:
 *   gtop_set_problem(ctx, N, m, ...)           the N problems, row i = caller i
 *   gtop_rendezvous_create(&r, ctx, N, m)
 *   thread i:  nlopt_set_min_objective(opt_i, gtop_cost_nlopt_shared,
 *                                      gtop_rendezvous_get_slot(r, i));
 *              nlopt_optimize(opt_i, ...);  gtop_rendezvous_leave(slot_i);
 * gtop_cost_nlopt_shared has exactly NLopt's nlopt_func shape.  It blocks until
 * every slot that has not left has arrived; the last arriver runs the batch;
 * each caller gets the cost and gradient of its own row — bit for bit what
 * gtop_eval_batch gives that row.  A caller MUST call gtop_rendezvous_leave
 * when its optimizer returns, or the others wait for it: for ever by default,
 * or — after gtop_rendezvous_set_timeout(r, seconds) — until one of them has
 * waited that long FOR A CALLER THAT HAS NOT ARRIVED, which breaks the
 * rendezvous for everybody (every call, pending or later, returns HUGE_VAL).
 * Time spent waiting for the elected caller's launch (everybody has arrived:
 * a module load on the first call, a large batch) never counts against the
 * timeout.  gtop_rendezvous_abort breaks it at once, from any thread (an
 * error path that cannot make every caller leave).  After a break no slot
 * counts as waiting: gtop_rendezvous_leave on it returns GTOP_OK.
 * gtop_rendezvous_destroy breaks the rendezvous, waits until no thread is
 * inside a call on it (a launch in flight uses its buffers) and frees it.
 * gtop_rendezvous_leave on a slot whose caller is inside the call right now
 * (i.e. from another thread) is refused with GTOP_ERR_STATE.  Returns
 * HUGE_VAL on misuse (wrong n, a slot that has left) or when the evaluation
 * failed (gtop_last_error(ctx)).  The context must not be used by anything
 * else while callers are inside. */
typedef struct gtop_rendezvous gtop_rendezvous;
typedef struct gtop_rendezvous_slot gtop_rendezvous_slot;
int gtop_rendezvous_create(gtop_rendezvous **out, gtop_ctx *ctx, int n_slots, int m);
int gtop_rendezvous_destroy(gtop_rendezvous *r);
gtop_rendezvous_slot *gtop_rendezvous_get_slot(gtop_rendezvous *r, int i);
double gtop_cost_nlopt_shared(unsigned n, const double *x, double *grad, void *slot);
int gtop_rendezvous_leave(gtop_rendezvous_slot *slot);
int gtop_rendezvous_set_timeout(gtop_rendezvous *r, double seconds);   /* 0 = wait for ever (default) */
int gtop_rendezvous_abort(gtop_rendezvous *r);
/* launches so far, seconds spent inside them, callbacks served (all slots) */
int gtop_rendezvous_stats(gtop_rendezvous *r, int64_t *launches, double *launch_seconds,
                          int64_t *callbacks);

/* Device-resident form: every pointer is a HIP device pointer of `dtype`
 * elements; launches on `hip_stream` (a hipStream_t, NULL = default stream)
 * and returns without synchronising.  Does not touch the problem set by
 * gtop_set_problem.  d_T: B*m (time_stride = m) or m (time_stride = 0). */
int gtop_eval_device(gtop_ctx *ctx, int dtype, int B, int m, const void *d_x,
                     const void *d_Df, const void *d_T, int time_stride,
                     void *d_cost, void *d_grad, void *hip_stream);

/* ---- one batch over several GPUs from one process -------- */
/* Nothing in the reference is multi-device (one NLopt instance per problem,
 * src/grad_traj_optimizer.cpp:137-195); the batched callback shards trivially:
 * trajectories are independent and the field is read-only.  A group owns one
 * gtop_ctx and one HIP stream per listed device (a device may be listed more
 * than once), keeps the distance field REPLICATED (gtop_group_init_sdf_map /
 * _update_sdf_map / _set_sdf build it on every device), and cuts the batch of
 * gtop_group_set_problem into contiguous slices of ceil(B / n) rows
 * (gtop_group_shard).  Every slice is launched on its own device before any is
 * waited for; there is no reduction, so results are bit-identical to the
 * unsharded evaluation.
 *   gtop_group_eval_batch        host buffers in and out (fp64), like
 *                                gtop_eval_batch
 *   gtop_group_upload_x + gtop_group_eval_resident(gather, synchronize)
 *                                everything resident; gather = 1 all-gathers
 *                                the costs, 2 also the gradients, so that EVERY
 *                                device holds the whole batch's results
 *                                (gtop_group_read_gathered copies one device's
 *                                set to the host; gtop_group_device_buffers
 *                                hands out the device pointers and the stream)
 *   gtop_group_optimize_batch_ex the batched optimizer, one launch per device
 * The all-gather is RCCL's (ncclAllGather in a group call, one communicator per
 * device, librccl.so loaded on first use) when all listed devices differ, peer
 * copies otherwise — gtop_group_gather_backend says which ("rccl" / "copy");
 * GTOP_GROUP_GATHER=copy|rccl in the environment at gtop_group_create forces
 * one.  One host thread at a time per group. */
typedef struct gtop_group gtop_group;
int gtop_group_create(gtop_group **out, const int *devices, int n_devices);
int gtop_group_destroy(gtop_group *g);
int gtop_group_size(const gtop_group *g);
gtop_ctx *gtop_group_context(gtop_group *g, int member);
/* Text of the group's last error; with g == NULL, the text of the calling
 * thread's last failed gtop_group_create (which member, and why). */
const char *gtop_group_last_error(const gtop_group *g);
const char *gtop_group_gather_backend(const gtop_group *g);
/* The backend and the reason for it in words, e.g. "copy: peer copies, because
 * librccl.so not found: ..." — a fallback from RCCL to copies is never silent. */
const char *gtop_group_gather_note(const gtop_group *g);
int gtop_group_set_params(gtop_group *g, const gtop_params *p);
int gtop_group_init_sdf_map(gtop_group *g, const double map_size[3], const double origin[3], double resolution);
int gtop_group_update_sdf_map(gtop_group *g, const double *pts, int npts);
int gtop_group_update_sdf_map_window(gtop_group *g, const double min_pos[3], const double max_pos[3], const double *pts,
                                     int npts);   /* gtop_update_sdf_map_window on every member */
int gtop_group_set_sdf(gtop_group *g, const double *dist_host, int nx, int ny, int nz, const double origin[3],
                       const double *map_size, double resolution);
int gtop_group_set_problem(gtop_group *g, int B, int m, const double *segment_time, int time_stride,
                           const double *Df);
int gtop_group_shard(const gtop_group *g, int member, int *first, int *count);
int gtop_group_eval_batch(gtop_group *g, int B, const double *x, double *cost, double *grad);
int gtop_group_upload_x(gtop_group *g, int B, const double *x);
int gtop_group_eval_resident(gtop_group *g, int gather, int synchronize);
int gtop_group_synchronize(gtop_group *g);
/* the gathered results as member `member` holds them: cost B, grad B*n host doubles (either may be NULL) */
int gtop_group_read_gathered(gtop_group *g, int member, double *cost, double *grad);
/* member's slice buffers (ceil(B/n) rows each), its gathered buffers (n*ceil(B/n) rows) and its hipStream_t */
int gtop_group_device_buffers(gtop_group *g, int member, void **d_x, void **d_cost, void **d_grad,
                              void **d_cost_all, void **d_grad_all, void **hip_stream);

/* ---- batched optimizer driver -------------------- */

/* Box bounds of GradTrajOptimizer::optimizeTrajectory
 * (src/grad_traj_optimizer.cpp:151-179): waypoint positions +-bos, velocities
 * +-vos, accelerations +-aos.  path: B x (m+1) x 3 waypoints; lb/ub: B x n.
 * Pure host helper (no context, no device). */
int gtop_default_bounds(int B, int m, const double *path, double bos,
                        double vos, double aos, double *lb, double *ub);

/* B independent bound-constrained CCSA-MMA solves on the device: what B calls
 * of nlopt::opt::optimize with algorithm 24 (LD_MMA) do one after another in
 * the reference (src/grad_traj_optimizer.cpp:137-195), here as max_evals
 * rounds of {cost/gradient, optimizer update} per trajectory — by default all
 * of them inside one kernel launch (gtop_set_optimizer_fusion).  The stop rule
 * is an evaluation count (the reference stops on wall-clock
 * maxtime, :144-148, which is not reproducible).  x: in = start point
 * (:182-187), out = best point found; min_cost: its cost.  fp64.
 * gtop_optimize_batch uses the problem of gtop_set_problem and host buffers;
 * gtop_optimize_device takes device pointers and only enqueues work. */
int gtop_optimize_batch(gtop_ctx *ctx, int B, double *x, const double *lb,
                        const double *ub, int max_evals, double *min_cost);
int gtop_optimize_device(gtop_ctx *ctx, int B, int m, void *d_x,
                         const void *d_Df, const void *d_T, int time_stride,
                         const void *d_lb, const void *d_ub, int max_evals,
                         void *d_min_cost, void *hip_stream);

/* The same with NLopt's other stop rules (nlopt::opt::set_ftol_rel / set_xtol_rel /
 * set_maxtime; the reference sets maxtime only, src/grad_traj_optimizer.cpp:144-148;
 * host twin csrc/mma.hpp:35-39, :127-137).  A trajectory stops by itself — inside
 * the one-launch loop — when, at the end of an outer MMA iteration,
 *   |f - f_prev| < ftol_rel (|f| + |f_prev|)/2, or
 *   |x_j - xprev_j| < xtol_rel (|x_j| + |xprev_j|)/2 for every j,
 * or, after at least one evaluation, when `maxtime` seconds of device wall
 * clock have passed since the launch (whole-loop-in-one-launch mode only; not
 * reproducible, as in the reference).  0 switches a rule off.  A stopped
 * trajectory costs no further evaluations.  nevals[b]: evaluations trajectory b
 * used; code[b]: 3 FTOL, 4 XTOL, 5 MAXEVAL, 6 MAXTIME (nlopt_result values).
 * nevals / code may be NULL (device pointers in the _device form). */
typedef struct {
  int32_t max_evals;       /* >= 1 */
  double ftol_rel, xtol_rel;
  double maxtime;          /* seconds */
} gtop_stop;
int gtop_optimize_batch_ex(gtop_ctx *ctx, int B, double *x, const double *lb,
                           const double *ub, const gtop_stop *stop, double *min_cost,
                           int32_t *nevals, int32_t *code);
int gtop_optimize_device_ex(gtop_ctx *ctx, int B, int m, void *d_x, const void *d_Df,
                            const void *d_T, int time_stride, const void *d_lb,
                            const void *d_ub, const gtop_stop *stop, void *d_min_cost,
                            int32_t *d_nevals, int32_t *d_code, void *hip_stream);
/* the same over a group's devices (B = the batch of gtop_group_set_problem): one launch per device */
int gtop_group_optimize_batch_ex(gtop_group *g, int B, double *x, const double *lb, const double *ub,
                                 const gtop_stop *stop, double *min_cost, int32_t *nevals, int32_t *code);

/* ---- post-processing ------------------------------ */

/* Batched GradTrajOptimizer::getCoefficient / getCoefficientFromDerivative
 * (src/grad_traj_optimizer.cpp:245-279): coeff is B x m x 18, row s =
 * [cx0..5 | cy0..5 | cz0..5], ascending powers. */
int gtop_coefficients_device(gtop_ctx *ctx, int B, int m, const void *d_x,
                             const void *d_Df, const void *d_T, int time_stride,
                             void *d_coeff, void *hip_stream);

/* The evaluation src/opti_node.cpp:135-142 runs on the optimised polynomials
 * through PolynomialTraj (include/grad_traj_optimization/polynomial_traj.hpp):
 * stats is B x GTOP_TRAJ_STATS doubles per trajectory =
 *   [getTimeSum, getLength (samples every dt_sample; the reference uses 0.01),
 *    getJerk, mean_v, max_v (getMeanAndMaxVel), mean_a, max_a
 *    (getMeanAndMaxAcc), getAccCost, number of getTraj samples].
 * The functions' quirks are kept (mean/max velocity and acceleration are
 * evaluated at each segment's END time, weighted by its sample count, as
 * :158-159 / :189-190 compute them). */
#define GTOP_TRAJ_STATS 9
int gtop_eval_trajectories_device(gtop_ctx *ctx, int B, int m, const void *d_coeff,
                                  const void *d_T, int time_stride,
                                  double dt_sample, void *d_stats,
                                  void *hip_stream);
/* The same, also returning the points PolynomialTraj::getTraj produces
 * (polynomial_traj.hpp:69-78: one every dt_sample, the sample time ACCUMULATED
 * as the reference does, while eval_t <= time_sum): samples is
 * B x max_samples x 3; trajectory b has stats[b][8] points, of which the first
 * min(stats[b][8], max_samples) are stored.  src/opti_node.cpp:108-120 publishes
 * exactly these. */
int gtop_sample_trajectories_device(gtop_ctx *ctx, int B, int m, const void *d_coeff,
                                    const void *d_T, int time_stride, double dt_sample,
                                    void *d_stats, void *d_samples, int max_samples,
                                    void *hip_stream);
/* Host-buffer form for the context's problem: coefficients (may be NULL) and
 * stats (may be NULL) of the first B trajectories at free variables x. */
int gtop_trajectory_stats(gtop_ctx *ctx, int B, const double *x, double dt_sample,
                          double *coeff, double *stats);
int gtop_trajectory_samples(gtop_ctx *ctx, int B, const double *x, double dt_sample,
                            double *coeff, double *stats, double *samples, int max_samples);

/* Distance queries against the static field plus moving obstacles:
 * EDTEnvironment::evaluateEDTWithGrad / distToBox / minDistToAllBox
 * (src/edt_environment.cpp:26-122).  Boxes are {p0, vel, scale} (nbox x 3 each,
 * host arrays, copied): centre p0 + vel*t (obj_predictor.h:57-66), extent
 * +-scale/2.  A query is (pos[3], time): trilinear value and gradient over
 * the 8 corner voxels with corner value min(static, nearest box at `time`);
 * time < 0 = static only, and then equal to getDistWithGradTrilinear
 * (src/sdf_map.cpp:185-242).  Outside the map: dist = -1, grad = 0.  fp64,
 * needs the fp64 field.  pos is N x 3, grad is N x 3. */
int gtop_set_moving_boxes(gtop_ctx *ctx, int nbox, const double *p0,
                          const double *vel, const double *scale);
int gtop_edt_query_device(gtop_ctx *ctx, int N, const void *d_pos,
                          const void *d_time, void *d_dist, void *d_grad,
                          void *hip_stream);
int gtop_edt_query(gtop_ctx *ctx, int N, const double *pos, const double *time,
                   double *dist, double *grad);
/* EDTEnvironment::evaluateCoarseEDT (src/edt_environment.cpp:124-136): no
 * interpolation and no gradient — the distance stored in the voxel that holds
 * pos (SDFMap::getDistance(pos), src/sdf_map.cpp:155-164; -1 outside the map),
 * and for time >= 0 its minimum with the distance from pos to the nearest box. */
int gtop_edt_coarse_query_device(gtop_ctx *ctx, int N, const void *d_pos,
                                 const void *d_time, void *d_dist, void *hip_stream);
int gtop_edt_coarse_query(gtop_ctx *ctx, int N, const double *pos, const double *time,
                          double *dist);

/* ---- bookkeeping the reference keeps inside the callback ------------ */

/* iter_num and total_time (src/grad_traj_optimizer.cpp:284, :436); reset as
 * optimizeTrajectory does after a step-2 solve (:236-239). */
int gtop_get_stats(const gtop_ctx *ctx, int64_t *iter_num, double *total_time);
int gtop_reset_stats(gtop_ctx *ctx);
/* Best-so-far cost curve (src/grad_traj_optimizer.cpp:439-447,
 * include/grad_traj_optimization/grad_traj_optimizer.h:127-130).  Copies up
 * to `cap` entries; *count receives the number available. */
int gtop_get_cost_curve(const gtop_ctx *ctx, double *cost, double *time,
                        int cap, int *count);
int gtop_clear_cost_curve(gtop_ctx *ctx);

/* ---- tuning knobs (not in the reference) ---------------------------- */

/* Launch geometry of the evaluation kernel (csrc/gtop_kernels.hip,
 * gtop_eval_plan).  A wavefront always holds whole segments; `waves` must be 0
 * or 1 (the number of wavefronts per workgroup follows from the rule below and
 * is not a knob).  samples_per_lane:
 *   3   ten lanes per polynomial segment.  Up to 6 segments: one trajectory per
 *       wavefront.  7 .. 12 segments: one trajectory over TWO wavefronts (a
 *       128-thread workgroup, wavefront w holds segments 6w .. 6w+5).  More
 *       than 12 segments (and the batched optimizer past 6): GTOP_ERR_INVALID
 *       at the evaluation.
 *   6   five lanes per segment: two trajectories of up to 6 segments per
 *       wavefront, one of up to 12, or 12 segments at a time beyond that.
 *   10  three lanes per segment, 21 / m whole trajectories per wavefront; up to
 *       10 segments, plain evaluations; more: GTOP_ERR_INVALID at the
 *       evaluation.  Five lanes per segment leave most of a wavefront idle for
 *       every length but 6, 11 and 12 segments: the rule takes three for
 *       2 .. 5 segments from 8 192 trajectories and for 7 .. 10 from 4 096.
 *   30  ONE lane per segment (a lane walks all 30 samples), 64 / m whole
 *       trajectories per wavefront; up to 64 segments, plain evaluations (the
 *       batched optimizer keeps its own rule); more segments: GTOP_ERR_INVALID
 *       at the evaluation.  Fewest instructions per trajectory, most distinct
 *       cache lines per load: the rule takes it for fp32 batches of 65 536
 *       six-segment trajectories and more, and past 12 segments wherever the
 *       12-segments-at-a-time body would end on a mostly idle chunk.
 *   0   choose from B, m and dtype (measured rule).
 * Other values are GTOP_ERR_INVALID at the call.  Results do not depend on the
 * geometry beyond fp summation order. */
int gtop_set_launch_geometry(gtop_ctx *ctx, int waves, int samples_per_lane);
/* The corner records every lookup reads exist in fp64 and fp32; the capturable
 * map-update entries (gtop_update_sdf_map_device, _window_device) rebuild both
 * so that a replayed graph leaves both current — a third of their record pass
 * (200^3: 98 us for both against 57 us for fp64 alone).  keep_fp32 = 0 tells the
 * context that it will never run fp32 evaluations: only fp64 records are kept,
 * and GTOP_F32 evaluations (and gtop_set_optimizer_precision(GTOP_F32) runs) fail
 * with GTOP_ERR_STATE.  1 (default) switches them back on; they are rebuilt from
 * the fp64 field at the next fp32 use. */
int gtop_set_field_precisions(gtop_ctx *ctx, int keep_fp32);
/* Batched optimizer: 2 (default) = one launch runs the whole loop (evaluate,
 * MMA update, evaluate, ... max_evals times) for every trajectory — they are
 * independent, so nothing has to return to the host or to HBM in between;
 * 1 = the MMA update runs as the epilogue of the evaluation kernel, one launch
 * per iteration; 0 = separate update launch.  Same arithmetic in all three. */
int gtop_set_optimizer_fusion(gtop_ctx *ctx, int fused);
/* Batched optimizer: the arithmetic of its EVALUATIONS.  GTOP_F64 (default) is
 * the reference's; GTOP_F32 runs them on the fp32 corner records in the
 * packed-fp32 bodies of gtop_eval_device(GTOP_F32) — the trial point, bounds,
 * Df, T, the CCSA-MMA update and every result stay fp64, so the interface of
 * gtop_optimize_* does not change.  Measured: 9 % less time for 1 024
 * trajectories of 6 segments, 17 % for 16 384, 20 % for 65 536.  The iterates
 * are those of an fp32 objective: an accept / reject decision within fp32's
 * noise (1e-5 relative) can fall the other way.  Needs a fused launch form
 * (fusion 1 or 2). */
int gtop_set_optimizer_precision(gtop_ctx *ctx, int dtype);

/* ---- collecting the shards' results (not in the reference) -- */

/* The all-gather of a rank's result rows as point-to-point stores: ONE kernel on
 * `hip_stream` copies `bytes` from d_src to each of the n_dsts (<= 16) device
 * pointers of d_dsts — this rank's slot in its own gathered buffer and in every
 * peer's, the peers' buffers mapped into this process (hipIpcOpenMemHandle
 * between the one-process-per-GPU ranks, peer access inside one process), each
 * destination over its own xGMI link.  Every rank's slot is written by that rank
 * alone, so no protocol is needed; an owner may read its buffer after any
 * synchronisation that orders the read behind the writers' kernels (the closing
 * barrier of a timed region).  All pointers 16-byte aligned.  Asynchronous,
 * capturable.  d_clock_minmax != NULL: the kernel also takes the device-clock
 * stamp of gtop_device_clock_stamp as it starts (behind the last evaluation, in
 * front of the stores) — a node of a graph saved.  bench.py's N > 1 runs gather their costs this way, with RCCL's
 * all-gather as the fallback when the peers' buffers cannot be mapped. */
int gtop_push_rows(gtop_ctx *ctx, const void *d_src, size_t bytes, void *const *d_dsts, int n_dsts,
                   void *d_clock_minmax, void *hip_stream);
/* Buffers for it that another process can map.  gtop_shared_alloc: a zeroed
 * device allocation of its own (an IPC handle names a whole allocation) on the
 * context's device and its 64-byte handle, to be sent to the peers by any means;
 * gtop_shared_open maps a PEER's buffer (a handle made in another process) for
 * the context's device — the device whose kernels will store into it; with
 * owner_device >= 0 (the owner's device ordinal as THIS process counts devices)
 * the call first checks hipDeviceCanAccessPeer and enables peer access, and
 * refuses (GTOP_ERR_STATE) where the owner's device cannot be reached, instead of
 * leaving a mapping a kernel would fault on; -1 = unknown: the lazy flag alone.
 * gtop_shared_close unmaps it; gtop_shared_free releases an
 * allocation of gtop_shared_alloc (after the peers have closed it). */
#define GTOP_IPC_HANDLE_BYTES 64
int gtop_shared_alloc(gtop_ctx *ctx, size_t bytes, void **d_ptr, unsigned char handle[GTOP_IPC_HANDLE_BYTES]);
int gtop_shared_open(gtop_ctx *ctx, const unsigned char handle[GTOP_IPC_HANDLE_BYTES], int owner_device,
                     void **d_ptr);
int gtop_shared_close(gtop_ctx *ctx, void *d_ptr);
int gtop_shared_free(gtop_ctx *ctx, void *d_ptr);

/* ---- measurement aid (not in the reference) --------------------------- */

/* Enqueues a one-lane kernel on `hip_stream` that reads the device's constant-
 * rate wall clock (the counter gtop_stop.maxtime is measured with) and folds it
 * into d_minmax[0] = min(d_minmax[0], t), d_minmax[1] = max(d_minmax[1], t)
 * (two uint64 in HBM; preset them to UINT64_MAX and 0).  Put one in front of and
 * one behind a run of launches — e.g. as first and last node of a captured
 * hipGraph — and (max - min) / gtop_device_clock_hz is the GPU's own time for
 * them: no host clock and no profiler instrumentation inside the interval.
 * bench.py times its region this way (`ms_per_step_gpu`). */
int gtop_device_clock_stamp(gtop_ctx *ctx, void *d_minmax, void *hip_stream);
/* Rate of that clock in Hz (hipDeviceAttributeWallClockRate; 100 MHz on MI355X). */
int gtop_device_clock_hz(gtop_ctx *ctx, double *hz);

#ifdef __cplusplus
}
#endif
#endif /* GTOP_H_ */
