// gtop_kernels.h — launch interface between the C-ABI layer (gtop_capi.cpp)
// and the gfx950 kernels (gtop_kernels.hip, gtop_esdf.hip).  Internal; the
// public boundary is include/gtop.h.
#ifndef GTOP_KERNELS_H_
#define GTOP_KERNELS_H_

#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

// Kernel arguments of gtop_eval_wave_kernel<R>, all in the arithmetic type R of the
// launch (double for GTOP_F64, float for GTOP_F32).
template <typename R>
struct GtopKernelArgs {
  // per-trajectory inputs/outputs (HBM)
  const R *x;    // [B][n]   free variables, axis-major
  const R *Df;   // [B][3][6]
  const R *T;    // [B][m] or [m]
  R *cost;       // [B]
  R *grad;       // [B][n]
  int B, m, t_stride;
  // shared distance field — SDFMap fields, sdf_map.h:13-23.  `sdf` is the resident CORNER-RECORD copy
  // (gtop_records.hip: record (ix+1, iy+1, level+1) = the four clamped (x,y) corners of base index (ix,iy) at one z
  // level, records z-fastest), not the boundary's z-fastest buffer; nx, ny, nz are the voxel grid's.
  const R *sdf;
  int nx, ny, nz;
  R origin[3];
  R lo[3], hi[3];   // min_range + 1e-4, max_range - 1e-4  (isInMap, sdf_map.cpp:55-69)
  float lo_f[3], hi_f[3];   // lo rounded up / hi rounded down to float: for a float p, p < lo <=> p < lo_f, p > hi <=> p > hi_f
  R res, res_inv;
  // posToIndex's constants in double whatever R is: the fp32 kernels decide which cell a sample reads as the reference
  // does, in double on the widened float position (gtop_kernels.hip IndexBox); the fp64 kernels do not read these
  double idx_origin[3], idx_half, idx_rinv;
  // parameters — grad_traj_optimizer.cpp:5-32
  R ws, wc, alpha, d0, alpha_v, r_v, v0, alpha_a, r_a, a0;
  R inv_r, alpha_over_r;   // 1/r, alpha/r  (:509, :514)
  R inv_r_v, inv_r_a, gv_scale, ga_scale;   // 1/r_v, 1/r_a, alpha_v/r_v, alpha_a/r_a  (:517-535, the DYN bodies)
  int step;
};

// ---- batched CCSA-MMA optimizer state (gtop_mma.hip), all fp64, [B][n] / [B] ----
struct GtopMmaState {
  double *x, *xcur, *xprev, *xprevprev, *dfdx, *sigma;   // [B][n]
  const double *lb, *ub;                                  // [B][n]
  double *rho, *minf, *gval, *wval, *fprev;               // [B]; fprev: f at the start of the outer iteration
  int *k, *state, *nevals;                                // [B]; state 0 first evaluation pending, 1 running,
                                                          //      >= 3 stopped with that nlopt_result code
  int iters;                                              // evaluations per launch of the fused kernel
  int max_evals;                                          // the optimisation's evaluation cap (all launches together)
  // stop rules beside the evaluation count (mma.hpp:35-39; nlopt set_ftol_rel / set_xtol_rel / set_maxtime,
  // grad_traj_optimizer.cpp:144-148); 0 = off.  max_ticks: wall clock in ticks of the 100 MHz device clock.
  double ftol_rel, xtol_rel;
  long long max_ticks;
  // The one-launch loop of gtop_eval_wave_kernel can do without the launches around it (all optional, NULL = off):
  // x0_init: start points [B][n] — the state is initialised in the kernel (mma_init_kernel's arithmetic) instead of
  // being loaded; out_*: where the results go when the loop ends (the best point, its cost, the nlopt_result-style
  // code and the evaluations used: what the copies and mma_finish_kernel deliver otherwise).
  const double *x0_init;
  double *out_x, *out_minf;
  int *out_code, *out_nevals;
};
enum { GTOP_MMA_FTOL_REACHED = 3, GTOP_MMA_XTOL_REACHED = 4, GTOP_MMA_MAXEVAL_REACHED = 5, GTOP_MMA_MAXTIME_REACHED = 6 };
// How one launch is laid out on the wavefronts: spl = samples per lane (3: ten lanes per segment, one trajectory of up
// to 6 segments per wavefront; 6: five lanes per segment, up to 12 segments), nt = trajectories per wavefront (2 only
// at spl 6 with up to 6 segments), is_long = more than 12 segments (the wavefront walks them 12 at a time).
struct GtopEvalPlan {
  int spl, nt;
  bool is_long;
  int nw;   // wavefronts per trajectory: 2 for 7 .. 12 segments at ten lanes per segment (small batches), else 1
};
// The launch rule.  pinned_spl: 0 = auto, 3 or 6; for_optimizer: the optimizer loop and the evaluations of its
// multi-launch forms (the same rule with the loop's own switch point to two trajectories per wavefront).  false: the request cannot be served (m < 2, spl 3 with more than 6
// segments, more segments than one wavefront's LDS holds — 227, in the optimizer loop 118).
bool gtop_eval_plan(int B, int m, size_t elem, int pinned_spl, bool for_optimizer, GtopEvalPlan *plan);
// dyn: enable_dyn (the kernel applies it at step 2 only, as the commented-out block would)
template <typename R>
hipError_t gtop_launch_eval(const GtopKernelArgs<R> &args, const GtopEvalPlan &plan, bool dyn, hipStream_t stream);

// `bytes` from src to each of the first n_dsts pointers of `dsts` (device memory of this or of a peer GPU mapped into
// this process; 16-byte aligned), one kernel: gtop_push.hip
#define GTOP_PUSH_MAX_DSTS 16
struct GtopPushDsts { void *p[GTOP_PUSH_MAX_DSTS]; };
hipError_t gtop_launch_push_rows(const void *src, size_t bytes, const GtopPushDsts &dsts, int n_dsts,
                                 unsigned long long *minmax /* optional clock stamp, NULL = none */, hipStream_t stream);

// minmax[0] = min(minmax[0], clock), minmax[1] = max(minmax[1], clock) of the device's constant-rate wall clock
hipError_t gtop_launch_clock_stamp(unsigned long long *minmax, hipStream_t stream);

// ---- ESDF construction (gtop_esdf.hip) -----------------------------------
struct GtopGrid {
  int nx, ny, nz;
  double origin[3], min_range[3], max_range[3];
  double res, res_inv;
};

// occupancy := 0, distance := 10000   (sdf_map.cpp:26-53)
hipError_t gtop_launch_esdf_reset(uint8_t *occ, double *dist, size_t nvox, hipStream_t stream);
// setOccupancy per point (sdf_map.cpp:80-99)
hipError_t gtop_launch_esdf_mark(const GtopGrid &g, const double *pts, int npts, uint8_t *occ,
                                 hipStream_t stream);
// updateESDF3d (sdf_map.cpp:310-368): the three sweeps z, y, x as exact integer minimisations,
// then res*sqrt(.) into dist (fp64; dist32, an fp32 copy, may be NULL and is since round 4: the fp32 path reads fp32
// corner records).  tmp1/tmp2: nvox int32 each.
bool gtop_esdf_supported(const GtopGrid &g);
size_t gtop_esdf_rows_ints(const GtopGrid &g);   // ints of row workspace the builder needs
hipError_t gtop_launch_esdf_build(const GtopGrid &g, const uint8_t *occ, int *tmp1, int *tmp2, int *rows,
                                  double *dist, float *dist32, hipStream_t stream);

// the local update of compare2.cpp:147-152 (gtop_esdf_window.hip): resetBuffer(min, max) over the inclusive voxel box
// lo .. hi, and updateESDF3d over the same box (sdf_map.cpp:28-53, :310-368)
hipError_t gtop_launch_esdf_window_reset(const GtopGrid &g, const int lo[3], const int hi[3], uint8_t *occ, double *dist,
                                         hipStream_t stream);
hipError_t gtop_launch_esdf_window_build(const GtopGrid &g, const int lo[3], const int hi[3], const uint8_t *occ, int *tmp1,
                                         int *tmp2, double *dist, hipStream_t stream);
// a compact grid's distances (z fastest) into the window
hipError_t gtop_launch_esdf_window_scatter(const GtopGrid &g, const int lo[3], const int hi[3], const double *sub,
                                           double *dist, hipStream_t stream);
// the compact path's reset + marking without a gather: clears the window's occupancy in the map and in `sub`, marks the
// points in the map and (inside the window) in `sub`; the distances are not reset (the scatter rewrites the window)
hipError_t gtop_launch_esdf_window_reset_mark_compact(const GtopGrid &g, const int lo[3], const int hi[3], const double *pts,
                                                      int npts, uint8_t *occ, uint8_t *sub, hipStream_t stream);

// ---- corner records (gtop_records.hip): the gather-friendly resident copy the lookups read ----
size_t gtop_record_count(const GtopGrid &g);   // (nx+1)(ny+1)(nz+2) records of 4 values
// S -> D in {double -> double, double -> float, float -> float}; rec32 (may be NULL): the fp32 records too, in the
// same pass; vlo / vhi: inclusive voxel box whose records are rebuilt (NULL = the whole field)
template <typename S, typename D>
hipError_t gtop_launch_build_records(const GtopGrid &g, const S *field, D *rec, float *rec32, const int *vlo,
                                     const int *vhi, hipStream_t stream);

// the optimizer loop: st.iters x {cost/gradient at st.xcur, CCSA-MMA update} per trajectory in one launch (fp64; a plan
// made with for_optimizer = true); honours st.x0_init / st.out_*
hipError_t gtop_launch_eval_mma(const GtopKernelArgs<double> &args, const GtopMmaState &st, const GtopEvalPlan &plan,
                                bool dyn, hipStream_t stream);
hipError_t gtop_launch_eval_mma(const GtopKernelArgs<float> &args, const GtopMmaState &st, const GtopEvalPlan &plan,
                                bool dyn, hipStream_t stream);
hipError_t gtop_launch_mma_init(const GtopMmaState &st, int B, int n, const double *x0, hipStream_t stream);
hipError_t gtop_launch_mma_update(const GtopMmaState &st, int B, int n, const double *fcur, const double *gcur,
                                  hipStream_t stream);
// code[b] = nlopt_result-style stop code, nevals[b] = evaluations used (either may be NULL)
hipError_t gtop_launch_mma_finish(const GtopMmaState &st, int B, int *code, int *nevals, hipStream_t stream);

// ---- setup + post-processing (gtop_setup.hip), fp64 ----
#define GTOP_TRAJ_STATS 9   // time_sum, length, jerk, mean_v, max_v, mean_a, max_a, acc_cost, n_samples
hipError_t gtop_launch_setup_paths(int B, int m, const double *wp, double mean_v, double init_time, double *T,
                                   double *Df, double *x0, hipStream_t stream);
hipError_t gtop_launch_coefficients(int B, int m, const double *x, const double *Df, const double *T, int t_stride,
                                    double *coeff, hipStream_t stream);
// samples (may be NULL): [B][max_samples][3] getTraj points
hipError_t gtop_launch_eval_trajectories(int B, int m, const double *coeff, const double *T, int t_stride,
                                         double dt_sample, double *out, double *samples, int max_samples,
                                         hipStream_t stream);

// ---- static field + moving boxes (gtop_edt.hip) -----------------------------
// field: the z-fastest fp64 buffer (the coarse query's voxel values); rec: its corner records (the interpolating query)
hipError_t gtop_launch_edt_query(const GtopGrid &g, const double *field, const double *rec, int nbox, const double *box_p0,
                                 const double *box_vel, const double *box_scale, int N, const double *pos,
                                 const double *time, double *dist, double *grad, hipStream_t stream);

#endif  // GTOP_KERNELS_H_
