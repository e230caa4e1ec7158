/*
 * gtop_oracle.h — interface of the CPU restatement (TEST INFRASTRUCTURE ONLY;
 * see the header of gtop_oracle.c: parity unpinned, never used by the product
 * path).  Citations are file:line into /root/reference/.
 */
#ifndef GTOP_ORACLE_H_
#define GTOP_ORACLE_H_

#ifdef __cplusplus
extern "C" {
#endif

/* ROS parameters read by the ctor (grad_traj_optimizer.cpp:5-32) that the
 * callback uses, plus `step` (:132) and the dyn-feasibility switch for the
 * commented-out block (:383-407). */
typedef struct {
  double ws, wc;          /* w_smooth, w_collision          :10-11 */
  double alpha, r, d0;    /* distance penalty               :13-15 */
  double alpha_v, r_v, v0; /* velocity penalty (dead code)  :17-19 */
  double alpha_a, r_a, a0; /* acceleration penalty (dead)   :21-23 */
  int step;               /* 0,1,2; 1 zeroes ws             :413-415 */
  int enable_dyn;         /* 0 = as shipped (block commented out) */
} oracle_params;

/* SDFMap fields the query reads (sdf_map.h:13-23) */
typedef struct {
  double origin[3], min_range[3], max_range[3];
  double resolution, resolution_inv;
  int grid[3];
  double *dist; /* x*ny*nz + y*nz + z, sdf_map.cpp:172-173 */
} oracle_sdf;

void oracle_segment_time(int npts, const double *path, double mean_v,
                         double init_time, double *T);
int oracle_generator(int m, const double *T, double *A, double *Q, double *Ct,
                     double *L, double *R);
void oracle_initial_d(int npts, const double *path, const double *vel,
                      const double *acc, double *Df, double *Dp);

void oracle_sdf_init(oracle_sdf *S, const double origin[3], double resolution,
                     const int grid[3], double *distance_buffer);
void oracle_sdf_init_size(oracle_sdf *S, const double origin[3],
                          double resolution, const double map_size[3],
                          int grid_out[3]);
double oracle_sdf_query(const oracle_sdf *S, const double pos[3],
                        double grad[3]);

/* EDTEnvironment::evaluateEDTWithGrad (src/edt_environment.cpp:75-122): trilinear
 * over min(static, moving boxes at `time`); time < 0: static only.  Boxes:
 * p0/vel/scale, nbox x 3 each.  See the .c file for the conventions taken. */
double oracle_edt_query(const oracle_sdf *S, int nbox, const double *box_p0, const double *box_vel,
                        const double *box_scale, const double pos[3], double time, double grad[3]);
int oracle_set_occupancy(const oracle_sdf *S, double *occupancy,
                         const double pos[3], int occ);
void oracle_esdf_build(const oracle_sdf *S, const double *occupancy,
                       double *distance);
/* the windowed update (sdf_map.cpp:28-53 resetBuffer(min, max), :244-264 setUpdateRange, :310-368 updateESDF3d over
 * min_vec .. max_vec) */
void oracle_window_ids(const oracle_sdf *S, const double min_pos[3], const double max_pos[3], int min_id[3],
                       int max_id[3]);
void oracle_reset_window(const oracle_sdf *S, double *occupancy, double *distance, const double min_pos[3],
                         const double max_pos[3]);
void oracle_esdf_build_window(const oracle_sdf *S, const double *occupancy, double *distance, const int lo[3],
                              const int hi[3]);

double oracle_cost_grad(int m, const double *L, const double *R,
                        const double *Df, const double *T,
                        const oracle_params *prm, const oracle_sdf *S,
                        const double *x, double *grad_out);
double oracle_eval_batch(int B, int m, const double *T, int t_stride,
                         const double *Df, const oracle_params *prm,
                         const oracle_sdf *S, const double *x, double *cost,
                         double *grad, int reps, int nthreads);

void oracle_traj_stats(int m, const double *coeff, const double *T, double dt_sample, double *out);
/* PolynomialTraj::getTraj (polynomial_traj.hpp:69-78): returns the number of points, stores the first max_samples */
int oracle_traj_samples(int m, const double *coeff, const double *T, double dt_sample, int max_samples,
                        double *samples);
/* EDTEnvironment::evaluateCoarseEDT (src/edt_environment.cpp:124-136) */
double oracle_edt_coarse(const oracle_sdf *S, int nbox, const double *box_p0, const double *box_vel,
                         const double *box_scale, const double pos[3], double time);
void oracle_coefficients(int m, const double *L, const double *Df, const double *x, double *coe);

#ifdef __cplusplus
}
#endif
#endif
