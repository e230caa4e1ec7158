/*
 * gtop_oracle.c — CPU restatement of GTOP's cost/gradient callback.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is product code: only
 * tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load
 * this library, and only as the checker / the reported CPU baseline.  The
 * product path (grad_traj_optimization_amd/) never links or calls it.
 *
 * PARITY UNPINNED: the reference ships no tests, golden vectors or fixtures
 * for this path and cannot be compiled in this image (it needs Eigen3, ROS
 * and NLopt headers, none present).  This file follows the cited reference
 * lines statement by statement (dense L/R from an LU inverse, pow(), the
 * float round-trip of pos/vel, the extra `cd` factor, +1e-3/+1e-5 offsets);
 * it is cross-checked against an independently written numpy twin
 * (oracle/np_twin.py) and against structural known answers derivable from
 * the reference source (tests/test_oracle.py).
 *
 * All citations are file:line into /root/reference/.
 * Matrices are dense row-major doubles.
 */
#define _POSIX_C_SOURCE 200809L
#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#include "gtop_oracle.h"

/* ------------------------------------------------------------------ */
/* dense helpers                                                      */
/* ------------------------------------------------------------------ */

static double *dalloc(size_t n) {
  double *p = (double *)calloc(n ? n : 1, sizeof(double));
  return p;
}

/* C(r×c) = A(r×k) · B(k×c) */
static void matmul(int r, int k, int c, const double *A, const double *B,
                   double *C) {
  for (int i = 0; i < r; ++i)
    for (int j = 0; j < c; ++j) {
      double s = 0.0;
      for (int p = 0; p < k; ++p) s += A[i * k + p] * B[p * c + j];
      C[i * c + j] = s;
    }
}

static void transpose(int r, int c, const double *A, double *At) {
  for (int i = 0; i < r; ++i)
    for (int j = 0; j < c; ++j) At[j * r + i] = A[i * c + j];
}

/* inverse by LU with partial pivoting — what Eigen's MatrixXd::inverse()
 * (PartialPivLU) does for dynamic sizes (qp_generator.cpp:390-393). */
static int lu_inverse(int n, const double *Ain, double *inv) {
  double *a = dalloc((size_t)n * n);
  int *piv = (int *)malloc(sizeof(int) * (size_t)n);
  memcpy(a, Ain, sizeof(double) * (size_t)n * n);
  for (int i = 0; i < n; ++i) piv[i] = i;
  for (int k = 0; k < n; ++k) {
    int p = k;
    double best = fabs(a[k * n + k]);
    for (int i = k + 1; i < n; ++i)
      if (fabs(a[i * n + k]) > best) {
        best = fabs(a[i * n + k]);
        p = i;
      }
    if (best == 0.0) {
      free(a);
      free(piv);
      return -1;
    }
    if (p != k) {
      for (int j = 0; j < n; ++j) {
        double t = a[k * n + j];
        a[k * n + j] = a[p * n + j];
        a[p * n + j] = t;
      }
      int t = piv[k];
      piv[k] = piv[p];
      piv[p] = t;
    }
    for (int i = k + 1; i < n; ++i) {
      a[i * n + k] /= a[k * n + k];
      double l = a[i * n + k];
      if (l != 0.0)
        for (int j = k + 1; j < n; ++j) a[i * n + j] -= l * a[k * n + j];
    }
  }
  /* solve A X = I column by column: L U x = P e_c */
  double *y = dalloc((size_t)n);
  for (int c = 0; c < n; ++c) {
    for (int i = 0; i < n; ++i) {
      double s = (piv[i] == c) ? 1.0 : 0.0;
      for (int j = 0; j < i; ++j) s -= a[i * n + j] * y[j];
      y[i] = s;
    }
    for (int i = n - 1; i >= 0; --i) {
      double s = y[i];
      for (int j = i + 1; j < n; ++j) s -= a[i * n + j] * inv[j * n + c];
      inv[i * n + c] = s / a[i * n + i];
    }
  }
  free(y);
  free(a);
  free(piv);
  return 0;
}

static int factorial(int x) { /* qp_generator.cpp:173-179 */
  int fac = 1;
  for (int i = x; i > 0; i--) fac = fac * i;
  return fac;
}

/* ------------------------------------------------------------------ */
/* setup side: segment times, generator matrices, initial Df/Dp       */
/* ------------------------------------------------------------------ */

/* grad_traj_optimizer.cpp:73-81.  Note `i == segment_time.size()` is never
 * true inside the loop, so only the FIRST segment receives init_time. */
void oracle_segment_time(int npts, const double *path, double mean_v,
                         double init_time, double *T) {
  int m = npts - 1;
  for (int i = 0; i < m; ++i) {
    double dx = path[3 * i + 0] - path[3 * (i + 1) + 0];
    double dy = path[3 * i + 1] - path[3 * (i + 1) + 1];
    double dz = path[3 * i + 2] - path[3 * (i + 1) + 2];
    double len = sqrt(dx * dx + dy * dy + dz * dz);
    if (i == 0 || i == m) {
      T[i] = len / mean_v + init_time;
    } else {
      T[i] = len / mean_v;
    }
  }
}

/* Mapping matrix A (qp_generator.cpp:181-197), jerk Hessian Q (:223-236),
 * selection Ct (:357-387), L = A^-1 Ct (:390),
 * R = C A^-T Q A^-1 Ct (:392-393).  Sizes: A,Q 6m×6m; Ct,L 6m×(3m+3);
 * R (3m+3)². Any output pointer may be NULL.  Returns 0, or -1 if m < 2
 * (StackOptiDep writes out of bounds for m == 1) or A is singular. */
int oracle_generator(int m, const double *T, double *A_out, double *Q_out,
                     double *Ct_out, double *L_out, double *R_out) {
  if (m < 2) return -1;
  int n6 = 6 * m, nd = 3 * m + 3;
  double *A = dalloc((size_t)n6 * n6), *Q = dalloc((size_t)n6 * n6);
  double *Ct = dalloc((size_t)n6 * nd);
  for (int k = 0; k < m; k++) {
    for (int i = 0; i < 3; i++) {
      A[(k * 6 + 2 * i) * n6 + k * 6 + i] = factorial(i);
      for (int j = i; j < 6; j++)
        A[(k * 6 + 2 * i + 1) * n6 + k * 6 + j] =
            factorial(j) / factorial(j - i) * pow(T[k], j - i);
    }
  }
  for (int k = 0; k < m; k++)
    for (int i = 3; i < 6; i++)
      for (int j = 3; j < 6; j++)
        Q[(k * 6 + i) * n6 + k * 6 + j] =
            i * (i - 1) * (i - 2) * j * (j - 1) * (j - 2) / (i + j - 5) *
            pow(T[k], (i + j - 5));

#define CT(r, c) Ct[(r) * nd + (c)]
  CT(0, 0) = 1;
  CT(2, 1) = 1;
  CT(4, 2) = 1;
  CT(1, 6) = 1;
  CT(3, 7) = 1;
  CT(5, 8) = 1;
  CT(6 * (m - 1) + 0, 3 * m + 0) = 1;
  CT(6 * (m - 1) + 2, 3 * m + 1) = 1;
  CT(6 * (m - 1) + 4, 3 * m + 2) = 1;
  CT(6 * (m - 1) + 1, 3) = 1;
  CT(6 * (m - 1) + 3, 4) = 1;
  CT(6 * (m - 1) + 5, 5) = 1;
  for (int j = 2; j < m; j++) {
    CT(6 * (j - 1) + 0, 6 + 3 * (j - 2) + 0) = 1;
    CT(6 * (j - 1) + 1, 6 + 3 * (j - 1) + 0) = 1;
    CT(6 * (j - 1) + 2, 6 + 3 * (j - 2) + 1) = 1;
    CT(6 * (j - 1) + 3, 6 + 3 * (j - 1) + 1) = 1;
    CT(6 * (j - 1) + 4, 6 + 3 * (j - 2) + 2) = 1;
    CT(6 * (j - 1) + 5, 6 + 3 * (j - 1) + 2) = 1;
  }
#undef CT

  double *Ainv = dalloc((size_t)n6 * n6);
  if (lu_inverse(n6, A, Ainv) != 0) {
    free(A);
    free(Q);
    free(Ct);
    free(Ainv);
    return -1;
  }
  double *L = dalloc((size_t)n6 * nd);
  matmul(n6, n6, nd, Ainv, Ct, L);

  /* R = ((( C · B^T ) · Q ) · A^-1 ) · Ct, left to right */
  double *C = dalloc((size_t)nd * n6), *Bt = dalloc((size_t)n6 * n6);
  transpose(n6, nd, Ct, C);
  transpose(n6, n6, Ainv, Bt);
  double *t1 = dalloc((size_t)nd * n6), *t2 = dalloc((size_t)nd * n6);
  matmul(nd, n6, n6, C, Bt, t1);
  matmul(nd, n6, n6, t1, Q, t2);
  matmul(nd, n6, n6, t2, Ainv, t1);
  double *R = dalloc((size_t)nd * nd);
  matmul(nd, n6, nd, t1, Ct, R);

  if (A_out) memcpy(A_out, A, sizeof(double) * (size_t)n6 * n6);
  if (Q_out) memcpy(Q_out, Q, sizeof(double) * (size_t)n6 * n6);
  if (Ct_out) memcpy(Ct_out, Ct, sizeof(double) * (size_t)n6 * nd);
  if (L_out) memcpy(L_out, L, sizeof(double) * (size_t)n6 * nd);
  if (R_out) memcpy(R_out, R, sizeof(double) * (size_t)nd * nd);
  free(A);
  free(Q);
  free(Ct);
  free(Ainv);
  free(L);
  free(C);
  free(Bt);
  free(t1);
  free(t2);
  free(R);
  return 0;
}

/* Straight-line initial derivatives (type == 2 branch of PolyQPGeneration,
 * qp_generator.cpp:199-221, then getInitialD :407-451).
 * Df 3×6 = [p_start, startVel, startAcc, p_end, 0, 0] per axis;
 * Dp 3×(3m-3): interior waypoint positions, vel = acc = 0. */
void oracle_initial_d(int npts, const double *path, const double *vel,
                      const double *acc, double *Df, double *Dp) {
  int m = npts - 1, ndp = 3 * m - 3, n6 = 6 * m;
  double *D[3];
  for (int a = 0; a < 3; ++a) D[a] = dalloc((size_t)n6);
  for (int k = 1; k < m + 1; k++) {
    for (int a = 0; a < 3; ++a) {
      D[a][(k - 1) * 6] = path[3 * (k - 1) + a];
      D[a][(k - 1) * 6 + 1] = path[3 * k + a];
      if (k == 1) {
        D[a][(k - 1) * 6 + 2] = vel[a];
        D[a][(k - 1) * 6 + 4] = acc[a];
      }
    }
  }
  for (int a = 0; a < 3; ++a) {
    for (int j = 0; j < 6; ++j) Df[a * 6 + j] = 0.0;
    Df[a * 6 + 0] = D[a][0];
    Df[a * 6 + 3] = D[a][n6 - 5];
    Df[a * 6 + 1] = vel[a];
    Df[a * 6 + 2] = acc[a];
    for (int k = 1; k < m; k++)
      for (int i = 0; i < 3; i++)
        Dp[a * ndp + (k - 1) * 3 + i] = D[a][(k - 1) * 6 + 2 * i + 1];
  }
  for (int a = 0; a < 3; ++a) free(D[a]);
}

/* ------------------------------------------------------------------ */
/* distance field                                                     */
/* ------------------------------------------------------------------ */

/* sdf_map.cpp:3-24 (grid passed in directly) */
void oracle_sdf_init(oracle_sdf *S, const double origin[3], double resolution,
                     const int grid[3], double *distance_buffer) {
  for (int i = 0; i < 3; ++i) {
    S->origin[i] = origin[i];
    S->grid[i] = grid[i];
    S->min_range[i] = origin[i];
    S->max_range[i] = origin[i] + grid[i] * resolution;
  }
  S->resolution = resolution;
  S->resolution_inv = 1 / resolution;
  S->dist = distance_buffer;
}

/* as sdf_map.cpp:3-24 computes it: map_size given, grid = ceil(size/res),
 * max_range = origin + map_size */
void oracle_sdf_init_size(oracle_sdf *S, const double origin[3],
                          double resolution, const double map_size[3],
                          int grid_out[3]) {
  for (int i = 0; i < 3; ++i) {
    S->origin[i] = origin[i];
    S->grid[i] = (int)ceil(map_size[i] / resolution);
    grid_out[i] = S->grid[i];
    S->min_range[i] = origin[i];
    S->max_range[i] = origin[i] + map_size[i];
  }
  S->resolution = resolution;
  S->resolution_inv = 1 / resolution;
  S->dist = NULL;
}

static int sdf_in_map(const oracle_sdf *S, const double pos[3]) {
  /* sdf_map.cpp:55-69 */
  if (pos[0] < S->min_range[0] + 1e-4 || pos[1] < S->min_range[1] + 1e-4 ||
      pos[2] < S->min_range[2] + 1e-4)
    return 0;
  if (pos[0] > S->max_range[0] - 1e-4 || pos[1] > S->max_range[1] - 1e-4 ||
      pos[2] > S->max_range[2] - 1e-4)
    return 0;
  return 1;
}

static void sdf_pos_to_index(const oracle_sdf *S, const double pos[3],
                             int id[3]) {
  /* sdf_map.cpp:71-74 */
  for (int i = 0; i < 3; ++i)
    id[i] = (int)floor((pos[i] - S->origin[i]) * S->resolution_inv);
}

static double sdf_get_distance(const oracle_sdf *S, int x, int y, int z) {
  /* sdf_map.cpp:166-174: per-axis index clamp */
  int gx = S->grid[0], gy = S->grid[1], gz = S->grid[2];
  x = x < gx - 1 ? x : gx - 1;
  x = x > 0 ? x : 0;
  y = y < gy - 1 ? y : gy - 1;
  y = y > 0 ? y : 0;
  z = z < gz - 1 ? z : gz - 1;
  z = z > 0 ? z : 0;
  return S->dist[(size_t)x * gy * gz + (size_t)y * gz + z];
}

/* sdf_map.cpp:185-242.  Out of map: returns -1 and (convention, SURVEY
 * A.4 Q4: the reference leaves grad uninitialised) grad = 0. */
double oracle_sdf_query(const oracle_sdf *S, const double pos[3],
                        double grad[3]) {
  if (!sdf_in_map(S, pos)) {
    grad[0] = grad[1] = grad[2] = 0.0;
    return -1;
  }
  double res = S->resolution, rinv = S->resolution_inv;
  double pos_m[3], idx_pos[3], diff[3];
  int idx[3];
  for (int i = 0; i < 3; ++i) pos_m[i] = pos[i] - 0.5 * res * 1.0;
  sdf_pos_to_index(S, pos_m, idx);
  for (int i = 0; i < 3; ++i) /* sdf_map.cpp:76-78 */
    idx_pos[i] = (idx[i] + 0.5) * res + S->origin[i];
  for (int i = 0; i < 3; ++i) diff[i] = (pos[i] - idx_pos[i]) * rinv;

  double values[2][2][2];
  for (int x = 0; x < 2; x++)
    for (int y = 0; y < 2; y++)
      for (int z = 0; z < 2; z++)
        values[x][y][z] = sdf_get_distance(S, idx[0] + x, idx[1] + y, idx[2] + z);

  double v00 = (1 - diff[0]) * values[0][0][0] + diff[0] * values[1][0][0];
  double v01 = (1 - diff[0]) * values[0][0][1] + diff[0] * values[1][0][1];
  double v10 = (1 - diff[0]) * values[0][1][0] + diff[0] * values[1][1][0];
  double v11 = (1 - diff[0]) * values[0][1][1] + diff[0] * values[1][1][1];
  double v0 = (1 - diff[1]) * v00 + diff[1] * v10;
  double v1 = (1 - diff[1]) * v01 + diff[1] * v11;
  double dist = (1 - diff[2]) * v0 + diff[2] * v1;

  grad[2] = (v1 - v0) * rinv;
  grad[1] = ((1 - diff[2]) * (v10 - v00) + diff[2] * (v11 - v01)) * rinv;
  grad[0] = (1 - diff[2]) * (1 - diff[1]) * (values[1][0][0] - values[0][0][0]);
  grad[0] += (1 - diff[2]) * diff[1] * (values[1][1][0] - values[0][1][0]);
  grad[0] += diff[2] * (1 - diff[1]) * (values[1][0][1] - values[0][0][1]);
  grad[0] += diff[2] * diff[1] * (values[1][1][1] - values[0][1][1]);
  grad[0] *= rinv;
  return dist;
}

/* EDTEnvironment::distToBox / minDistToAllBox, src/edt_environment.cpp:26-73.
 * A box is {p0, vel, scale}: its centre at `time` is the constant-velocity
 * prediction p0 + vel*time (ObjPrediction::evaluateConstVel,
 * include/grad_traj_optimization/obj_predictor.h:57-66). */
static double edt_min_dist_to_boxes(int nbox, const double *p0, const double *vel, const double *scale,
                                    const double pt[3], double time) {
  double dist = 10000000.0; /* :64 */
  for (int b = 0; b < nbox; ++b) {
    double d2 = 0.0;
    for (int i = 0; i < 3; ++i) {
      double c = p0[3 * b + i] + vel[3 * b + i] * time;
      double bmax = c + 0.5 * scale[3 * b + i], bmin = c - 0.5 * scale[3 * b + i]; /* :31-32 */
      double di = (pt[i] >= bmin && pt[i] <= bmax) ? 0.0 : fmin(fabs(pt[i] - bmin), fabs(pt[i] - bmax)); /* :36-40 */
      d2 += di * di;
    }
    double di = sqrt(d2); /* dist.norm(), :42 */
    if (di < dist) dist = di;
  }
  return dist;
}

/* EDTEnvironment::evaluateCoarseEDT, src/edt_environment.cpp:124-136: the distance of the voxel that holds
 * `pos` (SDFMap::getDistance(Vector3d), src/sdf_map.cpp:155-164: -1 outside the map), and with time >= 0 its
 * minimum with the distance from `pos` itself to the nearest moving box.  No interpolation, no gradient. */
double oracle_edt_coarse(const oracle_sdf *S, int nbox, const double *box_p0, const double *box_vel,
                         const double *box_scale, const double pos[3], double time) {
  double d1 = -1.0;
  if (sdf_in_map(S, pos)) {
    int id[3];
    sdf_pos_to_index(S, pos, id);
    d1 = sdf_get_distance(S, id[0], id[1], id[2]);
  }
  if (time < 0.0) return d1;
  double d2 = edt_min_dist_to_boxes(nbox, box_p0, box_vel, box_scale, pos, time);
  return d1 < d2 ? d1 : d2;
}

/* EDTEnvironment::evaluateEDTWithGrad, src/edt_environment.cpp:75-122: the
 * trilinear value/gradient over corner values min(static distance, distance
 * to the nearest moving box at `time`); time < 0 means static only (:91-94).
 * That file is not part of the reference's build and calls an SDFMap API the
 * in-tree class does not have (getInterpolationData); the interpolation data
 * here are the in-tree ones (sdf_map.cpp:201-219: base index, diff, per-axis
 * clamped corner loads), a corner's position is the centre of its (unclamped)
 * voxel, and a query outside the map returns -1 with a zero gradient as
 * oracle_sdf_query does.  PARITY UNPINNED (no reference output exists). */
double oracle_edt_query(const oracle_sdf *S, int nbox, const double *box_p0, const double *box_vel,
                        const double *box_scale, const double pos[3], double time, double grad[3]) {
  if (!sdf_in_map(S, pos)) {
    grad[0] = grad[1] = grad[2] = 0.0;
    return -1;
  }
  double res = S->resolution, rinv = S->resolution_inv;
  double pos_m[3], idx_pos[3], diff[3];
  int idx[3];
  for (int i = 0; i < 3; ++i) pos_m[i] = pos[i] - 0.5 * res * 1.0;
  sdf_pos_to_index(S, pos_m, idx);
  for (int i = 0; i < 3; ++i) idx_pos[i] = (idx[i] + 0.5) * res + S->origin[i];
  for (int i = 0; i < 3; ++i) diff[i] = (pos[i] - idx_pos[i]) * rinv;

  double values[2][2][2];
  for (int x = 0; x < 2; x++)
    for (int y = 0; y < 2; y++)
      for (int z = 0; z < 2; z++) {
        double d1 = sdf_get_distance(S, idx[0] + x, idx[1] + y, idx[2] + z);
        if (time < 0.0) { /* :91-94 */
          values[x][y][z] = d1;
        } else {
          double pt[3] = {(idx[0] + x + 0.5) * res + S->origin[0], (idx[1] + y + 0.5) * res + S->origin[1],
                          (idx[2] + z + 0.5) * res + S->origin[2]};
          double d2 = edt_min_dist_to_boxes(nbox, box_p0, box_vel, box_scale, pt, time);
          values[x][y][z] = d1 < d2 ? d1 : d2; /* :98 */
        }
      }

  double v00 = (1 - diff[0]) * values[0][0][0] + diff[0] * values[1][0][0]; /* :104-112 */
  double v01 = (1 - diff[0]) * values[0][0][1] + diff[0] * values[1][0][1];
  double v10 = (1 - diff[0]) * values[0][1][0] + diff[0] * values[1][1][0];
  double v11 = (1 - diff[0]) * values[0][1][1] + diff[0] * values[1][1][1];
  double v0 = (1 - diff[1]) * v00 + diff[1] * v10;
  double v1 = (1 - diff[1]) * v01 + diff[1] * v11;
  double dist = (1 - diff[2]) * v0 + diff[2] * v1;

  grad[2] = (v1 - v0) * rinv; /* :114-121 */
  grad[1] = ((1 - diff[2]) * (v10 - v00) + diff[2] * (v11 - v01)) * rinv;
  grad[0] = (1 - diff[2]) * (1 - diff[1]) * (values[1][0][0] - values[0][0][0]);
  grad[0] += (1 - diff[2]) * diff[1] * (values[1][1][0] - values[0][1][0]);
  grad[0] += diff[2] * (1 - diff[1]) * (values[1][0][1] - values[0][0][1]);
  grad[0] += diff[2] * diff[1] * (values[1][1][1] - values[0][1][1]);
  grad[0] *= rinv;
  return dist;
}

/* sdf_map.cpp:80-99: mark the voxel containing pos occupied (if in map). */
int oracle_set_occupancy(const oracle_sdf *S, double *occupancy,
                         const double pos[3], int occ) {
  if (occ != 1 && occ != 0) return -1;
  if (!sdf_in_map(S, pos)) return 0;
  int id[3];
  sdf_pos_to_index(S, pos, id);
  occupancy[(size_t)id[0] * S->grid[1] * S->grid[2] +
            (size_t)id[1] * S->grid[2] + id[2]] = occ;
  return 1;
}

/* sdf_map.cpp:266-308: 1-D lower-envelope squared-distance transform.
 * get/set are expressed through a strided view: f[q*stride]. */
static void fill_esdf(const double *fin, size_t in_stride, double *fout,
                      size_t out_stride, int start, int end, int n,
                      int final_pass, double resolution) {
  int *v = (int *)malloc(sizeof(int) * (size_t)(n > 0 ? n : 1));
  double *z = (double *)malloc(sizeof(double) * (size_t)(n + 1));
  int k = start;
  v[start] = start;
  z[start] = -DBL_MAX;
  z[start + 1] = DBL_MAX;
  for (int q = start + 1; q <= end; q++) {
    k++;
    double s;
    do {
      k--;
      s = ((fin[(size_t)q * in_stride] + q * q) -
           (fin[(size_t)v[k] * in_stride] + v[k] * v[k])) /
          (2 * q - 2 * v[k]);
    } while (s <= z[k]);
    k++;
    v[k] = q;
    z[k] = s;
    z[k + 1] = DBL_MAX;
  }
  k = start;
  for (int q = start; q <= end; q++) {
    while (z[k + 1] < q) k++;
    double val = (q - v[k]) * (q - v[k]) + fin[(size_t)v[k] * in_stride];
    if (final_pass) {
      /* sdf_map.cpp:355-361: min(res*sqrt(val), previous distance) */
      double d = resolution * sqrt(val);
      double old = fout[(size_t)q * out_stride];
      fout[(size_t)q * out_stride] = d < old ? d : old;
    } else {
      fout[(size_t)q * out_stride] = val;
    }
  }
  free(v);
  free(z);
}

/* sdf_map.cpp:310-368 over the full grid (min_vec = 0, max_vec = grid-1).
 * `distance` must hold the previous distances (10000 after a reset,
 * sdf_map.cpp:22,51). */
void oracle_esdf_build(const oracle_sdf *S, const double *occupancy,
                       double *distance) {
  int gx = S->grid[0], gy = S->grid[1], gz = S->grid[2];
  size_t N = (size_t)gx * gy * gz;
  double *tmp0 = (double *)malloc(sizeof(double) * N);
  double *tmp1 = (double *)malloc(sizeof(double) * N);
  double *tmp2 = (double *)malloc(sizeof(double) * N);
  for (size_t i = 0; i < N; ++i) tmp0[i] = occupancy[i] == 1 ? 0 : DBL_MAX;
  for (int x = 0; x < gx; x++)
    for (int y = 0; y < gy; y++) {
      size_t base = (size_t)x * gy * gz + (size_t)y * gz;
      fill_esdf(tmp0 + base, 1, tmp1 + base, 1, 0, gz - 1, gz, 0, 0.0);
    }
  for (int x = 0; x < gx; x++)
    for (int z = 0; z < gz; z++) {
      size_t base = (size_t)x * gy * gz + z;
      fill_esdf(tmp1 + base, (size_t)gz, tmp2 + base, (size_t)gz, 0, gy - 1, gy,
                0, 0.0);
    }
  for (int y = 0; y < gy; y++)
    for (int z = 0; z < gz; z++) {
      size_t base = (size_t)y * gz + z;
      fill_esdf(tmp2 + base, (size_t)gy * gz, distance + base, (size_t)gy * gz,
                0, gx - 1, gx, 1, S->resolution);
    }
  free(tmp0);
  free(tmp1);
  free(tmp2);
}

/* sdf_map.cpp:28-45 / :244-260: the voxel window of (min_pos, max_pos) as resetBuffer(min, max) and setUpdateRange
 * compute it — both clamp the positions to [min_range, max_range], then posToIndex(min_pos) and
 * posToIndex(max_pos - res/2). */
void oracle_window_ids(const oracle_sdf *S, const double min_pos_in[3],
                       const double max_pos_in[3], int min_id[3], int max_id[3]) {
  double min_pos[3], max_pos[3], shifted[3];
  for (int i = 0; i < 3; ++i) {
    min_pos[i] = min_pos_in[i] > S->min_range[i] ? min_pos_in[i] : S->min_range[i];   /* max(min_pos, min_range) */
    max_pos[i] = max_pos_in[i] < S->max_range[i] ? max_pos_in[i] : S->max_range[i];   /* min(max_pos, max_range) */
    shifted[i] = max_pos[i] - S->resolution / 2;
  }
  sdf_pos_to_index(S, min_pos, min_id);
  sdf_pos_to_index(S, shifted, max_id);
}

/* sdf_map.cpp:28-53: resetBuffer(min_pos, max_pos) — occupancy 0 and distance 10000 inside the window only. */
void oracle_reset_window(const oracle_sdf *S, double *occupancy, double *distance,
                         const double min_pos[3], const double max_pos[3]) {
  int lo[3], hi[3];
  oracle_window_ids(S, min_pos, max_pos, lo, hi);
  int gy = S->grid[1], gz = S->grid[2];
  for (int x = lo[0]; x <= hi[0]; ++x)
    for (int y = lo[1]; y <= hi[1]; ++y)
      for (int z = lo[2]; z <= hi[2]; ++z) {
        occupancy[(size_t)x * gy * gz + (size_t)y * gz + z] = 0.0;
        distance[(size_t)x * gy * gz + (size_t)y * gz + z] = 10000;
      }
}

/* sdf_map.cpp:310-368 with the window min_vec .. max_vec that setUpdateRange (:244-264) left behind: the three sweeps
 * run over the window only and see only the window's part of every line; distances outside keep their values, inside
 * they become min(res*sqrt(val), previous) (:355-361).  tmp1 / tmp2 are the object's tmp_buffer1 / tmp_buffer2. */
void oracle_esdf_build_window(const oracle_sdf *S, const double *occupancy,
                              double *distance, const int lo[3], const int hi[3]) {
  int gx = S->grid[0], gy = S->grid[1], gz = S->grid[2];
  size_t N = (size_t)gx * gy * gz;
  double *tmp0 = (double *)malloc(sizeof(double) * N);
  double *tmp1 = (double *)calloc(N, sizeof(double));
  double *tmp2 = (double *)calloc(N, sizeof(double));
  for (size_t i = 0; i < N; ++i) tmp0[i] = occupancy[i] == 1 ? 0 : DBL_MAX;
  int n = gx > gy ? (gx > gz ? gx : gz) : (gy > gz ? gy : gz);
  for (int x = lo[0]; x <= hi[0]; x++)
    for (int y = lo[1]; y <= hi[1]; y++) {
      size_t base = (size_t)x * gy * gz + (size_t)y * gz;
      fill_esdf(tmp0 + base, 1, tmp1 + base, 1, lo[2], hi[2], n, 0, 0.0);
    }
  for (int x = lo[0]; x <= hi[0]; x++)
    for (int z = lo[2]; z <= hi[2]; z++) {
      size_t base = (size_t)x * gy * gz + z;
      fill_esdf(tmp1 + base, (size_t)gz, tmp2 + base, (size_t)gz, lo[1], hi[1], n, 0, 0.0);
    }
  for (int y = lo[1]; y <= hi[1]; y++)
    for (int z = lo[2]; z <= hi[2]; z++) {
      size_t base = (size_t)y * gz + z;
      fill_esdf(tmp2 + base, (size_t)gy * gz, distance + base, (size_t)gy * gz, lo[0], hi[0], n, 1,
                S->resolution);
    }
  free(tmp0);
  free(tmp1);
  free(tmp2);
}

/* ------------------------------------------------------------------ */
/* the hot path                                                       */
/* ------------------------------------------------------------------ */

/* grad_traj_optimizer.cpp:281-432 with :253-279, :451-505, :507-551 and
 * sdf_map.cpp:185-242 underneath.  L: 6m×(3m+3), R: (3m+3)², Df: 3×6,
 * x/grad: 3·num_dp axis-major (:182-187, :428-432).  Returns the cost. */
double oracle_cost_grad(int m, const double *L, const double *R,
                        const double *Df, const double *T,
                        const oracle_params *prm, const oracle_sdf *S,
                        const double *x, double *grad_out) {
  const int num_df = 6, num_dp = 3 * m - 3, nd = num_df + num_dp, n6 = 6 * m;
  double cost_smooth = 0, cost_colli = 0, cost_vel = 0, cost_acc = 0;
  double *g_smooth = dalloc((size_t)3 * num_dp);
  double *g_colli = dalloc((size_t)3 * num_dp);
  double *g_vel = dalloc((size_t)3 * num_dp);
  double *g_acc = dalloc((size_t)3 * num_dp);
  double *d = dalloc((size_t)3 * nd); /* [df;dp] per axis, :302-323 */
  double *coe = dalloc((size_t)m * 18);
  double *tmp = dalloc((size_t)nd);

  for (int a = 0; a < 3; ++a) {
    for (int j = 0; j < 6; ++j) d[a * nd + j] = Df[a * 6 + j];
    for (int j = 0; j < num_dp; ++j) d[a * nd + 6 + j] = x[j + num_dp * a];
  }

  /* smoothness cost d'Rd (:326-327): (d' R) d */
  for (int a = 0; a < 3; ++a) {
    const double *da = d + a * nd;
    for (int j = 0; j < nd; ++j) {
      double s = 0;
      for (int i = 0; i < nd; ++i) s += da[i] * R[i * nd + j];
      tmp[j] = s;
    }
    double q = 0;
    for (int j = 0; j < nd; ++j) q += tmp[j] * da[j];
    cost_smooth += q;
  }
  /* smoothness gradient 2 Rfp' df + 2 Rpp dp (:330-336);
   * Rfp = R[0:6, 6:], Rpp = R[6:, 6:] (qp_generator.cpp:400-403) */
  for (int a = 0; a < 3; ++a) {
    const double *da = d + a * nd;
    for (int j = 0; j < num_dp; ++j) {
      double s1 = 0, s2 = 0;
      for (int i = 0; i < 6; ++i) s1 += (2 * R[i * nd + 6 + j]) * da[i];
      for (int i = 0; i < num_dp; ++i)
        s2 += (2 * R[(6 + j) * nd + 6 + i]) * da[6 + i];
      g_smooth[a * num_dp + j] = s1 + s2;
    }
  }
  /* coefficients coe = L d (:253-279) */
  for (int a = 0; a < 3; ++a) {
    const double *da = d + a * nd;
    for (int r = 0; r < n6; ++r) {
      double s = 0;
      for (int i = 0; i < nd; ++i) s += L[r * nd + i] * da[i];
      coe[(r / 6) * 18 + 6 * a + (r % 6)] = s;
    }
  }

  double *Ldp = dalloc((size_t)6 * num_dp);
  for (int s = 0; s < m; s++) {
    if (fabs(prm->wc) < 1e-4) break; /* :346 */
    for (int r = 0; r < 6; ++r)      /* :348 */
      for (int c = 0; c < num_dp; ++c)
        Ldp[r * num_dp + c] = L[(6 * s + r) * nd + 6 + c];
    double dt = T[s] / 30.0; /* :351 */
    const double *c = coe + s * 18;
    for (double t = 1e-3; t < T[s]; t += dt) { /* :353 */
      /* :451-468, :471-488 — note the float locals */
      double pos[3], vel[3], acc[3];
      for (int a = 0; a < 3; ++a) {
        const double *q = c + 6 * a;
        float p = q[0] + q[1] * t + q[2] * pow(t, 2) + q[3] * pow(t, 3) +
                  q[4] * pow(t, 4) + q[5] * pow(t, 5);
        float v = q[1] + 2 * q[2] * pow(t, 1) + 3 * q[3] * pow(t, 2) +
                  4 * q[4] * pow(t, 3) + 5 * q[5] * pow(t, 4);
        pos[a] = p;
        vel[a] = v;
      }
      double vel_norm =
          sqrt(vel[0] * vel[0] + vel[1] * vel[1] + vel[2] * vel[2]) + 1e-5;

      double dist, gd, cd, grad[3];
      dist = oracle_sdf_query(S, pos, grad);                       /* :363 */
      cd = prm->alpha * exp(-(dist - prm->d0) / prm->r);           /* :509 */
      gd = -(prm->alpha / prm->r) * exp(-(dist - prm->d0) / prm->r); /* :514 */

      double Tm[6], TV[6], TVV[6];
      for (int i = 0; i < 6; ++i) Tm[i] = pow(t, i); /* :544-551 */
      /* T·V with V(i,i+1)=i+1, all else 0 (:104-105) */
      TV[0] = 0;
      for (int i = 1; i < 6; ++i) TV[i] = Tm[i - 1] * i;
      TVV[0] = 0;
      for (int i = 1; i < 6; ++i) TVV[i] = TV[i - 1] * i;

      cost_colli += cd * vel_norm * dt; /* :373 */

      for (int k = 0; k < 3; k++) { /* :376-381 */
        double s1 = gd * grad[k] * cd * vel_norm;
        double s2 = cd * (vel[k] / vel_norm);
        for (int cc = 0; cc < num_dp; ++cc) {
          double a1 = 0, a2 = 0;
          for (int i = 0; i < 6; ++i) {
            a1 += (s1 * Tm[i]) * Ldp[i * num_dp + cc];
            a2 += (s2 * TV[i]) * Ldp[i * num_dp + cc];
          }
          g_colli[k * num_dp + cc] =
              g_colli[k * num_dp + cc] + (a1 + a2) * dt;
        }
      }

      /* dynamic-feasibility block, commented out in the reference
       * (:383-407); executed here only when enable_dyn != 0. */
      if (prm->enable_dyn && prm->step == 2) {
        double cv = 0, ca = 0, gv = 0, ga = 0;
        for (int a = 0; a < 3; ++a) { /* :491-505 */
          const double *q = c + 6 * a;
          float ac = 2 * q[2] + 6 * q[3] * pow(t, 1) + 12 * q[4] * pow(t, 2) +
                     20 * q[5] * pow(t, 3);
          acc[a] = ac;
        }
        for (int k = 0; k < 3; k++) {
          cv = prm->alpha_v * exp((fabs(vel[k]) - prm->v0) / prm->r_v); /* :519 */
          cost_vel += cv * vel_norm * dt;
          ca = prm->alpha_a * exp((fabs(acc[k]) - prm->a0) / prm->r_a); /* :529 */
          cost_acc += ca * vel_norm * dt;
        }
        for (int k = 0; k < 3; k++) {
          gv = (prm->alpha_v / prm->r_v) *
               exp((fabs(vel[k]) - prm->v0) / prm->r_v); /* :524 */
          ga = (prm->alpha_a / prm->r_a) *
               exp((fabs(acc[k]) - prm->a0) / prm->r_a); /* :534 */
          double s1 = gv * vel_norm, s2 = cv * (vel[k] / vel_norm);
          double s3 = ga * vel_norm, s4 = ca * (vel[k] / vel_norm);
          for (int cc = 0; cc < num_dp; ++cc) {
            double a1 = 0, a2 = 0, a3 = 0, a4 = 0;
            for (int i = 0; i < 6; ++i) {
              double l = Ldp[i * num_dp + cc];
              a1 += (s1 * TV[i]) * l;
              a2 += (s2 * TV[i]) * l;
              a3 += (s3 * TVV[i]) * l;
              a4 += (s4 * TV[i]) * l;
            }
            g_vel[k * num_dp + cc] += (a1 + a2) * dt;
            g_acc[k * num_dp + cc] += (a3 + a4) * dt;
          }
        }
      }
    }
  }

  /* :412-418 */
  double ws = prm->ws, wc = prm->wc, wv = 1.0, wa = 1.0;
  if (prm->step == 1) ws = 0.0;
  double cost =
      ws * cost_smooth + wc * cost_colli + wv * cost_vel + wa * cost_acc + 1e-3;
  /* :425-432 */
  for (int a = 0; a < 3; ++a)
    for (int i = 0; i < num_dp; ++i)
      grad_out[i + num_dp * a] =
          (ws * g_smooth[a * num_dp + i] + wc * g_colli[a * num_dp + i] +
           wv * g_vel[a * num_dp + i] + wa * g_acc[a * num_dp + i]) +
          1e-5;

  free(g_smooth);
  free(g_colli);
  free(g_vel);
  free(g_acc);
  free(d);
  free(coe);
  free(tmp);
  free(Ldp);
  return cost;
}

static double now_s(void) {
  struct timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return ts.tv_sec + 1e-9 * ts.tv_nsec;
}

/* Batch driver used by tests and by bench.py's cpu_baseline leg.
 * T: B×m (t_stride = m) or shared (t_stride = 0); Df: B×18; x, grad: B×n.
 * Setup (L, R per trajectory — what setPath does once per problem,
 * grad_traj_optimizer.cpp:88-98) is done first and NOT timed; `reps`
 * passes of the callback over all B trajectories are timed.
 * Returns seconds spent in the callbacks (all reps), or <0 on error. */
double oracle_eval_batch(int B, int m, const double *T, int t_stride,
                         const double *Df, const oracle_params *prm,
                         const oracle_sdf *S, const double *x, double *cost,
                         double *grad, int reps, int nthreads) {
  int nd = 3 * m + 3, n6 = 6 * m, n = 9 * (m - 1);
  size_t lsz = (size_t)n6 * nd, rsz = (size_t)nd * nd;
  int nprob = t_stride ? B : 1;
  double *Ls = dalloc(lsz * nprob), *Rs = dalloc(rsz * nprob);
  int err = 0;
  for (int b = 0; b < nprob; ++b)
    if (oracle_generator(m, T + (size_t)b * t_stride, NULL, NULL, NULL,
                         Ls + lsz * b, Rs + rsz * b) != 0)
      err = 1;
  if (err) {
    free(Ls);
    free(Rs);
    return -1.0;
  }
  if (nthreads < 1) nthreads = 1;
  double t0 = now_s();
  for (int rep = 0; rep < reps; ++rep) {
#ifdef _OPENMP
#pragma omp parallel for num_threads(nthreads) schedule(static)
#endif
    for (int b = 0; b < B; ++b) {
      int p = t_stride ? b : 0;
      cost[b] = oracle_cost_grad(m, Ls + lsz * p, Rs + rsz * p,
                                 Df + (size_t)b * 18, T + (size_t)b * t_stride,
                                 prm, S, x + (size_t)b * n, grad + (size_t)b * n);
    }
  }
  double t1 = now_s();
  free(Ls);
  free(Rs);
  return t1 - t0;
}

/* ------------------------------------------------------------------ */
/* post-processing: PolynomialTraj                                     */
/* ------------------------------------------------------------------ */

/* include/grad_traj_optimization/polynomial_traj.hpp:46-66 — evaluate(t) on
 * coefficients in ASCENDING order c[0..5] (the node reverses them into
 * descending order, src/opti_node.cpp:118-120, and evaluate() pairs
 * tv(order-1-i) = pow(t,i) with them, i.e. a dot product that starts at t^5). */
static double traj_poly_eval(const double *c, double t) {
  double s = 0.0;
  for (int i = 5; i >= 0; --i) s += pow(t, (double)i) * c[i];
  return s;
}

/* out[9] = time_sum, length, jerk, mean_v, max_v, mean_a, max_a, acc_cost, n_samples.
 * coeff: m x 18 (row s = [cx0..5 | cy0..5 | cz0..5]).  dt_sample = 0.01 in the reference.
 * Convention where the reference has UB: evaluate() walks past the last segment when
 * t == time_sum exactly (:48-51); the last segment is extended instead. */
/* PolynomialTraj::getTraj, polynomial_traj.hpp:69-78: the points every dt_sample (0.01 in the reference), the
 * sample time accumulated; returns their number and stores the first max_samples of them. */
int oracle_traj_samples(int m, const double *coeff, const double *T, double dt_sample, int max_samples,
                        double *samples) {
  double time_sum = 0.0;
  for (int i = 0; i < m; ++i) time_sum += T[i];
  int nsamp = 0;
  for (double eval_t = 0.0; eval_t <= time_sum; eval_t += dt_sample) {
    double t = eval_t;
    int idx = 0;
    while (idx < m - 1 && T[idx] <= t) {
      t -= T[idx];
      ++idx;
    }
    if (nsamp < max_samples)
      for (int a = 0; a < 3; ++a) samples[3 * nsamp + a] = traj_poly_eval(coeff + idx * 18 + 6 * a, t);
    ++nsamp;
  }
  return nsamp;
}

void oracle_traj_stats(int m, const double *coeff, const double *T, double dt_sample, double *out) {
  double time_sum = 0.0; /* :37-43 */
  for (int i = 0; i < m; ++i) time_sum += T[i];

  double length = 0.0, pl[3] = {0, 0, 0}; /* getTraj :69-78 + getLength :80-92 */
  int nsamp = 0;
  for (double eval_t = 0.0; eval_t <= time_sum; eval_t += dt_sample) {
    double t = eval_t;
    int idx = 0;
    while (idx < m - 1 && T[idx] <= t) {
      t -= T[idx];
      ++idx;
    }
    double pn[3];
    for (int a = 0; a < 3; ++a) pn[a] = traj_poly_eval(coeff + idx * 18 + 6 * a, t);
    if (nsamp > 0) {
      double dx = pn[0] - pl[0], dy = pn[1] - pl[1], dz = pn[2] - pl[2];
      length += sqrt(dx * dx + dy * dy + dz * dz);
    }
    pl[0] = pn[0]; pl[1] = pn[1]; pl[2] = pn[2];
    ++nsamp;
  }

  double acc_cost = 0.0; /* getAccCost :96-109: um = 2*c[order-3] with descending c = 2*c2 */
  for (int s = 0; s < m; ++s) {
    double ux = 2 * coeff[s * 18 + 2], uy = 2 * coeff[s * 18 + 8], uz = 2 * coeff[s * 18 + 14];
    acc_cost += (ux * ux + uy * uy + uz * uz) * T[s];
  }

  double jerk = 0.0; /* getJerk :111-142 */
  for (int s = 0; s < m; ++s) {
    double M[6][6];
    memset(M, 0, sizeof(M));
    for (double i = 3; i < 6; i += 1)
      for (double j = 3; j < 6; j += 1)
        M[(int)i][(int)j] = i * (i - 1) * (i - 2) * j * (j - 1) * (j - 2) * pow(T[s], i + j - 5) / (i + j - 5);
    for (int a = 0; a < 3; ++a) {
      const double *c = coeff + s * 18 + 6 * a;
      double acc = 0.0; /* (c' M) c */
      for (int j = 3; j < 6; ++j) {
        double col = 0.0;
        for (int i = 3; i < 6; ++i) col += c[i] * M[i][j];
        acc += col * c[j];
      }
      jerk += acc;
    }
  }

  /* getMeanAndMaxVel :144-173, getMeanAndMaxAcc :175-204 — tv(i) = pow(ts, i): the
   * segment duration, not eval_t (kept as written) */
  double mean_v = 0.0, max_v = -1.0, mean_a = 0.0, max_a = -1.0;
  int num_v = 0, num_a = 0;
  for (int s = 0; s < m; ++s) {
    double vel[3], acc[3];
    for (int a = 0; a < 3; ++a) {
      const double *c = coeff + s * 18 + 6 * a;
      double sv = 0.0, sa = 0.0;
      for (int i = 0; i < 5; ++i) sv += pow(T[s], (double)i) * ((double)(i + 1) * c[i + 1]);
      for (int i = 0; i < 4; ++i) sa += pow(T[s], (double)i) * ((double)((i + 2) * (i + 1)) * c[i + 2]);
      vel[a] = sv;
      acc[a] = sa;
    }
    double vn = sqrt(vel[0] * vel[0] + vel[1] * vel[1] + vel[2] * vel[2]);
    double an = sqrt(acc[0] * acc[0] + acc[1] * acc[1] + acc[2] * acc[2]);
    for (double eval_t = 0.0; eval_t < T[s]; eval_t += dt_sample) {
      mean_v += vn;
      if (vn > max_v) max_v = vn;
      ++num_v;
      mean_a += an;
      if (an > max_a) max_a = an;
      ++num_a;
    }
  }
  mean_v = mean_v / (double)num_v;
  mean_a = mean_a / (double)num_a;

  out[0] = time_sum; out[1] = length; out[2] = jerk; out[3] = mean_v; out[4] = max_v;
  out[5] = mean_a; out[6] = max_a; out[7] = acc_cost; out[8] = (double)nsamp;
}

/* getCoefficientFromDerivative (src/grad_traj_optimizer.cpp:253-279): coe = L d, m x 18 */
void oracle_coefficients(int m, const double *L, const double *Df, const double *x, double *coe) {
  const int num_dp = 3 * m - 3, nd = 6 + num_dp, n6 = 6 * m;
  double *d = dalloc((size_t)nd);
  for (int a = 0; a < 3; ++a) {
    for (int j = 0; j < 6; ++j) d[j] = Df[a * 6 + j];
    for (int j = 0; j < num_dp; ++j) d[6 + j] = x[j + num_dp * a];
    for (int r = 0; r < n6; ++r) {
      double s = 0;
      for (int i = 0; i < nd; ++i) s += L[r * nd + i] * d[i];
      coe[(r / 6) * 18 + 6 * a + (r % 6)] = s;
    }
  }
  free(d);
}
