/* asan_main.c — drives every entry point of the CPU restatement under
 * AddressSanitizer + UBSan (CPU build only; test infrastructure).
 * Built by `make -C oracle _build/oracle_asan`, run by tests/test_oracle.py. */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "gtop_oracle.h"

static double urand(unsigned *s) {
  *s = *s * 1664525u + 1013904223u;
  return (*s >> 8) / 16777216.0;
}

int main(void) {
  unsigned seed = 12345u;
  const int grid[3] = {24, 20, 16};
  const double origin[3] = {-2.4, -2.0, 0.0}, res = 0.2;
  const size_t nvox = (size_t)grid[0] * grid[1] * grid[2];
  double *dist = (double *)malloc(sizeof(double) * nvox), *occ = (double *)calloc(nvox, sizeof(double));
  for (size_t i = 0; i < nvox; ++i) dist[i] = 10000.0;
  oracle_sdf S;
  oracle_sdf_init(&S, origin, res, grid, dist);
  for (int k = 0; k < 200; ++k) {
    double p[3] = {origin[0] + urand(&seed) * 5.5 - 0.3, origin[1] + urand(&seed) * 4.6 - 0.3, urand(&seed) * 3.6 - 0.2};
    oracle_set_occupancy(&S, occ, p, 1);   /* some of them out of the map on purpose */
  }
  oracle_esdf_build(&S, occ, dist);
  {   /* the windowed update (sdf_map.cpp:28-53, :244-264, :310-368): boxes inside, across the map's border, the whole
       * map, a sliver, an empty one (max below min) */
    const double boxes[5][6] = {{-1.0, -1.0, 0.5, 0.8, 1.1, 2.0},   {-9.0, -9.0, -9.0, -1.5, 0.0, 1.0},
                                {-99.0, -99.0, -99.0, 99.0, 99.0, 99.0}, {0.1, 0.1, 0.1, 0.15, 1.9, 3.1},
                                {1.0, 1.0, 1.0, 0.5, 0.5, 0.5}};
    for (int b = 0; b < 5; ++b) {
      int lo[3], hi[3];
      oracle_reset_window(&S, occ, dist, boxes[b], boxes[b] + 3);
      for (int k = 0; k < 20; ++k) {
        double p[3] = {origin[0] + urand(&seed) * 5.5 - 0.3, origin[1] + urand(&seed) * 4.6 - 0.3, urand(&seed) * 3.6 - 0.2};
        oracle_set_occupancy(&S, occ, p, 1);
      }
      oracle_window_ids(&S, boxes[b], boxes[b] + 3, lo, hi);
      oracle_esdf_build_window(&S, occ, dist, lo, hi);
    }
  }
  {   /* static field + moving boxes, in and out of the map, t < 0 and t >= 0 */
    double p0[6] = {origin[0] + 1.0, origin[1] + 1.0, 1.0, origin[0] + 3.0, origin[1] + 2.0, 2.0};
    double vel[6] = {0.3, -0.2, 0.0, -0.5, 0.1, 0.1}, scale[6] = {0.5, 0.7, 0.9, 1.1, 0.4, 0.6}, g3[3];
    for (int k = 0; k < 300; ++k) {
      double p[3] = {origin[0] + urand(&seed) * 5.5 - 0.3, origin[1] + urand(&seed) * 4.6 - 0.3, urand(&seed) * 3.6 - 0.2};
      (void)oracle_edt_query(&S, 2, p0, vel, scale, p, urand(&seed) * 4.0 - 1.0, g3);
    }
  }

  int rc = 0;
  for (int m = 2; m <= 7; ++m) {
    const int npts = m + 1, n = 9 * (m - 1), nd = 3 * m + 3, n6 = 6 * m;
    double *path = (double *)malloc(sizeof(double) * 3 * npts), *T = (double *)malloc(sizeof(double) * m);
    for (int i = 0; i < npts; ++i) {
      path[3 * i + 0] = origin[0] + 0.5 + urand(&seed) * 3.5;
      path[3 * i + 1] = origin[1] + 0.5 + urand(&seed) * 2.8;
      path[3 * i + 2] = 0.4 + urand(&seed) * 2.2;
    }
    oracle_segment_time(npts, path, 1.8, 0.3, T);
    if (m == 3) T[1] = 0.03;     /* the 29-sample case */
    if (m == 4) T[2] = 0.0009;   /* no sample at all */
    double *L = (double *)malloc(sizeof(double) * n6 * nd), *R = (double *)malloc(sizeof(double) * nd * nd);
    if (oracle_generator(m, T, NULL, NULL, NULL, L, R) != 0) rc = 1;
    double Df[18], *Dp = (double *)malloc(sizeof(double) * n), *x = (double *)malloc(sizeof(double) * n),
           *g = (double *)malloc(sizeof(double) * n);
    const double zero[3] = {0, 0, 0};
    oracle_initial_d(npts, path, zero, zero, Df, Dp);
    for (int i = 0; i < n; ++i) x[i] = Dp[i] + (urand(&seed) - 0.5) * 0.3;
    if (m == 5) x[0] += 50.0;    /* samples leave the map */
    for (int dyn = 0; dyn < 2; ++dyn)
      for (int step = 1; step <= 2; ++step) {
        oracle_params p = {1.0, 5.0, 10.0, 0.5, 0.8, 2.0, 1.5, 2.5, 1.5, 1.5, 3.5, step, dyn};
        const double c = oracle_cost_grad(m, L, R, Df, T, &p, &S, x, g);
        if (!(c >= 1e-3)) rc = 1;
      }
    double *coe = (double *)malloc(sizeof(double) * m * 18), st[9];
    oracle_coefficients(m, L, Df, x, coe);
    oracle_traj_stats(m, coe, T, 0.01, st);
    if (!(st[0] > 0) || !(st[8] >= 1)) rc = 1;
    /* the batch driver, shared and per-trajectory times */
    double cost2[2], *x2 = (double *)malloc(sizeof(double) * 2 * n), *g2 = (double *)malloc(sizeof(double) * 2 * n),
           Df2[36], *T2 = (double *)malloc(sizeof(double) * 2 * m);
    memcpy(x2, x, sizeof(double) * n); memcpy(x2 + n, x, sizeof(double) * n);
    memcpy(Df2, Df, sizeof(Df)); memcpy(Df2 + 18, Df, sizeof(Df));
    memcpy(T2, T, sizeof(double) * m); memcpy(T2 + m, T, sizeof(double) * m);
    oracle_params p = {1.0, 5.0, 10.0, 0.5, 0.8, 0.0, 1.5, 2.5, 0.0, 1.5, 3.5, 2, 0};
    if (oracle_eval_batch(2, m, T2, m, Df2, &p, &S, x2, cost2, g2, 1, 1) < 0) rc = 1;
    if (oracle_eval_batch(2, m, T, 0, Df2, &p, &S, x2, cost2, g2, 1, 1) < 0) rc = 1;
    free(path); free(T); free(L); free(R); free(Dp); free(x); free(g); free(coe); free(x2); free(g2); free(T2);
  }
  if (oracle_generator(1, (double[]){1.0}, NULL, NULL, NULL, NULL, NULL) == 0) rc = 1;   /* m = 1 must be rejected */
  free(dist); free(occ);
  printf(rc ? "oracle_asan: FAILED\n" : "oracle_asan: ok\n");
  return rc;
}
