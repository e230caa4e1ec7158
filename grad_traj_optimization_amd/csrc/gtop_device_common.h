// gtop_device_common.h — device-side helpers shared by the evaluation kernel
// (gtop_kernels.hip) and the optimizer kernels (gtop_mma.hip).
#ifndef GTOP_DEVICE_COMMON_H_
#define GTOP_DEVICE_COMMON_H_

#include <hip/hip_runtime.h>

#include "gtop_kernels.h"

// Wavefront sum on the DPP cross-lane path (no LDS round trips): quad swaps,
// row shifts, then the two row broadcasts; every lane's contribution ends up in
// lane 63, which is read back with readlane.  __shfl_xor would go through
// ds_bpermute (~100+ cycles per step); this is ~10.
template <int CTRL, int ROW_MASK = 0xf>
__device__ __forceinline__ float gtop_dpp_move(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, ROW_MASK, 0xf, true));
}
template <int CTRL, int ROW_MASK = 0xf>
__device__ __forceinline__ double gtop_dpp_move(double v) {
  const unsigned long long u = __builtin_bit_cast(unsigned long long, v);
  const unsigned lo = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(unsigned)u, CTRL, ROW_MASK, 0xf, true);
  const unsigned hi = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(unsigned)(u >> 32), CTRL, ROW_MASK, 0xf, true);
  return __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
}
template <typename R>
__device__ __forceinline__ R gtop_wave_sum(R v) {
  v += gtop_dpp_move<0xb1>(v);         // quad_perm [1,0,3,2]
  v += gtop_dpp_move<0x4e>(v);         // quad_perm [2,3,0,1]
  v += gtop_dpp_move<0x114>(v);        // row_shr:4
  v += gtop_dpp_move<0x118>(v);        // row_shr:8   -> lane 15 of each row holds the row sum
  v += gtop_dpp_move<0x142, 0xa>(v);   // row_bcast:15 into rows 1 and 3
  v += gtop_dpp_move<0x143, 0xc>(v);   // row_bcast:31 into rows 2 and 3 -> lane 63 holds the total
  if constexpr (sizeof(R) == 8) {
    const unsigned long long u = __builtin_bit_cast(unsigned long long, v);
    const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)u, 63);
    const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(u >> 32), 63);
    return __builtin_bit_cast(R, ((unsigned long long)hi << 32) | lo);
  } else {
    return __builtin_bit_cast(R, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
  }
}

// ---------------------------------------------------------------------------
// One CCSA-MMA update of trajectory b by one wavefront (lanes stride its n
// variables): consumes f(xcur) and its gradient, decides accept / inner-done,
// adapts rho and the asymptotes, and writes the next trial point into xcur.
// Same arithmetic as csrc/mma.hpp (Svanberg 2002 as in NLopt 2.5.0's mma.c,
// zero nonlinear constraints: the separable minimiser is closed form).
// state[b]: 0 = first evaluation pending, 1 = inside an inner loop.
// ---------------------------------------------------------------------------
// The update's view of ONE trajectory's state: its n-vectors (base pointers of this trajectory's rows — in global
// memory for the launch forms that keep the state there between launches, in LDS for the one-launch loop, which then
// pays no global round trip for them between its passes) and its scalars.  `xcur` is what the update reads;
// `xcur_out` is where the next trial point is ALSO written when the evaluation reads it from elsewhere (the global
// row, for the loop) — the same pointer otherwise.
struct GtopMmaVecs {
  double *x, *xcur, *xprev, *xprevprev, *dfdx, *sigma;
  const double *lb, *ub;
  double *xcur_out;
};
struct GtopMmaScalars {
  double rho, minf, gval, wval, fprev;
  int k, state, nevals;
};

// NLopt's relstop (stop.c; see mma.hpp)
__device__ __forceinline__ bool gtop_mma_relstop(double vold, double vnew, double reltol) {
  if (isinf(vold)) return false;
  return fabs(vnew - vold) < reltol * (fabs(vnew) + fabs(vold)) * 0.5 || (reltol > 0 && vnew == vold);
}

__device__ __forceinline__ void gtop_mma_separable_step(int n, int lane, const double *x, const double *dfdx,
                                                        const double *sigma, double rho, const double *lb,
                                                        const double *ub, double *xcur, double *xcur_out, double &g,
                                                        double &w) {
  g = 0.0;
  w = 0.0;
  for (int j = lane; j < n; j += 64) {
    const double sg = sigma[j], xj = x[j];
    if (sg == 0.0) {
      xcur[j] = xj;
      if (xcur_out != xcur) xcur_out[j] = xj;
      continue;
    }
    const double df = dfdx[j];
    const double sigma2 = sg * sg;
    const double u = df * sigma2;
    const double v = fabs(df) * sg + 0.5 * rho;
    const double q = u / (v * sg);
    double dx = (u / v) / (-1.0 - sqrt(fabs(1.0 - q * q)));
    double xc = xj + dx;
    xc = xc > ub[j] ? ub[j] : (xc < lb[j] ? lb[j] : xc);
    const double hi = xj + 0.9 * sg, lo = xj - 0.9 * sg;
    xc = xc > hi ? hi : (xc < lo ? lo : xc);
    xcur[j] = xc;
    if (xcur_out != xcur) xcur_out[j] = xc;
    dx = xc - xj;
    const double dx2 = dx * dx;
    const double denominv = 1.0 / (sigma2 - dx2);
    g += (df * (sigma2 * dx) + (fabs(df) * sg + 0.5 * rho) * dx2) * denominv;
    w += 0.5 * dx2 * denominv;
  }
}

// One update on the state `v` / `sc` (see gtop_mma_update_trajectory).  st: the stop tolerances only.
// gcur: this trajectory's gradient at xcur (n values; global or LDS).
// G: double, or float when the evaluation ran in fp32 (gtop_set_optimizer_precision): the state stays fp64.
template <typename G>
__device__ __forceinline__ void gtop_mma_update_core(const GtopMmaState &st, const GtopMmaVecs &v, GtopMmaScalars &sc,
                                                     int n, int lane, double fcur, const G *gcur) {
  if (sc.state >= 3) return;   // stopped (wavefront-uniform): the trajectory stays as it was left
  double rho = sc.rho, minf = sc.minf;
  int k = sc.k;
  const int nevals = sc.nevals + 1;   // this evaluation
  bool new_outer;

  if (sc.state == 0) {
    // f(x0): base point = start (mma.hpp: first evaluation)
    minf = fcur;
    for (int j = lane; j < n; j += 64) {
      v.x[j] = v.xcur[j];
      v.dfdx[j] = (double)gcur[j];
    }
    new_outer = true;
  } else {
    const double gval = sc.gval, wval = sc.wval;
    const bool inner_done = gval >= fcur;
    if (fcur < minf) {   // accept: new base point
      minf = fcur;
      for (int j = lane; j < n; j += 64) {
        v.x[j] = v.xcur[j];
        v.dfdx[j] = (double)gcur[j];
      }
    }
    if (inner_done) {
      // stop rules, where the host twin has them (mma.hpp; NLopt's mma.c): after the inner loop, on the last
      // evaluated f against the f the outer iteration started from, and on xcur against xprev — x after f, its
      // verdict standing when both hold — and not at all when this evaluation is the last one allowed: NLopt looks
      // at the evaluation limit first, so that evaluation reports MAXEVAL (the caller's default for a running state)
      int stop = 0;
      const double fprev = sc.fprev;
      const bool at_cap = st.max_evals > 0 && nevals >= st.max_evals;
      if (!at_cap && gtop_mma_relstop(fprev, fcur, st.ftol_rel)) stop = GTOP_MMA_FTOL_REACHED;
      if (!at_cap && st.xtol_rel > 0) {
        bool all = true;
        for (int j = lane; j < n; j += 64) all = all && gtop_mma_relstop(v.xprev[j], v.xcur[j], st.xtol_rel);
        if (__all(all)) stop = GTOP_MMA_XTOL_REACHED;
      }
      if (stop) {
        sc.minf = minf;
        sc.nevals = nevals;
        sc.state = stop;
        return;
      }
      // end of the outer iteration: relax rho, adapt the asymptotes
      rho = fmax(0.1 * rho, 1e-5);
      if (k > 1) {
        for (int j = lane; j < n; j += 64) {
          const double dx2 = (v.xcur[j] - v.xprev[j]) * (v.xprev[j] - v.xprevprev[j]);
          const double gam = dx2 < 0 ? 0.7 : (dx2 > 0 ? 1.2 : 1.0);
          double s = v.sigma[j] * gam;
          const double range = v.ub[j] - v.lb[j];
          if (!isinf(v.ub[j]) && !isinf(v.lb[j])) {
            s = fmin(s, 10 * range);
            s = fmax(s, 0.01 * range);
          }
          v.sigma[j] = s;
        }
      }
      new_outer = true;
    } else {
      if (fcur > gval) rho = fmin(10 * rho, 1.1 * (rho + (fcur - gval) / wval));
      new_outer = false;
    }
  }
  if (new_outer) {
    ++k;
    for (int j = lane; j < n; j += 64) {
      if (k > 1) v.xprevprev[j] = v.xprev[j];
      v.xprev[j] = v.xcur[j];
    }
  }
  // every lane's writes above are to its own j; the step below reads x/dfdx/sigma
  // at the same j only, so no cross-lane hazard
  double g, w;
  gtop_mma_separable_step(n, lane, v.x, v.dfdx, v.sigma, rho, v.lb, v.ub, v.xcur, v.xcur_out, g, w);
  g = gtop_wave_sum(g);
  w = gtop_wave_sum(w);
  sc.gval = minf + g;
  sc.wval = w;
  sc.rho = rho;
  sc.minf = minf;
  sc.k = k;
  sc.state = 1;
  sc.nevals = nevals;
  if (new_outer) sc.fprev = fcur;
}

__device__ __forceinline__ GtopMmaScalars gtop_mma_load_scalars(const GtopMmaState &st, int b) {
  GtopMmaScalars sc;
  sc.rho = st.rho[b]; sc.minf = st.minf[b]; sc.gval = st.gval[b]; sc.wval = st.wval[b]; sc.fprev = st.fprev[b];
  sc.k = st.k[b]; sc.state = st.state[b]; sc.nevals = st.nevals[b];
  return sc;
}
__device__ __forceinline__ void gtop_mma_store_scalars(const GtopMmaState &st, int b, const GtopMmaScalars &sc) {
  st.rho[b] = sc.rho; st.minf[b] = sc.minf; st.gval[b] = sc.gval; st.wval[b] = sc.wval; st.fprev[b] = sc.fprev;
  st.k[b] = sc.k; st.state[b] = sc.state; st.nevals[b] = sc.nevals;
}

// The update with the state where the multi-launch forms keep it: global memory, read and written in place.
__device__ __forceinline__ void gtop_mma_update_trajectory(const GtopMmaState &st, int b, int n, int lane,
                                                           double fcur, const double *gcur) {
  const size_t o = (size_t)b * n;
  const GtopMmaVecs v = {st.x + o, st.xcur + o, st.xprev + o, st.xprevprev + o, st.dfdx + o, st.sigma + o,
                         st.lb + o, st.ub + o, st.xcur + o};
  GtopMmaScalars sc = gtop_mma_load_scalars(st, b);
  if (sc.state >= 3) return;
  gtop_mma_update_core(st, v, sc, n, lane, fcur, gcur);
  if (lane == 0) gtop_mma_store_scalars(st, b, sc);
}

#endif  // GTOP_DEVICE_COMMON_H_
