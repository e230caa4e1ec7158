// gtop_setup.hip — per-problem setup and post-processing on gfx950, batched.
//
// (1) setup_paths: what GradTrajOptimizer::setPath leaves behind for the
//     callback (src/grad_traj_optimizer.cpp:67-110 of
//     EpicOne1/grad_traj_optimization): segment_time (:73-81), Df and the
//     straight-line Dp (src/qp_generator.cpp:199-221, :407-451), for B
//     waypoint lists at once.  L and R are never formed: the evaluation kernel
//     derives what it needs from segment_time.
// (2) eval_trajectories: the post-processing the reference's node runs on the
//     optimised polynomials (include/grad_traj_optimization/polynomial_traj.hpp:
//     getTraj/getLength :69-92, getAccCost :96-109, getJerk :111-142,
//     getMeanAndMaxVel :144-173, getMeanAndMaxAcc :175-204), one lane per
//     trajectory, statement order kept (including the functions' quirks, see
//     the comments) so that results match the CPU restatement.
// Both are setup/report-side, HBM/latency bound, nowhere near the hot path.

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "gtop_device_common.h"
#include "gtop_kernels.h"

#pragma clang fp contract(off)   // keep the reference's unfused arithmetic (bit-exact setup)

namespace {

__global__ void __launch_bounds__(256)
setup_paths_kernel(int B, int m, const double *__restrict__ wp, double mean_v, double init_time,
                   double *__restrict__ T, double *__restrict__ Df, double *__restrict__ x0) {
  const int ndp = 3 * m - 3, n = 3 * ndp;
  const int per = m + 18 + n;   // outputs per trajectory
  const size_t total = (size_t)B * per;
  for (size_t q = (size_t)blockIdx.x * blockDim.x + threadIdx.x; q < total; q += (size_t)gridDim.x * blockDim.x) {
    const int b = (int)(q / per);
    int i = (int)(q - (size_t)b * per);
    const double *p = wp + (size_t)b * (m + 1) * 3;
    if (i < m) {
      // :73-81 — `i == segment_time.size()` never holds inside the loop: only segment 0 gets init_time
      const double dx = p[i * 3] - p[(i + 1) * 3], dy = p[i * 3 + 1] - p[(i + 1) * 3 + 1],
                   dz = p[i * 3 + 2] - p[(i + 1) * 3 + 2];
      const double len = sqrt(dx * dx + dy * dy + dz * dz);
      T[(size_t)b * m + i] = (i == 0 || i == m) ? len / mean_v + init_time : len / mean_v;
      continue;
    }
    i -= m;
    if (i < 18) {
      // getInitialD, src/qp_generator.cpp:418-431: [p_start, 0, 0, p_end, 0, 0] per axis
      const int axis = i / 6, j = i - axis * 6;
      Df[(size_t)b * 18 + i] = (j == 0) ? p[axis] : (j == 3 ? p[m * 3 + axis] : 0.0);
      continue;
    }
    i -= 18;
    {
      // :433-439 with the straight-line D of :204-221: interior positions, zero vel/acc;
      // layout i + axis*num_dp (src/grad_traj_optimizer.cpp:182-187)
      const int axis = i / ndp, c = i - axis * ndp, w = c / 3 + 1, der = c - 3 * (w - 1);
      x0[(size_t)b * n + i] = (der == 0) ? p[w * 3 + axis] : 0.0;
    }
  }
}

// value of sum_j c[j] t^j the way PolynomialTraj::evaluate forms it
// (polynomial_traj.hpp:57-66): tv(order-1-i) = pow(t, i), pt = tv . c_descending
// The powers by multiplication: pow(t, i) for i <= 5 is the correctly rounded t^i on the host and 1-2 ulp off
// that in a product chain (the device's pow() is itself only within an ulp), far inside the 1e-9 the points are
// held to — and a twelfth of the instructions: 18 pow() calls were ~95 % of this kernel (176 -> 31 us for 1 024
// trajectories / 538 k samples).
__device__ __forceinline__ double poly_eval(const double *c, double t) {
  const double t2 = t * t, t3 = t2 * t, t4 = t2 * t2, t5 = t4 * t;
  double s = 0.0;
  s += t5 * c[5];   // dot over descending powers: t^5 c5 first
  s += t4 * c[4];
  s += t3 * c[3];
  s += t2 * c[2];
  s += t * c[1];
  s += c[0];        // pow(t, 0) = 1
  return s;
}

// PolynomialTraj's evaluation of one optimised trajectory (polynomial_traj.hpp:37-204), one WAVEFRONT per
// trajectory:
//   * getTraj / getLength (:69-92): the samples every dt_sample are spread over the lanes, 64 at a time.  The
//     reference accumulates the sample time (eval_t += 0.01) and tests eval_t <= time_sum, so the time of sample
//     k is that sum, not k*dt: every chunk starts from the carried accumulated value and lane l adds dt l times
//     (64 predicated additions per chunk) — sample count and sample
//     times are the reference's exactly.  A lane's neighbour's point arrives through a wavefront shuffle, the
//     previous chunk's last point through a broadcast; the length is a wavefront sum (summation order differs from
//     the serial sum: ~1e-16).
//   * getAccCost, getJerk, getMeanAndMaxVel/Acc (:96-204): one lane per segment, then wavefront sums / maxima.
//     Quirk kept: the velocity/acceleration "samples" use pow(ts, i) — the segment DURATION — so each is the
//     segment's end-point value, counted once per accumulated eval_t < ts.
// samples (may be NULL): [B][max_samples][3], the getTraj points; stats[8] = their number (also when it exceeds
// max_samples: then only the first max_samples are stored).
// Bound: VALU (the accumulated sample times: 63 dependent additions per chunk); 24 B per sample out.
__global__ void __launch_bounds__(64)
eval_trajectories_kernel(int B, int m, const double *__restrict__ coeff, const double *__restrict__ T,
                         int t_stride, double dt_sample, double *__restrict__ out /*[B][GTOP_TRAJ_STATS]*/,
                         double *__restrict__ samples, int max_samples) {
  const int b = blockIdx.x, lane = threadIdx.x;
  if (b >= B) return;
  const double *cf = coeff + (size_t)b * m * 18;   // row s = [cx0..5 | cy0..5 | cz0..5], ascending powers
  const double *ts = T + (size_t)b * t_stride;
  double time_sum = 0.0;                            // init(), :37-43
  for (int s = 0; s < m; ++s) time_sum += ts[s];

  // ---- getTraj + getLength ----
  double length = 0.0, base_t = 0.0, carry[3] = {0, 0, 0};
  int nsamp = 0;
  bool first_chunk = true;
  while (base_t <= time_sum) {                      // wave-uniform: base_t is the time of the chunk's sample 0
    double eval_t = base_t;
    for (int i = 0; i < lane; ++i) eval_t += dt_sample;          // the accumulated time of sample chunk*64 + lane
    double next_base = base_t;
    for (int i = 0; i < 64; ++i) next_base += dt_sample;
    const bool live = eval_t <= time_sum;           // monotone in the lane index
    double pn[3] = {0, 0, 0};
    if (live) {
      double t = eval_t;
      int idx = 0;
      while (idx < m - 1 && ts[idx] <= t) {   // :48-51; the reference walks off the end when t == time_sum exactly,
        t -= ts[idx];                          // here the last segment is extended instead
        ++idx;
      }
      for (int a = 0; a < 3; ++a) pn[a] = poly_eval(cf + idx * 18 + 6 * a, t);
      if (samples && nsamp + lane < max_samples) {
        double *o = samples + ((size_t)b * max_samples + nsamp + lane) * 3;
        o[0] = pn[0]; o[1] = pn[1]; o[2] = pn[2];
      }
    }
    // previous point: lane - 1's, or the last point of the previous chunk
    double pl[3];
    for (int a = 0; a < 3; ++a) {
      const double up = __shfl_up(pn[a], 1);
      pl[a] = lane == 0 ? carry[a] : up;
    }
    double seg = 0.0;
    if (live && !(first_chunk && lane == 0)) {
      const double dx = pn[0] - pl[0], dy = pn[1] - pl[1], dz = pn[2] - pl[2];
      seg = sqrt(dx * dx + dy * dy + dz * dz);
    }
    length += gtop_wave_sum(seg);
    const unsigned long long mask = __ballot(live);
    const int cnt = __popcll(mask);
    nsamp += cnt;
    for (int a = 0; a < 3; ++a) carry[a] = __shfl(pn[a], cnt - 1);   // (cnt >= 1: lane 0 is live)
    first_chunk = false;
    base_t = next_base;
  }

  // ---- per-segment quantities: lane s < m (m <= 64 per pass) ----
  double acc_cost = 0.0, jerk = 0.0, sum_v = 0.0, sum_a = 0.0, max_v = -1.0, max_a = -1.0;
  int num = 0;
  for (int s0 = 0; s0 < m; s0 += 64) {
    const int s = s0 + lane;
    double ac = 0.0, jk = 0.0, sv_tot = 0.0, sa_tot = 0.0, vn = -1.0, an = -1.0;
    int c = 0;
    if (s < m) {
      const double Ts = ts[s];
      const double Tp[6] = {1.0, Ts, Ts * Ts, Ts * Ts * Ts, (Ts * Ts) * (Ts * Ts), (Ts * Ts) * (Ts * Ts) * Ts};   // pow(Ts, i)
      // getAccCost (:96-109): um = 2 * (coefficient of t^2) = a(0) per segment
      const double ux = 2 * cf[s * 18 + 2], uy = 2 * cf[s * 18 + 8], uz = 2 * cf[s * 18 + 14];
      ac = (ux * ux + uy * uy + uz * uz) * Ts;
      // getJerk (:111-142): c' M c with M(i,j) = i(i-1)(i-2) j(j-1)(j-2) ts^(i+j-5) / (i+j-5), i,j = 3..5
      for (int a = 0; a < 3; ++a) {
        const double *cc = cf + s * 18 + 6 * a;
        double acc = 0.0;
        for (int j = 3; j < 6; ++j) {      // (c' M)(j) then . c, as Eigen evaluates c.transpose() * M * c
          double col = 0.0;
          for (int i = 3; i < 6; ++i) {
            const double di = i, dj = j;
            col += cc[i] * (di * (di - 1) * (di - 2) * dj * (dj - 1) * (dj - 2) * Tp[i + j - 5] / (di + dj - 5));
          }
          acc += col * cc[j];
        }
        jk += acc;
      }
      // getMeanAndMaxVel / Acc (:144-204)
      double vel[3], acc3[3];
      for (int a = 0; a < 3; ++a) {
        const double *cc = cf + s * 18 + 6 * a;
        double sv = 0.0, sa = 0.0;
        for (int i = 0; i < 5; ++i) sv += Tp[i] * ((double)(i + 1) * cc[i + 1]);
        for (int i = 0; i < 4; ++i) sa += Tp[i] * ((double)((i + 2) * (i + 1)) * cc[i + 2]);
        vel[a] = sv;
        acc3[a] = sa;
      }
      vn = sqrt(vel[0] * vel[0] + vel[1] * vel[1] + vel[2] * vel[2]);
      an = sqrt(acc3[0] * acc3[0] + acc3[1] * acc3[1] + acc3[2] * acc3[2]);
      for (double eval_t = 0.0; eval_t < Ts; eval_t += dt_sample) {   // accumulated, as the reference's loops
        sv_tot += vn;
        sa_tot += an;
        ++c;
      }
      if (c == 0) vn = an = -1.0;   // a segment without a sample does not enter the maxima
    }
    acc_cost += gtop_wave_sum(ac);
    jerk += gtop_wave_sum(jk);
    sum_v += gtop_wave_sum(sv_tot);
    sum_a += gtop_wave_sum(sa_tot);
    num += (int)gtop_wave_sum((double)c);
    for (int off = 32; off > 0; off >>= 1) {
      vn = fmax(vn, __shfl_xor(vn, off));
      an = fmax(an, __shfl_xor(an, off));
    }
    max_v = fmax(max_v, vn);
    max_a = fmax(max_a, an);
  }
  if (lane == 0) {
    double *o = out + (size_t)b * GTOP_TRAJ_STATS;
    o[0] = time_sum; o[1] = length; o[2] = jerk; o[3] = sum_v / (double)num; o[4] = max_v;
    o[5] = sum_a / (double)num; o[6] = max_a; o[7] = acc_cost; o[8] = (double)nsamp;
  }
}

// coefficients from derivatives for B trajectories (getCoefficientFromDerivative,
// src/grad_traj_optimizer.cpp:253-279): one lane per (trajectory, segment, axis)
__global__ void __launch_bounds__(256)
coefficients_kernel(int B, int m, const double *__restrict__ x, const double *__restrict__ Df,
                    const double *__restrict__ T, int t_stride, double *__restrict__ coeff) {
  const int ndp = 3 * m - 3, n = 3 * ndp;
  const size_t total = (size_t)B * m * 3;
  for (size_t q = (size_t)blockIdx.x * blockDim.x + threadIdx.x; q < total; q += (size_t)gridDim.x * blockDim.x) {
    const int b = (int)(q / (3 * m));
    const int r = (int)(q - (size_t)b * 3 * m), s = r / 3, k = r - 3 * s;
    const double *xb = x + (size_t)b * n + (size_t)k * ndp, *df = Df + (size_t)b * 18 + k * 6;
    auto wpd = [&](int j, int der) -> double {
      if (j == 0) return df[der];
      if (j == m) return df[3 + der];
      return xb[3 * (j - 1) + der];
    };
    const double p0 = wpd(s, 0), v0 = wpd(s, 1), a0 = wpd(s, 2);
    const double pT = wpd(s + 1, 0), vT = wpd(s + 1, 1), aT = wpd(s + 1, 2);
    const double Ts = T[(size_t)b * t_stride + s], T2 = Ts * Ts, iT = 1.0 / Ts, iT3 = iT * iT * iT;
    const double P = pT - p0 - v0 * Ts - 0.5 * a0 * T2;
    const double V = (vT - v0 - a0 * Ts) * Ts;
    const double A = (aT - a0) * T2;
    double *c = coeff + ((size_t)b * m + s) * 18 + 6 * k;
    c[0] = p0; c[1] = v0; c[2] = 0.5 * a0;
    c[3] = (10 * P - 4 * V + 0.5 * A) * iT3;
    c[4] = (-15 * P + 7 * V - A) * (iT3 * iT);
    c[5] = (6 * P - 3 * V + 0.5 * A) * (iT3 * iT * iT);
  }
}

}  // namespace

hipError_t gtop_launch_setup_paths(int B, int m, const double *wp, double mean_v, double init_time, double *T,
                                   double *Df, double *x0, hipStream_t stream) {
  if (B <= 0) return hipSuccess;
  hipLaunchKernelGGL(setup_paths_kernel, dim3(1024), dim3(256), 0, stream, B, m, wp, mean_v, init_time, T, Df, x0);
  return hipGetLastError();
}

hipError_t gtop_launch_coefficients(int B, int m, const double *x, const double *Df, const double *T, int t_stride,
                                    double *coeff, hipStream_t stream) {
  if (B <= 0) return hipSuccess;
  hipLaunchKernelGGL(coefficients_kernel, dim3(1024), dim3(256), 0, stream, B, m, x, Df, T, t_stride, coeff);
  return hipGetLastError();
}

hipError_t gtop_launch_eval_trajectories(int B, int m, const double *coeff, const double *T, int t_stride,
                                         double dt_sample, double *out, double *samples, int max_samples,
                                         hipStream_t stream) {
  if (B <= 0) return hipSuccess;
  hipLaunchKernelGGL(eval_trajectories_kernel, dim3(B), dim3(64), 0, stream, B, m, coeff, T, t_stride, dt_sample, out,
                     samples, max_samples);
  return hipGetLastError();
}
