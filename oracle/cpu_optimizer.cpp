// cpu_optimizer.cpp — the reference's CPU path AROUND the callback, for bench.py's CPU leg beside the batched device
// optimizer: one serial LD_MMA run per trajectory calling the cost/gradient callback once per evaluation
// (src/grad_traj_optimizer.cpp:137-195 of EpicOne1/grad_traj_optimization: nlopt::opt(LD_MMA), bounds, x0 from Dp,
// optimize()).  TEST / MEASUREMENT INFRASTRUCTURE ONLY, like the rest of oracle/: the callback is the C restatement
// (gtop_oracle.c, parity unpinned), and NLopt — absent from this image — is stood in for by the CCSA-MMA restatement
// the product's host shim uses (grad_traj_optimization_amd/csrc/mma.hpp, header only; nothing of the product's
// device path is linked here).  L and R are built once per trajectory before the clock starts, as setPath does
// (:67-110) before optimizeTrajectory is called.
#include <omp.h>

#include <chrono>
#include <vector>

#include "../grad_traj_optimization_amd/csrc/mma.hpp"
#include "gtop_oracle.h"

namespace {

struct Problem {
  int m;
  const double *L, *R, *Df, *T;
  const oracle_params *prm;
  const oracle_sdf *sdf;
};

double callback(unsigned, const double *x, double *grad, void *data) {
  const Problem *p = static_cast<const Problem *>(data);
  return oracle_cost_grad(p->m, p->L, p->R, p->Df, p->T, p->prm, p->sdf, x, grad);
}

}  // namespace

// x: B*n in (start points) / out (best points); returns the seconds the B optimisations took on `nthreads` threads
// (setup excluded), or -1 on failure.
extern "C" double oracle_optimize_batch(int B, int m, const double *T, int t_stride, const double *Df,
                                        const oracle_params *prm, const oracle_sdf *sdf, double *x, const double *lb,
                                        const double *ub, int max_evals, double *min_cost, int *nevals, int nthreads) {
  if (B < 1 || m < 2 || max_evals < 1) return -1.0;
  const int n6 = 6 * m, nd = 3 * m + 3, n = 9 * (m - 1);
  const size_t szL = (size_t)n6 * nd, szR = (size_t)nd * nd;
  std::vector<double> L(szL * B), R(szR * B);
  int bad = 0;
#pragma omp parallel for num_threads(nthreads) reduction(+ : bad)
  for (int b = 0; b < B; ++b) {
    std::vector<double> A((size_t)n6 * n6), Q((size_t)n6 * n6), Ct(szL);
    bad += oracle_generator(m, T + (size_t)b * t_stride, A.data(), Q.data(), Ct.data(), &L[szL * b], &R[szR * b]) != 0;
  }
  if (bad) return -1.0;
  const auto t0 = std::chrono::steady_clock::now();
#pragma omp parallel for num_threads(nthreads) schedule(dynamic, 1)
  for (int b = 0; b < B; ++b) {
    Problem p{m, &L[szL * b], &R[szR * b], Df + (size_t)b * 18, T + (size_t)b * t_stride, prm, sdf};
    gtop_amd::MmaOptions opt;
    opt.maxeval = max_evals;
    const gtop_amd::MmaResult r =
        gtop_amd::mma_minimize((unsigned)n, callback, &p, lb + (size_t)b * n, ub + (size_t)b * n, x + (size_t)b * n, opt);
    if (min_cost) min_cost[b] = r.minf;
    if (nevals) nevals[b] = r.nevals;
  }
  return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
}

// ---- traces of the serial optimizer (tests/test_mma_twin.py, tests/golden/make_mma_golden.py) ----------------------
// The product's host optimizer (csrc/mma.hpp) run once with every evaluation recorded — point, value — so that an
// INDEPENDENT restatement of the published algorithm (oracle/mma_twin.py, numpy, written without this header) can be
// held against it evaluation by evaluation.  Two objectives: the oracle's cost/gradient callback, and a separable
// convex quadratic f = sum a_j (x_j - c_j)^2 / 2 whose box-constrained minimiser clip(c, lb, ub) is known in closed form.
namespace {

struct Traced {
  gtop_amd::mma_objective f;
  void *data;
  unsigned n;
  int cap, count;
  double *xs, *fs;   // [cap][n], [cap]
};

double traced_callback(unsigned n, const double *x, double *grad, void *data) {
  Traced *t = static_cast<Traced *>(data);
  const double v = t->f(n, x, grad, t->data);
  if (t->count < t->cap) {
    for (unsigned j = 0; j < n; ++j) t->xs[(size_t)t->count * n + j] = x[j];
    t->fs[t->count] = v;
  }
  t->count++;
  return v;
}

struct Quadratic {
  const double *a, *c;
};

double quadratic_callback(unsigned n, const double *x, double *grad, void *data) {
  const Quadratic *q = static_cast<const Quadratic *>(data);
  double v = 0.0;
  for (unsigned j = 0; j < n; ++j) {
    const double d = x[j] - q->c[j];
    v += 0.5 * q->a[j] * d * d;
    if (grad) grad[j] = q->a[j] * d;
  }
  return v;
}

int run_traced(Traced &t, const double *lb, const double *ub, double *x, int max_evals, double ftol_rel, double xtol_rel,
               double *minf, int *nevals) {
  gtop_amd::MmaOptions opt;
  opt.maxeval = max_evals;
  opt.ftol_rel = ftol_rel;
  opt.xtol_rel = xtol_rel;
  const gtop_amd::MmaResult r = gtop_amd::mma_minimize(t.n, traced_callback, &t, lb, ub, x, opt);
  if (minf) *minf = r.minf;
  if (nevals) *nevals = r.nevals;
  return r.code;
}

}  // namespace

// one trajectory; xs: cap x n, fs: cap (the evaluations in order; cap >= max_evals); returns mma.hpp's result code
extern "C" int oracle_mma_trace(int m, const double *T, const double *Df, const oracle_params *prm, const oracle_sdf *sdf,
                                double *x, const double *lb, const double *ub, int max_evals, double ftol_rel,
                                double xtol_rel, double *minf, int *nevals, double *xs, double *fs, int cap) {
  if (m < 2 || max_evals < 1) return -2;
  const int n6 = 6 * m, nd = 3 * m + 3, n = 9 * (m - 1);
  std::vector<double> L((size_t)n6 * nd), R((size_t)nd * nd), A((size_t)n6 * n6), Q((size_t)n6 * n6), Ct((size_t)n6 * nd);
  if (oracle_generator(m, T, A.data(), Q.data(), Ct.data(), L.data(), R.data()) != 0) return -2;
  Problem p{m, L.data(), R.data(), Df, T, prm, sdf};
  Traced t{callback, &p, (unsigned)n, cap, 0, xs, fs};
  return run_traced(t, lb, ub, x, max_evals, ftol_rel, xtol_rel, minf, nevals);
}

extern "C" int oracle_mma_trace_quadratic(int n, const double *a, const double *c, double *x, const double *lb,
                                          const double *ub, int max_evals, double ftol_rel, double xtol_rel, double *minf,
                                          int *nevals, double *xs, double *fs, int cap) {
  if (n < 1 || max_evals < 1) return -2;
  Quadratic q{a, c};
  Traced t{quadratic_callback, &q, (unsigned)n, cap, 0, xs, fs};
  return run_traced(t, lb, ub, x, max_evals, ftol_rel, xtol_rel, minf, nevals);
}
