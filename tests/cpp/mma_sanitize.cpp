// mma_sanitize.cpp — csrc/mma.hpp (the serial CCSA-MMA that stands in for NLopt's LD_MMA in the C++ shim and in the
// bench's CPU leg; the reference calls NLopt at src/grad_traj_optimizer.cpp:137-195) under AddressSanitizer and
// UndefinedBehaviorSanitizer on the CPU: separable quadratics over boxes with every kind of bound the shim can hand it
// (finite, one-sided, infinite, zero width), n = 1, a start outside its box, every stop rule.  Built and run by
// tests/test_mma_twin.py; known minimiser clip(c, lb, ub) checked on the way.
#include <cmath>
#include <cstdio>
#include <vector>

#include "mma.hpp"

namespace {
struct Quad { std::vector<double> a, c; int calls = 0; };
double quad(unsigned n, const double *x, double *g, void *data) {
  Quad *q = static_cast<Quad *>(data);
  q->calls++;
  double f = 0.0;
  for (unsigned j = 0; j < n; ++j) {
    const double d = x[j] - q->c[j];
    f += 0.5 * q->a[j] * d * d;
    if (g) g[j] = q->a[j] * d;
  }
  return f + 1.0;
}
unsigned long long lcg = 88172645463325252ull;
double uni() {
  lcg = lcg * 6364136223846793005ull + 1442695040888963407ull;
  return (double)(lcg >> 11) / 9007199254740992.0;
}
}  // namespace

int main() {
  using namespace gtop_amd;
  int bad = 0, runs = 0;
  for (int draw = 0; draw < 200; ++draw) {
    const unsigned n = draw % 7 == 0 ? 1u : 1u + (unsigned)(uni() * 40);
    Quad q;
    std::vector<double> lb(n), ub(n), x(n);
    for (unsigned j = 0; j < n; ++j) {
      q.a.push_back(0.1 + 10 * uni());
      q.c.push_back(-3 + 6 * uni());
      const int kind = (int)(uni() * 5);
      lb[j] = kind == 1 || kind == 3 ? -HUGE_VAL : -2 + uni();
      ub[j] = kind == 2 || kind == 3 ? HUGE_VAL : 1 + uni();
      if (kind == 4) ub[j] = lb[j];                       // zero-width box
      x[j] = -4 + 8 * uni();                              // possibly outside: the shim clamps, as documented
      if (std::isfinite(lb[j]) && x[j] < lb[j]) x[j] = draw % 3 ? lb[j] : x[j];
    }
    MmaOptions opt;
    opt.maxeval = draw % 4 == 0 ? 7 : 400;
    opt.ftol_rel = draw % 3 == 0 ? 1e-10 : 0.0;
    opt.xtol_rel = draw % 3 == 1 ? 1e-9 : 0.0;
    opt.maxtime = draw % 5 == 0 ? 5.0 : 0.0;
    const MmaResult r = mma_minimize(n, quad, &q, lb.data(), ub.data(), x.data(), opt);
    ++runs;
    if (r.nevals != q.calls || r.nevals > opt.maxeval || !(r.minf >= 1.0)) ++bad;
    for (unsigned j = 0; j < n; ++j) {
      if (!(x[j] >= lb[j] && x[j] <= ub[j])) ++bad;
      if (opt.maxeval == 400 && r.code != MMA_MAXEVAL_REACHED) {      // converged runs: the box-clipped centre
        const double want = std::fmin(std::fmax(q.c[j], lb[j]), ub[j]);
        if (std::fabs(x[j] - want) > 1e-3 * (1.0 + std::fabs(want))) ++bad;
      }
    }
  }
  std::printf("mma_sanitize: %s (%d runs, %d bad)\n", bad ? "FAILED" : "ok", runs, bad);
  return bad ? 1 : 0;
}
