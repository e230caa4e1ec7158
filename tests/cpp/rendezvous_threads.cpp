// rendezvous_threads.cpp — N host threads, each a serial optimizer on its own trajectory (csrc/mma.hpp standing
// in for NLopt's LD_MMA, which the reference uses one instance of per problem: src/grad_traj_optimizer.cpp:137-195),
// sharing launches through gtop_cost_nlopt_shared; compared bit for bit with the same optimizers run one after
// the other through gtop_cost_nlopt.  Prints one JSON object (tests/test_rendezvous.py, tools/host_api_rate.py).
// usage: gtop_rendezvous_demo [N=64] [m=6] [max_evals=30] [spl=0]
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

#include "gtop.h"
#include "mma.hpp"

using namespace gtop_amd;

namespace {
struct Lcg {   // small deterministic generator: the scene must be the same on every box
  unsigned long long s;
  double next() {
    s = s * 6364136223846793005ull + 1442695040888963407ull;
    return (double)(s >> 11) / 9007199254740992.0;
  }
};
#define CHK(call)                                                                         \
  do {                                                                                    \
    int rc_ = (call);                                                                     \
    if (rc_ != GTOP_OK) {                                                                 \
      std::fprintf(stderr, "%s failed: %d %s\n", #call, rc_, gtop_last_error(ctx));       \
      return 2;                                                                           \
    }                                                                                     \
  } while (0)
}  // namespace

int main(int argc, char **argv) {
  const int N = argc > 1 ? std::atoi(argv[1]) : 64, m = argc > 2 ? std::atoi(argv[2]) : 6;
  const int max_evals = argc > 3 ? std::atoi(argv[3]) : 30, spl = argc > 4 ? std::atoi(argv[4]) : 0;
  const unsigned n = 9u * (unsigned)(m - 1);
  gtop_ctx *ctx = nullptr;
  if (gtop_create(&ctx, 0) != GTOP_OK) {
    std::fprintf(stderr, "gtop_create: %s\n", gtop_last_error(nullptr));
    return 2;
  }
  gtop_params prm = {1.0, 5.0, 10.0, 0.5, 0.8, 0.0, 1.5, 2.5, 0.0, 1.5, 3.5, 2, 0};   // launch/opti_node.launch:3-28
  CHK(gtop_set_params(ctx, &prm));
  CHK(gtop_set_launch_geometry(ctx, 0, spl));
  // scene: 16 x 16 x 6 m map at 0.2 m, random pillars
  const double map_size[3] = {16, 16, 6}, origin[3] = {-8, -8, 0};
  CHK(gtop_init_sdf_map(ctx, map_size, origin, 0.2));
  Lcg rng{12345};
  std::vector<double> pts;
  for (int p = 0; p < 40; ++p) {
    const double cx = -7 + 14 * rng.next(), cy = -7 + 14 * rng.next(), h = 2 + 4 * rng.next();
    for (double x = cx; x < cx + 0.6; x += 0.2)
      for (double y = cy; y < cy + 0.6; y += 0.2)
        for (double z = 0.1; z < h; z += 0.2) { pts.push_back(x); pts.push_back(y); pts.push_back(z); }
  }
  CHK(gtop_update_sdf_map(ctx, pts.data(), (int)(pts.size() / 3)));
  // N random-walk waypoint lists, 1 m inside the map
  std::vector<double> wp((size_t)N * (m + 1) * 3);
  for (int i = 0; i < N; ++i) {
    double p[3] = {-6 + 12 * rng.next(), -6 + 12 * rng.next(), 1 + 4 * rng.next()};
    for (int k = 0; k <= m; ++k) {
      for (int a = 0; a < 3; ++a) wp[((size_t)i * (m + 1) + k) * 3 + a] = p[a];
      const double lo[3] = {-7, -7, 1}, hi[3] = {7, 7, 5};
      for (int a = 0; a < 3; ++a) {
        p[a] += (rng.next() < 0.5 ? -1 : 1) * (0.6 + 0.6 * rng.next());
        if (p[a] < lo[a]) p[a] = 2 * lo[a] - p[a];
        if (p[a] > hi[a]) p[a] = 2 * hi[a] - p[a];
      }
    }
  }
  std::vector<double> x0((size_t)N * n), T((size_t)N * m), Df((size_t)N * 18), lb((size_t)N * n), ub((size_t)N * n);
  CHK(gtop_set_paths(ctx, N, m, wp.data(), 1.8, 0.3, x0.data()));   // setPath for all N (:67-110)
  CHK(gtop_get_problem(ctx, T.data(), Df.data()));
  CHK(gtop_default_bounds(N, m, wp.data(), 3.0, 8.0, 10.0, lb.data(), ub.data()));   // :151-179
  auto evals_of = [&](int i) { return max_evals - (i % 5); };   // staggered: callers leave at different times

  // ---- one after the other: the reference's usage, one problem per optimizer, B = 1 per callback ----
  std::vector<double> xs((size_t)N * n), fs(N);
  std::vector<int> ns(N);
  long serial_calls = 0;
  const auto ta = std::chrono::steady_clock::now();
  for (int i = 0; i < N; ++i) {
    CHK(gtop_set_problem(ctx, 1, m, &T[(size_t)i * m], m, &Df[(size_t)i * 18]));
    std::memcpy(&xs[(size_t)i * n], &x0[(size_t)i * n], n * sizeof(double));
    MmaOptions opt;
    opt.maxeval = evals_of(i);
    const MmaResult r = mma_minimize(n, gtop_cost_nlopt, ctx, &lb[(size_t)i * n], &ub[(size_t)i * n],
                                     &xs[(size_t)i * n], opt);
    fs[i] = r.minf;
    ns[i] = r.nevals;
    serial_calls += r.nevals;
  }
  const double serial_s = std::chrono::duration<double>(std::chrono::steady_clock::now() - ta).count();

  // ---- N threads meeting in shared launches ----
  CHK(gtop_set_problem(ctx, N, m, T.data(), m, Df.data()));
  gtop_rendezvous *rdv = nullptr;
  CHK(gtop_rendezvous_create(&rdv, ctx, N, m));
  std::vector<double> xr((size_t)N * n), fr(N);
  std::vector<int> nr(N);
  const auto tb = std::chrono::steady_clock::now();
  {
    std::vector<std::thread> th;
    for (int i = 0; i < N; ++i)
      th.emplace_back([&, i] {
        gtop_rendezvous_slot *slot = gtop_rendezvous_get_slot(rdv, i);
        std::memcpy(&xr[(size_t)i * n], &x0[(size_t)i * n], n * sizeof(double));
        MmaOptions opt;
        opt.maxeval = evals_of(i);
        const MmaResult r = mma_minimize(n, gtop_cost_nlopt_shared, slot, &lb[(size_t)i * n], &ub[(size_t)i * n],
                                         &xr[(size_t)i * n], opt);
        fr[i] = r.minf;
        nr[i] = r.nevals;
        gtop_rendezvous_leave(slot);
      });
    for (auto &t : th) t.join();
  }
  const double shared_s = std::chrono::duration<double>(std::chrono::steady_clock::now() - tb).count();
  int64_t launches = 0, callbacks = 0;
  double launch_s = 0;
  gtop_rendezvous_stats(rdv, &launches, &launch_s, &callbacks);
  gtop_rendezvous_destroy(rdv);

  bool identical = true;
  double max_dx = 0, improved = 0;
  for (int i = 0; i < N; ++i) {
    identical = identical && ns[i] == nr[i] && std::memcmp(&fs[i], &fr[i], sizeof(double)) == 0 &&
                std::memcmp(&xs[(size_t)i * n], &xr[(size_t)i * n], n * sizeof(double)) == 0;
    for (unsigned j = 0; j < n; ++j) max_dx = std::fmax(max_dx, std::fabs(xs[(size_t)i * n + j] - xr[(size_t)i * n + j]));
  }
  // did the optimizers optimise?  cost at x0 vs the minimum found
  std::vector<double> c0(N), g0((size_t)N * n);
  CHK(gtop_eval_batch(ctx, N, x0.data(), c0.data(), g0.data()));
  for (int i = 0; i < N; ++i) improved += fr[i] < c0[i] ? 1 : 0;
  std::printf("{\"threads\": %d, \"m\": %d, \"max_evals\": %d, \"spl\": %d, \"identical\": %s, \"max_abs_dx\": %.3g,\n"
              " \"callbacks\": %lld, \"launches\": %lld, \"serial_callbacks\": %ld, \"fraction_improved\": %.3f,\n"
              " \"serial_us_per_callback\": %.3f, \"shared_us_per_callback\": %.3f, \"shared_us_per_launch\": %.3f,\n"
              " \"us_inside_launches_per_launch\": %.3f}\n",
              N, m, max_evals, spl, identical ? "true" : "false", max_dx, (long long)callbacks, (long long)launches,
              serial_calls, improved / N, 1e6 * serial_s / serial_calls, 1e6 * shared_s / (double)callbacks,
              1e6 * shared_s / (double)launches, 1e6 * launch_s / (double)launches);
  gtop_destroy(ctx);
  return identical ? 0 : 1;
}
