// rendezvous_tsan.cpp — the rendezvous layer's host logic under ThreadSanitizer, on the CPU.
//
// csrc/gtop_rendezvous.cpp is pure host code over ONE entry of the C-ABI (gtop_eval_batch).  This program links it
// with a test double of that entry — cost = sum of squares of the row, gradient = 2 x, an adjustable sleep standing in
// for the launch — and drives every protocol path from real threads: shared launches with callers leaving one by one
// while another thread polls the statistics, a caller that never arrives (timeout), an abort and a destroy with
// callers blocked, and a launch slower than the timeout (which must NOT break the rendezvous).  Built by
// tests/test_rendezvous_host.py with `g++ -fsanitize=thread`; a data race or a wrong result fails the test.
// (The double is test scaffolding for THIS repository's own code; nothing of the reference is built or stood in for.)
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <thread>
#include <vector>

#include "gtop.h"

namespace {
std::atomic<int> g_eval_sleep_ms{0};
std::atomic<int> g_launches{0};
constexpr int kM = 3;                    // segments
constexpr unsigned kN = 9u * (kM - 1);   // free variables per trajectory
std::atomic<int> g_failures{0};
#define EXPECT(cond)                                                             \
  do {                                                                           \
    if (!(cond)) {                                                               \
      std::fprintf(stderr, "%s:%d: EXPECT(%s) failed\n", __FILE__, __LINE__, #cond); \
      ++g_failures;                                                              \
    }                                                                            \
  } while (0)
}  // namespace

// ---- the test double of the one C-ABI entry the layer calls ----
extern "C" int gtop_eval_batch(gtop_ctx *, int B, const double *x, double *cost, double *grad) {
  const int ms = g_eval_sleep_ms.load();
  if (ms > 0) std::this_thread::sleep_for(std::chrono::milliseconds(ms));
  for (int b = 0; b < B; ++b) {
    double c = 0.0;
    for (unsigned j = 0; j < kN; ++j) {
      const double v = x[(size_t)b * kN + j];
      c += v * v;
      grad[(size_t)b * kN + j] = 2.0 * v;
    }
    cost[b] = c;
  }
  g_launches.fetch_add(1);
  return GTOP_OK;
}

namespace {

gtop_ctx *fake_ctx() { return reinterpret_cast<gtop_ctx *>(&g_launches); }   // never dereferenced by the layer

// one serial "optimizer": `iters` callbacks on its own row, every result checked, then leave
void caller(gtop_rendezvous_slot *slot, int id, int iters, std::atomic<int> *huge_seen) {
  std::vector<double> x(kN), g(kN);
  for (int it = 0; it < iters; ++it) {
    double want = 0.0;
    for (unsigned j = 0; j < kN; ++j) {
      x[j] = 0.01 * (id + 1) + 0.001 * it + 0.1 * j;
      want += x[j] * x[j];
    }
    const double c = gtop_cost_nlopt_shared(kN, x.data(), g.data(), slot);
    if (c == HUGE_VAL) {
      if (huge_seen) huge_seen->fetch_add(1);
      break;
    }
    EXPECT(c == want);
    for (unsigned j = 0; j < kN; ++j) EXPECT(g[j] == 2.0 * x[j]);
  }
  EXPECT(gtop_rendezvous_leave(slot) == GTOP_OK);
}

void scenario_shared_launches_and_staggered_leaves(const int N) {   // (N above the core count: callers sleep on the futex)
  gtop_rendezvous *r = nullptr;
  EXPECT(gtop_rendezvous_create(&r, fake_ctx(), N, kM) == GTOP_OK);
  std::atomic<bool> stop{false};
  std::thread poller([&] {   // statistics read while the callers run
    while (!stop.load()) {
      int64_t launches = 0, callbacks = 0;
      double secs = 0.0;
      EXPECT(gtop_rendezvous_stats(r, &launches, &secs, &callbacks) == GTOP_OK);
      EXPECT(callbacks <= launches * N);
      std::this_thread::yield();
    }
  });
  std::vector<std::thread> th;
  int total = 0;
  for (int i = 0; i < N; ++i) {
    const int iters = 40 + 7 * i;   // everybody stops at a different time: the others must not wait for it
    total += iters;
    th.emplace_back(caller, gtop_rendezvous_get_slot(r, i), i, iters, nullptr);
  }
  for (auto &t : th) t.join();
  stop.store(true);
  poller.join();
  int64_t launches = 0, callbacks = 0;
  EXPECT(gtop_rendezvous_stats(r, &launches, nullptr, &callbacks) == GTOP_OK);
  EXPECT(callbacks == total);
  EXPECT(launches == 40 + 7 * (N - 1));   // as many launches as the longest caller's callbacks
  EXPECT(gtop_rendezvous_destroy(r) == GTOP_OK);
}

void scenario_a_caller_never_arrives() {
  const int N = 4;
  gtop_rendezvous *r = nullptr;
  EXPECT(gtop_rendezvous_create(&r, fake_ctx(), N, kM) == GTOP_OK);
  EXPECT(gtop_rendezvous_set_timeout(r, 0.05) == GTOP_OK);
  std::atomic<int> huge{0};
  std::vector<std::thread> th;
  for (int i = 0; i < N - 1; ++i) th.emplace_back(caller, gtop_rendezvous_get_slot(r, i), i, 5, &huge);   // slot N-1 stays away
  for (auto &t : th) t.join();
  EXPECT(huge.load() == N - 1);   // everybody gave up, nobody hangs; their leave() succeeded (checked in caller)
  std::vector<double> x(kN, 0.5);
  EXPECT(gtop_cost_nlopt_shared(kN, x.data(), nullptr, gtop_rendezvous_get_slot(r, N - 1)) == HUGE_VAL);   // broken for good
  EXPECT(gtop_rendezvous_destroy(r) == GTOP_OK);
}

void scenario_abort_and_destroy_with_callers_blocked(bool destroy_instead) {
  const int N = 5;
  gtop_rendezvous *r = nullptr;
  EXPECT(gtop_rendezvous_create(&r, fake_ctx(), N, kM) == GTOP_OK);
  std::atomic<int> huge{0};
  std::vector<std::thread> th;
  for (int i = 0; i < N - 1; ++i) th.emplace_back(caller, gtop_rendezvous_get_slot(r, i), i, 3, &huge);
  std::this_thread::sleep_for(std::chrono::milliseconds(30));   // they are asleep on the generation word by now
  if (destroy_instead) {
    EXPECT(gtop_rendezvous_destroy(r) == GTOP_OK);   // wakes them, waits until the last one is out, frees
    for (auto &t : th) t.join();
  } else {
    EXPECT(gtop_rendezvous_abort(r) == GTOP_OK);
    for (auto &t : th) t.join();
    EXPECT(gtop_rendezvous_destroy(r) == GTOP_OK);
  }
  EXPECT(huge.load() == N - 1);
}

void scenario_a_slow_launch_is_not_a_missing_caller() {
  const int N = 4;
  gtop_rendezvous *r = nullptr;
  EXPECT(gtop_rendezvous_create(&r, fake_ctx(), N, kM) == GTOP_OK);
  EXPECT(gtop_rendezvous_set_timeout(r, 0.02) == GTOP_OK);
  g_eval_sleep_ms.store(120);   // six timeouts long
  std::atomic<int> huge{0};
  std::vector<std::thread> th;
  for (int i = 0; i < N; ++i) th.emplace_back(caller, gtop_rendezvous_get_slot(r, i), i, 2, &huge);
  for (auto &t : th) t.join();
  g_eval_sleep_ms.store(0);
  EXPECT(huge.load() == 0);
  EXPECT(gtop_rendezvous_destroy(r) == GTOP_OK);
}

void scenario_misuse() {
  gtop_rendezvous *r = nullptr;
  EXPECT(gtop_rendezvous_create(&r, fake_ctx(), 2, kM) == GTOP_OK);
  std::vector<double> x(kN, 1.0);
  EXPECT(gtop_cost_nlopt_shared(kN + 1, x.data(), nullptr, gtop_rendezvous_get_slot(r, 0)) == HUGE_VAL);   // wrong n
  EXPECT(gtop_rendezvous_get_slot(r, 2) == nullptr);
  // a leave from another thread while the slot's own caller is blocked inside the callback is refused
  std::atomic<bool> entering{false};
  std::thread blocked([&] {
    entering.store(true);
    (void)gtop_cost_nlopt_shared(kN, x.data(), nullptr, gtop_rendezvous_get_slot(r, 0));
  });
  while (!entering.load()) std::this_thread::yield();
  std::this_thread::sleep_for(std::chrono::milliseconds(100));   // (the call is a few instructions behind the flag; generous for a loaded machine)
  EXPECT(gtop_rendezvous_leave(gtop_rendezvous_get_slot(r, 0)) == GTOP_ERR_STATE);
  EXPECT(gtop_rendezvous_leave(gtop_rendezvous_get_slot(r, 1)) == GTOP_OK);   // the other one leaves: slot 0 is evaluated
  blocked.join();
  EXPECT(gtop_rendezvous_leave(gtop_rendezvous_get_slot(r, 0)) == GTOP_OK);
  EXPECT(gtop_rendezvous_destroy(r) == GTOP_OK);
}

}  // namespace

int main() {
  scenario_shared_launches_and_staggered_leaves(6);
  scenario_shared_launches_and_staggered_leaves(24);
  scenario_a_caller_never_arrives();
  scenario_abort_and_destroy_with_callers_blocked(false);
  scenario_abort_and_destroy_with_callers_blocked(true);
  scenario_a_slow_launch_is_not_a_missing_caller();
  scenario_misuse();
  if (g_failures.load()) {
    std::printf("rendezvous_tsan: %d failure(s)\n", g_failures.load());
    return 1;
  }
  std::printf("rendezvous_tsan: ok (%d launches of the test double)\n", g_launches.load());
  return 0;
}
