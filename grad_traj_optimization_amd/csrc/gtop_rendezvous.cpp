// gtop_rendezvous.cpp — N serial callers, one launch.
//
// The reference runs one NLopt instance per problem: optimizer.optimize() calls
// the cost/gradient callback serially, one trajectory per call
// (src/grad_traj_optimizer.cpp:137-195, :554-562).  A GPU evaluation of ONE
// trajectory is launch-bound (~20 us, most of it the launch and the
// synchronisation), so serial callers only profit if their callbacks share a
// launch.  This layer lets N host threads — each running its own serial
// optimizer (NLopt's LD_MMA, or csrc/mma.hpp) on its own trajectory — meet:
// every thread calls gtop_cost_nlopt_shared (exactly NLopt's nlopt_func shape)
// with its slot as func_data; the call blocks until all slots still in the
// game have arrived, the last arriver evaluates the whole batch with ONE
// gtop_eval_batch on the shared context, and every caller returns with its own
// cost and gradient.  A caller whose optimizer has stopped leaves
// (gtop_rendezvous_leave); the others no longer wait for it.
//
// Results are those of gtop_eval_batch on the same rows: the rows of a batch are
// independent, so with the launch geometry the context uses for N trajectories
// each caller sees, bit for bit, what a batch evaluation gives it.
//
// Pure host logic over the public C-ABI (include/gtop.h).  A mutex guards the
// arrival bookkeeping only (tens of nanoseconds per caller); callers sleep on a
// generation word (futex; a short poll first when every caller has a core), and
// copy their results out without any lock: a generation's buffers stay put
// until every caller has arrived again.
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstring>
#include <mutex>
#include <new>
#include <thread>
#include <vector>

#include <cstdio>
#include <cstdlib>

#include <linux/futex.h>
#include <sched.h>
#include <sys/syscall.h>
#include <unistd.h>

#include "gtop.h"
#include "gtop_guard.h"

struct gtop_rendezvous_slot {
  gtop_rendezvous *owner;
  int index;
  bool active;       // still taking part
  bool waiting;      // has arrived for the current generation
  bool in_call;      // its caller is inside gtop_cost_nlopt_shared (arrived, or awake again and copying its results out)
  int64_t calls;     // callbacks served (iter_num of this caller, grad_traj_optimizer.cpp:284)
  double best;       // running minimum of its costs (:439-447)
};

struct gtop_rendezvous {
  gtop_ctx *ctx = nullptr;
  int n_slots = 0;
  unsigned n = 0;                 // free variables per trajectory
  unsigned spin = 0;              // polls of the generation word before sleeping on it
  std::vector<gtop_rendezvous_slot> slots;
  std::vector<double> x, cost, grad;   // [n_slots][n], [n_slots], [n_slots][n]
  std::mutex mu;                  // guards the bookkeeping below only: never held across a launch or a sleep
  int active = 0, arrived = 0;
  bool evaluating = false;        // a leader has been elected and is inside gtop_eval_batch with x / cost / grad
  std::atomic<int> inside{0};     // threads inside gtop_cost_nlopt_shared / gtop_rendezvous_leave right now
  std::atomic<uint32_t> generation{0};   // waiters sleep on this word (futex)
  int last_status = GTOP_OK;      // of the generation just evaluated
  std::atomic<bool> broken{false};   // gtop_rendezvous_abort, or a waiter's timeout: every call returns HUGE_VAL from then on
  std::atomic<double> timeout_s{0.0};   // longest a caller waits for the others (0 = for ever); may be set while callers wait
  double test_delay_s = 0.0;      // diagnostic (GTOP_RENDEZVOUS_TEST_DELAY_MS at create): the leader sleeps this long
                                  // before its launch, standing in for a slow first launch in the timeout tests
  int64_t launches = 0;
  double launch_seconds = 0.0;
};

namespace {

// (a rendezvous has no error text of its own; the context's gtop_last_error belongs to the evaluation)
inline void note_exception(void *, const char *) noexcept {}
#define GTOP_CATCH_STATUS(x) GTOP_CATCH_WITH(note_exception, x, GTOP_ERR_INTERNAL)
#define GTOP_CATCH_HUGE(x) GTOP_CATCH_WITH(note_exception, x, HUGE_VAL)

// cores this process may really use: the affinity mask, capped by the cgroup CPU quota (a container sees the
// whole host in hardware_concurrency())
int usable_cores() {
  cpu_set_t set;
  int n = 1;
  if (sched_getaffinity(0, sizeof(set), &set) == 0) n = CPU_COUNT(&set);
  if (FILE *f = std::fopen("/sys/fs/cgroup/cpu.max", "r")) {
    char quota[32];
    long period = 0;
    if (std::fscanf(f, "%31s %ld", quota, &period) == 2 && std::strcmp(quota, "max") != 0 && period > 0) {
      const long q = std::atol(quota) / period;
      if (q >= 1 && q < n) n = (int)q;
    }
    std::fclose(f);
  }
  return n < 1 ? 1 : n;
}

void futex_wait(std::atomic<uint32_t> *w, uint32_t seen, double seconds = 0.0) {
  struct timespec ts;
  if (seconds > 0.0) {
    ts.tv_sec = (time_t)seconds;
    ts.tv_nsec = (long)((seconds - (double)ts.tv_sec) * 1e9);
  }
  syscall(SYS_futex, reinterpret_cast<uint32_t *>(w), FUTEX_WAIT_PRIVATE, seen, seconds > 0.0 ? &ts : nullptr, nullptr, 0);
}
void futex_wake_all(std::atomic<uint32_t> *w) {
  syscall(SYS_futex, reinterpret_cast<uint32_t *>(w), FUTEX_WAKE_PRIVATE, 0x7fffffff, nullptr, nullptr, 0);
}

// A caller gave up (timeout) or the owner aborted: nobody may block any more.  The arrival bookkeeping is cleared
// under the lock, so that the callers that return HUGE_VAL from here are no longer counted as waiting (a later
// gtop_rendezvous_leave on their slots succeeds).  A leader that is evaluating right now finishes its launch on its
// own; gtop_rendezvous_destroy waits for it.
void break_rendezvous(gtop_rendezvous *r) {
  {
    std::lock_guard<std::mutex> lk(r->mu);
    r->broken.store(true, std::memory_order_release);
    for (auto &s : r->slots) s.waiting = false;
    r->arrived = 0;
  }
  r->generation.fetch_add(1, std::memory_order_release);
  futex_wake_all(&r->generation);
}

struct CallGuard {   // the slot's caller is inside the callback until this goes out of scope; counts a served call
  gtop_rendezvous *r;
  gtop_rendezvous_slot *s;
  bool served = false;
  double cost = HUGE_VAL;
  ~CallGuard() {
    std::lock_guard<std::mutex> lk(r->mu);   // (gtop_rendezvous_stats / _leave look at these from other threads)
    s->in_call = false;
    if (served) {
      s->calls++;
      if (cost < s->best) s->best = cost;
    }
  }
};

struct InsideGuard {   // counts the threads inside an entry point (gtop_rendezvous_destroy waits for zero)
  gtop_rendezvous *r;
  explicit InsideGuard(gtop_rendezvous *r_) : r(r_) { r->inside.fetch_add(1, std::memory_order_acq_rel); }
  ~InsideGuard() { r->inside.fetch_sub(1, std::memory_order_acq_rel); }
};

// Every active slot has arrived and the caller was elected under the lock (which set r->evaluating): all the other
// callers are blocked on the generation word, so nothing else touches the buffers.  Evaluate, publish, wake everybody.
void run_generation(gtop_rendezvous *r) {
  if (r->test_delay_s > 0.0) std::this_thread::sleep_for(std::chrono::duration<double>(r->test_delay_s));
  const auto t0 = std::chrono::steady_clock::now();
  const int st = gtop_eval_batch(r->ctx, r->n_slots, r->x.data(), r->cost.data(), r->grad.data());
  {
    std::lock_guard<std::mutex> lk(r->mu);
    r->last_status = st;
    r->launch_seconds += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    r->launches++;
    for (auto &s : r->slots) s.waiting = false;
    r->arrived = 0;
    r->evaluating = false;
  }
  r->generation.fetch_add(1, std::memory_order_release);
  futex_wake_all(&r->generation);
}

}  // namespace

extern "C" {

int gtop_rendezvous_create(gtop_rendezvous **out, gtop_ctx *ctx, int n_slots, int m) try {
  if (!out) return GTOP_ERR_INVALID;
  *out = nullptr;
  if (!ctx || n_slots < 1 || m < 2) return GTOP_ERR_INVALID;
  gtop_rendezvous *r = new (std::nothrow) gtop_rendezvous();
  if (!r) return GTOP_ERR_INVALID;
  r->ctx = ctx;
  r->n_slots = n_slots;
  r->n = 9u * (unsigned)(m - 1);
  // with a core per caller a short poll beats a sleep/wake pair (~50 us); with more callers than cores polling
  // only steals the cores the late arrivers need
  r->spin = n_slots <= usable_cores() ? 20000u : 0u;
  if (const char *d = std::getenv("GTOP_RENDEZVOUS_TEST_DELAY_MS")) r->test_delay_s = std::atof(d) * 1e-3;
  r->slots.resize(n_slots);
  for (int i = 0; i < n_slots; ++i) r->slots[i] = gtop_rendezvous_slot{r, i, true, false, false, 0, HUGE_VAL};
  r->active = n_slots;
  r->x.assign((size_t)n_slots * r->n, 0.0);
  r->cost.assign(n_slots, 0.0);
  r->grad.assign((size_t)n_slots * r->n, 0.0);
  *out = r;
  return GTOP_OK;
} GTOP_CATCH_STATUS(nullptr)

int gtop_rendezvous_destroy(gtop_rendezvous *r) try {
  if (!r) return GTOP_ERR_INVALID;
  // Nobody may be left inside: callers asleep on the generation word are woken (the rendezvous is broken for good),
  // and a leader that is inside gtop_eval_batch with this object's buffers is waited for — freeing them under its
  // launch would be a use after free.
  break_rendezvous(r);
  for (;;) {
    bool busy;
    {
      std::lock_guard<std::mutex> lk(r->mu);
      busy = r->evaluating;
    }
    if (!busy && r->inside.load(std::memory_order_acquire) == 0) break;
    std::this_thread::sleep_for(std::chrono::microseconds(50));
  }
  delete r;
  return GTOP_OK;
} GTOP_CATCH_STATUS(nullptr)

gtop_rendezvous_slot *gtop_rendezvous_get_slot(gtop_rendezvous *r, int i) {
  if (!r || i < 0 || i >= r->n_slots) return nullptr;
  return &r->slots[i];
}

double gtop_cost_nlopt_shared(unsigned n, const double *x, double *grad, void *func_data) try {
  gtop_rendezvous_slot *s = static_cast<gtop_rendezvous_slot *>(func_data);
  if (!s || !s->owner || !x) return HUGE_VAL;
  gtop_rendezvous *r = s->owner;
  InsideGuard inside(r);
  if (n != r->n || r->broken.load(std::memory_order_acquire)) return HUGE_VAL;
  bool leader;
  uint32_t gen;
  {
    std::lock_guard<std::mutex> lk(r->mu);
    if (r->broken.load(std::memory_order_relaxed)) return HUGE_VAL;   // (broken between the test above and the lock)
    if (!s->active || s->waiting || s->in_call) return HUGE_VAL;   // left already / re-entered from a second thread
    // (the row is this caller's own; the previous generation's launch has completed, or it could not be here)
    std::memcpy(&r->x[(size_t)s->index * n], x, n * sizeof(double));
    s->waiting = true;
    s->in_call = true;
    r->arrived++;
    gen = r->generation.load(std::memory_order_relaxed);
    leader = r->arrived == r->active;
    if (leader) r->evaluating = true;
  }
  CallGuard call{r, s};
  if (leader) {
    run_generation(r);   // the last arriver evaluates for everybody
  } else {
    for (unsigned i = 0; i < r->spin && r->generation.load(std::memory_order_acquire) == gen; ++i) __builtin_ia32_pause();
    auto t0 = std::chrono::steady_clock::now();
    while (r->generation.load(std::memory_order_acquire) == gen) {
      double left = 0.0;
      const double timeout_s = r->timeout_s.load(std::memory_order_relaxed);
      if (timeout_s > 0.0) {
        left = timeout_s - std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        if (left <= 0.0) {
          // The timeout is for a caller that never ARRIVES (it exited without gtop_rendezvous_leave).  Once everybody
          // has arrived the wait is for the leader's launch — a module load on the first call, a large batch — which
          // is not the callers' fault and ends by itself: the clock starts again.
          bool launch_in_flight;
          {
            std::lock_guard<std::mutex> lk(r->mu);
            launch_in_flight = r->evaluating;
          }
          if (launch_in_flight) {
            t0 = std::chrono::steady_clock::now();
            continue;
          }
          if (r->generation.load(std::memory_order_acquire) != gen) break;   // (published while we looked)
          break_rendezvous(r);   // give up, for everybody
          break;
        }
      }
      futex_wait(&r->generation, gen, left);
    }
  }
  if (r->broken.load(std::memory_order_acquire)) return HUGE_VAL;
  // results of this generation stay put until every active caller — this one included — has arrived again
  if (r->last_status != GTOP_OK) return HUGE_VAL;
  const double c = r->cost[s->index];
  if (grad) std::memcpy(grad, &r->grad[(size_t)s->index * n], n * sizeof(double));
  call.served = true;
  call.cost = c;
  return c;
} GTOP_CATCH_HUGE(nullptr)

int gtop_rendezvous_leave(gtop_rendezvous_slot *s) try {
  if (!s || !s->owner) return GTOP_ERR_INVALID;
  gtop_rendezvous *r = s->owner;
  InsideGuard inside(r);
  bool leader;
  {
    std::lock_guard<std::mutex> lk(r->mu);
    if (!s->active) return GTOP_OK;
    // the slot's own caller is blocked inside gtop_cost_nlopt_shared right now (a leave from another thread): taking
    // it out here would leave `arrived` counting a caller that is gone, and the others waiting for ever
    // — or awake again and still copying its results out of the buffers the next launch would overwrite
    if (s->waiting || s->in_call) return GTOP_ERR_STATE;
    s->active = false;          // its row keeps its last x: evaluated along, never read
    r->active--;
    leader = r->active > 0 && r->arrived == r->active && !r->broken.load(std::memory_order_relaxed);   // the others were only waiting for this one
    if (leader) r->evaluating = true;
  }
  if (leader) run_generation(r);
  return GTOP_OK;
} GTOP_CATCH_STATUS(nullptr)

int gtop_rendezvous_set_timeout(gtop_rendezvous *r, double seconds) try {
  if (!r || !(seconds >= 0.0)) return GTOP_ERR_INVALID;
  r->timeout_s.store(seconds, std::memory_order_relaxed);
  return GTOP_OK;
} GTOP_CATCH_STATUS(nullptr)

int gtop_rendezvous_abort(gtop_rendezvous *r) try {
  if (!r) return GTOP_ERR_INVALID;
  break_rendezvous(r);
  return GTOP_OK;
} GTOP_CATCH_STATUS(nullptr)

int gtop_rendezvous_stats(gtop_rendezvous *r, int64_t *launches, double *launch_seconds, int64_t *callbacks) try {
  if (!r) return GTOP_ERR_INVALID;
  std::lock_guard<std::mutex> lk(r->mu);
  if (launches) *launches = r->launches;
  if (launch_seconds) *launch_seconds = r->launch_seconds;
  if (callbacks) {
    int64_t t = 0;
    for (const auto &s : r->slots) t += s.calls;
    *callbacks = t;
  }
  return GTOP_OK;
} GTOP_CATCH_STATUS(nullptr)

}  // extern "C"
