"""The rendezvous layer (include/gtop.h, gtop_rendezvous_*): N host threads, each running its own serial
optimizer on its own trajectory — the reference's usage, one NLopt instance per problem calling costFunc
serially (src/grad_traj_optimizer.cpp:137-195, :554-562) — meet in shared launches.  Every thread must see,
bit for bit, what it sees when the optimizers run one after the other through gtop_cost_nlopt."""
import json
import os
import subprocess
import threading

import numpy as np
import pytest

from grad_traj_optimization_amd import problem

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DEMO = os.path.join(ROOT, "grad_traj_optimization_amd", "gtop_rendezvous_demo")


@pytest.mark.parametrize("threads,m,evals,spl", [(64, 6, 30, 0), (96, 4, 20, 0), (300, 6, 12, 3)])
def test_threads_sharing_launches_equal_the_serial_runs(threads, m, evals, spl):
    """C++ threads + csrc/mma.hpp (tests/cpp/rendezvous_threads.cpp).  Callers stop after different numbers of
    evaluations, so the leave path runs too.  (300 threads: geometry pinned to one wavefront per trajectory —
    the auto rule would serve a lone trajectory and a batch of 300 with different bodies.)"""
    assert os.path.exists(DEMO), "build() did not produce gtop_rendezvous_demo"
    out = subprocess.run([DEMO, str(threads), str(m), str(evals), str(spl)], capture_output=True, text=True,
                         timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    r = json.loads(out.stdout)
    assert r["identical"] is True and r["max_abs_dx"] == 0.0
    assert r["callbacks"] == r["serial_callbacks"] == sum(evals - (i % 5) for i in range(threads))
    assert r["launches"] == evals                      # one launch per generation: the longest-running caller's count
    assert r["fraction_improved"] > 0.9
    # (wall-clock figures — shared_us_per_callback against serial_us_per_callback — are reported by
    # tools/host_api_rate.py, not asserted: with more threads than cores they depend on the box, not the code)


def test_python_threads_and_error_paths(gtop, oracle_mod):
    mp = problem.make_map((40, 40, 20), density=0.03, seed=51)
    b = problem.make_trajectories(8, 5, mp, seed=52)
    ctx = gtop.GtopContext(device=0)
    ctx.init_sdf_map(mp.map_size, mp.origin, mp.resolution)
    ctx.update_sdf_map(mp.obstacle_points())
    ctx.set_problem(b.T, b.Df)
    c_ref, g_ref = ctx.eval_batch(b.x)
    rdv = gtop.Rendezvous(ctx, 8, 5)
    got = {}

    def worker(i):
        try:
            for k in range(3 + i % 3):                  # unequal call counts
                got[(i, k)] = rdv.cost(i, b.x[i] + 0.01 * k)
        finally:
            rdv.leave(i)                                # (an error in one caller must not strand the others)

    th = [threading.Thread(target=worker, args=(i,), daemon=True) for i in range(8)]
    for t in th:
        t.start()
    for t in th:
        t.join(timeout=120)
        assert not t.is_alive()
    for i in range(8):
        c, g = got[(i, 0)]
        assert c == c_ref[i] and np.array_equal(g, g_ref[i])      # the row of a plain batch evaluation
    st = rdv.stats()
    assert st["launches"] == 5 and st["callbacks"] == sum(3 + i % 3 for i in range(8))
    with pytest.raises(gtop.GtopError):
        rdv.cost(0, b.x[0])                             # slot 0 has left
    rdv2 = gtop.Rendezvous(ctx, 1, 5)
    with pytest.raises(gtop.GtopError):
        rdv2.cost(0, b.x[0][:-1])                       # n != 9(m-1)
    c, _ = rdv2.cost(0, b.x[0], want_grad=False)        # a single caller never waits
    assert c == c_ref[0]


def test_misuse_does_not_strand_the_callers(gtop):
    """A caller that exits without leaving, and a leave from another thread on a slot that is waiting: with a timeout
    set the others come back with an error instead of sleeping for ever; abort wakes them at once."""
    import time
    mp = problem.make_map((40, 40, 20), density=0.03, seed=51)
    b = problem.make_trajectories(3, 5, mp, seed=53)
    ctx = gtop.GtopContext(device=0)
    ctx.init_sdf_map(mp.map_size, mp.origin, mp.resolution)
    ctx.update_sdf_map(mp.obstacle_points())
    ctx.set_problem(b.T, b.Df)
    for how in ("timeout", "abort"):
        rdv = gtop.Rendezvous(ctx, 3, 5)
        if how == "timeout":
            rdv.set_timeout(0.3)
        out = {}

        def waiter(i, rdv=rdv, out=out):
            t0 = time.perf_counter()
            try:
                rdv.cost(i, b.x[i])
                out[i] = "returned"
            except gtop.GtopError:
                out[i] = time.perf_counter() - t0

        th = [threading.Thread(target=waiter, args=(i,), daemon=True) for i in (0, 1)]     # caller 2 never shows up
        for t in th:
            t.start()
        time.sleep(0.05)
        assert rdv.leave(0) == 4                         # GTOP_ERR_STATE: slot 0's caller is inside the call
        if how == "abort":
            rdv.abort()
        for t in th:
            t.join(timeout=30)
            assert not t.is_alive()
        assert all(isinstance(out[i], float) for i in (0, 1)), out
        if how == "timeout":
            assert min(out.values()) >= 0.25
        with pytest.raises(gtop.GtopError):
            rdv.cost(2, b.x[2])                          # broken for everybody, late arrivals included


def test_leave_after_a_break_and_timeout_against_a_slow_launch(gtop, monkeypatch):
    """Round 4 (advisor): (1) a waiter's timeout must not fire while the elected caller is inside the launch — only a
    caller that never ARRIVES breaks the rendezvous; (2) after a break the callers that came back with an error are no
    longer counted as waiting, so leave() on their slots succeeds; (3) destroy waits for a launch in flight instead
    of freeing the buffers under it.  GTOP_RENDEZVOUS_TEST_DELAY_MS makes the leader sleep before its launch."""
    import time
    mp = problem.make_map((40, 40, 20), density=0.03, seed=51)
    b = problem.make_trajectories(3, 5, mp, seed=53)
    ctx = gtop.GtopContext(device=0)
    ctx.init_sdf_map(mp.map_size, mp.origin, mp.resolution)
    ctx.update_sdf_map(mp.obstacle_points())
    ctx.set_problem(b.T, b.Df)
    c_ref, g_ref = ctx.eval_batch(b.x)

    # (2) break by timeout (caller 2 never shows up), then leave on every slot
    rdv = gtop.Rendezvous(ctx, 3, 5)
    rdv.set_timeout(0.2)
    out = {}

    def waiter(i, rdv, out):
        try:
            out[i] = rdv.cost(i, b.x[i])
        except gtop.GtopError:
            out[i] = "broken"

    th = [threading.Thread(target=waiter, args=(i, rdv, out), daemon=True) for i in (0, 1)]
    for t in th:
        t.start()
    for t in th:
        t.join(timeout=30)
        assert not t.is_alive()
    assert out == {0: "broken", 1: "broken"}
    assert [rdv.leave(i) for i in range(3)] == [0, 0, 0]      # (round 3: GTOP_ERR_STATE for ever on slots 0 and 1)
    rdv.close()

    # (1) everybody arrives, the launch takes 0.6 s, the timeout is 0.15 s: no break, everybody gets its row
    monkeypatch.setenv("GTOP_RENDEZVOUS_TEST_DELAY_MS", "600")
    rdv = gtop.Rendezvous(ctx, 3, 5)
    rdv.set_timeout(0.15)
    out = {}
    th = [threading.Thread(target=waiter, args=(i, rdv, out), daemon=True) for i in range(3)]
    t0 = time.perf_counter()
    for t in th:
        t.start()
    for t in th:
        t.join(timeout=30)
        assert not t.is_alive()
    assert time.perf_counter() - t0 >= 0.55
    for i in range(3):
        assert out[i] != "broken", out
        assert out[i][0] == c_ref[i] and np.array_equal(out[i][1], g_ref[i])
    assert [rdv.leave(i) for i in range(3)] == [0, 0, 0]
    rdv.close()

    # (3) abort while the leader is inside its (slow) launch, destroy at once: destroy waits for the launch
    rdv = gtop.Rendezvous(ctx, 3, 5)
    out = {}
    th = [threading.Thread(target=waiter, args=(i, rdv, out), daemon=True) for i in range(3)]
    for t in th:
        t.start()
    time.sleep(0.1)                       # all three have arrived; the leader sleeps in front of its launch
    rdv.abort()
    t0 = time.perf_counter()
    rdv.close()                           # gtop_rendezvous_destroy
    waited = time.perf_counter() - t0
    for t in th:
        t.join(timeout=30)
        assert not t.is_alive()
    assert waited >= 0.3, waited          # it did not free the buffers under the launch
    assert all(v == "broken" for v in out.values()), out
    ctx.close()
