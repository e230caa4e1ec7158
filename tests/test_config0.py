"""BASELINE.json configs[0] at its stated size: ONE trajectory, 7 waypoints (6 segments, 21 control points, n = 45),
a 100 x 100 x 50 random map — the reference's single-problem usage (one optimizer instance calling costFunc
serially, src/grad_traj_optimizer.cpp:137-195, :554-562).  The CPU leg is plumbing (oracle callback + the serial
CCSA-MMA twin); the GPU legs go through the NLopt-shaped entry gtop_cost_nlopt and through the C++ shim's
optimizeTrajectory, host loop and one-launch device loop."""
import numpy as np
import pytest

from grad_traj_optimization_amd import problem
from tests import scenes

GRID = (100, 100, 50)


def _config0():
    mp = problem.make_map(GRID, density=0.02, seed=100)
    b = problem.make_trajectories(1, 6, mp, seed=101)
    return mp, b


@pytest.fixture(scope="module")
def cpu_scene(oracle_mod):
    mp, b = _config0()
    sdf = oracle_mod.Sdf.from_map_size(mp.origin, mp.resolution, mp.map_size)
    assert sdf.grid == GRID
    sdf.build_from_points(mp.obstacle_points())
    return mp, b, sdf


def test_config0_cpu_plumbing(oracle_mod, cpu_scene):
    """The CPU path on its own: setPath's time allocation and straight-line start, the callback, and a serial
    CCSA-MMA run around it (what the reference does with NLopt's LD_MMA) lowering the cost."""
    from tests.test_optimizer import mma_serial
    mp, b, sdf = cpu_scene
    path = b.waypoints[0]
    assert path.shape == (7, 3)
    T = oracle_mod.segment_time(path)
    Df, Dp = oracle_mod.initial_d(path)
    assert np.array_equal(T, b.T[0]) and np.array_equal(Df, b.Df[0])
    gen = oracle_mod.generator(T)
    prm = oracle_mod.make_params()
    x0 = Dp.reshape(-1)
    assert x0.size == 45

    def f(x):
        return oracle_mod.cost_grad(T, Df, x, sdf, prm, L=gen["L"], R=gen["R"])
    c0, g0 = f(x0)
    assert np.isfinite(c0) and c0 >= 1e-3 and g0.shape == (45,)
    lb = np.empty(45)
    ub = np.empty(45)
    for i in range(15):                      # bounds, grad_traj_optimizer.cpp:151-179 (bos 3, vos 8, aos 10)
        for a in range(3):
            j = i + 15 * a
            if i % 3 == 0:
                lb[j], ub[j] = path[i // 3 + 1, a] - 3.0, path[i // 3 + 1, a] + 3.0
            else:
                lb[j], ub[j] = ((-8.0, 8.0), (-10.0, 10.0))[i % 3 - 1]
    x, fmin, trace = mma_serial(f, x0, lb, ub, 30)
    assert fmin < c0 and np.all(np.diff(trace) <= 0)
    assert abs(f(x)[0] - fmin) <= 1e-12 * fmin


@pytest.mark.gpu
def test_config0_nlopt_entry_point(gtop, oracle_mod, cpu_scene):
    mp, b, sdf = cpu_scene
    ctx = gtop.GtopContext(device=0)
    ctx.init_sdf_map(mp.map_size, mp.origin, mp.resolution)
    assert tuple(ctx.grid) == GRID
    ctx.update_sdf_map(mp.obstacle_points())
    assert np.array_equal(ctx.get_sdf().reshape(-1), sdf.dist)
    x0 = ctx.set_paths(b.waypoints)                                  # setPath on the device
    T, Df = ctx.get_problem()
    assert np.array_equal(T, b.T) and np.array_equal(Df, b.Df)
    prm = oracle_mod.make_params()
    for x in (x0[0], b.x[0]):                                        # the straight-line start and a perturbed point
        c, g = ctx.cost_nlopt(x)
        c_ref, g_ref = oracle_mod.cost_grad(b.T[0], b.Df[0], x, sdf, prm)
        rc, rg = scenes.rel_err(c, g, c_ref, g_ref)
        assert rc <= 1e-5 and rg <= 1e-5, (rc, rg)
    assert ctx.stats()[0] == 2 and len(ctx.cost_curve()[0]) == 2     # iter_num, cost curve (:284, :439-447)


@pytest.mark.gpu
def test_config0_optimize_trajectory_through_the_shim(oracle_mod, cpu_scene, tmp_path):
    mp, b, sdf = cpu_scene
    f = scenes.write_scene(tmp_path / "config0.txt", mp.map_size, mp.origin, mp.resolution, mp.obstacle_points(),
                           b.waypoints[0])
    host = scenes.run_scene(f, 50)
    dev = scenes.run_scene(f, 50, on_device=1)
    T, Df = b.T[0], b.Df[0]
    assert np.array_equal(np.array(host["segment_time"]), T)
    _, Dp = oracle_mod.initial_d(b.waypoints[0])
    assert np.array_equal(np.array(host["x0"]), Dp.reshape(-1))
    # (before optimizeTrajectory the object's step member is 1, i.e. ws = 0; afterwards 2: grad_traj_optimizer.h)
    for xk, ck, gk, step in (("x0", "cost0", "grad0", 1), ("x1", "cost1", "grad1", 2)):
        for run in (host, dev):
            c_ref, g_ref = oracle_mod.cost_grad(T, Df, np.array(run[xk]), sdf, oracle_mod.make_params(step=step))
            rc, rg = scenes.rel_err(run[ck], run[gk], c_ref, g_ref)
            assert rc <= 1e-5 and rg <= 1e-5, (xk, rc, rg)
    c_start = oracle_mod.cost_grad(T, Df, np.array(host["x0"]), sdf, oracle_mod.make_params())[0]
    assert host["evals"] == dev["evals"] == 50
    assert host["cost1"] < c_start
    assert abs(dev["cost1"] - host["cost1"]) <= 1e-6 * host["cost1"]           # same algorithm, same cap
    assert np.max(np.abs(np.array(dev["x1"]) - np.array(host["x1"]))) <= 1e-6 * max(1.0, np.max(np.abs(host["x1"])))
    curve = np.array(host["cost_curve"])
    assert len(curve) >= 50 and np.all(np.diff(curve[:50]) <= 0)
