"""The C++ host shim (GradTrajOptimizer-compatible class) driven through the calls the
reference's opti_node executable makes (src/opti_node.cpp:58-106) on that executable's scene
(tests/scenes.py holds it as data; tests/cpp/scene_runner.cpp reads the scene file), checked
against the oracle on the same scene."""
import numpy as np
import pytest

from tests import scenes

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def scene_file(tmp_path_factory):
    f = tmp_path_factory.mktemp("scene") / "opti_node.txt"
    return scenes.write_scene(f, scenes.OPTI_NODE_MAP_SIZE, scenes.OPTI_NODE_ORIGIN, scenes.OPTI_NODE_RES,
                              scenes.opti_node_obstacles(), scenes.OPTI_NODE_PATH)


@pytest.fixture(scope="module")
def run(scene_file):
    return scenes.run_scene(scene_file, 40)


def test_costfunc_signature_matches_oracle(run, oracle_mod):
    sdf = oracle_mod.Sdf.from_map_size(scenes.OPTI_NODE_ORIGIN, scenes.OPTI_NODE_RES, scenes.OPTI_NODE_MAP_SIZE)
    sdf.build_from_points(scenes.opti_node_obstacles())
    T = oracle_mod.segment_time(scenes.OPTI_NODE_PATH)
    Df, Dp = oracle_mod.initial_d(scenes.OPTI_NODE_PATH)
    assert run["n_obstacle_points"] == 3100
    assert np.array_equal(np.array(run["segment_time"]), T)                 # setPath's time allocation
    assert np.array_equal(np.array(run["x0"]), Dp.reshape(-1))              # straight-line Dp
    # before optimizeTrajectory the object's `step` member still holds its
    # default 1 (grad_traj_optimizer.h: `int step = 1`), i.e. ws = 0; afterwards 2
    for xk, ck, gk, step in (("x0", "cost0", "grad0", 1), ("x1", "cost1", "grad1", 2)):
        c_ref, g_ref = oracle_mod.cost_grad(T, Df, np.array(run[xk]), sdf, oracle_mod.make_params(step=step))
        rc, rg = scenes.rel_err(run[ck], run[gk], c_ref, g_ref)
        assert rc <= 1e-5 and rg <= 1e-5, (xk, rc, rg)


def test_initial_coefficients_are_the_straight_line(run, oracle_mod):
    """getCoefficient before optimisation = A^-1 Dx of the straight-line init
    (src/qp_generator.cpp:334-351): compare with L d from the oracle's generator."""
    T = oracle_mod.segment_time(scenes.OPTI_NODE_PATH)
    Df, Dp = oracle_mod.initial_d(scenes.OPTI_NODE_PATH)
    L = oracle_mod.generator(T)["L"]
    coe = np.zeros((10, 18))
    for a in range(3):
        coe[:, 6 * a:6 * a + 6] = (L @ np.concatenate([Df[a], Dp[a]])).reshape(10, 6)
    got = np.array(run["coeff0"]).reshape(10, 18)
    assert np.allclose(got, coe, rtol=1e-9, atol=1e-9 * np.abs(coe).max())


def test_optimizer_improves_and_bookkeeping(run):
    curve = np.array(run["cost_curve"])
    assert curve[run["evals"] - 1] < 0.5 * curve[0]        # MMA made progress within its evaluation cap
    assert abs(curve[run["evals"] - 1] - run["cost1"]) <= 1e-9 * run["cost1"]   # the returned x is the best seen
    assert len(curve) >= run["evals"]                      # one entry per callback (:439-447)
    assert np.all(np.diff(curve[:run["evals"]]) <= 0)      # best-so-far is non-increasing
    # coefficients after optimisation are continuous at the interior waypoints
    T = np.array(run["segment_time"])
    c = np.array(run["coeff1"]).reshape(10, 18)
    for s in range(9):
        for a in range(3):
            end = sum(c[s, 6 * a + j] * T[s] ** j for j in range(6))
            assert abs(end - c[s + 1, 6 * a]) <= 1e-9 * max(1.0, abs(end))


def test_device_optimizer_option_follows_the_host_loop(run, scene_file):
    """Config::optimize_on_device: the same scene with the whole optimisation as one launch of the batched
    device optimizer (10 segments: the five-lanes-per-segment loop).  Same algorithm, same evaluation cap:
    the same minimum to rounding, no per-call cost curve."""
    dev = scenes.run_scene(scene_file, 40, on_device=1)
    assert dev["evals"] == run["evals"] == 40
    assert abs(dev["cost1"] - run["cost1"]) <= 1e-6 * run["cost1"]
    assert np.max(np.abs(np.array(dev["x1"]) - np.array(run["x1"]))) <= 1e-6 * max(1.0, np.max(np.abs(run["x1"])))
    assert len(dev["cost_curve"]) == 1          # the scene's own costFunc call after the optimisation, nothing per evaluation
    assert np.array_equal(np.array(dev["x0"]), np.array(run["x0"]))


def test_eigen_signature_adapter_delegates(run, scene_file):
    """include/grad_traj_optimization/grad_traj_optimizer.h: the global `GradTrajOptimizer` with the reference's Eigen
    signatures (grad_traj_optimizer.h:20-39), driven as src/opti_node.cpp:58-106 drives the reference's class.  It is a
    thin adapter over the shim, so the same scene must give the same bits as gtop_scene_runner.  (Compiled against
    tests/cpp/eigen_double — the image has no Eigen: a check of the adapter's syntax and delegation only.)"""
    ad = scenes.run_scene(scene_file, 40, exe_name="gtop_eigen_adapter")
    assert ad["evals"] == run["evals"] == 40
    assert np.array_equal(np.array(ad["segment_time"]), np.array(run["segment_time"]))
    assert np.array_equal(np.array(ad["coeff1"]), np.array(run["coeff1"]))
    curve = np.array(ad["cost_curve"])       # one entry per callback, best so far (:439-447); no call of its own in front
    assert len(curve) == 40 and np.all(np.diff(curve) <= 0)
    assert abs(curve[-1] - run["cost1"]) <= 1e-9 * run["cost1"]
