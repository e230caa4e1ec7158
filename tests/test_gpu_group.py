"""The multi-device entry of the C-ABI (include/gtop.h, gtop_group_*) on what one card allows: a group
that lists device 0 several times (peer-copy gather), and a group of one device with an RCCL communicator of size 1.
Results must be the unsharded evaluation's, bit for bit — sharding has no reduction anywhere.  (N > 1 devices is the
same code with different ordinals; it has not run on hardware from here — DESIGN §7.)"""
import numpy as np
import pytest

from grad_traj_optimization_amd import problem

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def scene(gtop):
    mp = problem.make_map((60, 50, 30), density=0.03, seed=11)
    ctx = gtop.GtopContext(device=0)
    ctx.init_sdf_map(mp.map_size, mp.origin, mp.resolution)
    ctx.update_sdf_map(mp.obstacle_points())
    return mp, ctx


def _group(gtop, mp, devices):
    g = gtop.GtopGroup(devices)
    g.init_sdf_map(mp.map_size, mp.origin, mp.resolution)
    g.update_sdf_map(mp.obstacle_points())
    return g


@pytest.mark.parametrize("members,B,m", [(3, 1000, 6), (4, 23, 6), (2, 4500, 12), (5, 3, 6)])
def test_slices_on_one_card_equal_the_unsharded_batch(scene, gtop, members, B, m):
    """n contexts on device 0, peer-copy gather: host-buffer results, the gathered copies every member holds, and
    the slice bounds (contiguous, ceil(B / n) rows; members past the batch's end hold nothing)."""
    mp, ctx = scene
    b = problem.make_trajectories(B, m, mp, seed=40 + members, step_len=(0.5, 1.2) if m > 6 else (1.0, 2.0),
                                  boundary="random")
    ctx.set_params()
    ctx.set_problem(b.T, b.Df)
    c_ref, g_ref = ctx.eval_batch(b.x)
    g = _group(gtop, mp, [0] * members)
    assert g.gather_backend == "copy"
    g.set_problem(b.T, b.Df)
    per = -(-B // members)
    assert g.shards() == [(min(B, i * per), min(B, (i + 1) * per) - min(B, i * per)) for i in range(members)]
    c, gr = g.eval_batch(b.x)
    assert np.array_equal(c, c_ref) and np.array_equal(gr, g_ref)
    for ci, gi in g.eval_resident(gather=2):                        # x is resident from the call above
        assert np.array_equal(ci, c_ref) and np.array_equal(gi, g_ref)
    x2 = b.x + 0.01
    c2_ref, _ = ctx.eval_batch(x2)
    for ci, gi in g.eval_resident(x2, gather=1):
        assert np.array_equal(ci, c2_ref) and gi is None
    g.close()


def test_group_of_one_with_an_rccl_communicator(scene, gtop, monkeypatch):
    """All listed devices differ (there is one): the device-side gather is RCCL's ncclAllGather, communicator of
    size 1 — the call path N > 1 devices take."""
    mp, ctx = scene
    monkeypatch.setenv("GTOP_GROUP_GATHER", "rccl")       # a missing or failing librccl is an error, not a fallback
    g = _group(gtop, mp, [0])
    monkeypatch.delenv("GTOP_GROUP_GATHER")
    assert g.gather_backend == "rccl"
    b = problem.make_trajectories(777, 6, mp, seed=50)
    ctx.set_params()
    ctx.set_problem(b.T, b.Df)
    c_ref, g_ref = ctx.eval_batch(b.x)
    g.set_problem(b.T, b.Df)
    for _ in range(2):
        (ci, gi), = g.eval_resident(b.x, gather=2)
        assert np.array_equal(ci, c_ref) and np.array_equal(gi, g_ref)
    with pytest.raises(gtop.GtopError):
        monkeypatch.setenv("GTOP_GROUP_GATHER", "rccl")
        gtop.GtopGroup([0, 0])                            # RCCL wants one rank per device
    g.close()


def test_group_optimizer_and_errors(scene, gtop):
    mp, ctx = scene
    B, m = 300, 6
    b = problem.make_trajectories(B, m, mp, seed=60)
    lb, ub = gtop.GtopContext.default_bounds(b.waypoints)
    ctx.set_params()
    ctx.set_problem(b.T, b.Df)
    ref = ctx.optimize_batch_ex(b.x, lb, ub, 20, ftol_rel=1e-3)
    g = _group(gtop, mp, [0, 0, 0])
    with pytest.raises(gtop.GtopError) as e:
        g.eval_batch(b.x)                                 # no problem yet
    assert e.value.code == 4
    g.set_problem(b.T, b.Df)
    got = g.optimize_batch_ex(b.x, lb, ub, 20, ftol_rel=1e-3)
    for a, r in zip(got, ref):
        assert np.array_equal(a, r)
    with pytest.raises(gtop.GtopError) as e:
        g.eval_batch(b.x[:10])                            # the group's batch is fixed by set_problem
    assert e.value.code == 1
    with pytest.raises(gtop.GtopError):
        gtop.GtopGroup([0, 99])                           # no such device
    g.close()


def test_cpp_batch_class_over_the_group(tmp_path):
    """GradTrajBatch (csrc/grad_traj_optimizer.hpp): the C++ host side for MANY trajectories on the listed devices —
    tests/cpp/batch_devices.cpp runs 200 copies of the reference's opti_node path (small offsets per copy) on three
    members (device 0 three times) and, for five of them, the same problem through one GradTrajOptimizer object:
    the same minimum and coefficients (the single-problem loop runs the same body, so to rounding of the final
    evaluation only)."""
    import json
    import os
    import subprocess
    from tests import scenes
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(root, "grad_traj_optimization_amd", "gtop_batch_devices")
    assert os.path.exists(exe), "build() did not produce gtop_batch_devices"
    f = scenes.write_scene(tmp_path / "opti_node.txt", scenes.OPTI_NODE_MAP_SIZE, scenes.OPTI_NODE_ORIGIN,
                           scenes.OPTI_NODE_RES, scenes.opti_node_obstacles(), scenes.OPTI_NODE_PATH)
    out = subprocess.run([exe, str(f), "200", "30", "0", "0", "0"], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr
    r = json.loads(out.stdout[out.stdout.index("{"):])
    assert r["B"] == 200 and r["devices"] == 3 and r["gather"] == "copy"
    assert r["min_evals"] == r["max_evals"] == 30
    assert r["max_rel_cost_diff"] <= 1e-9 and r["max_coeff_diff"] <= 1e-9, r


def test_cpp_batch_class_with_differing_waypoint_counts(tmp_path):
    """A ragged batch: the copies of the opti_node path with 11, 10 and 9 waypoints in turn (candidate paths seldom
    agree on their number of waypoints).  GradTrajBatch forms one device problem per segment count over the same
    group (slice buffers reused, the field untouched) and hands every trajectory its own result: the same minimum
    and coefficients as one GradTrajOptimizer object per path."""
    import json
    import os
    import subprocess
    from tests import scenes
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(root, "grad_traj_optimization_amd", "gtop_batch_devices")
    assert os.path.exists(exe), "build() did not produce gtop_batch_devices"
    f = scenes.write_scene(tmp_path / "opti_node.txt", scenes.OPTI_NODE_MAP_SIZE, scenes.OPTI_NODE_ORIGIN,
                           scenes.OPTI_NODE_RES, scenes.opti_node_obstacles(), scenes.OPTI_NODE_PATH)
    out = subprocess.run([exe, str(f), "100", "25", "0", "0", "ragged"], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, (out.returncode, out.stderr)
    r = json.loads(out.stdout[out.stdout.index("{"):])
    assert r["B"] == 100 and r["devices"] == 2
    assert (r["min_segments"], r["max_segments"]) == (8, 10)
    assert r["min_evals"] == r["max_evals"] == 25
    assert r["max_rel_cost_diff"] <= 1e-9 and r["max_coeff_diff"] <= 1e-9, r


def test_group_walks_through_problems_of_different_sizes(scene, gtop):
    """One group, problem after problem (as GradTrajBatch does per segment count): the slice buffers are reused where
    large enough and grown where not, the replicated maps are rebuilt in between — every result the unsharded one's."""
    mp, ctx = scene
    g = _group(gtop, mp, [0, 0, 0])
    ctx.set_params()
    rng = np.random.default_rng(5)
    for k, (B, m) in enumerate([(500, 6), (40, 12), (900, 3), (7, 17), (901, 6), (2, 2)]):
        if k == 3:      # another map half way: every member rebuilds its copy, the single context too
            mp2 = problem.make_map((60, 50, 30), density=0.05, seed=99)
            g.update_sdf_map(mp2.obstacle_points())
            ctx.update_sdf_map(mp2.obstacle_points())
        b = problem.make_trajectories(B, m, mp, seed=200 + k, step_len=(0.4, 0.9), boundary="random" if k % 2 else None)
        ctx.set_problem(b.T, b.Df)
        g.set_problem(b.T, b.Df)
        c_ref, g_ref = ctx.eval_batch(b.x)
        c, gr = g.eval_batch(b.x)
        assert np.array_equal(c, c_ref) and np.array_equal(gr, g_ref), (k, B, m)
        for ci, gi in g.eval_resident(gather=2):
            assert np.array_equal(ci, c_ref) and np.array_equal(gi, g_ref), (k, B, m)
        if k in (1, 4):
            lb, ub = gtop.GtopContext.default_bounds(b.waypoints)
            ref = ctx.optimize_batch_ex(b.x, lb, ub, int(rng.integers(5, 15)))
            got = g.optimize_batch_ex(b.x, lb, ub, ref[2].max())
            for a, r in zip(got, ref):
                assert np.array_equal(a, r), (k, B, m)
    ctx.update_sdf_map(mp.obstacle_points())     # (the module's scene as the other tests expect it)
    g.close()


def test_cpp_batch_class_with_fp32_evaluations(tmp_path):
    """Config::optimizer_fp32 through GradTrajBatch (every member context) and through single GradTrajOptimizer objects:
    the same loop on the same rows, so the batch's minima and coefficients are the objects' (the final cost is
    re-evaluated in fp64 by costFunc: equal to the fp32 bound)."""
    import json
    import os
    import subprocess
    from tests import scenes
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(root, "grad_traj_optimization_amd", "gtop_batch_devices")
    f = scenes.write_scene(tmp_path / "opti_node.txt", scenes.OPTI_NODE_MAP_SIZE, scenes.OPTI_NODE_ORIGIN,
                           scenes.OPTI_NODE_RES, scenes.opti_node_obstacles(), scenes.OPTI_NODE_PATH)
    out = subprocess.run([exe, str(f), "60", "20", "0", "0", "fp32"], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, (out.returncode, out.stderr)
    r = json.loads(out.stdout[out.stdout.index("{"):])
    assert r["B"] == 60 and r["min_evals"] == r["max_evals"] == 20
    assert r["max_rel_cost_diff"] <= 2e-4 and r["max_coeff_diff"] <= 1e-9, r


def test_new_problem_while_a_gather_is_in_flight(scene, gtop):
    """Regression for the race e18f164 closed (round 3, no test then): gtop_group_eval_resident(synchronize = 0) leaves
    the slices' kernels and every member's copies into the OTHER members' gathered rows in flight; a
    gtop_group_set_problem right behind it (another batch size: buffers are reused, grown, memset) must wait for all of
    them — every member's stream, not just the buffer owner's — before it touches anything.  Alternating a large and a
    small problem with nothing in between but the enqueue: every result must still be the unsharded evaluation's."""
    mp, ctx = scene
    g = _group(gtop, mp, [0, 0, 0])
    assert g.gather_backend == "copy" and "listed more than once" in g.gather_note()
    ctx.set_params()
    # (the auto rule picks the body from the batch size: a slice of 2 000 rows and the whole batch of 6 000 would sum
    # in different orders — pin five lanes per segment on both sides)
    ctx.set_launch_geometry(0, 6)
    g.set_launch_geometry(0, 6)
    sizes = [(6000, 6, 1), (37, 6, 2), (9000, 6, 3), (5, 4, 4), (7000, 12, 5), (64, 6, 6)]
    refs = []
    for B, m, seed in sizes:
        b = problem.make_trajectories(B, m, mp, seed=300 + seed, step_len=(0.5, 1.2) if m > 6 else (1.0, 2.0))
        ctx.set_problem(b.T, b.Df)
        refs.append((b, ctx.eval_batch(b.x)))
    for rep in range(3):
        for (b, (c_ref, g_ref)) in refs:
            g.set_problem(b.T, b.Df)              # behind whatever the previous round left in flight
            g.launch_resident(b.x, gather=2)      # enqueue only
            if rep == 2:                          # last pass: look at what every member holds
                g.synchronize()
                for i in range(3):
                    c, gr = g.read_gathered(i, grads=True)
                    assert np.array_equal(c, c_ref) and np.array_equal(gr, g_ref), (len(c_ref), i)
    g.close()
    ctx.set_launch_geometry(0, 0)


def test_more_than_one_distinct_device(gtop):
    """The paths only n > 1 DISTINCT devices take — ncclCommInitAll over several ordinals, one all-gather per device in
    a group call, cross-device events, the replicated map built on every card — need a multi-GPU box: skipped on the
    one-GPU boxes this repository has had, so those paths are UNVERIFIED ON HARDWARE (README)."""
    import torch
    n = torch.cuda.device_count()
    if n < 2:
        pytest.skip(f"needs at least 2 GPUs ({n} visible): the n > 1 distinct-device paths have not run on hardware")
    mp = problem.make_map((60, 50, 30), density=0.03, seed=11)
    ctx = gtop.GtopContext(device=0)
    ctx.init_sdf_map(mp.map_size, mp.origin, mp.resolution)
    ctx.update_sdf_map(mp.obstacle_points())
    ctx.set_params()
    devs = list(range(min(n, 6)))
    for force in ("rccl", "copy"):
        import os
        os.environ["GTOP_GROUP_GATHER"] = force
        try:
            g = _group(gtop, mp, devs)
        finally:
            del os.environ["GTOP_GROUP_GATHER"]
        assert g.gather_backend == force, g.gather_note()
        for B, m in ((4099, 6), (257, 12)):
            b = problem.make_trajectories(B, m, mp, seed=500 + B, step_len=(0.5, 1.2) if m > 6 else (1.0, 2.0))
            ctx.set_problem(b.T, b.Df)
            c_ref, g_ref = ctx.eval_batch(b.x)
            g.set_problem(b.T, b.Df)
            c, gr = g.eval_batch(b.x)
            assert np.array_equal(c, c_ref) and np.array_equal(gr, g_ref)
            for ci, gi in g.eval_resident(gather=2):
                assert np.array_equal(ci, c_ref) and np.array_equal(gi, g_ref)
        g.close()


def test_group_window_update_keeps_the_replicas_identical(scene, gtop, oracle_mod):
    """gtop_group_update_sdf_map_window: the reference's local map update on every member — afterwards every member's
    field is the oracle's, and a sharded evaluation equals the unsharded one on that field."""
    mp, _ = scene
    g = _group(gtop, mp, [0, 0])
    sdf = oracle_mod.Sdf.from_map_size(mp.origin, mp.resolution, mp.map_size)
    occ = sdf.build_from_points(mp.obstacle_points()).copy()
    rng = np.random.default_rng(5)
    a = mp.origin + 0.2 * mp.map_size
    b_ = mp.origin + 0.7 * mp.map_size
    pts = rng.uniform(a, b_, size=(60, 3))
    g.update_sdf_map_window(a, b_, pts)
    sdf.update_window(occ, a, b_, pts)
    ref = gtop.GtopContext(device=0)
    ref.set_sdf(sdf.dist, sdf.grid, mp.origin, mp.resolution, map_size=mp.map_size)
    ref.set_params()
    tr = problem.make_trajectories(300, 6, mp, seed=9)
    ref.set_problem(tr.T, tr.Df)
    c_ref, g_ref = ref.eval_batch(tr.x)
    g.set_problem(tr.T, tr.Df)
    c, gr = g.eval_batch(tr.x)
    assert np.array_equal(c, c_ref) and np.array_equal(gr, g_ref)
    g.close()
    ref.close()
