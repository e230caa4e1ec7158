"""CPU tests of the oracle itself: the C restatement against the independently
written numpy twin, against scipy's exact EDT, and against the structural
known answers that can be derived from the reference source.
The reference has no tests or golden vectors for this path: PARITY UNPINNED."""
import numpy as np
import pytest

from grad_traj_optimization_amd import problem
from oracle import np_twin
from tests import scenes


@pytest.fixture(scope="module")
def opti_scene(oracle_mod):
    sdf = oracle_mod.Sdf.from_map_size(scenes.OPTI_NODE_ORIGIN, scenes.OPTI_NODE_RES, scenes.OPTI_NODE_MAP_SIZE)
    obs = scenes.opti_node_obstacles()
    occ = sdf.build_from_points(obs)
    return sdf, obs, occ


def test_opti_node_scene_known_answers(oracle_mod, opti_scene):
    """grid 200x200x25, m = 10, n = 81; first segment alone gets init_time."""
    sdf, obs, occ = opti_scene
    assert sdf.grid == (200, 200, 25)
    assert len(obs) == 3100 and occ.sum() == 3100       # every point lands in its own voxel
    T = oracle_mod.segment_time(scenes.OPTI_NODE_PATH)
    assert T.shape == (10,)
    assert T[0] == np.sqrt(2.0) / 1.8 + 0.3 and T[9] == np.sqrt(2.0) / 1.8   # Q9: last gets no init_time
    assert T[1] == 1.0 / 1.8
    Df, Dp = oracle_mod.initial_d(scenes.OPTI_NODE_PATH)
    assert Dp.shape == (3, 27) and Dp.size == 81
    assert np.array_equal(Df[:, 0], scenes.OPTI_NODE_PATH[0]) and np.array_equal(Df[:, 3], scenes.OPTI_NODE_PATH[-1])
    assert np.count_nonzero(Df[:, [1, 2, 4, 5]]) == 0
    # unsigned field: occupied voxels are exactly 0, nothing negative (Q10)
    assert sdf.dist.min() == 0.0 and np.count_nonzero(sdf.dist == 0.0) == 3100


@pytest.mark.parametrize("m", [2, 3, 6, 10, 12])
def test_generator_structure(oracle_mod, m):
    rng = np.random.default_rng(m)
    T = rng.uniform(0.4, 1.6, m)
    g = oracle_mod.generator(T)
    L, R, A, Q, Ct = g["L"], g["R"], g["A"], g["Q"], g["Ct"]
    assert np.allclose(R, R.T, rtol=1e-10, atol=1e-8 * np.abs(R).max())       # R symmetric
    # A.3: row-block s of L touches only the columns of waypoints s and s+1 — exact zeros
    nd = 3 * m + 3

    def cols(wp):
        return range(0, 3) if wp == 0 else (range(3, 6) if wp == m else range(6 + 3 * (wp - 1), 9 + 3 * (wp - 1)))
    for s in range(m):
        allowed = set(cols(s)) | set(cols(s + 1))
        blk = L[6 * s:6 * s + 6]
        for c in range(nd):
            if c not in allowed:
                assert np.all(blk[:, c] == 0.0)
    # Q integer coefficients (A.1): 36, 72, 120, 192, 360, 720 times powers of T
    T0 = T[0]
    assert np.isclose(Q[3, 3], 36 * T0) and np.isclose(Q[3, 4], 72 * T0 ** 2) and np.isclose(Q[3, 5], 120 * T0 ** 3)
    assert np.isclose(Q[4, 4], 192 * T0 ** 3) and np.isclose(Q[4, 5], 360 * T0 ** 4) and np.isclose(Q[5, 5], 720 * T0 ** 5)
    # Ct is a 0/1 selection with exactly one 1 per row
    assert set(np.unique(Ct)) == {0.0, 1.0} and np.all(Ct.sum(axis=1) == 1)
    # the twin builds the same matrices from its own formulation
    gn = np_twin.generator(T)
    for k in ("A", "Q", "Ct"):
        assert np.array_equal(g[k], gn[k])
    for k in ("L", "R"):
        assert np.allclose(g[k], gn[k], rtol=1e-9, atol=1e-9 * np.abs(gn[k]).max())


def test_generator_rejects_single_segment(oracle_mod):
    with pytest.raises(ValueError):
        oracle_mod.generator(np.array([1.0]))   # StackOptiDep is out of bounds for m = 1


def test_cost_grad_matches_numpy_twin_opti_node(oracle_mod, opti_scene):
    sdf, _, _ = opti_scene
    T = oracle_mod.segment_time(scenes.OPTI_NODE_PATH)
    Df, Dp = oracle_mod.initial_d(scenes.OPTI_NODE_PATH)
    rng = np.random.default_rng(0)
    x = Dp.reshape(-1) + rng.normal(0, 0.05, Dp.size)
    nsdf = np_twin.Sdf(sdf.origin, sdf.resolution, sdf.grid, sdf.dist, max_range=list(sdf.c.max_range))
    for kw in (dict(), dict(step=1), dict(wc=0.0), dict(enable_dyn=1, alpha_v=2.0, alpha_a=1.5)):
        p = dict(oracle_mod.OPTI_NODE_PARAMS)
        p.update(kw)
        c, g = oracle_mod.cost_grad(T, Df, x, sdf, oracle_mod.make_params(**kw))
        cn, gn, info = np_twin.cost_grad(T, Df, x, nsdf, p)
        assert abs(c - cn) <= 1e-12 * abs(cn)
        assert np.max(np.abs(g - gn)) <= 1e-12 * np.max(np.abs(gn))
        if abs(p["wc"]) >= 1e-4:
            assert info["nsamples"] == [30] * 10          # 30 samples per segment for T > 0.031
        assert c >= 1e-3                                   # Q5


def test_sample_count_quirk_for_tiny_segment_time(oracle_mod):
    """A.2: `for (t = 1e-3; t < T; t += T/30)` takes 29 samples when T <= 0.03, 30 above."""
    grid = (20, 20, 20)
    sdf = np_twin.Sdf((-2, -2, 0), 0.2, grid, np.full(grid, 1.0))
    for Tval, want in ((0.03, 29), (0.031, 30), (0.5, 30), (0.0009, 0)):
        T = np.array([Tval, 0.5])
        path = np.array([(0, 0, 2.0), (0.001, 0, 2.0), (0.5, 0.2, 2.0)])
        Df, Dp = np_twin.initial_d(path)
        _, _, info = np_twin.cost_grad(T, Df, Dp.reshape(-1), sdf, dict(oracle_mod.OPTI_NODE_PARAMS))
        assert info["nsamples"][0] == want


def test_flags_and_offsets(oracle_mod, opti_scene):
    sdf, _, _ = opti_scene
    T = oracle_mod.segment_time(scenes.OPTI_NODE_PATH)
    Df, Dp = oracle_mod.initial_d(scenes.OPTI_NODE_PATH)
    x = Dp.reshape(-1) + 0.01
    # |wc| < 1e-4 skips the collision loop entirely (Q7); with ws = 0 too: cost = 1e-3, grad = 1e-5 (Q5)
    c, g = oracle_mod.cost_grad(T, Df, x, sdf, oracle_mod.make_params(ws=0.0, wc=5e-5))
    assert c == 1e-3 and np.all(g == 1e-5)
    # step == 1 zeroes the smoothness weight (:413-415)
    c1, g1 = oracle_mod.cost_grad(T, Df, x, sdf, oracle_mod.make_params(step=1))
    c2, g2 = oracle_mod.cost_grad(T, Df, x, sdf, oracle_mod.make_params(ws=0.0))
    assert c1 == c2 and np.array_equal(g1, g2)
    # alpha_v = alpha_a = 0 (opti_node.launch) makes the dyn block a no-op on the cost
    c3, _ = oracle_mod.cost_grad(T, Df, x, sdf, oracle_mod.make_params(enable_dyn=1))
    c4, _ = oracle_mod.cost_grad(T, Df, x, sdf, oracle_mod.make_params())
    assert c3 == c4


def test_sdf_query_conventions(oracle_mod):
    grid = (10, 8, 6)
    rng = np.random.default_rng(5)
    dist = rng.uniform(0, 3, grid)
    sdf = oracle_mod.Sdf((-1.0, -0.8, 0.0), 0.2, grid, dist)
    tw = np_twin.Sdf((-1.0, -0.8, 0.0), 0.2, grid, dist)
    # voxel centres return the voxel value; the gradient is finite
    for idx in ((0, 0, 0), (4, 3, 2), (9, 7, 5)):
        pos = (np.array(idx) + 0.5) * 0.2 + np.array((-1.0, -0.8, 0.0))
        d, g = sdf.query(pos)
        assert abs(d - dist[idx]) < 1e-12
    # border: base index -1 clamps (edge replicate, Q11) — value equals the border voxel
    d, g = sdf.query(np.array((-1.0 + 0.05, -0.8 + 0.05, 0.05)))
    assert abs(d - dist[0, 0, 0]) < 1e-12 and np.allclose(g, 0.0)
    # out of map (±1e-4 margins, Q12): dist = -1, grad = 0 by this build's convention (Q4)
    for pos in ((-1.0 + 5e-5, 0, 0.5), (1.0 - 5e-5, 0, 0.5), (0, 0, 1.3), (0, -0.9, 0.5)):
        d, g = sdf.query(np.array(pos))
        assert d == -1.0 and np.all(g == 0.0)
    for _ in range(200):
        pos = rng.uniform((-1.05, -0.85, -0.05), (1.05, 0.85, 1.25))
        d, g = sdf.query(pos)
        dn, gn = tw.query(pos)
        assert d == dn and np.array_equal(g, gn)


def test_edt_query_with_moving_boxes_known_answers(oracle_mod):
    """EDTEnvironment::evaluateEDTWithGrad restated (src/edt_environment.cpp:26-122): static-only
    queries are the trilinear query; a box's distance is the norm of per-axis face distances;
    the box centre moves with constant velocity."""
    grid = (12, 10, 8)
    rng = np.random.default_rng(7)
    dist = rng.uniform(1.0, 3.0, grid)
    org = np.array((-1.2, -1.0, 0.0))
    sdf = oracle_mod.Sdf(org, 0.2, grid, dist)
    none = np.zeros((0, 3))
    pos = rng.uniform(org + 0.2, org + np.array(grid) * 0.2 - 0.2, size=(50, 3))
    d, g = sdf.edt_query(pos, -1.0, none, none, none)
    for i, p in enumerate(pos):
        ds, gs = sdf.query(p)
        assert d[i] == ds and np.array_equal(g[i], gs)
    # a box given at t >= 0 is ignored at t < 0
    box = ([[0.0, 0.0, 0.8]], [[0.5, 0.0, 0.0]], [[0.5, 0.5, 0.5]])
    d2, _ = sdf.edt_query(pos, -0.5, *box)
    assert np.array_equal(d, d2)
    # every corner voxel centre inside the box -> value 0, gradient 0 (the box sits at x = 0.5 at t = 1)
    dq, gq = sdf.edt_query(np.array([[0.5, 0.0, 0.8]]), 1.0, *box)
    assert dq[0] == 0.0 and np.all(gq[0] == 0.0)
    # a query at a voxel centre returns that corner's value: min(static >= 1, box distance).  Box faces at
    # x in [0.25, 0.75] (t = 1), y in [-0.25, 0.25], z in [0.55, 1.05]; voxel centre (0.9, 0.1, 0.7) is 0.15 off in x only
    dq, _ = sdf.edt_query(np.array([[0.9, 0.1, 0.7]]), 1.0, *box)
    assert abs(dq[0] - 0.15) < 1e-9
    # off in two axes: norm of the per-axis distances (0.15, 0.25)
    dq, _ = sdf.edt_query(np.array([[0.9, 0.5, 0.7]]), 1.0, *box)
    assert abs(dq[0] - np.hypot(0.15, 0.25)) < 1e-9
    # out of the map: -1, zero gradient
    dq, gq = sdf.edt_query(np.array([[5.0, 0.0, 0.5]]), 1.0, *box)
    assert dq[0] == -1.0 and np.all(gq[0] == 0.0)


def test_esdf_against_twin_and_scipy(oracle_mod):
    """Three implementations of the same exact EDT: the C restatement, the
    numpy twin of the same lower-envelope sweeps, and scipy's EDT."""
    from scipy import ndimage
    mp = problem.make_map((18, 14, 10), density=0.06, seed=9, box_vox=(1, 3))
    sdf = oracle_mod.Sdf(mp.origin, mp.resolution, mp.grid)
    sdf.build_from_occupancy(mp.occupancy)
    d_c = sdf.dist.reshape(mp.grid)
    d_np = np_twin.esdf_build(mp.occupancy, mp.resolution)
    assert np.array_equal(d_c, d_np)
    d_sp = mp.resolution * ndimage.distance_transform_edt(mp.occupancy == 0)
    assert np.array_equal(d_c, np.minimum(d_sp, 10000.0))
    # an empty map keeps the 10000 sentinel (sdf_map.cpp:22)
    sdf.build_from_occupancy(np.zeros(mp.grid))
    assert np.all(sdf.dist == 10000.0)


def test_batch_driver_matches_single_calls(oracle_mod):
    mp = problem.make_map((30, 30, 20), density=0.04, seed=2)
    sdf = oracle_mod.Sdf(mp.origin, mp.resolution, mp.grid)
    sdf.build_from_occupancy(mp.occupancy)
    b = problem.make_trajectories(6, 4, mp, seed=3)
    prm = oracle_mod.make_params()
    c, g, _ = oracle_mod.eval_batch(b.T, b.Df, b.x, sdf, prm, nthreads=2)
    for i in range(6):
        ci, gi = oracle_mod.cost_grad(b.T[i], b.Df[i], b.x[i], sdf, prm)
        assert ci == c[i] and np.array_equal(gi, g[i])
    # setup helpers of the product's generator agree with the oracle's setup restatement
    assert np.array_equal(b.T, np.stack([oracle_mod.segment_time(w) for w in b.waypoints]))
    Df0, Dp0 = oracle_mod.initial_d(b.waypoints[0])
    Dfp, Dpp = problem.initial_derivatives(b.waypoints[:1])
    assert np.array_equal(Df0, Dfp[0]) and np.array_equal(Dp0, Dpp[0])


def test_restatement_is_sanitizer_clean():
    """ASan + UBSan build of the C restatement over every entry point (CPU only)."""
    import os
    import subprocess
    here = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle")
    subprocess.check_call(["make", "-C", here, "-s", "_build/oracle_asan"])
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:halt_on_error=1", UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1")
    env.pop("LD_PRELOAD", None)
    out = subprocess.run([os.path.join(here, "_build", "oracle_asan")], capture_output=True, text=True, env=env, timeout=120)
    assert out.returncode == 0 and "oracle_asan: ok" in out.stdout, out.stdout + out.stderr
