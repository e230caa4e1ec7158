"""GPU parity of the static-field + moving-boxes distance query (SURVEY §8f row f4,
src/edt_environment.cpp:26-122) against the oracle's restatement.  PARITY UNPINNED
against the reference itself: that file is outside its build (see oracle/gtop_oracle.c)."""
import numpy as np
import pytest

from grad_traj_optimization_amd import problem

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def scene(gtop, oracle_mod):
    mp = problem.make_map((60, 50, 30), density=0.03, seed=11)
    ctx = gtop.GtopContext(device=0)
    ctx.init_sdf_map(mp.map_size, mp.origin, mp.resolution)
    ctx.update_sdf_map(mp.obstacle_points())
    sdf = oracle_mod.Sdf.from_map_size(mp.origin, mp.resolution, mp.map_size)
    sdf.build_from_occupancy(mp.occupancy)
    return mp, ctx, sdf


def _queries(mp, n, seed):
    rng = np.random.default_rng(seed)
    lo, hi = mp.origin - 0.3, mp.origin + mp.map_size + 0.3     # some land outside the map
    pos = rng.uniform(lo, hi, size=(n, 3))
    pos[:8] = mp.origin + 0.05                                   # base index -1: clamped corners
    pos[8:16] = mp.origin + mp.map_size - 0.05
    time = rng.uniform(0.0, 3.0, size=n)
    time[::5] = -1.0                                             # static only
    return pos, time


def test_static_only_equals_trilinear_query(scene, oracle_mod):
    mp, ctx, sdf = scene
    pos, _ = _queries(mp, 512, 1)
    ctx.set_moving_boxes(np.zeros((0, 3)), np.zeros((0, 3)), np.zeros((0, 3)))
    d, g = ctx.edt_query(pos, -1.0)
    ref = [sdf.query(p) for p in pos]
    assert np.allclose(d, [r[0] for r in ref], rtol=1e-13, atol=1e-13)
    assert np.allclose(g, [r[1] for r in ref], rtol=1e-12, atol=1e-12)
    # no boxes, time >= 0: the box distance is the 1e7 sentinel, i.e. static again
    d2, g2 = ctx.edt_query(pos, 1.0)
    assert np.array_equal(d, d2) and np.array_equal(g, g2)


@pytest.mark.parametrize("nbox", [1, 7, 200])
def test_moving_boxes_parity(scene, oracle_mod, nbox):
    mp, ctx, sdf = scene
    rng = np.random.default_rng(100 + nbox)
    p0 = rng.uniform(mp.origin, mp.origin + mp.map_size, size=(nbox, 3))
    vel = rng.uniform(-1.0, 1.0, size=(nbox, 3))
    scale = rng.uniform(0.3, 1.5, size=(nbox, 3))
    pos, time = _queries(mp, 2048, 2 + nbox)
    ctx.set_moving_boxes(p0, vel, scale)
    d, g = ctx.edt_query(pos, time)
    d_ref, g_ref = sdf.edt_query(pos, time, p0, vel, scale)
    assert (d_ref[time >= 0] < sdf.edt_query(pos, -1.0, p0, vel, scale)[0][time >= 0] - 1e-9).any()   # boxes matter
    assert np.allclose(d, d_ref, rtol=1e-12, atol=1e-12)
    assert np.allclose(g, g_ref, rtol=1e-10, atol=1e-10)
    assert ((d == -1.0) == (d_ref == -1.0)).all() and (d == -1.0).any()


def test_device_entry_matches_host_entry(scene):
    import torch
    mp, ctx, sdf = scene
    rng = np.random.default_rng(5)
    p0 = rng.uniform(mp.origin, mp.origin + mp.map_size, size=(3, 3))
    ctx.set_moving_boxes(p0, np.ones((3, 3)) * 0.2, np.ones((3, 3)))
    pos, time = _queries(mp, 1000, 9)
    d, g = ctx.edt_query(pos, time)
    dev = torch.device("cuda:0")
    dd, gd = ctx.edt_query_device(torch.tensor(pos, device=dev), torch.tensor(time, device=dev))
    torch.cuda.synchronize()
    assert np.array_equal(dd.cpu().numpy(), d) and np.array_equal(gd.cpu().numpy(), g)


@pytest.mark.parametrize("nbox", [0, 5, 150])
def test_coarse_query_parity(scene, oracle_mod, nbox):
    """EDTEnvironment::evaluateCoarseEDT (src/edt_environment.cpp:124-136): the voxel's own distance
    (SDFMap::getDistance(pos), sdf_map.cpp:155-164) min'ed with the box distance from the position itself."""
    mp, ctx, sdf = scene
    rng = np.random.default_rng(300 + nbox)
    p0 = rng.uniform(mp.origin, mp.origin + mp.map_size, size=(nbox, 3))
    vel = rng.uniform(-1.0, 1.0, size=(nbox, 3))
    scale = rng.uniform(0.3, 1.5, size=(nbox, 3))
    pos, time = _queries(mp, 1500, 40 + nbox)       # 1500: a partial last workgroup
    ctx.set_moving_boxes(p0, vel, scale)
    d = ctx.edt_coarse_query(pos, time)
    d_ref = sdf.edt_coarse(pos, time, p0, vel, scale)
    assert np.allclose(d, d_ref, rtol=1e-13, atol=1e-13)
    assert ((d == -1.0) == (d_ref == -1.0)).all() and (d == -1.0).any()
    # static only = the voxel value, bit for bit
    ds = ctx.edt_coarse_query(pos, -1.0)
    inside = ds != -1.0
    idx = np.floor((pos[inside] - mp.origin) / mp.resolution).astype(int)
    assert np.array_equal(ds[inside], sdf.dist.reshape(sdf.grid)[idx[:, 0], idx[:, 1], idx[:, 2]])
