"""GPU parity: the HIP path (through the C-ABI) against the oracle on the same
seeded inputs.  Tolerance: 1e-5 relative in fp64 (BASELINE.json north_star);
the fp32 path's looser bound is stated where it is tested."""
import numpy as np
import pytest

from grad_traj_optimization_amd import problem

pytestmark = pytest.mark.gpu

TOL64 = 1e-5


def rel_err(c, g, c_ref, g_ref):
    rc = np.max(np.abs(c - c_ref) / np.abs(c_ref))
    rg = np.max(np.max(np.abs(g - g_ref), axis=1) / np.max(np.abs(g_ref), axis=1))
    return rc, rg


@pytest.fixture(scope="module")
def scene(gtop, oracle_mod):
    mp = problem.make_map((60, 50, 30), density=0.03, seed=11)
    ctx = gtop.GtopContext(device=0)
    ctx.init_sdf_map(mp.map_size, mp.origin, mp.resolution)
    ctx.update_sdf_map(mp.obstacle_points())
    sdf = oracle_mod.Sdf.from_map_size(mp.origin, mp.resolution, mp.map_size)
    assert sdf.grid == tuple(mp.grid) == tuple(ctx.grid)
    sdf.build_from_occupancy(mp.occupancy)
    return mp, ctx, sdf


def test_esdf_bit_exact(scene):
    mp, ctx, sdf = scene
    d = ctx.get_sdf().reshape(-1)
    assert np.array_equal(d, sdf.dist)


@pytest.mark.parametrize("grid,kind", [
    ((40, 24, 16), "random"),        # packed 16-bit x sweep (nz % 4 == 0, ny*nz % 8 == 0)
    ((300, 8, 8), "one_end"),        # free space wider than 255 voxels: saturated values -> the 32-bit scan
    ((100, 16, 16), "low_x"),        # slabs without any obstacle (no finite value after the y sweep)
    ((64, 64, 12), "single"),        # one obstacle voxel
    ((17, 10, 6), "random"),         # nz % 4 != 0: one voxel per lane
    ((20, 5, 4), "random"),          # ny*nz % 8 != 0: 32-bit x sweep with four voxels per lane
    ((3, 2, 8), "single"),           # lines shorter than a slab block
    ((260, 260, 8), "corner"),       # in-plane distances past 255 voxels next to small ones
    ((4, 2056, 8), "random"),        # ny past the y sweep's in-LDS candidate list: esdf_rows_kernel's lists
    ((70, 130, 68), "random"),       # two 64-voxel chunks per column, several ballots per slab's column flags
    ((6, 10, 320), "random"),        # five chunks per column: the scalar-mask z sweep's upper variants
    ((4, 6, 520), "random"),         # columns past 512 voxels: the LDS-mask z sweep
    ((8, 12, 320), "floor"),         # obstacles at z = 0 only: z distances past 255 voxels saturate the 16-bit copy the
    ((4, 6, 520), "floor"),          # packed y sweep reads -> its 32-bit walk (scalar-mask and LDS-mask z sweeps)
    ((40, 300, 16), "one_end"),      # the packed y sweep's own saturation: free space wider than 255 voxels along y
])
def test_esdf_grid_shapes_bit_exact(gtop, grid, kind):
    """The exact EDT on grid shapes that pick each sweep variant, against scipy's exact transform."""
    from scipy import ndimage
    nx, ny, nz = grid
    rng = np.random.default_rng(nx * 1000 + ny * 10 + nz)
    occ = np.zeros(grid, dtype=np.uint8)
    if kind == "random":
        occ[rng.random(grid) < 0.03] = 1
        occ[0, 0, 0] = 1
    elif kind == "one_end":
        occ[0, 3, 4] = 1
        occ[2, 0, 1] = 1
    elif kind == "low_x":
        occ[:5][rng.random((5, ny, nz)) < 0.1] = 1
        occ[1, 2, 3] = 1
    elif kind == "single":
        occ[nx // 2, ny // 3, nz // 2] = 1
    elif kind == "floor":
        occ[:, :, 0][rng.random((nx, ny)) < 0.3] = 1
        occ[0, 0, 0] = 1
    elif kind == "corner":
        occ[:6, :6, :][rng.random((6, 6, nz)) < 0.3] = 1
        occ[0, 0, 0] = 1
    res = 0.2
    mp = problem.MapSpec(grid, res, np.array([-nx * res / 2, -ny * res / 2, 0.0]), occ)
    ctx = gtop.GtopContext(device=0)
    ctx.init_sdf_map(mp.map_size, mp.origin, res)
    assert tuple(ctx.grid) == grid
    ctx.update_sdf_map(mp.obstacle_points())
    d = ctx.get_sdf()
    ref = res * ndimage.distance_transform_edt(occ == 0)
    assert np.array_equal(d, ref)


def test_esdf_every_distance_the_packed_sweep_can_produce(gtop):
    """One obstacle in the corner of a 256^3 map: the squared distances are ALL sums of three squares of 0 .. 255 —
    every value below 2^16 the packed x sweep's own square root (esdf_sqrt_u16: the hardware estimate + Goldschmidt,
    without the library routine's rescaling) can ever be handed, and the saturated ones beyond for the 32-bit path.
    Bit for bit res * sqrt(n) with numpy's exactly rounded sqrt."""
    n = 256
    res = 0.2
    occ = np.zeros((n, n, n), dtype=np.uint8)
    occ[0, 0, 0] = 1
    mp = problem.MapSpec((n, n, n), res, np.array([-n * res / 2, -n * res / 2, 0.0]), occ)
    ctx = gtop.GtopContext(device=0)
    ctx.init_sdf_map(mp.map_size, mp.origin, res)
    assert tuple(ctx.grid) == (n, n, n)
    ctx.update_sdf_map(mp.obstacle_points())
    d = ctx.get_sdf().reshape(n, n, n)
    i = np.arange(n, dtype=np.float64)
    sq = i[:, None, None] ** 2 + i[None, :, None] ** 2 + i[None, None, :] ** 2
    assert len(np.unique(sq[sq < 65535])) > 50000          # (most integers are sums of three squares)
    assert np.array_equal(d, res * np.sqrt(sq))
    ctx.close()


@pytest.mark.parametrize("m", [2, 3, 6, 7, 10, 12, 17])
@pytest.mark.parametrize("kw", [dict(), dict(step=1), dict(wc=0.0), dict(ws=0.0), dict(ws=20.0, wc=1.0)])
def test_fp64_parity_host_api(scene, oracle_mod, m, kw):
    mp, ctx, sdf = scene
    b = problem.make_trajectories(24, m, mp, seed=100 + m)
    ctx.set_launch_geometry(0, 0)
    ctx.set_params(**kw)
    ctx.set_problem(b.T, b.Df)
    c, g = ctx.eval_batch(b.x)
    c_ref, g_ref, _ = oracle_mod.eval_batch(b.T, b.Df, b.x, sdf, oracle_mod.make_params(**kw))
    rc, rg = rel_err(c, g, c_ref, g_ref)
    assert rc <= TOL64 and rg <= TOL64, (rc, rg)


@pytest.mark.parametrize("m", [2, 3, 4, 5, 6, 7, 12, 13, 24, 25, 37])
@pytest.mark.parametrize("spl", [0, 3, 6, 10, 30])
@pytest.mark.parametrize("waves", [0, 1])
def test_fp64_parity_every_launch_geometry(scene, oracle_mod, gtop, m, spl, waves):
    """Samples per lane (ten or five lanes per segment, one or two trajectories per wavefront, two wavefronts per
    trajectory, chunks of 12 segments past 12; three lanes or one lane per segment with as many trajectories per
    wavefront as fit) only changes the work split.  Ten lanes per segment hold up to 6 segments per wavefront, 12 on
    two: refused beyond; three lanes per segment serve up to 10 segments, one lane up to 64."""
    mp, ctx, sdf = scene
    b = problem.make_trajectories(23, m, mp, seed=200 + m,    # odd: exercises a partial last pair / padding workgroups
                                  step_len=(0.5, 1.2) if m > 6 else (1.0, 2.0))
    kw = dict(ws=0.0)          # collision term alone: the part the geometry touches
    ctx.set_launch_geometry(waves, spl)
    ctx.set_params(**kw)
    ctx.set_problem(b.T, b.Df)
    try:
        if (spl == 3 and m > 12) or (spl == 10 and m > 10) or (spl == 30 and m > 64):
            with pytest.raises(gtop.GtopError) as e:
                ctx.eval_batch(b.x)
            assert e.value.code == 1
            return
        c, g = ctx.eval_batch(b.x)
    finally:
        ctx.set_launch_geometry(0, 0)
    c_ref, g_ref, _ = oracle_mod.eval_batch(b.T, b.Df, b.x, sdf, oracle_mod.make_params(**kw))
    rc, rg = rel_err(c, g, c_ref, g_ref)
    assert rc <= TOL64 and rg <= TOL64, (rc, rg)


def test_launch_geometry_values(gtop):
    """One kernel family: a workgroup is one wavefront; samples per lane 0, 3, 6, 10 or 30."""
    ctx = gtop.GtopContext(device=0)
    for waves, spl in ((0, 0), (1, 3), (1, 6), (0, 6), (0, 10), (0, 30)):
        ctx.set_launch_geometry(waves, spl)
    for waves, spl in ((2, 3), (4, 0), (0, 1), (0, 5), (0, 15), (0, 2), (-1, 0)):
        with pytest.raises(gtop.GtopError) as e:
            ctx.set_launch_geometry(waves, spl)
        assert e.value.code == 1


def test_parity_wide_index_field(gtop, oracle_mod):
    """A flat map with nx*ny >= 2^24 takes the 64-bit-index variant of the lookup
    (corner_loads<WIDE>); same arithmetic, so the same bound.  The field is
    synthetic (smooth + noise), handed over with gtop_set_sdf."""
    import types
    grid = (4097, 4097, 6)
    res = 0.2
    origin = np.array([-grid[0] * res / 2, -grid[1] * res / 2, 0.0])
    map_size = np.array(grid) * res
    rng = np.random.default_rng(5)
    dist = rng.uniform(0.0, 2.0, size=grid[0] * grid[1] * grid[2])
    ms = types.SimpleNamespace(origin=origin, map_size=map_size)
    b = problem.make_trajectories(96, 6, ms, seed=77, margin=0.15, step_len=(0.5, 1.5))
    ctx = gtop.GtopContext(device=0)
    ctx.set_sdf(dist, grid, origin, res)
    sdf = oracle_mod.Sdf(origin, res, grid, dist)
    ctx.set_params()
    ctx.set_problem(b.T, b.Df)
    for spl in (3, 6, 10, 30):
        ctx.set_launch_geometry(0, spl)
        c, g = ctx.eval_batch(b.x)
        c_ref, g_ref, _ = oracle_mod.eval_batch(b.T, b.Df, b.x, sdf, oracle_mod.make_params())
        rc, rg = rel_err(c, g, c_ref, g_ref)
        assert rc <= TOL64 and rg <= TOL64, (spl, rc, rg)
    # fp32 entry on the same field: the packed path's wide variant (bound as in test_gpu_api)
    import torch
    dev = torch.device("cuda:0")
    x = torch.tensor(b.x, dtype=torch.float32, device=dev)
    Df = torch.tensor(b.Df.reshape(-1, 18), dtype=torch.float32, device=dev)
    T = torch.tensor(b.T, dtype=torch.float32, device=dev)
    for spl in (3, 6):
        ctx.set_launch_geometry(0, spl)
        c32, g32 = ctx.eval_device(x, Df, T)
        torch.cuda.synchronize()
        rc, rg = rel_err(c32.cpu().numpy().astype(np.float64), g32.cpu().numpy().astype(np.float64), c_ref, g_ref)
        assert rc <= 2e-3 and rg <= 2e-3, (spl, rc, rg)


@pytest.mark.parametrize("m", [40, 90, 200])
def test_fp64_parity_long_trajectories(scene, oracle_mod, gtop, m):
    """Many segments: the wavefront walks them 12 at a time; m = 90 takes more than 64 KB of LDS per workgroup (opt-in
    dynamic LDS), m = 200 nearly all 160 KB; past that the request is refused."""
    mp, ctx, sdf = scene
    b = problem.make_trajectories(5, m, mp, seed=4000 + m, step_len=(0.3, 0.8))
    if m == 200:
        too_long = problem.make_trajectories(2, 240, mp, seed=4001, step_len=(0.3, 0.8))
        ctx.set_problem(too_long.T, too_long.Df)
        with pytest.raises(gtop.GtopError) as e:
            ctx.eval_batch(too_long.x)
        assert e.value.code == 1
    ctx.set_launch_geometry(0, 0)
    ctx.set_params()
    ctx.set_problem(b.T, b.Df)
    c, g = ctx.eval_batch(b.x)
    c_ref, g_ref, _ = oracle_mod.eval_batch(b.T, b.Df, b.x, sdf, oracle_mod.make_params())
    rc, rg = rel_err(c, g, c_ref, g_ref)
    assert rc <= TOL64 and rg <= TOL64, (rc, rg)
