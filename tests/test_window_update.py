"""The reference's LOCAL map update — resetBuffer(min, max) + setOccupancy + [setUpdateRange] + updateESDF3d over the
window (src/sdf_map.cpp:28-53, :80-99, :244-264, :310-368; used by compare2.cpp:147-152) — restated in the oracle and,
on the GPU, as gtop_update_sdf_map_window.  CPU part: the oracle's restatement has the semantics the reference's code
has (checked against a brute-force minimum); GPU part: bit-exact against the oracle on random windows."""
import numpy as np
import pytest

from grad_traj_optimization_amd import problem


def _brute(occ, lo, hi, res, old):
    """What updateESDF3d leaves in the window: the exact Euclidean distance to the nearest occupied voxel INSIDE the
    window (the sweeps never look outside it), min'ed with the previous distance; elsewhere `old` untouched."""
    out = old.copy()
    sub = occ[lo[0]:hi[0] + 1, lo[1]:hi[1] + 1, lo[2]:hi[2] + 1]
    pts = np.argwhere(sub == 1)
    if len(pts) == 0:
        return out
    gx, gy, gz = np.meshgrid(*(np.arange(n) for n in sub.shape), indexing="ij")
    q = np.stack([gx, gy, gz], axis=-1).reshape(-1, 3)
    d2 = np.min(((q[:, None, :] - pts[None, :, :]) ** 2).sum(axis=2), axis=1).reshape(sub.shape)
    w = out[lo[0]:hi[0] + 1, lo[1]:hi[1] + 1, lo[2]:hi[2] + 1]
    w[...] = np.minimum(res * np.sqrt(d2.astype(np.float64)), w)
    return out


def _random_case(rng):
    grid = tuple(int(v) for v in rng.integers(6, 22, size=3))
    res = float(rng.choice([0.1, 0.2, 0.25]))
    origin = np.array([-grid[0] * res / 2, -grid[1] * res / 2, 0.0]) + rng.uniform(-1, 1, 3) * (rng.random() < 0.5)
    map_size = (np.array(grid) - rng.choice([0.5, 0.25, 0.75])) * res     # grid = ceil(size / res) (sdf_map.cpp:9); max_range inside the last voxel
    return grid, res, origin, map_size


def test_oracle_window_update_matches_a_brute_force_minimum(oracle_mod):
    rng = np.random.default_rng(7)
    for it in range(30):
        grid, res, origin, map_size = _random_case(rng)
        sdf = oracle_mod.Sdf.from_map_size(origin, res, map_size)
        assert sdf.grid == grid
        occ = np.zeros(int(np.prod(grid)))
        # a first, whole-map build from some points, then two local updates with new points
        centres = lambda idx: (idx + 0.5) * res + origin      # noqa: E731
        first = centres(np.argwhere(rng.random(grid) < 0.03))
        occ[:] = sdf.build_from_points(first)                                         # updateSDFMap: the whole map
        lo_all, hi_all = sdf.window_ids(origin - 1.0, origin + map_size + 1.0)        # (clamped to the map)
        assert np.all(lo_all == 0)
        # resetBuffer's own box can stop one voxel short of the grid (posToIndex(max_range - res/2) with a map size that
        # is not a multiple of the resolution): the reference's quirk, kept
        assert np.all((hi_all == np.array(grid) - 1) | (hi_all == np.array(grid) - 2))
        if np.all(hi_all == np.array(grid) - 1):
            again = oracle_mod.Sdf.from_map_size(origin, res, map_size)
            occ2 = np.zeros_like(occ)
            again.update_window(occ2, origin - 1.0, origin + map_size + 1.0, first)
            assert np.array_equal(again.dist, sdf.dist) and np.array_equal(occ2, occ)   # a window = the map: the full build
        for _ in range(2):
            a = origin + rng.uniform(0.0, 0.6, 3) * map_size
            b = a + rng.uniform(0.15, 0.6, 3) * map_size
            new_pts = rng.uniform(a - 0.3, b + 0.3, size=(int(rng.integers(0, 12)), 3))   # some fall outside the window
            before_dist, before_occ = sdf.dist.copy(), occ.copy()
            lo, hi = sdf.update_window(occ, a, b, new_pts)
            assert np.all(lo >= 0) and np.all(hi < np.array(grid))
            # expected occupancy: cleared in the window, then the in-map points marked wherever they fall
            exp_occ = before_occ.reshape(grid).copy()
            exp_occ[lo[0]:hi[0] + 1, lo[1]:hi[1] + 1, lo[2]:hi[2] + 1] = 0
            for p in new_pts:
                if np.all(p >= origin + 1e-4) and np.all(p <= origin + map_size - 1e-4):
                    i = np.floor((p - origin) / res + 0).astype(int)
                    i = np.floor((p - origin) * (1.0 / res)).astype(int)
                    exp_occ[tuple(i)] = 1
            assert np.array_equal(occ.reshape(grid), exp_occ)
            old = before_dist.reshape(grid).copy()
            old[lo[0]:hi[0] + 1, lo[1]:hi[1] + 1, lo[2]:hi[2] + 1] = 10000.0
            assert np.array_equal(sdf.dist.reshape(grid), _brute(exp_occ, lo, hi, res, old)), it


@pytest.mark.gpu
def test_gpu_window_update_is_bit_exact(gtop, oracle_mod):
    """gtop_update_sdf_map_window against the oracle's restatement on random maps and random windows, a sequence of
    updates on one context (occupancy and distances persist), the field read back bit for bit after every one — and
    the lookups (which read the corner records, rebuilt for the window only) agree with the oracle's on that field."""
    import torch
    rng = np.random.default_rng(11)
    dev = torch.device("cuda:0")
    for it in range(12):
        grid = tuple(int(v) for v in rng.integers(10, 48, size=3))
        res = float(rng.choice([0.1, 0.2, 0.25]))
        origin = np.array([-grid[0] * res / 2, -grid[1] * res / 2, 0.0])
        map_size = (np.array(grid) - 0.5) * res
        sdf = oracle_mod.Sdf.from_map_size(origin, res, map_size)
        assert sdf.grid == grid
        occ = np.zeros(int(np.prod(grid)))
        ctx = gtop.GtopContext(device=0)
        ctx.init_sdf_map(map_size, origin, res)
        assert tuple(ctx.grid) == grid
        first = (np.argwhere(rng.random(grid) < 0.02) + 0.5) * res + origin
        ctx.update_sdf_map(first)
        sdf.build_from_points(first)
        occ[:] = 0
        for p in first:
            occ[np.ravel_multi_index(tuple(np.floor((p - origin) * (1.0 / res)).astype(int)), grid)] = 1
        assert np.array_equal(ctx.get_sdf().reshape(-1), sdf.dist)
        if it % 3 == 0:
            ctx.eval_device(*[torch.zeros(1, n, dtype=torch.float32, device=dev) + 0.5 for n in (9, 18, 2)])   # fp32 records in use
        for step in range(4):
            a = origin + rng.uniform(-0.1, 0.6, 3) * map_size
            b = a + rng.uniform(0.1, 0.7, 3) * map_size
            if step == 3:
                a, b = origin - 1, origin + map_size + 1              # the whole map: the whole-grid builder
            pts = rng.uniform(a - 0.3, b + 0.3, size=(int(rng.integers(0, 40)), 3))
            if step % 2:
                ctx.update_sdf_map_window_device(a, b, torch.tensor(pts.reshape(-1, 3), device=dev))
                torch.cuda.synchronize()
            else:
                ctx.update_sdf_map_window(a, b, pts)
            sdf.update_window(occ, a, b, pts)
            assert np.array_equal(ctx.get_sdf().reshape(-1), sdf.dist), (it, step)
            # lookups through the corner records
            m = 3
            ms = type("M", (), {"origin": origin, "map_size": map_size})
            bt = problem.make_trajectories(16, m, ms, seed=100 * it + step, margin=0.05, step_len=(0.2, 0.6))
            ctx.set_params()
            ctx.set_problem(bt.T, bt.Df)
            c, g = ctx.eval_batch(bt.x)
            c_ref, g_ref, _ = oracle_mod.eval_batch(bt.T, bt.Df, bt.x, sdf, oracle_mod.make_params())
            assert np.max(np.abs(c - c_ref) / np.abs(c_ref)) <= 1e-9
            assert np.max(np.abs(g - g_ref)) <= 1e-9 * np.max(np.abs(g_ref))
            if it % 3 == 0:
                c32, g32 = ctx.eval_device(*[torch.tensor(v, dtype=torch.float32, device=dev) for v in (bt.x, bt.Df.reshape(-1, 18), bt.T)])
                torch.cuda.synchronize()
                assert np.max(np.abs(c32.double().cpu().numpy() - c_ref) / np.abs(c_ref)) <= 2e-4
        ctx.close()


@pytest.mark.gpu
def test_window_update_replays_from_a_hip_graph(gtop, oracle_mod):
    """The per-frame loop of compare2.cpp:147-152 as ONE graph: window update (points read from a fixed HBM buffer) +
    the evaluation behind it, captured once — after a first eager update of a SMALLER window has allocated the scratch,
    which serves any window — and replayed with new points in the buffer: every replay leaves the field and the costs
    of an eager context fed the same frames, bit for bit (fp64 and, the records of both precisions following, fp32)."""
    import torch
    rng = np.random.default_rng(5)
    dev = torch.device("cuda:0")
    grid, res = (40, 36, 20), 0.2
    origin = np.array([-4.0, -3.6, 0.0])
    map_size = (np.array(grid) - 0.5) * res
    ms = type("M", (), {"origin": origin, "map_size": map_size})
    bt = problem.make_trajectories(64, 4, ms, seed=9, margin=0.05, step_len=(0.3, 0.8))
    first = (np.argwhere(rng.random(grid) < 0.02) + 0.5) * res + origin
    a, b = origin + np.array([1.0, 1.2, 0.0]), origin + np.array([6.0, 5.5, 3.0])
    frames = [rng.uniform(a - 0.2, b + 0.2, size=(50, 3)) for _ in range(3)]
    ctxs = []
    for _ in range(2):
        c = gtop.GtopContext(device=0)
        c.init_sdf_map(map_size, origin, res)
        c.update_sdf_map(first)
        c.set_params()
        c.update_sdf_map_window(a + 1.0, a + 2.0, frames[0][:3])          # allocates the window scratch (uncaptured)
        ctxs.append(c)
    eager, graphed = ctxs
    pts = torch.zeros(50, 3, dtype=torch.float64, device=dev)
    ins = {td: [torch.tensor(v, dtype=td, device=dev) for v in (bt.x, bt.Df.reshape(-1, 18), bt.T)]
           for td in (torch.float64, torch.float32)}
    graphed.eval_device(*ins[torch.float32])                               # fp32 records in use: the window rebuilds both
    eager.eval_device(*ins[torch.float32])
    outs = {td: (torch.zeros(64, dtype=td, device=dev), torch.zeros(64, 27, dtype=td, device=dev)) for td in ins}
    torch.cuda.synchronize()
    gph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gph):
        graphed.update_sdf_map_window_device(a, b, pts)
        for td in ins:
            graphed.eval_device(*ins[td], *outs[td])
    for f in frames:
        pts.copy_(torch.tensor(f, device=dev))
        gph.replay()
        torch.cuda.synchronize()
        eager.update_sdf_map_window(a, b, f)
        assert np.array_equal(graphed.get_sdf(), eager.get_sdf())
        for td in ins:
            c_e, g_e = eager.eval_device(*ins[td])
            torch.cuda.synchronize()
            assert torch.equal(c_e, outs[td][0]) and torch.equal(g_e, outs[td][1])
    for c in ctxs:
        c.close()
