"""Non-zero velocity and acceleration at BOTH ends in the fixed derivatives Df — rows of the form
[p_start, v_start, a_start, p_end, v_end, a_end] (include/gtop.h) as a kinodynamic front end or a replanning start
state produces them (setKinoPath, src/grad_traj_optimizer.cpp:35-65; startVel / startAcc, src/qp_generator.cpp:425-431)
— through every kernel body, both precisions, the optimizer's launch forms and the rendezvous layer, HIP against
the oracle.  (Every other parity test draws Df with zero end derivatives, as setPath leaves them.)"""
import threading

import numpy as np
import pytest

from grad_traj_optimization_amd import problem
from tests import scenes
from tests.test_optimizer import mma_serial

pytestmark = pytest.mark.gpu
TOL64, TOL32 = 1e-5, 2e-4


@pytest.fixture(scope="module")
def scene(gtop, oracle_mod):
    mp = problem.make_map((60, 50, 30), density=0.03, seed=11)
    ctx = gtop.GtopContext(device=0)
    ctx.init_sdf_map(mp.map_size, mp.origin, mp.resolution)
    ctx.update_sdf_map(mp.obstacle_points())
    sdf = oracle_mod.Sdf.from_map_size(mp.origin, mp.resolution, mp.map_size)
    sdf.build_from_occupancy(mp.occupancy)
    return mp, ctx, sdf


def _kino(B, m, mp, seed):
    b = problem.make_trajectories(B, m, mp, seed=seed, step_len=(0.5, 1.2) if m > 6 else (1.0, 2.0),
                                  boundary="random")
    assert np.abs(b.Df[:, :, [1, 2, 4, 5]]).min() > 0.0
    return b


def _run(ctx, b, waves, spl, dtype, **params):
    import torch
    td = torch.float64 if dtype == "f64" else torch.float32
    dev = torch.device("cuda:0")
    x = torch.tensor(b.x, dtype=td, device=dev)
    Df = torch.tensor(b.Df.reshape(-1, 18), dtype=td, device=dev)
    T = torch.tensor(b.T, dtype=td, device=dev)
    try:
        ctx.set_params(**params)
        ctx.set_launch_geometry(waves, spl)
        c, g = ctx.eval_device(x, Df, T)
        torch.cuda.synchronize()
    finally:
        ctx.set_launch_geometry(0, 0)
        ctx.set_params()
    return c.double().cpu().numpy(), g.double().cpu().numpy()


def _check(oracle_mod, sdf, b, c, g, tol, **params):
    B = b.x.shape[0]
    idx = np.arange(B) if B <= 64 else np.r_[0:40, B - 40:B]
    c_ref, g_ref, _ = oracle_mod.eval_batch(b.T[idx], b.Df[idx], b.x[idx], sdf, oracle_mod.make_params(**params),
                                            nthreads=8)
    rc, rg = scenes.rel_err(c[idx], g[idx], c_ref, g_ref)
    assert rc <= tol and rg <= tol, (rc, rg)
    assert np.isfinite(c).all() and np.isfinite(g).all()


@pytest.mark.parametrize("dtype", ["f64", "f32"])
@pytest.mark.parametrize("spl,m", [(3, 2), (3, 3), (3, 5), (3, 6), (6, 5), (6, 6), (6, 7), (6, 9), (6, 11), (6, 12)])
@pytest.mark.parametrize("B", [23, 3101])
def test_kino_rows_every_wave_instantiation(scene, oracle_mod, dtype, spl, m, B):
    """The instantiations of tests/test_gpu_wave.py::test_every_instantiation (one trajectory per wavefront at 10 or
    5 lanes per segment, two per wavefront, latency and three-wavefront variants, packed fp32)."""
    mp, ctx, sdf = scene
    b = _kino(B, m, mp, 1600 + 13 * m + spl)
    c, g = _run(ctx, b, 1, spl, dtype)
    _check(oracle_mod, sdf, b, c, g, TOL64 if dtype == "f64" else TOL32)


@pytest.mark.parametrize("dtype", ["f64", "f32"])
@pytest.mark.parametrize("waves,spl,m,B", [(0, 6, 3, 23), (0, 6, 4, 200), (1, 6, 13, 23), (0, 0, 13, 300), (0, 0, 20, 5000),
                                           (0, 0, 30, 64), (0, 0, 6, 1), (0, 0, 9, 100), (0, 0, 12, 4100)])
def test_kino_rows_other_geometries(scene, oracle_mod, dtype, waves, spl, m, B):
    """Two short trajectories per wavefront, trajectories past 12 segments (walked 12 segments at a time), and what
    the auto rule picks at the batch sizes where it changes its mind."""
    mp, ctx, sdf = scene
    b = _kino(B, m, mp, 1700 + 7 * m + spl)
    c, g = _run(ctx, b, waves, spl, dtype)
    _check(oracle_mod, sdf, b, c, g, TOL64 if dtype == "f64" else TOL32)


@pytest.mark.parametrize("spl,m,B", [(0, 6, 37), (3, 6, 600), (3, 4, 3200), (6, 6, 600), (6, 12, 200), (0, 13, 64), (0, 27, 40),
                                     (0, 6, 9000)])
def test_kino_rows_dyn_feasibility(scene, oracle_mod, spl, m, B):
    """enable_dyn (the block commented out at grad_traj_optimizer.cpp:383-407) with such rows: the boundary
    velocities and accelerations enter its |v|, |a| penalties directly.  Every DYN body: ten and five lanes per
    segment, two trajectories per wavefront, 7 .. 12 and more than 12 segments."""
    mp, ctx, sdf = scene
    b = _kino(B, m, mp, 1800 + m + spl)
    p = dict(enable_dyn=1, alpha_v=2.0, r_v=4.0, alpha_a=1.5, r_a=15.0, step=2)
    c, g = _run(ctx, b, 0, spl, "f64", **p)
    _check(oracle_mod, sdf, b, c, g, TOL64, **p)
    if m <= 12 and B <= 600:          # the fp32 DYN bodies on rows of at least 0.25 s per segment (see test_gpu_api)
        keep = np.flatnonzero(b.T.min(axis=1) >= 0.25)
        bb = problem.permute(b, keep)
        c, g = _run(ctx, bb, 0, spl, "f32", **p)
        _check(oracle_mod, sdf, bb, c, g, TOL32, **p)


@pytest.mark.parametrize("kw", [dict(wc=0.0), dict(step=1), dict(ws=20.0, wc=1.0)])
def test_kino_rows_parameter_sets(scene, oracle_mod, kw):
    """|wc| < 1e-4 leaves the jerk term alone (:346) — where the boundary derivatives weigh most."""
    mp, ctx, sdf = scene
    for spl, m in ((3, 6), (6, 12)):
        b = _kino(37, m, mp, 1900 + m)
        c, g = _run(ctx, b, 1, spl, "f64", **kw)
        _check(oracle_mod, sdf, b, c, g, TOL64, **kw)


@pytest.mark.parametrize("m,evals", [(6, 25), (9, 20), (12, 15), (13, 12)])
def test_kino_rows_optimizer_launch_forms(scene, oracle_mod, gtop, m, evals):
    """The whole loop in one launch (the optimizer reads Df from LDS there), one launch per iteration, and the
    separate update launch: bit-identical at one pinned body, and each trajectory follows the serial CCSA-MMA twin
    driven by the oracle."""
    mp, ctx, sdf = scene
    B = 12
    b = _kino(B, m, mp, 2000 + m)
    lb, ub = gtop.GtopContext.default_bounds(b.waypoints)
    ctx.set_params()
    ctx.set_problem(b.T, b.Df)
    res = {}
    try:
        ctx.set_launch_geometry(0, 3 if m <= 6 else 6)
        for mode in (2, 1, 0):
            ctx.set_optimizer_fusion(mode)
            res[mode] = ctx.optimize_batch(b.x, lb, ub, evals)
    finally:
        ctx.set_optimizer_fusion(2)
        ctx.set_launch_geometry(0, 0)
    for mode in (1, 0):
        assert np.array_equal(res[mode][0], res[2][0]) and np.array_equal(res[mode][1], res[2][1])
    xs, costs = ctx.optimize_batch(b.x, lb, ub, evals)              # the auto rule's own choice
    prm = oracle_mod.make_params()
    c0, _, _ = oracle_mod.eval_batch(b.T, b.Df, b.x, sdf, prm)
    for i in range(B):
        gen = oracle_mod.generator(b.T[i])

        def f(x, i=i, gen=gen):
            return oracle_mod.cost_grad(b.T[i], b.Df[i], x, sdf, prm, L=gen["L"], R=gen["R"])
        x_ref, f_ref, _ = mma_serial(f, b.x[i], lb[i], ub[i], evals)
        for xo, co in ((xs, costs), res[2]):
            assert abs(co[i] - f_ref) <= 1e-6 * abs(f_ref), (i, co[i], f_ref)
            assert np.max(np.abs(xo[i] - x_ref)) <= 1e-6 * max(1.0, np.max(np.abs(x_ref)))
        assert costs[i] < c0[i]


def test_kino_rows_through_the_rendezvous_layer(scene, oracle_mod, gtop):
    """Serial callers on such rows meeting in shared launches (gtop_cost_nlopt_shared) see the rows of a plain batch
    evaluation, which match the oracle."""
    mp, ctx, sdf = scene
    n_callers, m = 8, 6
    b = _kino(n_callers, m, mp, 2100)
    ctx.set_params()
    ctx.set_problem(b.T, b.Df)
    c_ref, g_ref = ctx.eval_batch(b.x)
    rdv = gtop.Rendezvous(ctx, n_callers, m)
    got, errors = {}, []

    def worker(i):
        try:
            for k in range(2 + i % 3):
                got[(i, k)] = rdv.cost(i, b.x[i] + 0.01 * k)
        except Exception as e:      # noqa: BLE001 — reported below; the slot must leave either way
            errors.append(e)
        finally:
            rdv.leave(i)

    th = [threading.Thread(target=worker, args=(i,), daemon=True) for i in range(n_callers)]
    for t in th:
        t.start()
    for t in th:
        t.join(timeout=120)
        assert not t.is_alive()
    assert not errors, errors
    for i in range(n_callers):
        assert got[(i, 0)][0] == c_ref[i] and np.array_equal(got[(i, 0)][1], g_ref[i])
    c_or, g_or, _ = oracle_mod.eval_batch(b.T, b.Df, b.x, sdf, oracle_mod.make_params())
    rc, rg = scenes.rel_err(c_ref, g_ref, c_or, g_or)
    assert rc <= TOL64 and rg <= TOL64, (rc, rg)
    # the NLopt-shaped single-problem entry on row 0
    c1, g1 = ctx.cost_nlopt(b.x[0])
    rc, rg = scenes.rel_err(c1, g1, c_or[0], g_or[0])
    assert rc <= TOL64 and rg <= TOL64
