"""The batched CCSA-MMA driver on the device against a
serial restatement of the same algorithm driven by the oracle.

PARITY UNPINNED for the optimizer trajectory itself: the reference delegates
to NLopt 2.5.0 (LD_MMA), which is absent here.  What is checked: the device
lock-step driver follows, per trajectory, exactly the serial algorithm of
csrc/mma.hpp (restated below in numpy) when both are fed the same callback —
same accept/reject decisions, same iterates up to fp noise."""
import numpy as np
import pytest

from grad_traj_optimization_amd import problem
from tests import scenes

pytestmark = pytest.mark.gpu


def mma_serial(f, x0, lb, ub, max_evals, ftol_rel=0.0, xtol_rel=0.0, full=False):
    """numpy twin of csrc/mma.hpp::mma_minimize: evaluation-count stop, ftol_rel, xtol_rel (:127-137)."""
    n = x0.size
    sigma = np.where(np.isinf(lb) | np.isinf(ub), 1.0, 0.5 * (ub - lb))
    x = np.clip(x0, lb, ub).copy()
    rho = 1.0
    fcur, dfdx = f(x)
    nev = 1
    minf = fcur
    xcur = x.copy()
    xprev = x.copy()
    xprevprev = x.copy()
    k = 0
    trace = [minf]

    def step(x, dfdx, sigma, rho, fval):
        s2 = sigma * sigma
        u = dfdx * s2
        v = np.abs(dfdx) * sigma + 0.5 * rho
        q = u / (v * sigma)
        dx = (u / v) / (-1.0 - np.sqrt(np.abs(1.0 - q * q)))
        xc = np.clip(x + dx, lb, ub)
        xc = np.clip(xc, x - 0.9 * sigma, x + 0.9 * sigma)
        dx = xc - x
        dx2 = dx * dx
        den = 1.0 / (s2 - dx2)
        g = fval + np.sum((dfdx * (s2 * dx) + (np.abs(dfdx) * sigma + 0.5 * rho) * dx2) * den)
        w = np.sum(0.5 * dx2 * den)
        return xc, g, w

    code = 5
    capped = False
    while nev < max_evals:
        fprev = fcur
        k += 1
        if k > 1:
            xprevprev = xprev.copy()
        xprev = xcur.copy()
        inner_done = False
        while nev < max_evals:
            xcur, gval, wval = step(x, dfdx, sigma, rho, minf)
            fcur, dcur = f(xcur)
            nev += 1
            inner_done = gval >= fcur
            if fcur < minf:
                minf, x, dfdx = fcur, xcur.copy(), dcur
            trace.append(minf)
            if nev >= max_evals:                             # NLopt looks at the evaluation limit first (mma.c)
                capped = True
                break
            if inner_done:
                break
            if fcur > gval:
                rho = min(10 * rho, 1.1 * (rho + (fcur - gval) / wval))
        if capped:
            break

        def relstop(old, new, tol):                          # NLopt's stop.c
            return (np.abs(new - old) < tol * (np.abs(new) + np.abs(old)) * 0.5) | ((tol > 0) & (new == old))
        if relstop(fprev, fcur, ftol_rel):
            code = 3
        if xtol_rel > 0 and np.all(relstop(xprev, xcur, xtol_rel)):
            code = 4                                         # x after f: its verdict stands when both hold
        if code != 5:
            break
        rho = max(0.1 * rho, 1e-5)
        if k > 1:
            dx2 = (xcur - xprev) * (xprev - xprevprev)
            gam = np.where(dx2 < 0, 0.7, np.where(dx2 > 0, 1.2, 1.0))
            sigma = np.clip(sigma * gam, 0.01 * (ub - lb), 10 * (ub - lb))
    if full:
        return x, minf, trace, nev, code
    return x, minf, trace


@pytest.fixture(scope="module")
def scene(gtop, oracle_mod):
    mp = problem.make_map((60, 50, 30), density=0.03, seed=11)
    ctx = gtop.GtopContext(device=0)
    ctx.init_sdf_map(mp.map_size, mp.origin, mp.resolution)
    ctx.update_sdf_map(mp.obstacle_points())
    sdf = oracle_mod.Sdf.from_map_size(mp.origin, mp.resolution, mp.map_size)
    sdf.build_from_occupancy(mp.occupancy)
    return mp, ctx, sdf


def test_default_bounds(gtop):
    wp = np.arange(2 * 4 * 3, dtype=np.float64).reshape(2, 4, 3)   # B=2, m=3
    lb, ub = gtop.GtopContext.default_bounds(wp, bos=3.0, vos=8.0, aos=10.0)
    assert lb.shape == (2, 18)
    # x layout i + axis*num_dp, i%3: 0 pos (waypoint i/3+1), 1 vel, 2 acc   (:151-179)
    for b in range(2):
        for a in range(3):
            for i in range(6):
                j = i + a * 6
                if i % 3 == 0:
                    assert lb[b, j] == wp[b, i // 3 + 1, a] - 3.0 and ub[b, j] == wp[b, i // 3 + 1, a] + 3.0
                elif i % 3 == 1:
                    assert (lb[b, j], ub[b, j]) == (-8.0, 8.0)
                else:
                    assert (lb[b, j], ub[b, j]) == (-10.0, 10.0)


@pytest.mark.parametrize("m,evals,kw", [(6, 25, {}), (3, 40, {}),
                                        (9, 20, {}), (12, 15, {}),      # the five-lanes-per-segment loop (7..12 segments)
                                        (13, 12, {}), (30, 14, {}),     # past it: the loop walks the segments 12 at a time
                                        (6, 20, dict(enable_dyn=1, alpha_v=2.0, alpha_a=1.5)),   # the loop's DYN bodies
                                        (10, 12, dict(enable_dyn=1, alpha_v=2.0, r_v=4.0, alpha_a=1.5, r_a=15.0)),
                                        (14, 10, dict(enable_dyn=1, alpha_v=2.0, r_v=4.0, alpha_a=1.5, r_a=15.0))])
def test_device_lockstep_mma_follows_the_serial_algorithm(scene, oracle_mod, gtop, m, evals, kw):
    mp, ctx, sdf = scene
    B = 12
    b = problem.make_trajectories(B, m, mp, seed=300 + m)
    lb, ub = gtop.GtopContext.default_bounds(b.waypoints)
    ctx.set_params(**kw)
    ctx.set_problem(b.T, b.Df)
    try:
        xs, costs = ctx.optimize_batch(b.x, lb, ub, evals)
    finally:
        ctx.set_params()
    prm = oracle_mod.make_params(**kw)
    c0, _, _ = oracle_mod.eval_batch(b.T, b.Df, b.x, sdf, prm)
    improved = 0
    for i in range(B):
        gen = oracle_mod.generator(b.T[i])

        def f(x, i=i, gen=gen):
            return oracle_mod.cost_grad(b.T[i], b.Df[i], x, sdf, prm, L=gen["L"], R=gen["R"])
        x_ref, f_ref, trace = mma_serial(f, b.x[i], lb[i], ub[i], evals)
        assert abs(costs[i] - f_ref) <= 1e-6 * abs(f_ref), (i, costs[i], f_ref)
        assert np.max(np.abs(xs[i] - x_ref)) <= 1e-6 * max(1.0, np.max(np.abs(x_ref)))
        assert costs[i] <= c0[i] * (1 + 1e-12)                     # never worse than the start
        improved += costs[i] < c0[i] * (1 - 1e-9)
        assert np.all(xs[i] >= lb[i] - 1e-12) and np.all(xs[i] <= ub[i] + 1e-12)
        # the returned cost is the callback's value at the returned point
        c_chk, _ = oracle_mod.cost_grad(b.T[i], b.Df[i], xs[i], sdf, prm, L=gen["L"], R=gen["R"])
        assert abs(c_chk - costs[i]) <= 1e-5 * abs(c_chk)
    # it optimises (a long trajectory may spend a handful of evaluations making its first model conservative)
    assert improved >= B // 2, improved


def test_optimize_device_api_and_determinism(scene, gtop):
    import torch
    mp, ctx, sdf = scene
    b = problem.make_trajectories(256, 6, mp, seed=77)
    lb, ub = gtop.GtopContext.default_bounds(b.waypoints)
    dev = torch.device("cuda:0")
    Df = torch.tensor(b.Df.reshape(-1, 18), device=dev)
    T = torch.tensor(b.T, device=dev)
    lbt, ubt = torch.tensor(lb, device=dev), torch.tensor(ub, device=dev)
    ctx.set_params()
    out = []
    for _ in range(2):
        x = torch.tensor(b.x, device=dev)
        x, c = ctx.optimize_device(x, Df, T, lbt, ubt, 20)
        torch.cuda.synchronize()
        out.append((x.clone(), c.clone()))
    assert torch.equal(out[0][0], out[1][0]) and torch.equal(out[0][1], out[1][1])   # bitwise repeatable
    ctx.set_launch_geometry(0, 3)                # the optimizer's geometry: same summation order
    c0, _ = ctx.eval_device(torch.tensor(b.x, device=dev), Df, T)
    c1, _ = ctx.eval_device(out[0][0], Df, T)
    torch.cuda.synchronize()
    ctx.set_launch_geometry(0, 0)
    # min_cost is the cost of the returned x (a plain evaluation may run another body than the optimizer's:
    # summation order only)
    assert torch.max(torch.abs(c1 - out[0][1]) / out[0][1]).item() <= 1e-12
    assert (c1 <= c0).all() and (c1 < 0.5 * c0).float().mean() > 0.9


@pytest.mark.parametrize("B,m,pin", [(64, 6, 3), (5000, 6, 0), (4097, 6, 6), (33, 4, 6), (300, 12, 0), (100, 17, 0)])
def test_fused_optimizer_step_equals_separate_launches(scene, gtop, B, m, pin):
    """Three ways to run the same loop — the whole loop in one launch (default), the MMA
    update as the evaluation kernel's epilogue with one launch per iteration, and the
    two-launch form — do the same arithmetic: bit-identical results at the same launch
    geometry.  pin 6 with up to 6 segments: TWO trajectories per wavefront, both states in LDS, the update run once per
    trajectory (odd batches: a last wavefront with one) — what the launch rule picks by itself for fp32 evaluations
    from 3 072 trajectories (test_optimizer_with_fp32_evaluations)."""
    import torch
    mp, ctx, sdf = scene
    b = problem.make_trajectories(B, m, mp, seed=900 + m)
    lb, ub = gtop.GtopContext.default_bounds(b.waypoints)
    dev = torch.device("cuda:0")
    Df = torch.tensor(b.Df.reshape(-1, 18), device=dev)
    T = torch.tensor(b.T, device=dev)
    lbt, ubt = torch.tensor(lb, device=dev), torch.tensor(ub, device=dev)
    ctx.set_params()
    res = []
    ctx.set_launch_geometry(0, pin)
    for mode in (2, 1, 0):
        ctx.set_optimizer_fusion(mode)
        x = torch.tensor(b.x, device=dev)
        x, c = ctx.optimize_device(x, Df, T, lbt, ubt, 15)
        torch.cuda.synchronize()
        res.append((x.clone(), c.clone()))
    ctx.set_optimizer_fusion(2)
    ctx.set_launch_geometry(0, 0)
    for r in res[1:]:
        assert torch.equal(res[0][0], r[0]) and torch.equal(res[0][1], r[1])


def test_two_step_schedule_batched(scene, oracle_mod, gtop):
    """The reference's schedule (example_click.cpp:163-164: optimizeTrajectory(OPT_FIRST_STEP)
    then OPT_SECOND_STEP): step 1 drops the smoothness weight (:412-415), step 2 restores
    it.  Batched: set step, optimise, set step, optimise from the first result."""
    mp, ctx, sdf = scene
    B, m = 48, 6
    b = problem.make_trajectories(B, m, mp, seed=555)
    lb, ub = gtop.GtopContext.default_bounds(b.waypoints)
    ctx.set_problem(b.T, b.Df)
    ctx.set_params(step=1)
    x1, c1 = ctx.optimize_batch(b.x, lb, ub, 20)
    p1 = oracle_mod.make_params(step=1)
    c_start1, _, _ = oracle_mod.eval_batch(b.T, b.Df, b.x, sdf, p1)
    c_end1, _, _ = oracle_mod.eval_batch(b.T, b.Df, x1, sdf, p1)
    assert np.allclose(c1, c_end1, rtol=1e-6) and (c_end1 <= c_start1 + 1e-12).all()
    ctx.set_params(step=2)
    x2, c2 = ctx.optimize_batch(x1, lb, ub, 20)
    p2 = oracle_mod.make_params(step=2)
    c_start2, _, _ = oracle_mod.eval_batch(b.T, b.Df, x1, sdf, p2)
    c_end2, _, _ = oracle_mod.eval_batch(b.T, b.Df, x2, sdf, p2)
    assert np.allclose(c2, c_end2, rtol=1e-6) and (c_end2 <= c_start2 + 1e-12).all()
    assert (c_end2 < c_start2).mean() > 0.9
    ctx.set_params()


@pytest.mark.parametrize("rule", [dict(ftol_rel=5e-2), dict(xtol_rel=0.2), dict(ftol_rel=2e-2, xtol_rel=0.1)])
def test_device_stop_rules_follow_the_serial_algorithm(scene, oracle_mod, gtop, rule):
    """ftol_rel / xtol_rel inside the one-launch loop (mma.hpp:35-39, :127-137; the reference's own rule is
    set_maxtime, grad_traj_optimizer.cpp:144-148): every trajectory stops at the evaluation the serial twin stops
    at, with its code and its iterate, and uses fewer evaluations than the cap.  Compared on the jerk term alone
    (wc = 0: a convex quadratic, on which the two runs cannot drift apart through rounding — with the collision
    term an accept/reject decision near a tie flips after ~50 evaluations and moves the stop by an evaluation or
    two); the three launch forms are compared with the full cost."""
    mp, ctx, sdf = scene
    B, m, cap = 16, 6, 80
    b = problem.make_trajectories(B, m, mp, seed=700)
    lb, ub = gtop.GtopContext.default_bounds(b.waypoints)
    ctx.set_problem(b.T, b.Df)
    # (a) full cost: same decisions whatever the launch form (one body pinned: same bits)
    ctx.set_params()
    for pin in (3, 6):                       # (6: two trajectories per wavefront, each stopping on its own)
        res = {}
        ctx.set_launch_geometry(0, pin)
        for mode in (2, 1, 0):               # whole loop in one launch; one launch per iteration; separate update launch
            ctx.set_optimizer_fusion(mode)
            res[mode] = ctx.optimize_batch_ex(b.x, lb, ub, cap, **rule)
        ctx.set_optimizer_fusion(2)
        ctx.set_launch_geometry(0, 0)
        for mode in (1, 0):
            for a, r in zip(res[mode], res[2]):
                assert np.array_equal(a, r)
        assert (res[2][3] != 5).sum() >= B // 2 and res[2][2][res[2][3] != 5].max() < cap      # the rules did fire
        if pin == 6:
            assert (res[2][2][0::2] != res[2][2][1::2]).any()      # wavefronts whose two trajectories part ways
    # (b) jerk term alone against the serial twin
    kw = dict(wc=0.0)
    prm = oracle_mod.make_params(**kw)
    for pin in (0, 6):
        ctx.set_params(**kw)
        ctx.set_launch_geometry(0, pin)
        try:
            xs, costs, nev, code = ctx.optimize_batch_ex(b.x, lb, ub, cap, **rule)
        finally:
            ctx.set_params()
            ctx.set_launch_geometry(0, 0)
        for i in range(B):
            gen = oracle_mod.generator(b.T[i])

            def f(x, i=i, gen=gen):
                return oracle_mod.cost_grad(b.T[i], b.Df[i], x, sdf, prm, L=gen["L"], R=gen["R"])
            x_ref, f_ref, _, nev_ref, code_ref = mma_serial(f, b.x[i], lb[i], ub[i], cap, full=True, **rule)
            assert (nev[i], code[i]) == (nev_ref, code_ref), (pin, i, nev[i], code[i], nev_ref, code_ref)
            assert abs(costs[i] - f_ref) <= 1e-4 * abs(f_ref)          # (up to 80 iterations of rounding differences)
            assert np.max(np.abs(xs[i] - x_ref)) <= 1e-4 * max(1.0, np.max(np.abs(x_ref)))
        assert (code != 5).sum() >= B - 2 and nev[code != 5].max() < cap
    # no rule set: the cap is the only stop, as before
    x0, c0, n0, k0 = ctx.optimize_batch_ex(b.x, lb, ub, 12)
    x1, c1 = ctx.optimize_batch(b.x, lb, ub, 12)
    assert np.array_equal(x0, x1) and np.array_equal(c0, c1) and np.all(n0 == 12) and np.all(k0 == 5)


def test_device_wall_clock_stop(scene, gtop):
    """set_maxtime (grad_traj_optimizer.cpp:144-148) inside the one-launch loop: past the limit every trajectory
    still running stops after the evaluation it is in (code 6) and returns the best point it has; a generous limit
    changes nothing."""
    import torch
    mp, ctx, sdf = scene
    b = problem.make_trajectories(512, 6, mp, seed=701)
    lb, ub = gtop.GtopContext.default_bounds(b.waypoints)
    dev = torch.device("cuda:0")
    Df = torch.tensor(b.Df.reshape(-1, 18), device=dev)
    T = torch.tensor(b.T, device=dev)
    lbt, ubt = torch.tensor(lb, device=dev), torch.tensor(ub, device=dev)
    ctx.set_params()
    for pin in (3, 6):                       # (6: two trajectories per wavefront)
        ctx.set_launch_geometry(0, pin)
        x, c, nev, code = ctx.optimize_device_ex(torch.tensor(b.x, device=dev), Df, T, lbt, ubt, 400, maxtime=20e-6)
        torch.cuda.synchronize()
        assert (code == 6).all() and (nev >= 1).all() and (nev < 400).all()
        chk, _ = ctx.eval_device(x, Df, T)
        torch.cuda.synchronize()
        ctx.set_launch_geometry(0, 0)
        assert torch.max(torch.abs(chk - c) / c).item() <= 1e-12       # min_cost is the cost of the returned point
    xa, ca, na, ka = ctx.optimize_device_ex(torch.tensor(b.x, device=dev), Df, T, lbt, ubt, 15, maxtime=10.0)
    xb, cb = ctx.optimize_device(torch.tensor(b.x, device=dev), Df, T, lbt, ubt, 15)
    torch.cuda.synchronize()
    assert torch.equal(xa, xb) and torch.equal(ca, cb) and (ka == 5).all() and (na == 15).all()


def test_whole_optimisation_replays_from_a_hip_graph(scene, gtop):
    """The one-launch optimizer is a single kernel node: captured once into a hipGraph (after a first eager call has
    sized the context's workspace) it replays bit-identically — the launch-bound form a planner loop would keep."""
    import torch
    mp, ctx, sdf = scene
    b = problem.make_trajectories(200, 6, mp, seed=4321)
    lb, ub = gtop.GtopContext.default_bounds(b.waypoints)
    dev = torch.device("cuda:0")
    Df = torch.tensor(b.Df.reshape(-1, 18), device=dev)
    T = torch.tensor(b.T, device=dev)
    lbt, ubt = torch.tensor(lb, device=dev), torch.tensor(ub, device=dev)
    ctx.set_params()
    x0 = torch.tensor(b.x, device=dev)
    x_e, c_e = ctx.optimize_device(x0.clone(), Df, T, lbt, ubt, 25)
    torch.cuda.synchronize()
    xg = x0.clone()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        xr, cr = ctx.optimize_device(xg, Df, T, lbt, ubt, 25)
    for _ in range(2):
        xg.copy_(x0)
        g.replay()
        torch.cuda.synchronize()
        assert torch.equal(xr, x_e) and torch.equal(cr, c_e)


def test_stop_rule_precedence_follows_nlopt(scene, oracle_mod, gtop):
    """The corner cases of NLopt's own stop logic (mma.c / stop.c): xtol is tested after ftol and its verdict stands
    when both hold (code 4, not 3); the evaluation limit is looked at right after every evaluation, so an evaluation
    that is the last one allowed reports MAXEVAL (5) even where it also completes an outer iteration meeting ftol.
    Device (the one-launch loop and the per-iteration launch form) against the numpy twin, on the jerk term alone."""
    mp, ctx, sdf = scene
    B, m = 16, 6
    b = problem.make_trajectories(B, m, mp, seed=710)
    lb, ub = gtop.GtopContext.default_bounds(b.waypoints)
    kw = dict(wc=0.0)
    prm = oracle_mod.make_params(**kw)
    ctx.set_problem(b.T, b.Df)
    ctx.set_params(**kw)
    try:
        # (1) both tolerances loose: wherever f AND x pass at the same outer iteration the code is 4
        rule = dict(ftol_rel=0.3, xtol_rel=0.5)
        xs, costs, nev, code = ctx.optimize_batch_ex(b.x, lb, ub, 60, **rule)
        refs = []
        for i in range(B):
            gen = oracle_mod.generator(b.T[i])

            def f(x, i=i, gen=gen):
                return oracle_mod.cost_grad(b.T[i], b.Df[i], x, sdf, prm, L=gen["L"], R=gen["R"])
            refs.append(mma_serial(f, b.x[i], lb[i], ub[i], 60, full=True, **rule))
            assert (nev[i], code[i]) == (refs[i][3], refs[i][4]), (i, nev[i], code[i], refs[i][3:])
        assert (code == 4).any() and set(code.tolist()) <= {3, 4}
        # (2) the cap placed exactly on the evaluation at which ftol fires: MAXEVAL wins
        rule = dict(ftol_rel=5e-2)
        _, _, nev_free, code_free = ctx.optimize_batch_ex(b.x, lb, ub, 80, **rule)
        assert (code_free == 3).sum() >= B // 2
        cap = int(np.bincount(nev_free[code_free == 3]).argmax())        # the most common stopping evaluation
        for mode in (2, 1):
            ctx.set_optimizer_fusion(mode)
            _, _, nev_c, code_c = ctx.optimize_batch_ex(b.x, lb, ub, cap, **rule)
            hit = (nev_free == cap) & (code_free == 3)
            assert hit.any() and (code_c[hit] == 5).all() and (nev_c[hit] == cap).all()
            early = (nev_free < cap) & (code_free == 3)
            assert (code_c[early] == 3).all() and np.array_equal(nev_c[early], nev_free[early])
        for i in np.flatnonzero(hit)[:3]:
            gen = oracle_mod.generator(b.T[i])

            def f(x, i=i, gen=gen):
                return oracle_mod.cost_grad(b.T[i], b.Df[i], x, sdf, prm, L=gen["L"], R=gen["R"])
            assert mma_serial(f, b.x[i], lb[i], ub[i], cap, full=True, **rule)[3:] == (cap, 5)
    finally:
        ctx.set_optimizer_fusion(2)
        ctx.set_params()


@pytest.mark.parametrize("B,m,fusion", [(300, 6, 2), (300, 6, 1), (200, 9, 2), (64, 17, 2), (4000, 6, 2), (4201, 6, 2),
                                          (4201, 5, 1)])
def test_optimizer_with_fp32_evaluations(scene, oracle_mod, gtop, B, m, fusion):
    """gtop_set_optimizer_precision(GTOP_F32): the loop's evaluations on the fp32 field in the fp32 bodies (up to 6
    segments — from 3 072 trajectories two per wavefront —, 7 .. 12 in packed pairs, the chunked body), its state,
    update and results fp64.  The objective it
    minimises is the fp32 one, so its road may part from the fp64 loop's at a decision inside fp32's noise; what must
    hold: the returned cost IS the callback's value at the returned point (to the fp32 bound), the point is inside
    its bounds and no worse than the start, evaluation counts and codes as the fp64 loop's, and the batch as a whole
    gets as far as the fp64 loop does."""
    mp, ctx, sdf = scene
    b = problem.make_trajectories(B, m, mp, seed=800 + m, step_len=(0.5, 1.2) if m > 6 else (1.0, 2.0))
    lb, ub = gtop.GtopContext.default_bounds(b.waypoints)
    ctx.set_params()
    ctx.set_problem(b.T, b.Df)
    evals = 25
    x64, c64, n64, code64 = ctx.optimize_batch_ex(b.x, lb, ub, evals)
    try:
        ctx.set_optimizer_precision("f32")
        ctx.set_optimizer_fusion(fusion)
        x32, c32, n32, code32 = ctx.optimize_batch_ex(b.x, lb, ub, evals)
        again = ctx.optimize_batch_ex(b.x, lb, ub, evals)
        ctx.set_optimizer_fusion(0)
        with pytest.raises(gtop.GtopError):
            ctx.optimize_batch_ex(b.x, lb, ub, evals)          # the separate-update form is fp64 only
    finally:
        ctx.set_optimizer_precision("f64")
        ctx.set_optimizer_fusion(2)
    for a, r in zip(again, (x32, c32, n32, code32)):
        assert np.array_equal(a, r)                              # deterministic
    assert np.array_equal(n32, n64) and np.array_equal(code32, code64)
    assert np.all(x32 >= lb - 1e-12) and np.all(x32 <= ub + 1e-12)
    prm = oracle_mod.make_params()
    idx = np.arange(B) if B <= 300 else np.r_[0:100, B - 100:B]
    c_at, _, _ = oracle_mod.eval_batch(b.T[idx], b.Df[idx], x32[idx], sdf, prm, nthreads=8)
    assert np.max(np.abs(c32[idx] - c_at) / np.abs(c_at)) <= 2e-4           # the fp32 bound of the evaluation
    c0, _, _ = oracle_mod.eval_batch(b.T[idx], b.Df[idx], b.x[idx], sdf, prm, nthreads=8)
    assert np.all(c32[idx] <= c0 * (1 + 2e-4))
    assert 0.8 <= np.median(c32 / c64) <= 1.25, np.median(c32 / c64)
    ratio32, ratio64 = np.median(c32[idx] / c0), np.median(c64[idx] / c0)
    assert ratio32 <= max(2 * ratio64, ratio64 + 0.05), (ratio32, ratio64)
    # the context's fp64 loop is untouched by the excursion
    back = ctx.optimize_batch_ex(b.x, lb, ub, evals)
    for a, r in zip(back, (x64, c64, n64, code64)):
        assert np.array_equal(a, r)


@pytest.mark.parametrize("name", ["m3_maxeval", "m6_ftol", "m4_xtol", "m8_both_tols"])
def test_device_loop_lands_where_the_committed_traces_end(gtop, oracle_mod, name):
    """tests/golden/mma_traces.npz holds evaluation-by-evaluation traces of the serial optimizer that TWO restatements of
    NLopt's LD_MMA sharing no code agree on to 1e-12 (oracle/mma_twin.py and csrc/mma.hpp, tests/test_mma_twin.py).  The
    device's one-launch loop, given the same problem, stop rules and bounds, must end where they end: the same
    evaluation count and stop code, the best point and its cost to 1e-6."""
    import os
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "mma_traces.npz"))
    g = {k.split("/", 1)[1]: z[k] for k in z.files if k.startswith(name + "/")}
    grid = tuple(int(v) for v in g["grid"])
    occ = np.unpackbits(g["occupancy"])[:int(np.prod(grid))].reshape(grid)
    ctx = gtop.GtopContext(device=0)
    ctx.init_sdf_map(g["map_size"], g["origin"], float(g["resolution"]))
    assert tuple(ctx.grid) == grid
    pts = (np.argwhere(occ == 1) + 0.5) * float(g["resolution"]) + g["origin"]
    ctx.update_sdf_map(pts)
    sdf = oracle_mod.Sdf.from_map_size(g["origin"], float(g["resolution"]), g["map_size"])
    sdf.build_from_occupancy(occ)
    assert np.array_equal(ctx.get_sdf().reshape(-1), sdf.dist)
    ctx.set_params(**{k: float(v) for k, v in zip(("ws", "wc"), g["params"]) if not np.isnan(v)})
    ctx.set_problem(g["T"][None], g["Df"][None])
    maxeval, ftol, xtol = int(g["stop"][0]), float(g["stop"][1]), float(g["stop"][2])
    # the first evaluations: the best value after k of them is the running minimum of the committed trace (1e-6: the
    # device's callback differs from the oracle's by 1e-12 per evaluation, and the iteration has had no time to amplify it)
    for k in (5, 15, 30):
        if k > int(g["nevals"]):
            continue
        xs, costs, nev, code = ctx.optimize_batch_ex(g["x0"][None], g["lb"][None], g["ub"][None], k)
        best = float(np.min(g["fs"][:k]))
        assert int(nev[0]) == k and int(code[0]) == 5
        assert abs(costs[0] - best) <= 1e-6 * abs(best), (k, costs[0], best)
        assert np.max(np.abs(xs[0] - g["xs"][int(np.argmin(g["fs"][:k]))])) <= 1e-6 * max(1.0, np.max(np.abs(g["x"])))
    # the whole run with its stop rules: the same evaluation count and stop code; the minimum to 1e-3 (after 100+
    # evaluations the asymptote updates — factors 0.7 / 1.2 on the sign of a product of steps — have amplified the
    # callbacks' last-bit differences: 4.5e-5 seen after 166 evaluations)
    xs, costs, nev, code = ctx.optimize_batch_ex(g["x0"][None], g["lb"][None], g["ub"][None], maxeval, ftol_rel=ftol,
                                                 xtol_rel=xtol)
    ctx.close()
    assert int(nev[0]) == int(g["nevals"]) and int(code[0]) == int(g["code"]), (nev, code, g["nevals"], g["code"])
    assert abs(costs[0] - float(g["minf"])) <= 1e-3 * abs(float(g["minf"]))
    assert np.max(np.abs(xs[0] - g["x"])) <= 1e-2 * max(1.0, np.max(np.abs(g["x"])))


def test_launch_forms_agree_on_random_problems_at_both_geometries(scene, gtop):
    """Random batch sizes (odd ones too: a last wavefront with a single trajectory), segment counts 2 .. 6, evaluation
    caps and stop rules; one and two trajectories per wavefront: the three launch forms give the same bits, and a
    trajectory's result does not depend on its place in the batch (whom it shares a wavefront with, first or second):
    the same rows in reversed order give the reversed results."""
    mp, ctx, sdf = scene
    rng = np.random.default_rng(4242)
    ctx.set_params()
    try:
        for draw in range(10):
            B, m = int(rng.integers(1, 400)), int(rng.integers(2, 7))
            cap = int(rng.integers(3, 45))
            rule = [dict(), dict(ftol_rel=float(rng.choice([1e-2, 5e-2]))), dict(xtol_rel=float(rng.choice([0.05, 0.2])))][draw % 3]
            b = problem.make_trajectories(B, m, mp, seed=6000 + draw)
            lb, ub = gtop.GtopContext.default_bounds(b.waypoints)
            for pin in (3, 6):
                ctx.set_launch_geometry(0, pin)
                ctx.set_problem(b.T, b.Df)
                res = {}
                for mode in (2, 1, 0):
                    ctx.set_optimizer_fusion(mode)
                    res[mode] = ctx.optimize_batch_ex(b.x, lb, ub, cap, **rule)
                for mode in (1, 0):
                    for a, r in zip(res[mode], res[2]):
                        assert np.array_equal(a, r), (draw, B, m, pin, mode)
                ctx.set_optimizer_fusion(2)
                ctx.set_problem(b.T[::-1].copy(), b.Df[::-1].copy())
                rev = ctx.optimize_batch_ex(b.x[::-1].copy(), lb[::-1].copy(), ub[::-1].copy(), cap, **rule)
                for a, r in zip(rev, res[2]):
                    assert np.array_equal(a[::-1], r), (draw, B, m, pin, "reversed")
    finally:
        ctx.set_optimizer_fusion(2)
        ctx.set_launch_geometry(0, 0)
