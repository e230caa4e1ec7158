"""gtop_eval_wave_kernel (csrc/gtop_kernels.hip) through every instantiation: one trajectory of up
to 6 segments per wavefront (samples per lane 3; latency variant below 3 072 trajectories, three-wavefront variant
above), one of up to 12 or two of up to 6 (samples per lane 6), three lanes per segment with 21 / m whole trajectories per
wavefront (samples per lane 10), one lane per segment with 64 / m (samples per lane 30), fp64 and fp32 (packed pairs at 6,
10 and 30), odd batches (a partial last pair / group,
padding workgroups), the edge cases of the sample loop, and the collision-free instantiation."""
import numpy as np
import pytest

from grad_traj_optimization_amd import problem
from tests import scenes

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


def _run(ctx, b, spl, dtype, **params):
    import torch
    td = torch.float64 if dtype == "f64" else torch.float32
    dev = torch.device("cuda:0")
    x = torch.tensor(b.x, dtype=td, device=dev)
    Df = torch.tensor(b.Df.reshape(-1, 18), dtype=td, device=dev)
    T = torch.tensor(b.T, dtype=td, device=dev)
    try:
        ctx.set_params(**params)
        ctx.set_launch_geometry(1, spl)
        c, g = ctx.eval_device(x, Df, T)
        torch.cuda.synchronize()
    finally:
        ctx.set_launch_geometry(0, 0)
        ctx.set_params()
    return c.double().cpu().numpy(), g.double().cpu().numpy()


@pytest.mark.parametrize("dtype", ["f64", "f32"])
@pytest.mark.parametrize("spl,m", [(3, 2), (3, 3), (3, 5), (3, 6), (6, 5), (6, 6), (6, 7), (6, 9), (6, 11), (6, 12),
                                   (30, 2), (30, 3), (30, 5), (30, 6), (30, 7), (30, 10), (30, 12),
                                   (10, 2), (10, 3), (10, 4), (10, 5), (10, 6), (10, 7), (10, 8), (10, 10),
                                   (30, 13), (30, 17), (30, 22), (30, 33), (30, 64)])
@pytest.mark.parametrize("B", [23, 3101])
def test_every_instantiation(scene, oracle_mod, dtype, spl, m, B):
    mp, ctx, sdf = scene
    b = problem.make_trajectories(B, m, mp, seed=600 + 13 * m + spl,
                                  step_len=(0.2, 0.5) if m > 12 else (0.5, 1.2) if m > 6 else (1.0, 2.0))
    c, g = _run(ctx, b, spl, dtype)
    idx = np.arange(B) if B <= 64 else np.r_[0:40, B - 40:B]          # both ends: the last pair / padding workgroups
    c_ref, g_ref, _ = oracle_mod.eval_batch(b.T[idx], b.Df[idx], b.x[idx], sdf, oracle_mod.make_params(), nthreads=8)
    rc, rg = scenes.rel_err(c[idx], g[idx], c_ref, g_ref)
    tol = TOL64 if dtype == "f64" else TOL32
    assert rc <= tol and rg <= tol, (rc, rg)
    assert np.isfinite(c).all() and np.isfinite(g).all()


@pytest.mark.parametrize("dtype", ["f64", "f32"])
@pytest.mark.parametrize("spl", [3, 6, 10, 30])
def test_edge_cases_of_the_sample_loop(scene, oracle_mod, dtype, spl):
    """Out-of-map samples (dist = -1, grad = 0), 29 / 20 / 0 / 30-sample segments (T = 0.03, 0.02, 0.0009, 0.031:
    the replay of `t += dt`, grad_traj_optimizer.cpp:353), together in one batch with ordinary rows."""
    mp, ctx, sdf = scene
    b = problem.make_trajectories(10, 4, mp, seed=12)
    x, T = b.x.copy(), b.T.copy()
    x[:4, 0] += 30.0
    x[4:8, 2 * 9] = -2.0
    T[0, 1], T[1, 0], T[2, 2], T[3, 3] = 0.03, 0.02, 0.0009, 0.031
    bb = problem.Batch(b.waypoints, T, b.Df, x, 4)
    kw = dict(ws=1e-6)                                                # the collision term dominates
    c, g = _run(ctx, bb, spl, dtype, **kw)
    c_ref, g_ref, _ = oracle_mod.eval_batch(T, b.Df, x, sdf, oracle_mod.make_params(**kw))
    rc, rg = scenes.rel_err(c, g, c_ref, g_ref)
    tol = TOL64 if dtype == "f64" else 2e-3        # (fp32: T = 0.0009 and 30 m outside the map are beyond its digits)
    assert rc <= tol and rg <= tol, (rc, rg)


@pytest.mark.parametrize("kw", [dict(wc=0.0), dict(wc=5e-5), dict(step=1), dict(ws=0.0), dict(ws=20.0, wc=1.0)])
@pytest.mark.parametrize("spl,m", [(3, 6), (6, 6), (6, 12), (3, 12), (3, 7), (30, 6), (30, 12), (30, 4), (10, 4), (10, 8)])
def test_parameter_sets_and_the_collision_free_instantiation(scene, oracle_mod, spl, m, kw):
    """|wc| < 1e-4 skips the sample loop (:346: the COLLI = false instantiation); step 1 drops the jerk weight (:412-415,
    applied by the launcher)."""
    mp, ctx, sdf = scene
    b = problem.make_trajectories(37, m, mp, seed=900 + m, step_len=(0.5, 1.2) if m > 6 else (1.0, 2.0))
    c, g = _run(ctx, b, spl, "f64", **kw)
    c_ref, g_ref, _ = oracle_mod.eval_batch(b.T, b.Df, b.x, sdf, oracle_mod.make_params(**kw))
    rc, rg = scenes.rel_err(c, g, c_ref, g_ref)
    assert rc <= TOL64 and rg <= TOL64, (rc, rg)


@pytest.mark.parametrize("dtype", ["f64", "f32"])
@pytest.mark.parametrize("m", [7, 8, 10, 12])
@pytest.mark.parametrize("B", [1, 23, 1024, 1025])
def test_two_wavefronts_per_trajectory(scene, oracle_mod, dtype, m, B):
    """7 .. 12 segments at ten lanes per segment over two wavefronts of one workgroup (the auto rule up to 1 024
    trajectories; five lanes per segment on one wavefront past that): against the oracle, with out-of-map samples and
    a 29-sample segment in the batch, and — fp64 — bit for bit the same row whatever the batch around it."""
    mp, ctx, sdf = scene
    b = problem.make_trajectories(B, m, mp, seed=640 + m, step_len=(0.5, 1.2), boundary="random")
    x, T = b.x.copy(), b.T.copy()
    x[0, 0] += 30.0                      # row 0 leaves the map
    T[B // 2, m - 1] = 0.03              # a 29-sample segment in the second wavefront's half
    bb = problem.Batch(b.waypoints, T, b.Df, x, m)
    c, g = _run(ctx, bb, 0, dtype)
    idx = np.arange(B) if B <= 64 else np.r_[0:40, B // 2 - 2:B // 2 + 2, B - 40:B]
    c_ref, g_ref, _ = oracle_mod.eval_batch(T[idx], b.Df[idx], x[idx], sdf, oracle_mod.make_params(), nthreads=8)
    rc, rg = scenes.rel_err(c[idx], g[idx], c_ref, g_ref)
    tol = TOL64 if dtype == "f64" else 2e-3          # (fp32: 30 m outside the map is beyond its digits)
    assert rc <= tol and rg <= tol, (rc, rg)
    if B == 1024 and dtype == "f64":                  # the same body for a batch of 3 of its rows
        sub = [0, 511, 1023]
        cs, gs = _run(ctx, problem.Batch(b.waypoints[sub], T[sub], b.Df[sub], x[sub], m), 0, dtype)
        assert np.array_equal(cs, c[sub]) and np.array_equal(gs, g[sub])


def test_rows_do_not_depend_on_their_place(scene):
    """A trajectory's result is the same bits wherever it sits in the batch: first or second of a wavefront's pair,
    in a full or a partial last pair, in the latency or the three-wavefront variant's batch."""
    mp, ctx, sdf = scene
    b = problem.make_trajectories(3101, 6, mp, seed=77)
    for spl, dtype in ((3, "f64"), (6, "f64"), (6, "f32"), (30, "f64"), (30, "f32"), (10, "f64"), (10, "f32")):
        c, g = _run(ctx, b, spl, dtype)
        perm = np.random.default_rng(1).permutation(3101)
        cp, gp = _run(ctx, problem.permute(b, perm), spl, dtype)
        assert np.array_equal(cp, c[perm]) and np.array_equal(gp, g[perm])
        if spl >= 6:                                   # same body at any batch size
            cs, gs = _run(ctx, problem.permute(b, perm[:11]), spl, dtype)
            assert np.array_equal(cs, c[perm[:11]]) and np.array_equal(gs, g[perm[:11]])


@pytest.mark.timeout(900)
def test_hand_issued_loads_equal_compiler_issued_loads(scene, tmp_path):
    """The latency variant with its distance-field loads issued by hand (inline asm + hand-written s_waitcnt, the
    shipped build) against the same variant built with -DGTOP_ASM_LOADS=0 (the compiler's own loads and waits): bit
    for bit the same results.  The second library is compiled here (hipcc is on the GPU box) and driven in a child
    process through GTOP_HIP_LIB."""
    import os
    import subprocess
    import sys
    mp, ctx, sdf = scene
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    alt = str(tmp_path / "libgtop_noasm.so")
    subprocess.check_call(["make", "-C", os.path.join(root, "grad_traj_optimization_amd", "csrc"), "-s", "-j", "8", "lib",
                           f"OUT={alt}", "EXTRA=-DGTOP_ASM_LOADS=0"], timeout=800)
    script = r"""
import sys, numpy as np, torch
sys.path.insert(0, %r)
import grad_traj_optimization_amd as gtop
from grad_traj_optimization_amd import problem
mp = problem.make_map((60, 50, 30), density=0.03, seed=11)
ctx = gtop.GtopContext(device=0)
ctx.init_sdf_map(mp.map_size, mp.origin, mp.resolution)
ctx.update_sdf_map(mp.obstacle_points())
out = {}
for B, m in ((1, 6), (777, 6), (100, 3)):
    b = problem.make_trajectories(B, m, mp, seed=31 + B, boundary="random")
    x = b.x.copy()
    if B == 777:
        x[:5, 0] += 30.0                      # out-of-map samples: the rare branch
    ctx.set_problem(b.T, b.Df)
    c, g = ctx.eval_batch(x)
    out[f"c{B}"], out[f"g{B}"] = c, g
np.savez(sys.argv[1], **out)
""" % root
    res = {}
    for name, lib in (("asm", None), ("noasm", alt)):
        env = dict(os.environ)
        if lib:
            env["GTOP_HIP_LIB"] = lib
        else:
            env.pop("GTOP_HIP_LIB", None)
        f = str(tmp_path / f"{name}.npz")
        subprocess.check_call([sys.executable, "-c", script, f], env=env, timeout=300)
        res[name] = np.load(f)
    for k in res["asm"].files:
        assert np.array_equal(res["asm"][k], res["noasm"][k]), k


@pytest.mark.parametrize("dtype", ["f64", "f32"])
@pytest.mark.parametrize("spl", [3, 6, 10, 30])
def test_samples_past_the_loop_bound_cannot_overflow_into_the_sums(scene, oracle_mod, dtype, spl):
    """A 20 ms segment has 20 samples (`t += dt` up to T, src/grad_traj_optimizer.cpp:353); the lanes of the other ten
    evaluate too, with weight 0.  With the dyn block on and r_v just large enough for the reference's own samples
    to stay inside the exponent range, the polynomial EXTRAPOLATED past T overflows exp: those lanes are evaluated
    at the first sample's time instead, so that no 0 * inf reaches the sums."""
    mp, ctx, sdf = scene
    b = problem.make_trajectories(6, 4, mp, seed=77, boundary="random")
    T = b.T.copy()
    T[:, 1] = 0.02
    limit = 700.0 if dtype == "f64" else 70.0        # (exp's argument range, with a margin for the gradient's factors)
    kw = dict(enable_dyn=1, alpha_v=1.0, v0=1.0, alpha_a=0.0, r_a=1e6, a0=1.0)
    lo, hi = 1e-3, 1e3                                # smallest r_v for which the reference stays finite and in range
    for _ in range(40):
        mid = np.sqrt(lo * hi)
        c_try = oracle_mod.eval_batch(T, b.Df, b.x, sdf, oracle_mod.make_params(r_v=mid, **kw))[0]
        ok = np.isfinite(c_try).all() and c_try.max() < np.exp(limit)
        lo, hi = (lo, mid) if ok else (mid, hi)
    kw["r_v"] = float(hi * 1.02)
    c_ref, g_ref, _ = oracle_mod.eval_batch(T, b.Df, b.x, sdf, oracle_mod.make_params(**kw))
    assert np.isfinite(c_ref).all() and c_ref.max() > np.exp(0.8 * limit)      # live samples close to the limit
    c, g = _run(ctx, problem.Batch(b.waypoints, T, b.Df, b.x, 4), spl, dtype, **kw)
    assert np.isfinite(c).all() and np.isfinite(g).all()
    rc, rg = scenes.rel_err(c, g, c_ref, g_ref)
    tol = TOL64 if dtype == "f64" else 2e-2           # (fp32: exp(80) turns the velocity's rounding into 1e-3 and more)
    assert rc <= tol and rg <= tol, (rc, rg)


@pytest.mark.parametrize("dtype", ["f64", "f32"])
@pytest.mark.parametrize("m,B", [(3, 8200), (8, 4100), (13, 4100), (24, 2100), (40, 1030), (64, 1025), (65, 300)])
def test_the_launch_rule_left_to_itself_past_its_switch_points(scene, oracle_mod, dtype, m, B):
    """Batches just past the switch points of gtop_eval_plan, nothing pinned: three lanes per segment (3 and 8 segments),
    one lane per segment (13, 40, 64), the chunked body where that one stays ahead (24) or has to (65) — against the
    oracle on both ends of the batch (the last, partly filled wavefront included)."""
    mp, ctx, sdf = scene
    b = problem.make_trajectories(B, m, mp, seed=4000 + m, step_len=(0.2, 0.5) if m > 12 else (0.5, 1.2) if m > 6 else (1.0, 2.0))
    c, g = _run(ctx, b, 0, dtype)
    idx = np.r_[0:24, B - 24:B]
    c_ref, g_ref, _ = oracle_mod.eval_batch(b.T[idx], b.Df[idx], b.x[idx], sdf, oracle_mod.make_params(), nthreads=8)
    rc, rg = scenes.rel_err(c[idx], g[idx], c_ref, g_ref)
    tol = TOL64 if dtype == "f64" else TOL32 * (3 if m > 12 else 1)      # (fp32: long chains of short segments, as in test_gpu_kino)
    assert rc <= tol and rg <= tol, (rc, rg)
    assert np.isfinite(c).all() and np.isfinite(g).all()
