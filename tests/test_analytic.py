"""Hand-derived known answers (tests/golden/analytic.npz, derivation in tests/golden/ANALYTIC.md, generator
make_analytic.py — which builds no matrices and interpolates no field): a constant field, a field linear in x
(both exact under trilinear interpolation) with a straight constant-velocity path, and a single cubic whose jerk
integral is 36 T.  The oracle (CPU) and the HIP path (GPU) must reproduce them to 1e-9.

These pin the build's restatements to the reference's FORMULAS (grad_traj_optimizer.cpp:373-381, :417-432,
sdf_map.cpp:221-239, qp_generator.cpp:223-236) independently of one another; parity with the reference's
OUTPUTS stays unpinned (it cannot be built or run here)."""
import os

import numpy as np
import pytest

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
TOL = 1e-9


def check(c, g, c_ref, g_ref, what):
    """cost to 1e-9 relative; gradient to 1e-9 of its largest entry plus 1e-12 of the cost — the entries that are
    analytically zero (+1e-5) are sums of cancelling terms of the size of the cost."""
    assert abs(c - c_ref) <= TOL * abs(c_ref), (what, c, c_ref)
    tol_g = TOL * np.max(np.abs(g_ref)) + 1e-12 * abs(c_ref)
    assert np.max(np.abs(np.asarray(g) - g_ref)) <= tol_g, (what, np.max(np.abs(np.asarray(g) - g_ref)), tol_g)


@pytest.fixture(scope="module")
def ana():
    return np.load(os.path.join(GOLD, "analytic.npz"), allow_pickle=False)


def field_for(ana, case):
    nx, ny, nz = (int(g) for g in ana["grid"])
    res, org = float(ana["resolution"]), ana["origin"]
    if case == "A":
        return np.full((nx, ny, nz), float(ana["A_const"]))
    if case == "B":
        a, b = ana["B_lin"]
        xc = (np.arange(nx) + 0.5) * res + org[0]           # voxel centres (sdf_map.cpp:76-78)
        return np.broadcast_to((a + b * xc)[:, None, None], (nx, ny, nz)).copy()
    return np.full((nx, ny, nz), 5.0)                        # C: wc = 0, the field is never read


def params_of(ana, case):
    d = {str(k): float(v) for k, v in zip(ana["pkeys"], ana[f"{case}_params"])}
    d["step"] = int(d["step"])
    return d


@pytest.mark.parametrize("case", ["A", "B", "C"])
def test_oracle_reproduces_the_hand_derived_answers(oracle_mod, ana, case):
    sdf = oracle_mod.Sdf(ana["origin"], float(ana["resolution"]), tuple(int(g) for g in ana["grid"]),
                         dist=field_for(ana, case).reshape(-1))
    c, g = oracle_mod.cost_grad(ana[f"{case}_T"], ana[f"{case}_Df"], ana[f"{case}_x"], sdf,
                                oracle_mod.make_params(**params_of(ana, case)))
    check(c, g, float(ana[f"{case}_cost"]), ana[f"{case}_grad"], case)


def test_closed_forms_stated_in_the_derivation(ana):
    """A: cost = wc alpha e^{(d0-D)/r} (|v| + 1e-5) sum T + 1e-3;  C: cost = ws 36 (T1 + T2) + 1e-3, grad = 1e-5."""
    assert abs(float(ana["A_cost"]) - float(ana["A_cost_closed_form"])) <= 1e-13 * float(ana["A_cost"])
    p = params_of(ana, "C")
    assert float(ana["C_cost"]) == p["ws"] * 36.0 * float(ana["C_T"].sum()) + 1e-3
    assert np.all(ana["C_grad"] == 1e-5)


@pytest.mark.gpu
@pytest.mark.parametrize("case", ["A", "B", "C"])
def test_hip_path_reproduces_the_hand_derived_answers(gtop, ana, case):
    """Through every body that serves m <= 6 at fp64: ten lanes per segment (auto at B = 1, latency variant; spl 3
    pinned) and five lanes per segment with two trajectories per wavefront (spl 6), in batches of identical rows."""
    ctx = gtop.GtopContext(device=0)
    ctx.set_sdf(field_for(ana, case), ana["grid"], ana["origin"], float(ana["resolution"]))
    ctx.set_params(**params_of(ana, case))
    T, Df, x = ana[f"{case}_T"], ana[f"{case}_Df"], ana[f"{case}_x"]
    for spl, B in ((0, 1), (3, 7), (6, 7), (6, 2)):
        ctx.set_launch_geometry(0, spl)
        ctx.set_problem(np.repeat(T[None], B, 0), np.repeat(Df[None], B, 0))
        c, g = ctx.eval_batch(np.repeat(x[None], B, 0))
        for i in range(B):
            check(c[i], g[i], float(ana[f"{case}_cost"]), ana[f"{case}_grad"], (case, spl, i))
