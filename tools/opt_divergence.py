#!/usr/bin/env python3
"""Diagnostic for tests/test_gpu_fuzz.py::test_random_draw_optimizer: for one seed, find the rows whose device result
leaves the serial road, the first evaluation count at which they do, and how the device's callback compares with the
oracle's at the serial road's trial points around it.  usage: tools/opt_divergence.py SEED"""
import os
import sys

import numpy as np

sys.path.insert(0, os.getcwd())
import grad_traj_optimization_amd as gtop
from oracle import mma_twin, oracle
from tests.test_gpu_fuzz import _draw

seed = int(sys.argv[1])
mp, b, kw, shared_T = _draw(seed)
rng = np.random.default_rng(seed)
B = min(len(b.x), 24)
T = np.broadcast_to(b.T, (len(b.x), b.m))[:B].copy()
T = np.maximum(T, 0.05)
Df, x0 = b.Df[:B], np.clip(b.x[:B], -1e3, 1e3)
if kw["ws"] == 0.0:
    kw["ws"] = 1.0
lb, ub = gtop.GtopContext.default_bounds(b.waypoints[:B], bos=float(rng.choice([0.5, 3.0])),
                                         vos=float(rng.choice([2.0, 8.0])), aos=float(rng.choice([3.0, 10.0])))
evals = int(rng.integers(2, 30))
sdf = oracle.Sdf.from_map_size(mp.origin, mp.resolution, mp.map_size)
sdf.build_from_occupancy(mp.occupancy)
prm = oracle.make_params(**kw)
ctx = gtop.GtopContext(device=0)
ctx.init_sdf_map(mp.map_size, mp.origin, mp.resolution)
ctx.update_sdf_map(mp.obstacle_points())
ctx.set_params(**kw)
ctx.set_problem(T, Df)
fusion = int(rng.choice([2, 2, 1, 0]))
ctx.set_optimizer_fusion(fusion)
print(f"seed {seed}: m={b.m} B={B} evals={evals} fusion={fusion} kw={kw}")
xs, costs, nev, code = ctx.optimize_batch_ex(x0, lb, ub, evals)
for i in range(B):
    gen = oracle.generator(T[i])

    def f(x):
        return oracle.cost_grad(T[i], Df[i], x, sdf, prm, L=gen["L"], R=gen["R"])
    tr = []
    tw = mma_twin.minimize(f, x0[i], lb[i], ub[i], evals, trace=tr)
    if not np.isfinite(tw["minf"]):
        continue
    rel = abs(costs[i] - tw["minf"]) / abs(tw["minf"])
    if rel <= 1e-6:
        continue
    print(f" row {i}: device {costs[i]:.12g} serial {tw['minf']:.12g} rel {rel:.2e}")
    run_min = np.minimum.accumulate(tw["fs"])
    for k in range(1, evals + 1):
        _, ck, _, _ = ctx.optimize_batch_ex(x0[i:i + 1], lb[i:i + 1], ub[i:i + 1], k) if False else (None, None, None, None)
    # device loop with k evaluations on the whole batch (row i looked at): first k where the best value differs
    first = None
    for k in range(1, evals + 1):
        _, c_k, _, _ = ctx.optimize_batch_ex(x0, lb, ub, k)
        r = abs(c_k[i] - run_min[k - 1]) / abs(run_min[k - 1])
        if r > 1e-9 and first is None:
            first = k
            print(f"   first difference at {k} evaluations: device best {c_k[i]:.15g} serial {run_min[k - 1]:.15g} rel {r:.2e}")
    if first is not None:
        xk_dev, _, _, _ = ctx.optimize_batch_ex(x0, lb, ub, first)
        k_best = int(np.argmin(tw["fs"][:first]))
        d = xk_dev[i] - tw["xs"][k_best]
        j = int(np.argmax(np.abs(d)))
        print(f"   best point after {first} evaluations: serial's is trial point {k_best + 1}; max |dx| {np.max(np.abs(d)):.3e} at j={j} "
              f"(x {tw['xs'][k_best][j]:.6g}, lb {lb[i][j]:.6g}, ub {ub[i][j]:.6g}); #coords differing > 1e-12: {int(np.sum(np.abs(d) > 1e-12))}")
        for k in range(max(0, first - 6), min(len(tr), first + 1)):
            t = tr[k]
            print(f"   trial {k + 2:3d}: outer k={t['k']} rho {t['rho']:.6e} g {t['g']:.6e} f {t['f']:.6e} w {t['w']:.4e} fbest {t['fbest']:.6e} "
                  f"(f-g)/w {(t['f'] - t['g']) / t['w']:.3e}")
    # the device's callback at the serial road's trial points
    c_dev, g_dev = ctx.eval_batch(np.broadcast_to(tw["xs"][:, None, :], (len(tw["xs"]), 1, tw["xs"].shape[1])).reshape(-1, tw["xs"].shape[1])[:1]) if False else (None, None)
    for k, xk in enumerate(tw["xs"]):
        X = np.repeat(x0, 1, axis=0).copy()
        X[i] = xk
        cd, gd = ctx.eval_batch(X)
        cr, gr = f(xk)
        ec = abs(cd[i] - cr) / abs(cr)
        eg = np.max(np.abs(gd[i] - gr)) / np.max(np.abs(gr))
        flag = "  <==" if max(ec, eg) > 1e-9 else ""
        if first is not None and abs(k + 1 - first) <= 3 or flag:
            print(f"   eval {k + 1:3d}: f {cr:.6e} callback rel err cost {ec:.1e} grad {eg:.1e}{flag}")
ctx.close()
