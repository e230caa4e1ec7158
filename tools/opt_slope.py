#!/usr/bin/env python3
"""Diagnostic: time per pass of the one-launch optimizer loop (slope of wall time over the evaluation count, best of 7)."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.getcwd())
import grad_traj_optimization_amd as gtop
from grad_traj_optimization_amd import problem

mp = problem.make_map(200, density=0.02, seed=0)
ctx = gtop.GtopContext(0)
ctx.init_sdf_map(mp.map_size, mp.origin, mp.resolution)
ctx.update_sdf_map(mp.obstacle_points())
ctx.set_params()
dev = torch.device("cuda:0")
for B in [int(a) for a in sys.argv[1:]] or [1024, 16384]:
    b = problem.make_trajectories(B, 6, mp, seed=1)
    b = problem.permute(b, problem.spatial_order(b.waypoints, mp.origin, mp.map_size))
    lb, ub = gtop.GtopContext.default_bounds(b.waypoints)
    Df = torch.tensor(b.Df.reshape(-1, 18), device=dev)
    T = torch.tensor(b.T, device=dev)
    lbt, ubt = torch.tensor(lb, device=dev), torch.tensor(ub, device=dev)
    x0 = torch.tensor(b.x, device=dev)
    for mode in (2, 1, 0):
        ctx.set_optimizer_fusion(mode)
        ev = [50, 100, 200, 400]
        ts = []
        for evals in ev:
            best = 1e9
            for rep in range(7):
                x = x0.clone()
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                ctx.optimize_device(x, Df, T, lbt, ubt, evals)
                torch.cuda.synchronize()
                best = min(best, time.perf_counter() - t0)
            ts.append(best)
        slope, icpt = np.polyfit(ev, ts, 1)
        print(f"B={B} mode={mode}: " + " ".join(f"{e}:{t * 1e3:.3f}ms" for e, t in zip(ev, ts)) +
              f"  -> {slope * 1e6:.2f} us per pass + {icpt * 1e6:.0f} us fixed; {B / ts[0]:.3g} trajectories x 50 evals per s")
    ctx.set_optimizer_fusion(2)
