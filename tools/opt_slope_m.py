#!/usr/bin/env python3
"""Diagnostic: per-pass time of the one-launch optimizer loop for several segment counts (auto geometry vs pinned)."""
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
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
for m in (6, 9, 12):
    b = problem.make_trajectories(B, m, mp, seed=1)
    b = problem.permute(b, problem.spatial_order(b.waypoints, mp.origin, mp.map_size))
    lb, ub = gtop.GtopContext.default_bounds(b.waypoints)
    Df = torch.tensor(b.Df.reshape(-1, 18), device=dev)
    T = torch.tensor(b.T, device=dev)
    lbt, ubt = torch.tensor(lb, device=dev), torch.tensor(ub, device=dev)
    x0 = torch.tensor(b.x, device=dev)
    for label, spl in (("auto", 0), ("pinned 10 lanes/segment", 3)):
        ctx.set_launch_geometry(0, spl)
        ev = [50, 100, 200]
        ts = []
        for evals in ev:
            best = 1e9
            for rep in range(5):
                x = x0.clone()
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                ctx.optimize_device(x, Df, T, lbt, ubt, evals)
                torch.cuda.synchronize()
                best = min(best, time.perf_counter() - t0)
            ts.append(best)
        slope, icpt = np.polyfit(ev, ts, 1)
        print(f"B={B} m={m} {label}: 50 evals {ts[0] * 1e3:.3f} ms, {slope * 1e6:.2f} us per pass")
    ctx.set_launch_geometry(0, 0)
