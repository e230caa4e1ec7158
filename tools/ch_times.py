#!/usr/bin/env python3
"""Diagnostic: launch time of the six-samples-per-lane bodies on the HBM-bound and the cache-resident workloads
(A/B of library builds: GTOP_HIP_LIB).  usage: tools/ch_times.py"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.getcwd())
import grad_traj_optimization_amd as gtop
from grad_traj_optimization_amd import problem

dev = torch.device("cuda:0")
out = []
for grid, den, seed, cases in ((400, 0.04, 2, [(8192, 12, 3)]), (200, 0.02, 0, [(16384, 6, 7), (8192, 12, 5), (65536, 6, 9)])):
    mp = problem.make_map(grid, density=den, seed=seed)
    ctx = gtop.GtopContext(0)
    ctx.init_sdf_map(mp.map_size, mp.origin, mp.resolution)
    ctx.update_sdf_map(mp.obstacle_points())
    for B, m, s in cases:
        b = problem.make_trajectories(B, m, mp, seed=s)
        b = problem.permute(b, problem.spatial_order(b.waypoints, mp.origin, mp.map_size))
        x, Df, T = (torch.tensor(a, dtype=torch.float64, device=dev) for a in (b.x, b.Df.reshape(-1, 18), b.T))
        cost, grad = ctx.eval_device(x, Df, T)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, capture_error_mode="thread_local"):
            for _ in range(20):
                ctx.eval_device(x, Df, T, cost, grad)
        t0 = time.perf_counter()
        while time.perf_counter() - t0 < 0.05:
            g.replay()
            torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            g.replay()
        e1.record()
        torch.cuda.synchronize()
        out.append(f"{grid}^3 B={B} m={m}: {e0.elapsed_time(e1) * 1e3 / 400:.2f} us  (sum {float(cost.sum().item()):.10e})")
    ctx.close()
print(" | ".join(out))
