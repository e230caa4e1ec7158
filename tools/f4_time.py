#!/usr/bin/env python3
"""Diagnostic: the post-processing kernels of SURVEY §8f row f4 on bench-sized inputs, so that
`rocprofv3 --kernel-trace --stats -- python3 tools/f4_time.py` lists them."""
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
dev = torch.device("cuda:0")
rng = np.random.default_rng(0)
# distance queries: 1M random in-map positions, 32 moving boxes
N = 1 << 20
pos = torch.tensor(rng.uniform(mp.origin + 0.5, mp.origin + mp.map_size - 0.5, size=(N, 3)), device=dev)
tq = torch.tensor(rng.uniform(0.0, 3.0, size=N), device=dev)
nb = 32
ctx.set_moving_boxes(rng.uniform(mp.origin, mp.origin + mp.map_size, size=(nb, 3)), rng.uniform(-1, 1, size=(nb, 3)),
                     rng.uniform(0.3, 1.5, size=(nb, 3)))
for name, t in (("32 boxes", tq), ("static only", -torch.ones_like(tq))):
    ctx.edt_query_device(pos, t)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        ctx.edt_query_device(pos, t)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 10
    print(f"edt_query {name}: {N} queries in {dt * 1e6:.1f} us = {N / dt:.3e} queries/s, {128 * N / dt / 1e9:.0f} GB/s algorithmic")
# trajectory evaluation: 1024 optimised trajectories of 6 segments
b = problem.make_trajectories(1024, 6, mp, seed=1)
ctx.set_problem(b.T, b.Df)
ctx.trajectory_stats(b.x)
t0 = time.perf_counter()
for _ in range(5):
    ctx.trajectory_stats(b.x)
dt = (time.perf_counter() - t0) / 5
ns = ctx.trajectory_stats(b.x)[1][:, 8].sum()
print(f"trajectory_stats: 1024 trajectories ({int(ns)} getTraj samples) in {dt * 1e6:.1f} us wall (host copies included)")
