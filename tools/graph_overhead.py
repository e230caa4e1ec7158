#!/usr/bin/env python3
"""Diagnostic: where the fixed cost of a short timed region goes (one hipGraph replay of K evaluation kernels):
the host's launch call, the wait for completion, event records."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.getcwd())
import grad_traj_optimization_amd as gtop
from grad_traj_optimization_amd import problem

K = int(sys.argv[1]) if len(sys.argv) > 1 else 20
mp = problem.make_map(200, density=0.02, seed=0)
ctx = gtop.GtopContext(0)
ctx.init_sdf_map(mp.map_size, mp.origin, mp.resolution)
ctx.update_sdf_map(mp.obstacle_points())
b = problem.make_trajectories(1024, 6, mp, seed=1)
dev = torch.device("cuda:0")
x = torch.tensor(b.x, device=dev)
Df = torch.tensor(b.Df.reshape(-1, 18), device=dev)
T = torch.tensor(b.T, device=dev)
cost = torch.zeros(K, 1024, dtype=torch.float64, device=dev)
grad = torch.zeros(1024, 45, dtype=torch.float64, device=dev)
ctx.eval_device(x, Df, T, cost[0], grad)
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    for s in range(K):
        ctx.eval_device(x, Df, T, cost[s], grad)
for _ in range(3):
    g.replay()
torch.cuda.synchronize()
stream = torch.cuda.current_stream()
res = []
for mode in ("sync", "poll_stream", "poll_event"):
    rows = []
    for rep in range(200):
        torch.cuda.synchronize()
        ev = torch.cuda.Event()
        t0 = time.perf_counter()
        g.replay()
        t1 = time.perf_counter()
        if mode == "sync":
            torch.cuda.synchronize()
        elif mode == "poll_stream":
            while not stream.query():
                pass
        else:
            ev.record(stream)
            while not ev.query():
                pass
        t2 = time.perf_counter()
        rows.append((t1 - t0, t2 - t1, t2 - t0))
    r = np.median(np.array(rows), axis=0) * 1e6
    print(f"K={K} {mode:12s}: launch call {r[0]:6.1f} us, wait {r[1]:6.1f} us, total {r[2]:6.1f} us = {r[2] / K:5.2f} us per step")
