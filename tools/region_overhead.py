#!/usr/bin/env python3
"""Diagnostic: where the host's clock around a 20-step timed region goes (bench.py at the driver's --steps 20): the
graph launch call, the wait for completion, the final synchronize — against the device clock's span of the same graph."""
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
b = problem.permute(b, problem.spatial_order(b.waypoints, mp.origin, mp.map_size))
dev = torch.device("cuda:0")
x, Df, T = (torch.tensor(a, device=dev) for a in (b.x, b.Df.reshape(-1, 18), b.T))
cost = torch.zeros(K, 1024, dtype=torch.float64, device=dev)
grad = torch.zeros(1024, 45, dtype=torch.float64, device=dev)
INIT = torch.tensor([2 ** 63 - 1, 0], dtype=torch.int64, device=dev)
stamps = INIT.clone()
hz = ctx.clock_hz()
stream = torch.cuda.current_stream(dev)
gph = torch.cuda.CUDAGraph()
with torch.cuda.graph(gph):
    ctx.clock_stamp(stamps)
    for s in range(K):
        ctx.eval_device(x, Df, T, cost[s], grad)
    ctx.clock_stamp(stamps)
for _ in range(200):
    gph.replay()
torch.cuda.synchronize()
rows = []
for rep in range(30):
    t_w = time.perf_counter()
    while time.perf_counter() - t_w < 0.02:
        gph.replay()
        torch.cuda.synchronize()
    stamps.copy_(INIT)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    gph.replay()
    t1 = time.perf_counter()
    n = 0
    while not stream.query():
        n += 1
    t2 = time.perf_counter()
    torch.cuda.synchronize()
    t3 = time.perf_counter()
    st = stamps.tolist()
    rows.append(((t1 - t0) * 1e6, (t2 - t1) * 1e6, (t3 - t2) * 1e6, (t3 - t0) * 1e6, (st[1] - st[0]) / hz * 1e6, n))
r = np.array(rows)
print(f"K={K}: median us — replay() call {np.median(r[:,0]):.1f}, poll until query() true {np.median(r[:,1]):.1f} ({np.median(r[:,5]):.0f} polls), "
      f"synchronize {np.median(r[:,2]):.1f}, total host {np.median(r[:,3]):.1f}; device clock first->last stamp {np.median(r[:,4]):.1f}")
# the same with a blocking synchronize only
rows = []
for rep in range(30):
    gph.replay(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    gph.replay()
    torch.cuda.synchronize()
    rows.append((time.perf_counter() - t0) * 1e6)
print(f"      replay + synchronize only: median {np.median(rows):.1f} us")
