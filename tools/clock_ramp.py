#!/usr/bin/env python3
"""Diagnostic: per-launch time of the evaluation kernel in successive windows after an idle period — how long the
card takes to reach its sustained clocks, i.e. how much warm-up a probe needs before its figure means anything.
usage: tools/clock_ramp.py [B ...]"""
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
for B in [int(a) for a in sys.argv[1:]] or [1024, 16384]:
    b = problem.make_trajectories(B, 6, mp, seed=1)
    x, Df, T = (torch.tensor(a, dtype=torch.float64, device=dev) for a in (b.x, b.Df.reshape(-1, 18), b.T))
    cost, grad = ctx.eval_device(x, Df, T)
    torch.cuda.synchronize()
    per = 50
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, capture_error_mode="thread_local"):
        for _ in range(per):
            ctx.eval_device(x, Df, T, cost, grad)
    g.replay()
    torch.cuda.synchronize()
    for idle in (2.0, 0.0):
        time.sleep(idle)
        nwin, reps = 24, max(1, int(2000 // (per * (1 if B <= 1024 else 8))))
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(nwin + 1)]
        ev[0].record()
        for w in range(nwin):
            for _ in range(reps):
                g.replay()
            ev[w + 1].record()
        torch.cuda.synchronize()
        us = [ev[w].elapsed_time(ev[w + 1]) * 1e3 / (reps * per) for w in range(nwin)]
        t = np.cumsum([ev[w].elapsed_time(ev[w + 1]) for w in range(nwin)])
        print(f"B={B} after {idle:.0f} s idle: us per launch by window (window end, ms): " +
              " ".join(f"{u:.2f}@{tt:.0f}" for u, tt in zip(us, t)), flush=True)
