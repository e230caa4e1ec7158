#!/usr/bin/env python3
"""Diagnostic: run the ESDF build (gtop_update_sdf_map) on the bench maps so that
`rocprofv3 --kernel-trace --stats -- python3 tools/esdf_time.py` lists its kernels.
usage: tools/esdf_time.py [grid ...]   (default 200 400)"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.getcwd())
import grad_traj_optimization_amd as gtop
from grad_traj_optimization_amd import problem

for g in [int(a) for a in sys.argv[1:]] or [200, 400]:
    mp = problem.make_map(g, density=0.02 if g <= 200 else 0.04, seed=0)
    ctx = gtop.GtopContext(0)
    ctx.init_sdf_map(mp.map_size, mp.origin, mp.resolution)
    pts = mp.obstacle_points()
    ctx.update_sdf_map(pts)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        ctx.update_sdf_map(pts)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 3
    dp = torch.tensor(pts, device="cuda:0")
    ctx.update_sdf_map_device(dp)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        ctx.update_sdf_map_device(dp)
    e1.record()
    torch.cuda.synchronize()
    print(f"grid {g}^3: gtop_update_sdf_map_device (points resident) {e0.elapsed_time(e1) * 100:.1f} us per build (5 kernels)", flush=True)
    d = ctx.get_sdf()
    print(f"grid {g}^3: {len(pts)} obstacle points, update_sdf_map {dt * 1e3:.3f} ms wall (host copy of the points included), "
          f"checksum {float(np.sum(d)):.6f}", flush=True)
