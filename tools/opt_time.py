#!/usr/bin/env python3
"""Diagnostic: wall time of gtop_optimize_device for launch geometries and fusion modes.
usage: tools/opt_time.py [B ...]"""
import os
import sys
import time

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
for B in [int(a) for a in sys.argv[1:]] or [1024, 4096, 16384]:
    M = int(os.environ.get("GTOP_M", "6"))                    # segments per trajectory
    b = problem.make_trajectories(B, M, mp, seed=1, step_len=(0.5, 1.2) if M > 6 else (1.0, 2.0))
    perm = problem.spatial_order(b.waypoints, mp.origin, mp.map_size)
    b = problem.permute(b, perm)
    lb, ub = gtop.GtopContext.default_bounds(b.waypoints)
    Df = torch.tensor(b.Df.reshape(-1, 18), device=dev)
    T = torch.tensor(b.T, device=dev)
    lbt, ubt = torch.tensor(lb, device=dev), torch.tensor(ub, device=dev)
    x0 = torch.tensor(b.x, device=dev)
    rows = ((3, 2, "f64"), (3, 2, "f64"), (6, 2, "f64"), (3, 2, "f32"), (6, 2, "f32"), (0, 1, "f64"), (0, 1, "f32"))
    if os.environ.get("GTOP_OPT_ROWS"):                       # e.g. "0:f64,10:f64,0:f32,10:f32": pinned samples per lane : precision
        rows = tuple((int(r.split(":")[0]), 2, r.split(":")[1]) for r in os.environ["GTOP_OPT_ROWS"].split(","))
    for spl, mode, prec in rows:
        if True:
            ctx.set_launch_geometry(0, spl)
            ctx.set_optimizer_fusion(mode)
            ctx.set_optimizer_precision(prec)
            ts = []
            for evals in (50, 100):
                t_w = time.perf_counter()
                while time.perf_counter() - t_w < 0.04:      # sustained clocks (tools/clock_ramp.py)
                    ctx.optimize_device(x0.clone(), Df, T, lbt, ubt, evals)
                    torch.cuda.synchronize()
                x = x0.clone()
                t0 = time.perf_counter()
                ctx.optimize_device(x, Df, T, lbt, ubt, evals)
                torch.cuda.synchronize()
                ts.append(time.perf_counter() - t0)
            print(f"B={B} m={M} spl={spl} mode={mode} {prec}: 50 evals {ts[0]*1e3:.3f} ms, 100 evals {ts[1]*1e3:.3f} ms, "
                  f"slope {(ts[1]-ts[0])/50*1e6:.2f} us/eval-round", flush=True)
