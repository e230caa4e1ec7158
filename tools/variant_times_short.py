#!/usr/bin/env python3
"""A short list of tools/variant_times.py's workloads (A/B runs): rows given as B,m,dtype[,dyn] on the command line."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import grad_traj_optimization_amd as gtop  # noqa: E402
from grad_traj_optimization_amd import problem  # noqa: E402
from bench import _time_evals  # noqa: E402

GRID = int(os.environ.get("GTOP_GRID", "200"))          # GTOP_GRID=400: the field of configs[4] (4 % occupied)
mp = problem.make_map(GRID, density=0.02 if GRID <= 200 else 0.04, seed=0 if GRID <= 200 else 2)
ctx = gtop.GtopContext(device=0)
ctx.init_sdf_map(mp.map_size, mp.origin, mp.resolution)
ctx.update_sdf_map(mp.obstacle_points())
dev = torch.device("cuda:0")
DYN = dict(enable_dyn=1, alpha_v=2.0, r_v=4.0, alpha_a=1.5, r_a=15.0, step=2)
for spec in sys.argv[1:]:
    f = spec.split(",")
    B, m, dt = int(f[0]), int(f[1]), f[2]
    prm = DYN if len(f) > 3 and f[3] == "dyn" else {}
    b = problem.make_trajectories(B, m, mp, seed=5, step_len=(0.5, 1.2) if m > 6 else (1.0, 2.0))
    if B > 1:
        b = problem.permute(b, problem.spatial_order(b.waypoints, mp.origin, mp.map_size))
    td = torch.float64 if dt == "f64" else torch.float32
    x = torch.tensor(b.x, dtype=td, device=dev)
    Df = torch.tensor(b.Df.reshape(-1, 18), dtype=td, device=dev)
    T = torch.tensor(b.T, dtype=td, device=dev)
    ctx.set_params(**prm)
    SPL = int(os.environ.get("GTOP_SPL", "0"))      # pinned samples per lane (0 = the launch rule)
    ctx.set_launch_geometry(0, SPL)
    us = _time_evals(ctx, x, Df, T, 600)
    print(f"B={B:6d} m={m:3d} {dt} {'dyn' if prm else '   '} g{GRID} spl={SPL}: {us:8.2f} us", flush=True)
