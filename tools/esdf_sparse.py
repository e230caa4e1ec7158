#!/usr/bin/env python3
"""Diagnostic: the ESDF build on sparse maps (a few obstacles in a large empty map — the reference's own scenes)
next to the bench's 2 % map and a reference-sized 200 x 200 x 25 grid, so that
`rocprofv3 --kernel-trace -- python3 tools/esdf_sparse.py` lists the sweeps per map (four builds each)."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.getcwd())
import grad_traj_optimization_amd as gtop
from grad_traj_optimization_amd import problem
for g, dens in ((200, 0.0005), (200, 0.002), (200, 0.02), ((200,200,25), 0.002)):
    mp = problem.make_map(g, density=dens, seed=0)
    ctx = gtop.GtopContext(0)
    ctx.init_sdf_map(mp.map_size, mp.origin, mp.resolution)
    pts = torch.tensor(mp.obstacle_points(), device="cuda:0")
    ctx.update_sdf_map(mp.obstacle_points())
    torch.cuda.synchronize()
    empty = sum(1 for x in range(mp.grid[0]) if not mp.occupancy[x].any())
    print(f"grid {mp.grid} density {dens}: {len(mp.obstacle_points())} points, {empty} empty slabs", flush=True)
    for _ in range(3): ctx.update_sdf_map(mp.obstacle_points())
    torch.cuda.synchronize()
