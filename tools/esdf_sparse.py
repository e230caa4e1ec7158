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
