#!/usr/bin/env python3
"""us per launch with the samples per lane pinned to 3 and to 6 (auto rule beside them): tools/spl_compare.py B,m,dtype ..."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import grad_traj_optimization_amd as gtop  # noqa: E402
from grad_traj_optimization_amd import problem  # noqa: E402
from bench import _time_evals  # noqa: E402

mp = problem.make_map(200, density=0.02, seed=0)
ctx = gtop.GtopContext(device=0)
ctx.init_sdf_map(mp.map_size, mp.origin, mp.resolution)
ctx.update_sdf_map(mp.obstacle_points())
dev = torch.device("cuda:0")
for spec in sys.argv[1:]:
    f = spec.split(",")
    B, m, dt = int(f[0]), int(f[1]), f[2]
    b = problem.make_trajectories(B, m, mp, seed=5, step_len=(0.5, 1.2) if m > 6 else (1.0, 2.0))
    b = problem.permute(b, problem.spatial_order(b.waypoints, mp.origin, mp.map_size))
    td = torch.float64 if dt == "f64" else torch.float32
    x = torch.tensor(b.x, dtype=td, device=dev)
    Df = torch.tensor(b.Df.reshape(-1, 18), dtype=td, device=dev)
    T = torch.tensor(b.T, dtype=td, device=dev)
    out = []
    for spl in (0, 3, 6, 0, 3, 6):
        ctx.set_launch_geometry(0, spl)
        try:
            out.append("%d:%.2f" % (spl, _time_evals(ctx, x, Df, T, 600)))
        except Exception as e:
            out.append("%d:n/a" % spl)
    ctx.set_launch_geometry(0, 0)
    print(f"B={B} m={m} {dt}: " + "  ".join(out), flush=True)
