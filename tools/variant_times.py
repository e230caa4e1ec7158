#!/usr/bin/env python3
"""us per launch of gtop_eval_device for a list of workloads (hipGraph replays of 20 launches, HIP events, warm-up of
equal length) — the evaluation kernel's variants beside the bench line: enable_dyn at configs[1] / configs[2], more
than 12 segments, small batches of long trajectories.  usage: tools/variant_times.py [grid=200]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import grad_traj_optimization_amd as gtop  # noqa: E402
from grad_traj_optimization_amd import problem  # noqa: E402
from bench import _time_evals, algorithmic_bytes  # noqa: E402

grid = int(sys.argv[1]) if len(sys.argv) > 1 else 200
mp = problem.make_map(grid, density=0.02, seed=0)
ctx = gtop.GtopContext(device=0)
ctx.init_sdf_map(mp.map_size, mp.origin, mp.resolution)
ctx.update_sdf_map(mp.obstacle_points())
dev = torch.device("cuda:0")
DYN = dict(enable_dyn=1, alpha_v=2.0, r_v=4.0, alpha_a=1.5, r_a=15.0, step=2)
ROWS = [  # B, m, dtype, params, spl
    (1024, 6, "f64", {}, 0), (1024, 6, "f64", DYN, 0), (16384, 6, "f64", {}, 0), (16384, 6, "f64", DYN, 0),
    (16384, 6, "f32", {}, 0), (16384, 6, "f32", DYN, 0), (1024, 6, "f32", {}, 0), (1024, 6, "f32", DYN, 0),
    (1, 6, "f64", {}, 0), (1, 10, "f64", {}, 0), (1, 12, "f64", {}, 0), (256, 12, "f64", {}, 0), (1024, 12, "f64", {}, 0),
    (8192, 12, "f64", {}, 0), (8192, 12, "f64", DYN, 0), (1, 13, "f64", {}, 0), (1024, 13, "f64", {}, 0),
    (1024, 24, "f64", {}, 0), (8192, 24, "f64", {}, 0), (1024, 24, "f32", {}, 0), (8192, 24, "f64", DYN, 0),
    (256, 100, "f64", {}, 0),
]
print(f"# {grid}^3 field; us per launch (hipGraph replays), evaluations/s, fraction of 8 TB/s on algorithmic bytes")
for B, m, dt, prm, spl in ROWS:
    b = problem.make_trajectories(B, m, mp, seed=5, step_len=(0.5, 1.2) if m > 6 else (1.0, 2.0))
    if B > 1:
        b = problem.permute(b, problem.spatial_order(b.waypoints, mp.origin, mp.map_size))
    td = torch.float64 if dt == "f64" else torch.float32
    x = torch.tensor(b.x, dtype=td, device=dev)
    Df = torch.tensor(b.Df.reshape(-1, 18), dtype=td, device=dev)
    T = torch.tensor(b.T, dtype=td, device=dev)
    ctx.set_params(**prm)
    ctx.set_launch_geometry(0, spl)
    us = _time_evals(ctx, x, Df, T, 400)
    bpe = algorithmic_bytes(m, 8 if dt == "f64" else 4)
    print(f"B={B:6d} m={m:3d} {dt} {'dyn' if prm else '   '} spl={spl}: {us:8.2f} us  {B / us * 1e6:10.3e} evals/s  "
          f"frac {B * bpe / (us * 1e-6) / 8e12:.3f}", flush=True)
ctx.set_params()
