#!/usr/bin/env python3
"""Where the fp32 path's error comes from: the jerk term alone (wc = 0), the collision term alone (ws = 0) and both,
fp32 device result against the fp64 device result on configs[2]'s batch (max and quantiles over rows of the relative
cost error and of the inf-norm-relative gradient error)."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import grad_traj_optimization_amd as gtop  # noqa: E402
from grad_traj_optimization_amd import problem  # noqa: E402

mp = problem.make_map(200, density=0.02, seed=0)
ctx = gtop.GtopContext(device=0)
ctx.init_sdf_map(mp.map_size, mp.origin, mp.resolution)
ctx.update_sdf_map(mp.obstacle_points())
dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
b = problem.make_trajectories(B, 6, mp, seed=7)
t64 = [torch.tensor(a, device=dev) for a in (b.x, b.Df.reshape(-1, 18), b.T)]
t32 = [t.float() for t in t64]
for name, kw in (("jerk only (wc=0)", dict(wc=0.0)), ("collision only (ws=0)", dict(ws=0.0)), ("both (opti_node.launch)", {})):
    ctx.set_params(**kw)
    c64, g64 = ctx.eval_device(*t64)
    c32, g32 = ctx.eval_device(*t32)
    cq, gq = ctx.eval_device(*[t.double() for t in t32])      # fp64 arithmetic on the fp32-rounded INPUTS: the floor
    torch.cuda.synchronize()
    ec = ((c32.double() - c64).abs() / c64.abs()).cpu().numpy()
    eg = ((g32.double() - g64).abs().max(dim=1).values / g64.abs().max(dim=1).values).cpu().numpy()
    q = lambda e: "max %.2e  p99 %.2e  median %.2e" % (e.max(), np.quantile(e, 0.99), np.median(e))
    print(f"{name:28s} cost: {q(ec)}   grad: {q(eg)}", flush=True)
    fc = ((cq - c64).abs() / c64.abs()).cpu().numpy()
    fg = ((gq - g64).abs().max(dim=1).values / g64.abs().max(dim=1).values).cpu().numpy()
    print(f"{'  input rounding alone':28s} cost: {q(fc)}   grad: {q(fg)}", flush=True)
    worst = int(np.argmax(ec))
    print(f"    worst row {worst}: min segment time {b.T[worst].min():.3f} s, cost {float(c64[worst]):.4g}")
ctx.set_params()
