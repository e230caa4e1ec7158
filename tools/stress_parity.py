#!/usr/bin/env python3
"""Randomised parity sweep (diagnostic): random segment counts, batch sizes (hence launch geometries and
kernel variants) and penalty parameters, device result of the whole batch vs the oracle on a row subsample.
usage: tools/stress_parity.py [n_cases]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.getcwd())
import grad_traj_optimization_amd as gtop
from grad_traj_optimization_amd import problem
from oracle import oracle

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
mp = problem.make_map((80, 70, 40), density=0.03, seed=5)
ctx = gtop.GtopContext(0)
ctx.init_sdf_map(mp.map_size, mp.origin, mp.resolution)
ctx.update_sdf_map(mp.obstacle_points())
sdf = oracle.Sdf.from_map_size(mp.origin, mp.resolution, mp.map_size)
sdf.dist[:] = ctx.get_sdf().reshape(-1)
dev = torch.device("cuda:0")
rng = np.random.default_rng(2024)
worst = (0.0, 0.0)
for case in range(n_cases):
    m = int(rng.integers(2, 13))
    B = int(rng.choice([1, 7, 255, 300, 1000, 2049, 4097, 8193, 9000]))
    kw = dict(ws=float(rng.uniform(0.1, 5)), wc=float(rng.uniform(0.5, 10)), alpha=float(rng.uniform(1, 20)),
              r=float(rng.uniform(0.2, 1.0)), d0=float(rng.uniform(0.3, 1.5)), step=int(rng.integers(1, 3)))
    b = problem.make_trajectories(B, m, mp, seed=1000 + case, step_len=(0.4, 1.5), margin=0.5)
    ctx.set_params(**kw)
    x = torch.tensor(b.x, device=dev)
    Df = torch.tensor(b.Df.reshape(-1, 18), device=dev)
    T = torch.tensor(b.T, device=dev)
    c, g = ctx.eval_device(x, Df, T)
    torch.cuda.synchronize()
    idx = rng.choice(B, min(B, 48), replace=False)
    c_ref, g_ref, _ = oracle.eval_batch(b.T[idx], b.Df[idx], b.x[idx], sdf, oracle.make_params(**kw), nthreads=8)
    cc, gg = c[idx].cpu().numpy(), g[idx].cpu().numpy()
    rc = float(np.max(np.abs(cc - c_ref) / np.abs(c_ref)))
    rg = float(np.max(np.max(np.abs(gg - g_ref), axis=1) / np.max(np.abs(g_ref), axis=1)))
    worst = (max(worst[0], rc), max(worst[1], rg))
    flag = "" if (rc < 1e-9 and rg < 1e-9) else "   <-- CHECK"
    print(f"case {case:3d}: m={m:2d} B={B:5d} step={kw['step']} rel cost {rc:.2e} grad {rg:.2e}{flag}", flush=True)
print(f"worst: cost {worst[0]:.2e} grad {worst[1]:.2e}")
