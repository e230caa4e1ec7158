#!/usr/bin/env python3
"""Diagnostic: where a lone wavefront of gtop_eval_wave_kernel spends its cycles (-DGTOP_STAMPS build).
usage: GTOP_HIP_LIB=build_var/libS.so python tools/stamps_wave.py [B]"""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.getcwd())
import grad_traj_optimization_amd as gtop
from grad_traj_optimization_amd import problem

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
mp = problem.make_map(200, density=0.02, seed=0)
ctx = gtop.GtopContext(0)
ctx.set_launch_geometry(1, 3)
ctx.init_sdf_map(mp.map_size, mp.origin, mp.resolution)
ctx.update_sdf_map(mp.obstacle_points())
b = problem.make_trajectories(B, 6, mp, seed=1)
b = problem.permute(b, problem.spatial_order(b.waypoints, mp.origin, mp.map_size))
dev = torch.device("cuda:0")
x = torch.tensor(b.x, device=dev)
Df = torch.tensor(b.Df.reshape(-1, 18), device=dev)
T = torch.tensor(b.T, device=dev)
cost, grad = ctx.eval_device(x, Df, T)
for _ in range(20):
    ctx.eval_device(x, Df, T, cost, grad)
torch.cuda.synchronize()
L = gtop.load_library()
buf = np.zeros((4096, 16), dtype=np.uint64)
assert L.gtop_debug_read_stamps(buf.ctypes.data_as(C.c_void_p)) == 0
nb = min(4096, B)
s = buf[:nb, :12].astype(np.int64)
names = ["wave start -> kernel arguments in SGPRs", "indices, addresses, 13 input loads issued", "inputs landed (wait)",
         "coefficients, sample times, stage A (12 corner loads issued)", "in flight: jerk term, speeds",
         "corner loads landed (wait)", "stage B: blend, penalty, accumulation", "A^-T, tile writes",
         "(nothing: two stamps back to back = a stamp's own cost)", "tile reads, sums, gradient + cost stores issued", "stores acknowledged (wait)"]
d = np.diff(s, axis=1)
print(f"B={B}: median cycles between stamps (lane 0), {nb} wavefronts; each stamp costs ~50-100 cycles itself")
for i, nme in enumerate(names):
    print(f"  {i:2d}->{i + 1:2d} {nme:62s} median {np.median(d[:, i]):7.0f}  p10 {np.percentile(d[:, i], 10):7.0f}  p90 {np.percentile(d[:, i], 90):7.0f}")
print(f"  total (0 -> 11) median {np.median(s[:, 11] - s[:, 0]):.0f} cycles; first start -> last end {s[:, 11].max() - s[:, 0].min()}; "
      f"start spread {s[:, 0].max() - s[:, 0].min()}")
hw = buf[:nb, 14].astype(np.int64)
xcc = buf[:nb, 15].astype(np.int64) & 0xF
key = (xcc << 20) | (((hw >> 13) & 7) << 16) | (((hw >> 12) & 1) << 15) | (((hw >> 8) & 15) << 4) | ((hw >> 4) & 3)
u, c = np.unique(key, return_counts=True)
print(f"  placement: {nb} wavefronts on {len(u)} SIMDs, per-SIMD histogram {dict(zip(*np.unique(c, return_counts=True)))}")
