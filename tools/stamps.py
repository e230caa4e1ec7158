#!/usr/bin/env python3
"""Diagnostic: per-phase cycles of gtop_eval_kernel from a -DGTOP_STAMPS build.
usage: GTOP_HIP_LIB=build_var/libS.so python tools/stamps.py [B] [spl] [waves] [dtype]"""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.getcwd())
import grad_traj_optimization_amd as gtop
from grad_traj_optimization_amd import problem

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
spl = int(sys.argv[2]) if len(sys.argv) > 2 else 0
waves = int(sys.argv[3]) if len(sys.argv) > 3 else 0
dt = torch.float64 if (len(sys.argv) <= 4 or sys.argv[4] == "f64") else torch.float32
mp = problem.make_map(200, density=0.02, seed=0)
ctx = gtop.GtopContext(0)
ctx.set_launch_geometry(waves, spl)
ctx.init_sdf_map(mp.map_size, mp.origin, mp.resolution)
ctx.update_sdf_map(mp.obstacle_points())
b = problem.make_trajectories(B, 6, mp, seed=1)
dev = torch.device("cuda:0")
x = torch.tensor(b.x, dtype=dt, device=dev)
Df = torch.tensor(b.Df.reshape(-1, 18), dtype=dt, device=dev)
T = torch.tensor(b.T, dtype=dt, device=dev)
cost, grad = ctx.eval_device(x, Df, T)
for _ in range(20):
    ctx.eval_device(x, Df, T, cost, grad)
torch.cuda.synchronize()
L = gtop.load_library()
buf = np.zeros((4096, 16), dtype=np.uint64)
rc = L.gtop_debug_read_stamps(buf.ctypes.data_as(C.c_void_p))
assert rc == 0
nb = min(4096, B)
s = buf[:nb, :7].astype(np.int64)
d = np.diff(s, axis=1)
names = ["phase0 load+stage", "phase1 coeff+ttable", "phase2 samples", "reduction", "phase3 A^-T", "phase4 store+cost"]
print(f"B={B} spl={spl} waves={waves} dtype={dt}: median cycles per phase (lane 0 of wave 0), first {nb} blocks")
for i, nme in enumerate(names):
    print(f"  {nme:22s} median {np.median(d[:, i]):8.0f}   p10 {np.percentile(d[:, i], 10):8.0f}   p90 {np.percentile(d[:, i], 90):8.0f}")
print(f"  total                  median {np.median(s[:, 6] - s[:, 0]):8.0f}")
print(f"  first start -> last end: {(s[:, 6].max() - s[:, 0].min())} cycles; block start spread {(s[:, 0].max() - s[:, 0].min())}")

f = buf[:nb].astype(np.int64)
print("  fine (first sample chunk): start->issued %.0f, issued->loads landed %.0f, landed->consumed %.0f" % (
    np.median(f[:, 8] - f[:, 2]), np.median(f[:, 9] - f[:, 8]), np.median(f[:, 10] - f[:, 9])))
print("  fine: samples end->tile written %.0f, tile->sums %.0f, sums->barrier %.0f; phase4: gradient scatter %.0f, cost %.0f" % (
    np.median(f[:, 11] - f[:, 3]), np.median(f[:, 12] - f[:, 11]), np.median(f[:, 4] - f[:, 12]),
    np.median(f[:, 13] - f[:, 5]), np.median(f[:, 6] - f[:, 13])))

# placement: how many of the launch's wavefronts shared a SIMD (HW_ID bits: wave 3:0, simd 5:4, cu 11:8, sh 12, se 15:13)
hw = buf[:nb, 14].astype(np.int64)
xcc = buf[:nb, 15].astype(np.int64) & 0xF
simd_key = (xcc << 20) | (((hw >> 13) & 7) << 16) | (((hw >> 12) & 1) << 15) | (((hw >> 8) & 15) << 4) | ((hw >> 4) & 3)
uniq, cnt = np.unique(simd_key, return_counts=True)
print(f"  placement: {nb} workgroups on {len(uniq)} distinct SIMDs; workgroups per SIMD histogram "
      f"{dict(zip(*np.unique(cnt, return_counts=True)))}; XCC histogram {dict(zip(*np.unique(xcc, return_counts=True)))}")
tot = s[:, 6] - s[:, 0]
for c in np.unique(cnt):
    sel = np.isin(simd_key, uniq[cnt == c])
    print(f"    SIMDs holding {c} workgroup(s): median kernel-body cycles {np.median(tot[sel]):.0f}")
