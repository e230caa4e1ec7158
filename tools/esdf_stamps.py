#!/usr/bin/env python3
"""Tuning aid: timeline of the x sweep's wavefronts (a library built with -DGTOP_ESDF_STAMPS, named by
GTOP_HIP_LIB) — or of the y sweep's (-DGTOP_ESDF_STAMPS -DGTOP_ESDF_STAMP_Y; "steps" are then the trips of its two
candidate loops; set ESDF_WAVES_PER_BLOCK=4).  usage: GTOP_HIP_LIB=.../libgtop_st.so python3 tools/esdf_stamps.py [grid=200]"""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.getcwd())
import grad_traj_optimization_amd as gtop
from grad_traj_optimization_amd import problem, _lib

g = int(sys.argv[1]) if len(sys.argv) > 1 else 200
mp = problem.make_map(g, density=0.02 if g <= 200 else 0.04, seed=0)
ctx = gtop.GtopContext(0)
ctx.init_sdf_map(mp.map_size, mp.origin, mp.resolution)
pts = mp.obstacle_points()
for _ in range(3):
    ctx.update_sdf_map(pts)
torch.cuda.synchronize()
lib = ctypes.CDLL(os.environ["GTOP_HIP_LIB"])
n = 4 * 65536
buf = (ctypes.c_ulonglong * n)()
rc = lib.gtop_debug_esdf_stamps(buf, ctypes.c_size_t(n))
assert rc == 0, rc
a = np.frombuffer(buf, dtype=np.uint64).reshape(-1, 4).astype(np.int64)
a = a[a[:, 1] > 0]
t0 = a[:, 0].min()
start = (a[:, 0] - t0) / 100.0          # us (100 MHz clock)
end = (a[:, 1] - t0) / 100.0
dur = end - start
steps = a[:, 2]
print(f"grid {g}^3: {len(a)} wavefronts recorded; kernel span {end.max():.1f} us")
print(f"start: p50 {np.median(start):.1f} p90 {np.percentile(start, 90):.1f} max {start.max():.1f} us")
print(f"end:   p10 {np.percentile(end, 10):.1f} p50 {np.median(end):.1f} p90 {np.percentile(end, 90):.1f} max {end.max():.1f} us")
print(f"duration: p10 {np.percentile(dur, 10):.1f} p50 {np.median(dur):.1f} p90 {np.percentile(dur, 90):.1f} max {dur.max():.1f} us")
if (steps >= 0).any():
    for lo, hi in ((0, 8), (8, 16), (16, 32), (32, 48), (48, 64), (64, 300)):
        k = (steps >= lo) & (steps < hi)
        if k.any():
            print(f"  steps [{lo},{hi}): {k.sum():5d} waves, duration p50 {np.median(dur[k]):6.1f} max {dur[k].max():6.1f} us, "
                  f"us/step {np.median(dur[k] / np.maximum(steps[k], 1)):.2f}")
# waves alive over time
ts = np.linspace(0, end.max(), 21)
alive = [(int(((start <= t) & (end > t)).sum())) for t in ts]
print("alive wavefronts at", " ".join(f"{t:.0f}us:{n}" for t, n in zip(ts, alive)))
late = start > 5
print(f"wavefronts starting after 5 us: {late.sum()}; their start p50 {np.median(start[late]) if late.any() else 0:.1f}")
# per XCD (workgroups are dealt round-robin: xcd = blockIdx & 7) and per SIMD (HW_ID: wave 3:0, simd 5:4, cu 11:8, sh 12, se 15:13)
wpb = int(os.environ.get("ESDF_WAVES_PER_BLOCK", "2"))
idx = np.nonzero(np.frombuffer(buf, dtype=np.uint64).reshape(-1, 4)[:, 1] > 0)[0]
xcd = (idx // wpb) & 7
print("per XCD: waves, sum of steps, last end (us)")
for c in range(8):
    k = xcd == c
    print(f"  xcd {c}: {k.sum():5d} waves, steps sum {int(np.maximum(steps[k], 0).sum()):7d}, end p50 {np.median(end[k]):5.1f} max {end[k].max():5.1f}")
hw = a[:, 3] & 0xFFFF
simd_key = xcd * 65536 + (hw & 0xFFF0)     # (xcd, se, sh, cu, simd)
keys, inv = np.unique(simd_key, return_inverse=True)
last = np.zeros(len(keys)); cnt = np.zeros(len(keys)); st = np.zeros(len(keys))
np.maximum.at(last, inv, end); np.add.at(cnt, inv, 1); np.add.at(st, inv, np.maximum(steps, 0))
print(f"SIMDs used {len(keys)}: waves per SIMD min {cnt.min():.0f} p50 {np.median(cnt):.0f} max {cnt.max():.0f}; "
      f"last end per SIMD p10 {np.percentile(last, 10):.1f} p50 {np.median(last):.1f} p90 {np.percentile(last, 90):.1f} max {last.max():.1f} us")
if st.max() > 0:
    print(f"steps per SIMD: min {st.min():.0f} p50 {np.median(st):.0f} max {st.max():.0f}; corr(steps sum, last end) = {np.corrcoef(st, last)[0, 1]:.2f}")
