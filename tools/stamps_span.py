#!/usr/bin/env python3
"""Evidence for the bench default's launch time that needs neither the host's clock nor rocprofv3's dispatch
instrumentation: a -DGTOP_STAMPS=2 build of the library has lane 0 of every wavefront read the shader clock (s_memtime)
and the constant-rate wall clock (wall_clock64, 100 MHz) when it starts and when its last store has been acknowledged.
For the LAST launch of a 50-kernel hipGraph replayed at sustained clocks this prints: the wavefronts' lifetimes, the span
first start -> last end (what the kernel itself takes: no launch overhead in it), the start spread (dispatch of 1 024
workgroups), and the shader clock's rate.  Beside it the bench's own device-clock figure for the same graph (stamp
kernels around it): the difference between the two is the per-launch dispatch floor (tools/ubench/launch_floor2).
usage: GTOP_HIP_LIB=build_var/libgtop_stamps2.so python tools/stamps_span.py [B]"""
import ctypes as C
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.getcwd())
import grad_traj_optimization_amd as gtop
from grad_traj_optimization_amd import problem

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
mp = problem.make_map(200, density=0.02, seed=0)
ctx = gtop.GtopContext(0)
ctx.init_sdf_map(mp.map_size, mp.origin, mp.resolution)
ctx.update_sdf_map(mp.obstacle_points())
b = problem.make_trajectories(B, 6, mp, seed=1)
b = problem.permute(b, problem.spatial_order(b.waypoints, mp.origin, mp.map_size))
dev = torch.device("cuda:0")
x, Df, T = (torch.tensor(a, device=dev) for a in (b.x, b.Df.reshape(-1, 18), b.T))
cost, grad = ctx.eval_device(x, Df, T)
torch.cuda.synchronize()
K = 50
stamps = torch.tensor([2 ** 63 - 1, 0], dtype=torch.int64, device=dev)
gph = torch.cuda.CUDAGraph()
with torch.cuda.graph(gph):
    ctx.clock_stamp(stamps)
    for _ in range(K):
        ctx.eval_device(x, Df, T, cost, grad)
    ctx.clock_stamp(stamps)
t0 = time.perf_counter()
while time.perf_counter() - t0 < 0.05:      # sustained clocks
    for _ in range(20):
        gph.replay()
    torch.cuda.synchronize()
stamps.copy_(torch.tensor([2 ** 63 - 1, 0], dtype=torch.int64))
torch.cuda.synchronize()
gph.replay()
torch.cuda.synchronize()
st = stamps.tolist()
hz = ctx.clock_hz()
per_launch_us = (st[1] - st[0]) / hz / K * 1e6
L = gtop.load_library()
buf = np.zeros((4096, 16), dtype=np.uint64)
assert L.gtop_debug_read_stamps(buf.ctypes.data_as(C.c_void_p)) == 0
nb = min(4096, B)
s0, s1 = buf[:nb, 0].astype(np.int64), buf[:nb, 11].astype(np.int64)
w0, w1 = buf[:nb, 12].astype(np.int64), buf[:nb, 13].astype(np.int64)
life = s1 - s0
span_ticks = w1.max() - w0.min()
clock_ghz = float(np.sum(life)) / float(np.sum(w1 - w0)) * hz / 1e9
print(f"B={B}, the last of {K} launches of one graph replay (sustained clocks), {nb} wavefronts, lane 0 of each:")
print(f"  wavefront lifetime (first instruction -> stores acknowledged): median {np.median(life):.0f} shader cycles "
      f"(p10 {np.percentile(life, 10):.0f}, p90 {np.percentile(life, 90):.0f}, max {life.max()})")
print(f"  shader clock over those lifetimes: {clock_ghz:.2f} GHz (shader cycles / wall-clock ticks of {hz / 1e6:.0f} MHz)")
print(f"  median lifetime = {np.median(life) / clock_ghz * 1e-3:.3f} us;  start spread (first -> last wavefront start) "
      f"{(w0.max() - w0.min()) / hz * 1e6:.2f} us")
print(f"  KERNEL SPAN first start -> last end: {span_ticks / hz * 1e6:.2f} us (wall clock, 10 ns ticks)")
print(f"  the same graph by the bench's device-clock stamps (stamp kernel, {K} launches, stamp kernel): {per_launch_us:.3f} us per launch")
print(f"  => per-launch dispatch floor inside the graph = {per_launch_us - span_ticks / hz * 1e6:.2f} us "
      f"(tools/ubench/launch_floor2 measures it with an EMPTY kernel of the same grid)")
hw = buf[:nb, 14].astype(np.int64)
xcc = buf[:nb, 15].astype(np.int64) & 0xF
key = (xcc << 20) | (((hw >> 13) & 7) << 16) | (((hw >> 12) & 1) << 15) | (((hw >> 8) & 15) << 4) | ((hw >> 4) & 3)
u, c = np.unique(key, return_counts=True)
print(f"  placement: {nb} wavefronts on {len(u)} SIMDs, wavefronts-per-SIMD histogram {dict(zip(*[v.tolist() for v in np.unique(c, return_counts=True)]))}")
