#!/usr/bin/env python3
"""Diagnostic: the local map update (gtop_update_sdf_map_window_device: reset + mark + the three sweeps over the box +
the box's corner records) against the whole-map rebuild (gtop_update_sdf_map_device) on the bench maps.
usage: tools/window_time.py [grid ...]   (default 200 400)"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.getcwd())
import grad_traj_optimization_amd as gtop
from grad_traj_optimization_amd import problem


def timed(fn, reps=20):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


for g in [int(a) for a in sys.argv[1:]] or [200, 400]:
    mp = problem.make_map(g, density=0.02 if g <= 200 else 0.04, seed=0)
    ctx = gtop.GtopContext(0)
    ctx.init_sdf_map(mp.map_size, mp.origin, mp.resolution)
    pts = mp.obstacle_points()
    dp = torch.tensor(pts, device="cuda:0")
    full = timed(lambda: ctx.update_sdf_map_device(dp), 10)
    print(f"grid {g}^3: whole-map rebuild (points resident, both record precisions) {full:.1f} us", flush=True)
    centre = mp.origin + 0.5 * mp.map_size
    for box in ((10.0, 10.0, 5.0), (20.0, 20.0, 5.0), (20.0, 20.0, 20.0)):     # metres: a sensor's reach
        half = 0.5 * np.array(box)
        a, b = centre - half, centre + half
        sel = np.all((pts >= a) & (pts <= b), axis=1)
        dw = torch.tensor(pts[sel], device="cuda:0")
        us = timed(lambda: ctx.update_sdf_map_window_device(a, b, dw))
        # the same update captured once (its scratch exists after the eager calls) and replayed: one graph launch per frame
        gph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gph):
            ctx.update_sdf_map_window_device(a, b, dw)
        us_g = timed(gph.replay)
        vox = np.prod(np.ceil(np.array(box) / mp.resolution))
        print(f"grid {g}^3: window {box} m = {int(vox)} voxels ({100 * vox / g ** 3:.2f} % of the map), {int(sel.sum())} points: "
              f"{us:.1f} us eager, {us_g:.1f} us as a replayed hipGraph = {full / us_g:.1f}x faster than the whole map", flush=True)
    ctx.close()
