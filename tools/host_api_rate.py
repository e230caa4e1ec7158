#!/usr/bin/env python3
"""Diagnostic: throughput of the host-buffer entry gtop_eval_batch (PCIe copies of x in,
cost+grad out, synchronous) next to the resident-device entry.  usage: tools/host_api_rate.py [B ...]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.getcwd())
import grad_traj_optimization_amd as gtop
from grad_traj_optimization_amd import problem

mp = problem.make_map(200, density=0.02, seed=0)
ctx = gtop.GtopContext(0)
ctx.init_sdf_map(mp.map_size, mp.origin, mp.resolution)
ctx.update_sdf_map(mp.obstacle_points())
ctx.set_params()
ctx.set_launch_geometry(0, int(os.environ.get("SPL", "0")))
for B in [int(a) for a in sys.argv[1:]] or [1, 1024, 16384]:
    b = problem.make_trajectories(B, 6, mp, seed=1)
    ctx.set_problem(b.T, b.Df)
    ctx.eval_batch(b.x)
    reps = 200 if B <= 1024 else 50
    t0 = time.perf_counter()
    for _ in range(reps):
        ctx.eval_batch(b.x)
    dt = (time.perf_counter() - t0) / reps
    print(f"B={B}: gtop_eval_batch {dt * 1e6:.1f} us per call, {B / dt:.3e} evals/s (host buffers, PCIe both ways, synchronous)")
    if B == 1:
        t0 = time.perf_counter()
        for _ in range(reps):
            ctx.cost_nlopt(b.x[0])
        dt = (time.perf_counter() - t0) / reps
        print(f"B=1: gtop_cost_nlopt {dt * 1e6:.1f} us per call (the nlopt_func-shaped entry, through ctypes)")

# the rendezvous layer: N serial optimizers (C++ threads, csrc/mma.hpp) sharing launches, amortised per callback
import json
import subprocess
demo = os.path.join(os.getcwd(), "grad_traj_optimization_amd", "gtop_rendezvous_demo")
if os.path.exists(demo):
    for threads in (16, 64, 256, 1024):
        out = subprocess.run([demo, str(threads), "6", "30", "3" if threads > 256 else "0"], capture_output=True, text=True)
        if out.returncode != 0:
            print(f"rendezvous {threads}: failed {out.stderr[-200:]}")
            continue
        r = json.loads(out.stdout)
        print(f"rendezvous, {threads} threads: {r['shared_us_per_callback']:.2f} us per callback amortised "
              f"({r['shared_us_per_launch']:.1f} us per shared launch, {r['us_inside_launches_per_launch']:.1f} us of it inside "
              f"gtop_eval_batch) vs {r['serial_us_per_callback']:.2f} us per callback one after the other; identical = {r['identical']}")
