#!/usr/bin/env python3
"""Generates tests/golden/*.npz — inputs and expected outputs of the cost/gradient
callback.  The reference has no fixtures for this path and cannot be built or
run here (PARITY UNPINNED), so each vector is produced by the C restatement
(oracle/gtop_oracle.c) and accepted only if the independently written numpy
twin (oracle/np_twin.py) agrees to 1e-12 relative.  Run from the repo root:

    python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from grad_traj_optimization_amd import problem  # noqa: E402
from oracle import np_twin, oracle  # noqa: E402
from tests import scenes  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))
PKEYS = ["ws", "wc", "alpha", "r", "d0", "alpha_v", "r_v", "v0", "alpha_a", "r_a", "a0", "step", "enable_dyn"]


def pvec(p):
    return np.array([float(p[k]) for k in PKEYS])


def evaluate(T, Df, x, sdf, nsdf, p):
    """C oracle, cross-checked against the twin."""
    prm = oracle.make_params(**p)
    c, g, _ = oracle.eval_batch(T, Df, x, sdf, prm)
    for i in range(x.shape[0]):
        Ti = T[i] if T.ndim == 2 else T
        cn, gn, _ = np_twin.cost_grad(Ti, Df[i], x[i], nsdf, p)
        assert abs(c[i] - cn) <= 1e-12 * abs(cn), (c[i], cn)
        assert np.max(np.abs(g[i] - gn)) <= 1e-12 * np.max(np.abs(gn))
    return c, g


def case_opti_node():
    sdf = oracle.Sdf.from_map_size(scenes.OPTI_NODE_ORIGIN, scenes.OPTI_NODE_RES, scenes.OPTI_NODE_MAP_SIZE)
    sdf.build_from_points(scenes.opti_node_obstacles())
    nsdf = np_twin.Sdf(sdf.origin, sdf.resolution, sdf.grid, sdf.dist, max_range=list(sdf.c.max_range))
    T = oracle.segment_time(scenes.OPTI_NODE_PATH)
    Df, Dp = oracle.initial_d(scenes.OPTI_NODE_PATH)
    rng = np.random.default_rng(2024)
    x = np.stack([Dp.reshape(-1), Dp.reshape(-1) + rng.normal(0, 0.05, Dp.size),
                  Dp.reshape(-1) + rng.normal(0, 0.2, Dp.size)])
    Dfb = np.broadcast_to(Df, (3, 3, 6)).copy()
    sets = {"opti_node": {}, "step1": dict(step=1), "compare2": dict(ws=20.0, wc=1.0)}   # launch files, App. B
    out = dict(T=T, Df=Dfb, x=x, pkeys=np.array(PKEYS), set_names=np.array(list(sets)))
    for name, kw in sets.items():
        p = dict(oracle.OPTI_NODE_PARAMS)
        p.update(kw)
        c, g = evaluate(T, Dfb, x, sdf, nsdf, p)
        out[f"params_{name}"] = pvec(p)
        out[f"cost_{name}"] = c
        out[f"grad_{name}"] = g
    # the map is rebuilt from tests/scenes.py in the test; pin it with checksums and probes
    d = sdf.dist.reshape(sdf.grid)
    out["dist_sum"] = np.array([d.sum(), (d * d).sum()])
    probe_idx = np.array([(100, 100, 10), (101, 111, 3), (0, 0, 0), (199, 199, 24), (110, 112, 12), (90, 87, 20)])
    out["probe_idx"] = probe_idx
    out["probe_val"] = d[tuple(probe_idx.T)]
    np.savez_compressed(os.path.join(OUT, "opti_node_scene.npz"), **out)
    print("opti_node_scene: cost", out["cost_opti_node"])


def case_small_maps():
    mp = problem.make_map((24, 20, 16), density=0.05, seed=21, box_vox=(1, 4))
    sdf = oracle.Sdf.from_map_size(mp.origin, mp.resolution, mp.map_size)
    sdf.build_from_occupancy(mp.occupancy)
    nsdf = np_twin.Sdf(sdf.origin, sdf.resolution, sdf.grid, sdf.dist, max_range=list(sdf.c.max_range))
    out = dict(grid=np.array(mp.grid), origin=mp.origin, resolution=np.array(mp.resolution),
               map_size=mp.map_size, occupancy=mp.occupancy, dist=sdf.dist.reshape(mp.grid),
               pkeys=np.array(PKEYS))
    names = []

    def add(name, T, Df, x, p):
        c, g = evaluate(T, Df, x, sdf, nsdf, p)
        names.append(name)
        out[f"{name}_T"], out[f"{name}_Df"], out[f"{name}_x"] = T, Df, x
        out[f"{name}_params"], out[f"{name}_cost"], out[f"{name}_grad"] = pvec(p), c, g
        print(name, "cost", c)

    base = dict(oracle.OPTI_NODE_PARAMS)
    for m in (2, 3, 6, 12):
        b = problem.make_trajectories(4, m, mp, seed=30 + m, step_len=(0.5, 1.0), margin=0.4)
        add(f"m{m}", b.T, b.Df, b.x, base)
    b = problem.make_trajectories(4, 6, mp, seed=77, step_len=(0.5, 1.0), margin=0.4)
    add("m6_step1", b.T, b.Df, b.x, dict(base, step=1))
    add("m6_wc0", b.T, b.Df, b.x, dict(base, wc=0.0))
    add("m6_click", b.T, b.Df, b.x, dict(base, ws=20.0, wc=0.1, d0=0.7))             # click.launch, App. B
    add("m6_dyn", b.T, b.Df, b.x, dict(base, enable_dyn=1, alpha_v=2.0, alpha_a=1.5))  # the commented-out block
    add("m6_shared_T", b.T[0], b.Df, b.x, base)                                        # one time vector for the batch
    # non-zero velocity and acceleration at both ends in Df (a kinodynamic front end's rows: setKinoPath,
    # grad_traj_optimizer.cpp:35-65; replanning start state, qp_generator.cpp:425-431), 6 and 12 segments
    for m in (6, 12):
        bk = problem.make_trajectories(4, m, mp, seed=80 + m, step_len=(0.5, 1.0), margin=0.4, boundary="random")
        add(f"m{m}_kino", bk.T, bk.Df, bk.x, base)
    bk = problem.make_trajectories(4, 6, mp, seed=93, step_len=(0.5, 1.0), margin=0.4, boundary="random")
    add("m6_kino_dyn", bk.T, bk.Df, bk.x, dict(base, enable_dyn=1, alpha_v=2.0, alpha_a=1.5))
    # round 3's bodies: 9 segments (two wavefronts per trajectory for a batch this small), 17 and 30 (the chunked body past
    # 12 segments: two and three chunks), the latter with the dyn block as well
    for m in (9, 17, 30):
        bk = problem.make_trajectories(3, m, mp, seed=100 + m, step_len=(0.4, 0.8), margin=0.4, boundary="random")
        add(f"m{m}_kino", bk.T, bk.Df, bk.x, base)
    bk = problem.make_trajectories(3, 17, mp, seed=131, step_len=(0.4, 0.8), margin=0.4, boundary="random")
    add("m17_kino_dyn", bk.T, bk.Df, bk.x, dict(base, enable_dyn=1, alpha_v=2.0, r_v=4.0, alpha_a=1.5, r_a=15.0))
    # samples leaving the map: waypoints pushed through the boundary (dist = -1, grad = 0 convention)
    b2 = problem.make_trajectories(4, 4, mp, seed=78, step_len=(0.5, 1.0), margin=0.4)
    x2 = b2.x.copy()
    x2[:, 0] += 3.0          # x-position of waypoint 1, far past max_range
    x2[1, 3 * 3 * 2] = -1.5  # z-position of waypoint 1 below the floor
    add("m4_out_of_map", b2.T, b2.Df, x2, base)
    # tiny segment times: 29 samples at T = 0.03, none below 1e-3
    b3 = problem.make_trajectories(4, 3, mp, seed=79, step_len=(0.3, 0.5), margin=0.6)
    T3 = b3.T.copy()
    T3[0, 1] = 0.03
    T3[1, 0] = 0.02
    T3[2, 2] = 0.0009
    T3[3, 1] = 0.031
    add("m3_tiny_T", T3, b3.Df, b3.x, dict(base, ws=1e-6))
    out["case_names"] = np.array(names)
    np.savez_compressed(os.path.join(OUT, "small_maps.npz"), **out)


if __name__ == "__main__":
    case_opti_node()
    case_small_maps()
    for f in sorted(os.listdir(OUT)):
        if f.endswith(".npz"):
            print(f, os.path.getsize(os.path.join(OUT, f)), "bytes")
