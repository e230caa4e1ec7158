"""Synthetic inputs for the batched cost/gradient path.

Host-side setup only (numpy): obstacle maps and batches of trajectories in the
layout the C-ABI takes.  The setup formulas mirror what the reference's
setPath does once per problem (src/grad_traj_optimizer.cpp:67-110 and
src/qp_generator.cpp:199-221, :407-451 of EpicOne1/grad_traj_optimization);
the per-iteration work is the HIP kernel's, not this file's.
"""
from dataclasses import dataclass

import numpy as np


@dataclass
class MapSpec:
    grid: tuple          # (nx, ny, nz)
    resolution: float
    origin: np.ndarray   # (3,)
    occupancy: np.ndarray  # (nx, ny, nz) uint8

    @property
    def map_size(self):
        """Metric size whose ceil(size/res) (src/sdf_map.cpp:9) is exactly
        `grid`: grid*res, nudged down by ulps where the division rounds up."""
        size = np.asarray(self.grid, dtype=np.float64) * self.resolution
        for i in range(3):
            while int(np.ceil(size[i] / self.resolution)) > self.grid[i]:
                size[i] = np.nextafter(size[i], 0.0)
        return size

    def obstacle_points(self):
        """Voxel centres of the occupied voxels (what updateSDFMap is fed)."""
        idx = np.argwhere(self.occupancy == 1)
        return (idx + 0.5) * self.resolution + self.origin


def make_map(grid, resolution=0.2, density=0.02, seed=0, box_vox=(3, 12), pillar_frac=0.5):
    """Random axis-aligned pillars and boxes until `density` of the voxels is
    occupied.  origin = (-X/2, -Y/2, 0) as the reference's scenes have it
    (src/opti_node.cpp:61)."""
    if np.isscalar(grid):
        grid = (int(grid),) * 3
    nx, ny, nz = (int(g) for g in grid)
    rng = np.random.default_rng(seed)
    occ = np.zeros((nx, ny, nz), dtype=np.uint8)
    target = density * occ.size
    filled = 0
    guard = 0
    while filled < target and guard < 100000:
        guard += 1
        sx, sy = rng.integers(box_vox[0], box_vox[1] + 1, size=2)
        x0 = rng.integers(0, max(1, nx - sx))
        y0 = rng.integers(0, max(1, ny - sy))
        if rng.random() < pillar_frac:      # floor-to-top pillar
            z0, sz = 0, int(nz * rng.uniform(0.5, 1.0))
        else:                                # floating box
            sz = int(rng.integers(box_vox[0], box_vox[1] + 1))
            z0 = int(rng.integers(0, max(1, nz - sz)))
        blk = occ[x0:x0 + sx, y0:y0 + sy, z0:z0 + sz]
        filled += int(blk.size - blk.sum())
        blk[...] = 1
    origin = np.array([-nx * resolution / 2, -ny * resolution / 2, 0.0])
    return MapSpec((nx, ny, nz), float(resolution), origin, occ)


def segment_times(waypoints, mean_v=1.8, init_time=0.3):
    """(B, m+1, 3) -> (B, m).  len/mean_v, + init_time on the FIRST segment
    only (the reference's `i == segment_time.size()` clause is unreachable,
    src/grad_traj_optimizer.cpp:73-81)."""
    wp = np.asarray(waypoints, dtype=np.float64)
    d = wp[..., :-1, :] - wp[..., 1:, :]
    ln = np.sqrt(d[..., 0] * d[..., 0] + d[..., 1] * d[..., 1] + d[..., 2] * d[..., 2])
    T = ln / mean_v
    T[..., 0] = ln[..., 0] / mean_v + init_time
    return T


def initial_derivatives(waypoints, start_vel=None, start_acc=None, end_vel=None, end_acc=None):
    """Straight-line initialisation (src/qp_generator.cpp:199-221, :407-451):
    Df (B,3,6) = [p_start, v_start, a_start, p_end, v_end, a_end] per axis
    (the reference's getInitialD leaves v_end = a_end = 0, :418-431; the C-ABI
    takes any row, include/gtop.h);
    Dp (B,3,3m-3): interior waypoint positions, zero velocity/acceleration."""
    wp = np.asarray(waypoints, dtype=np.float64)
    B, npts, _ = wp.shape
    m = npts - 1
    Df = np.zeros((B, 3, 6))
    Df[:, :, 0] = wp[:, 0, :]
    Df[:, :, 3] = wp[:, m, :]
    if start_vel is not None:
        Df[:, :, 1] = start_vel
    if start_acc is not None:
        Df[:, :, 2] = start_acc
    if end_vel is not None:
        Df[:, :, 4] = end_vel
    if end_acc is not None:
        Df[:, :, 5] = end_acc
    Dp = np.zeros((B, 3, 3 * m - 3))
    Dp[:, :, 0::3] = np.transpose(wp[:, 1:m, :], (0, 2, 1))
    return Df, Dp


@dataclass
class Batch:
    waypoints: np.ndarray  # (B, m+1, 3)
    T: np.ndarray          # (B, m)
    Df: np.ndarray         # (B, 3, 6)
    x: np.ndarray          # (B, 9(m-1))  axis-major free variables
    m: int


def make_trajectories(B, m, mapspec, seed=1, step_len=(1.0, 2.0), margin=1.0, noise=0.05,
                      mean_v=1.8, init_time=0.3, boundary=None, boundary_scale=(1.0, 1.5)):
    """Random-walk waypoints kept `margin` metres inside the map; x is the
    straight-line Dp plus N(0, noise^2) so velocities are non-zero.
    boundary="random": the fixed derivatives Df carry non-zero velocity and
    acceleration at BOTH ends, N(0, boundary_scale^2) m/s and m/s^2 — the rows a
    kinodynamic front end hands over (setKinoPath, src/grad_traj_optimizer.cpp:35-65;
    a replanning start state, src/qp_generator.cpp:425-431)."""
    rng = np.random.default_rng(seed)
    lo = mapspec.origin + margin
    hi = mapspec.origin + mapspec.map_size - margin
    wp = np.empty((B, m + 1, 3))
    wp[:, 0, :] = rng.uniform(lo, hi, size=(B, 3))
    for i in range(1, m + 1):
        d = rng.normal(size=(B, 3))
        d /= np.linalg.norm(d, axis=1, keepdims=True)
        p = wp[:, i - 1, :] + d * rng.uniform(step_len[0], step_len[1], size=(B, 1))
        # reflect at the margin box
        p = np.where(p < lo, 2 * lo - p, p)
        p = np.where(p > hi, 2 * hi - p, p)
        wp[:, i, :] = p
    T = segment_times(wp, mean_v, init_time)
    Df, Dp = initial_derivatives(wp)
    x = Dp.reshape(B, -1) + rng.normal(0.0, noise, size=(B, 9 * (m - 1)))
    if boundary == "random":        # (drawn after x: the waypoints and x of a seed do not change with the flag)
        sv, sa = boundary_scale
        Df, _ = initial_derivatives(wp, rng.normal(0.0, sv, (B, 3)), rng.normal(0.0, sa, (B, 3)),
                                    rng.normal(0.0, sv, (B, 3)), rng.normal(0.0, sa, (B, 3)))
    elif boundary is not None:
        raise ValueError("boundary must be None or 'random'")
    return Batch(wp, T, Df, x, m)


def spatial_order(waypoints, origin, map_size, bits=7):
    """Permutation that orders trajectories along a Morton (Z-order) curve of
    their centroids.  Trajectories that run together on an XCD then read the
    same region of the distance field (L2 locality); results are unchanged."""
    wp = np.asarray(waypoints, dtype=np.float64)
    c = wp.mean(axis=1)
    q = np.clip(((c - origin) / map_size * (1 << bits)).astype(np.int64), 0, (1 << bits) - 1)
    code = np.zeros(len(c), dtype=np.int64)
    for b in range(bits):
        for a in range(3):
            code |= ((q[:, a] >> b) & 1) << (3 * b + a)
    return np.argsort(code, kind="stable")


def permute(batch, perm):
    return Batch(batch.waypoints[perm], batch.T[perm], batch.Df[perm], batch.x[perm], batch.m)


def shard_range(B, rank, world_size):
    """Contiguous slice of the batch owned by `rank`."""
    per = (B + world_size - 1) // world_size
    lo = min(B, rank * per)
    return lo, min(B, lo + per)
