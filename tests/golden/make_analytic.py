#!/usr/bin/env python3
"""Known answers derived BY HAND from the reference's formulas (tests/golden/ANALYTIC.md has the derivation),
independent of oracle/gtop_oracle.c, oracle/np_twin.py and the HIP kernels: no matrices are built or inverted
here and no field is interpolated.  Each case is chosen so that the quantities the reference computes
numerically have closed forms:

  A  constant distance field + straight constant-velocity path
       trilinear interpolation of a constant is the constant, its gradient 0 (sdf_map.cpp:221-239);
       the path's polynomials are p0 + v t, so jerk cost and jerk gradient vanish (qp_generator.cpp:226-234);
       cost = wc * alpha e^{(d0-D)/r} (|v| + 1e-5) sum_s 30 (T_s/30) + 1e-3   (grad_traj_optimizer.cpp:373, :417-418, :509)
       grad = wc * cd (v_k / vn) sum_s dt_s sum_i phi'_{s,j}(t_i) + 1e-5        (:376-381, second term only)
  B  distance field linear in x + the same path
       trilinear interpolation of a linear function is exact, gradient (b, 0, 0);
       cost and gradient are finite sums over the 30 sample times of explicit expressions, including the
       reference's extra cd factor in the distance-gradient term (:378).
  C  one cubic p(t) = t^3 along x over two segments, wc = 0
       the collision loop is skipped (:346); d'Rd is by definition the integral of the squared jerk
       (qp_generator.cpp:223-236), i.e. 36 (T_1 + T_2); the cubic is a stationary point, so the jerk gradient is 0:
       cost = ws * 36 (T_1 + T_2) + 1e-3, every gradient entry 1e-5   (:418, :429-431).

phi_{s,j} are the quintic Hermite basis functions of segment s (the columns of A_s^-1 for the derivative
ordering [p(0), p(T), v(0), v(T), a(0), a(T)] of qp_generator.cpp:185-195), written out below.

PARITY REMAINS UNPINNED: the reference cannot be built or run here, so these are answers to the reference's
FORMULAS as read from its source, not outputs of the reference.

usage: python tests/golden/make_analytic.py   (writes tests/golden/analytic.npz)"""
import math
import os
import struct
from fractions import Fraction as Fr

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))


def to_float32(v):
    """(double)(float)v — the reference stores pos/vel in float locals (grad_traj_optimizer.cpp:457-485)."""
    return struct.unpack("f", struct.pack("f", v))[0]


# d/dtau of the quintic Hermite basis on tau = t/T in [0, 1]; order [p0, pT, v0, vT, a0, aT].
# p(t) = p0 H0 + pT H1 + T (v0 H2 + vT H3) + T^2 (a0 H4 + aT H5) with
#   H0 = 1 - 10 tau^3 + 15 tau^4 - 6 tau^5      H1 = 10 tau^3 - 15 tau^4 + 6 tau^5
#   H2 = tau - 6 tau^3 + 8 tau^4 - 3 tau^5       H3 = -4 tau^3 + 7 tau^4 - 3 tau^5
#   H4 = tau^2/2 - 3/2 tau^3 + 3/2 tau^4 - tau^5/2   H5 = tau^3/2 - tau^4 + tau^5/2
def hermite(tau, T, order):
    """[d^order/dt^order of the six basis functions] at tau, scaled so that they multiply (p0,pT,v0,vT,a0,aT)."""
    c = [[1, 0, 0, -10, 15, -6], [0, 0, 0, 10, -15, 6], [0, 1, 0, -6, 8, -3], [0, 0, 0, -4, 7, -3],
         [0, 0, Fr(1, 2), Fr(-3, 2), Fr(3, 2), Fr(-1, 2)], [0, 0, 0, Fr(1, 2), -1, Fr(1, 2)]]
    scale = [1, 1, T, T, T * T, T * T]
    out = []
    for cj, sj in zip(c, scale):
        v = 0.0
        for p in range(order, 6):
            f = 1
            for q in range(order):
                f *= (p - q)
            v += float(cj[p]) * f * tau ** (p - order)
        out.append(v * sj / T ** order)
    return out


def sample_times(T):
    """for (t = 1e-3; t < T; t += dt), dt = T/30   (grad_traj_optimizer.cpp:351-353)"""
    dt = T / 30.0
    ts, t = [], 1e-3
    while t < T:
        ts.append(t)
        t += dt
    return ts, dt


def free_index(m, axis, wpt, der):
    """x[i + axis*num_dp], i = 3 (wpt-1) + der for interior waypoint wpt = 1..m-1   (:182-187)"""
    return axis * (3 * m - 3) + 3 * (wpt - 1) + der


def straight_line_case(m, p_start, v, T, field, prm):
    """Cases A and B.  field(px) -> (dist, d dist/dx): constant or linear in x.  Returns waypoints' state,
    cost, grad."""
    n = 9 * (m - 1)
    speed = math.sqrt(sum(c * c for c in v))
    wp = [[p_start[k] + v[k] * T * s for k in range(3)] for s in range(m + 1)]
    Df = np.zeros((3, 6))
    x = np.zeros(n)
    for k in range(3):
        Df[k] = [wp[0][k], v[k], 0.0, wp[m][k], v[k], 0.0]
        for w in range(1, m):
            x[free_index(m, k, w, 0)] = wp[w][k]
            x[free_index(m, k, w, 1)] = v[k]
    wc, alpha, r, d0 = prm["wc"], prm["alpha"], prm["r"], prm["d0"]
    cost_colli = 0.0
    g = np.zeros(n)
    for s in range(m):
        ts, dt = sample_times(T)
        assert len(ts) == 30
        for t in ts:
            pos = [to_float32(wp[s][k] + v[k] * t) for k in range(3)]      # c0 + c1 t, the other coefficients are 0
            vel = [to_float32(v[k]) for k in range(3)]
            vn = math.sqrt(sum(c * c for c in vel)) + 1e-5                  # :358
            dist, ddx = field(pos[0])
            e = math.exp(-(dist - d0) / r)
            cd, gd = alpha * e, -(alpha / r) * e                            # :509, :514
            cost_colli += cd * vn * dt                                      # :373
            grad_d = [ddx, 0.0, 0.0]
            h0 = hermite(t / T, T, 0)      # T * Ldp restricted to this segment's two waypoints  (:376)
            h1 = hermite(t / T, T, 1)      # T * V * Ldp
            for k in range(3):
                f_dist = gd * grad_d[k] * cd * vn     # the reference's formula, extra cd included (:378)
                f_vel = cd * vel[k] / vn
                for der in range(3):
                    # segment s ends at waypoint s+1 (entries pT, vT, aT = 1, 3, 5) and starts at waypoint s (0, 2, 4)
                    if 1 <= s + 1 <= m - 1:
                        g[free_index(m, k, s + 1, der)] += (f_dist * h0[2 * der + 1] + f_vel * h1[2 * der + 1]) * dt
                    if 1 <= s <= m - 1:
                        g[free_index(m, k, s, der)] += (f_dist * h0[2 * der] + f_vel * h1[2 * der]) * dt
    cost = wc * cost_colli + 1e-3                                           # jerk term = 0; :417-418
    grad = wc * g + 1e-5                                                    # :425-432
    return np.array(wp), np.full(m, T), Df, x, cost, grad, speed


def main():
    prm = dict(ws=1.0, wc=5.0, alpha=10.0, r=0.5, d0=0.8, step=2)          # launch/opti_node.launch:3-28
    out = {}
    grid, res, origin = (60, 30, 30), 0.2, (-6.0, -3.0, 0.0)
    out["grid"], out["resolution"], out["origin"] = np.array(grid), res, np.array(origin)
    out["pkeys"] = np.array(["ws", "wc", "alpha", "r", "d0", "step"])

    # --- A: constant field D = 1.3; v = (0.5, 0.25, 0.125) (exact in float), T = 2, m = 4
    D = 1.3
    m, v, T, p0 = 4, (0.5, 0.25, 0.125), 2.0, (-2.5, -1.0, 1.5)
    wp, Ts, Df, x, cost, grad, speed = straight_line_case(m, p0, v, T, lambda px: (D, 0.0), prm)
    closed = prm["wc"] * prm["alpha"] * math.exp((prm["d0"] - D) / prm["r"]) * (speed + 1e-5) * (m * 30 * (T / 30.0)) + 1e-3
    assert abs(closed - cost) <= 1e-13 * cost          # the finite sum IS the closed form
    out.update(A_const=D, A_T=Ts, A_Df=Df, A_x=x, A_cost=cost, A_cost_closed_form=closed, A_grad=grad,
               A_params=np.array([prm[k] for k in out["pkeys"]], dtype=float))

    # --- B: field a + b x_world with a = 2.0, b = 0.15 (> 0 over the whole map), same path
    a, b = 2.0, 0.15
    wp, Ts, Df, x, cost, grad, _ = straight_line_case(m, p0, v, T, lambda px: (a + b * px, b), prm)
    out.update(B_lin=np.array([a, b]), B_T=Ts, B_Df=Df, B_x=x, B_cost=cost, B_grad=grad,
               B_params=np.array([prm[k] for k in out["pkeys"]], dtype=float))

    # --- C: p(t) = t^3 along x over two segments (T1 = 1.25, T2 = 0.75), wc = 0
    T1, T2 = 1.25, 0.75
    pc = dict(prm, wc=0.0, ws=2.5)
    t_end = T1 + T2
    Df = np.zeros((3, 6))
    Df[0] = [0.0, 0.0, 0.0, t_end ** 3, 3 * t_end ** 2, 6 * t_end]
    Df[1] = [0.5, 0, 0, 0.5, 0, 0]          # y, z: constant (inside the map; irrelevant with wc = 0)
    Df[2] = [1.0, 0, 0, 1.0, 0, 0]
    x = np.zeros(9)
    x[0:3] = [T1 ** 3, 3 * T1 ** 2, 6 * T1]
    x[3], x[6] = 0.5, 1.0
    cost = pc["ws"] * 36.0 * (T1 + T2) + 1e-3
    out.update(C_T=np.array([T1, T2]), C_Df=Df, C_x=x, C_cost=cost, C_grad=np.full(9, 1e-5),
               C_params=np.array([pc[k] for k in out["pkeys"]], dtype=float))
    np.savez(os.path.join(HERE, "analytic.npz"), **out)
    print({k: (v if np.ndim(v) == 0 else np.shape(v)) for k, v in out.items()})


if __name__ == "__main__":
    main()
