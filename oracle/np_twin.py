"""numpy twin of the oracle — TEST INFRASTRUCTURE ONLY (see gtop_oracle.c).

An independent restatement of the same reference lines, written with numpy
matrix algebra instead of C loops, used to cross-check gtop_oracle.c when the
golden fixtures are generated (tests/golden/make_golden.py) and in
tests/test_oracle.py.  Pure-Python loops: small cases only.
PARITY UNPINNED (the reference has no fixtures for this path).

Citations are file:line into /root/reference/.
"""
import math

import numpy as np


def segment_time(path, mean_v, init_time):
    """src/grad_traj_optimizer.cpp:73-81 (last segment gets no init_time)."""
    path = np.asarray(path, dtype=np.float64)
    m = len(path) - 1
    T = np.zeros(m)
    for i in range(m):
        ln = np.linalg.norm(path[i] - path[i + 1])
        T[i] = ln / mean_v + init_time if (i == 0 or i == m) else ln / mean_v
    return T


def generator(T):
    """src/qp_generator.cpp:181-197 (A), :223-236 (Q), :357-393 (Ct, L, R)."""
    T = np.asarray(T, dtype=np.float64)
    m = len(T)
    assert m >= 2
    n6, nd = 6 * m, 3 * m + 3
    A = np.zeros((n6, n6))
    Q = np.zeros((n6, n6))
    for k in range(m):
        for i in range(3):
            A[6 * k + 2 * i, 6 * k + i] = math.factorial(i)
            for j in range(i, 6):
                A[6 * k + 2 * i + 1, 6 * k + j] = (math.factorial(j) // math.factorial(j - i)) * T[k] ** (j - i)
        for i in range(3, 6):
            for j in range(3, 6):
                Q[6 * k + i, 6 * k + j] = (i * (i - 1) * (i - 2) * j * (j - 1) * (j - 2) // (i + j - 5)) * T[k] ** (i + j - 5)
    Ct = np.zeros((n6, nd))
    # global d = [start p,v,a | end p,v,a | wp1 p,v,a | ... | wp(m-1) p,v,a]
    # segment-local rows = [p(0), p(T), v(0), v(T), a(0), a(T)]
    def col(wp, der):  # waypoint 0..m, derivative 0..2
        if wp == 0:
            return der
        if wp == m:
            return 3 + der
        return 6 + 3 * (wp - 1) + der
    for s in range(m):
        for der in range(3):
            Ct[6 * s + 2 * der, col(s, der)] = 1
            Ct[6 * s + 2 * der + 1, col(s + 1, der)] = 1
    Ainv = np.linalg.inv(A)
    L = Ainv @ Ct
    R = Ct.T @ Ainv.T @ Q @ Ainv @ Ct
    return dict(A=A, Q=Q, Ct=Ct, L=L, R=R)


def initial_d(path, vel=(0, 0, 0), acc=(0, 0, 0)):
    """src/qp_generator.cpp:199-221 + :407-451 (straight-line init)."""
    path = np.asarray(path, dtype=np.float64)
    m = len(path) - 1
    Df = np.zeros((3, 6))
    Dp = np.zeros((3, 3 * m - 3))
    for a in range(3):
        Df[a] = [path[0, a], vel[a], acc[a], path[m, a], 0, 0]
        for k in range(1, m):
            Dp[a, 3 * (k - 1)] = path[k, a]
    return Df, Dp


class Sdf:
    """src/sdf_map.cpp:3-24 fields; dist indexed x*ny*nz + y*nz + z (:172-173)."""

    def __init__(self, origin, resolution, grid, dist, max_range=None):
        self.origin = np.asarray(origin, dtype=np.float64)
        self.res = float(resolution)
        self.res_inv = 1 / self.res
        self.grid = np.asarray(grid, dtype=np.int64)
        self.dist = np.asarray(dist, dtype=np.float64).reshape(tuple(self.grid))
        self.min_range = self.origin.copy()
        self.max_range = (self.origin + self.grid * self.res) if max_range is None else np.asarray(max_range, float)

    def in_map(self, pos):
        """:55-69"""
        return not (np.any(pos < self.min_range + 1e-4) or np.any(pos > self.max_range - 1e-4))

    def query(self, pos):
        """:185-242; out of map -> (-1, 0) (SURVEY A.4 Q4 convention)."""
        pos = np.asarray(pos, dtype=np.float64)
        if not self.in_map(pos):
            return -1.0, np.zeros(3)
        pos_m = pos - 0.5 * self.res * np.ones(3)
        idx = np.floor((pos_m - self.origin) * self.res_inv).astype(np.int64)
        idx_pos = (idx + 0.5) * self.res + self.origin
        diff = (pos - idx_pos) * self.res_inv
        v = np.zeros((2, 2, 2))
        for x in range(2):
            for y in range(2):
                for z in range(2):
                    c = np.clip(idx + (x, y, z), 0, self.grid - 1)
                    v[x, y, z] = self.dist[c[0], c[1], c[2]]
        dx, dy, dz = diff
        v00 = (1 - dx) * v[0, 0, 0] + dx * v[1, 0, 0]
        v01 = (1 - dx) * v[0, 0, 1] + dx * v[1, 0, 1]
        v10 = (1 - dx) * v[0, 1, 0] + dx * v[1, 1, 0]
        v11 = (1 - dx) * v[0, 1, 1] + dx * v[1, 1, 1]
        v0 = (1 - dy) * v00 + dy * v10
        v1 = (1 - dy) * v01 + dy * v11
        dist = (1 - dz) * v0 + dz * v1
        g = np.zeros(3)
        g[2] = (v1 - v0) * self.res_inv
        g[1] = ((1 - dz) * (v10 - v00) + dz * (v11 - v01)) * self.res_inv
        g0 = (1 - dz) * (1 - dy) * (v[1, 0, 0] - v[0, 0, 0])
        g0 += (1 - dz) * dy * (v[1, 1, 0] - v[0, 1, 0])
        g0 += dz * (1 - dy) * (v[1, 0, 1] - v[0, 0, 1])
        g0 += dz * dy * (v[1, 1, 1] - v[0, 1, 1])
        g[0] = g0 * self.res_inv
        return dist, g


def _edt_1d(f):
    """src/sdf_map.cpp:266-308 for one line (start=0, end=n-1)."""
    n = len(f)
    big = np.finfo(np.float64).max
    v = [0] * n
    z = [0.0] * (n + 1)
    k = 0
    z[0], z[1] = -big, big
    for q in range(1, n):
        k += 1
        while True:
            k -= 1
            with np.errstate(over="ignore", invalid="ignore"):
                s = ((f[q] + q * q) - (f[v[k]] + v[k] * v[k])) / (2 * q - 2 * v[k])
            if not (s <= z[k]):
                break
        k += 1
        v[k] = q
        z[k] = s
        z[k + 1] = big
    out = np.zeros(n)
    k = 0
    for q in range(n):
        while z[k + 1] < q:
            k += 1
        out[q] = (q - v[k]) * (q - v[k]) + f[v[k]]
    return out


def esdf_build(occ, resolution, prev=None):
    """src/sdf_map.cpp:310-368, full grid.  occ: (nx,ny,nz) of 0/1."""
    occ = np.asarray(occ)
    nx, ny, nz = occ.shape
    big = np.finfo(np.float64).max
    t0 = np.where(occ == 1, 0.0, big)
    t1 = np.zeros_like(t0)
    t2 = np.zeros_like(t0)
    for x in range(nx):
        for y in range(ny):
            t1[x, y, :] = _edt_1d(t0[x, y, :])
    for x in range(nx):
        for z in range(nz):
            t2[x, :, z] = _edt_1d(t1[x, :, z])
    out = np.full(occ.shape, 10000.0) if prev is None else np.array(prev, dtype=np.float64).reshape(occ.shape)
    for y in range(ny):
        for z in range(nz):
            out[:, y, z] = np.minimum(resolution * np.sqrt(_edt_1d(t2[:, y, z])), out[:, y, z])
    return out


def cost_grad(T, Df, x, sdf, p, gen=None):
    """src/grad_traj_optimizer.cpp:281-432.  p: dict of params (ws, wc, alpha,
    r, d0, step, enable_dyn, alpha_v, r_v, v0, alpha_a, r_a, a0)."""
    T = np.asarray(T, dtype=np.float64)
    m = len(T)
    ndp = 3 * m - 3
    gen = gen or generator(T)
    L, R = gen["L"], gen["R"]
    Rfp, Rpp = R[:6, 6:], R[6:, 6:]
    Df = np.asarray(Df, dtype=np.float64).reshape(3, 6)
    dp = np.asarray(x, dtype=np.float64).reshape(3, ndp)  # axis-major (:182-187)
    d = np.hstack([Df, dp])  # 3 x (6+ndp)

    cost_smooth = sum(float(d[a] @ R @ d[a]) for a in range(3))            # :326-327
    g_smooth = np.stack([2 * Rfp.T @ Df[a] + 2 * Rpp @ dp[a] for a in range(3)])  # :330-336
    coe = np.zeros((m, 18))                                                 # :253-279
    for a in range(3):
        coe[:, 6 * a:6 * a + 6] = (L @ d[a]).reshape(m, 6)

    V = np.zeros((6, 6))                                                    # :104-105 (Q2: rest = 0)
    for i in range(5):
        V[i, i + 1] = i + 1

    g_colli = np.zeros((3, ndp))
    g_vel = np.zeros((3, ndp))
    g_acc = np.zeros((3, ndp))
    cost_colli = cost_vel = cost_acc = 0.0
    nsamples = []
    for s in range(m):
        if abs(p["wc"]) < 1e-4:                                             # :346
            break
        Ldp = L[6 * s:6 * s + 6, 6:]                                        # :348
        dt = T[s] / 30.0                                                    # :351
        t = 1e-3
        cnt = 0
        while t < T[s]:                                                     # :353
            cnt += 1
            pw = np.array([math.pow(t, i) for i in range(6)])
            pos = np.zeros(3)
            vel = np.zeros(3)
            acc = np.zeros(3)
            for a in range(3):
                c = coe[s, 6 * a:6 * a + 6]
                # left-to-right sums, then the float round trip (:457-465, :477-485)
                pv = c[0] + c[1] * t + c[2] * math.pow(t, 2) + c[3] * math.pow(t, 3) + c[4] * math.pow(t, 4) + c[5] * math.pow(t, 5)
                vv = c[1] + 2 * c[2] * math.pow(t, 1) + 3 * c[3] * math.pow(t, 2) + 4 * c[4] * math.pow(t, 3) + 5 * c[5] * math.pow(t, 4)
                av = 2 * c[2] + 6 * c[3] * math.pow(t, 1) + 12 * c[4] * math.pow(t, 2) + 20 * c[5] * math.pow(t, 3)
                pos[a] = np.float64(np.float32(pv))
                vel[a] = np.float64(np.float32(vv))
                acc[a] = np.float64(np.float32(av))
            vn = math.sqrt(vel[0] * vel[0] + vel[1] * vel[1] + vel[2] * vel[2]) + 1e-5   # :358
            dist, g = sdf.query(pos)                                        # :363
            e = math.exp(-(dist - p["d0"]) / p["r"])
            cd = p["alpha"] * e                                             # :509
            gd = -(p["alpha"] / p["r"]) * e                                 # :514
            Tm = pw.reshape(1, 6)                                           # :544-551
            cost_colli += cd * vn * dt                                      # :373
            for k in range(3):                                              # :376-381
                g_colli[k] = g_colli[k] + ((gd * g[k] * cd * vn * Tm @ Ldp + cd * (vel[k] / vn) * Tm @ V @ Ldp) * dt).ravel()
            if p.get("enable_dyn", 0) and p["step"] == 2:                   # :383-407 (dead in the reference)
                cv = ca = 0.0
                for k in range(3):
                    cv = p["alpha_v"] * math.exp((abs(vel[k]) - p["v0"]) / p["r_v"])
                    cost_vel += cv * vn * dt
                    ca = p["alpha_a"] * math.exp((abs(acc[k]) - p["a0"]) / p["r_a"])
                    cost_acc += ca * vn * dt
                for k in range(3):
                    gv = (p["alpha_v"] / p["r_v"]) * math.exp((abs(vel[k]) - p["v0"]) / p["r_v"])
                    g_vel[k] = g_vel[k] + ((gv * vn * Tm @ V @ Ldp + cv * (vel[k] / vn) * Tm @ V @ Ldp) * dt).ravel()
                    ga = (p["alpha_a"] / p["r_a"]) * math.exp((abs(acc[k]) - p["a0"]) / p["r_a"])
                    g_acc[k] = g_acc[k] + ((ga * vn * Tm @ V @ V @ Ldp + ca * (vel[k] / vn) * Tm @ V @ Ldp) * dt).ravel()
            t += dt
        nsamples.append(cnt)

    ws = 0.0 if p["step"] == 1 else p["ws"]                                 # :412-415
    cost = ws * cost_smooth + p["wc"] * cost_colli + cost_vel + cost_acc + 1e-3     # :417-418
    grad = (ws * g_smooth + p["wc"] * g_colli + g_vel + g_acc) + 1e-5       # :425-432
    return cost, grad.reshape(-1), dict(coe=coe, nsamples=nsamples, cost_smooth=cost_smooth, cost_colli=cost_colli)
