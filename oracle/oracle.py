"""ctypes binding of oracle/gtop_oracle.c — TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
import this module (as the checker / reported CPU baseline).  The product
package grad_traj_optimization_amd never does.  PARITY UNPINNED — see the
header of gtop_oracle.c.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libgtop_oracle.so")


class OracleParams(C.Structure):
    _fields_ = [
        ("ws", C.c_double), ("wc", C.c_double),
        ("alpha", C.c_double), ("r", C.c_double), ("d0", C.c_double),
        ("alpha_v", C.c_double), ("r_v", C.c_double), ("v0", C.c_double),
        ("alpha_a", C.c_double), ("r_a", C.c_double), ("a0", C.c_double),
        ("step", C.c_int), ("enable_dyn", C.c_int),
    ]


class OracleSdf(C.Structure):
    _fields_ = [
        ("origin", C.c_double * 3), ("min_range", C.c_double * 3),
        ("max_range", C.c_double * 3),
        ("resolution", C.c_double), ("resolution_inv", C.c_double),
        ("grid", C.c_int * 3),
        ("dist", C.POINTER(C.c_double)),
    ]


def build(force=False):
    """Compile the restatement with gcc (no-op if up to date)."""
    src = os.path.join(_HERE, "gtop_oracle.c")
    if (not force and os.path.exists(_SO)
            and os.path.getmtime(_SO) >= os.path.getmtime(src)
            and os.path.getmtime(_SO) >= os.path.getmtime(os.path.join(_HERE, "gtop_oracle.h"))):
        return _SO
    subprocess.check_call(["make", "-C", _HERE, "-s", "_build/libgtop_oracle.so"])
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_SO)
        dp = C.POINTER(C.c_double)
        ip = C.POINTER(C.c_int)
        L.oracle_segment_time.argtypes = [C.c_int, dp, C.c_double, C.c_double, dp]
        L.oracle_segment_time.restype = None
        L.oracle_generator.argtypes = [C.c_int, dp, dp, dp, dp, dp, dp]
        L.oracle_generator.restype = C.c_int
        L.oracle_initial_d.argtypes = [C.c_int, dp, dp, dp, dp, dp]
        L.oracle_initial_d.restype = None
        L.oracle_sdf_init.argtypes = [C.POINTER(OracleSdf), dp, C.c_double, ip, dp]
        L.oracle_sdf_init.restype = None
        L.oracle_sdf_init_size.argtypes = [C.POINTER(OracleSdf), dp, C.c_double, dp, ip]
        L.oracle_sdf_init_size.restype = None
        L.oracle_sdf_query.argtypes = [C.POINTER(OracleSdf), dp, dp]
        L.oracle_sdf_query.restype = C.c_double
        L.oracle_edt_query.argtypes = [C.POINTER(OracleSdf), C.c_int, dp, dp, dp, dp, C.c_double, dp]
        L.oracle_edt_query.restype = C.c_double
        L.oracle_set_occupancy.argtypes = [C.POINTER(OracleSdf), dp, dp, C.c_int]
        L.oracle_set_occupancy.restype = C.c_int
        L.oracle_esdf_build.argtypes = [C.POINTER(OracleSdf), dp, dp]
        L.oracle_esdf_build.restype = None
        ip3 = C.POINTER(C.c_int)
        L.oracle_window_ids.argtypes = [C.POINTER(OracleSdf), dp, dp, ip3, ip3]
        L.oracle_window_ids.restype = None
        L.oracle_reset_window.argtypes = [C.POINTER(OracleSdf), dp, dp, dp, dp]
        L.oracle_reset_window.restype = None
        L.oracle_esdf_build_window.argtypes = [C.POINTER(OracleSdf), dp, dp, ip3, ip3]
        L.oracle_esdf_build_window.restype = None
        L.oracle_cost_grad.argtypes = [C.c_int, dp, dp, dp, dp, C.POINTER(OracleParams),
                                       C.POINTER(OracleSdf), dp, dp]
        L.oracle_cost_grad.restype = C.c_double
        L.oracle_eval_batch.argtypes = [C.c_int, C.c_int, dp, C.c_int, dp,
                                        C.POINTER(OracleParams), C.POINTER(OracleSdf),
                                        dp, dp, dp, C.c_int, C.c_int]
        L.oracle_eval_batch.restype = C.c_double
        L.oracle_traj_stats.argtypes = [C.c_int, dp, dp, C.c_double, dp]
        L.oracle_traj_stats.restype = None
        L.oracle_traj_samples.argtypes = [C.c_int, dp, dp, C.c_double, C.c_int, dp]
        L.oracle_traj_samples.restype = C.c_int
        L.oracle_edt_coarse.argtypes = [C.POINTER(OracleSdf), C.c_int, dp, dp, dp, dp, C.c_double]
        L.oracle_edt_coarse.restype = C.c_double
        L.oracle_coefficients.argtypes = [C.c_int, dp, dp, dp, dp]
        L.oracle_coefficients.restype = None
        _lib = L
    return _lib


def _p(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


# opti_node.launch:3-28 — the only parameter set whose names match the ctor
OPTI_NODE_PARAMS = dict(ws=1.0, wc=5.0, alpha=10.0, r=0.5, d0=0.8,
                        alpha_v=0.0, r_v=1.5, v0=2.5, alpha_a=0.0, r_a=1.5, a0=3.5,
                        step=2, enable_dyn=0)


def make_params(**kw):
    d = dict(OPTI_NODE_PARAMS)
    d.update(kw)
    return OracleParams(**d)


def segment_time(path, mean_v=1.8, init_time=0.3):
    path = _f64(path)
    T = np.zeros(path.shape[0] - 1)
    lib().oracle_segment_time(path.shape[0], _p(path), mean_v, init_time, _p(T))
    return T


def generator(T):
    """Returns dict(A, Q, Ct, L, R) for segment times T (m >= 2)."""
    T = _f64(T)
    m = T.shape[0]
    n6, nd = 6 * m, 3 * m + 3
    out = dict(A=np.zeros((n6, n6)), Q=np.zeros((n6, n6)), Ct=np.zeros((n6, nd)),
               L=np.zeros((n6, nd)), R=np.zeros((nd, nd)))
    rc = lib().oracle_generator(m, _p(T), _p(out["A"]), _p(out["Q"]), _p(out["Ct"]),
                                _p(out["L"]), _p(out["R"]))
    if rc != 0:
        raise ValueError("oracle_generator failed (m < 2 or singular A)")
    return out


def initial_d(path, vel=(0, 0, 0), acc=(0, 0, 0)):
    path = _f64(path)
    m = path.shape[0] - 1
    Df = np.zeros((3, 6))
    Dp = np.zeros((3, 3 * m - 3))
    lib().oracle_initial_d(path.shape[0], _p(path), _p(_f64(vel)), _p(_f64(acc)), _p(Df), _p(Dp))
    return Df, Dp


class Sdf:
    """Owns the distance buffer and the C descriptor."""

    def __init__(self, origin, resolution, grid, dist=None):
        self.grid = tuple(int(g) for g in grid)
        n = self.grid[0] * self.grid[1] * self.grid[2]
        if dist is None:
            dist = np.full(n, 10000.0)  # sdf_map.cpp:22
        self.dist = _f64(dist).reshape(-1)
        assert self.dist.size == n
        self.c = OracleSdf()
        g = (C.c_int * 3)(*self.grid)
        lib().oracle_sdf_init(C.byref(self.c), _p(_f64(origin)), float(resolution), g, _p(self.dist))
        self.origin = np.array(origin, dtype=np.float64)
        self.resolution = float(resolution)

    @classmethod
    def from_map_size(cls, origin, resolution, map_size):
        """sdf_map.cpp:3-24 — grid = ceil(size/res), max_range = origin + size."""
        tmp = OracleSdf()
        g = (C.c_int * 3)()
        lib().oracle_sdf_init_size(C.byref(tmp), _p(_f64(origin)), float(resolution),
                                   _p(_f64(map_size)), g)
        s = cls(origin, resolution, tuple(g))
        for i in range(3):
            s.c.max_range[i] = tmp.max_range[i]
        return s

    def query(self, pos):
        pos = _f64(pos)
        g = np.zeros(3)
        d = lib().oracle_sdf_query(C.byref(self.c), _p(pos), _p(g))
        return d, g

    def edt_query(self, pos, time, p0, vel, scale):
        """EDTEnvironment::evaluateEDTWithGrad for each (pos, time); boxes {p0, vel, scale}."""
        pos = _f64(pos).reshape(-1, 3)
        time = _f64(np.broadcast_to(time, (pos.shape[0],)))
        p0, vel, scale = (_f64(a).reshape(-1, 3) for a in (p0, vel, scale))
        d = np.empty(pos.shape[0])
        g = np.empty((pos.shape[0], 3))
        gi = np.zeros(3)
        for i in range(pos.shape[0]):
            d[i] = lib().oracle_edt_query(C.byref(self.c), p0.shape[0], _p(p0), _p(vel), _p(scale),
                                          _p(np.ascontiguousarray(pos[i])), float(time[i]), _p(gi))
            g[i] = gi
        return d, g

    def edt_coarse(self, pos, time, p0, vel, scale):
        """EDTEnvironment::evaluateCoarseEDT for each (pos, time)."""
        pos = _f64(pos).reshape(-1, 3)
        time = _f64(np.broadcast_to(time, (pos.shape[0],)))
        p0, vel, scale = (_f64(a).reshape(-1, 3) for a in (p0, vel, scale))
        return np.array([lib().oracle_edt_coarse(C.byref(self.c), p0.shape[0], _p(p0), _p(vel), _p(scale),
                                                 _p(np.ascontiguousarray(pos[i])), float(time[i]))
                         for i in range(pos.shape[0])])

    def build_from_points(self, pts):
        """updateSDFMap (grad_traj_optimizer.cpp:117-126): reset, mark, EDT."""
        occ = np.zeros(self.dist.size)
        self.dist[:] = 10000.0
        pts = _f64(pts).reshape(-1, 3)
        for p in pts:
            lib().oracle_set_occupancy(C.byref(self.c), _p(occ), _p(_f64(p)), 1)
        lib().oracle_esdf_build(C.byref(self.c), _p(occ), _p(self.dist))
        return occ

    def build_from_occupancy(self, occ):
        occ = _f64(occ).reshape(-1)
        self.dist[:] = 10000.0
        lib().oracle_esdf_build(C.byref(self.c), _p(occ), _p(self.dist))

    def window_ids(self, min_pos, max_pos):
        """The voxel window resetBuffer(min, max) / setUpdateRange compute (sdf_map.cpp:28-45, :244-260)."""
        lo, hi = (C.c_int * 3)(), (C.c_int * 3)()
        lib().oracle_window_ids(C.byref(self.c), _p(_f64(min_pos)), _p(_f64(max_pos)), lo, hi)
        return np.array(lo[:]), np.array(hi[:])

    def update_window(self, occ, min_pos, max_pos, pts):
        """The reference's local map update (compare2.cpp:147-152): resetBuffer(min, max), setOccupancy per point,
        setUpdateRange(min, max), updateESDF3d — on the persistent occupancy array `occ` (float64, one per voxel) and
        this object's distances.  Distances outside the window keep their values."""
        assert occ.dtype == np.float64 and occ.size == self.dist.size and occ.flags.c_contiguous
        mn, mx = _f64(min_pos), _f64(max_pos)
        lib().oracle_reset_window(C.byref(self.c), _p(occ), _p(self.dist), _p(mn), _p(mx))
        for p in _f64(pts).reshape(-1, 3):
            lib().oracle_set_occupancy(C.byref(self.c), _p(occ), _p(_f64(p)), 1)
        lo, hi = (C.c_int * 3)(), (C.c_int * 3)()
        lib().oracle_window_ids(C.byref(self.c), _p(mn), _p(mx), lo, hi)
        lib().oracle_esdf_build_window(C.byref(self.c), _p(occ), _p(self.dist), lo, hi)
        return np.array(lo[:]), np.array(hi[:])


def cost_grad(T, Df, x, sdf, params, L=None, R=None):
    """One callback evaluation (grad_traj_optimizer.cpp:554-562)."""
    T = _f64(T)
    m = T.shape[0]
    if L is None or R is None:
        g = generator(T)
        L, R = g["L"], g["R"]
    x = _f64(x)
    grad = np.zeros(9 * (m - 1))
    c = lib().oracle_cost_grad(m, _p(_f64(L)), _p(_f64(R)), _p(_f64(Df)), _p(T),
                               C.byref(params), C.byref(sdf.c), _p(x), _p(grad))
    return c, grad


def eval_batch(T, Df, x, sdf, params, reps=1, nthreads=1):
    """T: (B,m) or (m,) shared; Df: (B,3,6); x: (B,n).  Returns cost, grad, seconds."""
    x = _f64(x)
    B, n = x.shape
    m = n // 9 + 1
    T = _f64(T)
    stride = m if T.ndim == 2 else 0
    Df = _f64(Df).reshape(B, 18)
    cost = np.zeros(B)
    grad = np.zeros((B, n))
    sec = lib().oracle_eval_batch(B, m, _p(T), stride, _p(Df), C.byref(params),
                                  C.byref(sdf.c), _p(x), _p(cost), _p(grad), reps, nthreads)
    if sec < 0:
        raise ValueError("oracle_eval_batch failed")
    return cost, grad, sec


_opt_lib = None


def optimize_batch(T, Df, x0, lb, ub, sdf, params, max_evals, nthreads=1):
    """bench.py's CPU leg beside the device optimizer (oracle/cpu_optimizer.cpp): per trajectory a serial CCSA-MMA
    (csrc/mma.hpp standing in for NLopt's LD_MMA, grad_traj_optimizer.cpp:137-195) around the oracle callback.
    Returns (x, min_cost, nevals, seconds)."""
    _optimizer_lib()
    x = _f64(x0).copy()
    B, n = x.shape
    m = n // 9 + 1
    T = _f64(T)
    assert T.shape == (B, m)
    Df = _f64(Df).reshape(B, 18)
    lb, ub = _f64(lb), _f64(ub)
    cost = np.zeros(B)
    nev = np.zeros(B, dtype=np.int32)
    sec = _opt_lib.oracle_optimize_batch(B, m, _p(T), m, _p(Df), C.byref(params), C.byref(sdf.c), _p(x), _p(lb), _p(ub),
                                         int(max_evals), _p(cost), nev.ctypes.data_as(C.POINTER(C.c_int)), int(nthreads))
    if sec < 0:
        raise ValueError("oracle_optimize_batch failed")
    return x, cost, nev, sec


def _optimizer_lib():
    global _opt_lib
    if _opt_lib is None:
        lib()
        so = os.path.join(_HERE, "_build", "libgtop_cpu_optimizer.so")
        subprocess.check_call(["make", "-C", _HERE, "-s", "_build/libgtop_cpu_optimizer.so"])
        L = C.CDLL(so)
        dp, ip = C.POINTER(C.c_double), C.POINTER(C.c_int)
        L.oracle_optimize_batch.argtypes = [C.c_int, C.c_int, dp, C.c_int, dp, C.POINTER(OracleParams),
                                            C.POINTER(OracleSdf), dp, dp, dp, C.c_int, dp, ip, C.c_int]
        L.oracle_optimize_batch.restype = C.c_double
        L.oracle_mma_trace.argtypes = [C.c_int, dp, dp, C.POINTER(OracleParams), C.POINTER(OracleSdf), dp, dp, dp, C.c_int,
                                       C.c_double, C.c_double, dp, ip, dp, dp, C.c_int]
        L.oracle_mma_trace.restype = C.c_int
        L.oracle_mma_trace_quadratic.argtypes = [C.c_int, dp, dp, dp, dp, dp, C.c_int, C.c_double, C.c_double, dp, ip, dp,
                                                 dp, C.c_int]
        L.oracle_mma_trace_quadratic.restype = C.c_int
        _opt_lib = L
    return _opt_lib


def _trace_result(code, x, minf, nev, xs, fs):
    k = int(nev.value)
    return dict(x=x, minf=float(minf.value), nevals=k, code=int(code), xs=xs[:k].copy(), fs=fs[:k].copy())


def mma_trace(T, Df, x0, lb, ub, sdf, params, max_evals, ftol_rel=0.0, xtol_rel=0.0):
    """The product's host optimizer (csrc/mma.hpp) around the oracle callback for ONE trajectory, every evaluation
    recorded (oracle/cpu_optimizer.cpp oracle_mma_trace): dict(x, minf, nevals, code, xs, fs)."""
    L = _optimizer_lib()
    x = _f64(x0).copy().reshape(-1)
    n = x.size
    m = n // 9 + 1
    T, Df, lb, ub = _f64(T).reshape(m), _f64(Df).reshape(18), _f64(lb).reshape(n), _f64(ub).reshape(n)
    xs, fs = np.zeros((max_evals + 1, n)), np.zeros(max_evals + 1)
    minf, nev = C.c_double(), C.c_int()
    code = L.oracle_mma_trace(m, _p(T), _p(Df), C.byref(params), C.byref(sdf.c), _p(x), _p(lb), _p(ub), int(max_evals),
                              float(ftol_rel), float(xtol_rel), C.byref(minf), C.byref(nev), _p(xs), _p(fs), max_evals + 1)
    if code == -2:
        raise ValueError("oracle_mma_trace failed")
    return _trace_result(code, x, minf, nev, xs, fs)


def mma_trace_quadratic(a, c, x0, lb, ub, max_evals, ftol_rel=0.0, xtol_rel=0.0):
    """The same on f(x) = sum a_j (x_j - c_j)^2 / 2 (known answers of tests/test_mma_twin.py)."""
    L = _optimizer_lib()
    x = _f64(x0).copy().reshape(-1)
    n = x.size
    a, c, lb, ub = (_f64(v).reshape(n) for v in (a, c, lb, ub))
    xs, fs = np.zeros((max_evals + 1, n)), np.zeros(max_evals + 1)
    minf, nev = C.c_double(), C.c_int()
    code = L.oracle_mma_trace_quadratic(n, _p(a), _p(c), _p(x), _p(lb), _p(ub), int(max_evals), float(ftol_rel),
                                        float(xtol_rel), C.byref(minf), C.byref(nev), _p(xs), _p(fs), max_evals + 1)
    return _trace_result(code, x, minf, nev, xs, fs)


def coefficients(T, Df, x, L=None):
    """getCoefficientFromDerivative: (m, 18) coefficients, ascending powers per axis."""
    T = _f64(T)
    m = T.shape[0]
    if L is None:
        L = generator(T)["L"]
    coe = np.zeros((m, 18))
    lib().oracle_coefficients(m, _p(_f64(L)), _p(_f64(Df)), _p(_f64(x)), _p(coe))
    return coe


def traj_stats(coeff, T, dt_sample=0.01):
    """PolynomialTraj evaluation: [time_sum, length, jerk, mean_v, max_v, mean_a, max_a, acc_cost, n_samples]."""
    T = _f64(T)
    out = np.zeros(9)
    lib().oracle_traj_stats(T.shape[0], _p(_f64(coeff)), _p(T), float(dt_sample), _p(out))
    return out


def traj_samples(coeff, T, dt_sample=0.01, max_samples=4096):
    """PolynomialTraj::getTraj: (count, points (min(count, max_samples), 3))."""
    T = _f64(T)
    buf = np.zeros((max_samples, 3))
    n = lib().oracle_traj_samples(T.shape[0], _p(_f64(coeff)), _p(T), float(dt_sample), int(max_samples), _p(buf))
    return n, buf[:min(n, max_samples)]
