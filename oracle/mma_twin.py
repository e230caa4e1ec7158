"""mma_twin.py — an INDEPENDENT restatement of the optimizer the reference runs around its callback, for tests.

TEST INFRASTRUCTURE ONLY (like everything under oracle/).  The reference hands `costFunc` to NLopt algorithm 24,
LD_MMA (src/grad_traj_optimizer.cpp:137-140, launch/opti_node.launch:3) with box bounds (:151-179) and nothing else —
no nonlinear constraints.  NLopt is a third-party dependency absent from this image (pinned by the reference's vendored
lib/libnlopt.so.0.9.0 = NLopt 2.5.0, which is not loaded), so the product restates the algorithm (csrc/mma.hpp on the
host, gtop_device_common.h on the device) and round 3's "oracle" for it was that same product header.  This file is a
second restatement of the PUBLISHED algorithm — K. Svanberg, "A class of globally convergent optimization methods
based on conservative convex separable approximations", SIAM J. Optim. 12 (2002) 555-573, in the form of NLopt 2.5.0's
src/algs/mma/mma.c and src/api/stop.c, summarised below — in another language, sharing no code with the product: it
neither includes, imports nor calls csrc/mma.hpp.  The two must agree to 1e-12 on every evaluation point, value, evaluation count and stop code (tests/test_mma_twin.py, goldens in
tests/golden/mma_traces.npz), and both must reproduce the hand-derived known answers there.  PARITY with NLopt's own
iterates stays UNPINNED (NLopt cannot be run here).

The algorithm, with zero nonlinear constraints (m = 0: the dual problem of mma.c is empty, every inner iteration is
the closed-form minimiser of the separable approximant):

  state     x (base point, the best point so far), f(x), df(x); trial point xcur; asymptote widths sigma_j;
            conservativeness rho; outer iteration counter k; xprev, xprevprev (trial points of the previous outers)
  start     sigma_j = (ub_j - lb_j)/2, or 1 if a bound is infinite; rho = 1; evaluate f at the start point
  outer     fprev = f(xcur); stop on the evaluation limit;  k += 1;  xprevprev = xprev (k > 1);  xprev = xcur
   inner    approximant about x:  g(y) = f(x) + sum_j [ df_j sigma_j^2 dx_j + (|df_j| sigma_j + rho/2) dx_j^2 ]
                                                       / (sigma_j^2 - dx_j^2),          dx = y - x
            its minimiser per coordinate (mma.c dual_func):  u = df_j sigma_j^2,  v = |df_j| sigma_j + rho/2,
                dx_j = (u/v) / (-1 - sqrt|1 - (u/(v sigma_j))^2|),
            then y_j clamped into [lb_j, ub_j] and into x_j -+ 0.9 sigma_j  (sigma_j = 0: y_j = x_j);
            w = sum_j dx_j^2 / (2 (sigma_j^2 - dx_j^2))
            evaluate f, df at xcur = y;  inner_done = g(xcur) >= f(xcur)
            if f(xcur) < f(x): x, f(x), df(x) := xcur, f(xcur), df(xcur)          (best point moves)
            stop on the evaluation limit (looked at after EVERY evaluation, before the inner loop may end)
            if inner_done: leave the inner loop
            if f(xcur) > g(xcur): rho = min(10 rho, 1.1 (rho + (f(xcur) - g(xcur)) / w))
   stop     ftol:  relstop(fprev, f(xcur));  xtol: relstop(xprev_j, xcur_j) for every j;  x after f, its verdict
            stands when both hold.  relstop(old, new) = |new - old| < tol (|new| + |old|)/2, or tol > 0 and new == old
            (stop.c; the absolute tolerances are never set by the reference)
   update   rho = max(rho/10, 1e-5);  for k > 1, per coordinate: s = (xcur - xprev)(xprev - xprevprev);
            sigma_j *= 0.7 if s < 0, 1.2 if s > 0;  then clamped to [0.01, 10] (ub_j - lb_j) when both bounds are finite

Sums over coordinates run in index order (as a C loop does), so that two restatements of the same arithmetic agree
to the last bits and an accept / reject decision cannot flip between them.
"""
import math

import numpy as np

FTOL_REACHED, XTOL_REACHED, MAXEVAL_REACHED = 3, 4, 5     # nlopt_result values
RHO_MIN = 1e-5


def relstop(old, new, tol):
    """NLopt 2.5.0 stop.c `relstop` with abstol = 0."""
    if math.isinf(old):
        return False
    return abs(new - old) < tol * (abs(new) + abs(old)) * 0.5 or (tol > 0 and new == old)


def separable_minimiser(x, fx, dfdx, sigma, rho, lb, ub):
    """The approximant's minimiser about base point x (closed form per coordinate), its value g and the weight w."""
    n = len(x)
    y = np.array(x, dtype=np.float64)
    g, w = float(fx), 0.0
    for j in range(n):
        s = float(sigma[j])
        if s == 0.0:
            continue
        s2 = s * s
        d = float(dfdx[j])
        v = abs(d) * s + 0.5 * rho
        u = d * s2
        r = u / (v * s)
        dx = (u / v) / (-1.0 - math.sqrt(abs(1.0 - r * r)))
        yj = float(x[j]) + dx
        if yj > ub[j]:
            yj = float(ub[j])
        elif yj < lb[j]:
            yj = float(lb[j])
        if yj > x[j] + 0.9 * s:
            yj = float(x[j]) + 0.9 * s
        elif yj < x[j] - 0.9 * s:
            yj = float(x[j]) - 0.9 * s
        y[j] = yj
        dx = yj - float(x[j])
        dx2 = dx * dx
        inv = 1.0 / (s2 - dx2)
        g += (d * (s2 * dx) + (abs(d) * s + 0.5 * rho) * dx2) * inv
        w += 0.5 * dx2 * inv
    return y, g, w


def minimize(f, x0, lb, ub, maxeval, ftol_rel=0.0, xtol_rel=0.0, start_outside="clamp", observe=None, observe_outer=None, trace=None):
    """f(x) -> (value, gradient).  Returns dict(x, minf, nevals, code, xs, fs): best point, its value, evaluations used,
    nlopt_result-style code, and the trace (every evaluation's point and value, in order).

    start_outside: NLopt 2.5.0 refuses a start point outside the box (NLOPT_INVALID_ARGS, optimize.c); the product
    clamps it into the box instead ("clamp", the default here so that the two can be compared; "reject" raises).
    observe(xcur, fcur, g, fbest): called after every inner evaluation with the trial point, its value, the
    approximant's value there and the best value before it — the two comparisons the road hangs on are g >= fcur
    (conservative: the inner loop ends) and fcur < fbest (the best point moves).
    observe_outer(xcur, xprev, xprevprev): called before every asymptote update (k > 1) — the third kind of decision
    the road hangs on: per coordinate the SIGN of (xcur - xprev)(xprev - xprevprev) picks the factor 0.7, 1 or 1.2."""
    lb = np.asarray(lb, dtype=np.float64)
    ub = np.asarray(ub, dtype=np.float64)
    x = np.array(x0, dtype=np.float64)
    n = x.size
    if np.any(x < lb) or np.any(x > ub):
        if start_outside == "reject":
            raise ValueError("start point outside the bounds (NLopt: NLOPT_INVALID_ARGS)")
        x = np.minimum(np.maximum(x, lb), ub)
    sigma = np.where(np.isinf(lb) | np.isinf(ub), 1.0, 0.5 * (ub - lb))
    rho = 1.0
    xs, fs = [], []

    def evaluate(p):
        v, g = f(p)
        xs.append(np.array(p, dtype=np.float64))
        fs.append(float(v))
        return float(v), np.array(g, dtype=np.float64)

    fx, dfdx = evaluate(x)
    fcur, xcur = fx, x.copy()
    xprev, xprevprev = xcur.copy(), xcur.copy()
    k, code = 0, 0
    while True:
        fprev = fcur
        if len(fs) >= maxeval:
            code = MAXEVAL_REACHED
            break
        k += 1
        if k > 1:
            xprevprev = xprev
        xprev = xcur
        while True:
            xcur, g, w = separable_minimiser(x, fx, dfdx, sigma, rho, lb, ub)
            fcur, dfcur = evaluate(xcur)
            inner_done = g >= fcur
            if trace is not None:      # (diagnostics: the state each trial point was made from)
                trace.append(dict(rho=rho, g=g, w=w, f=fcur, fbest=fx, k=k, sigma_min=float(np.min(sigma)), sigma_max=float(np.max(sigma))))
            if observe is not None:
                observe(xcur, fcur, g, fx)
            if fcur < fx:
                x, fx, dfdx = xcur.copy(), fcur, dfcur
            if len(fs) >= maxeval:
                code = MAXEVAL_REACHED
                break
            if inner_done:
                break
            if fcur > g:
                rho = min(10.0 * rho, 1.1 * (rho + (fcur - g) / w))
        if code:
            break
        if relstop(fprev, fcur, ftol_rel):
            code = FTOL_REACHED
        if xtol_rel > 0 and all(relstop(float(xprev[j]), float(xcur[j]), xtol_rel) for j in range(n)):
            code = XTOL_REACHED
        if code:
            break
        rho = max(0.1 * rho, RHO_MIN)
        if k > 1:
            if observe_outer is not None:
                observe_outer(xcur, xprev, xprevprev)
            for j in range(n):
                s = (xcur[j] - xprev[j]) * (xprev[j] - xprevprev[j])
                sigma[j] *= 0.7 if s < 0 else (1.2 if s > 0 else 1.0)
                if not (math.isinf(ub[j]) or math.isinf(lb[j])):
                    sigma[j] = min(sigma[j], 10.0 * (ub[j] - lb[j]))
                    sigma[j] = max(sigma[j], 0.01 * (ub[j] - lb[j]))
    return dict(x=x, minf=fx, nevals=len(fs), code=code, xs=np.array(xs), fs=np.array(fs))
