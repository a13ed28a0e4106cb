"""CPU ORACLE (test infrastructure, not product code) for the box-constrained dual-QP hot path.

A NumPy/SciPy restatement of the reference algorithms, written as plain functions over a dense
fp64 Hessian.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this module; the product package `optiml_amd` never does.

Parity status: PINNED.  tests/test_oracle_golden.py checks every function below against the
fixtures under tests/golden/, which were produced by running the reference itself
(tools/gen_golden.py): the reference's own unit-test problems
(optiml/opti/constrained/tests/test_*.py, generator optiml/opti/utils.py:54), per-iteration
trajectories and converged SVC/SVR fits.

Reference sites restated here (all under /root/reference):
  objective        optiml/opti/_base.py:282 (value), :291 (gradient)
  start point      optiml/opti/constrained/_base.py:61-65 (lb defaults to 0, x0 = mid-box)
  projected grad   optiml/opti/constrained/projected_gradient.py:81-136
  frank-wolfe      optiml/opti/constrained/frank_wolfe.py:90-158
  active set       optiml/opti/constrained/active_set.py:84-230
  interior point   optiml/opti/constrained/interior_point.py:180-274
Every solver returns a dict: x, f_x, g_x, iter, status, f_hist (objective seen at the top of each
iteration, i.e. what the reference's callback observes), plus solver-specific traces.
"""
import numpy as np
from scipy.linalg import cho_factor, cho_solve
from scipy.sparse.linalg import minres

ACT_TOL = 1e-12   # bound-activity threshold used by PG and AS
CURV_TOL = 1e-16  # "no curvature along d" threshold used by PG and FW


def qp_value(Q, q, x):
    return 0.5 * x @ Q @ x + q @ x          # opti/_base.py:282


def qp_grad(Q, q, x):
    return Q @ x + q                        # opti/_base.py:291


def x_star(Q, q):
    """Unconstrained minimiser, opti/_base.py:259-269: Cholesky when Q is positive definite, else scipy's default minres."""
    try:
        return cho_solve(cho_factor(Q), -q), 'cholesky'
    except np.linalg.LinAlgError:
        return minres(Q, -q)[0], 'minres'


def _box(ub, lb, x0):
    ub = np.asarray(ub, dtype=float)
    lb = np.zeros_like(ub) if lb is None else np.asarray(lb, dtype=float)
    x = (lb + ub) / 2 if x0 is None else np.array(x0, dtype=float)
    return lb, ub, x


def _ratio_min(num, den, mask, start=np.inf):
    return min(start, min(num[mask] / den[mask], default=np.inf))


def projected_gradient(Q, q, ub, lb=None, x0=None, eps=1e-6, max_iter=1000, keep_x=(), trace=False):
    lb, ub, x = _box(ub, lb, x0)
    it, status, f_hist, xs, tr = 0, 'unknown', [], {}, []
    while True:
        f, g = qp_value(Q, q, x), qp_grad(Q, q, x)
        d = -g
        d[(ub - x <= ACT_TOL) & (d > 0)] = 0
        d[(x - lb <= ACT_TOL) & (d < 0)] = 0
        ng = np.linalg.norm(d)
        f_hist.append(f)
        if it in keep_x:
            xs[it] = x.copy()
        if ng <= eps:
            status = 'optimal'
            break
        if it >= max_iter:
            status = 'stopped'
            break
        max_t = _ratio_min(ub - x, d, d > 0)
        max_t = _ratio_min(lb - x, d, d < 0, max_t)
        den = d @ Q @ d
        t = max_t if den <= CURV_TOL else min(-(g @ d) / den, max_t)
        if trace:
            tr.append(dict(f=f, ng=ng, max_t=max_t, den=den, t=t, d=d.copy(), g=g.copy()))
        x += t * d
        it += 1
    return dict(x=x, f_x=f, g_x=g, iter=it, status=status, f_hist=np.array(f_hist), x_at=xs, trace=tr, ng=ng)


def frank_wolfe(Q, q, ub, lb=None, x0=None, eps=1e-6, max_iter=1000, t=0.0, keep_x=(), trace=False):
    if not 0 <= t < 1:
        raise ValueError('t has to lie in [0, 1)')
    lb, ub, x = _box(ub, lb, x0)
    it, status, f_hist, xs, tr = 0, 'unknown', [], {}, []
    best = -np.inf
    while True:
        f, g = qp_value(Q, q, x), qp_grad(Q, q, x)
        y = np.where(g < 0, ub, lb)
        low = f + g @ (y - x)
        if low > best:
            best = low
        gap = (f - best) / max(abs(f), 1)
        f_hist.append(f)
        if it in keep_x:
            xs[it] = x.copy()
        if gap <= eps:
            status = 'optimal'
            break
        if it >= max_iter:
            status = 'stopped'
            break
        if t > 0:
            r = t * (ub - lb)
            y = np.clip(y, x - r, x + r)
        d = y - x
        den = d @ Q @ d
        a = 1 if den <= CURV_TOL else min(-(g @ d) / den, 1)
        if trace:
            tr.append(dict(f=f, lower=low, best=best, gap=gap, den=den, a=a, d=d.copy(), g=g.copy()))
        x += a * d
        it += 1
    return dict(x=x, f_x=f, g_x=g, iter=it, status=status, f_hist=np.array(f_hist), x_at=xs, trace=tr,
                gap=gap, best_lb=best)


def interior_point(Q, q, ub, lb=None, x0=None, eps=1e-10, max_iter=1000, keep_x=(), trace=False):
    lb, ub, x = _box(ub, lb, x0)
    n = len(x)
    g = qp_grad(Q, q, x)
    lp = np.full(n, 1e-6)
    lm = np.full(n, 1e-6)
    pos = g >= 0
    lm[pos] += g[pos]
    lp[~pos] -= g[~pos]
    it, status, f_hist, xs, tr = 0, 'unknown', [], {}, []
    while True:
        f = qp_value(Q, q, x)
        xQx = x @ Q @ x
        p = -(lp @ ub) + lm @ lb - 0.5 * xQx
        gap = (f - p) / max(abs(f), 1)
        f_hist.append(f)
        if it in keep_x:
            xs[it] = x.copy()
        if gap <= eps:
            status = 'optimal'
            break
        if it >= max_iter:
            status = 'stopped'
            break
        mu = (f - p) / (4 * n * n)
        umx, xml = ub - x, x - lb
        hd = lp / umx + lm / xml
        H = Q + np.diag(hd)
        w = mu * (ub + lb - 2 * x) / (umx * xml) + lp - lm
        dx = cho_solve(cho_factor(H), w)
        dlp = (mu * np.ones(n) + lp * dx) / umx - lp
        dlm = (mu * np.ones(n) - lm * dx) / xml - lm
        max_t = np.inf
        m = dx < 0
        if m.any():
            max_t = min((lb[m] - x[m]) / dx[m])
        m = dx > 0
        if m.any():
            max_t = min(max_t, min(umx[m] / dx[m]))
        m = dlp < 0
        if m.any():
            max_t = min(max_t, min(-lp[m] / dlp[m]))
        m = dlm < 0
        if m.any():
            max_t = min(max_t, min(-lm[m] / dlm[m]))
        max_t *= 0.9995
        if trace:
            tr.append(dict(f=f, p=p, gap=gap, mu=mu, hd=hd, w=w, dx=dx.copy(), dlp=dlp, dlm=dlm, max_t=max_t))
        x += max_t * dx
        lp += max_t * dlp
        lm += max_t * dlm
        it += 1
    return dict(x=x, f_x=f, g_x=g, iter=it, status=status, f_hist=np.array(f_hist), x_at=xs, trace=tr,
                gap=gap, p=p, lp=lp, lm=lm)


def active_set(Q, q, ub, lb=None, x0=None, max_iter=1000, keep_x=(), trace=False):
    lb, ub, x = _box(ub, lb, x0)
    n = len(x)
    f = qp_value(Q, q, x)
    g = np.zeros(0)
    L = np.zeros(n, dtype=bool)
    U = np.zeros(n, dtype=bool)
    A = np.ones(n, dtype=bool)
    it, status, f_hist, xs, tr = 0, 'unknown', [], {}, []
    while True:
        f_hist.append(f)
        if it in keep_x:
            xs[it] = x.copy()
        if it >= max_iter:
            status = 'stopped'
            break
        cand = np.zeros_like(x)
        cand[U] = ub[U]
        cand[L] = lb[L]
        rhs = q[A] + Q[A, :][:, U] @ ub[U] + Q[A, :][:, L] @ lb[L]
        QAA = Q[A, :][:, A]
        used_minres = False
        try:
            cand[A] = cho_solve(cho_factor(QAA), -rhs)
        except Exception:   # singular / not PD (bare except in the reference, active_set.py:142)
            used_minres = True
            cand[A] = minres(np.inner(QAA, QAA), -(QAA.T @ rhs))[0]
        ev = dict(kind=None, used_minres=used_minres, nA=int(A.sum()))
        if ((cand[A] <= ub[A] + ACT_TOL) & (cand[A] >= lb[A] - ACT_TOL)).all():
            x = cand
            f, g = qp_value(Q, q, x), qp_grad(Q, q, x)
            h = np.nonzero(L & (g < -ACT_TOL))[0]
            from_upper = False
            if h.size == 0:
                h = np.nonzero(U & (g > ACT_TOL))[0]
                from_upper = True
            if h.size == 0:
                status = 'optimal'
                if trace:
                    ev['kind'] = 'optimal'
                    tr.append(ev)
                break
            h = h[0]
            A[h] = True
            (U if from_upper else L)[h] = False
            ev.update(kind='release', h=int(h), upper=from_upper)
        else:
            d = np.zeros_like(x)
            d[A] = cand[A] - x[A]
            max_t = _ratio_min(ub - x, d, A & (d > 0))
            max_t = _ratio_min(lb - x, d, A & (d < 0), max_t)
            x += max_t * d
            f = qp_value(Q, q, x)
            nL = A & (x <= lb + ACT_TOL)
            L[nL] = True
            A[nL] = False
            nU = A & (x >= ub - ACT_TOL)
            U[nU] = True
            A[nU] = False
            ev.update(kind='step', max_t=max_t, nL=int(nL.sum()), nU=int(nU.sum()))
        if trace:
            tr.append(ev)
        it += 1
    return dict(x=x, f_x=f, g_x=g, iter=it, status=status, f_hist=np.array(f_hist), x_at=xs, trace=tr,
                L=L, U=U, A=A)


SOLVERS = {'pg': projected_gradient, 'fw': frank_wolfe, 'ip': interior_point, 'as': active_set}
