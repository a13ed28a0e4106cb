#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ by RUNNING the reference implementation.

Runs only in the build container (needs /root/reference, which never travels to the GPU box);
the committed outputs are plain data: inputs and the reference's outputs on them.

The reference's box-constrained solvers / SVC / SVR need only numpy + scipy + sklearn, but the
package import chain also pulls `autograd`, `qpsolvers`, `wurlitzer`, `cvxpy`, `casadi`, which are
not installed here and are never exercised on this path (`Quadratic` overrides jacobian/hessian;
its own `x_star()` is scipy's cho_solve / minres — only the box-constrained optimizers' `x_star`, which is never called,
would need qpsolvers).  They are satisfied by empty in-memory modules below; `autograd.numpy`
is a namespace view of the real numpy, so every arithmetic operation is numpy's own.
Nothing is written to /root/reference and no reference source is copied.

Usage:  python tools/gen_golden.py [--out tests/golden]
"""
import argparse
import contextlib
import os
import sys
import types

sys.dont_write_bytecode = True
import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)


def _install_placeholders():
    ag = types.ModuleType('autograd')
    ag.jacobian = lambda f: (lambda *a, **k: (_ for _ in ()).throw(RuntimeError('autograd absent')))
    ag.hessian = ag.jacobian
    agn = types.ModuleType('autograd.numpy')
    agn.__dict__.update({k: v for k, v in np.__dict__.items() if not k.startswith('__')})
    agn.random, agn.linalg = np.random, np.linalg   # lazily loaded numpy submodules
    ag.numpy = agn
    sys.modules['autograd'] = ag
    sys.modules['autograd.numpy'] = agn

    def _absent(name):
        def f(*a, **k):
            raise RuntimeError(name + ' is not available in this container')
        return f

    qp = types.ModuleType('qpsolvers')
    qp.solve_qp = _absent('qpsolvers')
    sys.modules['qpsolvers'] = qp

    wz = types.ModuleType('wurlitzer')
    wz.pipes = lambda *a, **k: contextlib.nullcontext()
    wz.STDOUT = object()
    sys.modules['wurlitzer'] = wz

    cp = types.ModuleType('cvxpy')
    for nm in ('Variable', 'Problem', 'Minimize', 'sum_squares', 'sum', 'pos', 'multiply', 'square', 'abs'):
        setattr(cp, nm, _absent('cvxpy'))
    cperr = types.ModuleType('cvxpy.error')
    cperr.SolverError = type('SolverError', (Exception,), {})
    cperr.DCPError = type('DCPError', (Exception,), {})
    cp.error = cperr
    cp.SolverError = cperr.SolverError
    cp.DCPError = cperr.DCPError
    sys.modules['cvxpy'] = cp
    sys.modules['cvxpy.error'] = cperr

    ca = types.ModuleType('casadi')
    ca.ldl = _absent('casadi')
    ca.ldl_solve = _absent('casadi')
    sys.modules['casadi'] = ca


_install_placeholders()
sys.path.insert(0, '/root/reference')

from optiml.opti import Quadratic  # noqa: E402
from optiml.opti.constrained import ProjectedGradient, FrankWolfe, ActiveSet, InteriorPoint  # noqa: E402
from optiml.opti.utils import generate_box_constrained_quadratic  # noqa: E402
from optiml.ml.svm import SVC, SVR  # noqa: E402
from optiml.ml.svm.kernels import (LinearKernel, PolyKernel, GaussianKernel,  # noqa: E402
                                   LaplacianKernel, SigmoidKernel, linear, gaussian, laplacian, sigmoid)
from optiml.ml.svm.losses import hinge, epsilon_insensitive  # noqa: E402

from optiml.opti.constrained import AugmentedLagrangianQuadratic  # noqa: E402
from optiml.opti.unconstrained.stochastic import (StochasticGradientDescent, Adam, AMSGrad, AdaMax,  # noqa: E402
                                                  AdaGrad, AdaDelta, RMSProp)

from optiml_amd.datasets import make_blobs, make_regression  # noqa: E402

SOLVERS = {'pg': ProjectedGradient, 'fw': FrankWolfe, 'as': ActiveSet, 'ip': InteriorPoint}


class Recorder:
    """Callback that snapshots (iter, f_x) every call and x at chosen iteration numbers."""

    def __init__(self, keep_x_at=(), keep_all_x=False):
        self.keep = set(keep_x_at)
        self.keep_all = keep_all_x
        self.f = []
        self.x = {}

    def __call__(self, opt):
        self.f.append(float(opt.f_x))
        if self.keep_all or opt.iter in self.keep:
            self.x[opt.iter] = np.array(opt.x, dtype=float, copy=True)


def run_solver(cls, Q, q, ub, lb=None, x0=None, keep=(), keep_all=False, **kw):
    rec = Recorder(keep, keep_all)
    opt = cls(quad=Quadratic(Q, q), ub=ub, lb=lb, x=None if x0 is None else x0.copy(), callback=rec, **kw)
    opt.minimize()
    out = {'x': np.array(opt.x, dtype=float), 'f_x': float(opt.f_x), 'iter': int(opt.iter),
           'status': str(opt.status), 'f_hist': np.array(rec.f)}
    if rec.x:
        ks = sorted(rec.x)
        out['x_iters'] = np.array(ks)
        out['x_at'] = np.stack([rec.x[k] for k in ks])
    return out


def flat(prefix, d):
    return {prefix + '_' + k: v for k, v in d.items()}


def gen_unit_problems(out):
    data = {}
    # optiml/opti/constrained/tests/test_{projected_gradient,frank_wolfe,active_set,interior_point}.py
    Q, q, ub = generate_box_constrained_quadratic(ndim=2)
    data.update(nd2_Q=Q, nd2_q=q, nd2_ub=ub, nd2_lb=np.zeros_like(ub))
    for s, cls in SOLVERS.items():
        data.update(flat('nd2_' + s, run_solver(cls, Q, q, ub)))
    # optiml/opti/constrained/tests/test_lower_bound.py
    Q, q, ub = generate_box_constrained_quadratic(ndim=5, seed=7)
    lb = ub / 4
    data.update(nd5_Q=Q, nd5_q=q, nd5_ub=ub, nd5_lb=lb)
    for s, cls in SOLVERS.items():
        data.update(flat('nd5_' + s, run_solver(cls, Q, q, ub, lb=lb, keep_all=(s != 'fw'))))
    # a larger generator instance with a non-trivial active set (same generator, other size/seed)
    Q, q, ub = generate_box_constrained_quadratic(ndim=64, seed=11)
    data.update(nd64_Q=Q, nd64_q=q, nd64_ub=ub, nd64_lb=np.zeros_like(ub))
    for s, cls in SOLVERS.items():
        data.update(flat('nd64_' + s, run_solver(cls, Q, q, ub, keep=(1, 2, 5, 10, 50, 100))))
    np.savez_compressed(os.path.join(out, 'unit_problems.npz'), **data)


def gen_x_star(out):
    """Quadratic.x_star() / f_star() of the reference (optiml/opti/_base.py:259-273): the Cholesky branch on the strictly convex
    generator problems and an RBF SVC dual, the minres branch on Hessians that are singular by construction (linear-kernel SVC
    dual: rank <= d + 1; SVR dual [[K,-K],[-K,K]] + ee': rank <= n + 1) and on an indefinite one (deterministic failure)."""
    from scipy.linalg import cho_factor
    from scipy.sparse.linalg import minres
    cases = {}
    for tag, kw in (('nd2', dict(ndim=2)), ('nd5', dict(ndim=5, seed=7)), ('nd64', dict(ndim=64, seed=11))):
        Q, q, _ = generate_box_constrained_quadratic(**kw)
        cases[tag] = (Q, q)
    X, y = make_blobs(200, 8, seed=3)
    K = GaussianKernel(gamma='scale')(X)
    cases['rbf_svc200'] = (K * np.outer(y, y) + np.outer(y, y), -np.ones(200))
    X, y = make_blobs(80, 5, seed=4)
    K = LinearKernel()(X)
    cases['lin_svc80'] = (K * np.outer(y, y) + np.outer(y, y), -np.ones(80))
    X, t = make_regression(40, 6, seed=5)
    K = GaussianKernel(gamma='scale')(X)
    e = np.hstack((np.ones(40), -np.ones(40)))
    cases['rbf_svr40'] = (np.vstack((np.hstack((K, -K)), np.hstack((-K, K)))) + np.outer(e, e), np.hstack((-t, t)) + 0.1)
    rs = np.random.RandomState(6)
    A = rs.standard_normal((48, 48))
    w = np.linspace(-1.0, 3.0, 48)
    U, _ = np.linalg.qr(A)
    Qi = (U * w) @ U.T
    cases['indef48'] = ((Qi + Qi.T) / 2, rs.standard_normal(48))
    data = {}
    for tag, (Q, q) in cases.items():
        quad = Quadratic(Q, q)
        x = np.array(quad.x_star(), dtype=float)
        f = float(quad.f_star())
        try:
            cho_factor(Q)
            method = 'cholesky'
            its = 0
        except np.linalg.LinAlgError:
            method = 'minres'
            cnt = [0]
            minres(Q, -q, callback=lambda xk: cnt.__setitem__(0, cnt[0] + 1))
            its = cnt[0]
        data.update({f'{tag}_Q': Q, f'{tag}_q': q, f'{tag}_x_star': x, f'{tag}_f_star': f, f'{tag}_method': method,
                     f'{tag}_minres_iters': its})
        print(f'  {tag}: {method} ({its} minres iterations), f* = {f:.12g}, |Qx+q| = {np.linalg.norm(Q @ x + q):.3e}')
    np.savez_compressed(os.path.join(out, 'x_star.npz'), **data)


def gen_kernels(out):
    rs = np.random.RandomState(0)
    X = 1.7 * rs.standard_normal((64, 8)) + 0.3
    Y = 0.9 * rs.standard_normal((16, 8)) - 0.2
    data = {'X': X, 'Y': Y}
    data['linear_XX'] = linear(X)
    data['linear_YX'] = linear(Y, X)
    pk = PolyKernel(degree=3, gamma='scale', coef0=1.)
    data['poly3_scale_c1_XX'] = pk(X)
    data['poly3_scale_c1_YX'] = pk(Y, X)
    data['poly3_default_XX'] = PolyKernel()(X)
    data['poly2_g05_c2_XX'] = PolyKernel(degree=2, gamma=0.5, coef0=2.)(X)
    data['rbf_scale_XX'] = gaussian(X)
    data['rbf_scale_YX'] = gaussian(Y, X)
    data['rbf_auto_XX'] = GaussianKernel(gamma='auto')(X)
    data['rbf_g037_XX'] = GaussianKernel(gamma=0.37)(X)
    data['gamma_scale_X'] = 1. / (X.shape[1] * X.var())
    data['gamma_scale_Y'] = 1. / (Y.shape[1] * Y.var())
    # float32 input: the reference (via sklearn's euclidean_distances) up-casts chunk-wise
    X32 = X.astype(np.float32)
    data['rbf_scale_XX_f32in'] = np.asarray(gaussian(X32), dtype=np.float64)
    np.savez_compressed(os.path.join(out, 'kernels.npz'), **data)


def gen_kernels_more(out):
    """the two remaining kernel functors (kernels.py:132-201) + one fit through each."""
    rs = np.random.RandomState(0)
    X = 1.7 * rs.standard_normal((64, 8)) + 0.3
    Y = 0.9 * rs.standard_normal((16, 8)) - 0.2
    data = {'X': X, 'Y': Y}
    data['laplacian_scale_XX'] = laplacian(X)
    data['laplacian_scale_YX'] = laplacian(Y, X)
    data['laplacian_g02_XX'] = LaplacianKernel(gamma=0.2)(X)
    data['sigmoid_scale_XX'] = sigmoid(X)
    data['sigmoid_auto_c05_YX'] = SigmoidKernel(gamma='auto', coef0=0.5)(Y, X)
    Xf, yf = make_blobs(300, 7, seed=77, sigma=6.0)
    Xte, _ = make_blobs(32, 7, seed=78, sigma=6.0)
    data.update(fit_X=Xf, fit_y=yf, fit_Xtest=Xte)
    est = SVC(loss=hinge, kernel=laplacian, C=1., reg_intercept=True, dual=True, optimizer=InteriorPoint).fit(Xf, yf)
    data.update(flat('laplacian_ip', _fit_record(est, Xte)))
    np.savez_compressed(os.path.join(out, 'kernels_more.npz'), **data)


def gen_trajectories(out):
    keep = (1, 2, 3, 10, 100, 500, 1000)
    # --- SVC, RBF, n=256 (dual dim 256) — svm/_base.py:552-559,628-629
    X, y = make_blobs(256, 16, seed=3)
    C = 1.0
    K = gaussian(X)
    Q = K * np.outer(y, y)
    Q += np.outer(y, y)
    q = -np.ones(len(y))
    ub = np.ones(len(y)) * C
    data = {'X': X, 'y': y, 'C': C, 'Q': Q, 'q': q, 'ub': ub}
    data.update(flat('pg', run_solver(ProjectedGradient, Q, q, ub, keep=keep)))
    data.update(flat('fw', run_solver(FrankWolfe, Q, q, ub, keep=keep)))
    data.update(flat('fwt', run_solver(FrankWolfe, Q, q, ub, keep=keep, t=0.1)))
    data.update(flat('ip', run_solver(InteriorPoint, Q, q, ub, keep_all=True)))
    data.update(flat('as', run_solver(ActiveSet, Q, q, ub, keep_all=True, max_iter=5000)))
    # warm start + general lb (the ctor's x= and lb= arguments)
    lb = 0.05 * ub
    x0 = np.linspace(0.1, 0.9, len(y))
    for s, cls in SOLVERS.items():
        data.update(flat('lbx0_' + s, run_solver(cls, Q, q, ub, lb=lb, x0=x0, keep=keep, max_iter=3000)))
    data['lbx0_lb'] = lb
    data['lbx0_x0'] = x0
    np.savez_compressed(os.path.join(out, 'traj_svc_rbf_n256.npz'), **data)

    # --- SVR, poly(3, scale, coef0=1), n=128 (dual dim 256) — svm/_base.py:1096-1104,1178
    X, y = make_regression(128, 8, seed=5)
    eps_ins = 0.1
    K = PolyKernel(degree=3, gamma='scale', coef0=1.)(X)
    Q = np.vstack((np.hstack((K, -K)), np.hstack((-K, K))))
    q = np.hstack((-y, y)) + eps_ins
    ub = np.ones(2 * len(y)) * C
    e = np.hstack((np.ones(len(y)), -np.ones(len(y))))
    Q += np.outer(e, e)
    data = {'X': X, 'y': y, 'C': C, 'epsilon': eps_ins, 'Q': Q, 'q': q, 'ub': ub}
    data.update(flat('pg', run_solver(ProjectedGradient, Q, q, ub, keep=keep)))
    data.update(flat('fw', run_solver(FrankWolfe, Q, q, ub, keep=keep)))
    data.update(flat('ip', run_solver(InteriorPoint, Q, q, ub, keep_all=True)))
    data.update(flat('as', run_solver(ActiveSet, Q, q, ub, keep_all=True, max_iter=5000)))
    np.savez_compressed(os.path.join(out, 'traj_svr_poly_n128.npz'), **data)


def gen_pg_converged(out):
    """ProjectedGradient run by the reference to its own stop test (|d| <= eps, status 'optimal') on two RBF SVC duals whose Hessian is
    well enough conditioned for the method to get there (the n = 256 trajectory problem is not: beyond k ~ 300 the reference's iterates
    depend on rounding): the converged alpha is path-independent, so it can be held to rtol 1e-6 — svm/_base.py:552-559, 628-629,
    projected_gradient.py:76-143.  eps = 1e-8: 471 and 364 iterations; at 1e-9 the reference itself never stops (rounding keeps |d| near 1e-8)."""
    data = {}
    for tag, (n, d, sigma, gamma, C) in {'a': (300, 6, 6.0, 0.5, 0.1), 'b': (500, 8, 3.0, 2.0, 1.0)}.items():
        X, y = make_blobs(n, d, seed=300 + n, sigma=sigma)
        K = GaussianKernel(gamma=gamma)(X)
        Q = K * np.outer(y, y)
        Q += np.outer(y, y)
        q = -np.ones(n)
        ub = np.ones(n) * C
        r = run_solver(ProjectedGradient, Q, q, ub, eps=1e-8, max_iter=20000)
        print(f'  pg converged {tag}: n={n} gamma={gamma} C={C}: iter={r["iter"]} status={r["status"]} f={r["f_x"]:.12f} '
              f'nsv={(r["x"] > 1e-6).sum()} at ub={(r["x"] > C - 1e-9).sum()}')
        data.update({f'{tag}_X': X, f'{tag}_y': y, f'{tag}_gamma': gamma, f'{tag}_C': C, f'{tag}_eps': 1e-8})
        data.update(flat(tag, r))
    np.savez_compressed(os.path.join(out, 'pg_converged.npz'), **data)


def _fit_record(est, Xtest):
    opt = est.optimizer
    rec = {'alphas': np.asarray(est.alphas_, dtype=float), 'support': np.asarray(est.support_),
           'dual_coef': np.asarray(est.dual_coef_, dtype=float), 'intercept': float(est.intercept_),
           'iter': int(opt.iter), 'status': str(opt.status), 'f_x': float(opt.f_x),
           'loss_hist': np.asarray(est.train_loss_history, dtype=float),
           'decision': np.asarray(est.decision_function(Xtest), dtype=float)}
    if hasattr(est, 'coef_') and np.size(est.coef_):
        rec['coef'] = np.asarray(est.coef_, dtype=float)
    return rec


def gen_fits(out):
    for n, d in ((200, 6), (600, 10)):
        X, y = make_blobs(n, d, seed=100 + n, sigma=6.0)
        Xte, _ = make_blobs(32, d, seed=900 + n, sigma=6.0)
        data = {'X': X, 'y': y, 'Xtest': Xte}
        for kname, kern in (('rbf', gaussian), ('linear', linear)):
            for s, cls in SOLVERS.items():
                if kname == 'linear' and s == 'as' and n > 200:
                    continue  # singular Q_AA -> minres-on-normal-equations path; kept to the small case
                mi = 5000 if s == 'as' else 1000
                est = SVC(loss=hinge, kernel=kern, C=1., reg_intercept=True, dual=True, optimizer=cls, max_iter=mi)
                est.fit(X, y)
                data.update(flat(f'{kname}_{s}', _fit_record(est, Xte)))
                print(f'  svc n={n} {kname} {s}: iter={est.optimizer.iter} status={est.optimizer.status} '
                      f'f={est.optimizer.f_x:.10f} nsv={len(est.support_)}')
        np.savez_compressed(os.path.join(out, f'fit_svc_n{n}.npz'), **data)

    for n, d in ((150, 5), (400, 8)):
        X, y = make_regression(n, d, seed=200 + n)
        Xte, _ = make_regression(32, d, seed=700 + n)
        data = {'X': X, 'y': y, 'Xtest': Xte, 'epsilon': 0.1}
        for kname, kern in (('poly', PolyKernel(degree=3, gamma='scale', coef0=1.)), ('rbf', gaussian),
                            ('linear', linear)):
            for s, cls in SOLVERS.items():
                if kname == 'linear' and s == 'as' and n > 150:
                    continue
                mi = 5000 if s == 'as' else 1000
                est = SVR(loss=epsilon_insensitive, epsilon=0.1, kernel=kern, C=1., reg_intercept=True, dual=True,
                          optimizer=cls, max_iter=mi)
                est.fit(X, y)
                data.update(flat(f'{kname}_{s}', _fit_record(est, Xte)))
                print(f'  svr n={n} {kname} {s}: iter={est.optimizer.iter} status={est.optimizer.status} '
                      f'f={est.optimizer.f_x:.10f} nsv={len(est.support_)}')
        np.savez_compressed(os.path.join(out, f'fit_svr_n{n}.npz'), **data)


def gen_cfg5(out):
    # BASELINE config 5 is not reachable through SVC.fit (svm/_base.py:771-774 raises); SURVEY 8(c) item 6
    # drives ActiveSet directly on the squared-hinge dual: Q = K*yy' + yy' + I/(2C), q=-1, 0 <= x (ub=+inf).
    n, d, C = 300, 8, 1.0
    X, y = make_blobs(n, d, seed=42, sigma=6.0)
    K = gaussian(X)
    Q = K * np.outer(y, y)
    Q += np.outer(y, y)
    Q += np.diag(np.ones(n) / (2 * C))
    q = -np.ones(n)
    ub = np.full(n, np.inf)
    x0 = np.ones(n)
    data = {'X': X, 'y': y, 'C': C, 'x0': x0}
    data.update(flat('as', run_solver(ActiveSet, Q, q, ub, x0=x0, keep=(1, 2, 10, 50, 100), max_iter=5000)))
    data.update(flat('pg', run_solver(ProjectedGradient, Q, q, ub, x0=x0, keep=(1, 2, 10, 100), max_iter=1000)))
    np.savez_compressed(os.path.join(out, 'cfg5_sqhinge_n300.npz'), **data)
    print(f"  cfg5: AS iter={data['as_iter']} status={data['as_status']} f={data['as_f_x']:.10f}")


def gen_cfg1(out):
    # BASELINE config 1: SVC hinge, linear kernel, Wolfe dual via ProjectedGradient, n=2000 d=20 synthetic blobs (SURVEY
    # 8(d) generator: standardised overlapping blobs, C=1) through the reference's own SVC.fit.  Q (32 MB) is not stored:
    # it is a function of the stored X, y.  x at a few iterations, the whole objective history, the fitted attributes.
    n, d = 2000, 20
    X, y = make_blobs(n, d, seed=0)
    Xte, _ = make_blobs(32, d, seed=901)
    keep = (1, 10, 80, 100, 120, 1000)
    est = SVC(loss=hinge, kernel=linear, C=1., reg_intercept=True, dual=True, optimizer=ProjectedGradient, max_iter=1000)
    est.fit(X, y)
    # SVC.fit takes no user callback: the iterates come from the same solver on the same Q (svm/_base.py:552-559, 628-629)
    K = linear(X)
    Q = K * np.outer(y, y)
    Q += np.outer(y, y)
    run = run_solver(ProjectedGradient, Q, -np.ones(n), np.ones(n), keep=keep, max_iter=1000)
    assert np.array_equal(run['f_hist'], np.asarray(est.train_loss_history, dtype=float)) and np.array_equal(run['x'], est.alphas_)
    data = {'X': X, 'y': y, 'Xtest': Xte, 'C': 1.0, 'pg_f_hist': run['f_hist'], 'pg_x_iters': run['x_iters'],
            'pg_x_at': run['x_at']}
    data.update(flat('pg', _fit_record(est, Xte)))
    np.savez_compressed(os.path.join(out, 'cfg1_linear_pg_n2000_d20.npz'), **data)
    print(f"  cfg1: PG iter={est.optimizer.iter} status={est.optimizer.status} f={est.optimizer.f_x:.10f} nsv={len(est.support_)}")


class ALRecorder:
    """Callback for the (augmented-)Lagrangian dual runs: AL value, primal value and x at chosen iterations."""

    def __init__(self, keep_x_at=()):
        self.keep = set(keep_x_at)
        self.f, self.pf, self.x = [], [], {}

    def __call__(self, opt):
        self.f.append(float(opt.f_x))
        self.pf.append(float(opt.primal_f_x))
        if opt.iter in self.keep:
            self.x[opt.iter] = np.array(opt.x, dtype=float, copy=True)


def run_al(cls, Q, q, a, lb, ub, x0, rho=1., keep=(), **kw):
    rec = ALRecorder(keep)
    obj = AugmentedLagrangianQuadratic(primal=Quadratic(Q, q), A=a, b=None if a is None else np.zeros(1),
                                       lb=lb, ub=ub, rho=rho)
    opt = cls(f=obj, x=x0.copy(), callback=rec, **kw).minimize()
    ks = sorted(rec.x)
    return {'x': np.array(opt.x, dtype=float), 'f_x': float(opt.f_x), 'g_x': np.array(opt.g_x, dtype=float),
            'iter': int(opt.iter), 'epoch': int(opt.epoch), 'status': str(opt.status),
            'dual_x': np.array(obj.dual_x, dtype=float), 'f_hist': np.array(rec.f), 'pf_hist': np.array(rec.pf),
            'x_iters': np.array(ks), 'x_at': np.stack([rec.x[k] for k in ks])}


AL_RULES = (  # name, class, kwargs  (every update rule of optiml/opti/unconstrained/stochastic, every momentum type once)
    ('sgd', StochasticGradientDescent, dict(step_size=0.004)),
    ('sgd_polyak', StochasticGradientDescent, dict(step_size=0.0003, momentum_type='polyak', momentum=0.9)),
    ('sgd_nesterov', StochasticGradientDescent, dict(step_size=0.0005, momentum_type='nesterov', momentum=0.8)),
    ('adam', Adam, dict(step_size=0.002)),
    ('adam_nesterov', Adam, dict(step_size=0.002, momentum_type='nesterov', momentum=0.5)),
    ('amsgrad', AMSGrad, dict(step_size=0.002)),
    ('amsgrad_polyak', AMSGrad, dict(step_size=0.002, momentum_type='polyak', momentum=0.5)),
    ('adamax', AdaMax, dict(step_size=0.002, beta1=0.8, beta2=0.99)),
    ('adagrad', AdaGrad, dict(step_size=1.)),
    ('adadelta', AdaDelta, dict(step_size=1., decay=0.9)),
    ('rmsprop', RMSProp, dict(step_size=0.01)),
    ('rmsprop_nesterov', RMSProp, dict(step_size=0.01, momentum_type='nesterov', momentum=0.5, decay=0.95)),
)


def gen_lagrangian(out):
    """augmented-Lagrangian dual + the stochastic update rules (full batch).
    optiml/opti/constrained/_base.py:224-410, optiml/opti/_base.py:96-169, optiml/opti/unconstrained/stochastic/*.py,
    dispatch optiml/ml/svm/_base.py:638-723 (SVC), :1188-1270 (SVR)."""
    data = {}
    # optiml/opti/constrained/tests/test_lagrangian_quadratic.py:18-22 (the expected value there is cvxopt's x*,
    # unavailable here: the reference's own result is recorded instead)
    Q, q, ub = generate_box_constrained_quadratic(ndim=2)
    x0 = np.random.RandomState(1).uniform(size=2)
    data.update(nd2_Q=Q, nd2_q=q, nd2_ub=ub, nd2_a=np.array([2., 7.]), nd2_x0=x0)
    data.update(flat('nd2_adagrad', run_al(AdaGrad, Q, q, [2, 7], np.zeros(2), ub, x0, keep=(1, 2, 10, 100, 1000),
                                           step_size=1, epochs=15000)))
    print(f"  nd2 adagrad: iter={data['nd2_adagrad_iter']} status={data['nd2_adagrad_status']} x={data['nd2_adagrad_x']}")

    # every rule on an SVC dual WITHOUT the regularised intercept (Q = K*yy', equality y'a = 0, 0 <= a <= C)
    X, y = make_blobs(128, 8, seed=21, sigma=6.0)
    K = gaussian(X)
    Q = K * np.outer(y, y)
    n = len(y)
    q, ub, lb = -np.ones(n), np.ones(n), np.zeros(n)
    x0 = np.random.RandomState(7).uniform(size=n)
    data.update(rules_X=X, rules_y=y, rules_x0=x0)
    keep = (1, 2, 10, 100, 299)
    # (the momentum and Adam-family rules are unstable against the multiplier update of the equality row — in the
    # reference too — so the equality-constrained fixture keeps the rules with a stable trajectory)
    for name, cls, kw in AL_RULES:
        if name not in ('sgd', 'adagrad', 'adadelta', 'rmsprop', 'rmsprop_nesterov'):
            continue
        r = run_al(cls, Q, q, y, lb, ub, x0, rho=1., keep=keep, epochs=300, tol=1e-10, **kw)
        data.update(flat('rules_' + name, r))
        print(f"  rules {name}: iter={r['iter']} status={r['status']} f={r['f_x']:.8f} pf={r['pf_hist'][-1]:.8f}")
    # every rule with the regularised intercept (no equality row), rho != 1, and a general lb
    Qb = Q + np.outer(y, y)
    for name, cls, kw in AL_RULES:
        r = run_al(cls, Qb, q, None, 0.05 * ub, ub, x0, rho=2.5, keep=keep, epochs=300, tol=1e-10, **kw)
        data.update(flat('rulesb_' + name, r))
        print(f"  rulesb {name}: iter={r['iter']} status={r['status']} f={r['f_x']:.8f} pf={r['pf_hist'][-1]:.8f}")
    # schedules (stochastic/schedules.py): an iterable step size and an iterable momentum, one value drawn per iteration
    from optiml.opti.unconstrained.stochastic import schedules as ref_sched
    r = run_al(StochasticGradientDescent, Qb, q, None, 0.05 * ub, ub, x0, rho=2.5, keep=keep, epochs=300, tol=1e-10,
               step_size=ref_sched.decaying(0.002, 0.997), momentum_type='polyak',
               momentum=ref_sched.sutskever_blend(0.9, 40))
    data.update(flat('sched_sgd_polyak', r))
    print(f"  sched sgd polyak: iter={r['iter']} status={r['status']} f={r['f_x']:.8f}")
    r = run_al(RMSProp, Qb, q, None, 0.05 * ub, ub, x0, rho=2.5, keep=keep, epochs=300, tol=1e-10,
               step_size=ref_sched.linear_annealing(0.02, 0.002, 200), momentum_type='nesterov',
               momentum=ref_sched.repeater([0.2, 0.4, 0.6], 100))
    data.update(flat('sched_rmsprop_nesterov', r))
    print(f"  sched rmsprop nesterov: iter={r['iter']} status={r['status']} f={r['f_x']:.8f}")
    # 'optimal' through the tolerance test (optiml/opti/_base.py:141-146)
    r = run_al(AdaGrad, Q, q, y, lb, ub, x0, rho=1., keep=(1, 10), epochs=20000, tol=2e-3, step_size=1.)
    data.update(flat('tol_adagrad', r))
    print(f"  tol adagrad: iter={r['iter']} status={r['status']} f={r['f_x']:.8f}")
    np.savez_compressed(os.path.join(out, 'al_dual.npz'), **data)

    # end to end through SVC.fit / SVR.fit (test_svc.py:134-147 shape: AdaGrad, learning_rate=1.)
    import warnings
    X, y = make_blobs(200, 6, seed=300, sigma=6.0)
    Xte, _ = make_blobs(32, 6, seed=1100, sigma=6.0)
    data = {'X': X, 'y': y, 'Xtest': Xte}
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        for reg in (False, True):
            for name, cls, kw in (('adagrad', AdaGrad, dict(learning_rate=1.)),
                                  ('adam', Adam, dict(learning_rate=0.002, momentum_type='nesterov', momentum=0.5))
                                  if reg else ('rmsprop', RMSProp, dict(learning_rate=0.01))):
                est = SVC(loss=hinge, kernel=gaussian, C=1., reg_intercept=reg, dual=True, optimizer=cls,
                          max_iter=1000, random_state=1, **kw).fit(X, y)
                tag = f"{name}_{'b' if reg else 'nob'}"
                rec = _fit_record(est, Xte)
                rec['dual_x'] = np.asarray(est.obj.dual_x, dtype=float)
                data.update(flat(tag, rec))
                print(f"  svc {tag}: iter={est.optimizer.iter} status={est.optimizer.status} f={est.optimizer.f_x:.8f} "
                      f"nsv={len(est.support_)} b={est.intercept_:.6f}")
    np.savez_compressed(os.path.join(out, 'fit_al_svc_n200.npz'), **data)

    X, y = make_regression(150, 5, seed=350)
    Xte, _ = make_regression(32, 5, seed=850)
    data = {'X': X, 'y': y, 'Xtest': Xte, 'epsilon': 0.1}
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        for reg in (False, True):
            est = SVR(loss=epsilon_insensitive, epsilon=0.1, kernel=gaussian, C=1., reg_intercept=reg, dual=True,
                      optimizer=AdaGrad, learning_rate=1., max_iter=1000, random_state=1).fit(X, y)
            tag = f"adagrad_{'b' if reg else 'nob'}"
            rec = _fit_record(est, Xte)
            rec['dual_x'] = np.asarray(est.obj.dual_x, dtype=float)
            data.update(flat(tag, rec))
            print(f"  svr {tag}: iter={est.optimizer.iter} status={est.optimizer.status} f={est.optimizer.f_x:.8f} "
                  f"nsv={len(est.support_)} b={est.intercept_:.6f}")
    np.savez_compressed(os.path.join(out, 'fit_al_svr_n150.npz'), **data)


def gen_smo(out):
    """SMO (optiml/ml/svm/smo.py) through SVC.fit / SVR.fit(dual=True, optimizer='smo').  The state
    after every outer iteration is captured through the objective the reference evaluates for its verbose line."""
    import io
    from optiml.ml.svm import smo as ref_smo
    KEEP = 40   # outer iterations whose state is stored (the final state is always stored)

    class Spy:
        """records (alphas, b_up, b_low, errors) whenever SMO evaluates its cost line (once per outer iteration)"""

        def __init__(self):
            self.opt = None
            self.alphas, self.b_up, self.b_low = [], [], []

        def function(self, x):
            self.alphas.append(np.array(x, dtype=float, copy=True))
            self.b_up.append(float(self.opt.b_up))
            self.b_low.append(float(self.opt.b_low))
            return 0.

    def run_svc(X, y, kern, C, tol):
        yb = np.where(y == np.unique(y)[-1], 1., -1.)
        K = kern(X)
        spy = Spy()
        opt = ref_smo.SMOClassifier(spy, X, yb, K, kern, C, tol, verbose=1)
        spy.opt = opt
        with contextlib.redirect_stdout(io.StringIO()):
            opt.minimize()
        rec = {'alphas': opt.alphas.copy(), 'b': float(opt.b), 'iter': int(opt.iter), 'b_up': float(opt.b_up),
               'b_low': float(opt.b_low), 'errors': opt.errors.copy(), 'outer_alphas': np.stack(spy.alphas[:KEEP]),
               'outer_b_up': np.array(spy.b_up[:KEEP]), 'outer_b_low': np.array(spy.b_low[:KEEP])}
        if isinstance(kern, LinearKernel):
            rec['w'] = np.asarray(opt.w, dtype=float)
        return rec

    def run_svr(X, y, kern, C, eps, tol):
        K = kern(X)
        spy = Spy()
        opt = ref_smo.SMORegression(spy, X, y, K, kern, C, eps, tol, verbose=1)
        spy.opt = opt
        with contextlib.redirect_stdout(io.StringIO()):
            opt.minimize()
        rec = {'alphas_p': opt.alphas_p.copy(), 'alphas_n': opt.alphas_n.copy(), 'b': float(opt.b),
               'iter': int(opt.iter), 'b_up': float(opt.b_up), 'b_low': float(opt.b_low), 'errors': opt.errors.copy(),
               'outer_alphas': np.stack(spy.alphas[:KEEP]), 'outer_b_up': np.array(spy.b_up[:KEEP]),
               'outer_b_low': np.array(spy.b_low[:KEEP])}
        if isinstance(kern, LinearKernel):
            rec['w'] = np.asarray(opt.w, dtype=float)
        return rec

    data = {}
    for n, d in ((200, 6), (600, 10)):
        X, y = make_blobs(n, d, seed=400 + n, sigma=6.0)
        data.update({f'svc{n}_X': X, f'svc{n}_y': y})
        for kname, kern in (('rbf', gaussian), ('linear', linear)):
            for tol in (1e-3, 1e-4):
                r = run_svc(X, y, kern, 1.0, tol)
                tag = f'svc{n}_{kname}_tol{tol:g}'
                data.update(flat(tag, r))
                print(f"  {tag}: outer={r['iter']} nsv={(r['alphas'] > 1e-6).sum()} b={r['b']:.8f}")
    r = run_svc(data['svc200_X'], data['svc200_y'], gaussian, 10.0, 1e-3)
    data.update(flat('svc200_rbf_C10', r))
    print(f"  svc200_rbf_C10: outer={r['iter']} b={r['b']:.8f}")
    for n, d in ((150, 5), (400, 8)):
        X, y = make_regression(n, d, seed=500 + n)
        data.update({f'svr{n}_X': X, f'svr{n}_y': y})
        for kname, kern in (('rbf', gaussian), ('linear', linear)):
            for tol in (1e-3, 1e-4):
                r = run_svr(X, y, kern, 1.0, 0.1, tol)
                tag = f'svr{n}_{kname}_tol{tol:g}'
                data.update(flat(tag, r))
                print(f"  {tag}: outer={r['iter']} nsv={((r['alphas_p'] > 1e-6) | (r['alphas_n'] > 1e-6)).sum()} "
                      f"b={r['b']:.8f}")
    np.savez_compressed(os.path.join(out, 'smo.npz'), **data)

    # end to end through SVC.fit / SVR.fit (optiml/ml/tests/test_svc.py:71-79, test_svr.py:86-94)
    X, y = make_blobs(300, 6, seed=450, sigma=6.0)
    Xte, _ = make_blobs(32, 6, seed=1250, sigma=6.0)
    data = {'X': X, 'y': y, 'Xtest': Xte}
    for kname, kern in (('rbf', gaussian), ('linear', linear)):
        est = SVC(loss=hinge, kernel=kern, C=1., dual=True, optimizer='smo').fit(X, y)
        rec = {'alphas': est.alphas_, 'support': est.support_, 'dual_coef': est.dual_coef_,
               'intercept': float(est.intercept_), 'iter': int(est.optimizer.iter),
               'decision': est.decision_function(Xte)}
        if kname == 'linear':
            rec['coef'] = np.asarray(est.coef_, dtype=float)
        data.update(flat('svc_' + kname, rec))
        print(f"  fit svc {kname}: outer={rec['iter']} nsv={len(est.support_)} b={rec['intercept']:.8f}")
    Xr, yr = make_regression(250, 5, seed=550)
    Xrt, _ = make_regression(32, 5, seed=1350)
    data.update(Xr=Xr, yr=yr, Xrtest=Xrt)
    for kname, kern in (('rbf', gaussian), ('linear', linear)):
        est = SVR(loss=epsilon_insensitive, epsilon=0.1, kernel=kern, C=1., dual=True, optimizer='smo').fit(Xr, yr)
        rec = {'alphas': est.alphas_, 'support': est.support_, 'dual_coef': est.dual_coef_,
               'intercept': float(est.intercept_), 'iter': int(est.optimizer.iter),
               'decision': est.decision_function(Xrt)}
        if kname == 'linear':
            rec['coef'] = np.asarray(est.coef_, dtype=float)
        data.update(flat('svr_' + kname, rec))
        print(f"  fit svr {kname}: outer={rec['iter']} nsv={len(est.support_)} b={rec['intercept']:.8f}")
    np.savez_compressed(os.path.join(out, 'fit_smo.npz'), **data)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--out', default=os.path.join(REPO, 'tests', 'golden'))
    ap.add_argument('--only', default=None, help='run a single generator, e.g. gen_kernels_more')
    args = ap.parse_args()
    os.makedirs(args.out, exist_ok=True)
    for fn in (gen_unit_problems, gen_x_star, gen_kernels, gen_kernels_more, gen_trajectories, gen_pg_converged, gen_fits, gen_cfg5, gen_cfg1, gen_lagrangian, gen_smo):
        if args.only and fn.__name__ != args.only:
            continue
        print(fn.__name__)
        fn(args.out)
    print('done ->', args.out)


if __name__ == '__main__':
    main()
