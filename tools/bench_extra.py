#!/usr/bin/env python3
"""Secondary measurements (not the driver's bench contract): one JSON line per run.

    python tools/bench_extra.py ip   --n 16000 --d 128 --iters 3     # InteriorPoint iteration / Cholesky rate
    python tools/bench_extra.py chol --n 8192                        # stand-alone dense Cholesky solve
    python tools/bench_extra.py fit  --n 20000 --d 64 --solver pg    # SVC.fit wall time (BASELINE config 2 shape)
    python tools/bench_extra.py smo  --n 20000 --d 64                # device SMO fit (CPU baseline: bench.py --solver smo)
    python tools/bench_extra.py cfg5 --n 250000 --d 256 --storage f32 --max-iter 200   # BASELINE config 5 shape on ONE GPU:
                                      # squared-hinge dual, ub = +inf, x0 = 1, ActiveSet with conjugate-gradient solves
"""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

FP64_MFMA_TF = 78.6


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('what', choices=['ip', 'chol', 'fit', 'smo', 'cfg5'])
    ap.add_argument('--storage', default='f64', choices=['f64', 'f32', 'stream'])
    ap.add_argument('--inner-tol', type=float, default=1e-13, help='cfg5 / ascg: residual level of the inner conjugate gradients')
    ap.add_argument('--n', type=int, default=8192)
    ap.add_argument('--d', type=int, default=128)
    ap.add_argument('--iters', type=int, default=3)
    ap.add_argument('--solver', default='pg')
    ap.add_argument('--task', choices=['svc', 'svr'], default='svc')
    ap.add_argument('--max-iter', type=int, default=1000)
    ap.add_argument('--tol', type=float, default=1e-3)
    ap.add_argument('--progress', action='store_true', help='print one line per solver iteration (long runs)')
    a = ap.parse_args()
    from optiml_amd import _lib, device
    from optiml_amd.datasets import make_blobs
    ctx = device.get_context()
    out = {'what': a.what, 'n': a.n, 'd': a.d, 'device': ctx.name}
    if a.what == 'chol':
        from optiml_amd.linalg import cho_solve_spd
        rs = np.random.RandomState(0)
        A = rs.uniform(-1, 1, (a.n, a.n))
        A = (A + A.T) / 2
        A[np.diag_indices(a.n)] = a.n       # strictly diagonally dominant -> SPD
        b = rs.standard_normal(a.n)
        cho_solve_spd(A[:256, :256].copy(), b[:256])
        x, ms = cho_solve_spd(A, b, return_factor_ms=True)
        res = np.abs(A @ x - b).max()
        out.update(factor_ms=ms, tflops=a.n ** 3 / 3 / (ms * 1e-3) / 1e12, resid=float(res))
        out['frac_of_fp64_mfma_peak'] = out['tflops'] / FP64_MFMA_TF
    elif a.what == 'smo':
        from optiml_amd.ml.svm import SVC, SVR
        from optiml_amd.ml.svm.kernels import gaussian
        from optiml_amd.ml.svm.losses import hinge, epsilon_insensitive
        from optiml_amd.datasets import make_regression
        if a.task == 'svr':
            X, y = make_regression(a.n, a.d, seed=0)
            y = (y - y.mean()) / y.std()
            est = SVR(loss=epsilon_insensitive, epsilon=0.1, kernel=gaussian, C=1., dual=True, optimizer='smo', tol=a.tol)
        else:
            X, y = make_blobs(a.n, a.d, seed=0)
            est = SVC(loss=hinge, kernel=gaussian, C=1., dual=True, optimizer='smo', tol=a.tol)
        t0 = time.perf_counter()
        est.fit(X, y)
        dt = time.perf_counter() - t0
        o = est.optimizer
        out.update(task=a.task, tol=a.tol, fit_s=dt, outer_iters=o.iter, pair_steps=o.steps, n_sv=int(len(est.support_)),
                   b=float(est.intercept_), steps_per_s=o.steps / dt, score=float(est.score(X[:5000], y[:5000])))
    elif a.what == 'ip':
        from optiml_amd.ml.svm.kernels import gaussian
        from optiml_amd.opti import KernelQuadratic
        from optiml_amd.opti.constrained._base import _DeviceSolver
        X, y = make_blobs(a.n, a.d, seed=0)
        quad = KernelQuadratic(X, -np.ones(a.n), 'svc', gaussian, y=y)
        ctx.profile(True)
        dev = quad.device_problem(ctx)
        ub = np.ones(a.n)
        s = _DeviceSolver(dev, _lib.IP, np.zeros(a.n), ub, ub / 2, 1e-10, 10 ** 6)
        s.run(1)
        ctx.profile_read(_lib.PROF_CHOL, reset=True)
        t0 = time.perf_counter()
        rows, status = s.run(a.iters)
        dt = time.perf_counter() - t0
        ms, cnt = ctx.profile_read(_lib.PROF_CHOL, reset=True)
        out.update(iters=len(rows), s_per_iter=dt / max(len(rows), 1), chol_ms=ms / max(cnt, 1),
                   chol_tflops=a.n ** 3 / 3 / (ms / max(cnt, 1) * 1e-3) / 1e12, gap_last=float(rows['r2'][-1]))
        out['chol_frac_of_fp64_mfma_peak'] = out['chol_tflops'] / FP64_MFMA_TF
        s.close()
    elif a.what == 'cfg5':
        # SURVEY 8(c).6 / BASELINE config 5: ActiveSet(Quadratic(K*yy' + yy' + I/(2C), -1), ub = +inf, x = 1); the restricted
        # systems by conjugate gradients on the (fp32-stored) panel — the reference's dense factor would need n^2 more
        from optiml_amd.ml.svm.kernels import gaussian
        from optiml_amd.opti import KernelQuadratic
        from optiml_amd.opti.constrained import ActiveSetCG
        X, y = make_blobs(a.n, a.d, seed=0)
        quad = KernelQuadratic(X, -np.ones(a.n), 'svc', gaussian, y=y, diag=0.5, storage=a.storage)
        t0 = time.perf_counter()
        quad.device_problem()      # returns after the Gram build has finished
        out['gram_s'] = time.perf_counter() - t0
        ub = np.full(a.n, np.inf)
        ActiveSetCG.inner_tol = a.inner_tol
        if a.progress:
            ActiveSetCG.chunk = 1
        t0 = time.perf_counter()
        o = ActiveSetCG(quad=quad, ub=ub, x=np.ones(a.n), max_iter=a.max_iter, verbose=bool(a.progress)).minimize()
        dt = time.perf_counter() - t0
        x = np.asarray(o.x, float)
        g = quad.jacobian(x)
        dd = -g
        dd[(x <= 1e-12) & (dd < 0)] = 0.
        out.update(storage=a.storage, inner_tol=a.inner_tol, solve_s=dt, iters=o.iter, status=o.status, f=o.f_x, inner_iters=int(o.inner_iters),
                   outer_iter_per_s=o.iter / dt, products_per_outer=(o.inner_iters + 2 * o.iter) / max(o.iter, 1),
                   n_bound=int(o.n_bound), proj_grad_norm=float(np.linalg.norm(dd)),
                   free_grad_norm=float(np.linalg.norm(g[~(o.L | o.U)])))
    else:
        from optiml_amd.ml.svm import SVC
        from optiml_amd.ml.svm.kernels import gaussian
        from optiml_amd.ml.svm.losses import hinge
        from optiml_amd.opti.constrained import ProjectedGradient, FrankWolfe, InteriorPoint, ActiveSet
        from optiml_amd.opti.constrained import ActiveSetCG
        cls = {'pg': ProjectedGradient, 'fw': FrankWolfe, 'ip': InteriorPoint, 'as': ActiveSet, 'ascg': ActiveSetCG}[a.solver]
        if a.progress:
            cls.chunk = 1      # one device run per iteration so that verbose lines appear as the solve advances
        if a.solver == 'ascg':
            cls.inner_tol = a.inner_tol
        if a.task == 'svr':
            from optiml_amd.ml.svm import SVR
            from optiml_amd.ml.svm.losses import epsilon_insensitive
            from optiml_amd.datasets import make_regression
            X, y = make_regression(a.n, a.d, seed=0)
            y = (y - y.mean()) / y.std()
            est = SVR(loss=epsilon_insensitive, epsilon=0.1, kernel=gaussian, C=1., reg_intercept=True, dual=True,
                      optimizer=cls, max_iter=a.max_iter, verbose=bool(a.progress))
        else:
            X, y = make_blobs(a.n, a.d, seed=0)
            est = SVC(loss=hinge, kernel=gaussian, C=1., reg_intercept=True, dual=True, optimizer=cls,
                      max_iter=a.max_iter, verbose=bool(a.progress))
        t0 = time.perf_counter()
        est.fit(X, y)
        dt = time.perf_counter() - t0
        o = est.optimizer
        # the same optimality measure for every solver — the 2-norm of d = -g with ProjectedGradient's
        # masks (projected_gradient.py:100-107), g = Qx + q taken from one more device product at the final point
        x = np.asarray(o.x, float)
        g = est.obj.jacobian(x)
        dd = -g
        dd[(np.asarray(o.ub) - x <= 1e-12) & (dd > 0)] = 0.
        dd[(x - np.asarray(o.lb) <= 1e-12) & (dd < 0)] = 0.
        out.update(proj_grad_norm=float(np.linalg.norm(dd)), f_recomputed=float(est.obj.function(x)))
        if hasattr(o, 'inner_iters'):
            out['inner_iters'] = int(o.inner_iters)
        out.update(task=a.task, solver=a.solver, fit_s=dt, iters=o.iter, status=o.status, f=o.f_x,
                   n_sv=int(len(est.support_)), iter_per_s=o.iter / dt, score=float(est.score(X[:5000], y[:5000])))
    print(json.dumps(out), flush=True)


if __name__ == '__main__':
    main()
