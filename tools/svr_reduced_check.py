#!/usr/bin/env python3
"""Diagnostic (run from the repo root on a GPU box): SVR + InteriorPoint on the two SVR fixtures, default reduced n x n system;
prints the distance of the multipliers from the reference's (the test of the same name in tests/test_gpu_parity.py asserts it)."""
import sys, os, json, numpy as np
sys.path.insert(0, os.getcwd())
from optiml_amd.ml.svm import SVR
from optiml_amd.ml.svm.kernels import gaussian, PolyKernel
from optiml_amd.ml.svm.losses import epsilon_insensitive
from optiml_amd.opti.constrained import InteriorPoint
for name in ('fit_svr_n400.npz', 'fit_svr_n150.npz'):
    g = np.load('tests/golden/' + name)
    est = SVR(loss=epsilon_insensitive, epsilon=0.1, kernel=gaussian, C=1., reg_intercept=True, dual=True,
              optimizer=InteriorPoint).fit(g['X'], g['y'])
    keys = [k for k in g.files if 'ip' in k]
    a = est.alphas_; n = len(a) // 2
    ref = g['rbf_ip_alphas'] if 'rbf_ip_alphas' in g.files else None
    out = dict(name=name, mode=os.environ.get('BQ_TEST_HOOKS', 'ip_svr_reduced=1'), iter=int(est.optimizer.iter), status=est.optimizer.status,
               f=float(est.optimizer.f_x), ref_iter=int(g['rbf_ip_iter']) if 'rbf_ip_iter' in g.files else None,
               ref_f=float(g['rbf_ip_f_x']) if 'rbf_ip_f_x' in g.files else None)
    if ref is not None:
        out['max_dalpha'] = float(np.abs((a[:n] - a[n:]) - (ref[:n] - ref[n:])).max())
    print(json.dumps(out))
