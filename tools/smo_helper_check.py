#!/usr/bin/env python3
"""Diagnostic: SMO runs for several helper-workgroup counts must give the same path (BQ_TEST_HOOKS=smo_helpers=N)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from optiml_amd.datasets import make_blobs, make_regression  # noqa: E402
from optiml_amd.ml.svm.kernels import gaussian  # noqa: E402
from optiml_amd.ml.svm.smo import SMOClassifier, SMORegression  # noqa: E402
from optiml_amd.opti import KernelQuadratic  # noqa: E402

task = sys.argv[1] if len(sys.argv) > 1 else 'svr'
n = int(sys.argv[2]) if len(sys.argv) > 2 else 3000
Xc, yc = make_blobs(6000, 16, seed=3)
ycb = np.where(yc == np.unique(yc)[-1], 1., -1.)
if task == 'svc':
    X, y = make_blobs(n, 16, seed=3)
    yb = np.where(y == np.unique(y)[-1], 1., -1.)
else:
    X, yr = make_regression(n, 8, seed=4)
    yr = (yr - yr.mean()) / yr.std()
for h in (sys.argv[3].split(',') if len(sys.argv) > 3 else ('0', '0', '16', '16', '128', '128', '128', '255', '255')):
    os.environ['BQ_TEST_HOOKS'] = 'smo_helpers=' + h
    if task == 'both':   # the sequence of tests/test_gpu_smo.py: a classifier run first, then the regression
        quad = KernelQuadratic(Xc, -np.ones(6000), 'svc', gaussian, y=ycb, rank_one=False)
        o = SMOClassifier(quad, Xc, ycb, None, gaussian, 1., 1e-3).minimize()
        print('   svc: outer %d steps %d' % (o.iter, o.steps), flush=True)
    if task == 'svc':
        quad = KernelQuadratic(X, -np.ones(n), 'svc', gaussian, y=yb, rank_one=False)
        o = SMOClassifier(quad, X, yb, None, gaussian, 1., 1e-3).minimize()
        sig = float(np.abs(o.alphas).sum())
    else:
        quad = KernelQuadratic(X, np.hstack((-yr, yr)) + 0.1, 'svr', gaussian, rank_one=False)
        o = SMORegression(quad, X, yr, None, gaussian, 1., 0.1, 1e-3).minimize()
        sig = float(np.abs(o.alphas_p).sum() + np.abs(o.alphas_n).sum())
    print('helpers %4s  outer %4d  steps %7d  |alpha|_1 %.15g  b %.15g' % (h, o.iter, o.steps, sig, o.b), flush=True)
