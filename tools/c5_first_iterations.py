#!/usr/bin/env python3
"""BASELINE config 5 (n = 250 000, d = 256, fp32 panel, ActiveSetCG at inner tolerance 1e-8): inner conjugate-gradient iterations of the
first three outer iterations — one cold solve from x0 = 1 and two warm ones — per second feature family of the preconditioner
(BQ_TEST_HOOKS=as_cg_pc_class=1, =2, default = 2 + the implicit order-2 remainder).  Measured: 40 / 32 / 22 (tests/test_gpu_fullsize.py holds the
default to <= 26)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from optiml_amd.datasets import make_blobs
from optiml_amd.ml.svm.kernels import gaussian
from optiml_amd.opti import KernelQuadratic
from optiml_amd.opti.constrained import ActiveSetCG
n, d = 250000, 256
X, y = make_blobs(n, d, seed=0)
for fam in ('1', '2', None):
    if fam is None: os.environ.pop('BQ_TEST_HOOKS', None)
    else: os.environ['BQ_TEST_HOOKS'] = 'as_cg_pc_class=' + fam
    quad = KernelQuadratic(X, -np.ones(n), 'svc', gaussian, y=y, diag=0.5, storage='f32')
    class S(ActiveSetCG):
        inner_tol = 1e-8
    opt = S(quad=quad, ub=np.full(n, np.inf), x=np.ones(n), max_iter=3).minimize()
    print('family', fam, 'inner iterations over 3 outer:', opt.inner_iters, 'f', opt.f_x, flush=True)
    quad.release()
