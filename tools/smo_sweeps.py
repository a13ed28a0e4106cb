#!/usr/bin/env python3
"""Per-sweep timing of the device SMO (one outer iteration per call):  python tools/smo_sweeps.py [n] [d]"""
import ctypes as C
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from optiml_amd import _lib, device  # noqa: E402
from optiml_amd.datasets import make_blobs  # noqa: E402
from optiml_amd.ml.svm.kernels import gaussian  # noqa: E402
from optiml_amd.opti import KernelQuadratic  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
d = int(sys.argv[2]) if len(sys.argv) > 2 else 128
device.get_context()
X, y = make_blobs(n, d, seed=0)
yb = np.where(y == np.unique(y)[-1], 1., -1.)
quad = KernelQuadratic(X, -np.ones(n), 'svc', gaussian, y=yb, rank_one=False)
lib = _lib.load()
dev = quad.device_problem()
h = C.c_void_p()
_lib.check(lib.bq_smo_create(dev.handle, _lib.SVC, _lib.ptr(yb), 1.0, 0.0, 1e-3, C.byref(h)))
outer, fin = C.c_int64(0), C.c_int(0)
sc = np.empty(6)
alphas = np.empty(n)
prev = 0
while not fin.value:
    t0 = time.perf_counter()
    _lib.check(lib.bq_smo_run(h, 1, C.byref(outer), C.byref(fin)))
    dt = time.perf_counter() - t0
    _lib.check(lib.bq_smo_get(h, _lib.SMO_SCALARS, _lib.ptr(sc)))
    _lib.check(lib.bq_smo_get(h, _lib.SMO_ALPHAS, _lib.ptr(alphas)))
    print('outer %3d  %8.2f ms  steps %6d  (+%d)  nnz %d  free %d' % (
        outer.value, dt * 1e3, int(sc[4]), int(sc[4]) - prev, int((alphas != 0).sum()),
        int(((alphas > 0) & (alphas < 1.0)).sum())), flush=True)
    prev = int(sc[4])
lib.bq_smo_destroy(h)
