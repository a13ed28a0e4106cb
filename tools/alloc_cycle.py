#!/usr/bin/env python3
"""Create / release cycles of the headline-size problem: what a second fit in the same process pays for its panel."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from optiml_amd import device  # noqa: E402
from optiml_amd.datasets import make_blobs  # noqa: E402
from optiml_amd.ml.svm.kernels import gaussian  # noqa: E402
from optiml_amd.opti import KernelQuadratic  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
device.get_context()
X, y = make_blobs(n, 128, seed=0)
yb = np.where(y == np.unique(y)[-1], 1., -1.)
for rep in range(4):
    quad = KernelQuadratic(X, -np.ones(n), 'svc', gaussian, y=yb)
    t0 = time.perf_counter()
    quad.device_problem()
    print('cycle %d: problem (panel + Gram build) ready in %.3f s' % (rep, time.perf_counter() - t0), flush=True)
    quad.release()
