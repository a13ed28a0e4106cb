#!/usr/bin/env python3
"""What the panel placement choice leaves behind (profiles/r05/placement_release_transient.txt): config 2's panel (1.6 GB) built
plain, with profiling switched on first, or with the placement choice; then `time_matvec` (50 products by events) three times.

    python tools/placement_ab.py plain|profile_first|tuned
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from optiml_amd import device  # noqa: E402
from optiml_amd.datasets import make_blobs  # noqa: E402
from optiml_amd.ml.svm.kernels import gaussian  # noqa: E402
from optiml_amd.opti import KernelQuadratic  # noqa: E402

mode = sys.argv[1] if len(sys.argv) > 1 else 'plain'
n, d = 20000, 64
ctx = device.get_context()
if mode == 'profile_first':
    ctx.profile(True)
X, y = make_blobs(n, d, seed=0)
quad = KernelQuadratic(X, -np.ones(n), 'svc', gaussian, y=y, tune_placement=(mode == 'tuned'))
dev = quad.device_problem(ctx)
print(mode, 'time_matvec', ' '.join('%.4f' % dev.time_matvec(50) for _ in range(3)), 'placement', dev.placement())
