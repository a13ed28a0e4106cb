#!/usr/bin/env python3
"""What does forming K_ij + 1 in registers cost the panel product?  One panel, the same tile kernel with (bq_problem_matvec: the
dual's Hessian) and without (bq_problem_gram_matvec: the raw Gram matrix) the +1, HIP-event time of symv_tiles_kernel each.

    python tools/addone_probe.py [n] [d] [storage]
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from optiml_amd import _lib, device  # noqa: E402
from optiml_amd.datasets import make_blobs  # noqa: E402
from optiml_amd.ml.svm.kernels import gaussian  # noqa: E402
from optiml_amd.opti import KernelQuadratic  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 250000
d = int(sys.argv[2]) if len(sys.argv) > 2 else 256
storage = sys.argv[3] if len(sys.argv) > 3 else 'f32'
X, y = make_blobs(n, d, seed=0)
ctx = device.get_context()
quad = KernelQuadratic(X, -np.ones(n), 'svc', gaussian, y=y, storage=storage)
dev = quad.device_problem(ctx)
v = np.random.RandomState(0).standard_normal(n)
ctx.profile(True)
for rnd in range(3):
    for name, fn in (('with +1 (matvec)', dev.matvec), ('without (gram_matvec)', dev.gram_matvec)):
        fn(v)
        ctx.profile_read(_lib.PROF_MATVEC, reset=True)
        for _ in range(5):
            fn(v)
        ms, cnt = ctx.profile_read(_lib.PROF_MATVEC, reset=True)
        print(f'n={n} {storage} round {rnd} {name:24s}: {ms / cnt:.3f} ms per launch ({cnt} launches)', flush=True)
