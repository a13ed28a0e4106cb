#!/usr/bin/env python3
"""Is a slow placement a matter of where the panel's BASE sits in the address-to-channel map?  K panels of the n = 100 000 headline
problem held at once; each allocation carries slack (BQ_PLACE_OFFSETS) and the product kernel is timed on it at every offset of
the list.  One line per panel: the launch time at each offset.

    BQ_PLACE_OFFSETS=0,4096,... python tools/placement_offsets.py [K]
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from optiml_amd import device  # noqa: E402
from optiml_amd.datasets import make_blobs  # noqa: E402
from optiml_amd.ml.svm.kernels import gaussian  # noqa: E402
from optiml_amd.opti import KernelQuadratic  # noqa: E402

K = int(sys.argv[1]) if len(sys.argv) > 1 else 4
n, d = 100000, 128
X, y = make_blobs(n, d, seed=0)
ctx = device.get_context()
offs = [int(v) & ~4095 for v in os.environ['BQ_PLACE_OFFSETS'].split(',')]
print('offsets      ' + ' '.join('%8s' % (f'{o >> 20}M' if o >= 1 << 20 else f'{o >> 10}K') for o in offs), flush=True)
for rnd in range(int(sys.argv[2]) if len(sys.argv) > 2 else 2):   # release everything and allocate again: other placements
    quads = [KernelQuadratic(X, -np.ones(n), 'svc', gaussian, y=y, tune_placement=True) for _ in range(K)]
    for k, q in enumerate(quads):
        dev = q.device_problem(ctx)
        print(f'round {rnd} panel {k}      ' + ' '.join('%8.3f' % v for v in dev.placement()) + f'   -> kept: {dev.time_matvec(10):.3f} ms', flush=True)
    for q in quads:
        q.release()
