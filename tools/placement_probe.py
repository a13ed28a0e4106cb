#!/usr/bin/env python3
"""Is the panel product's launch time a property of WHERE the panel landed?  K panels of the same n=100 000 problem held at the
same time in one process (different physical memory each), each timed with the same kernel; then released and rebuilt.

    python tools/placement_probe.py [K] [reps]
"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from optiml_amd import device
from optiml_amd.datasets import make_blobs
from optiml_amd.ml.svm.kernels import gaussian
from optiml_amd.opti import KernelQuadratic

K = int(sys.argv[1]) if len(sys.argv) > 1 else 5
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
n, d = 100000, 128
X, y = make_blobs(n, d, seed=0)
ctx = device.get_context()
for rnd in range(2):
    quads = [KernelQuadratic(X, -np.ones(n), 'svc', gaussian, y=y) for _ in range(K)]
    devs = [q.device_problem(ctx) for q in quads]
    for sweep in range(2):
        ms = [dv.time_matvec(reps) for dv in devs]
        print(f'round {rnd} sweep {sweep}: ' + ' '.join(f'{m:.3f}' for m in ms) + f'  (min {min(ms):.3f} max {max(ms):.3f} ms)', flush=True)
    for q in quads:
        q.release()
