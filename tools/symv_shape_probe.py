#!/usr/bin/env python3
"""Launch time of the symmetric panel product (`bq_problem_time_matvec`) by shape and storage type: is the fp32 panel of config 5
slower per byte than the fp64 headline panel because of its TYPE, its FOOTPRINT or its row pitch?

    python tools/symv_shape_probe.py n:d:storage[,n:d:storage...] [reps]
"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from optiml_amd import device
from optiml_amd.datasets import make_blobs
from optiml_amd.ml.svm.kernels import gaussian
from optiml_amd.opti import KernelQuadratic

shapes = [s.split(':') for s in (sys.argv[1] if len(sys.argv) > 1 else '100000:128:f64,100000:128:f32').split(',')]
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
ctx = device.get_context()
for n, d, st in shapes:
    n, d = int(n), int(d)
    X, y = make_blobs(n, d, seed=0)
    if st == 'f32':
        X = X.astype(np.float32).astype(np.float64)
    q = KernelQuadratic(X, -np.ones(n), 'svc', gaussian, y=y, storage=st)
    dv = q.device_problem(ctx)
    nb = (n + 255) // 256
    tiles = nb * (nb + 1) // 2
    s = 4 if st == 'f32' else 8
    gb = (tiles * (256 * 256 * s + 2048) + (nb * (nb // 8 + 1)) * 2048) / 1e9   # tiles + column parts + row parts
    for _ in range(int(os.environ.get('PROBE_SWEEPS', '1'))):
        dv.time_matvec(3)
        ms = [dv.time_matvec(reps) for _ in range(3)]
        print(f'n={n} d={d} {st}: panel {tiles * 65536 * s / 2**30:.1f} GiB, pitch of the last tile row {nb * 256 * s / 1024:.0f} KiB, '
              f'product {min(ms):.4f} .. {max(ms):.4f} ms = {gb / min(ms):.3f} TB/s ({gb / min(ms) / 8:.3f} of 8 TB/s)', flush=True)
    q.release()
