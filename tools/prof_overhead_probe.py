#!/usr/bin/env python3
"""What the per-launch timing of the product (ctx.profile(True): hipExtLaunchKernelGGL with a start and a stop event) costs a short
iteration: BASELINE config 2's PG loop, 400 iterations, with and without it, alternating in one process on one panel.

    python tools/prof_overhead_probe.py [n] [d] [steps]
"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from optiml_amd import _lib, device
from optiml_amd.datasets import make_blobs
from optiml_amd.ml.svm.kernels import gaussian
from optiml_amd.opti import KernelQuadratic
from optiml_amd.opti.constrained._base import _DeviceSolver

n = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
d = int(sys.argv[2]) if len(sys.argv) > 2 else 64
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 400
ctx = device.get_context()
X, y = make_blobs(n, d, seed=0)
quad = KernelQuadratic(X, -np.ones(n), 'svc', gaussian, y=y)
dev = quad.device_problem(ctx)
ub = np.ones(n)
for rep in range(3):
    for prof in (True, False):
        ctx.profile(prof)
        solver = _DeviceSolver(dev, _lib.PG, np.zeros(n), ub, ub / 2, 1e-6, 10 ** 9)
        solver.run(20)
        t0 = time.perf_counter()
        rows, status = solver.run(steps)
        t1 = time.perf_counter()
        note = ''
        if prof:
            ms, cnt = ctx.profile_read(_lib.PROF_MATVEC, reset=True)
            note = f'  product {ms / max(cnt, 1):.4f} ms over {cnt} launches'
        print(f'n={n} profiling={"on " if prof else "off"}: {1e3 * (t1 - t0) / len(rows):.4f} ms per iteration  ({len(rows) / (t1 - t0):.1f} iter/s){note}', flush=True)
        del solver
