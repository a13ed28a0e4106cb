#!/usr/bin/env python3
"""How many OUTER iterations does BASELINE config 5 (squared-hinge dual, ActiveSet, ub = +inf, x0 = 1) need to reach 'optimal'?

The reference's ActiveSet binds or releases about one index per outer iteration (optiml/opti/constrained/active_set.py:195-220), so the
count grows linearly with n; at n = 250 000 it cannot be run to the end inside a bench (hours).  This tool runs the SAME workload
(d = 256, fp32 panel, ActiveSetCG, inner tolerance 1e-8) to 'optimal' at sizes that finish in seconds to minutes and prints
iterations / n, the support-vector share and the bound share — the ratios bench.py's `time_to_kkt.c5_projected` multiplies the
measured seconds per outer iteration at n = 250 000 by (profiles/r05/c5_outer_iterations_scaling.json).

    python tools/c5_scaling.py 5000,10000,20000,40000 > profiles/r05/c5_outer_iterations_scaling.json
"""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from optiml_amd import _lib, device  # noqa: E402
from optiml_amd.datasets import make_blobs  # noqa: E402
from optiml_amd.ml.svm.kernels import gaussian  # noqa: E402
from optiml_amd.opti import KernelQuadratic  # noqa: E402
from optiml_amd.opti.constrained._base import _DeviceSolver  # noqa: E402


def main():
    sizes = [int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else '5000,10000,20000').split(',')]
    d = 256
    ctx = device.get_context()
    out = {'what': 'c5_outer_iterations_scaling', 'd': d, 'storage': 'f32', 'inner_tol': 1e-8, 'device': ctx.name, 'runs': []}
    for n in sizes:
        X, y = make_blobs(n, d, seed=0, sigma=8.0)
        quad = KernelQuadratic(X, -np.ones(n), 'svc', gaussian, y=y, storage='f32', diag=0.5)
        dev = quad.device_problem(ctx)
        solver = _DeviceSolver(dev, _lib.AS_CG, np.zeros(n), np.full(n, np.inf), np.ones(n), 1e-6, 10 ** 9)
        solver.set_inner(1e-8, 0)
        t0 = time.perf_counter()
        status, done = 'unknown', 0
        while status == 'unknown':
            rows, status = solver.run(500)
            done += len(rows)
            print(f'[c5-scaling] n={n}: {done} records, {time.perf_counter() - t0:.1f} s', file=sys.stderr, flush=True)
        dt = time.perf_counter() - t0
        it, status, f = solver.state()
        x = solver.get(_lib.GET_X_NOW)
        rec = {'n': n, 'iterations': int(it), 'status': status, 'f': float(f), 'wall_s': dt, 'iterations_per_n': it / n,
               'n_sv': int((x > 1e-6).sum()), 'sv_share': float((x > 1e-6).mean()), 'bound_share': float((x <= 1e-12).mean()),
               'inner_products': int(solver.inner_iters()), 'inner_products_per_outer_iteration': solver.inner_iters() / max(it, 1)}
        out['runs'].append(rec)
        solver.close()
        quad.release()
    print(json.dumps(out, indent=1))


if __name__ == '__main__':
    main()
