#!/usr/bin/env python3
"""Dense ActiveSet: the f(x) of ratio-step iterations from the line-search identity (bq_as.hip, as_step_min_kernel) against the same
fit with a panel product per iteration (BQ_TEST_HOOKS=as_f_chain=0).  Two child processes (the switch is read when the solver starts), same
seeded problem; prints iterations, status, how many iterations went without a product and the largest relative difference of the
recorded f along the trajectory.

    python tools/as_f_chain_check.py [n] [d] [max_iter]
"""
import json
import os
import subprocess
import sys

import numpy as np

CHILD = r'''
import json, sys, time
import numpy as np
sys.path.insert(0, %(root)r)
from optiml_amd.ml.svm import SVC
from optiml_amd.ml.svm.kernels import gaussian
from optiml_amd.opti.constrained import ActiveSet
n, d, max_iter = %(n)d, %(d)d, %(max_iter)d
rng = np.random.default_rng(7)
X = rng.standard_normal((n, d))
w = rng.standard_normal(d)
y = np.where(X @ w + 0.3 * rng.standard_normal(n) > 0, 1.0, -1.0)
from optiml_amd.ml.svm.losses import hinge
m = SVC(loss=hinge, kernel=gaussian, C=1.0, dual=True, reg_intercept=True, optimizer=ActiveSet, max_iter=max_iter)
t0 = time.perf_counter()
import warnings
with warnings.catch_warnings():
    warnings.simplefilter('ignore')
    m.fit(X, y)
dt = time.perf_counter() - t0
opt = m.optimizer
np.save(%(out)r, np.asarray(m.train_loss_history, dtype=float))
print(json.dumps({'fit_s': dt, 'iter': int(opt.iter), 'status': opt.status, 'f': float(opt.f_x),
                  'product_free': int(getattr(opt, 'product_free_iterations', -1))}))
'''


def run(n, d, max_iter, chain, out):
    env = dict(os.environ, BQ_TEST_HOOKS='as_f_chain=%d' % (1 if chain else 0))
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = CHILD % {'root': root, 'n': n, 'd': d, 'max_iter': max_iter, 'out': out}
    r = subprocess.run([sys.executable, '-c', code], env=env, capture_output=True, text=True)
    if r.returncode != 0:
        sys.stderr.write(r.stderr[-2000:])
        raise SystemExit(r.returncode)
    return json.loads(r.stdout.strip().splitlines()[-1])


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 4000
    d = int(sys.argv[2]) if len(sys.argv) > 2 else 32
    max_iter = int(sys.argv[3]) if len(sys.argv) > 3 else 100000
    a = run(n, d, max_iter, True, '/tmp/as_chain_on.npy')
    b = run(n, d, max_iter, False, '/tmp/as_chain_off.npy')
    fa, fb = np.load('/tmp/as_chain_on.npy'), np.load('/tmp/as_chain_off.npy')
    m = min(len(fa), len(fb))
    rel = np.abs(fa[:m] - fb[:m]) / np.maximum(np.abs(fb[:m]), 1e-300)
    print(json.dumps({'n': n, 'd': d, 'identity': a, 'product': b, 'records': [len(fa), len(fb)],
                      'max_rel_f_difference': float(rel.max()) if m else None,
                      'argmax': int(rel.argmax()) if m else None}))


if __name__ == '__main__':
    main()
