#!/usr/bin/env python3
"""Dense ActiveSet's product-free objective (hook as_f_chain, INTEGRATION.md "Deviations") on the panel kinds the parity tests do not
sweep for it: an fp32-stored panel and an SVR dual (2n variables on an n x n panel), beside the fp64 SVC dual.  Each fit once with
the identity and once with a product per iteration (fresh process each); prints iterations, status, the number of iterations without
a product and the largest difference of the recorded objective relative to its scale.      python tools/as_variants_check.py
(gpurun r6h: svc f64 / svc f32 / svr f64: 5 616 / 5 784 / 20 000 iterations, 3 478 / 3 559 / 8 441 without a product, 9e-16 / 9e-16 / 2e-15.)"""
import os, sys, json, subprocess
CHILD = r'''
import sys, json, warnings
import numpy as np
sys.path.insert(0, %(root)r)
from optiml_amd.ml.svm import SVC, SVR
from optiml_amd.ml.svm.kernels import gaussian
from optiml_amd.ml.svm.losses import hinge, epsilon_insensitive
from optiml_amd.opti.constrained import ActiveSet
rng = np.random.default_rng(3)
n, d = 1500, 8
X = rng.standard_normal((n, d)); w = rng.standard_normal(d)
out = {}
for name in ("svc_f64", "svc_f32", "svr_f64"):
    if name.startswith("svc"):
        y = np.where(X @ w + 0.3 * rng.standard_normal(n) > 0, 1.0, -1.0)
        m = SVC(loss=hinge, kernel=gaussian, C=1.0, dual=True, reg_intercept=True, optimizer=ActiveSet, max_iter=20000, storage=name[-3:])
    else:
        y = X @ w + 0.1 * rng.standard_normal(n)
        m = SVR(loss=epsilon_insensitive, epsilon=0.1, kernel=gaussian, C=1.0, dual=True, reg_intercept=True, optimizer=ActiveSet, max_iter=20000)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        m.fit(X, y)
    o = m.optimizer
    out[name] = dict(iter=int(o.iter), status=o.status, f=float(o.f_x), pf=int(o.product_free_iterations), hist=[float(v) for v in m.train_loss_history])
print(json.dumps(out))
'''
res = {}
for chain in ("1", "0"):
    r = subprocess.run([sys.executable, "-c", CHILD % {"root": os.path.dirname(os.path.dirname(os.path.abspath(__file__)))}], env=dict(os.environ, BQ_TEST_HOOKS='as_f_chain=' + chain), capture_output=True, text=True)
    if r.returncode: print(r.stderr[-1500:]); sys.exit(1)
    res[chain] = json.loads(r.stdout.strip().splitlines()[-1])
import numpy as np
for k in res["1"]:
    a, b = res["1"][k], res["0"][k]
    ha, hb = np.array(a["hist"]), np.array(b["hist"])
    m = min(len(ha), len(hb))
    print(k, "iter", a["iter"], b["iter"], a["status"], b["status"], "product-free", a["pf"], b["pf"], "f", a["f"], b["f"],
          "max |df| / max|f|", float(np.abs(ha[:m] - hb[:m]).max() / np.abs(hb).max()))
