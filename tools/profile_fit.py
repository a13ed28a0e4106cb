#!/usr/bin/env python3
"""Host-side profile of one SVC.fit (where the wall time outside the kernels goes):  python tools/profile_fit.py [n] [d]"""
import cProfile
import os
import pstats
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from optiml_amd import device  # noqa: E402
from optiml_amd.datasets import make_blobs  # noqa: E402
from optiml_amd.ml.svm import SVC  # noqa: E402
from optiml_amd.ml.svm.kernels import gaussian  # noqa: E402
from optiml_amd.ml.svm.losses import hinge  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
d = int(sys.argv[2]) if len(sys.argv) > 2 else 128
device.get_context()
X, y = make_blobs(n, d, seed=0)
est = SVC(loss=hinge, kernel=gaussian, C=1., dual=True, optimizer='smo', tol=1e-3)
pr = cProfile.Profile()
pr.enable()
est.fit(X, y)
pr.disable()
pstats.Stats(pr).sort_stats('cumulative').print_stats(14)
