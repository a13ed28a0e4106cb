import os, sys, time
import numpy as np
sys.path.insert(0, '/root/repo' if os.path.isdir('/root/repo/optiml_amd') else os.getcwd())
from optiml_amd import _lib, device
from optiml_amd.datasets import make_blobs
from optiml_amd.ml.svm.kernels import gaussian
from optiml_amd.opti import KernelQuadratic
mode = sys.argv[1]
n, d = 20000, 64
ctx = device.get_context()
if mode == 'profile_first':
    ctx.profile(True)
X, y = make_blobs(n, d, seed=0)
quad = KernelQuadratic(X, -np.ones(n), 'svc', gaussian, y=y, tune_placement=(mode == 'tuned'))
dev = quad.device_problem(ctx)
print(mode, 'time_matvec', ' '.join('%.4f' % dev.time_matvec(50) for _ in range(3)), 'placement', dev.placement())
