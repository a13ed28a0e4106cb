"""SMO on the device-resident Gram panel — SURVEY 8(f).4.

Interface of optiml/ml/svm/smo.py (`SMO` :11-74, `SMOClassifier` :77-357, `SMORegression` :360-797): constructor
arguments, the attributes SVC/SVR read back (`alphas` / `alphas_p` / `alphas_n`, `b`, `w` for the linear kernel,
`errors`, `b_up`, `b_low`, `b_up_idx`, `b_low_idx`, `iter`) and the verbose cost line.  The sweeps themselves run in
libbcqp_hip.so (`bq_smo_*`, csrc/bq_smo.hip) on the Gram panel that `quad` (a `KernelQuadratic`) keeps in HBM; the
`K` argument of the reference is accepted and ignored (the n x n host matrix is never needed).
"""
import ctypes as C
from abc import ABC

import numpy as np

from ... import _lib
from ...opti import KernelQuadratic
from .kernels import LinearKernel

__all__ = ['SMO', 'SMOClassifier', 'SMORegression']


class SMO(ABC):
    _task = None

    def __init__(self, quad, X, y, K, kernel, C, tol=1e-3, verbose=False):
        if not isinstance(quad, KernelQuadratic):
            raise TypeError('the device SMO needs a KernelQuadratic (its Gram panel) as quad')
        self.quad = quad
        self.X = np.ascontiguousarray(X, dtype=float)
        self.y = np.ascontiguousarray(y, dtype=float)
        self.K = K
        self.kernel = kernel
        if isinstance(kernel, LinearKernel):
            self.w = 0.
        self.b = 0.
        self.C = C
        self.errors = np.zeros(len(self.X))
        self.tol = tol
        self.iter = 0
        self.verbose = verbose

    def _epsilon(self):
        return 0.0

    def _coefficients(self):
        """the vector c with decision(x) = sum_j c_j k(x_j, x) + b"""
        raise NotImplementedError

    def _all_alphas(self):
        raise NotImplementedError

    def _pull(self, lib, h):
        raise NotImplementedError

    def minimize(self):
        lib = _lib.load()
        dev = self.quad.device_problem()
        h = C.c_void_p()
        _lib.check(lib.bq_smo_create(dev.handle, self._task, _lib.ptr(self.y), float(self.C), float(self._epsilon()),
                                     float(self.tol), C.byref(h)))
        try:
            if self.verbose:
                print('iter\t cost')
            outer, fin = C.c_int64(0), C.c_int(0)
            while not fin.value:
                # one sweep per call only when the cost line has to be printed in between
                _lib.check(lib.bq_smo_run(h, 1 if self.verbose else 64, C.byref(outer), C.byref(fin)))
                if self.verbose:
                    self.iter = outer.value - 1
                    if not self.iter % self.verbose:
                        self._pull(lib, h)
                        print('{:4d}\t{: 1.4e}'.format(self.iter, self.quad.function(self._all_alphas())))
            self.iter = outer.value
            self._pull(lib, h)
            sc = np.empty(6)
            _lib.check(lib.bq_smo_get(h, _lib.SMO_SCALARS, _lib.ptr(sc)))
            self.b_up, self.b_low, self.b_up_idx, self.b_low_idx = sc[0], sc[1], int(sc[2]), int(sc[3])
            self.steps, self.b = int(sc[4]), sc[5]
            st = np.empty(4)
            _lib.check(lib.bq_smo_get(h, _lib.SMO_STATS, _lib.ptr(st)))
            # self-checks of the helper hand-off (bq_smo.hip): the two rejection counts must be 0
            self.helper_stats = {'helpers': int(st[0]), 'delivered': int(st[1]), 'rejected_list_hash': int(st[2]),
                                 'rejected_checksum': int(st[3])}
            _lib.check(lib.bq_smo_get(h, _lib.SMO_ERRORS, _lib.ptr(self.errors)))
            if isinstance(self.kernel, LinearKernel):
                self.w = self._coefficients() @ self.X      # smo.py:193-196 / :595-598 accumulated over the steps
            if self.verbose:
                print()
        finally:
            lib.bq_smo_destroy(h)
        return self


class SMOClassifier(SMO):
    _task = _lib.SVC

    def __init__(self, quad, X, y, K, kernel, C, tol=1e-3, verbose=False):
        self.alphas = np.zeros(len(X))
        super(SMOClassifier, self).__init__(quad, X, y, K, kernel, C, tol, verbose)

    def _pull(self, lib, h):
        _lib.check(lib.bq_smo_get(h, _lib.SMO_ALPHAS, _lib.ptr(self.alphas)))

    def _all_alphas(self):
        return self.alphas

    def _coefficients(self):
        return self.alphas * self.y


class SMORegression(SMO):
    _task = _lib.SVR

    def __init__(self, quad, X, y, K, kernel, C, epsilon, tol=1e-3, verbose=False):
        self.alphas_p = np.zeros(len(X))
        self.alphas_n = np.zeros(len(X))
        super(SMORegression, self).__init__(quad, X, y, K, kernel, C, tol, verbose)
        self.epsilon = epsilon

    def _epsilon(self):
        return self.epsilon

    def _pull(self, lib, h):
        both = np.empty(2 * len(self.alphas_p))
        _lib.check(lib.bq_smo_get(h, _lib.SMO_ALPHAS, _lib.ptr(both)))
        self.alphas_p, self.alphas_n = both[:len(both) // 2].copy(), both[len(both) // 2:].copy()

    def _all_alphas(self):
        return np.concatenate((self.alphas_p, self.alphas_n))

    def _coefficients(self):
        return self.alphas_p - self.alphas_n
