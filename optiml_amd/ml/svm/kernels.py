"""Gram-matrix functors (linear, polynomial, Gaussian, Laplacian, sigmoid) evaluated on the device.

Same classes, ctor arguments and validation as optiml/ml/svm/kernels.py:40-208.  `kernel(X, Y=None)` returns
the dense Gram matrix computed by the fp64-MFMA tile kernel; gamma='scale' is 1 / (n_features * X.var()) of
the FIRST argument, 'auto' is 1 / n_features (kernels.py:93-94, :127-128).  The SVM estimators never call
these on the training set: they hand the kernel *spec* to the device (`device_spec`) and keep K in HBM.
"""
import ctypes as C
from abc import ABC

import numpy as np

from ... import _lib
from ...device import get_context

try:  # keeps get_params/set_params/clone working when scikit-learn is present
    from sklearn.base import BaseEstimator
except ImportError:  # pragma: no cover
    class BaseEstimator:
        def get_params(self, deep=True):
            import inspect
            names = [p for p in inspect.signature(type(self).__init__).parameters if p != 'self']
            return {k: getattr(self, k) for k in names}

        def set_params(self, **params):
            for k, v in params.items():
                setattr(self, k, v)
            return self

__all__ = ['Kernel', 'LinearKernel', 'PolyKernel', 'GaussianKernel', 'LaplacianKernel', 'SigmoidKernel',
           'linear', 'poly', 'gaussian', 'laplacian', 'sigmoid']


def _resolve_gamma(gamma, X):
    if isinstance(gamma, str):
        return 1. / (X.shape[1] * X.var()) if gamma == 'scale' else 1. / X.shape[1]
    return float(gamma)


def _check_pair(X, Y):
    X = np.ascontiguousarray(X, dtype=float)
    if X.ndim != 2:
        raise ValueError('X must be 2-dimensional')
    if Y is not None:
        Y = np.ascontiguousarray(Y, dtype=float)
        if Y.ndim != 2 or Y.shape[1] != X.shape[1]:
            raise ValueError('Incompatible dimension for X and Y matrices')
    return X, Y


class Kernel(BaseEstimator, ABC):
    _kind = None

    def device_spec(self, X):
        """(kind, gamma, coef0, degree) with gamma resolved against X."""
        raise NotImplementedError

    def __call__(self, X, Y=None):
        X, Y = _check_pair(X, Y)
        kind, gamma, coef0, degree = self.device_spec(X)
        lib = _lib.load()
        m, d = X.shape
        t = m if Y is None else Y.shape[0]
        out = np.empty((m, t))
        _lib.check(lib.bq_gram_matrix(get_context().handle, kind, gamma, coef0, degree, m, d, _lib.ptr(X), t,
                                      _lib.ptr(Y), _lib.ptr(out)))
        return out


class LinearKernel(Kernel):
    def device_spec(self, X):
        return _lib.KERNEL_LINEAR, 0.0, 0.0, 1


class PolyKernel(Kernel):
    def __init__(self, degree=3, gamma='scale', coef0=0.):
        if not degree > 0:
            raise ValueError('degree must be > 0')
        self.degree = degree
        if isinstance(gamma, str):
            if gamma not in ('scale', 'auto'):
                raise ValueError(f'unknown gamma type {gamma}')
        elif not gamma > 0:
            raise ValueError('gamma must be > 0')
        self.gamma = gamma
        self.coef0 = coef0

    def device_spec(self, X):
        return _lib.KERNEL_POLY, _resolve_gamma(self.gamma, X), float(self.coef0), int(self.degree)


class GaussianKernel(Kernel):
    def __init__(self, gamma='scale'):
        if isinstance(gamma, str):
            if gamma not in ('scale', 'auto'):
                raise ValueError(f'unknown gamma type {gamma}')
        elif not gamma > 0:
            raise ValueError('gamma must be > 0')
        self.gamma = gamma

    def device_spec(self, X):
        return _lib.KERNEL_RBF, _resolve_gamma(self.gamma, X), 0.0, 1


class LaplacianKernel(Kernel):
    """exp(-gamma |x - y|_1) — optiml/ml/svm/kernels.py:132-163 (L1 distance: VALU tile kernel, no GEMM form)."""

    def __init__(self, gamma='scale'):
        if isinstance(gamma, str):
            if gamma not in ('scale', 'auto'):
                raise ValueError(f'unknown gamma type {gamma}')
        elif not gamma > 0:
            raise ValueError('gamma must be > 0')
        self.gamma = gamma

    def device_spec(self, X):
        return _lib.KERNEL_LAPLACIAN, _resolve_gamma(self.gamma, X), 0.0, 1


class SigmoidKernel(Kernel):
    """tanh(gamma <x, y> + coef0) — optiml/ml/svm/kernels.py:166-201 (MFMA Gram tile + tanh epilogue)."""

    def __init__(self, gamma='scale', coef0=0.):
        if isinstance(gamma, str):
            if gamma not in ('scale', 'auto'):
                raise ValueError(f'unknown gamma type {gamma}')
        elif not gamma > 0:
            raise ValueError('gamma must be > 0')
        self.gamma = gamma
        self.coef0 = coef0

    def device_spec(self, X):
        return _lib.KERNEL_SIGMOID, _resolve_gamma(self.gamma, X), float(self.coef0), 1


linear = LinearKernel()
poly = PolyKernel()
gaussian = GaussianKernel()
laplacian = LaplacianKernel()
sigmoid = SigmoidKernel()
