"""Dense SPD solve on the device: the cho_factor/cho_solve pair of the interior-point and active-set solvers."""
import ctypes as C

import numpy as np

from . import _lib
from .device import get_context

__all__ = ['cho_solve_spd']


def cho_solve_spd(A, b, return_factor_ms=False):
    """x = A^-1 b for a symmetric positive definite A (lower triangle read); raises LinAlgError if A is not PD."""
    A = np.ascontiguousarray(A, dtype=float)
    n = A.shape[0]
    if A.ndim != 2 or A.shape[1] != n:
        raise ValueError('expected square matrix')
    b = _lib.as_f64(b, n, 'b')
    x = np.empty(n)
    ms = C.c_double(0)
    lib = _lib.load()
    rc = lib.bq_cholesky_solve(get_context().handle, n, _lib.ptr(A), _lib.ptr(b), _lib.ptr(x), C.byref(ms))
    if rc == _lib.ERR_NOT_PD:
        raise np.linalg.LinAlgError(lib.bq_last_error().decode())
    _lib.check(rc)
    return (x, ms.value) if return_factor_ms else x
