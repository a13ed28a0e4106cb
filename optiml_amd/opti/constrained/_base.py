"""Base of the box-constrained QP solvers:  min 1/2 x'Qx + q'x  s.t.  lb <= x <= ub.

Interface of optiml/opti/constrained/_base.py:10-73 (ctor arguments, defaults, type checks, mid-box start).
The iteration itself runs device-resident in libbcqp_hip.so; this class only drives `bq_solver_run` in
chunks and replays the per-iteration records to `verbose` printing and to the user callback.

Callback contract (optiml/opti/_base.py:119-127): `callback(opt, *callback_args)` once per iteration at the
top of the loop, may raise StopIteration.  An arbitrary callback forces one host round-trip per iteration
(x and g_x are downloaded before each call).  A callback that only needs `opt.iter` / `opt.f_x` can opt in
to batched replay by carrying the attribute `_bq_needs_state = False` (SVC/SVR's loss-history hook does).
"""
import ctypes as C
from abc import ABC

import numpy as np

from ... import _lib
from .._base import Optimizer, Quadratic

__all__ = ['BoxConstrainedQuadraticOptimizer']


class _DeviceSolver:
    def __init__(self, problem, kind, lb, ub, x0, eps, max_iter, t=0.0):
        self._lib = _lib.load()
        self.problem = problem
        self.N = problem.dims()[0]
        self._h = C.c_void_p()
        lb = _lib.as_f64(lb, self.N, 'lb')
        ub = _lib.as_f64(ub, self.N, 'ub')
        x0 = _lib.as_f64(x0, self.N, 'x')
        _lib.check(self._lib.bq_solver_create(problem.handle, kind, _lib.ptr(lb), _lib.ptr(ub), _lib.ptr(x0),
                                              float(eps), int(max_iter), float(t), C.byref(self._h)))

    def run(self, max_steps):
        stats = np.zeros(max_steps, dtype=_lib.STAT_DTYPE)
        n, status = C.c_int64(0), C.c_int(0)
        try:
            _lib.check(self._lib.bq_solver_run(self._h, max_steps, stats.ctypes.data_as(C.POINTER(_lib.IterStat)),
                                               max_steps, C.byref(n), C.byref(status)))
        except _lib.BcqpError as err:
            # the exceptions scipy's cho_factor raises inside the reference solvers
            if err.code == _lib.ERR_NOT_PD:
                raise np.linalg.LinAlgError(str(err)) from None
            if err.code == _lib.ERR_NONFINITE:
                raise ValueError('array must not contain infs or NaNs') from None
            raise
        return stats[:n.value], _lib.STATUS[status.value]

    def state(self):
        """(iter, status, f_x) as of the end of the last run."""
        it, st, f = C.c_int64(0), C.c_int(0), C.c_double(0)
        _lib.check(self._lib.bq_solver_state(self._h, C.byref(it), C.byref(st), C.byref(f)))
        return it.value, _lib.STATUS.get(st.value, 'unknown'), f.value

    def get(self, what):
        out = np.empty(self.N)
        _lib.check(self._lib.bq_solver_get(self._h, what, _lib.ptr(out)))
        return out

    def close(self):
        if self._h:
            self._lib.bq_solver_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class BoxConstrainedQuadraticOptimizer(Optimizer, ABC):
    _kind = None          # _lib.PG / FW / AS / IP
    _header = ''
    chunk = 256           # iterations per device-resident run when no per-iteration host state is needed

    def __init__(self, quad, ub, lb=None, x=None, eps=1e-6, tol=1e-8, max_iter=1000, callback=None,
                 callback_args=(), verbose=False):
        if not isinstance(quad, Quadratic):
            raise TypeError(f'{quad} is not an allowed quadratic function')
        ub = np.asarray(ub, dtype=float)
        lb = np.zeros_like(ub) if lb is None else np.asarray(lb, dtype=float)
        super(BoxConstrainedQuadraticOptimizer, self).__init__(f=quad,
                                                               x=x if x is not None else (lb + ub) / 2,
                                                               eps=eps, tol=tol, max_iter=max_iter,
                                                               callback=callback, callback_args=callback_args,
                                                               verbose=verbose)
        self.lb = lb
        self.ub = ub

    # -- hooks for the concrete solvers -------------------------------------------------------------
    def _solver_t(self):
        return 0.0

    def _line(self, row):
        raise NotImplementedError

    def _after_row(self, row):
        """Expose solver-specific scalars of an iteration record as attributes."""

    def _needs_state(self):
        if self.f.ndim <= 3:
            return True   # x0/x1 histories are appended every iteration (optiml/opti/_base.py:121-124)
        if callable(self._callback):
            return getattr(self._callback, '_bq_needs_state', True)
        return False

    def minimize(self):
        dev = self.f.device_problem()
        solver = _DeviceSolver(dev, self._kind, self.lb, self.ub, self.x, self.eps, self.max_iter, self._solver_t())
        self._solver = solver
        if self.verbose:
            print(self._header, end='')
        step_mode = self._needs_state()
        stop = False
        try:
            while not stop:
                rows, status = solver.run(1 if step_mode else self.chunk)
                for row in rows:
                    self.iter = int(row['iter'])
                    self.f_x = float(row['f'])
                    self._after_row(row)
                    if step_mode:
                        self.x = solver.get(_lib.GET_X)
                        self.g_x = solver.get(_lib.GET_G)
                    if self.is_verbose():
                        print(self._line(row), end='')
                    try:
                        self.callback()
                    except StopIteration:
                        stop = True
                        break
                if status != 'unknown':
                    self.status = status
                    break
            self.x = solver.get(_lib.GET_X if stop else _lib.GET_X_NOW)
            self.g_x = solver.get(_lib.GET_G if stop else _lib.GET_G_NOW)
            if not stop:
                # ActiveSet re-evaluates f after its last move (active_set.py:160); for the others this is the
                # value of the last record
                self.f_x = solver.state()[2]
            self._finalize(solver)
        finally:
            solver.close()
            self._solver = None
        if self.verbose:
            print('\n')
        return self

    def _finalize(self, solver):
        pass
