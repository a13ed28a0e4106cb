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
from .._base import Optimizer, OptimizationFunction, Quadratic

__all__ = ['BoxConstrainedQuadraticOptimizer', 'AugmentedLagrangianQuadratic']


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

    _STATE_VECS = (('x', _lib.STATE_X), ('g', _lib.STATE_G), ('lp', _lib.STATE_MULT), ('lm', _lib.STATE_MULT),
                   ('mask_l', _lib.STATE_MASKS), ('mask_u', _lib.STATE_MASKS))

    def get_state(self):
        """bq_solver_get_state: what the reference's loop holds at the top of the next iteration, as a dict of host arrays and
        scalars (`kind, iter, f, best_lb, x, g` + `lp, lm` for InteriorPoint, boolean `mask_l, mask_u` for ActiveSet)."""
        st = _lib.SolverState()
        bufs = {name: np.empty(self.N) for name, _ in self._STATE_VECS}
        for name, buf in bufs.items():
            setattr(st, name, _lib.ptr(buf))
        _lib.check(self._lib.bq_solver_get_state(self._h, C.byref(st)))
        out = {'kind': int(st.kind), 'iter': int(st.iter), 'f': float(st.f), 'best_lb': float(st.best_lb)}
        for name, bit in self._STATE_VECS:
            if st.have & bit:
                out[name] = bufs[name] != 0.0 if name.startswith('mask') else bufs[name]
        return out

    def set_state(self, state):
        """bq_solver_set_state: continue from a dict made by get_state (or by hand: `x` is required, the rest is formed as at a start
        point when absent).  Only before the first run."""
        st = _lib.SolverState()
        st.iter, st.kind = int(state.get('iter', 0)), int(state.get('kind', -1))
        st.f, st.best_lb = float(state.get('f', np.nan)), float(state.get('best_lb', np.nan))
        keep = []
        for name, bit in self._STATE_VECS:
            if state.get(name) is not None:
                buf = _lib.as_f64(np.asarray(state[name], dtype=float), self.N, name)
                keep.append(buf)
                setattr(st, name, _lib.ptr(buf))
                st.have |= bit
        _lib.check(self._lib.bq_solver_set_state(self._h, C.byref(st)))

    def set_inner(self, rtol, max_iter):
        _lib.check(self._lib.bq_solver_set_inner(self._h, float(rtol), int(max_iter)))

    def inner_iters(self):
        n = C.c_int64(0)
        _lib.check(self._lib.bq_solver_inner_iters(self._h, C.byref(n)))
        return n.value

    def counter(self, which):
        """ActiveSet bookkeeping totals: _lib.COUNT_MINRES / COUNT_REFACTOR / COUNT_REUSED / COUNT_INNER."""
        n = C.c_int64(0)
        _lib.check(self._lib.bq_solver_counter(self._h, int(which), C.byref(n)))
        return n.value

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

    # -- checkpoint / resume: the reference can only be restarted from `x=` (constrained/_base.py:61-65) and loses
    #    InteriorPoint's multipliers (interior_point.py:181-186) and ActiveSet's masks (active_set.py:91-92) -----------------
    def get_state(self):
        """The state the last `minimize()` stopped in (max_iter, a callback's StopIteration), to be handed to `set_state` of a
        new optimizer on the same problem: a dict of NumPy arrays and scalars (picklable).  It describes the top of iteration
        `state['iter']`: the step decided in the last iteration is applied."""
        if getattr(self, '_state', None) is None:
            raise RuntimeError('no state: minimize() has not run')
        return dict(self._state)

    def set_state(self, state):
        """Continue where another run stopped: the next `minimize()` starts from `state` (iteration counter included, so `max_iter`
        keeps its meaning) instead of from `x`."""
        if 'x' not in state:
            raise ValueError('a state holds x at least')
        if state.get('kind', self._kind) != self._kind:
            raise ValueError('the state was taken from another kind of solver')
        self._resume = dict(state)
        self.x = np.array(state['x'], dtype=float)
        self.iter = int(state.get('iter', 0))
        return self

    def minimize(self):
        dev = self.f.device_problem()
        solver = _DeviceSolver(dev, self._kind, self.lb, self.ub, self.x, self.eps, self.max_iter, self._solver_t())
        self._solver = solver
        self._configure(solver)
        if getattr(self, '_resume', None) is not None:
            solver.set_state(self._resume)
            self._resume = None
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
            self._state = solver.get_state()
        finally:
            solver.close()
            self._solver = None
        if self.verbose:
            print('\n')
        return self

    def _configure(self, solver):
        """Solver-specific settings, applied before the first iteration."""

    def _finalize(self, solver):
        pass


class AugmentedLagrangianQuadratic(Quadratic):
    r"""Augmented-Lagrangian relaxation of   min 1/2 x'Qx + q'x : A x = b, lb <= x <= ub   (SURVEY 8(f).3).

    Interface of optiml/opti/constrained/_base.py:224-410 (ctor arguments, checks, `primal`, `rho`, `n_eq`, `dual_x`,
    `past_dual_x`, `constraints/function/jacobian/function_jacobian`).  The stacked matrix [A; -I; I] is never
    formed: the rows are the single equality row and the coordinates.  What the device path covers is what the
    SVC/SVR duals need (optiml/ml/svm/_base.py:680-694, :1192-1207): ONE equality row with b = 0, bounds lb / ub, no
    general G x <= h rows — anything else raises NotImplementedError.  The multipliers are ordered as in the
    reference: [mu; lambda_lb; lambda_ub].
    """

    def __init__(self, primal, A=None, b=None, G=None, h=None, lb=None, ub=None, rho=1):
        if not isinstance(primal, Quadratic):
            raise TypeError(f'{primal} is not an allowed quadratic function')
        OptimizationFunction.__init__(self, primal.ndim)
        self.primal = primal
        if G is None and h is not None:
            raise ValueError('incomplete inequality constraint (missing G)')
        if G is not None and h is None:
            raise ValueError('incomplete inequality constraint (missing h)')
        if A is None and b is not None:
            raise ValueError('incomplete equality constraint (missing A)')
        if A is not None and b is None:
            raise ValueError('incomplete equality constraint (missing b)')
        if G is not None:
            raise NotImplementedError('general inequality rows G x <= h are not built: use lb / ub')
        if not rho > 0:
            raise ValueError('rho must be must > 0')
        self.A = np.atleast_2d(A).astype(float) if A is not None else None
        self.b = None if b is None else np.atleast_1d(np.asarray(b, dtype=float))
        if self.A is not None:
            if self.A.shape != (1, self.ndim):
                raise NotImplementedError('a single equality row of length ndim is built')
            if self.b.size != 1 or self.b[0] != 0:
                raise NotImplementedError('the equality row is built for b = 0')
        self.lb = np.asarray(lb, dtype=float) if lb is not None else None
        self.ub = np.asarray(ub, dtype=float) if ub is not None else None
        for v, name in ((self.lb, 'lb'), (self.ub, 'ub')):
            if v is not None and v.shape != (self.ndim,):
                raise ValueError(f'{name} size does not match with Q')
        self.rho = rho
        self.n_eq = 0 if self.A is None else 1
        self.dual_x = np.zeros(self.n_eq + (self.ndim if self.lb is not None else 0) +
                               (self.ndim if self.ub is not None else 0))   # mu_lmbda
        self.past_dual_x = self.dual_x.copy()

    # the Hessian / linear term are the primal's (lazy for KernelQuadratic)
    @property
    def Q(self):
        return self.primal.Q

    @property
    def q(self):
        return self.primal.q

    def device_problem(self, ctx=None):
        return self.primal.device_problem(ctx)

    def release(self):
        self.primal.release()

    @property
    def AG(self):
        """[A; -I; I] as a dense matrix (inspection / small problems only)."""
        rows = [] if self.A is None else [self.A]
        if self.lb is not None:
            rows.append(-np.eye(self.ndim))
        if self.ub is not None:
            rows.append(np.eye(self.ndim))
        return np.concatenate(rows) if rows else np.zeros((0, self.ndim))

    @property
    def bh(self):
        parts = [] if self.A is None else [self.b]
        if self.lb is not None:
            parts.append(-self.lb)
        if self.ub is not None:
            parts.append(self.ub)
        return np.concatenate(parts) if parts else np.zeros(0)

    def constraints(self, x):
        x = np.asarray(x, dtype=float)
        parts = [] if self.A is None else [self.A @ x - self.b]
        if self.lb is not None:
            parts.append(self.lb - x)
        if self.ub is not None:
            parts.append(x - self.ub)
        return np.concatenate(parts) if parts else np.zeros(0)

    def _clipped(self, c):
        cc = c.copy()
        cc[self.n_eq:] = np.clip(c[self.n_eq:], a_min=0, a_max=None)
        return cc

    def function(self, x):
        return self.function_jacobian(x)[0]

    def jacobian(self, x):
        return self.function_jacobian(x)[1]

    def function_jacobian(self, x):
        """constrained/_base.py:393-404 with one device product (value and gradient of the primal together)."""
        x = np.asarray(x, dtype=float)
        pf, pg = self.primal.function_jacobian(x)
        c = self.constraints(x)
        cc = self._clipped(c)
        fun = pf + self.dual_x @ c + 0.5 * self.rho * np.linalg.norm(cc) ** 2
        jac = pg.copy()
        k = 0
        if self.A is not None:
            jac += (self.dual_x[0] + self.rho * cc[0]) * self.A[0]
            k = 1
        n = self.ndim
        if self.lb is not None:
            jac -= self.dual_x[k:k + n] + self.rho * cc[k:k + n]
            k += n
        if self.ub is not None:
            jac += self.dual_x[k:k + n] + self.rho * cc[k:k + n]
        return fun, jac

    def hessian(self, x):
        raise NotImplementedError('the Hessian of the augmented Lagrangian is not built')
