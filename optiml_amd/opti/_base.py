"""Objective and optimizer bases of the box-constrained QP path (device-backed).

Mirrors the interface of optiml/opti/_base.py: `Optimizer` (:9-181, non-Lagrangian part),
`OptimizationFunction` (:184-225) and `Quadratic` (:228-300).  The arithmetic of
`Quadratic.function/jacobian` (:282, :291) runs in libbcqp_hip.so on the Hessian resident in HBM.
"""
import ctypes as C
from abc import ABC

import numpy as np

from .. import _lib
from ..device import get_context

__all__ = ['Optimizer', 'OptimizationFunction', 'Quadratic', 'KernelQuadratic']


class OptimizationFunction(ABC):

    def __init__(self, ndim=2):
        self.ndim = ndim

    def x_star(self):
        return np.full(fill_value=np.nan, shape=self.ndim)

    def f_star(self):
        return np.inf

    def args(self):
        return ()

    def function(self, x):
        raise NotImplementedError

    def jacobian(self, x):
        raise NotImplementedError

    def hessian(self, x):
        raise NotImplementedError

    def function_jacobian(self, *args, **kwargs):
        return self.function(*args, **kwargs), self.jacobian(*args, **kwargs)

    def __call__(self, *args, **kwargs):
        return self.function(*args, **kwargs)


class _DeviceProblem:
    """Owner of one bq_problem handle (freed with the object)."""

    def __init__(self, ctx, handle):
        self.ctx = ctx
        self._h = handle
        self._lib = _lib.load()

    @property
    def handle(self):
        return self._h

    def dims(self):
        a, b, c, d = (C.c_int64(0) for _ in range(4))
        _lib.check(self._lib.bq_problem_dims(self._h, C.byref(a), C.byref(b), C.byref(c), C.byref(d)))
        return a.value, b.value, c.value, d.value

    def layout(self):
        """{'packed': the packed lower tile rows (kernel-built panels; a dense Q == Q') or row blocks, 'streamed', 'panel_bytes'}"""
        a, b, c = C.c_int(0), C.c_int(0), C.c_int64(0)
        _lib.check(self._lib.bq_problem_layout(self._h, C.byref(a), C.byref(b), C.byref(c)))
        return {'packed': bool(a.value), 'streamed': bool(b.value), 'panel_bytes': c.value}

    def matvec(self, v):
        N = self.dims()[0]
        v = _lib.as_f64(v, N, 'v')
        out = np.empty(N)
        _lib.check(self._lib.bq_problem_matvec(self._h, _lib.ptr(v), _lib.ptr(out)))
        return out

    def eval(self, x, want_grad=True):
        N = self.dims()[0]
        x = _lib.as_f64(x, N, 'x')
        f = C.c_double(0)
        g = np.empty(N) if want_grad else None
        _lib.check(self._lib.bq_problem_eval(self._h, _lib.ptr(x), C.byref(f), _lib.ptr(g)))
        return f.value, g

    def x_star(self):
        """(x, method, minres_iterations): the unconstrained minimiser by Cholesky ('cholesky') or, when Q is not positive
        definite, scipy's default MINRES ('minres') — optiml/opti/_base.py:259-269."""
        N = self.dims()[0]
        x = np.empty(N)
        method, iters = C.c_int(0), C.c_int64(0)
        _lib.check(self._lib.bq_problem_x_star(self._h, _lib.ptr(x), C.byref(method), C.byref(iters)))
        return x, ('cholesky', 'minres')[method.value], iters.value

    def gram_matvec(self, w):
        n = self.dims()[1]
        w = _lib.as_f64(w, n, 'w')
        out = np.empty(n)
        _lib.check(self._lib.bq_problem_gram_matvec(self._h, _lib.ptr(w), _lib.ptr(out)))
        return out

    def panel_rows(self, row0=None, nrows=None):
        _, n, r0, r1 = self.dims()
        row0 = r0 if row0 is None else row0
        nrows = r1 - row0 if nrows is None else nrows
        out = np.empty((nrows, n))
        _lib.check(self._lib.bq_problem_panel_rows(self._h, row0, nrows, _lib.ptr(out)))
        return out

    def placement(self):
        """Launch time (ms) of the panel product on every placement of the panel that was tried (`tune_placement`); [] if none."""
        tried = C.c_int(0)
        ms = np.zeros(32)
        _lib.check(self._lib.bq_problem_placement(self._h, C.byref(tried), _lib.ptr(ms), 32))
        return [float(v) for v in ms[:tried.value]]

    def time_matvec(self, reps=10):
        ms = C.c_double(0)
        _lib.check(self._lib.bq_problem_time_matvec(self._h, reps, C.byref(ms)))
        return ms.value

    def close(self):
        if self._h:
            self._lib.bq_problem_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Quadratic(OptimizationFunction):
    r"""f(x) = 1/2 x'Qx + q'x with a dense Hessian (same ctor and checks as optiml/opti/_base.py:230-256).

    `storage='f32'` keeps the device copy of Q in fp32 (fp64 accumulation); the default is fp64.

    `symmetric=None` (default): the device copy is the packed lower triangle (half the memory, half the bytes per product) when
    `Q == Q.T` holds exactly — compared on the device while Q is uploaded — and whole rows otherwise, so a Q that is not
    symmetric keeps NumPy's `Q @ x` (the reference never checks, optiml/opti/_base.py:249-256).  `symmetric=True`: the caller
    vouches for it, only the lower triangle of Q is read (LAPACK's uplo='L').  `symmetric=False`: whole rows whatever Q is.
    `tune_placement=True`: as for `KernelQuadratic` (packed copies of >= 1 GB only).
    """

    def __init__(self, Q, q, storage='f64', symmetric=None, tune_placement=False, expected_products=0):
        Q = np.array(Q, dtype=float)
        q = np.array(q, dtype=float)
        n = len(Q)
        super(Quadratic, self).__init__(n)
        if n <= 1:
            raise ValueError('Q is too small')
        if Q.ndim != 2 or n != Q.shape[1]:
            raise ValueError('Q is not square')
        self.Q = Q
        if q.size != n:
            raise ValueError('q size does not match with Q')
        self.q = q
        self.storage = storage
        self.symmetric = symmetric
        self.tune_placement = bool(tune_placement)
        self.expected_products = float(expected_products or 0)
        self._dev = None

    # -- device residency -------------------------------------------------------------------
    def _build_problem(self, ctx):
        lib = _lib.load()
        h = C.c_void_p()
        Qc = np.ascontiguousarray(self.Q)
        flags = _lib.F32 if self.storage == 'f32' else _lib.F64
        if self.symmetric is not None:
            flags |= _lib.DENSE_LOWER if self.symmetric else _lib.DENSE_ROWS
        if self.tune_placement:
            ctx.set_placement_budget(self.expected_products)
            flags |= _lib.PLACE_PANEL
        _lib.check(lib.bq_problem_create_dense(ctx.handle, self.ndim, _lib.ptr(Qc), _lib.ptr(self.q.ravel()), flags, C.byref(h)))
        return _DeviceProblem(ctx, h)

    def device_problem(self, ctx=None):
        ctx = ctx or get_context()
        if self._dev is None or self._dev.ctx is not ctx:
            self._dev = self._build_problem(ctx)
        return self._dev

    def release(self):
        """Free the device copy (it is rebuilt on demand)."""
        if self._dev is not None:
            self._dev.close()
            self._dev = None

    # -- reference interface ------------------------------------------------------------------
    def x_star(self):
        """optiml/opti/_base.py:259-269: cho_solve(cho_factor(Q), -q), or minres(Q, -q)[0] when Q is not positive definite —
        both on the device (blocked MFMA Cholesky of the resident Hessian; MINRES on the panel product)."""
        if not hasattr(self, 'x_opt'):
            self.x_opt, self.x_opt_method, self.x_opt_iters = self.device_problem().x_star()
        return self.x_opt

    def f_star(self):
        return self.function(self.x_star())   # optiml/opti/_base.py:271-272

    def function(self, x):
        return self.device_problem().eval(x, want_grad=False)[0]

    def jacobian(self, x):
        return self.device_problem().eval(x)[1]

    def function_jacobian(self, x):
        return self.device_problem().eval(x)

    def hessian(self, x):
        return self.Q


class KernelQuadratic(Quadratic):
    """Lazy form of the SVM dual Hessian: holds X (and labels), never builds the n x n matrix on the host.

    structure 'svc': Q = K*yy' + yy' (+ diag*I), n variables      (optiml/ml/svm/_base.py:552-555, 628)
    structure 'svr': Q = [[K,-K],[-K,K]] + ee', 2n variables      (optiml/ml/svm/_base.py:1096-1099, 1178)
    structure 'plain': Q = K (+ diag*I)
    `rank_one=False` leaves out the yy' / ee' term of the regularised intercept: the reg_intercept=False duals
    Q = K*yy' and [[K,-K],[-K,K]] (:552-555, :1096-1099) that the augmented-Lagrangian path solves with an equality row.
    `.Q` materialises the dense matrix from the device panel on demand (inspection / small problems only).
    `tune_placement=True` lets the library pick the fastest of up to three allocations of the panel (`bq_problem_placement`);
    `expected_products` (the iteration cap of the solve that follows; `SVC` / `SVR.fit` pass their `max_iter`) lets the time that
    choice may take grow with the work ahead (`bq_ctx_set_placement_budget`: 2 % of it, 200 ms at least, 5 s at most).
    `storage='stream'` keeps NO panel: every product recomputes the Gram tiles on the MFMA (for n^2 beyond HBM; first-order
    solvers only, no `.Q`).
    """

    _STRUCT = {'plain': _lib.PLAIN, 'svc': _lib.SVC, 'svr': _lib.SVR}

    def __init__(self, X, q, structure, kernel, y=None, diag=0.0, storage='f64', rank_one=True, full_panel=False,
                 tune_placement=False, expected_products=0):
        X = np.ascontiguousarray(X, dtype=float)
        if structure not in self._STRUCT:
            raise ValueError(f'unknown structure {structure}')
        n = X.shape[0]
        N = 2 * n if structure == 'svr' else n
        q = np.array(q, dtype=float)
        OptimizationFunction.__init__(self, N)
        if n <= 1:
            raise ValueError('Q is too small')
        if q.size != N:
            raise ValueError('q size does not match with Q')
        if structure == 'svc' and (y is None or len(y) != n):
            raise ValueError('labels are required for the svc structure')
        if storage not in _lib.STORAGE:
            raise ValueError(f'unknown storage {storage}')
        self.X, self.q, self.structure, self.kernel = X, q, structure, kernel
        self.y = None if y is None else np.ascontiguousarray(y, dtype=float)
        self.diag = float(diag)
        self.rank_one = bool(rank_one)
        self.full_panel = bool(full_panel)   # whole rows instead of the packed triangle (2x the memory)
        # BQ_PLACE_PANEL: time the product on the fresh panel and try up to two more allocations if it streams slowly (panels of
        # >= 1 GB; for solvers whose every iteration streams the panel)
        self.tune_placement = bool(tune_placement)
        self.expected_products = float(expected_products or 0)
        self.storage = storage
        self.kind, self.gamma, self.coef0, self.degree = kernel.device_spec(X)
        self._dev = None

    def _build_problem(self, ctx):
        lib = _lib.load()
        h = C.c_void_p()
        n, d = self.X.shape
        if self.tune_placement:
            ctx.set_placement_budget(self.expected_products)
        _lib.check(lib.bq_problem_create_kernel(
            ctx.handle, self._STRUCT[self.structure] | (0 if self.rank_one else _lib.NO_RANK_ONE) |
            (_lib.FULL_PANEL if self.full_panel else 0) | (_lib.PLACE_PANEL if self.tune_placement else 0), n, d, _lib.ptr(self.X), _lib.ptr(self.y), self.kind,
            self.gamma, self.coef0, self.degree, self.diag, _lib.ptr(self.q),
            _lib.STORAGE[self.storage], C.byref(h)))
        return _DeviceProblem(ctx, h)

    def gram(self):
        """The n x n Gram matrix K (single-rank contexts).  The device keeps only the 256 x 256 tiles on or below the
        diagonal; the upper tiles are mirrored here."""
        dev = self.device_problem()
        if dev.ctx.world != 1:
            raise RuntimeError('materialising K needs the whole panel: single-rank contexts only')
        L = dev.panel_rows()
        if self.full_panel:
            return L
        n = L.shape[0]
        tile = np.arange(n) // 256
        upper = tile[None, :] > tile[:, None]
        return np.where(upper, L.T, L)

    @property
    def Q(self):
        dev = self.device_problem()
        if dev.ctx.world != 1:
            raise RuntimeError('materialising Q needs the whole panel: single-rank contexts only')
        K = self.gram()
        n = K.shape[0]
        if self.structure == 'plain':
            Q = K
        elif self.structure == 'svc':
            Q = (K + (1.0 if self.rank_one else 0.0)) * np.outer(self.y, self.y)
        else:
            P = K + (1.0 if self.rank_one else 0.0)
            Q = np.vstack((np.hstack((P, -P)), np.hstack((-P, P))))
        if self.diag:
            Q = Q + self.diag * np.eye(Q.shape[0])
        return Q


class Optimizer(ABC):
    """State and callback contract of optiml/opti/_base.py:9-181, including the (augmented-)Lagrangian bookkeeping
    of the callback (:96-117): primal value, duality gap, past_x."""

    def __init__(self, f, x=None, eps=1e-6, tol=1e-8, max_iter=1000, callback=None, callback_args=(),
                 random_state=None, verbose=False):
        if not isinstance(f, OptimizationFunction):
            raise TypeError(f'{f} is not an allowed optimization function')
        self.f = f
        if x is None:
            # optiml/opti/_base.py:36-57: uniform(0, 1) start (the multipliers of an augmented Lagrangian live in f)
            x = np.random.uniform if random_state is None else np.random.RandomState(random_state).uniform
        if callable(x):
            self.x = x(size=f.ndim)
        else:
            self.x = np.asarray(x, dtype=float)
        self.f_x = np.nan
        if self.is_lagrangian_dual():
            self.past_x = self.x.copy()
            self.primal_f_x = np.nan
            self.dgap = np.nan
        self.g_x = np.zeros(0)
        self.eps = eps
        self.tol = tol
        if not max_iter > 0:
            raise ValueError('max_iter must be > 0')
        self.max_iter = max_iter
        self.iter = 0
        self.status = 'unknown'
        if self.f.ndim <= 3 or (hasattr(self.f, 'primal') and self.f.primal.ndim <= 3):
            self.x0_history = []
            self.x1_history = []
            self.f_x_history = []
        self._callback = callback
        self.callback_args = callback_args
        self.random_state = random_state
        self.verbose = verbose

    def is_lagrangian_dual(self):
        return hasattr(self.f, 'primal')

    def is_augmented_lagrangian_dual(self):
        return self.is_lagrangian_dual() and hasattr(self.f, 'rho')

    def callback(self, args=()):
        if self.is_lagrangian_dual():
            # primal_f_x is set from the device iteration record by the driver (one product per iteration, not three)
            self.dgap = abs((self.primal_f_x - self.f_x) / max(abs(self.primal_f_x), 1))
            if self.is_verbose():
                print('\tpcost: {: 1.4e}'.format(self.primal_f_x), end='')
                print('\tdgap: {: 1.4e}'.format(self.dgap), end='')
            if self.f.primal.ndim == 2:
                self.x0_history.append(self.x[0])
                self.x1_history.append(self.x[1])
                self.f_x_history.append(self.primal_f_x)
            if callable(self._callback):
                self._callback(self, *args, *self.callback_args)
            self.past_x = self.x.copy()
        else:
            if self.f.ndim <= 3:
                self.x0_history.append(self.x[0])
                self.x1_history.append(self.x[1])
                self.f_x_history.append(self.f_x)
            if callable(self._callback):
                self._callback(self, *args, *self.callback_args)

    def is_verbose(self):
        return self.verbose and not self.iter % self.verbose

    def minimize(self):
        raise NotImplementedError
