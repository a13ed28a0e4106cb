"""Full-batch drivers of the first-order update rules on a quadratic or its augmented Lagrangian.

Interface of optiml/opti/unconstrained/stochastic/_base.py (`StochasticOptimizer` :13-158,
`StochasticMomentumOptimizer` :161-243): ctor arguments, defaults and checks, `epochs`/`epoch`/`step`, the verbose
line formats, the callback contract.  The loop itself (optiml/opti/unconstrained/stochastic/*.py `minimize` +
the multiplier update of optiml/opti/_base.py:129-146) runs device-resident in libbcqp_hip.so
(`bq_al_solver_create`, csrc/bq_al.hip): one panel product per iteration.

Scope: the objective is a `Quadratic` (plain rule, runs `epochs` iterations) or an
`AugmentedLagrangianQuadratic` (the SVC/SVR dual branch, optiml/ml/svm/_base.py:638-723).  Such objectives have no
samples to draw mini batches from (`f.args()` is empty), so `batch_size` must stay None as in the reference.
`step_size` / `momentum` may be scalars, iterables (one value is drawn per iteration — `schedules.py`) or, for the
step size, a callable returning an iterator (the reference calls it afresh every iteration and takes its first value,
i.e. a constant: stochastic/_base.py:88-93).
"""
import ctypes as C
import itertools
from abc import ABC
from collections.abc import Iterable

import numpy as np

from .... import _lib
from ..._base import Optimizer, Quadratic
from ...constrained._base import AugmentedLagrangianQuadratic

__all__ = ['StochasticOptimizer', 'StochasticMomentumOptimizer']


class _AlDeviceSolver:
    """Owner of one augmented-Lagrangian bq_solver handle."""

    def __init__(self, problem, prm, a, lb, ub, x0, dual0):
        self._lib = _lib.load()
        self.N = problem.dims()[0]
        self._h = C.c_void_p()
        vec = lambda v, name: None if v is None else _lib.as_f64(v, self.N, name)   # noqa: E731
        a, lb, ub, x0 = vec(a, 'A'), vec(lb, 'lb'), vec(ub, 'ub'), vec(x0, 'x')
        self.n_dual = (a is not None) + self.N * ((lb is not None) + (ub is not None))
        dual0 = None if dual0 is None or self.n_dual == 0 else _lib.as_f64(dual0, self.n_dual, 'dual_x')
        _lib.check(self._lib.bq_al_solver_create(problem.handle, C.byref(prm), _lib.ptr(a), _lib.ptr(lb), _lib.ptr(ub),
                                                 _lib.ptr(x0), _lib.ptr(dual0), C.byref(self._h)))

    def run(self, max_steps):
        stats = np.zeros(max_steps, dtype=_lib.STAT_DTYPE)
        n, status = C.c_int64(0), C.c_int(0)
        _lib.check(self._lib.bq_solver_run(self._h, max_steps, stats.ctypes.data_as(C.POINTER(_lib.IterStat)),
                                           max_steps, C.byref(n), C.byref(status)))
        return stats[:n.value], _lib.STATUS[status.value]

    def set_schedules(self, steps, moms, count):
        steps = None if steps is None else _lib.as_f64(steps, count, 'step sizes')
        moms = None if moms is None else _lib.as_f64(moms, count, 'momenta')
        _lib.check(self._lib.bq_al_solver_set_schedules(self._h, _lib.ptr(steps), _lib.ptr(moms), count))

    def state(self):
        it, st, f = C.c_int64(0), C.c_int(0), C.c_double(0)
        _lib.check(self._lib.bq_solver_state(self._h, C.byref(it), C.byref(st), C.byref(f)))
        return it.value, _lib.STATUS.get(st.value, 'unknown'), f.value

    def get(self, what):
        out = np.empty(self.n_dual if what == _lib.GET_DUAL else self.N)
        if out.size:
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


class StochasticOptimizer(Optimizer, ABC):
    _rule = None     # _lib.RULE_*
    chunk = 256      # iterations per device-resident run when no per-iteration host state is needed

    def __init__(self, f, x=None, step_size=0.01, batch_size=None, eps=1e-6, tol=1e-8, epochs=1000, callback=None,
                 callback_args=(), shuffle=True, random_state=None, verbose=False):
        super(StochasticOptimizer, self).__init__(f=f, x=x, eps=eps, tol=tol, max_iter=epochs, callback=callback,
                                                  callback_args=callback_args, random_state=random_state,
                                                  verbose=verbose)
        self._step_schedule = None
        if isinstance(step_size, Iterable):      # one value per iteration, drawn here (stochastic/_base.py:88-89)
            self._step_schedule = self._draw(step_size, epochs, 'step_size')
            self.step_size = float(self._step_schedule[0])
        elif callable(step_size):                # called afresh every iteration: always its first value (:90-91)
            self.step_size = float(next(iter(step_size(*f.args()))))
        else:
            if not step_size > 0:
                raise ValueError('step_size must be > 0 or a callable or an iterator')
            self.step_size = float(step_size)
        self.epochs = epochs
        self.epoch = 0
        self.shuffle = shuffle
        self.step = 0
        if batch_size is not None:
            # a quadratic has no samples: the reference fails on len(f.args()[0]) here
            raise NotImplementedError('mini batches need a sampled objective: batch_size must be None')
        self.batch_size = None

    @staticmethod
    def _draw(values, count, name):
        table = list(itertools.islice(iter(values), int(count)))
        if len(table) < count:
            raise ValueError(f'the {name} iterable ends after {len(table)} values, {count} are needed')
        return np.asarray(table, dtype=float)

    def is_batch_end(self):
        return True   # full batch

    def is_verbose(self):
        return self.verbose and not self.epoch % self.verbose

    def _print_header(self):
        if self.verbose:
            print('epoch\titer\t cost\t', end='')

    def _print_info(self):
        if self.is_verbose():
            print('\n{:4d}\t{:4d}\t{: 1.4e}'.format(self.epoch, self.iter, self.f_x), end='')

    # -- rule parameters ---------------------------------------------------------------------------------
    def _params(self):
        prm = _lib.AlParams()
        prm.rule = self._rule
        prm.momentum_type = _lib.MOM[getattr(self, 'momentum_type', 'none')]
        prm.step_size = self.step_size
        prm.momentum = float(getattr(self, 'momentum', 0.0)) if prm.momentum_type else 0.0
        prm.beta1 = float(getattr(self, 'beta1', 0.0))
        prm.beta2 = float(getattr(self, 'beta2', 0.0))
        prm.decay = float(getattr(self, 'decay', 0.0))
        prm.offset = float(getattr(self, 'offset', 1.0))
        prm.rho = float(getattr(self.f, 'rho', 1.0))
        prm.tol = float(self.tol)
        prm.epochs = int(self.epochs)
        return prm

    def _needs_state(self):
        nd = self.f.primal.ndim if self.is_lagrangian_dual() else self.f.ndim
        if nd <= 3:
            return True   # x0/x1 histories are appended every iteration (optiml/opti/_base.py:104-107, 121-124)
        if callable(self._callback):
            return getattr(self._callback, '_bq_needs_state', True)
        return False

    def minimize(self):
        f = self.f
        if isinstance(f, AugmentedLagrangianQuadratic):
            primal, a = f.primal, (None if f.A is None else f.A[0])
            lb, ub, dual0 = f.lb, f.ub, f.dual_x
        elif isinstance(f, Quadratic):
            primal, a, lb, ub, dual0 = f, None, None, None, None
        else:
            raise NotImplementedError('only Quadratic / AugmentedLagrangianQuadratic objectives run on the device')
        solver = _AlDeviceSolver(primal.device_problem(), self._params(), a, lb, ub, self.x, dual0)
        steps, moms = self._step_schedule, getattr(self, '_momentum_schedule', None)
        if getattr(self, 'momentum_type', 'none') == 'none':
            moms = None
        if steps is not None or moms is not None:
            solver.set_schedules(steps, moms, int(self.epochs))
        self._print_header()
        step_mode = self._needs_state()
        stop = False
        try:
            while not stop:
                rows, status = solver.run(1 if step_mode else self.chunk)
                for row in rows:
                    self.iter = int(row['iter'])
                    self.epoch = self.iter              # full batch: one epoch per evaluation
                    self.f_x = float(row['f'])
                    if self.is_lagrangian_dual():
                        self.primal_f_x = float(row['r1'])
                    if step_mode:
                        self.x = solver.get(_lib.GET_X)
                        self.g_x = solver.get(_lib.GET_G)
                    self._print_info()
                    try:
                        self.callback(f.args())
                    except StopIteration:
                        stop = True
                        break
                    self.epoch += 1
                if status != 'unknown':
                    self.status = status
                    break
            # after a callback stop the point is the one the last record was evaluated at; otherwise the current one
            # ('optimal': the point after the last update; 'stopped': no update follows the last evaluation)
            self.x = solver.get(_lib.GET_X if stop else _lib.GET_X_NOW)
            self.g_x = solver.get(_lib.GET_G)
            self.step = solver.get(_lib.GET_D)
            if self.is_lagrangian_dual():
                self.past_x = solver.get(_lib.GET_X)
                if self.is_augmented_lagrangian_dual():
                    f.past_dual_x = f.dual_x.copy()
                    f.dual_x = solver.get(_lib.GET_DUAL)
                    assert np.all(f.dual_x[f.n_eq:] >= 0)   # check_lagrangian_dual_conditions, optiml/opti/_base.py:163-169
        finally:
            solver.close()
        if self.verbose:
            print('\n')
        return self


class StochasticMomentumOptimizer(StochasticOptimizer, ABC):

    def __init__(self, f, x=None, step_size=0.01, momentum_type='none', momentum=0.9, batch_size=None, eps=1e-6,
                 tol=1e-8, epochs=1000, callback=None, callback_args=(), shuffle=True, random_state=None,
                 verbose=False):
        super(StochasticMomentumOptimizer, self).__init__(f=f, x=x, step_size=step_size, batch_size=batch_size,
                                                          eps=eps, tol=tol, epochs=epochs, callback=callback,
                                                          callback_args=callback_args, shuffle=shuffle,
                                                          random_state=random_state, verbose=verbose)
        if momentum_type not in ('polyak', 'nesterov', 'none'):
            raise ValueError(f'unknown momentum type {momentum_type}')
        self.momentum_type = momentum_type
        self._momentum_schedule = None
        if isinstance(momentum, Iterable):   # one value per iteration (stochastic/_base.py:242-243)
            self._momentum_schedule = self._draw(momentum, epochs, 'momentum')
            self.momentum = float(self._momentum_schedule[0])
        else:
            if not 0 <= momentum < 1:
                raise ValueError('momentum must be between 0 and 1 or an iterator')
            self.momentum = momentum
