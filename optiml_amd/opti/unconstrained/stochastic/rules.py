"""The seven update rules of optiml/opti/unconstrained/stochastic (ctor arguments, defaults and checks of
gradient_descent.py:24-62, adam.py:31-93, amsgrad.py:30-92, adamax.py:30-92, adagrad.py:22-79, adadelta.py:24-86,
rmsprop.py:26-91).  The update formulas themselves are in csrc/bq_al.hip (al_update_kernel)."""
import warnings

import numpy as np

from .... import _lib
from ._base import StochasticOptimizer, StochasticMomentumOptimizer

__all__ = ['StochasticGradientDescent', 'Adam', 'AMSGrad', 'AdaMax', 'AdaGrad', 'AdaDelta', 'RMSProp']


class StochasticGradientDescent(StochasticMomentumOptimizer):
    _rule = _lib.RULE_SGD

    def __init__(self, f, x=None, batch_size=None, eps=1e-6, tol=1e-8, epochs=1000, step_size=0.01,
                 momentum_type='none', momentum=0.9, callback=None, callback_args=(), shuffle=True,
                 random_state=None, verbose=False):
        super(StochasticGradientDescent, self).__init__(
            f=f, x=x, step_size=step_size, momentum_type=momentum_type, momentum=momentum, batch_size=batch_size,
            eps=eps, tol=tol, epochs=epochs, callback=callback, callback_args=callback_args, shuffle=shuffle,
            random_state=random_state, verbose=verbose)


class _AdamFamily(StochasticMomentumOptimizer):
    _default_step = 0.001

    def __init__(self, f, x=None, batch_size=None, eps=1e-6, tol=1e-8, epochs=1000, step_size=None,
                 momentum_type='none', momentum=0.9, beta1=0.9, beta2=0.999, offset=1e-8, callback=None,
                 callback_args=(), shuffle=True, random_state=None, verbose=False):
        super(_AdamFamily, self).__init__(
            f=f, x=x, step_size=self._default_step if step_size is None else step_size, momentum_type=momentum_type,
            momentum=momentum, batch_size=batch_size, eps=eps, tol=tol, epochs=epochs, callback=callback,
            callback_args=callback_args, shuffle=shuffle, random_state=random_state, verbose=verbose)
        if not 0 <= beta1 < 1:
            raise ValueError('beta1 has to lie in [0, 1)')
        self.beta1 = beta1
        if not 0 <= beta2 < 1:
            raise ValueError('beta2 has to lie in [0, 1)')
        self.beta2 = beta2
        if not self.beta1 < np.sqrt(self.beta2):
            warnings.warn('constraint from convergence analysis for adam not satisfied')
        if not offset > 0:
            raise ValueError('offset must be > 0')
        self.offset = offset


class Adam(_AdamFamily):
    _rule = _lib.RULE_ADAM


class AMSGrad(_AdamFamily):
    _rule = _lib.RULE_AMSGRAD


class AdaMax(_AdamFamily):
    _rule = _lib.RULE_ADAMAX
    _default_step = 0.002


class AdaGrad(StochasticOptimizer):
    _rule = _lib.RULE_ADAGRAD

    def __init__(self, f, x=None, batch_size=None, eps=1e-6, tol=1e-8, epochs=1000, step_size=1., offset=1e-8,
                 callback=None, callback_args=(), shuffle=True, random_state=None, verbose=False):
        super(AdaGrad, self).__init__(f=f, x=x, step_size=step_size, batch_size=batch_size, eps=eps, tol=tol,
                                      epochs=epochs, callback=callback, callback_args=callback_args,
                                      shuffle=shuffle, random_state=random_state, verbose=verbose)
        if not offset > 0:
            raise ValueError('offset must be > 0')
        self.offset = offset


class AdaDelta(StochasticOptimizer):
    _rule = _lib.RULE_ADADELTA

    def __init__(self, f, x=None, batch_size=None, eps=1e-6, tol=1e-8, epochs=1000, step_size=1., decay=0.9,
                 offset=1e-6, callback=None, callback_args=(), shuffle=True, random_state=None, verbose=False):
        super(AdaDelta, self).__init__(f=f, x=x, step_size=step_size, batch_size=batch_size, eps=eps, tol=tol,
                                       epochs=epochs, callback=callback, callback_args=callback_args,
                                       shuffle=shuffle, random_state=random_state, verbose=verbose)
        if not 0 <= decay < 1:
            raise ValueError('decay has to lie in [0, 1)')
        self.decay = decay
        if not offset > 0:
            raise ValueError('offset must be > 0')
        self.offset = offset


class RMSProp(StochasticMomentumOptimizer):
    _rule = _lib.RULE_RMSPROP

    def __init__(self, f, x=None, step_size=0.001, momentum_type='none', momentum=0.9, batch_size=None, eps=1e-6,
                 tol=1e-8, epochs=1000, decay=0.9, offset=1e-8, callback=None, callback_args=(), shuffle=True,
                 random_state=None, verbose=False):
        super(RMSProp, self).__init__(f=f, x=x, step_size=step_size, momentum_type=momentum_type, momentum=momentum,
                                      batch_size=batch_size, eps=eps, tol=tol, epochs=epochs, callback=callback,
                                      callback_args=callback_args, shuffle=shuffle, random_state=random_state,
                                      verbose=verbose)
        if not 0 <= decay < 1:
            raise ValueError('decay has to lie in [0, 1)')
        self.decay = decay
        if not offset > 0:
            raise ValueError('offset must be > 0')
        self.offset = offset
