__all__ = ['schedules', 'StochasticOptimizer', 'StochasticMomentumOptimizer', 'StochasticGradientDescent', 'Adam', 'AMSGrad',
           'AdaMax', 'AdaGrad', 'AdaDelta', 'RMSProp']

from . import schedules
from ._base import StochasticOptimizer, StochasticMomentumOptimizer
from .rules import StochasticGradientDescent, Adam, AMSGrad, AdaMax, AdaGrad, AdaDelta, RMSProp
