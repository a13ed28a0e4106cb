__all__ = ['StochasticOptimizer', 'StochasticMomentumOptimizer', 'StochasticGradientDescent', 'Adam', 'AMSGrad',
           'AdaMax', 'AdaGrad', 'AdaDelta', 'RMSProp']

from ._base import StochasticOptimizer, StochasticMomentumOptimizer
from .rules import StochasticGradientDescent, Adam, AMSGrad, AdaMax, AdaGrad, AdaDelta, RMSProp
