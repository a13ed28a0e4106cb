__all__ = ['Optimizer', 'OptimizationFunction', 'Quadratic', 'KernelQuadratic']

from ._base import Optimizer, OptimizationFunction, Quadratic, KernelQuadratic
