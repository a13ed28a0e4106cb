"""Parameter schedules as iterators (interface of optiml/opti/unconstrained/stochastic/schedules.py:14-89): pass one as
`step_size=` or `momentum=`; the optimizer draws one value per iteration, exactly `epochs` of them, up front, and hands
the table to the device loop (`bq_al_solver_set_schedules`)."""
import itertools
import math

__all__ = ['constant', 'decaying', 'linear_annealing', 'repeater', 'sutskever_blend']


def constant(start):
    return itertools.repeat(start)


def decaying(start, decay):
    """start, start*decay, start*decay**2, ..."""
    return (start * decay ** k for k in itertools.count(0))


def linear_annealing(start, stop, n_steps):
    """n_steps equal increments from start (first value) towards stop, then stop for ever"""
    start, stop = float(start), float(stop)
    inc = (stop - start) / n_steps
    return itertools.chain((start + k * inc for k in range(n_steps)), itertools.repeat(stop))


def repeater(values, n):
    """every element of `values` n times in a row"""
    return (v for v in values for _ in range(n))


def sutskever_blend(max_momentum, stretch=250):
    """1 - 2**(-1 - log2(floor(k / stretch) + 1)) for k = 1, 2, ..., capped at max_momentum (Sutskever et al., 2013)"""
    return (float(min(1 - 2 ** (-1 - math.log2(k // stretch + 1)), max_momentum)) for k in itertools.count(1))
