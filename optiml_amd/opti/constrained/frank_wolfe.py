from ... import _lib
from ._base import BoxConstrainedQuadraticOptimizer

__all__ = ['FrankWolfe']


class FrankWolfe(BoxConstrainedQuadraticOptimizer):
    """(Stabilised) Frank-Wolfe with exact line search on the box QP.

    Interface and semantics of optiml/opti/constrained/frank_wolfe.py:25-165: vertex y_i = ub_i where g_i < 0
    else lb_i, lower bound f + g'(y - x), stop on relative gap <= eps, optional trust box of relative size
    `t` in [0, 1) around x.  One panel product per iteration on the device.
    """
    _kind = _lib.FW
    _header = 'iter\t cost\t\t lb\t\t gap'

    def __init__(self, quad, ub, lb=None, x=None, t=0., eps=1e-6, tol=1e-8, max_iter=1000, callback=None,
                 callback_args=(), verbose=False):
        super(FrankWolfe, self).__init__(quad=quad, ub=ub, lb=lb, x=x, eps=eps, tol=tol, max_iter=max_iter,
                                         callback=callback, callback_args=callback_args, verbose=verbose)
        if not 0 <= t < 1:
            raise ValueError('t has to lie in [0, 1)')
        self.t = t

    def _solver_t(self):
        return self.t

    def _after_row(self, row):
        self.best_lb = float(row['r1'])
        self.gap = float(row['r2'])

    def _line(self, row):
        return '\n{:4d}\t{: 1.4e}\t{: 1.4e}\t{: 1.4e}'.format(int(row['iter']), float(row['f']), float(row['r1']),
                                                            float(row['r2']))
