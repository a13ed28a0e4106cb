from ... import _lib
from ._base import BoxConstrainedQuadraticOptimizer

__all__ = ['InteriorPoint']


class InteriorPoint(BoxConstrainedQuadraticOptimizer):
    """Primal-dual feasible interior point for the box QP (Cholesky of H = Q + diag each iteration).

    Interface and semantics of optiml/opti/constrained/interior_point.py:45-281: eps defaults to 1e-10, stop
    on relative primal-dual gap, mu = (f - p) / (4 n^2), 0.9995 of the maximum feasible step.  g_x keeps the
    gradient at the starting point, as in the reference (:180).  The factorisation and both triangular solves
    run on the device (blocked right-looking Cholesky on fp64 matrix cores).
    """
    _kind = _lib.IP
    _header = 'iter\t cost\t\t p\t\t gap'

    def __init__(self, quad, ub, lb=None, x=None, eps=1e-10, tol=1e-8, max_iter=1000, callback=None,
                 callback_args=(), verbose=False):
        super(InteriorPoint, self).__init__(quad=quad, ub=ub, lb=lb, x=x, eps=eps, tol=tol, max_iter=max_iter,
                                            callback=callback, callback_args=callback_args, verbose=verbose)

    def _after_row(self, row):
        self.p = float(row['r1'])
        self.gap = float(row['r2'])

    def _line(self, row):
        return '\n{:4d}\t{: 1.4e}\t{: 1.4e}\t{: 1.4e}'.format(int(row['iter']), float(row['f']), float(row['r1']),
                                                            float(row['r2']))

    def _finalize(self, solver):
        self.lp = solver.get(_lib.GET_LP)
        self.lm = solver.get(_lib.GET_LM)
