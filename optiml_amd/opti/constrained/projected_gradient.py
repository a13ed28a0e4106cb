from ... import _lib
from ._base import BoxConstrainedQuadraticOptimizer

__all__ = ['ProjectedGradient']


class ProjectedGradient(BoxConstrainedQuadraticOptimizer):
    """Projected gradient with exact line search on the box QP.

    Same ctor, stopping rule (|projected gradient|_2 <= eps -> 'optimal', iter >= max_iter -> 'stopped') and
    thresholds (1e-12 activity, 1e-16 curvature) as optiml/opti/constrained/projected_gradient.py:76-143.
    Device formulation: one panel product Q d per iteration; g is carried as g += t Q d and the objective
    as 1/2 x'(g + q).  After `minimize()`: x, f_x, g_x, iter, status, and `ng` (last projected-gradient norm).
    """
    _kind = _lib.PG
    _header = 'iter\t cost\t\t gnorm'

    def _after_row(self, row):
        self.ng = float(row['r1'])

    def _line(self, row):
        return '\n{:4d}\t{: 1.4e}\t{: 1.4e}'.format(int(row['iter']), float(row['f']), float(row['r1']))
