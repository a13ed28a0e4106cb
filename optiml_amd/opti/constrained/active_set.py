from ... import _lib
from ._base import BoxConstrainedQuadraticOptimizer

__all__ = ['ActiveSet']


class ActiveSet(BoxConstrainedQuadraticOptimizer):
    """Primal active-set method for the box QP.

    Interface and semantics of optiml/opti/constrained/active_set.py:33-237: masks L/U/A, the restricted
    system Q_AA x_A = -(q_A + Q_AU ub_U + Q_AL lb_L) solved by Cholesky, jump-or-ratio-step, release of the
    first wrong-sign multiplier (Bland), 1e-12 tolerances.  Mask bookkeeping, gathers, the factorisation and the
    KKT sign scan all run on the device.
    """
    _kind = _lib.AS
    _header = 'iter\t cost\t\t|B|'

    def _after_row(self, row):
        self.n_bound = int(row['r1'])

    def _line(self, row):
        s = '\n{:4d}\t{: 1.4e}\t{:d}\t'.format(int(row['iter']), float(row['f']), int(row['r1']))
        ev, arg = int(row['r2']), row['r3']
        if ev == 1:
            s += '\tI/O: O {:d}(L)'.format(int(arg))
        elif ev == 2:
            s += '\tI/O: O {:d}(U)'.format(int(arg))
        elif ev == 0:
            nl = int(arg) >> 32
            nu = int(arg) & 0xffffffff
            s += '\tI/O: I {:d}+{:d}'.format(nl, nu)
        return s

    def _finalize(self, solver):
        self.L = solver.get(_lib.GET_MASK_L) > 0
        self.U = solver.get(_lib.GET_MASK_U) > 0
