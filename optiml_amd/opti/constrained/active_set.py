from ... import _lib
from ._base import BoxConstrainedQuadraticOptimizer

__all__ = ['ActiveSet', 'ActiveSetCG']


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
        # iterations that took the reference's minres branch (active_set.py:142-151: Q[A,A] not factorisable)
        self.minres_iterations = solver.counter(_lib.COUNT_MINRES)
        # ratio-step iterations whose f(x) came from the line-search identity instead of a product with Q
        self.product_free_iterations = solver.counter(_lib.COUNT_NO_PRODUCT)


class ActiveSetCG(ActiveSet):
    """ActiveSet whose restricted systems are solved by conjugate gradients on the masked panel product (`BQ_AS_CG`).

    Not a reference class: the reference re-factorises Q_AA by Cholesky in every iteration (active_set.py:141), which
    needs an |A| x |A| dense copy and a single device.  This variant keeps the outer logic unchanged and replaces only
    the linear solve, so ActiveSet also runs where the dense factor cannot: sharded over several GPUs, on fp32-stored
    and on streamed panels (BASELINE config 5).  `inner_tol` is the relative residual the inner iteration stops at,
    `inner_max_iter` its cap (0: 2 |A| + 50); `inner_iters` reports the total after `minimize()`.  Pass the class (or
    a subclass with other settings) as `optimizer=` to SVC / SVR like any other.
    """
    _kind = _lib.AS_CG
    inner_tol = 1e-13
    inner_max_iter = 0

    def _configure(self, solver):
        if not 0 < self.inner_tol < 1:
            raise ValueError('inner_tol has to lie in (0, 1)')
        if self.inner_max_iter < 0:
            raise ValueError('inner_max_iter must be >= 0')
        solver.set_inner(self.inner_tol, self.inner_max_iter)

    def _finalize(self, solver):
        super()._finalize(solver)
        self.inner_iters = solver.inner_iters()
