"""Parity at BASELINE.json's full sizes through size-independent properties (the oracle cannot hold an n x n matrix at
n = 100 000): linearity and symmetry of the Hessian product, spot rows against the oracle's kernel, consistency of the
objective/gradient pair, descent and feasibility of the solver iterates, record bookkeeping."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def headline():
    from optiml_amd.datasets import make_blobs
    from optiml_amd.ml.svm.kernels import gaussian
    from optiml_amd.opti import KernelQuadratic
    n, d = 100000, 128                       # BASELINE metric: RBF SVC n=100k d=128
    X, y = make_blobs(n, d, seed=0)
    quad = KernelQuadratic(X, -np.ones(n), 'svc', gaussian, y=y)
    quad.device_problem()
    yield n, X, y, quad
    quad.release()


def test_product_is_linear_and_symmetric(headline):
    n, X, y, quad = headline
    dev = quad.device_problem()
    rs = np.random.RandomState(1)
    u, v = rs.standard_normal(n), rs.standard_normal(n)
    Qu, Qv = dev.matvec(u), dev.matvec(v)
    a, b = 0.37, -1.9
    np.testing.assert_allclose(dev.matvec(a * u + b * v), a * Qu + b * Qv, rtol=1e-10, atol=1e-7)
    assert abs(u @ Qv - v @ Qu) <= 1e-10 * (np.linalg.norm(u) * np.linalg.norm(Qv))
    assert u @ Qu > 0 and v @ Qv > 0          # K*yy' + yy' is positive semidefinite
    assert np.array_equal(dev.matvec(u), Qu)  # bit-reproducible


def test_spot_rows_against_the_oracle_kernel(headline):
    """Rows of Q recomputed on the CPU with the oracle's kernel formula, dotted with v, against the device product."""
    from oracle import svm_oracle as so
    n, X, y, quad = headline
    dev = quad.device_problem()
    rs = np.random.RandomState(2)
    v = rs.standard_normal(n)
    Qv = dev.matvec(v)
    Kw = dev.gram_matvec(v)
    rows = np.array([0, 1, 255, 256, 257, 4095, 50000, 77777, 99743, 99999])
    g = so.resolve_gamma('scale', X)
    xx = np.einsum('ij,ij->i', X, X)
    for i in rows:
        d2 = np.maximum(-2 * (X @ X[i]) + xx[i] + xx, 0)
        d2[i] = 0
        Ki = np.exp(-g * d2)
        Qi = (Ki + 1.0) * y[i] * y
        np.testing.assert_allclose(Qv[i], Qi @ v, rtol=1e-9, atol=1e-9)
        np.testing.assert_allclose(Kw[i], Ki @ v, rtol=1e-9, atol=1e-9)


def test_objective_gradient_consistency(headline):
    n, X, y, quad = headline
    x = np.random.RandomState(3).uniform(0, 1, n)
    f, g = quad.function_jacobian(x)
    Qx = quad.device_problem().matvec(x)
    np.testing.assert_allclose(g, Qx - 1.0, rtol=1e-12, atol=1e-9)
    np.testing.assert_allclose(f, 0.5 * x @ Qx - x.sum(), rtol=1e-11)


@pytest.mark.parametrize('name', ['pg', 'fw'])
def test_solver_iterates_descend_and_stay_in_the_box(headline, name):
    from optiml_amd.opti.constrained import ProjectedGradient, FrankWolfe
    n, X, y, quad = headline
    cls = {'pg': ProjectedGradient, 'fw': FrankWolfe}[name]
    hist = []
    cb = lambda o: hist.append((o.iter, o.f_x))
    cb._bq_needs_state = False
    opt = cls(quad=quad, ub=np.ones(n), max_iter=25, callback=cb).minimize()
    assert opt.status == 'stopped' and opt.iter == 25
    assert [k for k, _ in hist] == list(range(26))
    f = np.array([v for _, v in hist])
    assert np.all(np.diff(f) <= 0)
    assert np.all(opt.x >= -1e-12) and np.all(opt.x <= 1 + 1e-12)   # a ratio step lands on a bound up to rounding
    # the incrementally carried gradient / objective agree with a fresh evaluation at the final point
    f_chk, g_chk = quad.function_jacobian(opt.x)
    np.testing.assert_allclose(opt.f_x, f_chk, rtol=1e-10)
    np.testing.assert_allclose(opt.g_x, g_chk, rtol=1e-8, atol=1e-6)


def test_config2_shape_against_the_oracle():
    """BASELINE config 2 (RBF, ProjectedGradient, n=20 000, d=64): the first 30 iterations against the CPU oracle run
    on the same input (the oracle's dense Q is 3.2 GB: still practical on the host)."""
    from oracle import svm_oracle as so, bcqp_oracle as bo
    from optiml_amd.datasets import make_blobs
    from optiml_amd.ml.svm.kernels import gaussian
    from optiml_amd.opti import KernelQuadratic
    from optiml_amd.opti.constrained import ProjectedGradient
    n, d = 20000, 64
    X, y = make_blobs(n, d, seed=0)
    Q, q, ub = so.svc_dual(so.gram('rbf', X), y, 1.0)
    ref = bo.projected_gradient(Q, q, ub, max_iter=30, keep_x=(30,))
    del Q
    hist = []
    cb = lambda o: hist.append(o.f_x)
    cb._bq_needs_state = False
    quad = KernelQuadratic(X, q, 'svc', gaussian, y=y)
    opt = ProjectedGradient(quad=quad, ub=ub, max_iter=30, callback=cb).minimize()
    np.testing.assert_allclose(hist, ref['f_hist'], rtol=1e-9)
    np.testing.assert_allclose(opt.x, ref['x_at'][30], rtol=1e-6, atol=1e-9)
    quad.release()


def test_active_set_kept_factor_at_config2_shape(monkeypatch):
    """BASELINE config 2's shape (n=20 000, d=64), ActiveSet from the reference's start: 230 iterations with the factor
    of a base set kept (base factorisations every 96 changed indices, or none at all within these iterations — the largest
    factors carry 512) against
    the same iterations with Q[A,A] re-factorised every time, as the reference does — same events, objectives to 1e-11."""
    from optiml_amd.datasets import make_blobs
    from optiml_amd.ml.svm.kernels import gaussian
    from optiml_amd.opti import KernelQuadratic
    from optiml_amd.opti.constrained import ActiveSet
    n, d = 20000, 64
    X, y = make_blobs(n, d, seed=0)
    runs = []
    for mode, limit in (('0', None), ('1', None), ('1', '512')):   # 512: what factors of 80 000 rows and more carry
        monkeypatch.setenv('BQ_AS_SCHUR', mode)
        if limit is None:
            monkeypatch.delenv('BQ_AS_SCHUR_LIMIT', raising=False)
        else:
            monkeypatch.setenv('BQ_AS_SCHUR_LIMIT', limit)
        hist = []
        cb = lambda o: hist.append((o.f_x, o.n_bound))
        cb._bq_needs_state = False
        quad = KernelQuadratic(X, -np.ones(n), 'svc', gaussian, y=y)
        opt = ActiveSet(quad=quad, ub=np.ones(n), max_iter=230, callback=cb).minimize()
        runs.append((np.array(hist), opt.x))
        quad.release()
    (h0, x0) = runs[0]
    for h1, x1 in runs[1:]:
        assert np.array_equal(h0[:, 1], h1[:, 1])
        np.testing.assert_allclose(h1[:, 0], h0[:, 0], rtol=1e-11)
        np.testing.assert_allclose(x1, x0, rtol=1e-9, atol=1e-12)


# ---------------------------------------------------------------------------------------------------------------
# the other dual branches and the streamed mode at the headline size
# ---------------------------------------------------------------------------------------------------------------
def test_streamed_product_equals_the_resident_panel_at_full_size(headline):
    """storage='stream' (Gram tiles recomputed inside the product) against the resident triangular panel."""
    from optiml_amd.ml.svm.kernels import gaussian
    from optiml_amd.opti import KernelQuadratic
    n, X, y, quad = headline
    v = np.random.RandomState(7).standard_normal(n)
    ref = quad.device_problem().matvec(v)
    st = KernelQuadratic(X, -np.ones(n), 'svc', gaussian, y=y, storage='stream')
    try:
        np.testing.assert_allclose(st.device_problem().matvec(v), ref, rtol=1e-10, atol=1e-7)
    finally:
        st.release()


def test_smo_reaches_the_kkt_conditions_at_full_size(headline):
    """SVC.fit(optimizer='smo') at n = 100 000: the optimality conditions of Keerthi et al. recomputed from the returned
    multipliers with ONE independent device product (F = K (alpha*y) - y), the equality constraint, the box."""
    from optiml_amd.ml.svm import SVC
    from optiml_amd.ml.svm.kernels import gaussian
    from optiml_amd.ml.svm.losses import hinge
    n, X, y, quad = headline
    tol = 1e-3
    est = SVC(loss=hinge, kernel=gaussian, C=1., dual=True, optimizer='smo', tol=tol).fit(X, y)
    a = est.alphas_
    assert a.min() >= 0 and a.max() <= 1
    assert abs(a @ y) <= 1e-8 * max(1.0, a.sum())            # y'alpha = 0 is maintained by every pair step
    F = quad.device_problem().gram_matvec(a * y) - y           # resident panel of the fixture: an independent product
    free = (a > 0) & (a < 1)
    up = free | ((y == 1) & (a == 0)) | ((y == -1) & (a == 1))
    low = free | ((y == 1) & (a == 1)) | ((y == -1) & (a == 0))
    b_up, b_low = F[up].min(), F[low].max()
    assert b_low <= b_up + 2 * tol + 1e-9
    assert b_low - tol - 1e-9 <= -est.intercept_ <= b_up + tol + 1e-9
    assert 100 < len(est.support_) < 5000 and est.score(X[:2000], y[:2000]) > 0.99
    est.obj.release()


def test_augmented_lagrangian_records_are_consistent_at_full_size(headline):
    """AdaGrad on the augmented Lagrangian of the reg_intercept=False dual, 12 iterations at n = 100 000: the recorded
    primal values equal an independent evaluation at the recorded point, multipliers stay feasible, the value decreases."""
    from optiml_amd.ml.svm.kernels import gaussian
    from optiml_amd.opti import KernelQuadratic
    from optiml_amd.opti.constrained import AugmentedLagrangianQuadratic
    from optiml_amd.opti.unconstrained.stochastic import AdaGrad
    n, X, y, quad = headline
    primal = KernelQuadratic(X, -np.ones(n), 'svc', gaussian, y=y, rank_one=False)
    al = AugmentedLagrangianQuadratic(primal=primal, A=y, b=np.zeros(1), lb=np.zeros(n), ub=np.ones(n), rho=1.)
    seen = []

    def cb(opt):     # full state: forces one host round trip per iteration
        seen.append((opt.iter, opt.f_x, opt.primal_f_x, opt.x.copy()))
    try:
        opt = AdaGrad(f=al, x=np.random.RandomState(0).uniform(size=n), step_size=1., epochs=12, callback=cb).minimize()
        assert opt.status == 'stopped' and opt.iter == 11 and len(seen) == 12
        for it, f, pf, x in (seen[0], seen[5], seen[11]):
            np.testing.assert_allclose(pf, primal.function(x), rtol=1e-10)
        fs = [s[1] for s in seen]
        assert fs[-1] < fs[0]
        assert np.all(al.dual_x[1:] >= 0) and np.isfinite(al.dual_x).all()
        # the multiplier of the equality row after the last update: rho * sum_k y'x_k over the updates
        assert np.isfinite(opt.g_x).all() and opt.g_x.shape == (n,)
    finally:
        primal.release()
