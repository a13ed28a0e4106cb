"""GPU parity tests of the augmented-Lagrangian dual path: libbcqp_hip.so's bq_al_* through the
Python classes, against the golden fixtures generated from the reference (tools/gen_golden.py gen_lagrangian) and the
CPU oracle (oracle/al_oracle.py).

Tolerances (fp64): value / primal-value histories rtol 1e-9 while the iterates agree to rtol 1e-6 (atol 1e-9), as for
the box-constrained solvers.
"""
import warnings

import numpy as np
import pytest

from conftest import load_golden
from test_oracle_golden import AL_RULES, AL_NOB, _al_rules_problem

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def amd():
    import optiml_amd
    from optiml_amd import _lib
    from optiml_amd.device import get_context
    _lib.load()
    get_context()
    return optiml_amd


def _classes():
    from optiml_amd.opti.unconstrained import stochastic as st
    return {'sgd': st.StochasticGradientDescent, 'adam': st.Adam, 'amsgrad': st.AMSGrad, 'adamax': st.AdaMax,
            'adagrad': st.AdaGrad, 'adadelta': st.AdaDelta, 'rmsprop': st.RMSProp}


class Rec:
    """callback replayed from the device records: needs no per-iteration state"""
    _bq_needs_state = False

    def __init__(self):
        self.f, self.pf, self.gap = [], [], []

    def __call__(self, opt):
        self.f.append(opt.f_x)
        self.pf.append(opt.primal_f_x)
        self.gap.append(opt.dgap)


def _cmp(opt, al, rec, g, p, rtol=1e-6, atol=1e-9):
    assert opt.status == str(g[p + '_status'])
    assert opt.iter == int(g[p + '_iter'])
    assert opt.epoch == int(g[p + '_epoch'])
    np.testing.assert_allclose(rec.f, g[p + '_f_hist'], rtol=1e-9, atol=1e-10)
    np.testing.assert_allclose(rec.pf, g[p + '_pf_hist'], rtol=1e-9, atol=1e-10)
    np.testing.assert_allclose(opt.f_x, float(g[p + '_f_x']), rtol=1e-9)
    np.testing.assert_allclose(opt.x, g[p + '_x'], rtol=rtol, atol=atol)
    np.testing.assert_allclose(opt.g_x, g[p + '_g_x'], rtol=rtol, atol=1e-8)
    np.testing.assert_allclose(al.dual_x, g[p + '_dual_x'], rtol=rtol, atol=atol)


def _run(amd, name, Q, q, a, lb, ub, rho, x0, epochs, tol, primal=None):
    from optiml_amd.opti import Quadratic
    from optiml_amd.opti.constrained import AugmentedLagrangianQuadratic
    rule, kw = AL_RULES[name]
    al = AugmentedLagrangianQuadratic(primal=primal if primal is not None else Quadratic(Q, q), A=a,
                                      b=None if a is None else np.zeros(1), lb=lb, ub=ub, rho=rho)
    rec = Rec()
    opt = _classes()[rule](f=al, x=x0.copy(), epochs=epochs, tol=tol, callback=rec, **kw).minimize()
    return opt, al, rec


@pytest.mark.parametrize('name', AL_NOB)
def test_rules_equality_constrained(amd, name):
    g = load_golden('al_dual.npz')
    Q, q, a, lb, ub, rho = _al_rules_problem(g, False)
    opt, al, rec = _run(amd, name, Q, q, a, lb, ub, rho, g['rules_x0'], 300, 1e-10)
    _cmp(opt, al, rec, g, 'rules_' + name)


@pytest.mark.parametrize('name', sorted(AL_RULES))
def test_rules_box_only(amd, name):
    g = load_golden('al_dual.npz')
    Q, q, a, lb, ub, rho = _al_rules_problem(g, True)
    opt, al, rec = _run(amd, name, Q, q, a, lb, ub, rho, g['rules_x0'], 300, 1e-10)
    _cmp(opt, al, rec, g, 'rulesb_' + name)


@pytest.mark.parametrize('name', ['adagrad', 'rmsprop_nesterov'])
@pytest.mark.parametrize('intercept', [False, True])
def test_rules_on_the_kernel_built_panel(amd, name, intercept):
    """Same fixtures through the lazy KernelQuadratic: Gram panel built on the device, BQ_NO_RANK_ONE when the
    intercept is not regularised (Q = K*yy', svm/_base.py:552-555)."""
    from optiml_amd.opti import KernelQuadratic
    from optiml_amd.ml.svm.kernels import gaussian
    g = load_golden('al_dual.npz')
    Q, q, a, lb, ub, rho = _al_rules_problem(g, intercept)
    primal = KernelQuadratic(g['rules_X'], q, 'svc', gaussian, y=g['rules_y'], rank_one=intercept)
    np.testing.assert_allclose(primal.Q, Q, rtol=1e-12, atol=1e-14)
    opt, al, rec = _run(amd, name, None, None, a, lb, ub, rho, g['rules_x0'], 300, 1e-10, primal=primal)
    _cmp(opt, al, rec, g, ('rulesb_' if intercept else 'rules_') + name)


def test_tolerance_stop_is_optimal(amd):
    # optiml/opti/_base.py:141-146
    g = load_golden('al_dual.npz')
    Q, q, a, lb, ub, rho = _al_rules_problem(g, False)
    opt, al, rec = _run(amd, 'adagrad', Q, q, a, lb, ub, rho, g['rules_x0'], 20000, 2e-3)
    assert opt.status == 'optimal'
    _cmp(opt, al, rec, g, 'tol_adagrad')


def test_reference_unit_problem_step_mode(amd, capsys):
    """optiml/opti/constrained/tests/test_lagrangian_quadratic.py:18-22 (ndim = 2: the x0/x1 histories force one host
    round trip per iteration) + the verbose line formats of stochastic/_base.py:142-158 and opti/_base.py:101-103."""
    from optiml_amd.opti import Quadratic
    from optiml_amd.opti.constrained import AugmentedLagrangianQuadratic
    from optiml_amd.opti.unconstrained.stochastic import AdaGrad
    g = load_golden('al_dual.npz')
    al = AugmentedLagrangianQuadratic(primal=Quadratic(g['nd2_Q'], g['nd2_q']), A=g['nd2_a'], b=np.zeros(1),
                                      lb=np.zeros(2), ub=g['nd2_ub'], rho=1)
    xs = {}

    def cb(opt):
        xs[opt.iter] = opt.x.copy()
    opt = AdaGrad(al, x=g['nd2_x0'].copy(), step_size=1, epochs=15000, callback=cb, verbose=50).minimize()
    p = 'nd2_adagrad'
    assert opt.status == str(g[p + '_status']) == 'optimal' and opt.iter == int(g[p + '_iter'])
    np.testing.assert_allclose(opt.x, g[p + '_x'], rtol=1e-6, atol=1e-9)
    np.testing.assert_allclose(opt.f_x_history, g[p + '_pf_hist'], rtol=1e-9, atol=1e-10)   # primal values
    np.testing.assert_allclose(al.dual_x, g[p + '_dual_x'], rtol=1e-6, atol=1e-9)
    for k, xk in zip(g[p + '_x_iters'], g[p + '_x_at']):
        np.testing.assert_allclose(xs[int(k)], xk, rtol=1e-6, atol=1e-9)
        np.testing.assert_allclose([opt.x0_history[int(k)], opt.x1_history[int(k)]], xk, rtol=1e-6, atol=1e-9)
    out = capsys.readouterr().out
    assert out.startswith('epoch\titer\t cost\t')
    line = '\n{:4d}\t{:4d}\t{: 1.4e}'.format(50, 50, g[p + '_f_hist'][50])
    assert line + '\tpcost: {: 1.4e}'.format(g[p + '_pf_hist'][50]) in out


def test_plain_quadratic_runs_the_rule_for_all_epochs(amd):
    """No Lagrangian: the loop has no stopping test but the epoch count (adagrad.py:81-123 with a plain Quadratic)."""
    from oracle import al_oracle as ao
    from optiml_amd.opti import Quadratic
    from optiml_amd.opti.unconstrained.stochastic import Adam
    rs = np.random.RandomState(5)
    B = rs.standard_normal((40, 40))
    Q, q, x0 = B @ B.T / 40 + np.eye(40), rs.standard_normal(40), rs.uniform(size=40)
    ref = ao.minimize(ao.AugLag(Q, q), x0, 'adam', epochs=120, step_size=0.05, momentum_type='polyak', momentum=0.3)
    hist = []
    cb = lambda o: hist.append(o.f_x)   # noqa: E731
    cb._bq_needs_state = False
    opt = Adam(Quadratic(Q, q), x=x0.copy(), epochs=120, step_size=0.05, momentum_type='polyak', momentum=0.3,
               callback=cb).minimize()
    assert opt.status == 'stopped' and opt.iter == ref['iter'] == 119 and not opt.is_lagrangian_dual()
    np.testing.assert_allclose(hist, ref['f_hist'], rtol=1e-9)
    np.testing.assert_allclose(opt.x, ref['x'], rtol=1e-6, atol=1e-9)


def test_multipliers_warm_start_from_the_objective(amd):
    """dual_x lives in the AugmentedLagrangianQuadratic and survives across optimizers (constrained/_base.py:297-299):
    two runs of 150 epochs = one run of 300 for a rule without per-coordinate state (plain gradient steps)."""
    g = load_golden('al_dual.npz')
    Q, q, a, lb, ub, rho = _al_rules_problem(g, False)
    from optiml_amd.opti import Quadratic
    from optiml_amd.opti.constrained import AugmentedLagrangianQuadratic
    from optiml_amd.opti.unconstrained.stochastic import StochasticGradientDescent as SGD
    al = AugmentedLagrangianQuadratic(primal=Quadratic(Q, q), A=a, b=np.zeros(1), lb=lb, ub=ub, rho=rho)
    o1 = SGD(f=al, x=g['rules_x0'].copy(), epochs=151, tol=1e-10, step_size=0.004).minimize()
    assert o1.iter == 150
    # the first run stops at its 151st evaluation without moving: continue from there
    o2 = SGD(f=al, x=o1.x.copy(), epochs=150, tol=1e-10, step_size=0.004).minimize()
    np.testing.assert_allclose(o2.x, g['rules_sgd_x'], rtol=1e-9, atol=1e-12)
    np.testing.assert_allclose(al.dual_x, g['rules_sgd_dual_x'], rtol=1e-9, atol=1e-12)


def _check_fit(est, g, p, Xte):
    assert est.optimizer.status == str(g[p + '_status']) and est.optimizer.iter == int(g[p + '_iter'])
    np.testing.assert_allclose(est.train_loss_history, g[p + '_loss_hist'], rtol=1e-9, atol=1e-10)
    np.testing.assert_allclose(est.alphas_, g[p + '_alphas'], rtol=1e-6, atol=1e-9)
    np.testing.assert_allclose(est.obj.dual_x, g[p + '_dual_x'], rtol=1e-6, atol=1e-9)
    ref_sup = g[p + '_support']
    if not np.array_equal(est.support_, ref_sup):  # only entries sitting on the 1e-6 threshold may differ
        diff = np.setxor1d(est.support_, ref_sup)
        a = g[p + '_alphas']
        a = a if len(a) == len(g['y']) else np.maximum(a[:len(a) // 2], a[len(a) // 2:])
        assert np.all(np.abs(a[diff] - 1e-6) < 1e-8)
    else:
        np.testing.assert_allclose(est.dual_coef_, g[p + '_dual_coef'], rtol=1e-6, atol=1e-9)
        np.testing.assert_allclose(est.intercept_, float(g[p + '_intercept']), rtol=1e-6, atol=1e-9)
        np.testing.assert_allclose(est.decision_function(Xte), g[p + '_decision'], rtol=1e-6, atol=1e-8)


@pytest.mark.parametrize('tag', ['adagrad_nob', 'rmsprop_nob', 'adagrad_b', 'adam_b'])
def test_fit_svc(amd, tag):
    """optiml/ml/tests/test_svc.py:134-147 shape: SVC(dual=True, optimizer=AdaGrad, learning_rate=1.), both intercepts"""
    from optiml_amd.ml.svm import SVC
    from optiml_amd.ml.svm.kernels import gaussian
    from optiml_amd.ml.svm.losses import hinge
    from optiml_amd.opti.unconstrained.stochastic import AdaGrad, Adam, RMSProp
    g = load_golden('fit_al_svc_n200.npz')
    name, reg = tag.split('_')
    cls, kw = {'adagrad': (AdaGrad, dict(learning_rate=1.)),
               'adam': (Adam, dict(learning_rate=0.002, momentum_type='nesterov', momentum=0.5)),
               'rmsprop': (RMSProp, dict(learning_rate=0.01))}[name]
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter('always')
        est = SVC(loss=hinge, kernel=gaussian, C=1., reg_intercept=(reg == 'b'), dual=True, optimizer=cls,
                  max_iter=1000, random_state=1, **kw).fit(g['X'], g['y'])
    assert any('max_iter reached' in str(x.message) for x in w)   # ConvergenceWarning, svm/_base.py:719-721
    _check_fit(est, g, tag, g['Xtest'])


@pytest.mark.parametrize('tag', ['adagrad_nob', 'adagrad_b'])
def test_fit_svr(amd, tag):
    from optiml_amd.ml.svm import SVR
    from optiml_amd.ml.svm.kernels import gaussian
    from optiml_amd.ml.svm.losses import epsilon_insensitive
    from optiml_amd.opti.unconstrained.stochastic import AdaGrad
    g = load_golden('fit_al_svr_n150.npz')
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        est = SVR(loss=epsilon_insensitive, epsilon=0.1, kernel=gaussian, C=1., reg_intercept=tag.endswith('_b'),
                  dual=True, optimizer=AdaGrad, learning_rate=1., max_iter=1000, random_state=1).fit(g['X'], g['y'])
    _check_fit(est, g, tag, g['Xtest'])


@pytest.mark.parametrize('reg', [False, True])
def test_squared_hinge_dual_against_oracle(amd, reg):
    """svm/_base.py:727-730, :778-794: Q += I/(2C), lower bound only.  No reference fixture: checked against the
    (pinned) oracle run on the same assembly."""
    from oracle import al_oracle as ao, svm_oracle as so
    from optiml_amd.datasets import make_blobs
    from optiml_amd.ml.svm import SVC
    from optiml_amd.ml.svm.kernels import gaussian
    from optiml_amd.ml.svm.losses import squared_hinge
    from optiml_amd.opti.unconstrained.stochastic import AdaGrad
    X, y = make_blobs(160, 5, seed=9, sigma=6.0)
    C = 2.0
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        est = SVC(loss=squared_hinge, kernel=gaussian, C=C, reg_intercept=reg, dual=True, optimizer=AdaGrad,
                  learning_rate=1., max_iter=400, random_state=4, rho=1.5).fit(X, y)
    Q = so.gram('rbf', X) * np.outer(y, y) + (np.outer(y, y) if reg else 0.) + np.eye(len(y)) / (2 * C)
    al = ao.AugLag(Q, -np.ones(len(y)), a=None if reg else y, lb=np.zeros(len(y)), ub=None, rho=1.5)
    ref = ao.minimize(al, np.random.RandomState(4).uniform(size=len(y)), 'adagrad', epochs=400, tol=1e-4, step_size=1.)
    assert est.optimizer.iter == ref['iter'] and est.optimizer.status == ref['status']
    np.testing.assert_allclose(est.train_loss_history, ref['pf_hist'], rtol=1e-9, atol=1e-10)
    np.testing.assert_allclose(est.alphas_, ref['x'], rtol=1e-6, atol=1e-9)
    np.testing.assert_allclose(est.obj.dual_x, ref['dual_x'], rtol=1e-6, atol=1e-9)


def test_box_solvers_refuse_the_dual_without_rank_one(amd):
    from optiml_amd import _lib
    from optiml_amd.opti import KernelQuadratic
    from optiml_amd.opti.constrained import InteriorPoint, ProjectedGradient
    from optiml_amd.ml.svm.kernels import gaussian
    from optiml_amd.datasets import make_blobs
    X, y = make_blobs(64, 4, seed=1)
    quad = KernelQuadratic(X, -np.ones(64), 'svc', gaussian, y=y, rank_one=False)
    with pytest.raises(_lib.BcqpError):
        InteriorPoint(quad=quad, ub=np.ones(64)).minimize()
    opt = ProjectedGradient(quad=quad, ub=np.ones(64), max_iter=5).minimize()   # products only: fine
    assert opt.iter == 5


@pytest.mark.parametrize('name', ['sched_sgd_polyak', 'sched_rmsprop_nesterov'])
def test_schedules(amd, name):
    """iterable step_size / momentum: the optimizer draws one value per iteration (stochastic/_base.py:88-93, :242-245)
    and hands the table to the device loop"""
    from optiml_amd.opti import Quadratic
    from optiml_amd.opti.constrained import AugmentedLagrangianQuadratic
    from optiml_amd.opti.unconstrained.stochastic import schedules as sch
    g = load_golden('al_dual.npz')
    Q, q, a, lb, ub, rho = _al_rules_problem(g, True)
    al = AugmentedLagrangianQuadratic(primal=Quadratic(Q, q), lb=lb, ub=ub, rho=rho)
    rec = Rec()
    if name == 'sched_sgd_polyak':
        opt = _classes()['sgd'](f=al, x=g['rules_x0'].copy(), epochs=300, tol=1e-10, callback=rec,
                                step_size=sch.decaying(0.002, 0.997), momentum_type='polyak',
                                momentum=sch.sutskever_blend(0.9, 40)).minimize()
    else:
        opt = _classes()['rmsprop'](f=al, x=g['rules_x0'].copy(), epochs=300, tol=1e-10, callback=rec,
                                    step_size=sch.linear_annealing(0.02, 0.002, 200), momentum_type='nesterov',
                                    momentum=sch.repeater([0.2, 0.4, 0.6], 100)).minimize()
    _cmp(opt, al, rec, g, name)
    with pytest.raises(ValueError):    # an iterable that runs dry before `epochs` values
        _classes()['sgd'](f=al, epochs=10, step_size=iter([0.1, 0.1]))
