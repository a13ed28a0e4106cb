"""Checkpoint / resume of the box-constrained solvers (SURVEY 5 "checkpoint / resume"; bq_solver_get_state / _set_state).

The reference can only be restarted from `x=` (optiml/opti/constrained/_base.py:61-65): InteriorPoint's multipliers lp / lm
(interior_point.py:181-186) and ActiveSet's masks L / U (active_set.py:91-92) are locals of `minimize()` and are lost.  Here a
stopped run hands out the state its loop holds at the top of the next iteration, and a NEW optimizer continues from it:
InteriorPoint / ProjectedGradient / FrankWolfe bit-for-bit (every later quantity is a function of that state), ActiveSet along the
reference's golden trajectory (its factor is rebuilt, the stopped run's was updated incrementally).
"""
import pickle

import numpy as np
import pytest

from conftest import load_golden
from test_gpu_parity import AS_KINDS, _solvers, amd, as_factor_mode  # noqa: F401  (fixtures)

pytestmark = pytest.mark.gpu


def _run(kind, g, stop_at=None, by_callback=False, state=None, **kw):
    """One minimize() on the n = 256 RBF SVC dual of the golden trajectory; returns (opt, {iter: f}, {iter: x})."""
    from optiml_amd.opti import Quadratic
    fs, xs = {}, {}

    def cb(o):
        fs[o.iter] = o.f_x
        xs[o.iter] = o.x.copy()
        if by_callback and stop_at is not None and o.iter == stop_at:
            raise StopIteration

    if stop_at is not None and not by_callback:
        kw['max_iter'] = stop_at
    opt = _solvers()[kind](quad=Quadratic(g['Q'], g['q']), ub=g['ub'], callback=cb, **kw)
    if state is not None:
        opt.set_state(state)
    return opt.minimize(), fs, xs


@pytest.mark.parametrize('by_callback', [False, True], ids=['max_iter', 'StopIteration'])
def test_interior_point_resumes_bit_for_bit_on_the_golden_trajectory(amd, by_callback):
    """stop InteriorPoint at iteration 10, restore into a fresh solver, iterations 11... of the reference's
    trajectory follow bit-for-bit (against the uninterrupted device run) and to the golden's tolerance (against the reference)."""
    g = load_golden('traj_svc_rbf_n256.npz')
    full, f_full, x_full = _run('ip', g)
    assert full.status == 'optimal' and full.iter == int(g['ip_iter'])
    first, f1, _ = _run('ip', g, stop_at=10, by_callback=by_callback)
    st = pickle.loads(pickle.dumps(first.get_state()))          # a checkpoint is plain data
    # stopped by max_iter: the loop ended at the top of iteration 10; by the callback (raised at the top of iteration 10): the
    # device had already decided that iteration's step, the state is the top of iteration 11
    assert st['iter'] == (11 if by_callback else 10) and st['kind'] == 3
    assert set(st) >= {'x', 'g', 'lp', 'lm'} and np.all(st['lp'] > 0) and np.all(st['lm'] > 0)
    assert np.array_equal(st['x'], x_full[st['iter']])            # host-applied step == device-applied step, to the bit
    second, f2, x2 = _run('ip', g, state=st)
    assert second.status == 'optimal' and second.iter == full.iter and min(f2) == st['iter']
    for k in f2:
        assert f2[k] == f_full[k] and np.array_equal(x2[k], x_full[k]), k
    assert np.array_equal(second.x, full.x) and second.f_x == full.f_x
    assert np.array_equal(second.lp, full.lp) and np.array_equal(second.lm, full.lm)
    assert np.array_equal(second.g_x, full.g_x)                   # the reference's self.g_x: the START gradient, kept through the state
    hist = [f1[k] for k in sorted(f1) if k < st['iter']] + [f2[k] for k in sorted(f2)]
    np.testing.assert_allclose(hist, g['ip_f_hist'], rtol=1e-9, atol=1e-11)
    np.testing.assert_allclose(second.x, g['ip_x'], rtol=1e-6, atol=1e-9)
    for k, xk in zip(g['ip_x_iters'], g['ip_x_at']):
        if int(k) in x2:
            np.testing.assert_allclose(x2[int(k)], xk, rtol=1e-6, atol=1e-9)


@pytest.mark.parametrize('kind,kw', [('pg', {}), ('fw', {}), ('fw', {'t': 0.1})], ids=['pg', 'fw', 'fw_t0.1'])
@pytest.mark.parametrize('by_callback', [False, True], ids=['max_iter', 'StopIteration'])
def test_projected_gradient_and_frank_wolfe_resume_bit_for_bit(amd, kind, kw, by_callback):
    """(x, g) is the whole state of these two (+ FrankWolfe's best lower bound): the resumed run repeats the uninterrupted one
    to the bit — including PG, whose iteration is chaotic under any rounding difference."""
    g = load_golden('traj_svc_rbf_n256.npz')
    full, f_full, x_full = _run(kind, g, max_iter=120, **kw)
    first, _, _ = _run(kind, g, stop_at=37, by_callback=by_callback, **kw)
    st = first.get_state()
    assert st['iter'] == (38 if by_callback else 37) and np.array_equal(st['x'], x_full[st['iter']])
    if kind == 'fw':
        assert np.isfinite(st['best_lb'])
    second, f2, x2 = _run(kind, g, state=st, max_iter=120, **kw)
    assert second.iter == full.iter == 120 and second.status == full.status == 'stopped'
    assert sorted(f2) == list(range(st['iter'], 121))
    for k in f2:
        assert f2[k] == f_full[k] and np.array_equal(x2[k], x_full[k]), k
    assert np.array_equal(second.x, full.x) and np.array_equal(second.g_x, full.g_x)
    # x alone (what the reference can be restarted from): g = Qx + q is formed afresh, the iterates agree to rounding, not to the bit
    third, f3, _ = _run(kind, g, state={'x': st['x'], 'iter': st['iter'], 'best_lb': st['best_lb']}, max_iter=60, **kw)
    np.testing.assert_allclose([f3[k] for k in sorted(f3)], [f_full[k] for k in sorted(f3)], rtol=1e-9)


@pytest.mark.parametrize('kind', AS_KINDS)
def test_active_set_resumes_on_the_golden_trajectory(amd, kind, as_factor_mode):
    """ActiveSet stopped at iteration 60 of the reference's n = 256 trajectory: masks, point and counter go into a fresh solver,
    which finishes the trajectory (same releases, same iteration count, x and f-history within the golden tolerances)."""
    g = load_golden('traj_svc_rbf_n256.npz')
    first, f1, _ = _run(kind, g, stop_at=60)
    st = pickle.loads(pickle.dumps(first.get_state()))
    assert first.status == 'stopped' and st['iter'] == 60 and st['mask_l'].dtype == bool
    assert np.array_equal(st['mask_l'], first.L) and np.array_equal(st['mask_u'], first.U) and not np.any(st['mask_l'] & st['mask_u'])
    assert st['mask_l'].sum() + st['mask_u'].sum() > 0
    second, f2, x2 = _run(kind, g, state=st, max_iter=5000)
    assert second.status == str(g['as_status']) and second.iter == int(g['as_iter'])
    hist = [f1[k] for k in sorted(f1) if k < 60] + [f2[k] for k in sorted(f2)]
    np.testing.assert_allclose(hist, g['as_f_hist'], rtol=1e-9, atol=1e-11)
    np.testing.assert_allclose(second.x, g['as_x'], rtol=1e-6, atol=1e-9)
    np.testing.assert_allclose(second.f_x, float(g['as_f_x']), rtol=1e-9, atol=1e-12)
    for k, xk in zip(g['as_x_iters'], g['as_x_at']):
        if int(k) in x2:
            np.testing.assert_allclose(x2[int(k)], xk, rtol=1e-6, atol=1e-9)


def test_state_argument_checks(amd):
    from optiml_amd import _lib
    from optiml_amd.opti import Quadratic
    from optiml_amd.opti.constrained._base import _DeviceSolver
    g = load_golden('traj_svc_rbf_n256.npz')
    quad = Quadratic(g['Q'], g['q'])
    n = len(g['q'])
    ip = _solvers()['ip'](quad=quad, ub=g['ub'])
    with pytest.raises(RuntimeError):
        ip.get_state()                                             # nothing has run
    with pytest.raises(ValueError):
        ip.set_state({'iter': 3})                                  # no x
    with pytest.raises(ValueError):
        ip.set_state({'x': np.ones(n) / 2, 'kind': _lib.PG})       # another solver's state
    dev = quad.device_problem()
    s = _DeviceSolver(dev, _lib.AS, np.zeros(n), g['ub'], g['ub'] / 2, 1e-6, 100)
    both = np.zeros(n, bool)
    both[3] = True
    with pytest.raises(_lib.BcqpError, match='both masks'):
        s.set_state({'x': g['ub'] / 2, 'mask_l': both, 'mask_u': both})
    with pytest.raises(_lib.BcqpError, match='InteriorPoint only'):
        s.set_state({'x': g['ub'] / 2, 'lp': np.ones(n), 'lm': np.ones(n)})
    st0 = s.get_state()                                            # before the first run: the start point, nothing else
    assert st0['iter'] == 0 and np.array_equal(st0['x'], g['ub'] / 2) and 'g' not in st0 and 'mask_l' not in st0
    s.run(2)
    with pytest.raises(_lib.BcqpError, match='before the solver'):
        s.set_state({'x': g['ub'] / 2})
    s.close()
    quad.release()
