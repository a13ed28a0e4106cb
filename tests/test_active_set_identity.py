"""The identity behind the device ActiveSet's product-free objective (INTEGRATION.md "Deviations"; csrc/bq_as.hip, as_step_min_kernel),
checked on the CPU against the oracle's trajectory — which, like the reference (active_set.py:172-176), evaluates f with products by
Q after every ratio step:

    d = cand - x lives on the free set A and cand solves the restricted system  =>  Q_AA d_A = -g_A(x)  =>
    f(x + t d) = f(x) + (t - t^2/2) g_A'd_A      and      g_A(x + t d) = (1 - t) g_A(x),

so along a run of ratio steps g_A is the gradient g0 of the run's first point (the starting point, or the point of the last release
iteration) times gamma = prod (1 - t_j).  No GPU, no library: numpy on the oracle's recorded iterates."""
import numpy as np
import pytest

from oracle import bcqp_oracle as bo


def _problems():
    rng = np.random.default_rng(3)
    n = 90
    M = rng.standard_normal((n, n))
    yield 'random spd, lb != 0', M @ M.T + 0.1 * np.eye(n), 3.0 * rng.standard_normal(n), np.full(n, -0.2), np.full(n, 0.5)
    n = 140
    X = rng.standard_normal((n, 5))
    y = np.where(X[:, 0] + 0.5 * rng.standard_normal(n) > 0, 1.0, -1.0)
    K = np.exp(-0.4 * ((X[:, None, :] - X[None, :, :]) ** 2).sum(-1))
    yield 'hinge dual, rbf', (K + 1.0) * np.outer(y, y), -np.ones(n), np.zeros(n), np.ones(n)


@pytest.mark.parametrize('name,Q,q,lb,ub', list(_problems()), ids=lambda v: v if isinstance(v, str) else '')
def test_ratio_step_objective_identity_along_the_oracle_trajectory(name, Q, q, lb, ub):
    probe = bo.active_set(Q, q, ub, lb=lb, max_iter=2000, trace=True)
    assert probe['status'] == 'optimal'
    its = probe['iter']
    ref = bo.active_set(Q, q, ub, lb=lb, max_iter=2000, trace=True, keep_x=tuple(range(its + 1)))
    xs, tr, f_hist = ref['x_at'], ref['trace'], ref['f_hist']
    assert not any(ev['used_minres'] for ev in tr)       # the identity needs exact restricted solves
    g0 = Q @ xs[0] + q
    gamma, f = 1.0, f_hist[0]
    steps = longest = run = 0
    scale = np.abs(f_hist).max()
    for k, ev in enumerate(tr):
        if ev['kind'] == 'step':
            t = ev['max_t']
            if t > 0:
                d = (xs[k + 1] - xs[k]) / t
                f = f + (t - 0.5 * t * t) * gamma * (g0 @ d)
                gamma *= 1.0 - t
            steps += 1
            run += 1
            longest = max(longest, run)
            # the chained value against the oracle's fresh evaluation (a product with Q) of the same point
            assert abs(f - f_hist[k + 1]) <= 1e-11 * scale, (name, k, f, f_hist[k + 1])
        elif ev['kind'] == 'release':
            f = f_hist[k + 1]                             # a release iteration forms f and g by a product (the reference needs g there too)
            g0 = Q @ xs[k + 1] + q
            gamma, run = 1.0, 0
    assert steps >= its // 4 and longest >= 3, (name, steps, longest, its)
