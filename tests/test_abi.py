"""CPU-side checks of the drop-in boundary: the shared library builds, loads, and exports every symbol that
include/bcqp.h declares; pure-host entry points behave; the Python surface mirrors the reference's names."""
import os
import re

import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope='module')
def lib():
    from optiml_amd import build, _lib
    build.build()           # hipcc cross-compiles without a GPU
    return _lib.load()


def _declared_symbols():
    text = open(os.path.join(REPO, 'include', 'bcqp.h')).read()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    return sorted(set(re.findall(r'\b(bq_[a-z0-9_]+)\s*\(', text)) - {'bq_exchange_fn'})


def test_every_declared_symbol_is_exported_and_bound(lib):
    from optiml_amd import _lib
    syms = _declared_symbols()
    assert len(syms) >= 25
    for s in syms:
        assert hasattr(lib, s), f'{s} declared in include/bcqp.h but not exported'
        assert s in _lib.PROTOTYPES, f'{s} has no ctypes prototype'
    assert set(_lib.PROTOTYPES) == set(syms)
    assert lib.bq_abi_version() == _lib.ABI_VERSION


@pytest.mark.parametrize('n', [2, 127, 128, 129, 2000, 100000, 250000])
@pytest.mark.parametrize('world', [1, 2, 3, 4, 8])
def test_row_blocks_partition_the_panel(lib, n, world):
    from optiml_amd.device import row_block
    from optiml_amd.dist import block_size, SocketComm
    blocks = [row_block(n, r, world) for r in range(world)]
    assert blocks[0][0] == 0 and blocks[-1][1] == n
    for (b0, e0), (b1, e1) in zip(blocks, blocks[1:]):
        assert e0 == b1 and b0 <= e0
    blk = block_size(n, world)
    assert blk % 128 == 0 and blk * world >= n
    for r, (b, e) in enumerate(blocks):
        assert b == min(n, r * blk) and e == min(n, b + blk)
        c = SocketComm.__new__(SocketComm)
        c.rank, c.world_size = r, world
        assert c.rows_of(n) == (b, e)


@pytest.mark.parametrize('n', [2, 255, 256, 257, 700, 100000, 250000])
@pytest.mark.parametrize('world', [1, 2, 4, 8])
def test_symmetric_row_blocks_balance_the_triangle(lib, n, world):
    from optiml_amd.device import row_block
    blocks = [row_block(n, r, world, symmetric=True) for r in range(world)]
    assert blocks[0][0] == 0 and blocks[-1][1] == n
    for (b0, e0), (b1, e1) in zip(blocks, blocks[1:]):
        assert e0 == b1 and b0 <= e0 and (b0 % 256 == 0 or b0 == n)
    if n >= 100000:   # every rank streams about the same number of lower-triangle tiles
        nb = -(-n // 256)
        tiles = []
        for b, e in blocks:
            i0, i1 = b // 256, -(-e // 256)
            tiles.append(i1 * (i1 + 1) // 2 - i0 * (i0 + 1) // 2)
        assert sum(tiles) == nb * (nb + 1) // 2
        assert max(tiles) <= 1.06 * (sum(tiles) / world)


@pytest.mark.parametrize('n', [700, 20000, 100000, 250000])
def test_symmetric_partitions_nest_on_the_eight_canonical_segments(lib, n):
    """1, 2, 4 and 8 ranks own whole runs of the same 8 segments (the unit of the fixed-order sum that makes the product
    bit-identical across rank counts); 3 ranks own 2 + 3 + 3 of them."""
    from optiml_amd.device import row_block
    cuts8 = [row_block(n, r, 8, symmetric=True)[0] for r in range(8)] + [n]
    for world in (1, 2, 4):
        step = 8 // world
        for r in range(world):
            assert row_block(n, r, world, symmetric=True) == (cuts8[r * step], cuts8[(r + 1) * step])
    firsts = [0, 2, 5, 8]
    for r in range(3):
        assert row_block(n, r, 3, symmetric=True) == (cuts8[firsts[r]], cuts8[firsts[r + 1]])


def test_bad_arguments_are_reported_not_crashed(lib):
    import ctypes as C
    from optiml_amd import _lib
    assert lib.bq_row_block(10, 3, 2, None, None) == _lib.ERR_BADARG
    assert b'rank' in lib.bq_last_error()
    with pytest.raises(_lib.BcqpError):
        _lib.check(lib.bq_problem_matvec(None, None, None))


def test_more_ranks_than_the_segment_table_holds_are_refused(lib):
    """bq_seg_table (passed to kernels by value) holds 64 canonical segments = 64 ranks: a larger world is a bad argument in
    every multi-rank constructor, before any device is touched (it used to write past the struct)."""
    import ctypes as C
    from optiml_amd import _lib
    h = C.c_void_p()
    uid = C.create_string_buffer(128)
    cb = _lib.EXCHANGE_FN(lambda *a: 0)
    for world in (65, 72, 1000):
        assert lib.bq_ctx_create_rccl(0, 0, world, uid, C.byref(h)) == _lib.ERR_BADARG
        assert b'64 ranks' in lib.bq_last_error()
        assert lib.bq_ctx_create_exchange(0, 0, world, cb, None, C.byref(h)) == _lib.ERR_BADARG
        assert lib.bq_ctx_create_share(0, 0, world, C.byref(h)) == _lib.ERR_BADARG
    assert lib.bq_ctx_create_share(0, 3, 2, C.byref(h)) == _lib.ERR_BADARG and b'rank' in lib.bq_last_error()
    assert not h


def test_python_surface_matches_reference_names():
    import optiml_amd.opti as opti
    import optiml_amd.opti.constrained as con
    import optiml_amd.ml.svm as svm
    from optiml_amd.ml.svm import kernels, losses
    for name in ('Optimizer', 'OptimizationFunction', 'Quadratic'):
        assert hasattr(opti, name)
    for name in ('BoxConstrainedQuadraticOptimizer', 'ProjectedGradient', 'ActiveSet', 'FrankWolfe', 'InteriorPoint'):
        assert issubclass(getattr(con, name), con.BoxConstrainedQuadraticOptimizer)
    for name in ('SVM', 'SVC', 'SVR'):
        assert hasattr(svm, name)
    for name in ('linear', 'poly', 'gaussian', 'laplacian', 'sigmoid', 'LinearKernel', 'PolyKernel', 'GaussianKernel',
                 'LaplacianKernel', 'SigmoidKernel'):
        assert hasattr(kernels, name)
    for name in ('hinge', 'squared_hinge', 'epsilon_insensitive', 'squared_epsilon_insensitive'):
        assert hasattr(losses, name)


def test_constructor_validation_mirrors_reference():
    from optiml_amd.opti import Quadratic
    from optiml_amd.opti.constrained import ProjectedGradient, FrankWolfe, InteriorPoint
    from optiml_amd.ml.svm import SVC, SVR
    from optiml_amd.ml.svm.kernels import PolyKernel, GaussianKernel, gaussian
    from optiml_amd.ml.svm.losses import hinge, squared_hinge, epsilon_insensitive
    Q = np.eye(3)
    quad = Quadratic(Q, np.zeros(3))
    assert quad.ndim == 3 and quad.hessian(None) is quad.Q
    with pytest.raises(ValueError):
        Quadratic(np.eye(1), np.zeros(1))                       # opti/_base.py:249-250
    with pytest.raises(ValueError):
        Quadratic(np.eye(3), np.zeros(2))                       # opti/_base.py:255-256
    with pytest.raises(TypeError):
        ProjectedGradient(quad=object(), ub=np.ones(3))         # constrained/_base.py:59-60
    with pytest.raises(ValueError):
        ProjectedGradient(quad=quad, ub=np.ones(3), max_iter=0)  # opti/_base.py:73-74
    with pytest.raises(ValueError):
        FrankWolfe(quad=quad, ub=np.ones(3), t=1.0)             # frank_wolfe.py:84-85
    opt = InteriorPoint(quad=quad, ub=np.array([2., 4., 6.]))
    assert opt.eps == 1e-10 and opt.status == 'unknown' and opt.iter == 0 and np.isnan(opt.f_x)
    np.testing.assert_array_equal(opt.x, [1., 2., 3.])           # mid-box start, constrained/_base.py:65
    np.testing.assert_array_equal(opt.lb, np.zeros(3))
    assert hasattr(opt, 'x0_history')                            # ndim <= 3, opti/_base.py:78-82
    with pytest.raises(ValueError):
        PolyKernel(degree=0)
    with pytest.raises(ValueError):
        GaussianKernel(gamma='bogus')
    with pytest.raises(ValueError):
        SVC(loss=hinge, C=0)
    with pytest.raises(TypeError):
        SVC(loss=epsilon_insensitive)
    with pytest.raises(TypeError):
        SVR(loss=hinge)
    with pytest.raises(TypeError):
        SVC(loss=hinge, kernel='rbf')
    X = np.random.RandomState(0).standard_normal((8, 2))
    y = np.array([0, 1] * 4)
    with pytest.raises(NotImplementedError):                     # svm/_base.py:621-624
        SVC(loss=hinge, kernel=gaussian, dual=True, reg_intercept=False, optimizer=ProjectedGradient).fit(X, y)
    with pytest.raises(NotImplementedError):                     # svm/_base.py:771-774
        SVC(loss=squared_hinge, kernel=gaussian, dual=True, reg_intercept=True, optimizer=ProjectedGradient).fit(X, y)
    with pytest.raises(ValueError):                              # svm/_base.py:437-439
        SVC(loss=hinge, dual=True, reg_intercept=True, optimizer=ProjectedGradient).fit(X, np.arange(8))
    est = SVC(loss=hinge, dual=True, reg_intercept=True, optimizer=ProjectedGradient)
    assert est.get_params()['C'] == 1 and est.train_loss_history == []


def _build_c_consumer(tmp_path):
    import subprocess
    from optiml_amd import build
    lib = build.build()
    exe = str(tmp_path / 'abi_smoke')
    src = os.path.join(REPO, 'tests', 'c', 'abi_smoke.c')
    cmd = ['gcc', '-std=c99', '-Wall', '-Werror', '-I', os.path.join(REPO, 'include'), src, '-o', exe,
           '-L', os.path.dirname(lib), '-lbcqp_hip', '-Wl,-rpath,' + os.path.dirname(lib), '-Wl,-rpath,/opt/rocm/lib']
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    return exe


def test_c_consumer(tmp_path):
    """include/bcqp.h is valid C99 and the library links and runs from a plain C program (host-only calls here)."""
    import subprocess
    exe = _build_c_consumer(tmp_path)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, (r.returncode, r.stdout, r.stderr)
    assert 'c abi ok' in r.stdout


@pytest.mark.gpu
def test_c_consumer_solves_on_gpu(tmp_path):
    import subprocess
    exe = _build_c_consumer(tmp_path)
    r = subprocess.run([exe, 'gpu'], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, (r.returncode, r.stdout, r.stderr)
    assert 'c abi gpu ok' in r.stdout


def test_lagrangian_surface_and_validation_mirror_reference():
    """SURVEY 8(f).3 host side: class names, ctor checks and defaults of optiml/opti/constrained/_base.py:242-277 and
    optiml/opti/unconstrained/stochastic/*.py (no device call is made)."""
    import warnings
    from optiml_amd.opti import Quadratic
    from optiml_amd.opti.constrained import AugmentedLagrangianQuadratic
    from optiml_amd.opti.unconstrained import stochastic as st
    quad = Quadratic(np.eye(4) * 2, -np.ones(4))
    with pytest.raises(TypeError):
        AugmentedLagrangianQuadratic(primal=object())
    with pytest.raises(ValueError):
        AugmentedLagrangianQuadratic(primal=quad, A=np.ones(4))                 # missing b
    with pytest.raises(ValueError):
        AugmentedLagrangianQuadratic(primal=quad, b=np.zeros(1))                # missing A
    with pytest.raises(ValueError):
        AugmentedLagrangianQuadratic(primal=quad, h=np.zeros(1))                # missing G
    with pytest.raises(ValueError):
        AugmentedLagrangianQuadratic(primal=quad, lb=np.zeros(4), rho=0)
    with pytest.raises(NotImplementedError):
        AugmentedLagrangianQuadratic(primal=quad, G=np.eye(4), h=np.ones(4))    # general G rows: not built
    with pytest.raises(NotImplementedError):
        AugmentedLagrangianQuadratic(primal=quad, A=np.ones((2, 4)), b=np.zeros(2))
    al = AugmentedLagrangianQuadratic(primal=quad, A=[1, -1, 1, -1], b=np.zeros(1), lb=np.zeros(4), ub=np.ones(4), rho=2)
    assert al.ndim == 4 and al.n_eq == 1 and al.rho == 2 and al.primal is quad
    assert al.dual_x.shape == (9,) and not al.dual_x.any() and al.AG.shape == (9, 4) and al.bh.shape == (9,)
    x = np.array([1.5, -0.5, 0.25, 0.5])
    np.testing.assert_allclose(al.constraints(x), al.AG @ x - al.bh)
    for cls in (st.StochasticGradientDescent, st.Adam, st.AMSGrad, st.AdaMax, st.AdaGrad, st.AdaDelta, st.RMSProp):
        assert issubclass(cls, st.StochasticOptimizer)
        opt = cls(f=al, random_state=3)
        assert opt.is_lagrangian_dual() and opt.is_augmented_lagrangian_dual()
        np.testing.assert_array_equal(opt.x, np.random.RandomState(3).uniform(size=4))   # opti/_base.py:36-57
        np.testing.assert_array_equal(opt.past_x, opt.x)
        assert opt.epochs == 1000 and opt.epoch == 0 and opt.iter == 0 and opt.status == 'unknown'
        assert np.isnan(opt.f_x) and np.isnan(opt.primal_f_x) and np.isnan(opt.dgap)
        with pytest.raises(ValueError):
            cls(f=al, step_size=0)
        with pytest.raises(ValueError):
            cls(f=al, epochs=0)
        with pytest.raises(TypeError):
            cls(f=object())
    assert (st.StochasticGradientDescent(f=al).step_size, st.Adam(f=al).step_size, st.AdaMax(f=al).step_size,
            st.AdaGrad(f=al).step_size, st.AdaDelta(f=al).step_size, st.RMSProp(f=al).step_size) == \
        (0.01, 0.001, 0.002, 1., 1., 0.001)
    assert st.AdaDelta(f=al).offset == 1e-6 and st.AdaGrad(f=al).offset == 1e-8 and st.RMSProp(f=al).decay == 0.9
    for cls in (st.StochasticGradientDescent, st.Adam, st.AMSGrad, st.AdaMax, st.RMSProp):
        assert issubclass(cls, st.StochasticMomentumOptimizer)
        with pytest.raises(ValueError):
            cls(f=al, momentum_type='heavy')
        with pytest.raises(ValueError):
            cls(f=al, momentum=1.0)
    with pytest.raises(ValueError):
        st.Adam(f=al, beta1=1.0)
    with pytest.raises(ValueError):
        st.AdaDelta(f=al, decay=1.0)
    with pytest.raises(ValueError):
        st.AdaGrad(f=al, offset=0)
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter('always')
        st.Adam(f=al, beta1=0.99, beta2=0.9)                                     # adam.py: convergence-analysis warning
        assert len(w) == 1
    with pytest.raises(NotImplementedError):
        st.AdaGrad(f=al, batch_size=2)                                           # no samples to batch
    sched = st.StochasticGradientDescent(f=al, epochs=3, step_size=st.schedules.decaying(1., .5),
                                         momentum_type='polyak', momentum=st.schedules.sutskever_blend(0.9, 1))
    np.testing.assert_array_equal(sched._step_schedule, [1., .5, .25])          # one value per iteration, drawn up front
    np.testing.assert_allclose(sched._momentum_schedule, [0.75, 1 - 2 ** (-1 - np.log2(3)), 0.875])
    assert st.AdaGrad(f=al, step_size=lambda: st.schedules.decaying(2., .5)).step_size == 2.   # callable: first value


@pytest.mark.gpu
def test_collective_timeout_is_inert_without_an_rccl_communicator():
    """bq_ctx_set_collective_timeout on a plain (single-GPU) context is accepted and does nothing: a global
    BQ_COLLECTIVE_TIMEOUT_S must not cut short the long, legitimate waits of a single-GPU run.  (The watchdog itself: one-rank
    RCCL context, tests/test_distributed.py.)"""
    from optiml_amd import device
    ctx = device.Context(collective_timeout=0.1)
    ctx.probe_stall(500.0)          # five times the "limit": returns normally
    assert ctx.comm_info()['kind'] == 'none'
    r, c = ctx.probe_bandwidth(1 << 26, 2)
    assert r > 0 and c > 0
    ctx.close()


def test_the_product_package_imports_no_torch():
    """BASELINE's north star: Python host code over ctypes, no PyTorch.  The one torch user of earlier rounds — the gloo rendezvous of
    launcher-started ranks — lives outside the package since round 5 (bench_rendezvous.py); optiml_amd itself names torch nowhere."""
    import re
    pkg = os.path.join(REPO, 'optiml_amd')
    pat = re.compile(r'^\s*(import|from)\s+torch\b', re.M)
    hits = []
    for root, _, files in os.walk(pkg):
        for f in files:
            if f.endswith('.py') and pat.search(open(os.path.join(root, f)).read()):
                hits.append(os.path.join(root, f))
    assert hits == []
    import subprocess
    import sys
    code = 'import sys; import optiml_amd, optiml_amd.dist, optiml_amd.opti.constrained, optiml_amd.ml.svm; assert "torch" not in sys.modules'
    assert subprocess.run([sys.executable, '-c', code], cwd=REPO).returncode == 0
