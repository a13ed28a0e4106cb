#!/usr/bin/env python3
"""Headline benchmark: dual-QP iterations/sec of the device-resident ProjectedGradient solver on the RBF SVC
Wolfe dual, n=100 000, d=128, fp64 (BASELINE.json `metric`), on N GPUs of one node.

    python bench.py --gpus 1 --steps 50 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A "step" is one solver iteration: one symmetric panel product Q d (this rank's lower-triangle tiles, streamed once)
+ the fused O(n) kernels + one all-reduce of the n-vector for N > 1.  The Gram panel is built once before the timed
region and stays resident in HBM (its build time is reported separately).  N > 1: one process per GPU, rank r owns a
balanced triangular share of the tile rows, RCCL all-reduce per product; total work is fixed as N grows ("strong"
scaling).  torch.distributed (gloo) is used only for rendezvous / barrier / max-over-ranks — no torch tensor touches
the compute path.

Rank 0 prints ONE JSON line with `roofline` (HIP-event timing of the panel-product kernel on its own stream against
8 TB/s HBM, on the bytes that kernel has to move: its tiles + partial-product slab; `row_block_equivalent_GBs` restates
the rate in SURVEY 8(d)'s n^2*s bytes) and `cpu_baseline` (the NumPy oracle in the reference formulation on a bounded
sample).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.abspath(__file__))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 GB/s is the measured copy ceiling


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=50)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--samples', '--n', dest='n', type=int, default=100000)
    ap.add_argument('--features', '--d', dest='d', type=int, default=128)
    ap.add_argument('--solver', default='pg', choices=['pg', 'fw', 'adagrad', 'smo', 'ip', 'as'],
                    help='adagrad: AdaGrad on the augmented Lagrangian of the reg_intercept=False dual (SURVEY 8f.3); '
                         'smo: time-to-KKT-tol of SVC.fit(optimizer="smo") (SURVEY 8f.4; --steps/--warmup unused); '
                         'ip / as: time-to-KKT-tol of SVC.fit with InteriorPoint / ActiveSet (their own stop tests; pick '
                         '--samples to taste: n=100000 takes 473 s with ip)')
    ap.add_argument('--task', default='svc', choices=['svc', 'svr'], help='svr: eps-insensitive dual, dim 2n (config 4)')
    ap.add_argument('--kernel', default='rbf', choices=['rbf', 'poly', 'linear'], help='poly: degree 3, coef0 1')
    ap.add_argument('--storage', default='f64', choices=['f64', 'f32', 'stream'],
                    help='stream: no resident panel, Gram tiles recomputed on the MFMA inside every product')
    ap.add_argument('--exchange', default='rccl', choices=['rccl', 'host'])
    ap.add_argument('--cpu-n', type=int, default=12000, help='sample size of the CPU baseline leg')
    ap.add_argument('--cpu-steps', type=int, default=150)
    ap.add_argument('--no-cpu', action='store_true')
    ap.add_argument('--sigma', type=float, default=8.0, help='blob spread of the synthetic data (SURVEY 8d: 8 overlapping, 3 separable)')
    ap.add_argument('--cpu-study', action='store_true',
                    help='CPU only (SURVEY 8d): the oracle timed at three sizes to check the n^2 (PG) / n^3 (Cholesky) laws '
                         'behind the extrapolated baseline, plus a blocked Gram-streaming product at the full n')
    return ap.parse_args()


def measured_traffic(workload, world):
    """HBM bytes per launch of the panel-product kernel from the committed PMC passes (profiles/rNN/pmc_traffic_*.json,
    produced by tools/pmc_summary.py from separate `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` runs of this same
    command); None when no pass matches this workload."""
    import glob
    best = None
    for path in sorted(glob.glob(os.path.join(REPO, 'profiles', 'r*', 'pmc_traffic_*.json'))):
        try:
            rec = json.load(open(path))
        except Exception:
            continue
        meta = rec.get('meta', {})
        if meta.get('workload') != workload or int(meta.get('n_gpus', 1)) != world:
            continue
        for name, k in rec.get('kernels', {}).items():
            if name.startswith('symv_tiles'):
                best = {'hbm_bytes': k['hbm_bytes'], 'source': os.path.relpath(path, REPO)}
    return best


def cpu_baseline(args):
    """The oracle (NumPy restatement of the reference: dense Q on the host, 3 products per PG iteration) timed on
    this host's cores at a bounded n, then scaled by (n_sample / n)^2 to the headline size (the per-iteration
    cost is 3 streams of the n x n fp64 Hessian)."""
    from oracle import bcqp_oracle as bo
    from oracle import svm_oracle as so
    from optiml_amd.datasets import make_blobs
    try:
        from threadpoolctl import threadpool_info
        threads = max([p.get('num_threads', 1) for p in threadpool_info()] or [os.cpu_count() or 1])
    except Exception:
        threads = os.cpu_count() or 1
    ns = min(args.cpu_n, args.n)
    from optiml_amd.datasets import make_regression
    X, y = make_blobs(ns, args.d, seed=0) if args.task == 'svc' else make_regression(ns, args.d, seed=0)
    t0 = time.perf_counter()
    K = so.gram(args.kernel, X, None, 'scale', 1.0 if args.kernel == 'poly' else 0.0, 3)
    Q, q, ub = so.svc_dual(K, y, 1.0) if args.task == 'svc' else so.svr_dual(K, y, 1.0, 0.1)
    del K
    t_build = time.perf_counter() - t0
    if args.solver == 'adagrad':
        return cpu_baseline_al(args, threads)
    solve = bo.projected_gradient if args.solver == 'pg' else bo.frank_wolfe
    solve(Q, q, ub, max_iter=2)  # warm
    t0 = time.perf_counter()
    res = solve(Q, q, ub, max_iter=args.cpu_steps)
    dt = time.perf_counter() - t0
    its = res['iter']
    rate = its / dt
    scaled = rate * (ns / args.n) ** 2
    model = ''
    try:
        for line in open('/proc/cpuinfo'):
            if line.startswith('model name'):
                model = line.split(':', 1)[1].strip()
                break
    except OSError:
        pass
    return {'value': scaled, 'unit': 'iter/s', 'cores': int(threads), 'kind': 'port', 'cpu_model': model,
            'os_cpu_count': os.cpu_count(), 'OMP_NUM_THREADS': os.environ.get('OMP_NUM_THREADS'),
            'OPENBLAS_NUM_THREADS': os.environ.get('OPENBLAS_NUM_THREADS'),
            'sample': f'oracle {args.solver.upper()} (dense fp64 Q on host, 3 products/iter), n={ns} d={args.d}, '
                      f'{its} iterations in {dt:.2f}s = {rate:.3f} iter/s measured; value scaled by (n_s/n)^2 to '
                      f'n={args.n}; Gram+Q assembly {t_build:.2f}s excluded',
            'measured_iter_per_s_at_sample': rate, 'sample_n': ns}


def cpu_baseline_al(args, threads):
    """Oracle AdaGrad on the augmented Lagrangian in the reference formulation: dense Q and the dense stacked
    constraint matrix [a; -I; I] ((2N+1) x N), three products with Q per iteration."""
    from oracle import al_oracle as ao, svm_oracle as so
    from optiml_amd.datasets import make_blobs, make_regression
    ns = min(args.cpu_n, args.n, 4000)
    X, y = make_blobs(ns, args.d, seed=0) if args.task == 'svc' else make_regression(ns, args.d, seed=0)
    K = so.gram(args.kernel, X, None, 'scale', 1.0 if args.kernel == 'poly' else 0.0, 3)
    if args.task == 'svc':
        Q, q, a = K * np.outer(y, y), -np.ones(ns), y
    else:
        Q, q = np.vstack((np.hstack((K, -K)), np.hstack((-K, K)))), np.hstack((-y, y)) + 0.1
        a = np.hstack((np.ones(ns), -np.ones(ns)))
    N = len(q)
    al = ao.AugLag(Q, q, a=a, lb=np.zeros(N), ub=np.ones(N), rho=1.)
    steps = max(10, args.cpu_steps // 5)
    t0 = time.perf_counter()
    res = ao.minimize(al, np.random.RandomState(0).uniform(size=N), 'adagrad', epochs=steps + 1, step_size=1.)
    dt = time.perf_counter() - t0
    rate = res['iter'] / dt
    return {'value': rate * (ns / args.n) ** 2, 'unit': 'iter/s', 'cores': int(threads), 'kind': 'port',
            'sample': f'oracle AdaGrad on the augmented Lagrangian (dense fp64 Q and dense [a;-I;I] on host), n={ns} '
                      f'd={args.d}, {res["iter"]} iterations in {dt:.2f}s = {rate:.3f} iter/s measured; value scaled '
                      f'by (n_s/n)^2 to n={args.n}',
            'measured_iter_per_s_at_sample': rate, 'sample_n': ns}


def cpu_study(args):
    """SURVEY 8(d), 'CPU baseline timed beside it': the reference formulation cannot be held on the host at the headline
    size, so (1) time the oracle's PG at three sizes that fit and check the n^2 law the extrapolation rests on, (2) the
    same for the dense Cholesky InteriorPoint / ActiveSet spend their time in (n^3), and (3) time a blocked,
    Gram-streaming product (K never materialised, ONE product per iteration — the cheapest thing a CPU can do) at the
    full n for a few iterations: a measured bound on the CPU rate that needs no extrapolation."""
    import platform
    import scipy.linalg as sla
    from oracle import bcqp_oracle as bo
    from oracle import svm_oracle as so
    from optiml_amd.datasets import make_blobs
    try:
        from threadpoolctl import threadpool_info
        pools = [{k: p.get(k) for k in ('internal_api', 'num_threads', 'version')} for p in threadpool_info()]
    except Exception:
        pools = []
    model = platform.processor()
    try:
        for line in open('/proc/cpuinfo'):
            if line.startswith('model name'):
                model = line.split(':', 1)[1].strip()
                break
    except OSError:
        pass
    out = {'what': 'cpu_study', 'cpu_model': model, 'os_cpu_count': os.cpu_count(),
           'sched_affinity': len(os.sched_getaffinity(0)), 'blas_pools': pools,
           'OMP_NUM_THREADS': os.environ.get('OMP_NUM_THREADS'),
           'OPENBLAS_NUM_THREADS': os.environ.get('OPENBLAS_NUM_THREADS'), 'd': args.d}
    pg = []
    for ns in (12000, 20000, 30000):
        X, y = make_blobs(ns, args.d, seed=0)
        Q, q, ub = so.svc_dual(so.gram('rbf', X), y, 1.0)
        bo.projected_gradient(Q, q, ub, max_iter=2)
        steps = max(10, int(40 * (12000 / ns) ** 2))
        t0 = time.perf_counter()
        res = bo.projected_gradient(Q, q, ub, max_iter=steps)
        dt = (time.perf_counter() - t0) / res['iter']
        pg.append({'n': ns, 'iters': int(res['iter']), 's_per_iter': dt, 'GBs': 3 * ns * ns * 8 / dt / 1e9})
        print(f'[cpu-study] PG n={ns}: {dt * 1e3:.1f} ms/iter', file=sys.stderr, flush=True)
        del Q
    out['pg_reference_formulation'] = pg
    out['pg_exponent'] = float(np.polyfit(np.log([r['n'] for r in pg]), np.log([r['s_per_iter'] for r in pg]), 1)[0])
    out['pg_iter_per_s_extrapolated_to_n'] = {'n': args.n, 'value': 1.0 / (pg[-1]['s_per_iter'] * (args.n / pg[-1]['n']) ** 2)}
    ch = []
    for ns in (6000, 12000, 20000):
        X, y = make_blobs(ns, args.d, seed=0)
        H = so.gram('rbf', X)
        H[np.diag_indices(ns)] += 1.0
        t0 = time.perf_counter()
        sla.cho_factor(H, overwrite_a=True, check_finite=False)
        dt = time.perf_counter() - t0
        ch.append({'n': ns, 's': dt, 'GFLOPs': ns ** 3 / 3 / dt / 1e9})
        print(f'[cpu-study] cho_factor n={ns}: {dt:.2f} s', file=sys.stderr, flush=True)
        del H
    out['cholesky'] = ch
    out['cholesky_exponent'] = float(np.polyfit(np.log([r['n'] for r in ch]), np.log([r['s'] for r in ch]), 1)[0])
    out['cholesky_s_extrapolated'] = {str(m): ch[-1]['s'] * (m / ch[-1]['n']) ** 3 for m in (50000, 100000)}
    # blocked Gram-streaming product at the full n: rows of K are formed block by block and applied at once
    n, blk = args.n, 2000
    X, y = make_blobs(n, args.d, seed=0)
    gamma = so.resolve_gamma('scale', X)
    sq = np.einsum('ij,ij->i', X, X)
    v = np.random.RandomState(0).uniform(size=n) * y
    times = []
    for _ in range(3):
        t0 = time.perf_counter()
        o = np.empty(n)
        for r0 in range(0, n, blk):
            r1 = min(n, r0 + blk)
            D = X[r0:r1] @ X.T
            D *= -2.0
            D += sq[r0:r1, None]
            D += sq[None, :]
            np.maximum(D, 0.0, out=D)
            D *= -gamma
            np.exp(D, out=D)
            o[r0:r1] = D @ v + v.sum()
        times.append(time.perf_counter() - t0)
        print(f'[cpu-study] streamed product n={n}: {times[-1]:.2f} s', file=sys.stderr, flush=True)
    out['streamed_product_full_n'] = {'n': n, 'block_rows': blk, 's_per_product': times, 'best_iter_per_s_1_product': 1.0 / min(times)}
    print(json.dumps(out), flush=True)


def bench_smo(args):
    """BASELINE.json's second metric, time-to-KKT-tol, on the route that reaches it fastest: SVC.fit(optimizer='smo') end
    to end (Gram build + sweeps) on one GPU; CPU baseline: the oracle's SMO sweeps at a bounded n (Gram excluded)."""
    from oracle import smo_oracle as smo, svm_oracle as so
    from optiml_amd import device
    from optiml_amd.datasets import make_blobs
    from optiml_amd.ml.svm import SVC
    from optiml_amd.ml.svm.kernels import gaussian
    from optiml_amd.ml.svm.losses import hinge
    ctx = device.get_context()
    X, y = make_blobs(args.n, args.d, seed=0, sigma=args.sigma)
    t0 = time.perf_counter()
    est = SVC(loss=hinge, kernel=gaussian, C=1., dual=True, optimizer='smo', tol=1e-3).fit(X, y)
    dt = time.perf_counter() - t0
    out = {'metric': 'time_to_kkt_tol', 'value': dt, 'unit': 's', 'n_gpus': 1, 'steps': int(est.optimizer.iter), 'warmup': 0,
           'ms_per_step': 1e3 * dt / max(est.optimizer.iter, 1), 'higher_is_better': False, 'scaling': 'strong',
           'vs_baseline': None, 'dtype': 'f64', 'data': 'synthetic',
           'config': {'workload': f'svc_hinge_rbf_smo_dual_n{args.n}_d{args.d}', 'n': args.n, 'd': args.d, 'C': 1.0,
                      'tol': 1e-3, 'gamma': 'scale', 'solver': 'smo', 'device': ctx.name},
           'roofline': None, 'pair_steps': int(est.optimizer.steps), 'outer_iterations': int(est.optimizer.iter),
           'n_sv': int(len(est.support_))}
    # `value` is the first fit of a fresh process (it includes the first 40 GB device allocation, which varies from 0.05 to
    # 1 s between boxes); the same fit again in the warm process is reported next to it
    t0 = time.perf_counter()
    SVC(loss=hinge, kernel=gaussian, C=1., dual=True, optimizer='smo', tol=1e-3).fit(X, y)
    out['second_fit_s'] = time.perf_counter() - t0
    if not args.no_cpu:
        ns = min(args.cpu_n, args.n)
        K = so.gram('rbf', X[:ns])
        yb = np.where(y[:ns] == np.unique(y)[-1], 1., -1.)
        t0 = time.perf_counter()
        r = smo.smo_svc(K, yb, 1., 1e-3)
        dtc = time.perf_counter() - t0
        out['cpu_baseline'] = {'value': dtc, 'unit': 's', 'cores': int(os.cpu_count() or 1), 'kind': 'port',
                               'sample': f'oracle SMO sweeps (reference algorithm in NumPy, dense K on host, Gram build '
                                         f'excluded) at n={ns}: {r["iter"]} outer iterations, {r["steps"]} pair steps'}
    else:
        out['cpu_baseline'] = None
    print(json.dumps(out), flush=True)


def bench_kkt(args):
    """BASELINE.json's second metric, time-to-KKT-tol, for the two box solvers that reach their own stop test:
    SVC.fit(optimizer=InteriorPoint | ActiveSet) end to end on one GPU.  CPU baseline: the oracle (reference algorithm,
    dense Q on the host, scipy's cho_factor per iteration) at a bounded n — its measured seconds, not extrapolated."""
    from oracle import bcqp_oracle as bo, svm_oracle as so
    from optiml_amd import device
    from optiml_amd.datasets import make_blobs
    from optiml_amd.ml.svm import SVC
    from optiml_amd.ml.svm.kernels import gaussian
    from optiml_amd.ml.svm.losses import hinge
    from optiml_amd.opti.constrained import ActiveSet, InteriorPoint
    ctx = device.get_context()
    X, y = make_blobs(args.n, args.d, seed=0, sigma=args.sigma)
    cls = InteriorPoint if args.solver == 'ip' else ActiveSet
    t0 = time.perf_counter()
    est = SVC(loss=hinge, kernel=gaussian, C=1., reg_intercept=True, dual=True, optimizer=cls, max_iter=10 ** 7).fit(X, y)
    dt = time.perf_counter() - t0
    o = est.optimizer
    out = {'metric': 'time_to_kkt_tol', 'value': dt, 'unit': 's', 'n_gpus': 1, 'steps': int(o.iter), 'warmup': 0,
           'ms_per_step': 1e3 * dt / max(o.iter, 1), 'higher_is_better': False, 'scaling': 'strong', 'vs_baseline': None,
           'dtype': 'f64', 'data': 'synthetic',
           'config': {'workload': f'svc_hinge_rbf_{args.solver}_dual_n{args.n}_d{args.d}', 'n': args.n, 'd': args.d, 'C': 1.0,
                      'gamma': 'scale', 'solver': args.solver, 'device': ctx.name},
           'roofline': None, 'status': o.status, 'f': float(o.f_x), 'n_sv': int(len(est.support_))}
    if not args.no_cpu:
        # bounded CPU samples: ActiveSet needs ~n iterations of an n^3/3 factorisation each — minutes already at n=3000
        ns = min(args.cpu_n, args.n, 3000 if args.solver == 'ip' else 800)
        Q, q, ub = so.svc_dual(so.gram('rbf', X[:ns]), y[:ns], 1.0)
        fn = bo.interior_point if args.solver == 'ip' else bo.active_set
        t0 = time.perf_counter()
        r = fn(Q, q, ub, max_iter=10 ** 7)
        dtc = time.perf_counter() - t0
        out['cpu_baseline'] = {'value': dtc, 'unit': 's', 'cores': int(os.cpu_count() or 1), 'kind': 'port',
                               'sample': f'oracle {args.solver.upper()} (reference algorithm in NumPy/SciPy, dense Q on host, Gram '
                                         f'and Q assembly excluded) run to its stop test at n={ns}: {r["iter"]} iterations, '
                                         f'status {r["status"]}'}
    else:
        out['cpu_baseline'] = None
    print(json.dumps(out), flush=True)


def main():
    args = parse()
    if args.cpu_study:
        return cpu_study(args)
    if args.solver in ('ip', 'as'):
        if args.gpus != 1:
            raise SystemExit('InteriorPoint / ActiveSet factorise on one GPU (replicas only): --gpus 1')
        return bench_kkt(args)
    if args.solver == 'smo':
        if args.gpus != 1:
            raise SystemExit('SMO walks the samples sequentially on one GPU (replicas only): --gpus 1')
        return bench_smo(args)
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit('launch N > 1 with: python -m torch.distributed.run --nproc-per-node N bench.py --gpus N')
        raise SystemExit(f'--gpus {args.gpus} does not match WORLD_SIZE={world}')

    from optiml_amd import _lib
    from optiml_amd import device
    from optiml_amd.datasets import make_blobs
    from optiml_amd.ml.svm.kernels import gaussian
    from optiml_amd.opti import KernelQuadratic
    from optiml_amd.opti.constrained._base import _DeviceSolver

    comm = None
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group(backend='gloo', rank=rank, world_size=world)
        from optiml_amd.dist import TorchComm
        comm = TorchComm()
        # RCCL over xGMI is the data path.  If the communicator cannot be created on every rank (reported, never
        # silent), the row-block exchange falls back to the host (gloo) transport so that the run still measures the
        # sharded GPU path; `config.exchange` in the JSON line says which transport was used.
        exchange, ctx, err = args.exchange, None, ''
        if exchange == 'rccl':
            try:
                ctx = device.Context(comm=comm, exchange='rccl')
            except Exception as exc:  # noqa: BLE001
                err = repr(exc)
            if comm.max_float(0.0 if ctx is not None else 1.0) > 0.0:
                print(f'[bench] rank {rank}: RCCL context unavailable ({err or "failed on another rank"}); '
                      'using the host exchange', file=sys.stderr, flush=True)
                if ctx is not None:
                    ctx.close()
                ctx, exchange = None, 'host'
        if ctx is None:
            ctx = device.Context(comm=comm, exchange=exchange)
        device.set_context(ctx)
    else:
        ctx = device.get_context()

    def barrier():
        if comm is not None:
            comm.barrier()

    n, d = args.n, args.d
    from optiml_amd.datasets import make_regression
    from optiml_amd.ml.svm.kernels import PolyKernel, linear
    kern = {'rbf': gaussian, 'poly': PolyKernel(3, 'scale', 1.0), 'linear': linear}[args.kernel]
    al = args.solver == 'adagrad'   # reg_intercept=False dual: no rank-one term, equality row handled by the multiplier
    if args.task == 'svc':
        X, y = make_blobs(n, d, seed=0, sigma=args.sigma)
        quad = KernelQuadratic(X, -np.ones(n), 'svc', kern, y=y, storage=args.storage, rank_one=not al)
        a_eq = y
    else:   # eps-insensitive SVR dual: 2n variables on one n x n panel (BASELINE config 4 shape)
        X, y = make_regression(n, d, seed=0)
        quad = KernelQuadratic(X, np.hstack((-y, y)) + 0.1, 'svr', kern, storage=args.storage, rank_one=not al)
        a_eq = np.hstack((np.ones(n), -np.ones(n)))
    N = quad.ndim
    ctx.profile(os.environ.get('BQ_BENCH_NOPROF', '0') != '1')   # HIP-event timing of the dominant kernel (roofline)
    t0 = time.perf_counter()
    dev = quad.device_problem(ctx)
    t_gram_total = time.perf_counter() - t0
    gram_ms, _ = ctx.profile_read(_lib.PROF_GRAM, reset=True)
    _, _, r0, r1 = dev.dims()

    ub = np.ones(N)
    if al:
        from optiml_amd.opti.unconstrained.stochastic._base import _AlDeviceSolver
        prm = _lib.AlParams(rule=_lib.RULE_ADAGRAD, momentum_type=0, step_size=1., momentum=0., beta1=0., beta2=0.,
                            decay=0., offset=1e-8, rho=1., tol=1e-12, epochs=10 ** 9)
        solver = _AlDeviceSolver(dev, prm, a_eq, np.zeros(N), ub, np.random.RandomState(0).uniform(size=N), None)
    else:
        kind = _lib.PG if args.solver == 'pg' else _lib.FW
        solver = _DeviceSolver(dev, kind, np.zeros(N), ub, ub / 2, 1e-6, 10 ** 9)

    rows, status = solver.run(max(args.warmup, 1))      # includes the start-up gradient product
    ctx.profile_read(_lib.PROF_MATVEC, reset=True)
    ctx.profile_read(_lib.PROF_EXCH, reset=True)
    barrier()
    t0 = time.perf_counter()
    rows, status = solver.run(args.steps)                # device-resident; returns after the stream has drained
    t1 = time.perf_counter()
    barrier()
    elapsed = t1 - t0
    if comm is not None:
        elapsed = comm.max_float(elapsed)
    done = len(rows)
    mv_ms, mv_cnt = ctx.profile_read(_lib.PROF_MATVEC, reset=True)
    ex_ms, ex_cnt = ctx.profile_read(_lib.PROF_EXCH, reset=True)

    esz = 8 if args.storage == 'f64' else 4
    probe = (0.0, 0.0)
    if rank == 0:
        try:   # outside the timed region: what a plain streaming read / copy reaches on this GPU (8 GiB scratch)
            probe = ctx.probe_bandwidth(4 << 30, 5)
        except Exception as exc:  # noqa: BLE001  (e.g. not enough free HBM beside a very large panel)
            print(f'[bench] bandwidth probe skipped: {exc!r}', file=sys.stderr, flush=True)
    if rank == 0:
        avg_ms = mv_ms / max(mv_cnt, 1)
        # the dominant kernel is the symmetric tile product: this rank streams the 256 x 256 tiles on/below the
        # diagonal of its tile rows once and writes two 256-vectors per tile into the partial-product slab
        T = 256
        i0, i1 = r0 // T, -(-r1 // T)
        tiles = i1 * (i1 + 1) // 2 - i0 * (i0 + 1) // 2
        alg_bytes = tiles * (T * T * esz + T * 8) + (tiles // 8 + i1 - i0) * T * 8 + 2 * n * 8   # tiles + col parts + row parts
        achieved = alg_bytes / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
        full_equiv = ((r1 - r0) * n * esz + 3 * n * 8) / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
        workload = (f'svc_hinge_{args.kernel}_{args.solver}_dual_n{n}_d{d}' if args.task == 'svc' else
                    f'svr_epsins_{args.kernel}_{args.solver}_dual_n{n}_d{d}')
        traffic = measured_traffic(workload, world) if args.storage == 'f64' else None
        out = {
            'metric': 'dual_qp_iterations_per_sec', 'value': done / elapsed, 'unit': 'iter/s',
            'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': 1e3 * elapsed / max(done, 1),
            'higher_is_better': True, 'scaling': 'strong', 'vs_baseline': None,
            'dtype': 'f64' if args.storage in ('f64', 'stream') else 'f32-storage/f64-accumulate', 'data': 'synthetic',
            'config': {'workload': workload, 'n': n, 'd': d, 'dual_dim': N, 'C': 1.0,
                       'gamma': 'scale', 'solver': args.solver, 'exchange': ctx.exchange, 'blob_sigma': args.sigma,
                       'rows_per_gpu': r1 - r0, 'device': ctx.name},
            'roofline': {'bound': 'hbm', 'kernel': 'symv_tiles_kernel (symmetric panel product Q*d)', 'achieved': achieved,
                         'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': achieved / HBM_PEAK_GBS,
                         'traffic': traffic['hbm_bytes'] if traffic else None,
                         'traffic_source': traffic['source'] if traffic else None, 'avg_launch_ms': avg_ms, 'launches': mv_cnt,
                         'algorithmic_bytes_per_launch': alg_bytes, 'tiles_per_launch': tiles,
                         'row_block_equivalent_GBs': full_equiv, 'measured_stream_read_GBs': probe[0],
                         'measured_copy_GBs': probe[1],
                         'frac_of_measured_stream_read': (achieved / probe[0]) if probe[0] else None},
            'steps_done': done, 'solver_status': status,
            'f_last': float(rows['f'][-1]) if done else None,
            'kkt_resid_last': float(rows['r1'][-1]) if done and args.solver == 'pg' else None,
            'gram_build_s': gram_ms * 1e-3, 'problem_setup_s': t_gram_total,
            'exchange_ms_per_step': (ex_ms / max(ex_cnt, 1)) if ex_cnt else 0.0,
        }
        if args.storage == 'stream':   # no panel: the product is the fused Gram-tile x vector kernel, MFMA-bound
            flops = 2.0 * (-(-(r1 - r0) // 128) * 128) * (-(-n // 128) * 128) * (-(-d // 16) * 16)
            tf = flops / (avg_ms * 1e-3) / 1e12 if avg_ms > 0 else 0.0
            out['roofline'] = {'bound': 'mfma', 'kernel': 'gram_stream_kernel (Gram tiles recomputed, fused with the product)',
                               'achieved': tf, 'peak': 78.6, 'unit': 'TFLOP/s', 'frac': tf / 78.6, 'traffic': None,
                               'avg_launch_ms': avg_ms, 'launches': mv_cnt, 'flops_per_launch': flops}
        if world == 1 and not args.no_cpu:
            out['cpu_baseline'] = cpu_baseline(args)
            out['speedup_vs_cpu_baseline'] = out['value'] / out['cpu_baseline']['value']
        else:
            out['cpu_baseline'] = None
        print(json.dumps(out), flush=True)
    barrier()
    solver.close()
    if comm is not None:
        import torch.distributed as dist
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
