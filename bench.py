#!/usr/bin/env python3
"""Headline benchmark: dual-QP iterations/sec of the device-resident ProjectedGradient solver on the RBF SVC
Wolfe dual, n=100 000, d=128, fp64 (BASELINE.json `metric`), on N GPUs of one node, plus BASELINE's second metric,
time-to-KKT-tol, as sub-records of the same JSON line (N = 1).

    python bench.py --gpus N --steps K --warmup W          # N > 1: spawns N fresh rank processes itself
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W             # ... or runs as one rank of a launcher

A "step" is one solver iteration: one symmetric panel product Q d (this rank's lower-triangle tiles, streamed once)
+ the fused O(n) kernels + ONE collective for N > 1 (RCCL all-gather of the per-segment partial vectors, which every rank
adds in a fixed order: iterates are bit-identical for N = 1, 2, 4, 8).  The Gram panel is built once before the timed
region and stays resident in HBM (its build time is reported separately).  N > 1: one process per GPU, rank r owns a
balanced triangular share of the tile rows; total work is fixed as N grows ("strong" scaling).  torch.distributed
(gloo) is used only for rendezvous / barrier / max-over-ranks — no torch tensor touches the compute path.  If the RCCL
communicator cannot be created the run FAILS (exit 3) unless --allow-host-exchange is given.

Rank 0 prints ONE JSON line with `roofline` (HIP-event timing of the panel-product kernel on its own stream against
8 TB/s HBM, on the bytes that kernel has to move: its lower-triangle tiles + partial-product slab; `frac_survey_8d_bytes`
restates it in SURVEY 8(d)'s n^2*s bytes, which the kernel does not move), `cpu_baseline` (the NumPy oracle in the reference
formulation timed at two sizes that fit the host and extrapolated with the fitted exponent) and `time_to_kkt`.

The DEFAULT single-GPU line (no workload flag: what the driver runs) also carries, after the headline record is complete and each
in a fresh child process of its own with a time budget (`--records`, `--budget-s`): `configs` = BASELINE's other GPU configs on
this one GPU (c2, c4, c5, each with its own roofline), `shares` = every rank's share of the 2-, 4- and 8-way partitions of the
headline panel (and of c4 over 4, c5 over 8) timed alone with the iteration rate they predict, and `collective_floor_us` = the
closing collectives of the real message sizes on a ONE-rank RCCL communicator.  The parent process of that mode never touches HIP.

Other workloads: --config c2|c3|c4|c5 (BASELINE.json configs), --solver fw|adagrad|ascg|smo|ip|as, --task svr, ...
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.abspath(__file__))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 GB/s is the measured copy ceiling
# `value` is a steady-state rate (iterations per second, set-up reported separately): the panel placement choice is told
# that many products follow, so that it may spend its maximum (5 s) instead of SVC.fit's 2 % of max_iter products (config.placement_budget)
STEADY_STATE_PRODUCTS = 1e9
FP64_MFMA_PEAK_TF = 78.6

LINE_LIMIT = 4096       # bytes of the last stdout line (the driver keeps a 12.8 KB tail of stdout: r04's 25.6 KB line was cut)

SIDE_RECORDS = ('c2', 'c4', 'c5', 'shares', 'collective', 'shares_c4', 'shares_c5', 'fixed_cap', 'fixed_cap_c5', 'as_c2', 'dense_c2')   # in the order they are measured

CONFIGS = {   # BASELINE.json `configs` (C2 .. C5) and the headline
    'headline': dict(n=100000, d=128, solver='pg', task='svc', kernel='rbf', storage='f64'),
    'c2': dict(n=20000, d=64, solver='pg', task='svc', kernel='rbf', storage='f64'),
    'c3': dict(n=50000, d=128, solver='ip', task='svc', kernel='rbf', storage='f64'),
    'c4': dict(n=100000, d=128, solver='fw', task='svr', kernel='poly', storage='f64'),
    'c5': dict(n=250000, d=256, solver='ascg', task='svc', kernel='rbf', storage='f32'),
}


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=50)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--config', default=None, choices=sorted(CONFIGS), help='preset of --samples/--features/--solver/--task/--kernel/--storage')
    ap.add_argument('--samples', '--n', dest='n', type=int, default=None)
    ap.add_argument('--features', '--d', dest='d', type=int, default=None)
    ap.add_argument('--solver', default=None, choices=['pg', 'fw', 'adagrad', 'ascg', 'smo', 'ip', 'as'],
                    help='adagrad: AdaGrad on the augmented Lagrangian of the reg_intercept=False dual (SURVEY 8f.3); '
                         'ascg: ActiveSet with conjugate-gradient restricted solves on the squared-hinge dual (BASELINE config 5: '
                         'a step is one OUTER iteration = tens of panel products); '
                         'smo: time-to-KKT-tol of SVC.fit(optimizer="smo") (SURVEY 8f.4; --steps/--warmup unused); '
                         'ip / as: time-to-KKT-tol of SVC.fit with InteriorPoint / ActiveSet (their own stop tests; pick '
                         '--samples to taste: n=100000 takes 473 s with ip)')
    ap.add_argument('--task', default=None, choices=['svc', 'svr'], help='svr: eps-insensitive dual, dim 2n (config 4)')
    ap.add_argument('--kernel', default=None, choices=['rbf', 'poly', 'linear'], help='poly: degree 3, coef0 1')
    ap.add_argument('--storage', default=None, choices=['f64', 'f32', 'stream'],
                    help='stream: no resident panel, Gram tiles recomputed on the MFMA inside every product')
    ap.add_argument('--exchange', default='rccl', choices=['rccl', 'host'])
    ap.add_argument('--sym-exchange', default='gather', choices=['gather', 'allreduce'],
                    help='closing collective of a symmetric product: all-gather of segment partials + ordered sum (default, '
                         'bit-identical for any N) or one all-reduce(sum)')
    ap.add_argument('--rccl-init-timeout', type=float, default=240.0,
                    help='seconds a rank waits inside the RCCL communicator bootstrap before the run is given up (exit 3)')
    ap.add_argument('--collective-timeout', type=float, default=None,
                    help='N > 1 over RCCL: seconds a host wait on the stream may last before the library aborts the communicator '
                         '(bq_ctx_set_collective_timeout: a rank whose peer stopped taking part then FAILS instead of sitting in the '
                         'collective until the caller\'s limit).  Default: 120 + 0.1 per step; 0: no bound')
    ap.add_argument('--allow-host-exchange', action='store_true',
                    help='fall back to the host (gloo) exchange when the RCCL communicator cannot be created (default: exit 3)')
    ap.add_argument('--cpu-sizes', default='20000,30000,40000',
                    help='sample sizes of the CPU baseline leg (SURVEY 8d); a size whose dense fp64 Q (+ one temporary of the same size) '
                         'does not fit into half of the host\'s free memory is left out, and the record says so')
    ap.add_argument('--cpu-seconds', type=float, default=6.0, help='timed seconds per CPU sample size')
    ap.add_argument('--no-cpu', action='store_true')
    ap.add_argument('--no-placement', action='store_true',
                    help='take the panel where the first allocation puts it.  Default: the library\'s placement choice (BQ_PLACE_PANEL / '
                         'KernelQuadratic(tune_placement=True), opt-in for SVC / SVR): the product is timed on the empty panel and up to '
                         'two more allocations are tried if it streams below ~6.5 TB/s, the fastest is kept; the times of all candidates '
                         'are in config.panel_placement_ms, the cost in problem_setup_s')
    ap.add_argument('--records', default=None, metavar='all|none|NAME[,NAME...]',
                    help='side records of the default single-GPU line, each measured in a fresh child process after the headline '
                         'record is complete: ' + ','.join(SIDE_RECORDS) + '.  Default: all when no workload flag is given '
                         '(the driver\'s command), none otherwise')
    ap.add_argument('--line', default='compact', choices=['compact', 'full'],
                    help='what the LAST stdout line is: compact (default) = the headline record cut to what the driver parses, under '
                         f'{LINE_LIMIT} bytes, with the complete record written to --records-file; full = the complete record itself '
                         '(what the child processes of the default line print for their parent)')
    ap.add_argument('--records-file', default=os.path.join(REPO, 'bench_records.json'),
                    help='where the complete (uncut) record of this run is written as indented JSON (also one line on stderr)')
    ap.add_argument('--budget-s', type=float, default=400.0,
                    help='wall-clock budget of the whole default line: a side record that would not fit is skipped (and says so)')
    ap.add_argument('--fixed-cap', type=int, default=0, metavar='CAP',
                    help='SURVEY 8(d): the solvers that do not reach their tolerance at these sizes, run to a fixed iteration cap: for '
                         'each workload of --fixed-cap-configs (default: this one) iterations done, status, f and the projected-gradient '
                         'norm computed the SAME way for every solver (d = -g with PG\'s masks, g from a fresh product at the final point), '
                         'wall time; ActiveSetCG also |L| + |U| and the inner products')
    ap.add_argument('--fixed-cap-configs', default=None, metavar='NAME[,NAME...]', help='--fixed-cap: workloads (headline, c2, c4, c5)')
    ap.add_argument('--collective-floor', action='store_true',
                    help='ONE-rank RCCL communicator: ncclAllGather / ncclAllReduce of the real message sizes of the headline, c4 '
                         'and c5 products timed with HIP events (launch + local-copy floor of the one collective per product)')
    ap.add_argument('--cpu-stream-iters', type=int, default=1,
                    help='also time a blocked Gram-streaming CPU product at the FULL n this many times (~35 s each at n=100000): a '
                         'measured CPU bound beside the extrapolated reference-formulation figure (SURVEY 8d)')
    ap.add_argument('--kkt', default='all', choices=['none', 'smo', 'ip', 'all', 'ip100k'],
                    help="time_to_kkt sub-records of the default line (N=1): smo = SVC.fit(optimizer='smo') at the workload's "
                         'n (tol 1e-3), ip = InteriorPoint at BASELINE config 3 shape n=50000 d=128 (gap 1e-10, ~50 s); ip100k = all of '
                         'these + InteriorPoint at the HEADLINE size n=100000 d=128 (~6.5 min: not in the default line)')
    ap.add_argument('--sigma', type=float, default=8.0, help='blob spread of the synthetic data (SURVEY 8d: 8 overlapping, 3 separable)')
    ap.add_argument('--inner-tol', type=float, default=1e-8, help='ascg: relative residual of the inner conjugate gradients')
    ap.add_argument('--emulate-shares', default=None, metavar='G[,G...]',
                    help='ONE GPU, one process: for every rank k of a G-way partition build only that rank\'s share of the panel '
                         '(bq_ctx_create_share: every per-rank kernel of the iteration, collectives are no-ops) and time it; prints '
                         'one JSON record with per-share kernel time / GB/s / fixed cost and the iteration rate they predict for '
                         'G GPUs (e.g. --emulate-shares 1,2,4,8)')
    ap.add_argument('--probe-gib', default='', metavar='GiB[,GiB...]',
                    help='--emulate-shares: also run the bare streaming-read / copy probe on scratch buffers of these sizes (is the '
                         'read rate a function of the footprint?)')
    ap.add_argument('--exchange-us', type=float, default=50.0,
                    help='--emulate-shares: ASSUMED duration of the one collective per product (not measurable on one GPU); the '
                         'prediction is printed with it and with 0')
    ap.add_argument('--dense', action='store_true',
                    help='dense Quadratic(Q, q) of --samples variables (optiml/opti/_base.py:228-300; a banded symmetric Q whose product '
                         'is known in closed form): the product and ProjectedGradient iterations on the packed lower-triangle copy the '
                         'library keeps when Q == Q\' exactly, and on the row blocks it keeps otherwise (forced), each with its roofline')
    ap.add_argument('--wall-limit', type=float, default=0.0,
                    help='--solver ip | as: stop the fit after this many seconds (0: none) and report how far it got — for the fits that '
                         'need ~n outer iterations at the headline size (tools/profile_kkt_headline.sh)')
    ap.add_argument('--cpu-study', action='store_true',
                    help='CPU only (SURVEY 8d): the oracle timed at three sizes to check the n^2 (PG) / n^3 (Cholesky) laws '
                         'behind the extrapolated baseline, plus a blocked Gram-streaming product at the full n')
    args = ap.parse_args(argv)
    # no workload flag at all = the driver's command: the line then carries the side records too
    args.default_workload = args.config is None and all(getattr(args, k) is None for k in ('n', 'd', 'solver', 'task', 'kernel', 'storage'))
    if args.records is None:
        args.records = 'all' if args.default_workload else 'none'
    preset = CONFIGS[args.config or 'headline']
    for k, v in preset.items():
        if getattr(args, k) is None:
            setattr(args, k, v)
    return args


# ---------------------------------------------------------------------------------------------------------------------
# self-launch: python bench.py --gpus N from a plain shell
# ---------------------------------------------------------------------------------------------------------------------
def spawn_ranks(n):
    """Start N fresh rank processes (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in their environment) and relay them.  This
    parent process never loads the HIP library or touches the GPU: it only waits, forwards rank 0's JSON line and returns the
    worst exit code.  A rank that dies takes the others down (their collectives could never complete)."""
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY='0')
        # few host threads per rank: the ranks do no heavy CPU work, and N x (cores / N) spinning OpenMP threads beside the HIP and
        # gloo threads oversubscribe the host (r03: torch's intra-op pool turned a 0.5 MB host all-gather into 600 ms)
        env.setdefault('OMP_NUM_THREADS', str(max(1, min(4, (os.cpu_count() or 8) // n))))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=None if r == 0 else subprocess.DEVNULL))   # rank 0 prints the JSON line
    rc = 0
    try:
        while any(p.poll() is None for p in procs):
            time.sleep(0.2)
            bad = [p.returncode for p in procs if p.poll() is not None and p.returncode != 0]
            if bad:
                rc = rc or bad[0]
                for q in procs:          # exact PIDs of our own children
                    if q.poll() is None:
                        q.kill()
        for p in procs:
            if p.returncode != 0:
                rc = rc or p.returncode
    finally:
        for q in procs:
            if q.poll() is None:
                q.kill()
    return rc if rc >= 0 else 1


# ---------------------------------------------------------------------------------------------------------------------
# CPU legs (the oracle is imported HERE only: it is the thing timed beside the device path, never part of it)
# ---------------------------------------------------------------------------------------------------------------------
def _blas_threads():
    """(threads the BLAS behind numpy's `@` actually runs, how that was found out).  threadpoolctl when importable; otherwise the
    OpenBLAS that numpy loaded is asked directly (openblas_get_num_threads through ctypes on the library in /proc/self/maps) —
    os.cpu_count() is NOT the answer on a 256-thread host where OpenBLAS was built for 64."""
    try:
        from threadpoolctl import threadpool_info
        pools = [p for p in threadpool_info() if p.get('user_api') == 'blas'] or threadpool_info()
        if pools:
            return max(p.get('num_threads', 1) for p in pools), 'threadpoolctl'
    except Exception:  # noqa: BLE001
        pass
    try:
        import ctypes
        np.dot(np.ones(4), np.ones(4))   # make sure the BLAS is loaded
        libs = sorted({line.split()[-1] for line in open('/proc/self/maps') if 'openblas' in line.lower() and '.so' in line})
        for path in libs:
            lib = ctypes.CDLL(path)
            for sym in ('openblas_get_num_threads', 'openblas_get_num_threads64_', 'scipy_openblas_get_num_threads64_',
                        'scipy_openblas_get_num_threads'):
                fn = getattr(lib, sym, None)
                if fn is not None:
                    fn.restype = ctypes.c_int
                    return int(fn()), f'{sym} of {os.path.basename(path)}'
    except Exception:  # noqa: BLE001
        pass
    env = os.environ.get('OPENBLAS_NUM_THREADS') or os.environ.get('OMP_NUM_THREADS')
    if env and env.isdigit():
        return int(env), 'environment'
    return os.cpu_count() or 1, 'os.cpu_count() (BLAS not identified)'


def _cpu_info():
    threads, how = _blas_threads()
    model = ''
    try:
        for line in open('/proc/cpuinfo'):
            if line.startswith('model name'):
                model = line.split(':', 1)[1].strip()
                break
    except OSError:
        pass
    return {'cores': int(threads), 'cores_source': how, 'cpu_model': model, 'os_cpu_count': os.cpu_count(),
            'OMP_NUM_THREADS': os.environ.get('OMP_NUM_THREADS'), 'OPENBLAS_NUM_THREADS': os.environ.get('OPENBLAS_NUM_THREADS')}


def cpu_baseline(args):
    """The oracle (NumPy restatement of the reference: dense fp64 Q on the host, 3 products per PG/FW iteration) timed on
    this host's cores at the sizes SURVEY 8(d) names (n = 20 000 and 30 000: the reference formulation cannot hold n = 100 000),
    then EXTRAPOLATED to the workload's n: with the exponent fitted between the two samples (`value`) and, next to it, with the
    n^2 law of a bandwidth-bound product (`value_n2_law`)."""
    from oracle import bcqp_oracle as bo
    from oracle import svm_oracle as so
    from optiml_amd.datasets import make_blobs, make_regression
    info = _cpu_info()
    if args.solver == 'adagrad':
        return cpu_baseline_al(args, info)
    if args.solver == 'ascg':
        return cpu_baseline_as(args, info)
    sizes = sorted({min(int(s), args.n) for s in args.cpu_sizes.split(',') if s})
    solve = bo.projected_gradient if args.solver == 'pg' else bo.frank_wolfe
    samples = []
    left_out = []
    try:
        free = next(int(line.split()[1]) * 1024 for line in open('/proc/meminfo') if line.startswith('MemAvailable'))
    except Exception:  # noqa: BLE001
        free = 32 << 30
    dual = 1 if args.task == 'svc' else 2
    for ns in sizes:
        if 3 * (dual * ns) ** 2 * 8 > free // 2 and len(samples) >= 2:   # Q, K and one temporary during assembly
            left_out.append(ns)
            continue
        X, y = make_blobs(ns, args.d, seed=0) if args.task == 'svc' else make_regression(ns, args.d, seed=0)
        t0 = time.perf_counter()
        K = so.gram(args.kernel, X, None, 'scale', 1.0 if args.kernel == 'poly' else 0.0, 3)
        Q, q, ub = so.svc_dual(K, y, 1.0) if args.task == 'svc' else so.svr_dual(K, y, 1.0, 0.1)
        del K
        t_build = time.perf_counter() - t0
        solve(Q, q, ub, max_iter=2)  # warm
        its, dt = 0, 0.0
        chunk = 4
        while dt < args.cpu_seconds and its < 400:
            t0 = time.perf_counter()
            res = solve(Q, q, ub, max_iter=chunk)
            dt += time.perf_counter() - t0
            its += res['iter']
        # each call restarts from the mid-box point: per-iteration cost is what is timed (3 dense products), not progress
        samples.append({'n': ns, 'iters': int(its), 's_per_iter': dt / its, 'build_s': t_build,
                        'GBs_of_Q': 3 * (len(q) ** 2) * 8 / (dt / its) / 1e9})
        del Q
    last = samples[-1]
    n2 = 1.0 / (last['s_per_iter'] * (args.n / last['n']) ** 2)
    if len(samples) >= 2 and samples[0]['n'] != last['n']:
        fit = samples[-2:]   # the two largest sizes: nearest the asymptote
        expo = float(np.polyfit(np.log([s['n'] for s in fit]), np.log([s['s_per_iter'] for s in fit]), 1)[0])
    else:
        expo = 2.0
    fitted = 1.0 / (last['s_per_iter'] * (args.n / last['n']) ** expo) if args.n > last['n'] else 1.0 / last['s_per_iter']
    desc = ', '.join(f"n={s['n']}: {s['iters']} it, {1e3 * s['s_per_iter']:.1f} ms/it ({s['GBs_of_Q']:.0f} GB/s of Q)" for s in samples)
    out = {'value': n2, 'unit': 'iter/s', 'kind': 'port', 'kind_detail': 'extrapolated from the sizes sampled', 'extrapolated': args.n > last['n'],
           'law': 'n^2 (three dense n x n products per iteration)', 'fitted_exponent': expo, 'value_fitted_exponent': fitted,
           # the honest statement is the interval: the n^2 law is the asymptote of a bandwidth-bound dense product, the fitted exponent
           # is what THIS host showed between the samples (below 2 while the BLAS threads are still ramping up)
           'value_range': sorted([n2, fitted]),
           'sample': f'oracle {args.solver.upper()} (reference formulation: dense fp64 Q on host, 3 products/iter), d={args.d}: {desc}; '
                     f'value = rate at n={last["n"]} scaled to n={args.n} by (n_s/n)^2, the law of a bandwidth-bound dense product '
                     f'(value_fitted_exponent: with the exponent {expo:.2f} fitted between the two largest samples — on a shared '
                     f'host it moves from run to run; value_range = [both]); Gram+Q assembly excluded',
           'samples': samples, 'sizes_left_out_for_memory': left_out}
    if args.cpu_stream_iters > 0 and args.task == 'svc' and args.kernel == 'rbf':
        out['streamed_product_full_n'] = cpu_streamed_product(args.n, args.d, args.cpu_stream_iters)
    out.update(info)
    return out


def cpu_streamed_product(n, d, iters, blk=2000):
    """SURVEY 8(d)'s measured bound that needs no extrapolation: ONE product Q v at the FULL n on the host cores with K never
    materialised — row blocks of the RBF Gram matrix formed (BLAS GEMM + exp) and applied at once.  The cheapest thing a CPU
    can do per iteration (the reference formulation does three products on a resident Q it could not hold)."""
    from oracle import svm_oracle as so
    from optiml_amd.datasets import make_blobs
    X, y = make_blobs(n, d, seed=0)
    gamma = so.resolve_gamma('scale', X)
    sq = np.einsum('ij,ij->i', X, X)
    v = np.random.RandomState(0).uniform(size=n) * y
    times = []
    for _ in range(iters):
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
        print(f'[bench] CPU streamed product n={n}: {times[-1]:.2f} s', file=sys.stderr, flush=True)
    return {'n': n, 'block_rows': blk, 's_per_product': times, 'iter_per_s_at_one_product_per_iteration': 1.0 / min(times),
            'kind': 'measured at full n (not the reference formulation: K recomputed per product, one product per iteration)'}


def cpu_baseline_as(args, info):
    """Config 5 on the CPU: the reference ActiveSet (dense Cholesky of Q[A,A] in every iteration) on the squared-hinge dual at
    a bounded n, scaled by the n^3 law of the factorisation to the workload's n — labelled, never a raw time."""
    from oracle import bcqp_oracle as bo, svm_oracle as so
    from optiml_amd.datasets import make_blobs
    ns = min(args.n, 4000)
    X, y = make_blobs(ns, args.d, seed=0)
    K = so.gram('rbf', X)
    Q = K * np.outer(y, y) + np.outer(y, y) + np.eye(ns) / 2.0
    t0 = time.perf_counter()
    res = bo.active_set(Q, -np.ones(ns), np.full(ns, np.inf), x0=np.ones(ns), max_iter=12)
    dt = (time.perf_counter() - t0) / max(res['iter'], 1)
    out = {'value': 1.0 / (dt * (args.n / ns) ** 3), 'unit': 'iter/s', 'kind': 'port', 'kind_detail': 'extrapolated from the sizes sampled', 'extrapolated': True,
           'sample': f'oracle ActiveSet (reference algorithm: cho_factor of Q[A,A] per iteration, dense fp64 Q on host) on the '
                     f'squared-hinge dual at n={ns} d={args.d}: {res["iter"]} iterations, {dt:.3f} s/it; value scaled by (n_s/n)^3 to n={args.n}',
           'measured_iter_per_s_at_sample': 1.0 / dt, 'sample_n': ns}
    out.update(info)
    return out


def cpu_baseline_al(args, info):
    """Oracle AdaGrad on the augmented Lagrangian in the reference formulation: dense Q and the dense stacked
    constraint matrix [a; -I; I] ((2N+1) x N), three products with Q per iteration."""
    from oracle import al_oracle as ao, svm_oracle as so
    from optiml_amd.datasets import make_blobs, make_regression
    ns = min(args.n, 4000)
    X, y = make_blobs(ns, args.d, seed=0) if args.task == 'svc' else make_regression(ns, args.d, seed=0)
    K = so.gram(args.kernel, X, None, 'scale', 1.0 if args.kernel == 'poly' else 0.0, 3)
    if args.task == 'svc':
        Q, q, a = K * np.outer(y, y), -np.ones(ns), y
    else:
        Q, q = np.vstack((np.hstack((K, -K)), np.hstack((-K, K)))), np.hstack((-y, y)) + 0.1
        a = np.hstack((np.ones(ns), -np.ones(ns)))
    N = len(q)
    al = ao.AugLag(Q, q, a=a, lb=np.zeros(N), ub=np.ones(N), rho=1.)
    t0 = time.perf_counter()
    res = ao.minimize(al, np.random.RandomState(0).uniform(size=N), 'adagrad', epochs=31, step_size=1.)
    dt = time.perf_counter() - t0
    rate = res['iter'] / dt
    out = {'value': rate * (ns / args.n) ** 2, 'unit': 'iter/s', 'kind': 'port', 'kind_detail': 'extrapolated from the sizes sampled', 'extrapolated': True,
           'sample': f'oracle AdaGrad on the augmented Lagrangian (dense fp64 Q and dense [a;-I;I] on host), n={ns} '
                     f'd={args.d}, {res["iter"]} iterations in {dt:.2f}s = {rate:.3f} iter/s measured; value scaled '
                     f'by (n_s/n)^2 to n={args.n}',
           'measured_iter_per_s_at_sample': rate, 'sample_n': ns}
    out.update(info)
    return out


def cpu_study(args):
    """SURVEY 8(d), 'CPU baseline timed beside it': the reference formulation cannot be held on the host at the headline
    size, so (1) time the oracle's PG at three sizes that fit and check the n^2 law the extrapolation rests on, (2) the
    same for the dense Cholesky InteriorPoint / ActiveSet spend their time in (n^3), and (3) time a blocked,
    Gram-streaming product (K never materialised, ONE product per iteration — the cheapest thing a CPU can do) at the
    full n for a few iterations: a measured bound on the CPU rate that needs no extrapolation."""
    import scipy.linalg as sla
    from oracle import bcqp_oracle as bo
    from oracle import svm_oracle as so
    from optiml_amd.datasets import make_blobs
    try:
        from threadpoolctl import threadpool_info
        pools = [{k: p.get(k) for k in ('internal_api', 'num_threads', 'version')} for p in threadpool_info()]
    except Exception:
        pools = []
    out = {'what': 'cpu_study', 'sched_affinity': len(os.sched_getaffinity(0)), 'blas_pools': pools, 'd': args.d}
    out.update(_cpu_info())
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


# ---------------------------------------------------------------------------------------------------------------------
# time-to-KKT-tol records (BASELINE.json's second metric): the routes that reach a stop test
# ---------------------------------------------------------------------------------------------------------------------
def kkt_smo(n, d, sigma, X=None, y=None, cpu=True, cpu_n=12000):
    """SVC.fit(optimizer='smo') end to end (Gram build + sweeps, tol 1e-3) on one GPU; CPU: the oracle's SMO sweeps at a
    bounded n, reported AT that n and scaled (pair steps grow ~linearly with n and each costs O(n): n^2), labelled."""
    from optiml_amd import device
    from optiml_amd.datasets import make_blobs
    from optiml_amd.ml.svm import SVC
    from optiml_amd.ml.svm.kernels import gaussian
    from optiml_amd.ml.svm.losses import hinge
    if X is None:
        X, y = make_blobs(n, d, seed=0, sigma=sigma)
    t0 = time.perf_counter()
    est = SVC(loss=hinge, kernel=gaussian, C=1., dual=True, optimizer='smo', tol=1e-3).fit(X, y)
    dt = time.perf_counter() - t0
    rec = {'route': "SVC.fit(optimizer='smo')", 'n': n, 'd': d, 'tol': 1e-3, 'value': dt, 'unit': 's',
           'outer_iterations': int(est.optimizer.iter), 'pair_steps': int(est.optimizer.steps), 'n_sv': int(len(est.support_)),
           'includes': 'Gram build + sweeps + intercept', 'roofline': None,
           'roofline_note': 'sequential pair steps: latency-bound (one walking workgroup + helper workgroups), no roofline applies'}
    est.obj.release()
    # `value` is the first fit of a fresh process (it includes the first large device allocation); the same fit again
    t0 = time.perf_counter()
    est = SVC(loss=hinge, kernel=gaussian, C=1., dual=True, optimizer='smo', tol=1e-3).fit(X, y)
    rec['second_fit_s'] = time.perf_counter() - t0
    est.obj.release()
    if cpu:
        from oracle import smo_oracle as smo, svm_oracle as so
        samples = []
        for ns in sorted({min(cpu_n // 2, n), min(cpu_n, n)}):
            K = so.gram('rbf', X[:ns])
            yb = np.where(y[:ns] == np.unique(y)[-1], 1., -1.)
            t0 = time.perf_counter()
            r = smo.smo_svc(K, yb, 1., 1e-3)
            samples.append({'n': ns, 's': time.perf_counter() - t0, 'outer_iterations': int(r['iter']), 'pair_steps': int(r['steps'])})
        last = samples[-1]
        # NOT extrapolated: between the two samples the oracle's time grows with an exponent of ~6 (outer sweeps and
        # pair steps do not grow smoothly with n), so no power law carries the larger sample to the workload's n.  What is reported is
        # what was measured: the two samples, and `value` = the larger one AT ITS OWN n.
        rec['cpu_baseline'] = {'value': last['s'], 'unit': 's', 'n': last['n'], 'kind': 'port', 'kind_detail': f'measured at n={last["n"]}, NOT at the workload\'s n={n}',
                               'extrapolated': False, 'comparable_to_value': last['n'] == n,
                               'cores': 1, 'cores_note': 'SMO is a sequential chain of pair steps: one core is the algorithm\'s nature, not a choice',
                               'samples': samples,
                               'sample': 'oracle SMO sweeps (reference algorithm in NumPy, dense K on host, Gram build excluded) at '
                                         + ', '.join(f"n={v['n']}: {v['pair_steps']} pair steps in {v['s']:.2f} s" for v in samples)
                                         + '; does not follow a power law between the samples, so it is not scaled to the workload\'s n'}
    return rec


def kkt_box(solver, n, d, sigma, cpu=True, wall_limit=0.0):
    """SVC.fit(optimizer=InteriorPoint | ActiveSet) end to end on one GPU, to the solver's own stop test.  roofline: fp64 MFMA
    fraction of the Cholesky factorisations (n^3/3 flop each, HIP-event time of the factorisation kernels).  CPU: the oracle's
    per-iteration cost at a bounded n scaled by the n^3 law to THIS n, times the iteration count — labelled."""
    from optiml_amd import _lib, device
    from optiml_amd.datasets import make_blobs
    from optiml_amd.ml.svm import SVC
    from optiml_amd.ml.svm.kernels import gaussian
    from optiml_amd.ml.svm.losses import hinge
    from optiml_amd.opti.constrained import ActiveSet, InteriorPoint
    ctx = device.get_context()
    X, y = make_blobs(n, d, seed=0, sigma=sigma)
    cls = InteriorPoint if solver == 'ip' else ActiveSet
    if wall_limit > 0:
        base = cls

        class cls(base):   # the same solver; its callback also looks at the clock every 256th iteration
            def callback(self, args=()):
                base.callback(self, args)
                if not self.iter % 256:
                    if not self.iter % 4096:
                        print(f'[bench] {base.__name__} n={n}: iteration {self.iter}, f = {self.f_x:.8g}, {time.perf_counter() - t0:.0f} s',
                              file=sys.stderr, flush=True)
                    if time.perf_counter() - t0 > wall_limit:
                        raise StopIteration
        cls.__name__ = base.__name__
    ctx.profile(True)
    ctx.profile_read(_lib.PROF_CHOL, reset=True)
    import threading
    stop = threading.Event()

    def heartbeat():   # long fits (n=100 000 InteriorPoint: 6 min) must not look hung to whoever watches stderr
        while not stop.wait(60.0):
            print(f'[bench] {cls.__name__} n={n}: {time.perf_counter() - t0:.0f} s', file=sys.stderr, flush=True)
    t0 = time.perf_counter()
    hb = threading.Thread(target=heartbeat, daemon=True)
    hb.start()
    try:
        est = SVC(loss=hinge, kernel=gaussian, C=1., reg_intercept=True, dual=True, optimizer=cls, max_iter=10 ** 7).fit(X, y)
    finally:
        dt = time.perf_counter() - t0
        stop.set()
    o = est.optimizer
    ch_ms, ch_cnt = ctx.profile_read(_lib.PROF_CHOL, reset=True)
    rec = {'route': f'SVC.fit(optimizer={cls.__name__})', 'n': n, 'd': d, 'value': dt, 'unit': 's', 'iterations': int(o.iter),
           'status': o.status, 'f': float(o.f_x), 'n_sv': int(len(est.support_)), 's_per_iteration': dt / max(o.iter, 1),
           'stop_test': 'relative gap <= 1e-10' if solver == 'ip' else 'no wrong-sign multiplier (exact)',
           'includes': 'Gram build + iterations + intercept'}
    if wall_limit > 0:
        rec['wall_limit_s'] = wall_limit
        rec['cut_at_wall_limit'] = o.status == 'unknown'
    if solver == 'as':
        rec['bounds_at_the_end'] = {'at_lower': int(o.L.sum()), 'at_upper': int(o.U.sum()), 'free': int(n - o.L.sum() - o.U.sum())}
        rec['counters'] = {'minres_iterations': int(o.minres_iterations), 'product_free_iterations': int(o.product_free_iterations),
                           'base_factorisations': int(ch_cnt)}
    if solver == 'ip' and ch_cnt:
        flops = n ** 3 / 3.0
        tf = flops / (ch_ms / ch_cnt * 1e-3) / 1e12
        try:   # what back-to-back fp64 MFMAs on register operands sustain on THIS GPU (outside the timed region)
            probe = ctx.probe_mfma_f64(1.0)
        except Exception:  # noqa: BLE001
            probe = None
        rec['roofline'] = {'bound': 'mfma', 'kernel': 'blocked Cholesky of H = Q + diag (syrk/trsm on v_mfma_f64_16x16x4_f64)',
                           'achieved': tf, 'peak': FP64_MFMA_PEAK_TF, 'unit': 'TFLOP/s', 'frac': tf / FP64_MFMA_PEAK_TF,
                           'traffic': None, 'flops_per_launch': flops, 'avg_factor_ms': ch_ms / ch_cnt, 'factorisations': ch_cnt,
                           'factor_share_of_wall': ch_ms * 1e-3 / dt, 'measured_mfma_f64_TFs': probe,
                           'frac_of_measured_mfma': (tf / probe) if probe else None}
    elif ch_cnt:
        rec['roofline'] = {'bound': 'mfma', 'kernel': 'base-set Cholesky factorisations (kept across iterations, Schur updates between)',
                           'achieved': None, 'peak': FP64_MFMA_PEAK_TF, 'unit': 'TFLOP/s', 'frac': None, 'traffic': None,
                           'avg_factor_ms': ch_ms / ch_cnt, 'factorisations': ch_cnt, 'factor_share_of_wall': ch_ms * 1e-3 / dt,
                           'note': 'the free set differs per factorisation, so n_A^3/3 is not a fixed flop count'}
    else:
        rec['roofline'] = None
    est.obj.release()
    if cpu:
        from oracle import bcqp_oracle as bo, svm_oracle as so
        fn = bo.interior_point if solver == 'ip' else bo.active_set
        # InteriorPoint: a third, larger sample (one iteration: ~10 s) so that the exponent the extrapolation uses is fitted between
        # n = 12 000 and 24 000, where a threaded dpotrf is much nearer its n^3 asymptote than between 6 000 and 12 000 (r03: 1.99)
        sizes = ((6000, 3), (12000, 3), (24000, 1)) if solver == 'ip' else ((2000, 12), (4000, 12))
        samples = []
        for ns, k in sorted({(min(v, n), it) for v, it in sizes}):
            Q, q, ub = so.svc_dual(so.gram('rbf', X[:ns]), y[:ns], 1.0)
            t0 = time.perf_counter()
            r = fn(Q, q, ub, max_iter=k)
            samples.append({'n': ns, 's_per_iteration': (time.perf_counter() - t0) / max(r['iter'], 1), 'iterations': int(r['iter'])})
            del Q
        last = samples[-1]
        fit = samples[-2:]   # the two largest sizes
        expo = float(np.polyfit(np.log([v['n'] for v in fit]), np.log([v['s_per_iteration'] for v in fit]), 1)[0]) \
            if len(fit) > 1 and fit[0]['n'] != fit[1]['n'] else 3.0
        info = _cpu_info()
        rec['cpu_baseline'] = {'value': last['s_per_iteration'] * (n / last['n']) ** expo * o.iter, 'unit': 's',
                               'kind': 'port', 'kind_detail': 'extrapolated from the sizes sampled', 'extrapolated': n > last['n'], 'cores': info['cores'],
                               'cores_source': info['cores_source'], 'fitted_exponent': expo,
                               'law': 'exponent fitted between the two LARGEST samples (a threaded dpotrf approaches its n^3 asymptote slowly)',
                               'value_n3_law': last['s_per_iteration'] * (n / last['n']) ** 3 * o.iter,
                               'measured_s_per_iteration_at_sample': last['s_per_iteration'], 'sample_n': last['n'], 'samples': samples,
                               'sample': f'oracle {solver.upper()} (reference algorithm in NumPy/SciPy: cho_factor per iteration, dense Q on host): '
                                         + ', '.join(f"n={v['n']}: {v['s_per_iteration']:.3f} s/iteration" for v in samples)
                                         + f'; value = the larger sample x (n/n_s)^{expo:.2f} (fitted) x the {o.iter} iterations the device run '
                                           f'needed at n={n} (value_n3_law: with exponent 3); Gram and Q assembly excluded'}
    return rec


def bench_kkt_line(args):
    """--solver smo | ip | as: one JSON line with metric time_to_kkt_tol."""
    from optiml_amd import device
    ctx = device.get_context()
    if args.solver == 'smo':
        rec = kkt_smo(args.n, args.d, args.sigma, cpu=not args.no_cpu)
    else:
        rec = kkt_box(args.solver, args.n, args.d, args.sigma, cpu=not args.no_cpu, wall_limit=args.wall_limit)
    steps = rec.get('iterations', rec.get('outer_iterations', 0))
    out = {'metric': 'time_to_kkt_tol', 'value': rec['value'], 'unit': 's', 'n_gpus': 1, 'steps': int(steps), 'warmup': 0,
           'ms_per_step': 1e3 * rec['value'] / max(steps, 1), 'higher_is_better': False, 'scaling': 'strong', 'vs_baseline': None,
           'dtype': 'f64', 'data': 'synthetic',
           'config': {'workload': f'svc_hinge_rbf_{args.solver}_dual_n{args.n}_d{args.d}', 'n': args.n, 'd': args.d, 'C': 1.0,
                      'gamma': 'scale', 'solver': args.solver, 'device': ctx.name},
           'roofline': rec.pop('roofline', None), 'cpu_baseline': rec.pop('cpu_baseline', None)}
    out.update({k: v for k, v in rec.items() if k not in ('value', 'unit', 'n', 'd')})
    print(json.dumps(out), flush=True)


# ---------------------------------------------------------------------------------------------------------------------
# what the 1 -> 8 GPU curve will look like, measured on ONE GPU: every rank's share of the partition, one at a time
# ---------------------------------------------------------------------------------------------------------------------
def share_timing(args):
    """For every G in --emulate-shares and every rank k < G: a share context (rank k of G, no transport), the rank's part of
    the Gram panel, and --steps solver iterations of exactly the kernels that rank runs in a G-GPU job (tile product over its
    canonical segments, per-segment reduction, ordered segment sum, the fused O(n) kernels) — everything but the collective,
    which moves nothing here.  The iterates are meaningless (partial products); the TIMES are the rank's.  Prediction:
    G-GPU iteration time = the slowest share + the assumed collective (--exchange-us), stated with and without it."""
    from optiml_amd import _lib, device
    from optiml_amd.datasets import make_blobs, make_regression
    from optiml_amd.ml.svm.kernels import PolyKernel, gaussian, linear
    from optiml_amd.opti import KernelQuadratic
    from optiml_amd.opti.constrained._base import _DeviceSolver
    n, d = args.n, args.d
    kern = {'rbf': gaussian, 'poly': PolyKernel(3, 'scale', 1.0), 'linear': linear}[args.kernel]
    if args.task == 'svc':
        X, y = make_blobs(n, d, seed=0, sigma=args.sigma)
        q, yy, struct = -np.ones(n), y, 'svc'
    else:
        X, y = make_regression(n, d, seed=0)
        q, yy, struct = np.hstack((-y, y)) + 0.1, None, 'svr'
    N = len(q)
    esz = 8 if args.storage == 'f64' else 4
    kind = _lib.FW if args.solver == 'fw' else _lib.PG
    T = 256
    out = {'what': 'share_timing', 'metric': 'dual_qp_iterations_per_sec (predicted from single-GPU share timings)',
           'config': {'workload': f'{args.task}_{args.kernel}_{args.solver}_dual_n{n}_d{d}', 'n': n, 'd': d, 'storage': args.storage,
                      'steps': args.steps, 'warmup': args.warmup},
           'assumed_exchange_us': args.exchange_us, 'partitions': []}
    name = None
    for G in [int(g) for g in args.emulate_shares.split(',') if g]:
        shares = []
        for k in range(G):
            ctx = device.Context(share=(k, G)) if G > 1 else device.Context()
            name = name or ctx.name
            quad = KernelQuadratic(X, q, struct, kern, y=yy, storage=args.storage, tune_placement=not args.no_placement, expected_products=STEADY_STATE_PRODUCTS)
            ctx.profile(True)
            dev = quad.device_problem(ctx)
            gram_ms, _ = ctx.profile_read(_lib.PROF_GRAM, reset=True)
            _, _, r0, r1 = dev.dims()
            # eps = -1: no stop test can fire on the meaningless iterates of a partial product (FW's gap did, on C4's shares)
            solver = _DeviceSolver(dev, kind, np.zeros(N), np.ones(N), np.ones(N) / 2, -1.0, 10 ** 9)
            solver.run(max(args.warmup, 1))
            ctx.profile_read(_lib.PROF_MATVEC, reset=True)
            t0 = time.perf_counter()
            rows, _ = solver.run(args.steps)
            dt = time.perf_counter() - t0
            mv_ms, mv_cnt = ctx.profile_read(_lib.PROF_MATVEC, reset=True)
            i0, i1 = r0 // T, -(-r1 // T)
            tiles = i1 * (i1 + 1) // 2 - i0 * (i0 + 1) // 2
            alg = tiles * (T * T * esz + T * 8) + (tiles // 8 + i1 - i0) * T * 8 + 2 * n * 8
            avg = mv_ms / max(mv_cnt, 1)
            step_ms = 1e3 * dt / max(len(rows), 1)
            shares.append({'rank': k, 'tile_rows': [i0, i1], 'rows': r1 - r0, 'tiles': tiles, 'strips': sum(I // 8 + 1 for I in range(i0, i1)),
                           'panel_GB': tiles * T * T * esz / 1e9, 'gram_build_ms': gram_ms, 'panel_placement_ms': dev.placement(), 'steps_done': len(rows),
                           'symv_tiles_ms': avg, 'symv_GBs': alg / (avg * 1e-3) / 1e9 if avg > 0 else 0.0,
                           'symv_frac_of_8TBs': alg / (avg * 1e-3) / 1e9 / HBM_PEAK_GBS if avg > 0 else 0.0,
                           'ms_per_step': step_ms, 'fixed_cost_ms': step_ms - avg})
            print(f'[share] G={G} k={k}: tiles {tiles}, symv {avg:.3f} ms ({shares[-1]["symv_GBs"]:.0f} GB/s), step {step_ms:.3f} ms',
                  file=sys.stderr, flush=True)
            solver.close()
            quad.release()
            ctx.close()
        worst = max(s['ms_per_step'] for s in shares)
        out['partitions'].append({
            'G': G, 'shares': shares, 'slowest_share_ms_per_step': worst,
            'predicted_iter_per_s_no_exchange': 1e3 / worst,
            'predicted_iter_per_s': 1e3 / (worst + (args.exchange_us * 1e-3 if G > 1 else 0.0)),
            'min_symv_frac_of_8TBs': min(s['symv_frac_of_8TBs'] for s in shares)})
    if args.probe_gib:
        ctx = device.Context()
        out['probe'] = []
        for gib in [float(x) for x in args.probe_gib.split(',') if x]:
            r, c = ctx.probe_bandwidth(int(gib * (1 << 30)), 5)
            out['probe'].append({'GiB': gib, 'read_GBs': r, 'copy_GBs': c})
            print(f'[share] probe {gib} GiB: read {r:.0f} GB/s, copy {c:.0f} GB/s', file=sys.stderr, flush=True)
        ctx.close()
    base = next((p for p in out['partitions'] if p['G'] == 1), None)
    if base:
        for p in out['partitions']:
            p['predicted_speedup_vs_1'] = p['predicted_iter_per_s'] / base['predicted_iter_per_s']
    out['config']['device'] = name
    print(json.dumps(out), flush=True)



# ---------------------------------------------------------------------------------------------------------------------
# dense Quadratic(Q, q): packed lower triangle (Q == Q' exactly) against row blocks
# ---------------------------------------------------------------------------------------------------------------------
DENSE_BANDS = ((0, 4.0), (1, -1.0), (257, 0.5), (4099, 0.25))   # offset, value: symmetric, strictly diagonally dominant


def dense_band_matrix(n):
    """n x n fp64 host matrix with a few symmetric bands.  np.zeros maps its pages lazily, so only the pages a band touches (and
    what the upload reads) ever become resident: the n = 100 000 matrix is 80 GB of address space, not of host memory writes."""
    Q = np.zeros((n, n))
    flat = Q.reshape(-1)
    for off, val in DENSE_BANDS:
        if off < n:
            flat[off:(n - off) * n:n + 1] = val          # (i, i + off)
            flat[off * n::n + 1] = val                    # (i + off, i)
    return Q


def dense_band_product(n, v):
    out = np.zeros(n)
    for off, val in DENSE_BANDS:
        if off == 0:
            out += val * v
        elif off < n:
            out[:n - off] += val * v[off:]
            out[off:] += val * v[:n - off]
    return out


def dense_record(args):
    """--dense: one JSON record with the two layouts of a dense symmetric Hessian side by side."""
    from optiml_amd import _lib, device
    from optiml_amd.opti import Quadratic
    from optiml_amd.opti.constrained._base import _DeviceSolver
    ctx = device.get_context()
    n = args.n
    t0 = time.perf_counter()
    Q = dense_band_matrix(n)
    rs = np.random.RandomState(0)
    q = rs.standard_normal(n)
    v = rs.standard_normal(n)
    ref = dense_band_product(n, v)
    out = {'what': 'dense_quadratic', 'metric': 'dual_qp_iterations_per_sec', 'unit': 'iter/s', 'n': n, 'dtype': 'f64', 'device': ctx.name,
           'host_matrix_s': time.perf_counter() - t0,
           'hessian': 'banded symmetric (offsets %s), handed over as a full n x n row-major fp64 host array' % ', '.join(str(o) for o, _ in DENSE_BANDS),
           'reference': 'Quadratic(Q, q): optiml/opti/_base.py:228-300 (Q @ x at :282, :291)', 'layouts': {}}
    T = 256
    nb = -(-n // T)
    tiles = nb * (nb + 1) // 2
    ld = -(-n // 1024) * 1024
    for name, sym, place in (('packed', None, not args.no_placement), ('rows', False, False)):
        quad = Quadratic(Q, q, symmetric=sym, tune_placement=place, expected_products=STEADY_STATE_PRODUCTS)
        ctx.profile(True)
        t0 = time.perf_counter()
        dev = quad.device_problem(ctx)
        setup = time.perf_counter() - t0
        lay = dev.layout()
        assert lay['packed'] == (name == 'packed'), lay
        err = float(np.abs(dev.matvec(v) - ref).max())
        solver = _DeviceSolver(dev, _lib.PG, np.zeros(n), np.ones(n), np.ones(n) / 2, -1.0, 10 ** 9)
        solver.run(max(args.warmup, 1))
        ctx.profile_read(_lib.PROF_MATVEC, reset=True)
        t0 = time.perf_counter()
        rows, _ = solver.run(args.steps)
        dt = time.perf_counter() - t0
        mv_ms, mv_cnt = ctx.profile_read(_lib.PROF_MATVEC, reset=True)
        avg = mv_ms / max(mv_cnt, 1)
        if name == 'packed':
            alg = tiles * (T * T * 8 + T * 8) + (tiles // 8 + nb) * T * 8 + 2 * n * 8
            kern = 'symv_tiles_kernel<double, false, 8, SR>'
        else:
            alg = n * ld * 8 + 3 * n * 8
            kern = 'gemv_rows_kernel'
        gbs = alg / (avg * 1e-3) / 1e9 if avg > 0 else 0.0
        out['layouts'][name] = {
            'value': len(rows) / dt, 'ms_per_step': 1e3 * dt / max(len(rows), 1), 'steps': len(rows), 'upload_s': setup,
            'panel_GB': lay['panel_bytes'] / 1e9, 'panel_placement_ms': dev.placement(), 'product_max_abs_error_vs_closed_form': err,
            'roofline': {'bound': 'hbm', 'kernel': kern, 'achieved': gbs, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': gbs / HBM_PEAK_GBS,
                         'traffic': None, 'avg_launch_ms': avg, 'launches': mv_cnt, 'algorithmic_bytes_per_launch': alg}}
        print(f'[dense] n={n} {name}: product {avg:.3f} ms ({gbs:.0f} GB/s), {len(rows) / dt:.1f} iter/s, upload {setup:.1f} s, '
              f'|error| {err:.2e}', file=sys.stderr, flush=True)
        solver.close()
        quad.release()
        ctx.release_held_memory()
    pk, rw = out['layouts']['packed'], out['layouts']['rows']
    out['value'] = pk['value']
    out['packed_vs_rows_product_time'] = pk['roofline']['avg_launch_ms'] / rw['roofline']['avg_launch_ms'] if rw['roofline']['avg_launch_ms'] else None
    print(json.dumps(out), flush=True)


# ---------------------------------------------------------------------------------------------------------------------
# the default line: headline record + side records, each measured in a fresh child process (this parent never touches HIP)
# ---------------------------------------------------------------------------------------------------------------------
SIDE_COMMANDS = {   # name -> (arguments of the child, its time cap in seconds)
    'c2': (['--config', 'c2', '--steps', '400', '--warmup', '20', '--no-cpu', '--kkt', 'none'], 90.0),
    'c4': (['--config', 'c4', '--steps', '50', '--warmup', '5', '--no-cpu', '--kkt', 'none'], 90.0),
    'c5': (['--config', 'c5', '--steps', '10', '--warmup', '2', '--no-cpu', '--kkt', 'none'], 150.0),
    'shares': (['--emulate-shares', '2,4,8', '--steps', '30', '--warmup', '3'], 150.0),
    'collective': (['--collective-floor'], 150.0),   # (a cold box pages librccl.so in: seen to take > 90 s once)
    'shares_c4': (['--config', 'c4', '--emulate-shares', '4', '--steps', '30', '--warmup', '3'], 90.0),
    'shares_c5': (['--config', 'c5', '--solver', 'pg', '--emulate-shares', '1,8', '--steps', '20', '--warmup', '3'], 150.0),
    'fixed_cap': (['--fixed-cap', '1000', '--fixed-cap-configs', 'headline,c2,c4'], 90.0),
    # the reference's ActiveSet needs ~n outer iterations at config 5 (hours): the driver line holds 100 of them, profiles/r04/fixed_cap_c5.json 1 000
    'fixed_cap_c5': (['--fixed-cap', '100', '--fixed-cap-configs', 'c5'], 90.0),
    # dense ActiveSet (the reference's own class, active_set.py:82-237) to 'optimal' on config 2's shape: lands in time_to_kkt
    'as_c2': (['--solver', 'as', '--samples', '20000', '--features', '64', '--no-cpu'], 90.0),
    # dense Quadratic(Q, q) of config 2's size: the packed copy a Q == Q' gets against the row blocks any other Q keeps
    'dense_c2': (['--dense', '--samples', '20000', '--steps', '300', '--warmup', '10'], 60.0),
}


def _run_child(argv, timeout):
    """One fresh `python bench.py ... --records none` process; (last JSON object on its stdout | None, error text | None,
    seconds).  Its stderr is this process's stderr (progress stays visible).  On expiry exactly that child is killed."""
    t0 = time.perf_counter()
    proc = subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv + ['--records', 'none', '--line', 'full'], stdout=subprocess.PIPE,
                            text=True, cwd=REPO)
    try:
        out, _ = proc.communicate(timeout=timeout)
    except subprocess.TimeoutExpired:
        proc.kill()
        proc.communicate()
        return None, f'not finished within its {timeout:.0f} s cap (killed)', time.perf_counter() - t0
    dt = time.perf_counter() - t0
    rec = None
    for line in reversed(out.splitlines()):
        line = line.strip()
        if line.startswith('{'):
            try:
                rec = json.loads(line)
                break
            except ValueError:
                continue
    if proc.returncode != 0 or rec is None:
        return rec, f'exit code {proc.returncode}' + ('' if rec is not None else ', no JSON line'), dt
    return rec, None, dt


def _compact_shares(rec):
    """The share table of one --emulate-shares run without the per-share bulk."""
    out = {'workload': rec['config']['workload'], 'steps': rec['config']['steps'], 'assumed_exchange_us': rec['assumed_exchange_us'],
           'assumed_exchange_note': 'the one collective per product cannot be measured on one GPU: ASSUMED, see collective_floor_us '
                                    'for its measured lower bound',
           'partitions': []}
    for part in rec['partitions']:
        sh = part['shares']
        out['partitions'].append({
            'G': part['G'], 'slowest_share_ms_per_step': part['slowest_share_ms_per_step'],
            'slowest_share_symv_tiles_ms': max(v['symv_tiles_ms'] for v in sh),
            'min_symv_frac_of_8TBs': part['min_symv_frac_of_8TBs'],
            'max_fixed_cost_ms': max(v['fixed_cost_ms'] for v in sh),
            'predicted_iter_per_s_no_exchange': part['predicted_iter_per_s_no_exchange'],
            'predicted_iter_per_s': part['predicted_iter_per_s'],
            'per_share': [{'rank': v['rank'], 'tiles': v['tiles'], 'strips': v['strips'], 'symv_tiles_ms': v['symv_tiles_ms'],
                           'symv_frac_of_8TBs': v['symv_frac_of_8TBs'], 'ms_per_step': v['ms_per_step']} for v in sh]})
    return out


def prefetch_rccl():
    """Read librccl.so (573 MB) into the page cache on a background thread — plain file reads, nothing here touches HIP or RCCL.
    dlopen of that library is where a cold box spends the start-up of an RCCL context (bq_comm_init_report: 4.9 s of 5.5 s on a
    fresh box, profiles/r05/rccl_init_stages.txt; one box of round 5 took 126 s for the same record): started when a run begins,
    the read overlaps the work that comes before the first communicator (the headline child; rendezvous and data generation)."""
    import threading

    def _read():
        for path in ('/opt/rocm/lib/librccl.so.1', '/opt/rocm/lib/librccl.so'):
            try:
                with open(os.path.realpath(path), 'rb', buffering=0) as fh:
                    while fh.read(8 << 20):
                        pass
                return
            except OSError:
                continue
    th = threading.Thread(target=_read, daemon=True)
    th.start()
    return th


def orchestrate(args):
    """`python bench.py [--gpus 1 --steps K --warmup W]`: the headline record from a child running exactly this command, then
    the side records (SIDE_RECORDS) from one child each, skipped with a reason when the budget would not hold them, then ONE
    JSON line.  A failing side record never costs the line; a failing headline child is this process's failure."""
    t_start = time.perf_counter()
    prefetch_rccl()
    want = list(SIDE_RECORDS) if args.records == 'all' else [r for r in args.records.split(',') if r]
    bad = [r for r in want if r not in SIDE_COMMANDS]
    if bad:
        raise SystemExit(f'--records: unknown record(s) {bad}; known: {", ".join(SIDE_RECORDS)}')

    def left():
        return args.budget_s - (time.perf_counter() - t_start)

    print('[bench] headline record (child process)', file=sys.stderr, flush=True)
    head, err, dt = _run_child(sys.argv[1:], max(left(), 60.0))
    if head is None or err is not None:
        print(f'[bench] the headline run failed: {err}', file=sys.stderr, flush=True)
        raise SystemExit(1)
    timing = {'headline_s': dt}
    # THE line once already, as soon as the headline record exists: a run that is cut before the side records are in (the caller's
    # limit, a box that pages slowly) still leaves a parseable last line with every contract field; the completed line follows
    early = dict(head)
    early['records'] = {'requested': want, 'state': 'headline record only: this line was printed before the side records were measured',
                        'wall_s': dict(timing, total_s=time.perf_counter() - t_start)}
    emit(early, args)
    side = {}
    for name in want:
        argv, cap = SIDE_COMMANDS[name]
        if left() < 0.5 * cap:   # a record is started only when at least half of its cap is left; it is cut at what IS left
            side[name] = {'skipped': f'{left():.0f} s of the {args.budget_s:.0f} s budget left, cap of this record {cap:.0f} s'}
            continue
        print(f'[bench] side record {name}: bench.py {" ".join(argv)}', file=sys.stderr, flush=True)
        rec, err, dt = _run_child(argv, min(cap, max(left(), 10.0)))
        timing[name + '_s'] = dt
        if err is not None:
            side[name] = {'error': err}
            print(f'[bench] side record {name} failed: {err}', file=sys.stderr, flush=True)
        else:
            rec.pop('cpu_baseline', None)
            rec['command'] = 'bench.py ' + ' '.join(argv)
            side[name] = rec
    configs = {k: side[k] for k in ('c2', 'c4', 'c5') if k in side}
    if configs:
        head['configs'] = configs
    shares = {}
    for key, name in (('headline', 'shares'), ('c4_over_4', 'shares_c4'), ('c5_over_8', 'shares_c5')):
        if name in side:
            shares[key] = _compact_shares(side[name]) if 'partitions' in side[name] else side[name]
    if 'collective' in side:
        head['collective_floor_us'] = side['collective']
    caps = {}
    for name in ('fixed_cap', 'fixed_cap_c5'):
        rec = side.get(name)
        if rec is None:
            continue
        if 'runs' in rec:
            for k, v in rec['runs'].items():
                caps[k] = dict(v, cap=rec['cap'])
            caps.setdefault('residual', rec['residual'])
        else:
            caps[name] = rec
    if caps:
        head['fixed_cap'] = caps
    if isinstance(side.get('c5'), dict) and side['c5'].get('time_to_kkt_projected'):
        head.setdefault('time_to_kkt', {})['c5_projected'] = side['c5']['time_to_kkt_projected']
    if 'dense_c2' in side:
        head['dense_quadratic'] = {'config2_size': side['dense_c2']}
    committed = committed_time_to_kkt()
    if committed:
        head.setdefault('time_to_kkt', {}).update(committed)
    if 'as_c2' in side:
        rec = side['as_c2']
        keep = ('value', 'unit', 'iterations', 'status', 'f', 'n_sv', 's_per_iteration', 'stop_test', 'includes', 'route', 'config', 'roofline',
                'command', 'error', 'skipped')
        head.setdefault('time_to_kkt', {})['as_config2_shape'] = {k: rec[k] for k in keep if k in rec}
    # the G = 1 row of the headline table is the headline record itself; predictions also at the measured collective floor
    floor = side.get('collective', {}).get('headline', {}) if 'collective' in side else {}
    tab = shares.get('headline')
    if tab and 'partitions' in tab:
        base = head['ms_per_step']
        tab['partitions'].insert(0, {'G': 1, 'slowest_share_ms_per_step': base, 'slowest_share_symv_tiles_ms': head['roofline']['avg_launch_ms'],
                                     'min_symv_frac_of_8TBs': head['roofline']['frac'], 'predicted_iter_per_s': head['value'],
                                     'predicted_iter_per_s_no_exchange': head['value'], 'source': 'the headline record of this line'})
        for part in tab['partitions']:
            part['predicted_speedup_vs_1'] = part['predicted_iter_per_s'] / head['value']
            us = floor.get('gather_8_segments', {}).get('mean_us')
            if us is not None and part['G'] > 1:
                part['predicted_iter_per_s_at_collective_floor'] = 1e3 / (part['slowest_share_ms_per_step'] + us * 1e-3)
    for key, cfg in (('c4_over_4', 'c4'), ('c5_over_8', 'c5')):   # the one-GPU record of the same config is the base of these
        tab, one = shares.get(key), side.get(cfg)
        if tab and 'partitions' in tab and one and 'ms_per_step' in one:
            for part in tab['partitions']:
                part['one_gpu_ms_per_step'] = one['ms_per_step']
                if cfg == 'c5':
                    # The shares run PG's iteration: one product (tiles + slab reduction) + a few O(n) kernels = P_G ms on a 1/G share.
                    # An ActiveSetCG outer iteration is `inner_products_per_step` products + work every rank repeats (preconditioner
                    # passes, O(n) kernels of the conjugate gradients, the m x m inverse): R = outer(1 GPU) - products x P_1, measured
                    # on this GPU; predicted outer(G) = products x (P_G + the ASSUMED collective) + R.
                    prods = one.get('inner_products_per_step')
                    g1 = next((q for q in tab['partitions'] if q['G'] == 1), None)
                    if prods and g1 and part['G'] > 1:
                        # the product's own time is taken from the SAME process as the outer iteration it is subtracted from (the
                        # panel's placement moves it by +-3 % from process to process: a difference of two processes' numbers is
                        # noise); the share run contributes only what a product costs beside its tile kernel (slab reduction, O(n))
                        p1 = one['roofline']['avg_launch_ms'] + (g1['slowest_share_ms_per_step'] - g1['slowest_share_symv_tiles_ms'])
                        # ... of which the preconditioner's applications are sharded by samples (round 5: the implicit remainder; round 6:
                        # the explicit model's passes too): 1 / G of their one-GPU time (HIP events around them in the c5 record) + their
                        # collectives (preconditioner_collectives_per_call per application)
                        shard = one.get('preconditioner_sharded_ms_per_step', 0.0)
                        calls = one.get('preconditioner_sharded_calls_per_step', 0.0)
                        repl = one['ms_per_step'] - prods * p1 - shard
                        share_ms = part['slowest_share_ms_per_step'] + tab['assumed_exchange_us'] * 1e-3
                        part['ascg_outer_iteration_ms_predicted'] = prods * share_ms + repl + shard / part['G'] + one.get('preconditioner_collectives_per_call', 2) * calls * tab['assumed_exchange_us'] * 1e-3
                        part['ascg_replicated_ms_per_outer_iteration'] = repl
                        part['ascg_sample_sharded_ms_per_outer_iteration_one_gpu'] = shard
                        part['ascg_one_gpu_product_step_ms'] = p1
                        part['predicted_speedup_vs_1'] = one['ms_per_step'] / part['ascg_outer_iteration_ms_predicted']
                else:
                    part['predicted_speedup_vs_1'] = one['ms_per_step'] / (1e3 / part['predicted_iter_per_s'])
    if shares:
        head['shares'] = shares
    timing['total_s'] = time.perf_counter() - t_start
    head['records'] = {'requested': want, 'budget_s': args.budget_s, 'wall_s': timing,
                       'note': 'headline record first (its own process, unchanged command); every side record in a fresh process '
                               'of its own after it, inputs resident in HBM inside each timed region; this parent never touches HIP'}
    emit(head, args)


def committed_time_to_kkt():
    """time_to_kkt records that take longer than a bench line may (InteriorPoint and dense ActiveSet at the headline size): measured
    by tools/profile_all.sh with this same bench.py, committed as profiles/rNN/time_to_kkt_headline.json and quoted in the complete
    record with their source — never re-measured inside the default line.  {} when no such file is committed."""
    import glob
    out = {}
    for path in sorted(glob.glob(os.path.join(REPO, 'profiles', 'r*', 'time_to_kkt_headline.json'))):
        try:
            rec = json.load(open(path))
        except Exception:  # noqa: BLE001
            continue
        for key in ('ip_headline', 'as_headline'):
            if isinstance(rec.get(key), dict):
                out[key] = dict(rec[key], source=os.path.relpath(path, REPO), measured='in a run of its own, committed file (not in this run)')
    return out


def fixed_cap(args):
    """--fixed-cap CAP: SURVEY 8(d)'s report for the solvers that do not reach their own tolerance at these sizes (in the reference
    either): each workload run for CAP iterations from the reference's start point, then, computed identically for all of them from
    the final point: f = 1/2 x'Qx + q'x and g = Qx + q by ONE fresh product (bq_problem_eval), the projected-gradient norm |d|_2 with
    d = -g, d_i = 0 where x_i sits on ub and d_i > 0 or on lb and d_i < 0 (projected_gradient.py:99-104, tolerance 1e-12)."""
    from optiml_amd import _lib, device
    from optiml_amd.datasets import make_blobs, make_regression
    from optiml_amd.ml.svm.kernels import PolyKernel, gaussian, linear
    from optiml_amd.opti import KernelQuadratic
    from optiml_amd.opti.constrained._base import _DeviceSolver
    ctx = device.get_context()
    names = [c for c in (args.fixed_cap_configs or (args.config or 'headline')).split(',') if c]
    out = {'what': 'fixed_cap', 'cap': args.fixed_cap, 'device': ctx.name,
           'residual': '|d|_2, d = -g masked as ProjectedGradient masks it (active bounds, 1e-12); g = Qx + q from one fresh product at the final point',
           'runs': {}}
    for name in names:
        cfg = CONFIGS[name]
        n, d = cfg['n'], cfg['d']
        kern = {'rbf': gaussian, 'poly': PolyKernel(3, 'scale', 1.0), 'linear': linear}[cfg['kernel']]
        ascg = cfg['solver'] == 'ascg'
        if cfg['solver'] not in ('pg', 'fw', 'ascg'):
            out['runs'][name] = {'skipped': f"{cfg['solver']} reaches its own stop test: see time_to_kkt"}
            continue
        if cfg['task'] == 'svc':
            X, y = make_blobs(n, d, seed=0, sigma=args.sigma)
            quad = KernelQuadratic(X, -np.ones(n), 'svc', kern, y=y, storage=cfg['storage'], diag=0.5 if ascg else 0.0)
        else:
            X, y = make_regression(n, d, seed=0)
            quad = KernelQuadratic(X, np.hstack((-y, y)) + 0.1, 'svr', kern, storage=cfg['storage'])
        N = quad.ndim
        t0 = time.perf_counter()
        dev = quad.device_problem(ctx)
        t_setup = time.perf_counter() - t0
        lb = np.zeros(N)
        if ascg:   # squared-hinge dual: ub = +inf, x0 = 1
            ub, x0 = np.full(N, np.inf), np.ones(N)
            solver = _DeviceSolver(dev, _lib.AS_CG, lb, ub, x0, 1e-6, args.fixed_cap)
            solver.set_inner(args.inner_tol, 0)
        else:
            ub = np.ones(N)
            solver = _DeviceSolver(dev, _lib.PG if cfg['solver'] == 'pg' else _lib.FW, lb, ub, ub / 2, 1e-6, args.fixed_cap)
        t0 = time.perf_counter()
        done, status = 0, 'unknown'
        while status == 'unknown' and done < args.fixed_cap + 1:
            rows, status = solver.run(min(100, args.fixed_cap + 1 - done))   # chunks: a long run stays visible on stderr
            done += len(rows)
            print(f'[fixed-cap] {name}: {done} records, f = {rows["f"][-1] if len(rows) else float("nan"):.6f}, '
                  f'{time.perf_counter() - t0:.1f} s', file=sys.stderr, flush=True)
            if len(rows) == 0:
                break
        dt = time.perf_counter() - t0
        it, status, f_solver = solver.state()
        x = solver.get(_lib.GET_X_NOW)
        f, g = dev.eval(x)
        dvec = -g
        dvec[(ub - x <= 1e-12) & (dvec > 0)] = 0.0
        dvec[(x - lb <= 1e-12) & (dvec < 0)] = 0.0
        rec = {'workload': name, 'n': n, 'd': d, 'dual_dim': N, 'solver': cfg['solver'], 'storage': cfg['storage'], 'iterations': int(it),
               'status': status, 'f': float(f), 'f_solver_last_record': float(f_solver), 'proj_grad_norm_2': float(np.linalg.norm(dvec)),
               'proj_grad_norm_inf': float(np.abs(dvec).max()), 'wall_s': dt, 'iter_per_s': it / dt if dt > 0 else None,
               'problem_setup_s': t_setup, 'n_at_lower': int((x - lb <= 1e-12).sum()),
               'n_at_upper': int((ub - x <= 1e-12).sum()), 'n_sv_alpha_gt_1e-6': int((x > 1e-6).sum())}
        if ascg:
            rec['inner_products'] = int(solver.inner_iters())
            rec['inner_products_per_outer_iteration'] = rec['inner_products'] / max(int(it), 1)
            rec['inner_tol'] = args.inner_tol
        out['runs'][name] = rec
        solver.close()
        quad.release()
    print(json.dumps(out), flush=True)


def collective_floor(args):
    """--collective-floor: the closing collective of a product on a ONE-rank RCCL communicator, for the real message sizes: the
    default all-gather of segment partial vectors (8 segments of nb*256 doubles in all, whatever the rank count) and the
    alternative all-reduce(sum) of nb*256 doubles.  With one rank RCCL moves nothing between GPUs: this is the launch + local
    copy floor of the call as the library issues it (same stream, same in-place buffers), a LOWER bound for N > 1."""
    from optiml_amd import device
    from optiml_amd.dist import SocketComm
    ctx = device.Context(comm=SocketComm(0, 1), exchange='rccl')
    info = ctx.comm_info()
    out = {'what': 'collective_floor', 'unit': 'us', 'rccl_ranks': info['rccl_ranks'], 'device': ctx.name,
           'rccl_init_stages': info['rccl_init_stages'],
           'note': 'ONE-rank RCCL communicator on one GPU: launch + local-copy floor of the one collective per product, HIP events '
                   'around each of 50 calls on the compute stream; a lower bound of the N > 1 cost over xGMI, not an estimate of it'}
    for key, n in (('headline', 100000), ('c4', 100000), ('c5', 250000)):
        ln = -(-n // 256) * 256
        g = ctx.probe_exchange('gather', 8 * ln, 50)        # one rank owns all 8 canonical segments: the whole gathered buffer
        a = ctx.probe_exchange('allreduce', ln, 50)
        out[key] = {'n': n, 'gather_8_segments': {'bytes': 8 * ln * 8, 'mean_us': g[0], 'min_us': g[1]},
                    'allreduce': {'bytes': ln * 8, 'mean_us': a[0], 'min_us': a[1]}}
    ctx.close()
    print(json.dumps(out), flush=True)


# ---------------------------------------------------------------------------------------------------------------------
# the last stdout line: the headline record cut to what the driver parses
# ---------------------------------------------------------------------------------------------------------------------
def _r(v, sig=6):
    """Floats to `sig` significant digits (the line is a summary: the uncut numbers are in --records-file)."""
    if isinstance(v, float):
        return float(f'{v:.{sig}g}') if v == v and abs(v) != float('inf') else None
    if isinstance(v, dict):
        return {k: _r(x, sig) for k, x in v.items()}
    if isinstance(v, (list, tuple)):
        return [_r(x, sig) for x in v]
    return v


def _pick(src, keys):
    return {k: src[k] for k in keys if isinstance(src, dict) and k in src}


def compact_line(full, records_file=None):
    """The record the driver parses: BASELINE's metric on BASELINE's config with `roofline` and `cpu_baseline`, one-number
    summaries of every side record, nothing else.  Always shorter than LINE_LIMIT: if a summary would push it over, summaries
    are dropped from the end (and named in `dropped`) — the contract fields never are."""
    out = _pick(full, ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling',
                       'vs_baseline', 'dtype', 'data'))
    cfg = full.get('config', {})
    out['config'] = _pick(cfg, ('workload', 'n', 'd', 'dual_dim', 'solver', 'C', 'gamma', 'exchange', 'rccl_ranks', 'sym_exchange',
                                'rows_per_gpu'))
    if cfg.get('panel_placement_ms') is not None:
        out['config']['panel_placement_ms'] = [float(f'{v:.4g}') for v in cfg['panel_placement_ms']]
        out['config']['placement_budget'] = (cfg.get('placement_budget') or '').split(':')[0].split(' (')[0][:40]
    if 'device' in cfg:
        out['config']['device'] = cfg['device'].split(' [')[0] + (' ' + cfg['device'][cfg['device'].rfind('('):] if '(' in cfg['device'] else '')
    roof = full.get('roofline') or {}
    out['roofline'] = _pick(roof, ('bound', 'kernel', 'achieved', 'peak', 'unit', 'frac', 'traffic', 'traffic_source', 'avg_launch_ms',
                                   'launches', 'algorithmic_bytes_per_launch', 'flops_per_launch', 'frac_survey_8d_bytes',
                                   'measured_stream_read_GBs', 'frac_first_placement'))
    if roof.get('bound') == 'hbm':
        out['roofline']['bytes_basis'] = 'bytes the kernel moves (lower-triangle tiles of the symmetric panel + partial slab); ' \
                                         'frac_survey_8d_bytes = same launch priced at n^2*s, never read'
    cpu = full.get('cpu_baseline')
    if cpu:
        c = _pick(cpu, ('value', 'unit', 'kind', 'extrapolated', 'cores', 'cpu_model', 'law', 'fitted_exponent', 'value_fitted_exponent',
                        'value_range'))
        if 'samples' in cpu:
            c['samples'] = [_pick(x, ('n', 'iters', 's_per_iter', 's_per_iteration')) for x in cpu['samples']]
        c['sample'] = (cpu.get('sample') or '')[:240]
        sp = cpu.get('streamed_product_full_n')
        if sp:
            c['streamed_product_full_n'] = {'n': sp.get('n'), 's_per_product': min(sp['s_per_product']) if sp.get('s_per_product') else None}
        out['cpu_baseline'] = c
    else:
        out['cpu_baseline'] = None
    for k in ('speedup_vs_cpu_baseline', 'steps_done', 'solver_status', 'gram_build_s', 'exchange_ms_per_product', 'inner_products_per_step'):
        if k in full:
            out[k] = full[k]
    # ---- one-number summaries, in the order they are dropped LAST -> FIRST when the line would be too long
    extra = []
    per = full.get('per_rank')
    if per:
        extra.append(('per_rank', {k: [p.get(k) for p in per] for k in ('tiles', 'ms_per_step', 'symv_tiles_ms', 'exchange_ms_per_product')}))
    xc = full.get('exchange_compare')
    if xc:
        extra.append(('exchange_compare', xc if 'error' in xc else {m: _pick(v, ('exchange_ms_per_product', 'ms_per_step')) for m, v in xc.items()}))
    side = {}
    for name, rec in (full.get('configs') or {}).items():
        if 'value' in rec:
            key = 'c5_outer_it_s' if name == 'c5' else f'{name}_iter_s'
            side[key] = rec['value']
            side[f'{name}_frac'] = (rec.get('roofline') or {}).get('frac')
            if 'inner_products_per_step' in rec:
                side[f'{name}_products_per_outer_it'] = rec['inner_products_per_step']
        else:
            side[f'{name}_iter_s'] = rec.get('error') or rec.get('skipped')
    tab = (full.get('shares') or {}).get('headline') or {}
    parts = {p['G']: p for p in tab.get('partitions', []) if 'G' in p}
    if parts:
        side['predicted_x_at_2_4_8'] = [parts.get(g, {}).get('predicted_speedup_vs_1') for g in (2, 4, 8)]
        side['predicted_symv_frac_at_2_4_8'] = [parts.get(g, {}).get('min_symv_frac_of_8TBs') for g in (2, 4, 8)]
        side['predicted_assumed_exchange_us'] = tab.get('assumed_exchange_us')
    for key, g, name in (('c4_over_4', 4, 'c4_predicted_x_at_4'), ('c5_over_8', 8, 'c5_predicted_x_at_8')):
        for p in ((full.get('shares') or {}).get(key) or {}).get('partitions', []):
            if p.get('G') == g:
                side[name] = p.get('predicted_speedup_vs_1')
    floor = full.get('collective_floor_us') or {}
    if 'headline' in floor:
        side['collective_floor_us'] = {k: floor['headline'][k].get('mean_us') for k in ('gather_8_segments', 'allreduce') if k in floor['headline']}
    kk = full.get('time_to_kkt') or {}
    dq = ((full.get('dense_quadratic') or {}).get('config2_size') or {})
    if 'layouts' in dq:
        side['dense_c2_packed_rows_iter_s'] = [dq['layouts'][k]['value'] for k in ('packed', 'rows') if k in dq['layouts']]
        side['dense_c2_packed_rows_frac'] = [dq['layouts'][k]['roofline']['frac'] for k in ('packed', 'rows') if k in dq['layouts']]
    elif dq:
        side['dense_c2_packed_rows_iter_s'] = dq.get('error') or dq.get('skipped')
    for key, name in (('ip_config3', 'ip_c3_s'), ('as_config2_shape', 'as_c2_s'), ('smo', 'smo_headline_s'), ('c5_projected', 'c5_projected_s'),
                      ('ip_headline', 'ip_headline_s'), ('as_headline', 'as_headline_s')):
        if key in kk and isinstance(kk[key], dict):
            side[name] = kk[key].get('value', kk[key].get('error'))
            fr = (kk[key].get('roofline') or {}).get('frac')
            if fr is not None:
                side[name[:-2] + '_frac'] = fr
    caps = full.get('fixed_cap') or {}
    fc = {k: [v.get('iterations'), v.get('proj_grad_norm_2'), v.get('wall_s')] for k, v in caps.items() if isinstance(v, dict) and 'iterations' in v}
    if fc:
        side['fixed_cap_iters_projgrad_wall_s'] = fc
    if side:
        extra.append(('side', side))
    wall = (full.get('records') or {}).get('wall_s') or {}
    if wall:
        extra.append(('wall_s', wall.get('total_s')))
    if (full.get('records') or {}).get('state'):
        extra.append(('state', full['records']['state']))
    if records_file:
        extra.append(('full_record', os.path.relpath(records_file, REPO) if records_file.startswith(REPO) else records_file))
    for k, v in extra:
        out[k] = v
    out = _r(out)
    dropped = []   # never reached by the records this file produces today (2.8 KB); the bound holds whatever is added later
    while len(json.dumps(out)) >= LINE_LIMIT:
        if isinstance(out.get('side'), dict) and out['side']:
            dropped.append('side.' + out['side'].popitem()[0])
        else:
            victims = [k for k, _ in extra if k in out]
            if not victims:
                break
            out.pop(victims[-1])
            dropped.append(victims[-1])
        out['dropped'] = dropped
    return out


def emit(full, args, stream=None):
    """The complete record to --records-file (indented) and to stderr (one line), then THE line on stdout: the complete record with
    --line full (children of the default line), the compact one otherwise."""
    stream = stream or sys.stdout
    if args.line == 'full':
        print(json.dumps(full), file=stream, flush=True)
        return
    path = args.records_file
    try:
        with open(path, 'w') as fh:
            json.dump(full, fh, indent=1)
            fh.write('\n')
    except OSError as exc:   # a read-only checkout must not cost the line
        print(f'[bench] complete record not written to {path}: {exc!r}', file=sys.stderr, flush=True)
        path = None
    print('[bench] complete record: ' + json.dumps(full), file=sys.stderr, flush=True)
    print(json.dumps(compact_line(full, path)), file=stream, flush=True)

# ---------------------------------------------------------------------------------------------------------------------
def main():
    args = parse()
    if args.cpu_study:
        return cpu_study(args)
    if args.dense:
        if args.gpus != 1 or 'WORLD_SIZE' in os.environ:
            raise SystemExit('--dense is a one-GPU record (the sharding of a packed dense Hessian is that of the kernel panels: --gpus N on the default workload)')
        return dense_record(args)
    if args.emulate_shares:
        return share_timing(args)
    if args.collective_floor:
        return collective_floor(args)
    if args.fixed_cap > 0:
        return fixed_cap(args)
    if args.records != 'none' and args.default_workload and args.gpus == 1 and 'WORLD_SIZE' not in os.environ:
        return orchestrate(args)   # before anything in this process touches HIP
    if args.solver in ('ip', 'as', 'smo'):
        if args.gpus != 1:
            raise SystemExit('InteriorPoint / ActiveSet factorise on one GPU and SMO walks the samples sequentially '
                             '(replicas only): --gpus 1')
        return bench_kkt_line(args)
    if 'WORLD_SIZE' not in os.environ and args.gpus > 1:
        raise SystemExit(spawn_ranks(args.gpus))   # before anything in this process touches HIP
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    # the host driver of this pool supports dmabuf IPC only: without it RCCL's cross-process buffer sharing fails with
    # `hipIpcGetMemHandle: invalid argument` (set before anything loads the HIP runtime in this process; a launcher's own value wins)
    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    if world != args.gpus:
        raise SystemExit(f'--gpus {args.gpus} does not match WORLD_SIZE={world}')
    # stdout carries ONE JSON line and nothing else: gloo and RCCL (NCCL_DEBUG) write to fd 1 from C++, so fd 1 is pointed at
    # stderr for the rest of the run and the line goes to a private copy of the original stdout
    sys.stdout.flush()
    json_out = os.fdopen(os.dup(1), 'w')
    os.dup2(2, 1)

    from optiml_amd import _lib
    from optiml_amd import device
    from optiml_amd.datasets import make_blobs, make_regression
    from optiml_amd.ml.svm.kernels import PolyKernel, gaussian, linear
    from optiml_amd.opti import KernelQuadratic
    from optiml_amd.opti.constrained._base import _DeviceSolver

    comm = None
    if world > 1:
        if args.exchange == 'rccl':
            prefetch_rccl()
        # rendezvous / barrier / max-over-ranks: torch.distributed (gloo) when torch is importable, else the package's own
        # TCP communicator (BQ_RENDEZVOUS=socket forces it) — the data path is RCCL either way
        from bench_rendezvous import from_env
        comm = from_env(timeout=600.0)
        # RCCL over xGMI is the data path.  A communicator that cannot be created on EVERY rank ends the run with exit code
        # 3 — a scaling number must never silently be a host-transport number — unless --allow-host-exchange asks for the
        # gloo fallback (reported in config.exchange).
        exchange, ctx, err = args.exchange, None, ''
        if exchange == 'rccl':
            # ncclCommInitRank has no timeout of its own: a bootstrap that never completes (a rank that died after the
            # pre-flight, an unusable network interface) must end this run with a message, not sit until the caller's limit
            import threading

            def _give_up():
                print(f'[bench] rank {rank}: the RCCL communicator was not created within {args.rccl_init_timeout:.0f} s '
                      '(NCCL_DEBUG=INFO shows the bootstrap; NCCL_SOCKET_IFNAME selects its interface)', file=sys.stderr, flush=True)
                os._exit(3)
            watchdog = threading.Timer(args.rccl_init_timeout, _give_up)
            watchdog.daemon = True
            watchdog.start()
            try:
                tmo = args.collective_timeout if args.collective_timeout is not None else 120.0 + 0.1 * (args.steps + args.warmup)
                ctx = device.Context(comm=comm, exchange='rccl', sym_exchange=args.sym_exchange, collective_timeout=tmo or None)
            except Exception as exc:  # noqa: BLE001
                err = repr(exc)
            finally:
                watchdog.cancel()
            if comm.max_float(0.0 if ctx is not None else 1.0) > 0.0:
                print(f'[bench] rank {rank}: RCCL context unavailable ({err or "failed on another rank"})', file=sys.stderr, flush=True)
                if ctx is not None:
                    ctx.close()
                if not args.allow_host_exchange:
                    comm.close()
                    raise SystemExit(3)
                print(f'[bench] rank {rank}: --allow-host-exchange: using the host (gloo) exchange', file=sys.stderr, flush=True)
                ctx, exchange = None, 'host'
        if ctx is None:
            ctx = device.Context(comm=comm, exchange=exchange, sym_exchange=args.sym_exchange)
        device.set_context(ctx)
    else:
        ctx = device.get_context()
    cinfo = ctx.comm_info()

    def barrier():
        if comm is not None:
            comm.barrier()

    n, d = args.n, args.d
    kern = {'rbf': gaussian, 'poly': PolyKernel(3, 'scale', 1.0), 'linear': linear}[args.kernel]
    al = args.solver == 'adagrad'   # reg_intercept=False dual: no rank-one term, equality row handled by the multiplier
    ascg = args.solver == 'ascg'    # squared-hinge dual: K*yy' + yy' + I/(2C), ub = +inf, x0 = 1
    if args.task == 'svc':
        X, y = make_blobs(n, d, seed=0, sigma=args.sigma)
        quad = KernelQuadratic(X, -np.ones(n), 'svc', kern, y=y, storage=args.storage, rank_one=not al,
                               diag=0.5 if ascg else 0.0, tune_placement=not args.no_placement, expected_products=STEADY_STATE_PRODUCTS)
        a_eq = y
    else:   # eps-insensitive SVR dual: 2n variables on one n x n panel (BASELINE config 4 shape)
        X, y = make_regression(n, d, seed=0)
        quad = KernelQuadratic(X, np.hstack((-y, y)) + 0.1, 'svr', kern, storage=args.storage, rank_one=not al,
                               tune_placement=not args.no_placement, expected_products=STEADY_STATE_PRODUCTS)
        a_eq = np.hstack((np.ones(n), -np.ones(n)))
    N = quad.ndim
    ctx.profile(True)   # timing of the dominant kernel by its own dispatch timestamps (roofline)
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
    elif ascg:
        solver = _DeviceSolver(dev, _lib.AS_CG, np.zeros(N), np.full(N, np.inf), np.ones(N), 1e-6, 10 ** 9)
        solver.set_inner(args.inner_tol, 0)
    else:
        kind = _lib.PG if args.solver == 'pg' else _lib.FW
        solver = _DeviceSolver(dev, kind, np.zeros(N), ub, ub / 2, 1e-6, 10 ** 9)

    rows, status = solver.run(max(args.warmup, 1))      # includes the start-up gradient product
    ctx.profile_read(_lib.PROF_MATVEC, reset=True)
    ctx.profile_read(_lib.PROF_EXCH, reset=True)
    ctx.profile_read(_lib.PROF_PCSHARD, reset=True)
    inner0 = solver.inner_iters() if ascg else 0
    barrier()
    t0 = time.perf_counter()
    rows, status = solver.run(args.steps)                # device-resident; returns after the stream has drained
    t1 = time.perf_counter()
    barrier()
    own_elapsed = elapsed = t1 - t0
    if comm is not None:
        elapsed = comm.max_float(elapsed)
    done = len(rows)
    mv_ms, mv_cnt = ctx.profile_read(_lib.PROF_MATVEC, reset=True)
    ex_ms, ex_cnt = ctx.profile_read(_lib.PROF_EXCH, reset=True)
    pcs_ms, pcs_cnt = ctx.profile_read(_lib.PROF_PCSHARD, reset=True) if ascg else (0.0, 0)
    inner = (solver.inner_iters() - inner0) if ascg else 0
    per_rank = None
    if comm is not None:   # every rank's share and timings in the one line rank 0 prints
        cols = [comm.allgather_float(v) for v in (own_elapsed * 1e3 / max(done, 1), mv_ms / max(mv_cnt, 1),
                                                  (ex_ms / ex_cnt) if ex_cnt else 0.0)]
        per_rank = []
        for k in range(world):
            b, e = device.row_block(n, k, world, symmetric=True)
            j0, j1 = b // 256, -(-e // 256)
            per_rank.append({'rank': k, 'rows': e - b, 'tile_rows': [j0, j1], 'tiles': j1 * (j1 + 1) // 2 - j0 * (j0 + 1) // 2,
                             'ms_per_step': cols[0][k], 'symv_tiles_ms': cols[1][k], 'exchange_ms_per_product': cols[2][k]})

    esz = 8 if args.storage == 'f64' else 4
    probe = (0.0, 0.0)
    if rank == 0:
        try:   # outside the timed region: what a plain streaming read / copy reaches on this GPU (4 GiB scratch)
            probe = ctx.probe_bandwidth(4 << 30, 5)
        except Exception as exc:  # noqa: BLE001  (e.g. not enough free HBM beside a very large panel)
            print(f'[bench] bandwidth probe skipped: {exc!r}', file=sys.stderr, flush=True)
    out = None
    if rank == 0:
        avg_ms = mv_ms / max(mv_cnt, 1)
        # the dominant kernel is the symmetric tile product: this rank streams the 256 x 256 tiles on/below the
        # diagonal of its tile rows once and writes two 256-vectors per tile into the partial-product slab
        T = 256
        i0, i1 = r0 // T, -(-r1 // T)
        tiles = i1 * (i1 + 1) // 2 - i0 * (i0 + 1) // 2
        alg_bytes = tiles * (T * T * esz + T * 8) + (tiles // 8 + i1 - i0) * T * 8 + 2 * n * 8   # tiles + col parts + row parts
        achieved = alg_bytes / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
        # SURVEY 8(d)'s per-product bytes for a row-block panel, n^2 s / G + 3 n s per GPU — twice what this kernel moves
        survey_bytes = n * n * esz / world + 3 * n * 8
        full_equiv = survey_bytes / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
        if args.task == 'svc':
            loss = 'sqhinge' if ascg else 'hinge'
            workload = f'svc_{loss}_{args.kernel}_{args.solver}_dual_n{n}_d{d}'
        else:
            workload = f'svr_epsins_{args.kernel}_{args.solver}_dual_n{n}_d{d}'
        out = {
            'metric': 'dual_qp_iterations_per_sec', 'value': done / elapsed, 'unit': 'iter/s',
            'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': 1e3 * elapsed / max(done, 1),
            'higher_is_better': True, 'scaling': 'strong', 'vs_baseline': None,
            'dtype': 'f64' if args.storage in ('f64', 'stream') else 'f32-storage/f64-accumulate', 'data': 'synthetic',
            'config': {'workload': workload, 'n': n, 'd': d, 'dual_dim': N, 'C': 1.0,
                       'gamma': 'scale', 'solver': args.solver, 'exchange': ctx.exchange, 'rccl_ranks': cinfo['rccl_ranks'],
                       'sym_exchange': cinfo['sym_exchange'] if world > 1 else 'none', 'blob_sigma': args.sigma,
                       'rccl_init_stages': cinfo['rccl_init_stages'],
                       'rows_per_gpu': r1 - r0, 'device': ctx.name,
                       'panel_placement_ms': dev.placement(),
                       'placement_budget': 'off' if args.no_placement else 'steady state: up to 5 s (SVC.fit: 2 % of max_iter products, 0.2 s at least)'},
            'roofline': {'bound': 'hbm', 'kernel': 'symv_tiles_kernel (symmetric panel product Q*d)', 'achieved': achieved,
                         'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': achieved / HBM_PEAK_GBS,
                         'traffic': None, 'traffic_source': None, 'avg_launch_ms': avg_ms, 'launches': mv_cnt,
                         'algorithmic_bytes_per_launch': alg_bytes, 'tiles_per_launch': tiles,
                         'survey_8d_bytes_per_launch': survey_bytes, 'survey_8d_equivalent_GBs': full_equiv,
                         'frac_survey_8d_bytes': full_equiv / HBM_PEAK_GBS,
                         'note': 'frac is on the bytes the kernel moves: lower-triangle tiles only (the Gram panel is symmetric; '
                                 'both contributions of a tile are formed from one read). frac_survey_8d_bytes prices the same '
                                 'launch at SURVEY 8(d)\'s n^2*s row-block bytes, which are never read — it can exceed 1.',
                         'measured_stream_read_GBs': probe[0], 'measured_copy_GBs': probe[1],
                         'frac_of_measured_stream_read': (achieved / probe[0]) if probe[0] else None,
                         # what the line would read WITHOUT the placement choice: the rate of the first allocation as the choice timed
                         # it (product + its closing kernel on the still empty panel: a few % below the tile kernel alone)
                         'frac_first_placement': None},
            'steps_done': done, 'solver_status': status,
            'f_last': float(rows['f'][-1]) if done else None,
            'kkt_resid_last': float(rows['r1'][-1]) if done and args.solver == 'pg' else None,
            'gram_build_s': gram_ms * 1e-3, 'problem_setup_s': t_gram_total,
            'exchange_ms_per_step': (ex_ms / max(done, 1)) if ex_cnt else 0.0,
            'exchange_ms_per_product': (ex_ms / ex_cnt) if ex_cnt else 0.0,
        }
        placed = out['config']['panel_placement_ms']
        if placed:
            out['roofline']['frac_first_placement'] = alg_bytes / (placed[0] * 1e-3) / 1e9 / HBM_PEAK_GBS
            out['roofline']['frac_chosen_placement_as_timed_by_the_choice'] = alg_bytes / (min(placed) * 1e-3) / 1e9 / HBM_PEAK_GBS
        if per_rank is not None:
            out['per_rank'] = per_rank
        if ascg:
            out['inner_products_per_step'] = inner / max(done, 1)
            out['inner_tol'] = args.inner_tol
            hooks = dict(h.split('=', 1) for h in os.environ.get('BQ_TEST_HOOKS', '').split(',') if '=' in h)
            fam = hooks.get('as_cg_pc_class', '2')
            out['inner_preconditioner'] = 'none (BQ_AS_CG_PC=0)' if os.environ.get('BQ_AS_CG_PC') == '0' else \
                'diagonal + Taylor features of the RBF kernel through Woodbury: orders 0-1 (d + 2 columns)' + \
                {'2': ' + the order-2 term projected onto the 2d class-mean directions (3d + 2 columns in all)' +
                      (', the rest of the order-2 term applied without features behind a degree-1 Chebyshev polynomial' if n >= 65536 and 'as_cg_pc_class' not in hooks else ''),
                 '3': ' + the order-2 term projected onto the 2d class-mean directions + its implicit remainder (forced)',
                 '1': ' + the class-mean cross term of rounds 3-4 (2d + 2 columns in all)'}.get(fam, '')
            out['time_to_kkt_projected'] = c5_projection(n, d, 1e3 * elapsed / max(done, 1))
            out['inner_warm_start'] = hooks.get('as_cg_warm') != '0'
            out['products_per_sec'] = mv_cnt / elapsed
            # one application of the preconditioner (HIP events around it): every pass over samples in it is sharded since round 6 — the
            # explicit model's two passes over the features and the implicit order-2 remainder; on G ranks it costs 1/G of this + its
            # collectives (2 for the explicit model alone; 6 with the remainder: t, z, M + Phi_top'y, v, t, z)
            remainder = fam == '3' or (n >= 65536 and 'as_cg_pc_class' not in hooks and fam == '2')
            out['preconditioner_sharded_ms_per_step'] = pcs_ms / max(done, 1)
            out['preconditioner_sharded_calls_per_step'] = pcs_cnt / max(done, 1)
            out['preconditioner_collectives_per_call'] = 6 if remainder else 2
        traffic = measured_traffic(workload, world) if args.storage == 'f64' else None
        if traffic:
            out['roofline']['traffic'] = traffic['hbm_bytes']
            out['roofline']['traffic_source'] = traffic['source']
            out['roofline']['traffic_note'] = 'PMC pass (FETCH_SIZE x2 per the gfx950 note, + WRITE_SIZE) of this command, committed file'
        if args.storage == 'stream':   # no panel: the product is the fused Gram-tile x vector kernel, MFMA-bound
            dp = -(-d // 16) * 16
            full = 2.0 * (-(-(r1 - r0) // 128) * 128) * (-(-n // 128) * 128) * dp
            t0, t1 = r0 // 128, -(-r1 // 128)   # every lower-triangle tile of this rank's tile rows once, used for rows and columns
            flops = 2.0 * ((t1 * (t1 + 1) - t0 * (t0 + 1)) // 2) * 128 * 128 * dp
            tf = flops / (avg_ms * 1e-3) / 1e12 if avg_ms > 0 else 0.0
            out['roofline'] = {'bound': 'mfma',
                               'kernel': 'gram_stream_sym_kernel (lower-triangle Gram tiles recomputed, each used for its rows and its '
                                         'columns, fused with the product)',
                               'achieved': tf, 'peak': FP64_MFMA_PEAK_TF, 'unit': 'TFLOP/s', 'frac': tf / FP64_MFMA_PEAK_TF, 'traffic': None,
                               'avg_launch_ms': avg_ms, 'launches': mv_cnt, 'flops_per_launch': flops,
                               'row_block_equivalent_TFs': full / (avg_ms * 1e-3) / 1e12 if avg_ms > 0 else 0.0,
                               'note': 'frac counts the MFMA flops actually issued (the tiles on/below the diagonal in this rank\'s tile rows); '
                                       'row_block_equivalent_TFs prices the launch at 2 rows n d, which is not what runs'}
    if comm is not None and not al and not ascg and args.storage != 'stream':
        # AFTER the headline record is complete and OUTSIDE the timed region: the same few iterations with the OTHER closing
        # collective of the symmetric product, so that one multi-GPU run shows both (all-gather of 8 segment vectors + ordered sum,
        # bit-identical for any N, against one all-reduce of n doubles) — collective time per product, max over ranks.  A side
        # record must never cost the line: if it has not finished within 90 s, rank 0 prints the line without it and every rank
        # leaves.
        import copy
        import threading
        snapshot = copy.deepcopy(out) if rank == 0 else None   # what the timer thread may print: never the dict the main thread edits
        stage = {'at': 'start', 'steps': 0}

        def _bail():
            # nothing but a collective or a kernel that never completes can expire this timer: that is a hang, and a hang is a
            # failure — every rank says where it was and leaves with a non-zero code
            print(f'[bench] rank {rank}: exchange_compare hung at "{stage["at"]}" (sym_exchange={stage.get("mode")}, '
                  f'{stage["steps"]} comparison steps enqueued) — not finished within 90 s', file=sys.stderr, flush=True)
            if rank == 0:
                snapshot['exchange_compare'] = {'error': 'the comparison run did not finish within 90 s', 'stuck_at': stage['at'],
                                                'sym_exchange': stage.get('mode')}
                snapshot['cpu_baseline'] = None
                emit(snapshot, args, json_out)
            os._exit(4)
        watchdog = threading.Timer(90.0, _bail)
        watchdog.daemon = True
        watchdog.start()
        exchange_compare = None
        try:
            other = 'allreduce' if cinfo['sym_exchange'] == 'gather' else 'gather'
            stage.update(at='switching the closing collective', mode=other)
            ctx.set_sym_exchange(other)
            s2 = _DeviceSolver(dev, _lib.PG if args.solver == 'pg' else _lib.FW, np.zeros(N), ub, ub / 2, 1e-6, 10 ** 9)
            stage.update(at='3 warm-up iterations', steps=3)
            s2.run(3)
            ctx.profile_read(_lib.PROF_EXCH, reset=True)
            stage.update(at='barrier before the timed comparison')
            barrier()
            t2 = time.perf_counter()
            stage.update(at='10 timed iterations', steps=13)
            rows2, _ = s2.run(10)
            dt2 = time.perf_counter() - t2
            stage.update(at='max over ranks of the comparison times')
            ex2_ms, ex2_cnt = ctx.profile_read(_lib.PROF_EXCH, reset=True)
            s2.close()
            exchange_compare = {
                cinfo['sym_exchange']: {'exchange_ms_per_product': comm.max_float((ex_ms / ex_cnt) if ex_cnt else 0.0),
                                        'ms_per_step': 1e3 * elapsed / max(done, 1)},
                other: {'exchange_ms_per_product': comm.max_float((ex2_ms / ex2_cnt) if ex2_cnt else 0.0),
                        'ms_per_step': comm.max_float(1e3 * dt2 / max(len(rows2), 1)), 'steps': len(rows2)}}
        except Exception as exc:  # noqa: BLE001
            exchange_compare = {'error': repr(exc)}
        finally:
            watchdog.cancel()
            ctx.set_sym_exchange(cinfo['sym_exchange'])
        if rank == 0:
            out['exchange_compare'] = exchange_compare
    barrier()
    solver.close()
    quad.release()
    if rank == 0:
        if world == 1 and args.kkt != 'none' and args.solver == 'pg' and args.task == 'svc' and args.storage == 'f64':
            # BASELINE's second metric in the same line: the routes that reach a KKT tolerance
            kk = {}
            try:
                if args.kkt in ('smo', 'all', 'ip100k'):
                    kk['smo'] = kkt_smo(n, d, args.sigma, X=X, y=y, cpu=not args.no_cpu)
                if args.kkt in ('ip', 'all', 'ip100k'):
                    kk['ip_config3'] = kkt_box('ip', 50000, 128, args.sigma, cpu=not args.no_cpu)
                if args.kkt == 'ip100k':
                    kk['ip_headline'] = kkt_box('ip', n, d, args.sigma, cpu=False)
            except Exception as exc:  # noqa: BLE001 — the headline number must survive a failing side record
                kk['error'] = repr(exc)
            out['time_to_kkt'] = kk
        if world == 1 and not args.no_cpu:
            out['cpu_baseline'] = cpu_baseline(args)
            out['speedup_vs_cpu_baseline'] = out['value'] / out['cpu_baseline']['value']
        else:
            out['cpu_baseline'] = None
        emit(out, args, json_out)
    barrier()
    if comm is not None:
        comm.close()


def c5_projection(n, d, ms_per_outer_iteration):
    """Config 5 cannot be run to 'optimal' inside a bench (the reference algorithm moves about one index per outer iteration: of the
    order of n of them).  Projection: the iterations / n the SAME workload needed to 'optimal' at the sizes tools/c5_scaling.py ran
    (committed: profiles/rNN/c5_outer_iterations_scaling.json, largest n) x this n x the seconds per outer iteration measured here.
    None when no scaling file is committed."""
    import glob
    best = None
    for path in sorted(glob.glob(os.path.join(REPO, 'profiles', 'r*', 'c5_outer_iterations_scaling.json'))):
        try:
            rec = json.load(open(path))
        except Exception:  # noqa: BLE001
            continue
        runs = [r for r in rec.get('runs', []) if r.get('status') == 'optimal' and rec.get('d') == d]
        if runs:
            best = (max(runs, key=lambda r: r['n']), os.path.relpath(path, REPO))
    if best is None:
        return None
    run, src = best
    iters = run['iterations_per_n'] * n
    return {'value': iters * ms_per_outer_iteration * 1e-3, 'unit': 's', 'kind': 'projected, not measured',
            'outer_iterations_projected': iters, 'iterations_per_n': run['iterations_per_n'], 'from_n': run['n'], 'source': src,
            'ms_per_outer_iteration': ms_per_outer_iteration,
            'note': 'iterations / n of the same workload run to its stop test at a size that finishes, times n, times the measured time per '
                    'outer iteration at this n (which grows with the free set shrinking: an upper estimate of the rate, a projection of the time)'}


def measured_traffic(workload, world):
    """HBM bytes per launch of the panel-product kernel from the committed PMC passes (profiles/rNN/pmc_traffic_*.json,
    produced by tools/pmc_summary.py from separate `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` runs of this same
    command); the newest round that matches this workload, None when none does."""
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


if __name__ == '__main__':
    main()
