"""Rank program for the multi-process tests (started by tests/test_distributed.py, one process per rank).

usage: python tests/_dist_worker.py MODE OUTDIR      with RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT in the env
  cpu-torch / cpu-socket / cpu-fromenv-socket : communicator primitives + a row-sharded product with the CPU oracle's Hessian (no GPU)
  gpu-host               : every rank on GPU 0, row-block panels, host (gloo) exchange, PG + FW solves
  gpu-rccl               : RCCL exchange (world size 1 on a one-GPU box exercises init / all-gather / destroy)
  gpu-host-allreduce / gpu-rccl-allreduce : the same with sym_exchange='allreduce'
"""
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)


def make_comm(kind):
    rank, world = int(os.environ['RANK']), int(os.environ['WORLD_SIZE'])
    if kind == 'fromenv-socket':   # what bench.py does on a host without torch (BQ_RENDEZVOUS=socket forces it)
        from optiml_amd.dist import SocketComm
        from bench_rendezvous import from_env
        os.environ['BQ_RENDEZVOUS'] = 'socket'
        comm = from_env(timeout=60.0)
        assert isinstance(comm, SocketComm)
        return comm
    if kind == 'socket':
        from optiml_amd.dist import SocketComm
        return SocketComm(rank, world, os.environ.get('MASTER_ADDR', '127.0.0.1'),
                          int(os.environ['MASTER_PORT']) + 1)
    import torch.distributed as dist
    dist.init_process_group(backend='gloo', rank=rank, world_size=world)
    from bench_rendezvous import TorchComm
    return TorchComm()


def cpu_mode(kind, outdir):
    from oracle import svm_oracle as so
    from optiml_amd.datasets import make_blobs
    comm = make_comm(kind)
    blob = comm.broadcast_bytes(bytes(range(128)) if comm.rank == 0 else b'\0' * 128, src=0)
    assert blob == bytes(range(128))
    res = {}
    for n in (300, 1000):
        X, y = make_blobs(n, 5, seed=n)
        Q, q, _ = so.svc_dual(so.gram('rbf', X), y, 1.0)
        v = np.random.RandomState(1).standard_normal(n)
        r0, r1 = comm.rows_of(n)
        buf = np.zeros(n)
        buf[r0:r1] = Q[r0:r1] @ v            # this rank's row block of the product
        comm.allgather_rows(buf, r0, r1)
        res[f'full_{n}'] = np.concatenate([Q[b:e] @ v for b, e in (comm.rows_of(n, r) for r in range(comm.world_size))])
        res[f'gathered_{n}'] = buf
    res['tmax'] = comm.max_float(float(comm.rank + 1))
    comm.barrier()
    np.savez(os.path.join(outdir, f'rank{comm.rank}.npz'), **res)


def gpu_mode(exchange, outdir, sym_exchange=None):
    from optiml_amd import device, _lib
    from optiml_amd.datasets import make_blobs, make_regression
    from optiml_amd.ml.svm.kernels import gaussian, PolyKernel
    from optiml_amd.opti import KernelQuadratic
    from optiml_amd.opti.constrained import ProjectedGradient, FrankWolfe
    comm = make_comm('torch')
    ctx = device.init_distributed(comm, exchange=exchange, device=0, sym_exchange=sym_exchange)
    assert ctx.world == comm.world_size
    assert ctx.exchange == (exchange if (comm.world_size > 1 or exchange == 'rccl') else 'none')
    info = ctx.comm_info()
    res = {'rccl_ranks': np.array(info['rccl_ranks']), 'sym_exchange': np.array(info['sym_exchange'])}
    n = int(os.environ.get('BQ_TEST_DIST_N', '700'))   # 700 = 3 tile rows of 256; the tests also run 257 and 2100
    X, y = make_blobs(n, 12, seed=5)
    quad = KernelQuadratic(X, -np.ones(n), 'svc', gaussian, y=y)
    dev = quad.device_problem()
    res['rows'] = np.array(dev.dims()[2:])
    v = np.random.RandomState(2).standard_normal(n)
    res['matvec'] = dev.matvec(v)
    res['gram_matvec'] = dev.gram_matvec(v)
    hist = []
    cb = lambda o: hist.append(o.f_x)
    cb._bq_needs_state = False
    opt = ProjectedGradient(quad=quad, ub=np.ones(n), max_iter=60, callback=cb).minimize()
    res['pg_x'], res['pg_hist'] = opt.x, np.array(hist)
    Xr, yr = make_regression(300, 6, seed=9)
    quad = KernelQuadratic(Xr, np.hstack((-yr, yr)) + 0.1, 'svr', PolyKernel(3, 'scale', 1.))
    opt = FrankWolfe(quad=quad, ub=np.ones(600), max_iter=40).minimize()
    res['fw_x'], res['fw_f'] = opt.x, opt.f_x
    # dense Quadratic, Q == Q' exactly: the packed lower tile rows of the kernel panels (segments + all-gather of the segment
    # partials) -> bit-identical for any world size
    from optiml_amd.opti import Quadratic
    rs = np.random.RandomState(4)
    G = rs.standard_normal((500, 520))
    Qd = G @ G.T / 500
    Qd = (Qd + Qd.T) / 2
    dq = Quadratic(Qd, rs.standard_normal(500))
    res['dense_rows'] = np.array(dq.device_problem().dims()[2:])
    res['dense_packed'] = np.array(dq.device_problem().layout()['packed'])
    # a dense Q that is NOT symmetric keeps whole row blocks + the all-gather of disjoint slices (NumPy's Q @ x)
    vq = np.random.RandomState(12).standard_normal(500)
    Qn = Qd + 0.01 * np.triu(rs.standard_normal((500, 500)), 1)
    nq = Quadratic(Qn, rs.standard_normal(500))
    res['rowsq_rows'] = np.array(nq.device_problem().dims()[2:])
    res['rowsq_packed'] = np.array(nq.device_problem().layout()['packed'])
    res['rowsq_matvec'] = nq.device_problem().matvec(vq)
    res['rowsq_ref'] = Qn @ vq
    res['rowsq_pg_x'] = ProjectedGradient(quad=nq, ub=np.ones(500), max_iter=40).minimize().x
    # one element one ulp away from its mirror image, placed where only the rank that owns row 400 meets the pair (its
    # column strip above its own rows): the ranks must AGREE on row blocks
    Qh = Qd.copy()
    Qh[10, 400] = np.nextafter(Qh[10, 400], np.inf)
    hq = Quadratic(Qh, np.zeros(500))
    res['hid_packed'] = np.array(hq.device_problem().layout()['packed'])
    res['hid_matvec'] = hq.device_problem().matvec(vq)
    res['hid_ref'] = Qh @ vq
    hq.release()
    # six tile rows (ranks own several, and rows above their own): a symmetric Q is packed on every rank count, and ONE element one
    # ulp off its mirror image sends every rank to row blocks wherever the pair sits — inside a rank's own tile rows (compared while
    # that rank's rows go up) or between two ranks' rows (met only in the later rank's column strip)
    nd = 1300
    G2 = np.random.RandomState(21).standard_normal((nd, 40))
    Q6 = G2 @ G2.T / nd + np.eye(nd)
    Q6 = (Q6 + Q6.T) / 2
    v6 = np.random.RandomState(22).standard_normal(nd)
    q6 = Quadratic(Q6, np.zeros(nd))
    res['six_packed'] = np.array(q6.device_problem().layout()['packed'])
    res['six_rows'] = np.array(q6.device_problem().dims()[2:])
    res['six_matvec'] = q6.device_problem().matvec(v6)
    res['six_ref'] = Q6 @ v6
    q6.release()
    for tag, (i, j) in (('a', (300, 1290)), ('b', (1100, 1101)), ('c', (520, 700)), ('d', (5, 260))):
        Qx = Q6.copy()
        Qx[i, j] = np.nextafter(Qx[i, j], np.inf)
        qx = Quadratic(Qx, np.zeros(nd))
        res[f'six_{tag}_packed'] = np.array(qx.device_problem().layout()['packed'])
        res[f'six_{tag}_matvec'] = qx.device_problem().matvec(v6)
        res[f'six_{tag}_ref'] = Qx @ v6
        qx.release()
    res['dense_matvec'] = dq.device_problem().matvec(vq)
    opt = ProjectedGradient(quad=dq, ub=np.ones(500), max_iter=40).minimize()
    res['dense_pg_x'] = opt.x
    # augmented-Lagrangian dual: same sharded product, everything else replicated
    from optiml_amd.opti.constrained import AugmentedLagrangianQuadratic
    from optiml_amd.opti.unconstrained.stochastic import AdaGrad, Adam
    quad = KernelQuadratic(X, -np.ones(n), 'svc', gaussian, y=y, rank_one=False)
    al = AugmentedLagrangianQuadratic(primal=quad, A=y, b=np.zeros(1), lb=np.zeros(n), ub=np.ones(n))
    opt = AdaGrad(f=al, x=np.random.RandomState(3).uniform(size=n), step_size=1., epochs=80).minimize()
    res['al_x'], res['al_dual'], res['al_f'] = opt.x, al.dual_x, opt.f_x
    al = AugmentedLagrangianQuadratic(primal=dq, lb=np.zeros(500), ub=np.ones(500), rho=2.)
    opt = Adam(f=al, x=np.random.RandomState(3).uniform(size=500), step_size=0.01, epochs=60,
               momentum_type='nesterov', momentum=0.5).minimize()
    res['dense_al_x'] = opt.x
    # streamed mode: row blocks of recomputed Gram tiles + all-gather
    sq = KernelQuadratic(X, -np.ones(n), 'svc', gaussian, y=y, storage='stream')
    res['stream_rows'] = np.array(sq.device_problem().dims()[2:])
    res['stream_matvec'] = sq.device_problem().matvec(v)
    opt = FrankWolfe(quad=sq, ub=np.ones(n), max_iter=30).minimize()
    res['stream_fw_x'] = opt.x
    # ActiveSet with conjugate-gradient restricted solves: the only ActiveSet that shards (no dense factor); squared
    # hinge dual of BASELINE config 5 (ub = +inf, x0 = 1) on a dense (packed symmetric) panel -> bit-identical for any world
    from optiml_amd.opti.constrained import ActiveSetCG
    Qs = Qd + np.eye(500) / 2
    opt = ActiveSetCG(quad=Quadratic(Qs, -np.ones(500)), ub=np.full(500, np.inf), x=np.ones(500), max_iter=400).minimize()
    res['ascg_x'], res['ascg_iter'], res['ascg_inner'] = opt.x, np.array(opt.iter), np.array(opt.inner_iters)
    res['ascg_status'] = np.array(opt.status == 'optimal')
    quad = KernelQuadratic(X, -np.ones(n), 'svc', gaussian, y=y, diag=0.5)
    opt = ActiveSetCG(quad=quad, ub=np.full(n, np.inf), x=np.ones(n), max_iter=60).minimize()
    res['ascg_kernel_x'], res['ascg_kernel_f'] = opt.x, np.array(opt.f_x)
    ms, cnt = ctx.profile_read(_lib.PROF_EXCH)
    comm.barrier()
    np.savez(os.path.join(outdir, f'rank{comm.rank}.npz'), **res)


def gpu_c4(outdir):
    """BASELINE config 4's shape (SVR eps-insensitive, poly(3, scale, 1), FrankWolfe) at n=20 000, d=128, on this rank's
    share of the symmetric panel, host exchange."""
    from optiml_amd import device
    from optiml_amd.datasets import make_regression
    from optiml_amd.ml.svm.kernels import PolyKernel
    from optiml_amd.opti import KernelQuadratic
    from optiml_amd.opti.constrained import FrankWolfe
    import time
    t0 = time.time()
    say = lambda what: print(f'[c4 rank {os.environ["RANK"]}] {time.time() - t0:7.2f}s {what}', file=sys.stderr, flush=True)
    comm = make_comm('torch')
    say('comm up')
    device.init_distributed(comm, exchange='host', device=0)
    say('context up')
    n, d = 20000, 128
    X, y = make_regression(n, d, seed=0)
    quad = KernelQuadratic(X, np.hstack((-y, y)) + 0.1, 'svr', PolyKernel(3, 'scale', 1.0))
    quad.device_problem()
    say('panel built')
    res = {'matvec': quad.device_problem().matvec(np.random.RandomState(2).standard_normal(2 * n))}
    say('panel built, one product done')
    hist = []
    cb = lambda o: hist.append(o.f_x)
    cb._bq_needs_state = False
    opt = FrankWolfe(quad=quad, ub=np.ones(2 * n), max_iter=25, callback=cb).minimize()
    res['fw_x'], res['fw_hist'] = opt.x, np.array(hist)
    say('25 FrankWolfe iterations done')
    comm.barrier()
    np.savez(os.path.join(outdir, f'rank{comm.rank}.npz'), **res)


def gpu_watchdog(outdir):
    """One rank, an RCCL communicator of its own, collectives bounded to 0.4 s.  Short waits pass; so does a 1.2 s wait with only
    this rank's own work ahead of it (a long factorisation must never cost the communicator: ADVICE r4); a collective that sits
    behind a 1.5 s occupation of the stream (bq_ctx_probe_stall(behind_collective): a lane spinning on the wall clock, it ends by
    itself — what a late peer looks like) trips the watchdog: the communicator is aborted, the call raises ERR_RCCL, later
    collectives fail at once, closing the context does not hang.  NOT exercised on this one-GPU box: ncclCommAbort ending an RCCL
    kernel that is really spinning on a peer (the collective here completes by itself once the occupation ends)."""
    import time
    from optiml_amd import _lib, device
    from optiml_amd.datasets import make_blobs
    from optiml_amd.dist import SocketComm
    from optiml_amd.ml.svm.kernels import gaussian
    from optiml_amd.opti import KernelQuadratic
    ctx = device.set_context(device.Context(comm=SocketComm(0, 1), exchange='rccl', device=0, collective_timeout=0.4))
    res = {'rccl_ranks': np.array(ctx.comm_info()['rccl_ranks'])}
    X, y = make_blobs(700, 12, seed=5)
    quad = KernelQuadratic(X, -np.ones(700), 'svc', gaussian, y=y)
    v = np.random.RandomState(2).standard_normal(700)
    res['matvec'] = quad.device_problem().matvec(v)            # a product with its (one-rank) all-gather: no false alarm
    ctx.probe_stall(100.0, behind_collective=True)              # a peer that is late by less than the limit: nothing happens
    res['matvec_again'] = quad.device_problem().matvec(v)
    t0 = time.perf_counter()
    ctx.probe_stall(1200.0)                                     # three times the limit, but no collective is outstanding
    res['own_work_s'] = np.array(time.perf_counter() - t0)
    res['matvec_third'] = quad.device_problem().matvec(v)
    t0 = time.perf_counter()
    try:
        ctx.probe_stall(1500.0, behind_collective=True)
        res['stall_error'] = np.array(0)
    except _lib.BcqpError as err:
        res['stall_error'], res['stall_msg'] = np.array(err.code), np.array(str(err))
    res['stall_s'] = np.array(time.perf_counter() - t0)
    t0 = time.perf_counter()
    try:
        quad.device_problem().matvec(v)
        res['after_error'] = np.array(0)
    except _lib.BcqpError as err:
        res['after_error'] = np.array(err.code)
    res['after_s'] = np.array(time.perf_counter() - t0)
    res['rccl_ranks_after'] = np.array(ctx.comm_info()['rccl_ranks'])
    t0 = time.perf_counter()
    quad.release()
    ctx.close()
    res['close_s'] = np.array(time.perf_counter() - t0)
    np.savez(os.path.join(outdir, 'rank0.npz'), **res)


def gpu_peer_leaves(outdir):
    """Two ranks on GPU 0, host exchange over the package's TCP communicator (5 s timeout).  Rank 1 meets a rank-local error
    after the first product and stops taking part (it leaves WITHOUT a non-zero exit: nothing outside kills the job); rank 0
    enters its second product and must come back with an error, not hang."""
    import time
    from optiml_amd import _lib, device
    from optiml_amd.datasets import make_blobs
    from optiml_amd.dist import SocketComm
    from optiml_amd.ml.svm.kernels import gaussian
    from optiml_amd.opti import KernelQuadratic
    rank, world = int(os.environ['RANK']), int(os.environ['WORLD_SIZE'])
    comm = SocketComm(rank, world, '127.0.0.1', int(os.environ['MASTER_PORT']) + 1, timeout=5.0)
    device.init_distributed(comm, exchange='host', device=0)
    X, y = make_blobs(700, 12, seed=5)
    quad = KernelQuadratic(X, -np.ones(700), 'svc', gaussian, y=y)
    v = np.random.RandomState(2).standard_normal(700)
    res = {'first': quad.device_problem().matvec(v)}
    if rank == 1:
        try:
            raise RuntimeError('rank-local failure before the second product')
        except RuntimeError as err:
            res['left_because'] = np.array(str(err))
        np.savez(os.path.join(outdir, f'rank{rank}.npz'), **res)
        comm.close()
        return
    t0 = time.perf_counter()
    try:
        quad.device_problem().matvec(v)
        res['second_error'] = np.array(0)
    except _lib.BcqpError as err:
        res['second_error'], res['second_msg'] = np.array(err.code), np.array(str(err))
    res['second_s'] = np.array(time.perf_counter() - t0)
    np.savez(os.path.join(outdir, f'rank{rank}.npz'), **res)


if __name__ == '__main__':
    import faulthandler
    faulthandler.dump_traceback_later(90, exit=False)   # a stuck rank shows where (its log is printed by the launcher)
    mode, outdir = sys.argv[1], sys.argv[2]
    if mode == 'cpu-torch':
        cpu_mode('torch', outdir)
    elif mode == 'cpu-fromenv-socket':
        cpu_mode('fromenv-socket', outdir)
    elif mode == 'cpu-socket':
        cpu_mode('socket', outdir)
    elif mode == 'gpu-host':
        gpu_mode('host', outdir)
    elif mode == 'gpu-rccl':
        gpu_mode('rccl', outdir)
    elif mode == 'gpu-host-c4':
        gpu_c4(outdir)
    elif mode == 'gpu-host-allreduce':
        gpu_mode('host', outdir, 'allreduce')
    elif mode == 'gpu-rccl-allreduce':
        gpu_mode('rccl', outdir, 'allreduce')
    elif mode == 'gpu-watchdog':
        gpu_watchdog(outdir)
    elif mode == 'gpu-peer-leaves':
        gpu_peer_leaves(outdir)
    else:
        raise SystemExit('unknown mode ' + mode)
