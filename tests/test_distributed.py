"""Multi-process coverage of the N > 1 path.

CPU (always run): world_size-2 gloo and socket communicators — rendezvous, the 128-byte id broadcast, the row-block
all-gather that completes a sharded Q v, barrier / max — with the CPU oracle's Hessian standing in for the panel.
GPU (-m gpu): two and three ranks on GPU 0 (host exchange) must reproduce the single-rank solver iterates BIT FOR BIT —
row-block panels (all-gather of slices) and symmetric kernel panels (all-gather of the per-segment partial vectors, added
in segment order) alike; a single-rank RCCL context exercises ncclCommInitRank / ncclAllGather / ncclAllReduce / destroy.
"""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

from conftest import set_hooks, hooks_env, hook_value

HERE = os.path.dirname(os.path.abspath(__file__))
WORKER = os.path.join(HERE, '_dist_worker.py')


def _free_port():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def _launch(mode, world, outdir, timeout=300, extra_env=None):
    """One process per rank; each rank's output goes to OUTDIR/rank<r>.log and is shown when a rank fails or hangs."""
    import time
    port = _free_port()
    outdir = str(outdir)
    os.makedirs(outdir, exist_ok=True)
    procs, logs = [], []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK='0', WORLD_SIZE=str(world), MASTER_ADDR='127.0.0.1',
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY='0', OMP_NUM_THREADS='2', **(extra_env or {}))
        logs.append(open(os.path.join(outdir, f'rank{r}.log'), 'w'))
        procs.append(subprocess.Popen([sys.executable, WORKER, mode, outdir], env=env, stdout=logs[-1],
                                      stderr=subprocess.STDOUT))

    def tails():
        for f in logs:
            f.close()
        return '\n'.join(f'---- rank {r} ----\n' + open(os.path.join(outdir, f'rank{r}.log')).read()[-3000:] for r in range(world))

    deadline = time.time() + timeout
    while any(p.poll() is None for p in procs):
        if time.time() > deadline or any(p.poll() not in (None, 0) for p in procs):
            for q in procs:
                if q.poll() is None:
                    q.kill()
            for q in procs:
                q.wait()
            raise AssertionError(f'{mode} x{world}: a rank failed or did not finish within {timeout} s\n' + tails())
        time.sleep(0.05)
    out = tails()
    for r, p in enumerate(procs):
        assert p.returncode == 0, f'rank {r} failed:\n{out}'
    return [np.load(os.path.join(outdir, f'rank{r}.npz')) for r in range(world)]


@pytest.mark.parametrize('kind', ['cpu-torch', 'cpu-socket', 'cpu-fromenv-socket'])
def test_two_rank_row_block_exchange_cpu(tmp_path, kind):
    res = _launch(kind, 2, tmp_path)
    for r in res:
        assert float(r['tmax']) == 2.0
        for n in (300, 1000):
            # the gathered vector is exactly the concatenation of the per-rank row-block products
            assert np.array_equal(r[f'gathered_{n}'], r[f'full_{n}'])
    for n in (300, 1000):
        assert np.array_equal(res[0][f'gathered_{n}'], res[1][f'gathered_{n}'])


def run_thread_ranks(world, body, timeout=600.0):
    """body(member) on `world` threads of THIS process, one ThreadComm member each; results in rank order.  A rank that raises
    aborts the group's barrier so that the others fail too instead of waiting for it."""
    import threading
    from optiml_amd.dist import ThreadComm
    members = ThreadComm.group(world, timeout=timeout)
    res, err = [None] * world, []

    def run(k):
        try:
            res[k] = body(members[k])
        except BaseException as exc:  # noqa: BLE001
            err.append((k, exc))
            members[k].abort()

    threads = [threading.Thread(target=run, args=(k,), name=f'rank{k}') for k in range(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    if err:
        first = [e for e in err if not isinstance(e[1], threading.BrokenBarrierError)] or err
        raise AssertionError(f'rank {first[0][0]} of {world} failed: {first[0][1]!r}') from first[0][1]
    return res


def test_thread_ranks_primitives():
    """ThreadComm: the communicator contract with the ranks as threads of one process (what the full-size 4- and 8-way
    partition tests run on a one-GPU box)."""
    def body(c):
        assert c.broadcast_bytes(b'abc' if c.rank == 0 else b'', src=0) == b'abc'
        n = 1000
        r0, r1 = c.rows_of(n)
        buf = np.zeros(n)
        buf[r0:r1] = np.arange(r0, r1) + 0.5
        c.allgather_rows(buf, r0, r1)
        assert np.array_equal(buf, np.arange(n) + 0.5)
        v = np.full(7, float(c.rank + 1))
        c.allreduce_sum(v)
        assert np.array_equal(v, np.full(7, sum(range(1, c.world_size + 1))))
        c.barrier()
        assert c.max_float(c.rank * 1.5) == 1.5 * (c.world_size - 1)
        assert c.allgather_float(10.0 + c.rank) == [10.0 + k for k in range(c.world_size)]
        return c.rank

    for world in (1, 4, 8):
        assert run_thread_ranks(world, body, timeout=30.0) == list(range(world))

    def bad(c):
        if c.rank == 2:
            raise ValueError('boom')
        c.barrier()

    with pytest.raises(AssertionError, match='rank 2 of 4 failed.*boom'):
        run_thread_ranks(4, bad, timeout=30.0)


SYM_KEYS = ('matvec', 'gram_matvec', 'pg_x', 'pg_hist', 'fw_x', 'fw_f', 'al_x', 'al_dual', 'al_f', 'ascg_kernel_x',
            'ascg_kernel_f')
# dense Hessians: a Q == Q' is packed like a kernel panel since round 6 (segments + all-gather of segment partials), any other Q keeps
# row blocks (all-gather of disjoint slices)
DENSE_SYM_KEYS = ('dense_matvec', 'dense_pg_x', 'dense_al_x', 'ascg_x', 'ascg_iter', 'ascg_inner', 'six_matvec')
DENSE_ROW_KEYS = ('rowsq_matvec', 'rowsq_pg_x', 'hid_matvec', 'six_a_matvec', 'six_b_matvec', 'six_c_matvec', 'six_d_matvec')
ROW_KEYS = DENSE_SYM_KEYS + DENSE_ROW_KEYS
# streamed kernel problems use the symmetric segment exchange since round 2b (every lower-triangle tile formed once)
SYM_KEYS = SYM_KEYS + ('stream_matvec', 'stream_fw_x')


@pytest.fixture(scope='module')
def one_rank(tmp_path_factory):
    return _launch('gpu-host', 1, tmp_path_factory.mktemp('w1'))[0]


@pytest.mark.gpu
def test_two_ranks_reproduce_single_rank_bitwise(tmp_path, one_rank):
    one = one_rank
    two = _launch('gpu-host', 2, tmp_path / 'w2')
    # kernel panels: symmetric tile storage, balanced triangular partition (700 rows = 3 tile rows -> 2 + 1)
    assert tuple(two[0]['rows']) == (0, 512) and tuple(two[1]['rows']) == (512, 700)
    # a dense Q == Q': packed like a kernel panel (500 rows = 2 tile rows -> 1 + 1); any other dense Q: equal 128-aligned row
    # blocks; a Q with one element one ulp off its mirror image, seen by rank 1 only: both ranks fall back to row blocks
    assert tuple(two[0]['dense_rows']) == (0, 256) and tuple(two[1]['dense_rows']) == (256, 500)
    assert tuple(two[0]['rowsq_rows']) == (0, 256) and tuple(two[1]['rowsq_rows']) == (256, 500)
    assert tuple(two[0]['six_rows']) == (0, 1024) and tuple(two[1]['six_rows']) == (1024, 1300)   # six tile rows: 4 + 2
    for r in two + [one]:
        assert bool(r['dense_packed']) and not bool(r['rowsq_packed']) and not bool(r['hid_packed'])
        assert bool(r['six_packed']) and not any(bool(r[f'six_{t}_packed']) for t in 'abcd')
        for t in ('', 'a_', 'b_', 'c_', 'd_'):
            np.testing.assert_allclose(r[f'six_{t}matvec'], r[f'six_{t}ref'], rtol=1e-12, atol=1e-12)
        np.testing.assert_allclose(r['rowsq_matvec'], r['rowsq_ref'], rtol=1e-12, atol=1e-12)
        np.testing.assert_allclose(r['hid_matvec'], r['hid_ref'], rtol=1e-12, atol=1e-12)
    # streamed kernel problems: the segment partition of the tile rows, like the resident symmetric panels
    assert tuple(two[0]['stream_rows']) == (0, 512) and tuple(two[1]['stream_rows']) == (512, 700)
    np.testing.assert_allclose(one['stream_matvec'], one['matvec'], rtol=1e-11, atol=1e-11)
    for r in two:
        assert str(r['sym_exchange']) == 'gather'
        # row-block panels end in an all-gather of disjoint slices; symmetric tile panels in an all-gather of the per-segment
        # partial vectors, added in segment order on every rank: both bit-identical for any rank count
        for key in ROW_KEYS + SYM_KEYS:
            assert np.array_equal(r[key], one[key]), key
        assert bool(r['ascg_status']) and int(r['ascg_inner']) > 0


@pytest.mark.gpu
def test_rccl_context_single_rank(tmp_path, one_rank):
    got = _launch('gpu-rccl', 1, tmp_path / 'rccl')[0]
    assert int(got['rccl_ranks']) == 1
    for key in ('matvec', 'pg_x', 'pg_hist', 'fw_x', 'dense_matvec', 'dense_pg_x', 'al_x', 'dense_al_x'):
        assert np.array_equal(got[key], one_rank[key]), key
    # the all-reduce variant of the symmetric exchange (ncclAllReduce on one rank)
    got = _launch('gpu-rccl-allreduce', 1, tmp_path / 'rccl_ar')[0]
    assert str(got['sym_exchange']) == 'allreduce'
    for key in ('matvec', 'pg_x', 'fw_x'):
        assert np.array_equal(got[key], one_rank[key]), key


@pytest.mark.gpu
def test_three_ranks_uneven_partitions(tmp_path, one_rank):
    """World size 3: uneven partitions (700 rows = 3 tile rows -> segments 2 + 3 + 3 of the canonical 8 = tile rows
    2 + 0 + 1; 500 dense rows -> 256 + 244 + 0) — still bit-identical to one rank."""
    three = _launch('gpu-host', 3, tmp_path / 'w3')
    assert [tuple(r['rows']) for r in three] == [(0, 512), (512, 512), (512, 700)]
    for r in three:
        assert bool(r['six_packed']) and not any(bool(r[f'six_{t}_packed']) for t in 'abcd') and not bool(r['hid_packed'])
        for key in ROW_KEYS + SYM_KEYS:
            assert np.array_equal(r[key], one_rank[key]), key


@pytest.mark.gpu
def test_four_ranks_two_segments_each(tmp_path, one_rank):
    """World size 4 — the C4 rank count: two canonical segments per rank; 700 rows = 3 tile rows -> tile rows 2 + 0 + 1 + 0
    (two ranks own no tile row and contribute zero vectors).  Still bit-identical to one rank."""
    four = _launch('gpu-host', 4, tmp_path / 'w4')
    assert [tuple(r['rows']) for r in four] == [(0, 512), (512, 512), (512, 700), (700, 700)]
    for r in four:
        assert bool(r['six_packed']) and not any(bool(r[f'six_{t}_packed']) for t in 'abcd') and not bool(r['hid_packed'])
        for key in ROW_KEYS + SYM_KEYS:
            assert np.array_equal(r[key], one_rank[key]), key


@pytest.mark.gpu
@pytest.mark.parametrize('n', [257, 2100])
def test_other_sizes_split_the_same_way(tmp_path, n):
    """257 rows = two tile rows, the second with ONE row (two ranks: 1 + 1 tile rows; three ranks: one of them owns nothing);
    2100 rows = nine tile rows (segments of 3 / 1 / 1 / 1 / 1 / 1 / 1 / 0 tile rows — an empty canonical segment).  Symmetric,
    streamed and row-block products and every solve on them: one rank = two ranks = three ranks, bit for bit."""
    env = {'BQ_TEST_DIST_N': str(n)}
    one = _launch('gpu-host', 1, tmp_path / 'w1', extra_env=env)[0]
    for world in (2, 3):
        for r in _launch('gpu-host', world, tmp_path / f'w{world}', extra_env=env):
            for key in ROW_KEYS + SYM_KEYS:
                assert np.array_equal(r[key], one[key]), (world, key)


@pytest.mark.gpu
def test_allreduce_variant_of_the_symmetric_exchange(tmp_path, one_rank):
    """sym_exchange='allreduce' (rank partials added by the transport, the collective BASELINE's north star names): the
    same values up to the association of the rank sum."""
    two = _launch('gpu-host-allreduce', 2, tmp_path / 'w2ar')
    for r in two:
        assert str(r['sym_exchange']) == 'allreduce'
        for key in DENSE_ROW_KEYS:   # row blocks end in an all-gather of disjoint slices whatever the symmetric exchange is
            assert np.array_equal(r[key], one_rank[key]), key
        for key in DENSE_SYM_KEYS:   # packed dense Hessians take the symmetric exchange: the rank sum's association is the transport's
            np.testing.assert_allclose(r[key], one_rank[key], rtol=1e-9, atol=1e-11, err_msg=key)
        for key in ('matvec', 'gram_matvec', 'pg_hist', 'fw_f', 'al_f', 'stream_matvec'):
            np.testing.assert_allclose(r[key], one_rank[key], rtol=1e-12, atol=1e-12, err_msg=key)
        for key in ('pg_x', 'fw_x', 'al_x', 'al_dual', 'stream_fw_x'):
            np.testing.assert_allclose(r[key], one_rank[key], rtol=1e-9, atol=1e-11, err_msg=key)
    for key in ('matvec', 'pg_x', 'fw_x'):
        assert np.array_equal(two[0][key], two[1][key]), key


@pytest.mark.gpu
@pytest.mark.parametrize('world', [2, 4, 8])
def test_share_contexts_hold_one_ranks_part_of_the_product(world):
    """bq_ctx_create_share: rank k's share of a `world`-way partition with no transport (what bench.py --emulate-shares times).
    Its product is that rank's contribution alone: the shares of all ranks add up to the one-rank product (to rounding — the
    sum over shares is not the canonical segment order), every share owns the rows bq_sym_row_block names, and a share of a
    one-way partition is the plain context."""
    from optiml_amd import device
    from optiml_amd.datasets import make_blobs
    from optiml_amd.ml.svm.kernels import gaussian
    from optiml_amd.opti import KernelQuadratic
    n, d = 3000, 16
    X, y = make_blobs(n, d, seed=5)
    v = np.random.RandomState(1).standard_normal(n)
    ctx = device.Context(device=0)
    quad = KernelQuadratic(X, -np.ones(n), 'svc', gaussian, y=y)
    full = quad.device_problem(ctx).matvec(v)
    quad.release()
    ctx.close()
    acc = np.zeros(n)
    for k in range(world):
        ctx = device.Context(device=0, share=(k, world))
        assert (ctx.rank, ctx.world, ctx.exchange) == (k, world, 'share') and ctx.comm_info()['kind'] == 'share'
        quad = KernelQuadratic(X, -np.ones(n), 'svc', gaussian, y=y)
        dev = quad.device_problem(ctx)
        assert dev.dims()[2:] == device.row_block(n, k, world, symmetric=True)
        part = dev.matvec(v)
        assert np.array_equal(part, dev.matvec(v))
        acc += part
        quad.release()
        ctx.close()
    np.testing.assert_allclose(acc, full, rtol=1e-12, atol=1e-12 * np.abs(full).max())


@pytest.mark.gpu
def test_rows_per_step_variants_are_bit_identical(tmp_path, one_rank):
    """The symmetric product streams 4 or 8 rows per step depending on how full the last round of workgroups of its grid is
    (bq_symv.hip).  The choice must not change a bit — otherwise a rank count that changes the grid would change the iterates:
    products, PG / FW / augmented-Lagrangian / ActiveSetCG runs forced to 4 and to 8 rows per step equal the default run."""
    for rows in ('4', '8'):
        res = _launch('gpu-host', 1, tmp_path / f'rows{rows}', extra_env=hooks_env(rows_per_step=rows))[0]
        for key in ('matvec', 'gram_matvec', 'pg_x', 'pg_hist', 'fw_x', 'fw_f', 'al_x', 'al_f', 'ascg_kernel_x', 'ascg_kernel_f'):
            assert np.array_equal(res[key], one_rank[key]), (rows, key)


@pytest.mark.gpu
def test_order2_remainder_of_the_preconditioner_is_replicated_bitwise(tmp_path):
    """ActiveSetCG's preconditioner is sharded by sample segments: the implicit order-2 remainder (csrc/bq_as_pc2.hip: two MFMA
    products with a split-K sum, a power-iteration spectrum bound, a Chebyshev polynomial — a rank forms the moment slices and the
    x'Mx values of its own samples) and, since round 6, the explicit model's two passes over the features too (csrc/bq_as_pc.hip:
    t = Phi' D^-1 r over a rank's own sample blocks, z for its own samples); the per-segment sums of M, of t and of Phi_top' y and
    the vectors v and z are gathered and added / unpacked in segment order — six collectives per application.  Forced on (hook
    as_cg_pc_class=3), one, two and three ranks give the same bits — and the same outer trajectory as without the remainder.
    (The explicit model alone, two collectives per application, is what every `ascg_kernel_*` key of the tests above runs.)"""
    env = hooks_env(as_cg_pc_class=3)
    one = _launch('gpu-host', 1, tmp_path / 'one', extra_env=env)[0]
    two = _launch('gpu-host', 2, tmp_path / 'two', extra_env=env)
    three = _launch('gpu-host', 3, tmp_path / 'three', extra_env=env)     # uneven runs of the eight sample segments: 2 / 3 / 3
    plain = _launch('gpu-host', 1, tmp_path / 'plain', extra_env=hooks_env(as_cg_pc_class='2'))[0]
    for key in ('ascg_kernel_x', 'ascg_kernel_f'):
        for r in two + three:
            assert np.array_equal(r[key], one[key]), key
    np.testing.assert_allclose(one['ascg_kernel_x'], plain['ascg_kernel_x'], rtol=1e-7, atol=1e-9)


@pytest.mark.gpu
def test_collective_watchdog_aborts_a_wait_that_outlasts_the_timeout(tmp_path):
    """bq_ctx_set_collective_timeout on a one-rank RCCL context (a child process: an aborted communicator is the end of its
    context): normal products, a collective that is late by less than the limit and a LONG wait for this rank's own work all pass;
    a collective behind a 1.5 s occupation against a 0.4 s limit makes the watchdog abort the communicator — ERR_RCCL from the call
    in progress, ERR_RCCL from the next collective, and closing the context returns.  Only error codes and generous upper bounds
    are asserted (the child also initialises HIP and RCCL on a shared box: ADVICE r4); ncclCommAbort ending a collective that is
    really waiting for a peer cannot be exercised with one GPU."""
    from optiml_amd import _lib
    got = _launch('gpu-watchdog', 1, tmp_path, timeout=400)[0]
    assert int(got['rccl_ranks']) == 1
    assert np.array_equal(got['matvec'], got['matvec_again']) and np.array_equal(got['matvec'], got['matvec_third'])
    assert float(got['own_work_s']) >= 1.0                       # the whole occupation was waited for, and nothing fired
    assert int(got['stall_error']) == _lib.ERR_RCCL and 'did not complete within' in str(got['stall_msg'])
    assert float(got['stall_s']) < 0.4 + 1.5 + 10.0              # the limit + the occupation's own length + slack: it did not hang
    assert int(got['after_error']) == _lib.ERR_RCCL and float(got['after_s']) < 30.0
    assert int(got['rccl_ranks_after']) == 0 and float(got['close_s']) < 30.0


@pytest.mark.gpu
def test_a_rank_that_leaves_does_not_hang_its_peer(tmp_path):
    """a rank-local failure mid-run on one of two ranks (host exchange, 5 s communicator timeout); the rank
    that goes on must get an error from its next product in bounded time, with nothing outside to kill it."""
    from optiml_amd import _lib
    r0, r1 = _launch('gpu-peer-leaves', 2, tmp_path, timeout=120)
    assert 'rank-local failure' in str(r1['left_because'])
    assert np.array_equal(r0['first'], r1['first'])
    assert int(r0['second_error']) == _lib.ERR_RCCL, r0['second_error']
    assert float(r0['second_s']) < 15.0
