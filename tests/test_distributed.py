"""Multi-process coverage of the N > 1 path.

CPU (always run): world_size-2 gloo and socket communicators — rendezvous, the 128-byte id broadcast, the row-block
all-gather that completes a sharded Q v, barrier / max — with the CPU oracle's Hessian standing in for the panel.
GPU (-m gpu): two ranks on GPU 0 with row-block Gram panels and the host exchange must reproduce the single-rank
solver iterates BIT FOR BIT; a single-rank RCCL context exercises ncclCommInitRank / ncclAllGather / destroy.
"""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
WORKER = os.path.join(HERE, '_dist_worker.py')


def _free_port():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def _launch(mode, world, outdir, timeout=300):
    port = _free_port()
    os.makedirs(outdir, exist_ok=True)
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK='0', WORLD_SIZE=str(world), MASTER_ADDR='127.0.0.1',
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY='0', OMP_NUM_THREADS='2')
        procs.append(subprocess.Popen([sys.executable, WORKER, mode, str(outdir)], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    outs = []
    for p in procs:
        try:
            out, _ = p.communicate(timeout=timeout)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        outs.append(out)
    for r, (p, out) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, f'rank {r} failed:\n{out}'
    return [np.load(os.path.join(outdir, f'rank{r}.npz')) for r in range(world)]


@pytest.mark.parametrize('kind', ['cpu-torch', 'cpu-socket'])
def test_two_rank_row_block_exchange_cpu(tmp_path, kind):
    res = _launch(kind, 2, tmp_path)
    for r in res:
        assert float(r['tmax']) == 2.0
        for n in (300, 1000):
            # the gathered vector is exactly the concatenation of the per-rank row-block products
            assert np.array_equal(r[f'gathered_{n}'], r[f'full_{n}'])
    for n in (300, 1000):
        assert np.array_equal(res[0][f'gathered_{n}'], res[1][f'gathered_{n}'])


@pytest.mark.gpu
def test_two_ranks_reproduce_single_rank_bitwise(tmp_path):
    one = _launch('gpu-host', 1, tmp_path / 'w1')[0]
    two = _launch('gpu-host', 2, tmp_path / 'w2')
    assert tuple(two[0]['rows']) == (0, 384) and tuple(two[1]['rows']) == (384, 700)
    for r in two:
        for key in ('matvec', 'gram_matvec', 'pg_x', 'pg_hist', 'fw_x', 'fw_f'):
            assert np.array_equal(r[key], one[key]), key


@pytest.mark.gpu
def test_rccl_context_single_rank(tmp_path):
    ref = _launch('gpu-host', 1, tmp_path / 'ref')[0]
    got = _launch('gpu-rccl', 1, tmp_path / 'rccl')[0]
    for key in ('matvec', 'pg_x', 'pg_hist', 'fw_x'):
        assert np.array_equal(got[key], ref[key]), key
