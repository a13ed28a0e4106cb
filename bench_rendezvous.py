"""Rendezvous over torch.distributed (gloo) for `bench.py` and the multi-process tests — plumbing OUTSIDE the product package
(`optiml_amd` imports no torch; its own communicators are `optiml_amd.dist.SocketComm` / `ThreadComm`).

`TorchComm` rides on an initialised torch.distributed process group (the launcher `python -m torch.distributed.run` sets
RANK / WORLD_SIZE / MASTER_*); `from_env` picks it when torch is importable (BQ_RENDEZVOUS=socket forces the package's TCP
communicator, BQ_RENDEZVOUS=torch insists on torch).  The data path is RCCL either way.
"""
import os
import time

import numpy as np

from optiml_amd.dist import SocketComm, _Base, block_size
from optiml_amd import dist as _pkg_dist


class TorchComm(_Base):
    def __init__(self, group=None):
        import torch.distributed as dist
        if not dist.is_initialized():
            raise RuntimeError('torch.distributed is not initialised')
        self._dist = dist
        self.group = group
        self.rank = dist.get_rank(group)
        self.world_size = dist.get_world_size(group)

    def broadcast_bytes(self, data, src=0):
        import torch
        n = len(data)
        t = torch.zeros(n, dtype=torch.uint8)
        if self.rank == src:
            t = torch.frombuffer(bytearray(data), dtype=torch.uint8).clone()
        self._dist.broadcast(t, src=src, group=self.group)
        return bytes(t.numpy().tobytes())

    def allgather_rows(self, buf, r0, r1):
        import torch
        dbg = os.environ.get('NCCL_DEBUG', '') in ('INFO', 'TRACE')
        t = [time.perf_counter()]
        n = buf.shape[0]
        blk = block_size(n, self.world_size)
        send = torch.zeros(blk, dtype=torch.float64)
        if r1 > r0:
            send[:r1 - r0] = torch.from_numpy(np.array(buf[r0:r1], copy=True))
        t.append(time.perf_counter())
        parts = [torch.zeros(blk, dtype=torch.float64) for _ in range(self.world_size)]
        self._dist.all_gather(parts, send, group=self.group)
        t.append(time.perf_counter())
        for r, part in enumerate(parts):
            b, e = self.rows_of(n, r)
            if e > b:
                buf[b:e] = part[:e - b].numpy()
        t.append(time.perf_counter())
        if dbg and self.rank == 0:
            import sys
            print('[allgather_rows] read %.3f ms, all_gather %.3f ms, write %.3f ms' % tuple(1e3 * (b - a) for a, b in zip(t, t[1:])),
                  file=sys.stderr, flush=True)

    def allreduce_sum(self, buf):
        import torch
        t = torch.from_numpy(np.array(buf, copy=True))
        self._dist.all_reduce(t, op=self._dist.ReduceOp.SUM, group=self.group)
        buf[:] = t.numpy()

    def barrier(self):
        self._dist.barrier(group=self.group)

    def max_float(self, x):
        import torch
        t = torch.tensor([float(x)], dtype=torch.float64)
        self._dist.all_reduce(t, op=self._dist.ReduceOp.MAX, group=self.group)
        return float(t.item())

    def close(self):
        if getattr(self, '_owns_group', False) and self._dist.is_initialized():
            self._dist.destroy_process_group()
            self._owns_group = False


def from_env(prefer_torch=True, timeout=600.0):
    """Communicator for the current launcher environment (RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT): torch.distributed
    (gloo) when torch is importable, otherwise — or with BQ_RENDEZVOUS=socket — optiml_amd.dist.SocketComm on MASTER_PORT + 33."""
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    if world == 1:
        return SocketComm(0, 1)
    mode = os.environ.get('BQ_RENDEZVOUS', '')
    if mode not in ('', 'torch', 'socket'):
        raise ValueError(f"BQ_RENDEZVOUS='{mode}' (use 'torch' or 'socket')")
    if (prefer_torch and mode != 'socket') or mode == 'torch':
        try:
            import datetime
            import torch.distributed as dist
            owns = not dist.is_initialized()
            if owns:
                dist.init_process_group(backend='gloo', rank=rank, world_size=world,
                                        timeout=datetime.timedelta(seconds=timeout))
            comm = TorchComm()
            comm._owns_group = owns
            return comm
        except ImportError:
            if mode == 'torch':
                raise
    return _pkg_dist.from_env(timeout=timeout)
