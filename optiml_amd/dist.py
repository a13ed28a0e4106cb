"""Host-side communicators used to bootstrap multi-GPU runs (rendezvous only — the data path is RCCL).

A communicator provides: rank, world_size, broadcast_bytes(data, src), allgather_rows(buf, r0, r1),
allreduce_sum(buf), barrier(), max_float(x).  `TorchComm` rides on an initialised torch.distributed process group (gloo on the host; the
launcher `python -m torch.distributed.run` sets RANK/WORLD_SIZE/MASTER_*); `SocketComm` is a dependency-free
TCP star for environments without torch.  torch is plumbing here, never on the compute path.
"""
import os
import pickle
import socket
import struct
import time

import numpy as np

__all__ = ['TorchComm', 'SocketComm', 'from_env', 'block_size']


def block_size(n, world):
    """Per-rank row-block size: ceil(n / world) rounded up to 128 (matches bq_row_block)."""
    per = max(1, -(-n // world))
    return -(-per // 128) * 128


class _Base:
    rank = 0
    world_size = 1

    def rows_of(self, n, rank=None):
        blk = block_size(n, self.world_size)
        r = self.rank if rank is None else rank
        b = min(n, r * blk)
        return b, min(n, b + blk)


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
        n = buf.shape[0]
        blk = block_size(n, self.world_size)
        send = torch.zeros(blk, dtype=torch.float64)
        if r1 > r0:
            send[:r1 - r0] = torch.from_numpy(np.array(buf[r0:r1], copy=True))
        parts = [torch.zeros(blk, dtype=torch.float64) for _ in range(self.world_size)]
        self._dist.all_gather(parts, send, group=self.group)
        for r, part in enumerate(parts):
            b, e = self.rows_of(n, r)
            if e > b:
                buf[b:e] = part[:e - b].numpy()

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


def _send_msg(sock, obj):
    data = pickle.dumps(obj, protocol=4)
    sock.sendall(struct.pack('!Q', len(data)) + data)


def _recv_exact(sock, n):
    chunks = []
    while n:
        c = sock.recv(min(n, 1 << 20))
        if not c:
            raise ConnectionError('peer closed the rendezvous socket')
        chunks.append(c)
        n -= len(c)
    return b''.join(chunks)


def _recv_msg(sock):
    (n,) = struct.unpack('!Q', _recv_exact(sock, 8))
    return pickle.loads(_recv_exact(sock, n))


class SocketComm(_Base):
    """TCP star through rank 0 (gather + scatter); small control-plane messages only."""

    def __init__(self, rank, world_size, addr='127.0.0.1', port=29533, timeout=120.0):
        self.rank, self.world_size = rank, world_size
        self._peers = []
        self._sock = None
        if world_size == 1:
            return
        if rank == 0:
            srv = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
            srv.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
            srv.bind((addr, port))
            srv.listen(world_size)
            srv.settimeout(timeout)
            peers = {}
            while len(peers) < world_size - 1:
                conn, _ = srv.accept()
                conn.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
                peers[_recv_msg(conn)] = conn
            srv.close()
            self._peers = [peers[r] for r in range(1, world_size)]
        else:
            deadline = time.time() + timeout
            while True:
                try:
                    s = socket.create_connection((addr, port), timeout=5.0)
                    break
                except OSError:
                    if time.time() > deadline:
                        raise
                    time.sleep(0.05)
            s.settimeout(timeout)
            s.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
            _send_msg(s, rank)
            self._sock = s

    def _gather(self, obj):
        if self.world_size == 1:
            return [obj]
        if self.rank == 0:
            return [obj] + [_recv_msg(p) for p in self._peers]
        _send_msg(self._sock, obj)
        return None

    def _bcast(self, obj):
        if self.world_size == 1:
            return obj
        if self.rank == 0:
            for p in self._peers:
                _send_msg(p, obj)
            return obj
        return _recv_msg(self._sock)

    def broadcast_bytes(self, data, src=0):
        if src != 0:
            raise ValueError('SocketComm broadcasts from rank 0 only')
        return self._bcast(bytes(data) if self.rank == 0 else None)

    def allgather_rows(self, buf, r0, r1):
        parts = self._gather((r0, r1, np.array(buf[r0:r1], copy=True)))
        parts = self._bcast(parts)
        for b, e, arr in parts:
            if e > b:
                buf[b:e] = arr

    def allreduce_sum(self, buf):
        parts = self._gather(np.array(buf, copy=True))
        total = None
        if parts is not None:
            total = parts[0].copy()
            for p in parts[1:]:      # fixed rank order -> deterministic
                total += p
        buf[:] = self._bcast(total)

    def barrier(self):
        self._bcast(self._gather(None) and None)

    def max_float(self, x):
        vals = self._gather(float(x))
        return self._bcast(max(vals) if vals is not None else None)

    def close(self):
        for p in self._peers:
            p.close()
        if self._sock:
            self._sock.close()


def from_env(prefer_torch=True):
    """Communicator for the current launcher environment (RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT)."""
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    if world == 1:
        return SocketComm(0, 1)
    if prefer_torch:
        try:
            import torch.distributed as dist
            if not dist.is_initialized():
                dist.init_process_group(backend='gloo', rank=rank, world_size=world)
            return TorchComm()
        except ImportError:
            pass
    addr = os.environ.get('MASTER_ADDR', '127.0.0.1')
    port = int(os.environ.get('MASTER_PORT', '29500')) + 33
    return SocketComm(rank, world, addr, port)
