"""Host-side communicators used to bootstrap multi-GPU runs (rendezvous only — the data path is RCCL).

A communicator provides: rank, world_size, broadcast_bytes(data, src), allgather_rows(buf, r0, r1),
allreduce_sum(buf), barrier(), max_float(x).  `SocketComm` is a dependency-free TCP star (loopback by default, HMAC hello handshake,
raw length-bounded frames); `ThreadComm` joins the threads of one process.  Nothing in this package imports torch: the communicator that
rides on a torch.distributed (gloo) process group — what `bench.py` and the tests use under `python -m torch.distributed.run` — lives
outside it, in `bench_rendezvous.py` at the repository root (round 5).
"""
import hashlib
import hmac
import os
import socket
import struct
import time

import numpy as np

__all__ = ['SocketComm', 'ThreadComm', 'from_env', 'block_size']


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

    def close(self):
        pass

    def allgather_float(self, x):
        """[x of rank 0, x of rank 1, ...] on every rank (control plane: one max per rank)."""
        return [self.max_float(float(x) if self.rank == k else float('-inf')) for k in range(self.world_size)]


# Wire format of SocketComm: a fixed 16-byte header (magic, kind, payload length) followed by raw bytes — byte strings and
# float64 arrays only, length-bounded; nothing received is ever unpickled or evaluated.
_MAGIC = b'BQRV'
_HDR = struct.Struct('!4sIQ')
_K_NONE, _K_BYTES, _K_F64, _K_ROWS, _K_HELLO = range(5)
_MAX_MSG = 1 << 31


def _send_msg(sock, kind, payload=b''):
    if len(payload) > _MAX_MSG:
        raise ValueError('rendezvous message too large')
    sock.sendall(_HDR.pack(_MAGIC, kind, len(payload)) + payload)


def _recv_exact(sock, n):
    chunks = []
    while n:
        c = sock.recv(min(n, 1 << 20))
        if not c:
            raise ConnectionError('peer closed the rendezvous socket')
        chunks.append(c)
        n -= len(c)
    return b''.join(chunks)


def _recv_msg(sock, expect=None):
    magic, kind, n = _HDR.unpack(_recv_exact(sock, _HDR.size))
    if magic != _MAGIC or n > _MAX_MSG or (expect is not None and kind not in expect):
        raise ConnectionError('malformed rendezvous message')
    return kind, _recv_exact(sock, n)


def _secret(port):
    """Shared secret of the hello handshake: BQ_RENDEZVOUS_SECRET, or a token of (port, uid) so that unrelated jobs of one
    host do not cross-connect.  Set the variable on every rank when the rendezvous port is reachable by others."""
    tok = os.environ.get('BQ_RENDEZVOUS_SECRET')
    if tok is None:
        tok = 'bq-%d-%d' % (port, os.getuid() if hasattr(os, 'getuid') else 0)
    return hashlib.sha256(tok.encode()).digest()


class SocketComm(_Base):
    """TCP star through rank 0 (gather + scatter); small control-plane messages only.  Rank 0 binds the given address
    (loopback by default), admits a peer only after an HMAC handshake on the shared secret, and exchanges nothing but
    length-bounded byte strings / float64 arrays."""

    def __init__(self, rank, world_size, addr='127.0.0.1', port=29533, timeout=120.0):
        self.rank, self.world_size = rank, world_size
        self._peers = []
        self._sock = None
        if world_size == 1:
            return
        key = _secret(port)
        if rank == 0:
            srv = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
            srv.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
            srv.bind((addr, port))
            srv.listen(world_size)
            srv.settimeout(timeout)
            peers = {}
            while len(peers) < world_size - 1:
                conn, _ = srv.accept()
                conn.settimeout(timeout)
                conn.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
                try:   # challenge / response: the peer proves it holds the secret before anything else is read from it
                    nonce = os.urandom(16)
                    _send_msg(conn, _K_HELLO, nonce)
                    _, reply = _recv_msg(conn, (_K_HELLO,))
                    r, mac = struct.unpack('!I', reply[:4])[0], reply[4:]
                    good = hmac.compare_digest(mac, hmac.new(key, nonce + reply[:4], hashlib.sha256).digest())
                    if not good or not (0 < r < world_size) or r in peers:
                        raise ConnectionError('rendezvous handshake refused')
                    peers[r] = conn
                except (ConnectionError, struct.error, socket.timeout):
                    conn.close()
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
            _, nonce = _recv_msg(s, (_K_HELLO,))
            me = struct.pack('!I', rank)
            _send_msg(s, _K_HELLO, me + hmac.new(key, nonce + me, hashlib.sha256).digest())
            self._sock = s

    def _gather(self, kind, payload):
        """rank 0: list of every rank's payload (own first); other ranks: None"""
        if self.world_size == 1:
            return [payload]
        if self.rank == 0:
            return [payload] + [_recv_msg(p, (kind,))[1] for p in self._peers]
        _send_msg(self._sock, kind, payload)
        return None

    def _bcast(self, kind, payload):
        if self.world_size == 1:
            return payload
        if self.rank == 0:
            for p in self._peers:
                _send_msg(p, kind, payload)
            return payload
        return _recv_msg(self._sock, (kind,))[1]

    def broadcast_bytes(self, data, src=0):
        if src != 0:
            raise ValueError('SocketComm broadcasts from rank 0 only')
        return self._bcast(_K_BYTES, bytes(data) if self.rank == 0 else b'')

    def allgather_rows(self, buf, r0, r1):
        mine = struct.pack('!qq', r0, r1) + np.ascontiguousarray(buf[r0:r1], dtype=np.float64).tobytes()
        parts = self._gather(_K_ROWS, mine)
        blob = self._bcast(_K_ROWS, b''.join(struct.pack('!Q', len(p)) + p for p in parts) if parts is not None else b'')
        off = 0
        while off < len(blob):
            (ln,) = struct.unpack_from('!Q', blob, off)
            b, e = struct.unpack_from('!qq', blob, off + 8)
            if not (0 <= b <= e <= buf.shape[0]) or ln != 16 + 8 * (e - b):
                raise ConnectionError('malformed row block in the rendezvous exchange')
            if e > b:
                buf[b:e] = np.frombuffer(blob, dtype=np.float64, count=e - b, offset=off + 24)
            off += 8 + ln

    def allreduce_sum(self, buf):
        parts = self._gather(_K_F64, np.ascontiguousarray(buf, dtype=np.float64).tobytes())
        total = b''
        if parts is not None:
            acc = np.frombuffer(parts[0], dtype=np.float64).copy()
            for p in parts[1:]:      # fixed rank order -> deterministic
                if len(p) != acc.nbytes:
                    raise ConnectionError('length mismatch in the rendezvous all-reduce')
                acc += np.frombuffer(p, dtype=np.float64)
            total = acc.tobytes()
        out = self._bcast(_K_F64, total)
        if len(out) != 8 * buf.shape[0]:
            raise ConnectionError('length mismatch in the rendezvous all-reduce')
        buf[:] = np.frombuffer(out, dtype=np.float64)

    def barrier(self):
        self._gather(_K_NONE, b'')
        self._bcast(_K_NONE, b'')

    def max_float(self, x):
        vals = self._gather(_K_F64, struct.pack('!d', float(x)))
        out = self._bcast(_K_F64, struct.pack('!d', max(struct.unpack('!d', v)[0] for v in vals)) if vals is not None else b'')
        return struct.unpack('!d', out)[0]

    def close(self):
        for p in self._peers:
            p.close()
        if self._sock:
            self._sock.close()


class ThreadComm(_Base):
    """Ranks as THREADS of one process (one context and one HIP stream each, on one or several devices): the partition, the
    per-rank kernels and the exchange protocol of a G-rank run inside a single process — how the full-size 4- and 8-way
    partitions are checked on a one-GPU box, where at most a handful of processes may hold the device.  Create the G
    members with ThreadComm.group(G) and hand member k to thread k."""

    def __init__(self, rank, shared):
        self.rank, self.world_size, self._s = rank, shared['world'], shared

    @classmethod
    def group(cls, world, timeout=600.0):
        import threading
        shared = {'world': world, 'barrier': threading.Barrier(world, timeout=timeout), 'slots': [None] * world}
        return [cls(r, shared) for r in range(world)]

    def _exchange(self, mine):
        """every member's object, in rank order"""
        s = self._s
        s['slots'][self.rank] = mine
        s['barrier'].wait()
        out = list(s['slots'])
        s['barrier'].wait()   # nobody overwrites a slot before everybody has read them
        return out

    def broadcast_bytes(self, data, src=0):
        return self._exchange(bytes(data) if self.rank == src else None)[src]

    def allgather_rows(self, buf, r0, r1):
        for b, e, part in self._exchange((r0, r1, np.array(buf[r0:r1], copy=True))):
            if e > b:
                buf[b:e] = part

    def allreduce_sum(self, buf):
        parts = self._exchange(np.array(buf, copy=True))
        acc = parts[0].copy()
        for p in parts[1:]:      # fixed rank order -> deterministic
            acc += p
        buf[:] = acc

    def barrier(self):
        self._exchange(None)

    def max_float(self, x):
        return max(self._exchange(float(x)))

    def abort(self):
        """a member that failed wakes the others (they raise BrokenBarrierError instead of waiting for it)"""
        self._s['barrier'].abort()


def from_env(timeout=600.0):
    """Communicator for the current launcher environment (RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT): the dependency-free
    SocketComm on MASTER_PORT + 33.  (bench_rendezvous.from_env at the repository root prefers torch.distributed's gloo group when
    torch is importable.)"""
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    if world == 1:
        return SocketComm(0, 1)
    addr = os.environ.get('MASTER_ADDR', '127.0.0.1')
    port = int(os.environ.get('MASTER_PORT', '29500')) + 33
    return SocketComm(rank, world, addr, port, timeout=timeout)
