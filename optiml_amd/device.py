"""Device contexts: one HIP stream (and, for multi-GPU runs, one RCCL communicator) per process.

One process drives one GPU.  `get_context()` returns the process-wide default context, created on first
use on device LOCAL_RANK (or 0).  Multi-GPU runs call `init_distributed(comm)` first: rank r then owns a
contiguous row block of every Gram/Hessian panel and each product Q v ends in one all-gather of the row
blocks over RCCL/xGMI (exchange='rccl'), or through the communicator's host all-gather
(exchange='host': tests and machines without RCCL).
"""
import ctypes as C
import os

import numpy as np

from . import _lib

__all__ = ['Context', 'get_context', 'set_context', 'init_distributed', 'row_block', 'device_count']


def device_count():
    lib = _lib.load()
    n = C.c_int(0)
    _lib.check(lib.bq_device_count(C.byref(n)))
    return n.value


def row_block(n, rank, world, symmetric=False):
    """Rows [begin, end) of an n-row panel owned by `rank`: equal 128-aligned blocks for dense panels; with
    symmetric=True the balanced triangular partition (256-aligned) of the kernel-built symmetric panels."""
    lib = _lib.load()
    b, e = C.c_int64(0), C.c_int64(0)
    fn = lib.bq_sym_row_block if symmetric else lib.bq_row_block
    _lib.check(fn(int(n), int(rank), int(world), C.byref(b), C.byref(e)))
    return b.value, e.value


def _agree(comm, failure, what):
    """Every rank learns whether ANY rank failed `what`, and then all of them raise — a rank that gave up alone would leave
    the others waiting inside the next collective (RCCL's communicator bootstrap has no timeout of its own)."""
    if comm.max_float(0.0 if failure is None else 1.0) > 0.0:
        if failure is not None:
            raise failure
        raise RuntimeError(f'{what} failed on another rank')


class Context:
    def __init__(self, device=None, comm=None, exchange='rccl', sym_exchange=None, share=None, collective_timeout=None):
        """exchange: 'rccl' (RCCL over xGMI) or 'host' (the communicator's host collectives; tests).  sym_exchange: how the
        products of symmetric kernel panels are closed — 'gather' (default: all-gather of the per-segment partial vectors,
        added in segment order on every rank, bit-identical iterates for any rank count) or 'allreduce' (one all-reduce(sum);
        the association of the rank sum then belongs to the transport).  None: the BQ_SYM_EXCHANGE environment variable.
        share=(k, G): rank k's share of a G-way partition with no transport (collectives are no-ops, products are partial) —
        for timing one share on one GPU, never for solving.

        collective_timeout (seconds; None: the BQ_COLLECTIVE_TIMEOUT_S environment variable, else no bound): RCCL never gives up
        on a collective whose peer does not arrive; with a timeout a watchdog aborts the communicator when the host has waited
        on the stream for longer, the call in progress raises BcqpError(ERR_RCCL) and the context is unusable afterwards.  (The
        host exchange is bounded by the communicator's own timeout.)

        device=None: LOCAL_RANK.  An RCCL rank needs a device of its own: LOCAL_RANK >= the visible device count is an
        error there (agreed on by all ranks before the communicator is created); ranks of the host exchange may share a
        device (LOCAL_RANK modulo the device count: the multi-process tests on a one-GPU box)."""
        lib = _lib.load()
        self._lib = lib
        self._h = C.c_void_p()
        self._cb = None
        self.comm = comm
        rccl = share is None and comm is not None and exchange == 'rccl'
        failure = None
        if device is None:
            device = int(os.environ.get('LOCAL_RANK', '0'))
            try:
                ndev = device_count()
            except _lib.BcqpError as err:
                if not rccl:
                    raise
                ndev, failure = 0, err
            if rccl:
                if failure is None and device >= ndev:
                    failure = RuntimeError(f'LOCAL_RANK={device} but only {ndev} device(s) are visible: an RCCL rank needs a '
                                           f'device of its own (HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES too narrow?)')
            elif ndev > 0:
                device %= ndev
        self.device = device
        if share is not None:
            k, g = share
            self.rank, self.world, self.exchange = int(k), int(g), 'share'
            _lib.check(lib.bq_ctx_create_share(device, int(k), int(g), C.byref(self._h)))
        elif comm is None or (comm.world_size == 1 and exchange != 'rccl'):
            self.rank, self.world, self.exchange = 0, 1, 'none'
            _lib.check(lib.bq_ctx_create(device, C.byref(self._h)))
        elif exchange == 'rccl':
            self.rank, self.world, self.exchange = comm.rank, comm.world_size, 'rccl'
            # pre-flight, agreed on by every rank BEFORE anyone enters ncclCommInitRank: this rank's device exists, can be
            # selected and can hold a stream (a plain context is created and destroyed)
            if failure is None:
                try:
                    probe = C.c_void_p()
                    _lib.check(lib.bq_ctx_create(device, C.byref(probe)))
                    lib.bq_ctx_destroy(probe)
                except _lib.BcqpError as err:
                    failure = err
            _agree(comm, failure, 'the device pre-flight of the RCCL context')
            uid = C.create_string_buffer(128)
            failure = None
            if comm.rank == 0:
                try:
                    _lib.check(lib.bq_comm_unique_id(uid))
                except _lib.BcqpError as err:   # e.g. librccl.so missing: tell the other ranks instead of leaving them in the broadcast
                    failure = err
            raw = comm.broadcast_bytes((b'\0' if failure else b'\1') + uid.raw, src=0)
            if raw[:1] != b'\1':
                raise failure if failure is not None else RuntimeError('rank 0 could not create an RCCL unique id')
            uid = C.create_string_buffer(raw[1:], 128)
            _lib.check(lib.bq_ctx_create_rccl(device, comm.rank, comm.world_size, uid, C.byref(self._h)))
        elif exchange == 'host':
            self.rank, self.world, self.exchange = comm.rank, comm.world_size, 'host'

            debug = os.environ.get('NCCL_DEBUG', '') in ('INFO', 'TRACE')

            def _exchange(user, buf, n, r0, r1, op):
                try:
                    if debug:
                        import sys
                        import time
                        t0 = time.perf_counter()
                    view = np.ctypeslib.as_array(buf, shape=(n,))
                    if op == 0:
                        comm.allgather_rows(view, r0, r1)
                    else:
                        comm.allreduce_sum(view)
                    if debug:
                        print(f'[exchange] rank {comm.rank} op {op} n {n}: {1e3 * (time.perf_counter() - t0):.3f} ms in the communicator',
                              file=sys.stderr, flush=True)
                    return 0
                except Exception as exc:  # never let an exception cross the C boundary
                    import traceback
                    traceback.print_exc()
                    return 1

            self._cb = _lib.EXCHANGE_FN(_exchange)
            # the native context can outlive this object (it goes with the last problem built on it): the trampoline must too
            _CALLBACKS.append(self._cb)
            _lib.check(lib.bq_ctx_create_exchange(device, comm.rank, comm.world_size, self._cb, None, C.byref(self._h)))
        else:
            raise ValueError(f"unknown exchange '{exchange}' (use 'rccl' or 'host')")
        if sym_exchange is not None:
            if sym_exchange not in ('gather', 'allreduce'):
                raise ValueError(f"unknown sym_exchange '{sym_exchange}' (use 'gather' or 'allreduce')")
            _lib.check(lib.bq_ctx_set_sym_allreduce(self._h, 1 if sym_exchange == 'allreduce' else 0))
        if collective_timeout is None and os.environ.get('BQ_COLLECTIVE_TIMEOUT_S'):
            collective_timeout = float(os.environ['BQ_COLLECTIVE_TIMEOUT_S'])
        if collective_timeout:
            self.set_collective_timeout(collective_timeout)

    @property
    def handle(self):
        if not self._h:
            raise RuntimeError('context has been closed')
        return self._h

    @property
    def name(self):
        buf = C.create_string_buffer(128)
        _lib.check(self._lib.bq_ctx_info(self.handle, None, None, None, buf, 128))
        return buf.value.decode()

    def comm_info(self):
        """{'kind': 'none'|'rccl'|'callback'|'share', 'rccl_ranks': ncclCommCount of the live communicator (0 without RCCL),
        'sym_exchange': 'gather'|'allreduce', 'rccl_init_stages': 'dlopen(librccl.so) 0.4 s; ...'}"""
        kind, ranks, ar = C.c_int(0), C.c_int(0), C.c_int(0)
        _lib.check(self._lib.bq_ctx_comm_info(self.handle, C.byref(kind), C.byref(ranks), C.byref(ar)))
        buf = C.create_string_buffer(1024)
        _lib.check(self._lib.bq_comm_init_report(buf, len(buf)))
        return {'kind': ('none', 'rccl', 'callback', 'share')[kind.value], 'rccl_ranks': ranks.value,
                'sym_exchange': 'allreduce' if ar.value else 'gather',
                'rccl_init_stages': buf.value.decode()}   # where RCCL's start-up time went in this process ('' without RCCL)

    def set_sym_exchange(self, mode):
        """'gather' or 'allreduce' for the products that follow (every rank must switch at the same point)."""
        if mode not in ('gather', 'allreduce'):
            raise ValueError(f"unknown sym_exchange '{mode}' (use 'gather' or 'allreduce')")
        _lib.check(self._lib.bq_ctx_set_sym_allreduce(self.handle, 1 if mode == 'allreduce' else 0))

    def profile(self, enable=True):
        _lib.check(self._lib.bq_ctx_profile(self.handle, 1 if enable else 0))

    def profile_read(self, which, reset=False):
        """(total_ms, launches) of the HIP-event timers: which in _lib.PROF_*."""
        ms, cnt = C.c_double(0), C.c_int64(0)
        _lib.check(self._lib.bq_ctx_profile_read(self.handle, which, C.byref(ms), C.byref(cnt), 1 if reset else 0))
        return ms.value, cnt.value

    def probe_bandwidth(self, nbytes=4 << 30, reps=5):
        """(read_GBs, copy_GBs): measured streaming-read and device-copy bandwidth of this GPU."""
        r, c = C.c_double(0), C.c_double(0)
        _lib.check(self._lib.bq_ctx_probe_bandwidth(self.handle, int(nbytes), int(reps), C.byref(r), C.byref(c)))
        return r.value, c.value

    def set_placement_budget(self, expected_products=0., min_ms=-1., max_ms=5000.):
        """Budget of the panel placement choice (`KernelQuadratic(tune_placement=True)`) for problems created from now on: up to 2 %
        of `expected_products` products of the panel, within [min_ms (default: BQ_PLACE_BUDGET_MS / 200 ms), max_ms]."""
        _lib.check(self._lib.bq_ctx_set_placement_budget(self.handle, float(min_ms), float(max_ms), float(expected_products)))

    def release_held_memory(self):
        """Give the panel-sized allocations the placement choice is holding back (until the solve they were tried for is over) to
        the driver now; returns the bytes released.  For callers that need the memory for something the library does not see."""
        n = C.c_int64(0)
        _lib.check(self._lib.bq_ctx_release_held_memory(self.handle, C.byref(n)))
        return n.value

    def set_collective_timeout(self, seconds):
        """Abort the RCCL communicator when a wait on the stream lasts longer than `seconds` (0: never)."""
        _lib.check(self._lib.bq_ctx_set_collective_timeout(self.handle, float(seconds)))

    def probe_stall(self, milliseconds, behind_collective=False):
        """Occupy the stream for `milliseconds` and wait for it through the bounded wait; behind_collective: with the context's
        collective enqueued behind the occupation, i.e. a peer that is that late (the watchdog's one-GPU test)."""
        _lib.check(self._lib.bq_ctx_probe_stall(self.handle, float(milliseconds), int(bool(behind_collective))))

    def probe_exchange(self, kind, count, reps=50):
        """(mean_us, min_us) of this context's closing collective timed on its own: kind 'gather' = the in-place all-gather of
        `count` doubles per rank, 'allreduce' = the all-reduce(sum) of `count` doubles (every rank calls it alike)."""
        mean, mn = C.c_double(0), C.c_double(0)
        _lib.check(self._lib.bq_ctx_probe_exchange(self.handle, {'gather': 0, 'allreduce': 1}[kind], int(count), int(reps),
                                                   C.byref(mean), C.byref(mn)))
        return mean.value, mn.value

    def probe_mfma_f64(self, seconds=1.0):
        """TFLOP/s that back-to-back fp64 MFMAs on register operands sustain on this GPU (clock under load included)."""
        t = C.c_double(0)
        _lib.check(self._lib.bq_ctx_probe_mfma_f64(self.handle, float(seconds), C.byref(t)))
        return t.value

    def close(self):
        if self._h:
            self._lib.bq_ctx_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


_CALLBACKS = []   # ctypes trampolines of host-exchange contexts, kept for the life of the process (a few bytes each)
_default = None


def get_context():
    global _default
    if _default is None:
        _default = Context()
    return _default


def set_context(ctx):
    global _default
    _default = ctx
    return ctx


def init_distributed(comm, exchange='rccl', device=None, sym_exchange=None):
    """Make the default context a multi-rank one (call once per process, before building problems)."""
    return set_context(Context(device=device, comm=comm, exchange=exchange, sym_exchange=sym_exchange))
