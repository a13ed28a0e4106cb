#!/usr/bin/env python3
"""What would overlapping the dense ActiveSet's base re-factorisation with the iterations before it cost the iterations?  (VERDICT r5
item 6 — measured before anything is built.)

Two contexts (two streams) on one GPU, two threads: A runs dense ActiveSet iterations on BASELINE config 2's shape (kept factor,
sweeps, Schur slots: short kernels, latency- and bandwidth-bound); B factorises Hessians of the order A's base has, back to back, the
way a background re-factorisation would (InteriorPoint iterations on a second problem: one blocked MFMA Cholesky each).  Printed: A's
time per iteration alone and beside B, B's time per factorisation alone and beside A.

    python tools/as_overlap_probe.py [--n 20000] [--d 64] [--nb 16000] [--iters 1500]
"""
import argparse
import json
import os
import sys
import threading
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--n', type=int, default=20000)
    ap.add_argument('--d', type=int, default=64)
    ap.add_argument('--nb', type=int, default=16000, help='order of the Hessians factorised in the background')
    ap.add_argument('--iters', type=int, default=1500)
    ap.add_argument('--warm', type=int, default=3000, help="ActiveSet iterations before the timing starts (the free set has left its first, largest sizes)")
    a = ap.parse_args()
    from optiml_amd import _lib, device
    from optiml_amd.datasets import make_blobs
    from optiml_amd.ml.svm.kernels import gaussian
    from optiml_amd.opti import KernelQuadratic
    from optiml_amd.opti.constrained._base import _DeviceSolver

    ctx_a, ctx_b = device.Context(), device.Context()
    X, y = make_blobs(a.n, a.d, seed=0)
    qa = KernelQuadratic(X, -np.ones(a.n), 'svc', gaussian, y=y)
    sa = _DeviceSolver(qa.device_problem(ctx_a), _lib.AS, np.zeros(a.n), np.ones(a.n), np.ones(a.n) / 2, 1e-6, 10 ** 9)
    Xb, yb = make_blobs(a.nb, a.d, seed=1)
    qb = KernelQuadratic(Xb, -np.ones(a.nb), 'svc', gaussian, y=yb)
    devb = qb.device_problem(ctx_b)
    new_b = lambda: _DeviceSolver(devb, _lib.IP, np.zeros(a.nb), np.ones(a.nb), np.ones(a.nb) / 2, 1e-10, 10 ** 9)
    box = {'s': new_b(), 'done': 0}   # (a fresh InteriorPoint every 16 iterations: far from convergence, every iteration one factorisation)
    sa.run(a.warm)
    box['s'].run(2)

    def run_a(k):
        t0 = time.perf_counter()
        rows, _ = sa.run(k)
        return (time.perf_counter() - t0) / max(len(rows), 1)

    def run_b(k):
        if box['done'] >= 16:
            box['s'].close()
            box['s'], box['done'] = new_b(), 0
            box['s'].run(1)   # the start-up product and first factorisation: not timed
        t0 = time.perf_counter()
        rows, _ = box['s'].run(k)
        box['done'] += k
        return (time.perf_counter() - t0) / max(len(rows), 1)

    out = {'what': 'as_overlap_probe', 'n': a.n, 'd': a.d, 'background_order': a.nb, 'device': ctx_a.name}
    out['as_ms_per_iteration_alone'] = 1e3 * run_a(a.iters)
    out['factorisation_ms_alone'] = 1e3 * run_b(6)
    stop = threading.Event()
    b_times = []

    def background():
        while not stop.is_set():
            b_times.append(run_b(2))

    th = threading.Thread(target=background)
    th.start()
    time.sleep(0.2)
    out['as_ms_per_iteration_beside_factorisations'] = 1e3 * run_a(a.iters)
    stop.set()
    th.join()
    out['factorisation_ms_beside_as'] = 1e3 * float(np.mean(b_times)) if b_times else None
    out['as_ms_per_iteration_alone_again'] = 1e3 * run_a(a.iters)
    out['as_slowdown'] = out['as_ms_per_iteration_beside_factorisations'] / out['as_ms_per_iteration_alone']
    out['factorisation_slowdown'] = (out['factorisation_ms_beside_as'] / out['factorisation_ms_alone']) if b_times else None
    print(json.dumps(out))


if __name__ == '__main__':
    main()
