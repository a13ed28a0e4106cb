#!/usr/bin/env python3
"""Study tool (torch on the GPU as a calculator; NOT the product path): what would cut BASELINE config 5's ~10 panel products per
outer ActiveSet iteration?

VERDICT r4 item 2 proposed Nystrom landmark columns of the resident panel.  tools/pc_nystrom_study.py (CPU, n = 6 000) shows they
do not help: beyond the 2d + 2 first-order directions the spectrum of K*yy' + yy' + I/(2C) at gamma = 'scale', d = 256 is FLAT
(514 eigenvalues > 1.5, the other 5 486 in [0.8, 1.2]) — there is no fast-decaying tail for landmarks to capture.  What grows with
n is the SECOND-order Taylor block of exp(2 gamma x.x'): d (d + 1) / 2 = 32 896 directions with eigenvalues ~ 0.27 n / 32 896
(2.05 at n = 250 000 against a bulk of ~0.8): that block sets the conjugate-gradient rate at config-5 scale.

This script measures, at n = 100 000 ... 160 000 and d = 256 on one GPU (K in fp32, 40 - 102 GB):
  * preconditioned CG iterations from zero to 1e-8 with the Woodbury preconditioner P = D + Phi Phi' for
      order 0-1 features (d + 2 columns: what the library has, without the class split) and
      order 0-2 features (d + 2 + d (d + 1) / 2 columns)
  * the warm-started re-solve after ONE index has been bound (the real per-outer-iteration cost), for both.

    python tools/pc_order2_study.py --n 100000 --d 256 [--binds 3]
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from optiml_amd.datasets import make_blobs  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--n', type=int, default=100000)
    ap.add_argument('--d', type=int, default=256)
    ap.add_argument('--binds', type=int, default=3)
    ap.add_argument('--tol', type=float, default=1e-8)
    args = ap.parse_args()
    n, d = args.n, args.d
    dev = torch.device('cuda:0')
    X64, y64 = make_blobs(n, d, seed=0, sigma=8.0)
    X = torch.tensor(X64.astype(np.float32), device=dev)
    y = torch.tensor(y64, dtype=torch.float64, device=dev)
    gamma = 1.0 / (d * float(X64.astype(np.float32).var()))
    sq = (X.double() ** 2).sum(1)
    t0 = time.time()
    K = torch.empty((n, n), dtype=torch.float32, device=dev)
    blk = 8192
    for r0 in range(0, n, blk):
        r1 = min(n, r0 + blk)
        D2 = sq[r0:r1, None].float() + sq[None, :].float() - 2.0 * (X[r0:r1] @ X.T)
        K[r0:r1] = torch.exp(-gamma * D2.clamp_(min=0))
        del D2
    torch.cuda.synchronize()
    print(f'[study] n={n} d={d} gamma={gamma:.6g}: K built in {time.time() - t0:.1f} s ({K.numel() * 4 / 1e9:.0f} GB)', flush=True)

    products = [0]

    def Qmv(v, mask):
        """(Q (m.v)) masked: Q = K*yy' + yy' + I/2, fp64 vectors, fp32 panel (the product is done in fp32 blocks, summed in fp64)."""
        products[0] += 1
        w = (mask * v * y)
        out = torch.empty(n, dtype=torch.float64, device=dev)
        w32 = w.float()
        for r0 in range(0, n, 32768):
            out[r0:r0 + 32768] = (K[r0:r0 + 32768] @ w32).double()
        out = y * out + y * (y @ (mask * v)) + 0.5 * mask * v
        return mask * out

    e = torch.exp(-gamma * sq)
    ye = (y * e)
    feats01 = torch.cat([ye[:, None], ye[:, None] * np.sqrt(2 * gamma) * X.double(), y[:, None]], 1).float()   # n x (d + 2)
    iu = torch.triu_indices(d, d, device=dev)
    scale = torch.where(iu[0] == iu[1], 1.0, float(np.sqrt(2.0))).float().to(dev)

    def feats2_block(r0, r1):
        xb = X[r0:r1]
        return (ye[r0:r1, None].float() * float(2 * gamma / np.sqrt(2.0))) * (xb[:, iu[0]] * xb[:, iu[1]] * scale)

    def make_pc(order, mask):
        """Woodbury data of P = D + Phi Phi' on the free set `mask`: (Phi fp32, dinv, Cholesky factor of G = I + Phi_A' D_A^-1 Phi_A)."""
        t0 = time.time()
        if order == 1:
            Phi = feats01
        else:
            m2 = iu.shape[1]
            Phi = torch.empty((n, d + 2 + m2), dtype=torch.float32, device=dev)
            Phi[:, :d + 2] = feats01
            for r0 in range(0, n, 16384):
                r1 = min(n, r0 + 16384)
                Phi[r0:r1, d + 2:] = feats2_block(r0, r1)
        diagQ = torch.full((n,), 2.5, dtype=torch.float64, device=dev)   # K_ii + 1 + 1/2
        dg = torch.empty(n, dtype=torch.float64, device=dev)
        for r0 in range(0, n, 16384):
            dg[r0:r0 + 16384] = (Phi[r0:r0 + 16384].double() ** 2).sum(1)
        dg = (diagQ - dg).clamp_(min=0.5)
        dinv = 1.0 / dg
        return Phi, dinv, time.time() - t0

    def factor(Phi, dinv, mask):
        t0 = time.time()
        m = Phi.shape[1]
        G = torch.zeros((m, m), dtype=torch.float32, device=dev)
        for r0 in range(0, n, 16384):
            P = Phi[r0:r0 + 16384]
            G += (P * (mask[r0:r0 + 16384] * dinv[r0:r0 + 16384]).float()[:, None]).T @ P
        G = G.double()
        G += torch.eye(m, dtype=torch.float64, device=dev)
        L = torch.linalg.cholesky(G)
        del G
        torch.cuda.synchronize()
        return L, time.time() - t0

    def apply_pc(Phi, dinv, L, mask, r):
        s = mask * dinv * r
        t = torch.zeros(Phi.shape[1], dtype=torch.float64, device=dev)
        for r0 in range(0, n, 32768):
            t += (Phi[r0:r0 + 32768].T @ s[r0:r0 + 32768].float()).double()
        u = torch.cholesky_solve(t[:, None], L)[:, 0].float()
        out = torch.empty(n, dtype=torch.float64, device=dev)
        for r0 in range(0, n, 32768):
            out[r0:r0 + 32768] = (Phi[r0:r0 + 32768] @ u).double()
        return mask * dinv * (r - out)

    def pcg(pc, mask, x0, rhs, tol):
        Phi, dinv, L = pc
        x = x0.clone()
        r = mask * (rhs - Qmv(x, mask)) if x0.abs().sum() > 0 else mask * rhs
        level = tol * float(torch.linalg.norm(mask * rhs))
        z = apply_pc(Phi, dinv, L, mask, r)
        p = z.clone()
        rz = r @ z
        hist = [float(torch.linalg.norm(r))]
        while hist[-1] > level and len(hist) < 400:
            Qp = Qmv(p, mask)
            a = rz / (p @ Qp)
            x += a * p
            r -= a * Qp
            hist.append(float(torch.linalg.norm(r)))
            z = apply_pc(Phi, dinv, L, mask, r)
            rz2 = r @ z
            p = z + (rz2 / rz) * p
            rz = rz2
        return x, hist

    out = {'n': n, 'd': d, 'gamma': gamma, 'tol': args.tol, 'orders': {}}
    rhs = torch.ones(n, dtype=torch.float64, device=dev)     # -q
    for order in (1, 2):
        mask = torch.ones(n, dtype=torch.float64, device=dev)
        Phi, dinv, t_feat = make_pc(order, mask)
        L, t_fac = factor(Phi, dinv, mask)
        products[0] = 0
        t0 = time.time()
        x, hist = pcg((Phi, dinv, L), mask, torch.zeros(n, dtype=torch.float64, device=dev), rhs, args.tol)
        torch.cuda.synchronize()
        rec = {'features': int(Phi.shape[1]), 'features_s': t_feat, 'G_and_cholesky_s': t_fac, 'cold_iterations': len(hist) - 1,
               'cold_s': time.time() - t0, 'cold_rate_last5': (hist[-1] / hist[-6]) ** 0.2 if len(hist) > 6 else None, 'warm': []}
        print(f'[study] order 0-{order}: {Phi.shape[1]} features, G + Cholesky {t_fac:.1f} s, cold solve {len(hist) - 1} iterations', flush=True)
        # the outer iteration's real cost: bind the free index with the smallest candidate value (what a ratio step does), re-solve
        # from the previous candidate
        for b in range(args.binds):
            j = int(torch.argmin(torch.where(mask > 0, x, torch.full_like(x, float('inf')))))
            mask[j] = 0.0
            x[j] = 0.0
            L, t_fac = factor(Phi, dinv, mask)   # (the library updates G^-1 by Sherman-Morrison; here simply afresh)
            products[0] = 0
            x, hist = pcg((Phi, dinv, L), mask, x, rhs, args.tol)
            rec['warm'].append({'bound_index': j, 'iterations': len(hist) - 1, 'products': products[0],
                                'start_rel_residual': hist[0] / float(torch.linalg.norm(rhs))})
            print(f'[study]   bind {j}: warm re-solve {len(hist) - 1} iterations ({products[0]} products), start residual '
                  f'{rec["warm"][-1]["start_rel_residual"]:.2e}', flush=True)
        out['orders'][f'0-{order}'] = rec
        del Phi, L
        torch.cuda.empty_cache()
    print(json.dumps(out))


if __name__ == '__main__':
    main()
