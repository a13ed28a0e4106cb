#!/usr/bin/env python3
"""Study tool (CPU, NumPy): first-order against first+second-order Taylor features of the RBF kernel in the Woodbury preconditioner
P = D + Phi Phi' of the config-5 dual, at a size where the second-order block stands out of the bulk on a CPU (d = 64: 2 080
second-order directions with eigenvalues ~ 0.27 n / 2080).

    python tools/pc_order2_cpu_study.py 20000 64

Result (profiles/r05/pc_order2_cpu_study.txt): 55 -> 17 iterations to 1e-8 (asymptotic rate 0.63 -> 0.39).
"""
import sys, time
import numpy as np
sys.path.insert(0, '/root/repo')
from optiml_amd.datasets import make_blobs
n, d = int(sys.argv[1]), int(sys.argv[2])
X, y = make_blobs(n, d, seed=0, sigma=8.0)
gamma = 1.0 / (d * X.var())
sq = (X * X).sum(1)
K = sq[:, None] + sq[None, :] - 2 * X @ X.T
np.maximum(K, 0, out=K); K *= -gamma; np.exp(K, out=K)
Q = K * np.outer(y, y); Q += np.outer(y, y); Q[np.diag_indices(n)] += 0.5
del K
dq = np.diag(Q).copy()
rhs = np.ones(n)
def pcg(apply_pc, tol=1e-8, cap=300):
    x = np.zeros(n); r = rhs.copy()
    z = apply_pc(r); p = z.copy(); rz = r @ z; nb = np.linalg.norm(rhs); hist = []
    for k in range(cap):
        Qp = Q @ p; a = rz / (p @ Qp); x += a * p; r -= a * Qp
        hist.append(np.linalg.norm(r) / nb)
        if hist[-1] <= tol: break
        z = apply_pc(r); rz2 = r @ z; p = z + (rz2 / rz) * p; rz = rz2
    return hist
def woodbury(Phi):
    dg = np.maximum(dq - (Phi * Phi).sum(1), 0.5)
    G = np.eye(Phi.shape[1]) + Phi.T @ (Phi / dg[:, None])
    c = np.linalg.cholesky(G)
    def ap(r):
        t = Phi.T @ (r / dg); u = np.linalg.solve(c.T, np.linalg.solve(c, t)); return r / dg - (Phi @ u) / dg
    return ap, dg
e = np.exp(-gamma * sq)
P0 = (y * e)[:, None]
P1 = (y * e)[:, None] * np.sqrt(2 * gamma) * X
iu = np.triu_indices(d)
scale = np.where(iu[0] == iu[1], 1.0, np.sqrt(2.0))
P2 = (y * e)[:, None] * (2 * gamma) / np.sqrt(2.0) * (X[:, iu[0]] * X[:, iu[1]] * scale)
one = y[:, None]
for name, Phi in [('order 0-1 (d+2)', np.hstack([P0, P1, one])), ('order 0-2 (d+2 + d(d+1)/2 = %d)' % (d + 2 + len(iu[0])), np.hstack([P0, P1, P2, one]))]:
    t0 = time.time(); ap, dg = woodbury(Phi); h = pcg(ap)
    print(f'{name:40s} iterations to 1e-8: {len(h):3d}  1e-5: {next(i+1 for i,v in enumerate(h) if v<=1e-5):3d}  rate(last5) {(h[-1]/h[-6])**0.2:.3f}   D in [{dg.min():.3f}, {dg.max():.3f}]  ({time.time()-t0:.0f}s)', flush=True)
h = pcg(lambda r: r); print(f'{"none":40s} iterations to 1e-8: {len(h):3d}  rate(last5) {(h[-1]/h[-6])**0.2:.3f}')
