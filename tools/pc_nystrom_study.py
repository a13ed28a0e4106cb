#!/usr/bin/env python3
"""Study tool (CPU, NumPy): would Nystrom landmark columns of the Gram panel precondition BASELINE config 5's inner conjugate
gradients better than the first-order Taylor features the library uses?

    python tools/pc_nystrom_study.py 6000 256        # config 5's d, gamma = 'scale', blobs sigma 8, Q = K*yy' + yy' + I/(2C)

Prints preconditioned-CG iteration counts to 1e-8 for: none, Jacobi, first-order Taylor (d + 2 features), Nystrom with 512 / 1024 /
2048 uniformly sampled landmarks (+ the y column), and the spectrum of Q.  Result (profiles/r05/pc_nystrom_study.txt): Nystrom needs
m = 1024 landmarks to equal the d + 2 = 258 Taylor features and m = 2048 (a third of n) to beat them by a quarter — the spectrum
beyond the first-order directions is flat, there is no decaying tail for landmarks to capture.
"""
import sys, time
import numpy as np
sys.path.insert(0, '/root/repo')
from optiml_amd.datasets import make_blobs
n, d = int(sys.argv[1]), int(sys.argv[2])
X, y = make_blobs(n, d, seed=0, sigma=8.0)
X = X.astype(np.float32).astype(np.float64)
gamma = 1.0 / (d * X.var())
sq = (X * X).sum(1)
D2 = np.maximum(sq[:, None] + sq[None, :] - 2 * X @ X.T, 0)
K = np.exp(-gamma * D2); del D2
Q = K * np.outer(y, y) + np.outer(y, y) + 0.5 * np.eye(n)
print('n', n, 'd', d, 'gamma', gamma, 'mean offdiag K', (K.sum() - n) / (n * (n - 1)))
rhs = np.ones(n)  # -q
def pcg(apply_pc, tol=1e-8, x0=None, cap=200):
    x = np.zeros(n) if x0 is None else x0.copy()
    r = rhs - Q @ x
    z = apply_pc(r); p = z.copy(); rz = r @ z
    nb = np.linalg.norm(rhs)
    hist = []
    for k in range(cap):
        Qp = Q @ p
        a = rz / (p @ Qp)
        x += a * p; r -= a * Qp
        hist.append(np.linalg.norm(r) / nb)
        if hist[-1] <= tol: break
        z = apply_pc(r); rz2 = r @ z; p = z + (rz2 / rz) * p; rz = rz2
    return x, hist
def woodbury(Phi):
    # P = Dg + Phi Phi', Dg = diag(Q) - diag(Phi Phi') clipped
    dg = np.maximum(np.diag(Q) - (Phi * Phi).sum(1), 0.5)
    G = np.eye(Phi.shape[1]) + Phi.T @ (Phi / dg[:, None])
    L = np.linalg.cholesky(G)
    def ap(r):
        t = Phi.T @ (r / dg)
        u = np.linalg.solve(L.T, np.linalg.solve(L, t))
        return r / dg - (Phi @ u) / dg
    return ap
def taylor():
    e = np.exp(-gamma * sq)
    return np.hstack([(y * e)[:, None], (y * e)[:, None] * np.sqrt(2 * gamma) * X, y[:, None]])
def nystrom(m, seed=0):
    S = np.random.RandomState(seed).choice(n, m, replace=False)
    C = K[:, S]; W = K[np.ix_(S, S)]
    Lw = np.linalg.cholesky(W + 1e-10 * np.eye(m))
    F = np.linalg.solve(Lw, C.T).T   # F F' = C W^-1 C'
    return np.hstack([y[:, None] * F, y[:, None]])
for name, ap in [('none', lambda r: r), ('jacobi', lambda r: r / np.diag(Q)), ('taylor d+2', woodbury(taylor()))] + \
                [(f'nystrom {m}', woodbury(nystrom(m))) for m in (512, 1024, 2048, 4096) if m < n // 2] + \
                [(f'taylor + nystrom-of-residual {m}', None) for m in ()]:
    t0 = time.time(); x, h = pcg(ap); print(f'{name:28s} iterations to 1e-8: {len(h):3d}   to 1e-4: {next(i+1 for i,v in enumerate(h) if v<=1e-4):3d}   ({time.time()-t0:.1f}s)')
# spectrum of the Taylor-preconditioned operator (is it a flat bulk?)
if n <= 8000:
    w = np.linalg.eigvalsh(Q)
    print('eig(Q): min %.3f  median %.3f  90%% %.3f  99%% %.3f  max %.1f' % (w[0], np.median(w), np.quantile(w, .9), np.quantile(w, .99), w[-1]))
    print('eigenvalues > 10:', (w > 10).sum(), ' > 3:', (w > 3).sum(), ' > 1.5:', (w > 1.5).sum())
