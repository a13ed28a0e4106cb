#!/usr/bin/env python3
"""Study tool (CPU, NumPy/SciPy): which explicit directions of the ORDER-2 Taylor term of the RBF kernel are worth putting into the
Woodbury preconditioner P = D + Phi Phi' of the config-5 dual, and is an implicit (feature-free) order-2 remainder behind a Chebyshev
polynomial worth its products?

    python tools/pc_projected_cpu_study.py 20000 64

"top directions set 1" = the exact projection of sqrt(2) g y e vec(x x') onto span{m_c e_k' + e_k m_c'} (2d features) — what
csrc/bq_as.hip builds since round 5 (as_pc_project_kernel); set 2 adds I and the class second moments (no gain).  Result
(profiles/r05/pc_projected_study.txt): order 0-1 alone 55 iterations, + the class-mean cross term of rounds 3-4 39, + the projected
directions 23, + one implicit order-2 remainder application per call 18 (exact order-2 Woodbury: 17).
"""
import sys, time
import numpy as np
import scipy.sparse.linalg as sla
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
dq = np.diag(Q).copy(); rhs = np.ones(n)
e = np.exp(-gamma * sq); ye = y * e; c2 = 2 * gamma * gamma
Phi0 = np.hstack([ye[:, None], ye[:, None] * np.sqrt(2 * gamma) * X, y[:, None]])
# analytic top directions of the second-order block: U_r symmetric d x d matrices; feature_r(i) = sqrt(c2) ye_i x_i' U_r x_i
mp_, mm_ = X[y > 0].mean(0), X[y < 0].mean(0)
def vecs(which):
    Us = []
    for m in (mp_, mm_):
        for k in range(d):
            U = np.zeros((d, d)); U[:, k] += m; U[k, :] += m; Us.append(U)
    if which >= 2:
        Us.append(np.eye(d))
        for m, sel in ((mp_, y > 0), (mm_, y < 0)):
            Us.append(X[sel].T @ X[sel] / sel.sum())
    return np.array(Us).reshape(len(Us), -1)
for which in (1, 2):
    V = vecs(which)                                   # r x d^2 (Frobenius inner product = the feature-space inner product)
    Qr, _ = np.linalg.qr(V.T)                         # d^2 x r orthonormal
    Z = (X[:, :, None] * X[:, None, :]).reshape(n, d * d)
    Phitop = (np.sqrt(c2) * ye)[:, None] * (Z @ Qr)   # n x r: exact projection of the second-order features
    del Z
    d2diag = c2 * e * e * sq * sq
    dg = np.maximum(dq - (Phi0 * Phi0).sum(1) - d2diag, 0.3)
    Phi1 = np.hstack([Phi0, Phitop])
    G = np.eye(Phi1.shape[1]) + Phi1.T @ (Phi1 / dg[:, None]); c = np.linalg.cholesky(G)
    P1inv = lambda r: r / dg - (Phi1 @ np.linalg.solve(c.T, np.linalg.solve(c, Phi1.T @ (r / dg)))) / dg
    def Rop(v):   # remainder of the second-order block: B v - Phitop Phitop' v  (PSD: a projection was removed)
        M = X.T @ ((ye * v)[:, None] * X)
        return c2 * ye * np.einsum('ij,ij->i', X @ M, X) - Phitop @ (Phitop.T @ v)
    T = sla.LinearOperator((n, n), matvec=lambda v: v + P1inv(Rop(v)))
    hi = sla.eigs(T, k=1, which='LR', tol=1e-3, return_eigenvectors=False).real[0]
    lo = sla.eigs(T, k=1, which='SR', tol=1e-3, return_eigenvectors=False, maxiter=5000).real[0]
    print(f'top directions set {which}: {Phitop.shape[1]} projected order-2 features; spectrum of P1^-1 P in [{lo:.3f}, {hi:.3f}]')
    def cheb_pc(k, a_, b_):
        th, de = (a_ + b_) / 2, (b_ - a_) / 2; s1 = th / de
        def ap(r):
            res = P1inv(r); rho = 1 / s1; dd = res / th; z = np.zeros(n)
            for it in range(k):
                z += dd
                if it == k - 1: break
                res -= dd + P1inv(Rop(dd))
                rho1 = 1 / (2 * s1 - rho); dd = rho1 * rho * dd + (2 * rho1 / de) * res; rho = rho1
            return z
        return ap
    def pcg(apply_pc, tol=1e-8, cap=300):
        x = np.zeros(n); r = rhs.copy(); z = apply_pc(r); p = z.copy(); rz = r @ z; nb = np.linalg.norm(rhs); hist = []
        for k in range(cap):
            Qp = Q @ p; al = rz / (p @ Qp); x += al * p; r -= al * Qp; hist.append(np.linalg.norm(r) / nb)
            if hist[-1] <= tol: break
            z = apply_pc(r); rz2 = r @ z; p = z + (rz2 / rz) * p; rz = rz2
        return hist
    h = pcg(P1inv); print('   P1 alone: %d iterations' % len(h))
    for k in (2, 3, 4):
        h = pcg(cheb_pc(k, 1.0, 1.05 * hi)); print('   Chebyshev, %d remainder applications per call: %d iterations, rate(last5) %.3f' % (k - 1, len(h), (h[-1] / h[-6]) ** 0.2))
