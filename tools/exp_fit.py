#!/usr/bin/env python3
"""Coefficients of the kernel maps' exp (optiml_amd/csrc/bq_exp.h): (exp(r) - 1 - r) / r^2 interpolated at the Chebyshev nodes of
|r| <= ln(2) / 2 (degree 9, extended precision, iterative refinement), printed as C hex floats with the error of the assembled
1 + r + r^2 q(r).    python tools/exp_fit.py"""
import numpy as np
from numpy.polynomial import chebyshev as C, polynomial as P
L = np.longdouble
h = L(np.log(2.0))/2 * L(1.0005)
# fit q(r) = (exp(r) - 1 - r) / r^2 with degree 9 on [-h, h] at Chebyshev nodes (near-minimax), in extended precision
deg = 9
k = np.arange(deg + 1, dtype=L)
nodes = np.cos((2*k + 1) * L(np.pi) / (2*(deg + 1))) * h
def q(r):
    r = np.asarray(r, dtype=L)
    out = np.empty_like(r)
    small = np.abs(r) < 1e-3
    # series for small r
    rs = r[small]; s = np.zeros_like(rs); term = np.ones_like(rs) / 2
    for n in range(2, 30):
        s = s + term; term = term * rs / (n + 1)
    out[small] = s
    rb = r[~small]
    out[~small] = (np.expm1(rb) - rb) / (rb*rb)
    return out
V = np.vander(nodes / h, deg + 1, increasing=True).astype(L)
coef_scaled = np.linalg.solve(V.astype(np.float64), q(nodes).astype(np.float64))  # initial
# refine in long double via a few steps of iterative refinement
c = coef_scaled.astype(L)
for _ in range(5):
    res = q(nodes) - V @ c
    c = c + np.linalg.solve(V.astype(np.float64), res.astype(np.float64)).astype(L)
coef = c / (h ** np.arange(deg + 1, dtype=L))
xs = np.linspace(-float(h), float(h), 200001).astype(L)
approx = np.zeros_like(xs)
for cc in coef[::-1]:
    approx = approx * xs + cc
full = 1 + xs + xs*xs*approx
err = np.max(np.abs(full - np.exp(xs)) / np.exp(xs))
print('max rel err (extended arithmetic):', float(err))
for i, cc in enumerate(coef):
    print('c%d = %s' % (i + 2, float(cc).hex()), repr(float(cc)))
