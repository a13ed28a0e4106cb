"""Synthetic inputs for the dual-QP hot path (pure NumPy, identical bytes on every machine).

These follow the measurement contract of SURVEY.md section 8(d): two overlapping Gaussian blobs,
column-standardised, labels in {-1,+1}; regression targets are a noisy linear map of the same X.
Nothing here touches the device; bench.py, the golden generator and the tests all share it.
"""
import numpy as np

__all__ = ['make_blobs', 'make_regression']


def make_blobs(n, d, seed=0, sigma=8.0, dtype=np.float64):
    """Two d-dimensional Gaussian blobs with centers U(-10,10)^d, std `sigma`, standardised columns.

    Returns (X, y) with X (n, d) C-contiguous and y in {-1.0, +1.0}.
    """
    rs = np.random.RandomState(seed)
    centers = rs.uniform(-10.0, 10.0, (2, d))
    label = np.zeros(n, dtype=np.int64)
    label[(n + 1) // 2:] = 1
    label = label[rs.permutation(n)]
    X = centers[label] + sigma * rs.standard_normal((n, d))
    X = (X - X.mean(axis=0)) / X.std(axis=0)
    y = 2.0 * label - 1.0
    return np.ascontiguousarray(X, dtype=dtype), y


def make_regression(n, d, seed=0, sigma=8.0, noise=0.1, dtype=np.float64):
    """Same X as `make_blobs`; targets y = X w / sqrt(d) + noise * N(0,1)."""
    X, _ = make_blobs(n, d, seed=seed, sigma=sigma)
    w = np.random.RandomState(seed + 1).standard_normal(d)
    y = X @ w / np.sqrt(d) + noise * np.random.RandomState(seed + 2).standard_normal(n)
    return np.ascontiguousarray(X, dtype=dtype), y
