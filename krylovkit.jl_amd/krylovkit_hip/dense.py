"""Host-side small dense helpers (k x k work, k <= krylovdim): mirrors src/dense/*.jl.

These act on the projected problem only (O(krylovdim^3) work on <= ~116 x 116 matrices); the
reference does the same on the CPU through LAPACK (`docs/src/index.md:74-77`).  Nothing here
touches N-length vectors.
"""
from __future__ import annotations

import math

import numpy as np
import scipy.linalg as sla


def householder(x: np.ndarray, i: int):
    """_householder!(v, i) (dense/reflector.jl:40-69), real case, i 0-based.
    Returns (beta, v, nu): (I - beta v v') x = nu e_i with nu = |x| >= 0 and v[i] = 1."""
    v = np.array(x, dtype=np.float64)
    sigma = float(v[:i] @ v[:i] + v[i + 1:] @ v[i + 1:])
    vi = float(v[i])
    nu = math.sqrt(vi * vi + sigma)
    if sigma == 0.0 and vi == nu:
        return 0.0, v, nu
    vi = vi - nu if vi < 0 else -sigma / (vi + nu)
    v[:i] /= vi
    v[i + 1:] /= vi
    v[i] = 1.0
    return -vi / nu, v, nu


def lmul_householder(beta: float, v: np.ndarray, rows, A: np.ndarray) -> np.ndarray:
    """lmul!(H, A) (reflector.jl:89-112): A[rows, :] -= beta * v (v' A[rows, :])"""
    if beta != 0.0:
        A[rows, :] -= np.outer(v, beta * (v @ A[rows, :]))
    return A


def rmul_householder(A: np.ndarray, beta: float, v: np.ndarray, cols, rows=slice(None)) -> np.ndarray:
    """rmul!(A, H) (reflector.jl:113-142): A[rows, cols] -= (A[rows, cols] v) beta v'"""
    if beta != 0.0:
        A[rows, cols] -= np.outer(A[rows, cols] @ v, beta * v)
    return A


def givens(f: float, g: float):
    """LinearAlgebra.givens (LAPACK dlartg): (c, s, r) with [c s; -s c][f; g] = [r; 0]."""
    if g == 0.0:
        return 1.0, 0.0, f
    if f == 0.0:
        return 0.0, 1.0, g
    r = math.hypot(f, g)
    c, s = f / r, g / r
    if abs(f) > abs(g) and c < 0:
        c, s, r = -c, -s, -r
    return c, s, r


def eigsort(which: str):
    """eigsort (eigsolve/eigsolve.jl:334-355) for a real spectrum."""
    if which == "LM":
        return np.abs, True
    if which == "LR":
        return (lambda d: d), True
    if which == "SR":
        return (lambda d: d), False
    raise ValueError(f"invalid specification of which eigenvalues to target: which = {which}")


def sortperm(D: np.ndarray, which: str) -> np.ndarray:
    by, rev = eigsort(which)
    key = by(np.asarray(D))
    return np.argsort(-key if rev else key, kind="stable")


def tridiageigh(alphas: np.ndarray, betas: np.ndarray):
    """tridiageigh! (dense/linalg.jl:109-116 -> LAPACK stegr)"""
    if len(alphas) == 1:
        return np.array([alphas[0]]), np.ones((1, 1))
    return sla.eigh_tridiagonal(alphas, betas, lapack_driver="stemr")


def bidiagsvd(B: np.ndarray):
    """bidiagsvd! (dense/linalg.jl:123-130 -> LAPACK bdsqr): B = P diag(S) Q, S descending."""
    return np.linalg.svd(B)


def ldiv_upper(R: np.ndarray, y: np.ndarray, k: int) -> np.ndarray:
    """ldiv!(UpperTriangular(R), y, 1:k) (dense/linalg.jl:96-106)"""
    for j in range(k - 1, -1, -1):
        if R[j, j] == 0.0:
            raise ZeroDivisionError(f"SingularException({j + 1})")
        y[j] = y[j] / R[j, j]
        y[:j] -= R[:j, j] * y[j]
    return y
