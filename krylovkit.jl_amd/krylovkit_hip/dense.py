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


# ---- Schur machinery of the Arnoldi eigensolver (dense/linalg.jl:152-300, 335-383), real Float64 case
def eigsort_general(which: str):
    """eigsort (eigsolve/eigsolve.jl:334-355) for a complex spectrum."""
    table = {"LM": (np.abs, True), "LR": (np.real, True), "SR": (np.real, False), "LI": (np.imag, True), "SI": (np.imag, False)}
    if which not in table:
        raise ValueError(f"invalid specification of which eigenvalues to target: which = {which}")
    return table[which]


def sortperm_general(values: np.ndarray, which: str) -> np.ndarray:
    by, rev = eigsort_general(which)
    key = by(np.asarray(values))
    return np.argsort(-key if rev else key, kind="stable")


def hschur(H: np.ndarray):
    """hschur!(H, Z) (dense/linalg.jl:152-154 -> LAPACK hseqr): real Schur form H = U T U' of an upper Hessenberg
    matrix; returns (T, U, values)."""
    T, U = sla.schur(H, output="real")
    return T, U, schur2eigvals(T)


def schur2eigvals(T: np.ndarray) -> np.ndarray:
    """schur2eigvals(T::Real) (dense/linalg.jl:166-189): eigenvalues in diagonal order, 2x2 blocks -> conjugate pairs."""
    n = T.shape[0]
    D = np.zeros(n, dtype=np.complex128)
    for i in range(n):
        if i < n - 1 and T[i + 1, i] != 0:
            halftr = (T[i, i] + T[i + 1, i + 1]) / 2
            diff = (T[i, i] - T[i + 1, i + 1]) / 2
            d = diff * diff + T[i, i + 1] * T[i + 1, i]
            D[i] = halftr + 1j * math.sqrt(-d)
        elif i > 0 and T[i, i - 1] != 0:
            halftr = (T[i, i] + T[i - 1, i - 1]) / 2
            diff = -(T[i, i] - T[i - 1, i - 1]) / 2
            d = diff * diff + T[i, i - 1] * T[i - 1, i]
            D[i] = halftr - 1j * math.sqrt(-d)
        else:
            D[i] = T[i, i]
    return D


def permuteschur(T: np.ndarray, Q: np.ndarray, order):
    """permuteschur!(T, Q, order) (dense/linalg.jl:356-383): bring eigenvalue order[i] to position i with LAPACK
    trexc, never splitting a 2x2 block.  Returns (T, Q, values)."""
    from scipy.linalg import lapack
    n = T.shape[0]
    p = [int(k) + 1 for k in order]            # 1-based like LAPACK
    T = np.asfortranarray(T, dtype=np.float64)
    Q = np.asfortranarray(Q, dtype=np.float64)
    i = 0
    while i < len(p):
        ifirst, ilast = p[i], i + 1
        single = ifirst == n or T[ifirst, ifirst - 1] == 0      # T[ifirst+1, ifirst] in 1-based terms
        if not single and not (i + 1 < len(p) and p[i + 1] == ifirst + 1):
            raise RuntimeError("cannot split 2x2 blocks when permuting schur decomposition")
        T, Q, info = lapack.dtrexc(T, Q, ifirst, ilast)
        if info != 0:
            raise RuntimeError(f"LAPACK trexc failed with info = {info}")
        step = 1 if single else 2
        for k in range(i + step, len(p)):
            if p[k] < p[i]:
                p[k] += step
        i += step
    return T, Q, schur2eigvals(T)


def schur2eigvecs(T: np.ndarray) -> np.ndarray:
    """schur2eigvecs(T::Real) (dense/linalg.jl:223-246, LAPACK trevc + pairing): unit-norm right eigenvectors of a
    real quasi-triangular T in diagonal order (complex for 2x2 blocks).  Done by back-substitution on the complex
    Schur form (rsf2csf); eigenvectors are defined up to a phase, which is fixed here by a real positive pivot."""
    n = T.shape[0]
    Tc, Zc = sla.rsf2csf(T, np.eye(n))
    lam = schur2eigvals(T)
    VR = np.zeros((n, n), dtype=np.complex128)
    diag = np.diag(Tc).copy()
    used = np.zeros(n, dtype=bool)
    for col in range(n):
        # position of lam[col] on the complex diagonal (rsf2csf may swap the two members of a pair)
        cand = [k for k in range(n) if not used[k]]
        k = min(cand, key=lambda q: abs(diag[q] - lam[col]))
        used[k] = True
        y = np.zeros(n, dtype=np.complex128)
        y[k] = 1.0
        for i in range(k - 1, -1, -1):
            den = Tc[i, i] - Tc[k, k]
            if den == 0:
                den = np.finfo(float).eps * max(abs(Tc[k, k]), 1.0)
            y[i] = -(Tc[i, i + 1:k + 1] @ y[i + 1:k + 1]) / den
        v = Zc @ y
        v /= np.linalg.norm(v)
        j = int(np.argmax(np.abs(v)))
        v *= np.conj(v[j]) / abs(v[j])
        VR[:, col] = v
    return VR


def restorearnoldiform(U: np.ndarray, H: np.ndarray, f: np.ndarray, keep: int, on_reflector=None):
    """_restorearnoldiform!(U, H, f, keep) (eigsolve/arnoldi.jl:466-480): put the residual row f[:keep] under the
    leading keep x keep block of the Schur form and chase it back to Hessenberg form with Householder reflectors
    from the bottom, accumulating them in U."""
    H[keep, :keep] = f[:keep]
    for j in range(keep, 0, -1):
        hb, hv, nu = householder(H[j, :j], j - 1)
        H[j, j - 1] = nu
        H[j, : j - 1] = 0.0
        rr = np.arange(j)
        lmul_householder(hb, hv, rr, H)
        rmul_householder(H, hb, hv, rr, slice(0, j))
        rmul_householder(U, hb, hv, rr)
    return U, H

