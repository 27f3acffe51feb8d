"""eigsolve / linsolve / svdsolve drivers: the reference's host control flow
(src/eigsolve/lanczos.jl, src/linsolve/gmres.jl, src/eigsolve/svdsolve.jl) re-stated over the
device factorizations.  In a Julia deployment these loops are KrylovKit's own, unchanged (they
only touch the L1/L2/L3 surface that KrylovKitHIP.jl overloads -- INTEGRATION.md); this Python
mirror exists because the image has no Julia toolchain, and issues the identical call sequence
into libkrylov_hip.so.  Small dense work (k x k) is done on the host exactly as in the reference.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import List, Optional

import numpy as np

from . import dense
from .core import DeviceBasis, HipVec, KrylovDefaults, ModifiedGramSchmidt, Orthogonalizer, SparseOperator
from .factorizations import (ArnoldiIterator, GKLIterator, LanczosIterator, _as_operator, expand_, initialize,
                             initialize_, shrink_)


@dataclass
class ConvergenceInfo:  # KrylovKit.jl:212-218
    converged: int
    residual: object
    normres: object
    numiter: int
    numops: int


# -------------------------------------------------------------------- algorithm structs
@dataclass
class Lanczos:  # algorithms.jl:110-127
    orth: Orthogonalizer = KrylovDefaults.orth
    krylovdim: int = KrylovDefaults.krylovdim
    maxiter: int = KrylovDefaults.maxiter
    tol: float = KrylovDefaults.tol
    eager: bool = False
    verbosity: int = 0


@dataclass
class Arnoldi:  # algorithms.jl:235-252
    orth: Orthogonalizer = KrylovDefaults.orth
    krylovdim: int = KrylovDefaults.krylovdim
    maxiter: int = KrylovDefaults.maxiter
    tol: float = KrylovDefaults.tol
    eager: bool = False
    verbosity: int = 0


@dataclass
class GMRES:  # algorithms.jl:373-390
    orth: Orthogonalizer = KrylovDefaults.orth
    maxiter: int = KrylovDefaults.maxiter
    krylovdim: int = KrylovDefaults.krylovdim
    tol: float = KrylovDefaults.tol
    verbosity: int = 0


@dataclass
class GKL:  # algorithms.jl:200-217
    orth: Orthogonalizer = KrylovDefaults.orth
    krylovdim: int = KrylovDefaults.krylovdim
    maxiter: int = KrylovDefaults.maxiter
    tol: float = KrylovDefaults.tol
    eager: bool = False
    verbosity: int = 0


# -------------------------------------------------------------------- eigsolve (Lanczos)
def eigsolve(A, x0, howmany: int = 1, which: str = "LM", alg: Optional[Lanczos] = None, *, return_device: bool = False,
             iterator=None, **kw):
    """eigsolve(A, x0, howmany, which, alg::Lanczos) (src/eigsolve/lanczos.jl:1-155).

    A: SparseOperator or scipy.sparse matrix (must be symmetric).  Returns
    (values, vectors, ConvergenceInfo); vectors are numpy arrays unless return_device.
    With alg::Arnoldi the general (non-symmetric) method of src/eigsolve/arnoldi.jl is used."""
    if isinstance(alg, Arnoldi):
        return _eigsolve_arnoldi(A, x0, howmany, which, alg)
    alg = alg or Lanczos(**kw)
    krylovdim, maxiter = alg.krylovdim, alg.maxiter
    if howmany > krylovdim:
        raise ValueError(f"krylov dimension {krylovdim} too small to compute {howmany} eigenvalues")
    if iterator is not None:  # e.g. dist.DistLanczosIterator: same control flow, sharded vectors
        it = iterator
    else:
        it = LanczosIterator(_as_operator(A), x0, alg.orth, True, capacity=krylovdim + 2)
    fact = initialize(it)
    numops = 1
    numiter = 1
    tol = alg.tol
    HH = np.zeros((krylovdim + 1, krylovdim))
    converged = 0
    D = U = f = None
    while True:
        beta = fact.normres
        K = len(fact)
        if K == krylovdim or beta <= tol or (alg.eager and K >= howmany):  # :45
            a, b = fact.rayleighquotient()
            if K == 1:
                D = np.array([a[0]])
                U = np.ones((1, 1))
                f = np.array([beta])
                converged = int(beta <= tol)
            else:
                D, U = dense.tridiageigh(a, b)  # :59
                p = dense.sortperm(D, which)
                D, U = D[p], np.array(U[:, p])
                f = U[K - 1, :] * beta  # :61
                converged = 0
                while converged < K and abs(f[converged]) <= tol:
                    converged += 1
            if converged >= howmany or beta <= tol:
                break
        if K < krylovdim:  # :77-79
            fact = expand_(it, fact)
            numops += 1
        else:  # thick restart :80-116
            if numiter == maxiter:
                break
            keep = (3 * krylovdim + 2 * converged) // 5
            H = HH[: keep + 1, :keep]
            H[:] = 0.0
            for j in range(keep):
                H[j, j] = D[j]
                H[keep, j] = f[j]
            for j in range(keep, 0, -1):  # :94-101
                hb, hv, nu = dense.householder(H[j, :j], j - 1)
                H[j, j - 1] = nu
                H[j, : j - 1] = 0.0
                rr = np.arange(j)
                dense.lmul_householder(hb, hv, rr, H)
                dense.rmul_householder(H, hb, hv, rr, slice(0, j))
                dense.rmul_householder(U, hb, hv, rr)
            for j in range(keep):
                fact.alphas[j] = H[j, j]
                fact.betas[j] = H[j + 1, j]
            B = fact.basis()
            B.basistransform(U[:, :keep])  # :109
            HipVec(B, keep).scale_from_(fact.r, 1.0 / beta)  # B[keep+1] = scale!!(r, 1/beta)  :111
            fact = shrink_(fact, keep)  # :114
            if hasattr(it, "recompute_gram"):
                it.recompute_gram(fact)
            numiter += 1
    hm = howmany
    if converged > howmany:
        hm = converged
    elif len(D) < howmany:
        hm = len(D)
    values = D[:hm]
    Vc = U[:, :hm]
    B = fact.basis()
    K = len(fact)
    out = DeviceBasis(B.n, max(hm, 1), B.ctx)
    for i in range(hm):  # vectors = [B*v for v in cols(V)]   :131-133
        B.times(Vc[:, i], HipVec(out, i), 0, K)
    out.length = hm
    normres = np.abs(f[:hm])
    info = ConvergenceInfo(converged, None, normres, numiter, numops)
    if return_device:
        return values, out, info
    vectors = [out.download(i) for i in range(hm)]
    info.residual = [fact.r.get() * Vc[-1, i] for i in range(hm)]  # :134-136
    return values, vectors, info


# -------------------------------------------------------------------- eigsolve / schursolve (Arnoldi)
def _set_packed_hessenberg(fact, H: np.ndarray, K: int):
    """copy!(rayleighquotient(fact), H) (eigsolve/arnoldi.jl:443): write the K x K dense H back into the packed
    Hessenberg storage (dense/packedhessenberg.jl:32-48)."""
    from .factorizations import packed_index
    for j in range(1, K + 1):
        for i in range(1, min(j + 1, K) + 1):
            fact.H[packed_index(i, j)] = float(H[i - 1, j - 1])


def _schursolve(A, x0, howmany: int, which: str, alg: Arnoldi):
    """_schursolve (src/eigsolve/arnoldi.jl:361-464): Krylov-Schur restarted Arnoldi.  Device work: kk_arnoldi_expand,
    kk_basistransform, one scaled copy per restart; the K x K Schur algebra is host LAPACK as in the reference."""
    krylovdim, maxiter = alg.krylovdim, alg.maxiter
    if howmany > krylovdim:
        raise ValueError(f"krylov dimension {krylovdim} too small to compute {howmany} eigenvalues")
    numiter = 1
    it = ArnoldiIterator(_as_operator(A), x0, alg.orth, capacity=krylovdim + 2)
    fact = initialize(it)
    numops = 1
    tol = alg.tol
    converged = 0
    T = U = f = None
    while True:
        beta = fact.normres
        K = len(fact)
        if K == krylovdim or beta <= tol or (alg.eager and K >= howmany):   # process  :389
            T, U, values = dense.hschur(fact.rayleighquotient())
            T, U, values = dense.permuteschur(T, U, dense.sortperm_general(values, which))
            f = U[K - 1, :] * beta                                           # :400
            converged = 0
            while converged < K and abs(f[converged]) <= tol:
                converged += 1
            if 0 < converged < K and T[converged, converged - 1] != 0:       # do not split a 2x2 block  :405
                converged -= 1
            if converged >= howmany or beta <= tol:
                break
        if K < krylovdim:                                                    # expand  :416
            fact = expand_(it, fact)
            numops += 1
        else:                                                                # shrink  :419-449
            if numiter == maxiter:
                break
            keep = (3 * krylovdim + 2 * converged) // 5
            if T[keep, keep - 1] != 0:
                if keep > 1:
                    keep -= 1
                else:
                    keep += 1
                    if krylovdim == 2:
                        break
            H = np.array(T)                                                  # the K x K view the reference reuses
            dense.restorearnoldiform(U, H, f, keep)                          # :441
            _set_packed_hessenberg(fact, H, K)
            B = fact.basis()
            B.basistransform(np.ascontiguousarray(U[:, :keep]))              # :444
            HipVec(B, keep).scale_from_(fact.r, 1.0 / beta)                  # B[keep+1] = scale!!(residual, 1/beta)
            fact = shrink_(fact, keep)
            numiter += 1
    return T, U, fact, converged, numiter, numops


def _arnoldi_howmany(T, fact, howmany, converged):
    hm = howmany                                                             # :286-293
    if howmany < len(fact) and T[howmany, howmany - 1] != 0:
        hm += 1
    elif T.shape[0] < howmany:
        hm = T.shape[0]
    if converged > howmany:
        hm = converged
    return hm


def _times_complex(B: DeviceBasis, K: int, coeffs: np.ndarray):
    """[B * v for v in cols(V)] with complex coefficient vectors: real and imaginary parts are two real kk_unproject
    calls on the device; the complex vector is assembled on the host."""
    out = DeviceBasis(B.n, 2, B.ctx)
    vecs = []
    for i in range(coeffs.shape[1]):
        c = coeffs[:, i]
        B.times(np.ascontiguousarray(c.real), HipVec(out, 0), 0, K)
        re = out.download(0)
        if np.any(c.imag != 0):
            B.times(np.ascontiguousarray(c.imag), HipVec(out, 1), 0, K)
            vecs.append(re + 1j * out.download(1))
        else:
            vecs.append(re)
    return vecs


def schursolve(A, x0, howmany: int = 1, which: str = "LM", alg: Optional[Arnoldi] = None, **kw):
    """schursolve(A, x0, howmany, which, alg::Arnoldi) (src/eigsolve/arnoldi.jl:237-275):
    returns (T, vectors, values, info) with A * vectors ~ vectors * T."""
    alg = alg or Arnoldi(**kw)
    T, U, fact, converged, numiter, numops = _schursolve(A, x0, howmany, which, alg)
    hm = _arnoldi_howmany(T, fact, howmany, converged)
    TT = T[:hm, :hm]
    values = dense.schur2eigvals(TT)
    K = len(fact)
    vectors = _times_complex(fact.basis(), K, U[:, :hm])
    r = fact.r.get()
    residuals = [r * U[K - 1, i] for i in range(hm)]
    normres = np.array([fact.normres * abs(U[K - 1, i]) for i in range(hm)])
    return TT, vectors, values, ConvergenceInfo(converged, residuals, normres, numiter, numops)


def _eigsolve_arnoldi(A, x0, howmany: int, which: str, alg: Arnoldi):
    """eigsolve(A, x0, howmany, which, alg::Arnoldi) (src/eigsolve/arnoldi.jl:277-316)."""
    T, U, fact, converged, numiter, numops = _schursolve(A, x0, howmany, which, alg)
    hm = _arnoldi_howmany(T, fact, howmany, converged)
    TT = T[:hm, :hm]
    values = dense.schur2eigvals(TT)
    V = U[:, :hm] @ dense.schur2eigvecs(TT)                                   # :297
    K = len(fact)
    vectors = _times_complex(fact.basis(), K, V)
    r = fact.r.get()
    residuals = [r * V[K - 1, i] for i in range(hm)]
    normres = np.array([fact.normres * abs(V[K - 1, i]) for i in range(hm)])
    return values, vectors, ConvergenceInfo(converged, residuals, normres, numiter, numops)


# -------------------------------------------------------------------- linsolve (GMRES)
def linsolve(A, b, x0=None, alg: Optional[GMRES] = None, a0: float = 0.0, a1: float = 1.0, *, atol: Optional[float] = None,
             rtol: Optional[float] = None, return_device: bool = False, **kw):
    """linsolve(operator, b, x0, alg::GMRES, a0, a1) (src/linsolve/gmres.jl:1-151), with the
    tolerance handling of the front-end (`tol = max(atol, rtol*norm(b))`, linsolve/linsolve.jl:135-140).
    Like the reference's method table, `alg::CG` and `alg::BiCGStab` select those solvers (linsolve/cg.jl, bicgstab.jl)."""
    if isinstance(alg, CG):
        return linsolve_cg(A, b, x0, alg, a0, a1)
    if isinstance(alg, BiCGStab):
        return linsolve_bicgstab(A, b, x0, alg, a0, a1)
    op = _as_operator(A)
    ctx = op.ctx
    n = op.shape[0]
    b = np.asarray(b, dtype=np.float64)
    alg = alg or GMRES(**{k: v for k, v in kw.items() if k in ("orth", "maxiter", "krylovdim", "tol", "verbosity")})
    if atol is not None or rtol is not None:
        alg = GMRES(alg.orth, alg.maxiter, alg.krylovdim, max(atol or 0.0, (rtol or 0.0) * float(np.linalg.norm(b))))
    krylovdim, maxiter, tol = alg.krylovdim, alg.maxiter, alg.tol
    # work vectors: 0 = b, 1 = x, 2 = r, 3 = tmp
    W = DeviceBasis(n, 4, ctx)
    vb, vx, vr, vt = HipVec(W, 0), HipVec(W, 1), HipVec(W, 2), HipVec(W, 3)
    vb.set(b)
    if x0 is None:
        vx.zero_()
    else:
        vx.set(np.asarray(x0, dtype=np.float64))
    # r = b - a0 x0 - a1 A x0   :3-12
    op.apply(vx, vt)
    vr.scale_from_(vb, 1.0)
    if a0 != 0:
        vr.add_(vx, -a0)
    vr.add_(vt, -a1)
    beta = vr.norm()
    if beta < tol:
        x = W if return_device else vx.get()
        return x, ConvergenceInfo(1, vr.get(), beta, 0, 1)
    y = np.zeros(krylovdim + 1)
    gs: List[Optional[tuple]] = [None] * krylovdim
    R = np.zeros((krylovdim, krylovdim))
    numiter = 0
    numops = 1
    it = ArnoldiIterator(op, vr, alg.orth, capacity=krylovdim + 2)
    fact = initialize(it)
    numops += 1
    while True:
        numiter += 1
        y[0] = beta
        k = 1
        H = fact.rayleighquotient()
        R[0, 0] = a0 + a1 * H[0, 0]
        c, s, R[0, 0] = dense.givens(R[0, 0], a1 * fact.normres)
        gs[0] = (0, 1, c, s)
        y[1] = 0.0
        y[0], y[1] = c * y[0] + s * y[1], -s * y[0] + c * y[1]
        beta = abs(y[1])
        while R[k - 1, k - 1] != 0 and beta > tol and len(fact) < krylovdim:  # :55
            fact = expand_(it, fact)
            numops += 1
            k = len(fact)
            # new Hessenberg column straight from the packed storage
            base = ((k * k + k - 2) >> 1)
            hcol = fact.H[base: base + k]
            for i in range(k - 1):
                R[i, k - 1] = a1 * hcol[i]
            R[k - 1, k - 1] = a0 + a1 * hcol[k - 1]
            Rk = R[:, k - 1]
            for i in range(k - 1):  # :72-75
                i1, i2, c, s = gs[i]
                Rk[i1], Rk[i2] = c * Rk[i1] + s * Rk[i2], -s * Rk[i1] + c * Rk[i2]
            if math.hypot(R[k - 1, k - 1], a1 * fact.normres) < tol:  # :78-85
                c, s, y[k] = dense.givens(0.0, y[k - 1])
                gs[k - 1] = (k, k - 1, c, s)
                y[k - 1] = 0.0
                R[k - 1, k - 1] = 0.0
            else:
                c, s, R[k - 1, k - 1] = dense.givens(R[k - 1, k - 1], a1 * fact.normres)
                gs[k - 1] = (k - 1, k, c, s)
                y[k] = 0.0
                y[k - 1], y[k] = c * y[k - 1] + s * y[k], -s * y[k - 1] + c * y[k]
            beta = abs(y[k])
        kk = k - 1 if (R[k - 1, k - 1] == 0 and y[k - 1] == 0) else k  # :98-102
        dense.ldiv_upper(R, y, kk)
        V = fact.basis()
        V.unproject(vx, y[:k], 0, k, 1.0, 1.0)  # x += sum V[i] y[i]   :105-108
        if beta > tol and numiter < maxiter:  # :110-117
            fact.r.scale_(1.0 / fact.normres)  # push!(V, scale!!(w, 1/normres))
            V.length = k + 1
            for i in range(k):
                i1, i2, c, s = gs[i]
                V.rmul_givens(i1, i2, c, -s)  # rmul!(V, gs[i]')
            vr.scale_from_(HipVec(V, k), y[k])  # r = scale!!(r, V[k+1], y[k+1])
            V.length = k
        else:  # :119-132
            vr.scale_from_(vb, 1.0)
            op.apply_affine(vx, vt, a0, a1)
            vr.add_(vt, -1.0)
            numops += 1
            beta = vr.norm()
            if beta < tol:
                x = W if return_device else vx.get()
                return x, ConvergenceInfo(1, vr.get(), beta, numiter, numops)
        if numiter >= maxiter:
            x = W if return_device else vx.get()
            return x, ConvergenceInfo(0, vr.get(), beta, numiter, numops)
        it = ArnoldiIterator(op, vr, alg.orth, capacity=krylovdim + 2)  # :147-148
        fact = initialize_(it, fact)


# -------------------------------------------------------------------- svdsolve (GKL)
def svdsolve(A, x0, howmany: int = 1, which: str = "LR", alg: Optional[GKL] = None, **kw):
    """svdsolve(A, x0, howmany, which, alg::GKL) (src/eigsolve/svdsolve.jl:144-314)."""
    alg = alg or GKL(**kw)
    krylovdim, maxiter, tol = alg.krylovdim, alg.maxiter, alg.tol
    if howmany > krylovdim:
        raise ValueError(f"krylov dimension {krylovdim} too small to compute {howmany} singular values")
    op = _as_operator(A)
    numiter = 1
    it = GKLIterator(op, x0, alg.orth, capacity=krylovdim + 2)
    fact = initialize(it)
    numops = 2
    HH = np.zeros((krylovdim + 1, krylovdim))
    converged = 0
    P = Q = f = S = None
    while True:
        beta = fact.normres
        K = len(fact)
        if K == krylovdim or beta <= tol or (alg.eager and K >= howmany):
            P, S, Q = dense.bidiagsvd(fact.rayleighquotient())  # :198
            if which == "SR":
                P, S, Q = P[:, ::-1], S[::-1], Q[::-1, :]
            elif which != "LR":
                raise ValueError(f"invalid specification of which singular values to target: which = {which}")
            f = Q.T[K - 1, :] * beta  # :208
            converged = 0
            while converged < K and abs(f[converged]) < tol:
                converged += 1
            if converged >= howmany or beta <= tol:
                break
        if K < krylovdim:
            fact = expand_(it, fact)
            numops += 2
        else:
            if numiter == maxiter:
                break
            keep = (3 * krylovdim + 2 * converged) // 5
            U, V = fact.basis("U"), fact.basis("V")
            U.basistransform(np.ascontiguousarray(P[:, :keep]))       # :233
            V.basistransform(np.ascontiguousarray(Q.T[:, :keep]))     # :240
            HipVec(U, keep).scale_from_(fact.r, 1.0 / fact.normres)   # U[keep+1] = scale!!(r, 1/normres)  :249
            H = HH[: keep + 1, :keep]
            H[:] = 0.0
            for j in range(keep):
                H[j, j] = S[j]
                H[keep, j] = f[j]
            for j in range(keep, 0, -1):  # :257-269
                hb, hv, nu = dense.householder(H[j, :j], j - 1)
                H[j, j - 1] = nu
                H[j, : j - 1] = 0.0
                rr = np.arange(j)
                dense.rmul_householder(H, hb, hv, rr, slice(0, j))
                V.rmul_householder(hb, hv, 0, j)  # rmul!(V, h')
                hb, hv, nu = dense.householder(H[:j, j - 1], j - 1)
                H[j - 1, j - 1] = nu
                H[: j - 1, j - 1] = 0.0
                dense.lmul_householder(hb, hv, rr, H[:, : j - 1])
                U.rmul_householder(hb, hv, 0, j)  # rmul!(U, h')
            for j in range(keep):
                fact.alphas[j] = H[j, j]
                fact.betas[j] = H[j + 1, j]
            fact = shrink_(fact, keep)
            numiter += 1
    if converged > howmany:
        howmany = converged
    values = S[:howmany]
    Pv, Qv = P[:, :howmany], Q[:howmany, :]
    K = len(fact)
    U, V = fact.basis("U"), fact.basis("V")
    outU = DeviceBasis(U.n, max(howmany, 1), op.ctx)
    outV = DeviceBasis(V.n, max(howmany, 1), op.ctx)
    left, right = [], []
    for i in range(howmany):
        U.times(Pv[:, i], HipVec(outU, i), 0, K)
        V.times(Qv[i, :], HipVec(outV, i), 0, K)
        left.append(outU.download(i))
        right.append(outV.download(i))
    normres = np.abs(f[:howmany])
    return values, left, right, ConvergenceInfo(converged, None, normres, numiter, numops)


# -------------------------------------------------------------------- eigsolve (BlockLanczos)
@dataclass
class BlockLanczos:  # algorithms.jl:152-171 (blockkrylovdim default 100, algorithms.jl:561)
    orth: Orthogonalizer = KrylovDefaults.orth
    krylovdim: int = 100
    maxiter: int = KrylovDefaults.maxiter
    tol: float = KrylovDefaults.tol
    qr_tol: float = KrylovDefaults.tol
    eager: bool = False
    verbosity: int = 0


def eigsolve_block(A, x0, howmany: int = 1, which: str = "SR", alg: Optional[BlockLanczos] = None, **kw):
    """eigsolve(A, x0::Block, howmany, which, alg::BlockLanczos) (src/eigsolve/blocklanczos.jl:1-144).
    x0: list of numpy start vectors (the Block)."""
    from .factorizations import Block, BlockLanczosIterator, block_inner

    alg = alg or BlockLanczos(**kw)
    krylovdim, maxiter, tol = alg.krylovdim, alg.maxiter, alg.tol
    if howmany > krylovdim:
        raise ValueError(f"krylov dimension {krylovdim} too small to compute {howmany} eigenvalues")
    op = _as_operator(A)
    bs = len(x0)
    it = BlockLanczosIterator(op, x0, krylovdim + bs, alg.orth, alg.qr_tol)
    fact = it.initialize()
    numops = bs + 1
    numiter = 1
    converged = 0
    normresiduals = D = U = None
    while True:
        K = len(fact)
        beta = fact.normres
        if K >= krylovdim or beta <= tol or (alg.eager and K >= howmany):  # :39
            BTD = fact.H[:K, :K]
            D, U = np.linalg.eigh((BTD + BTD.T) / 2)  # eigen(Hermitian(BTD))  :42
            p = dense.sortperm(D, which)
            D, U = D[p], np.array(U[:, p])
            bs_R = fact.R_size
            r = fact.residual()
            UU = U[K - bs_R:K, :]
            Rm = block_inner(r, r)  # :51
            normresiduals = np.sqrt(np.maximum(np.einsum("ik,ij,jk->k", UU, Rm, UU), 0.0))
            converged = 0
            while converged < K and normresiduals[converged] <= tol:
                converged += 1
            if converged >= howmany or beta <= tol:
                break
        if K < krylovdim:
            fact = it.expand(fact)
            numops += fact.R_size
        else:  # :68-104
            if numiter >= maxiter:
                break
            keep = max((3 * krylovdim + 2 * converged) // (5 * bs), 1) * bs
            H = np.zeros((keep + bs, keep))
            for j in range(keep):
                H[j, j] = D[j]
                H[keep:, j] = U[K - bs:K, j]
            for j in range(keep, 0, -1):  # :80-87
                hb, hv, nu = dense.householder(H[j + bs - 1, :j], j - 1)
                H[j + bs - 1, j - 1] = nu
                H[j + bs - 1, : j - 1] = 0.0
                rr = np.arange(j)
                dense.lmul_householder(hb, hv, rr, H)
                dense.rmul_householder(H, hb, hv, rr, slice(0, j + bs - 1))
                dense.rmul_householder(U, hb, hv, rr)
            fact.H[:] = 0.0
            Hk = H[:keep, :keep]
            fact.H[:keep, :keep] = (Hk + Hk.T) / 2
            B = fact.basis()
            B.basistransform(U[:, :keep])  # :92
            view_H = H[keep + bs - bs_R:keep + bs, keep - bs_R:keep]
            B.length = fact.r_col + bs_R  # address the residual block as a basis range
            B.basistransform(np.ascontiguousarray(view_H), c0=fact.r_col)  # basistransform!(R_new, view_H)  :96
            B.length = keep
            fact.k = keep
            numiter += 1
    hm = howmany
    if converged > howmany:
        hm = converged
    elif len(D) < howmany:
        hm = len(D)
    values = D[:hm]
    K = len(fact)
    B = fact.basis()
    out = DeviceBasis(B.n, max(hm, 1), op.ctx)
    vectors = []
    for i in range(hm):
        B.times(U[:, i], HipVec(out, i), 0, K)
        vectors.append(out.download(i))
    return values, vectors, ConvergenceInfo(converged, None, normresiduals[:hm], numiter, numops)


# -------------------------------------------------------------------- linsolve (CG)
@dataclass
class CG:  # algorithms.jl:325-337
    maxiter: int = KrylovDefaults.maxiter
    tol: float = KrylovDefaults.tol
    verbosity: int = 0


def linsolve_cg(A, b, x0=None, alg: Optional[CG] = None, a0: float = 0.0, a1: float = 1.0, **kw):
    """linsolve(operator, b, x0, alg::CG, a0, a1) (src/linsolve/cg.jl:1-103) for a symmetric positive
    definite a0 + a1*A.  Per iteration: one SpMV with the fused <p, q>, one fused update
    (x += alpha p; r -= alpha q; |r|), one axpby (p = r + beta p)."""
    import ctypes as C
    from ._lib import check
    op = _as_operator(A)
    n = op.shape[0]
    alg = alg or CG(**kw)
    maxiter, tol = alg.maxiter, alg.tol
    W = DeviceBasis(n, 5, op.ctx)  # 0 = b, 1 = x, 2 = r, 3 = p, 4 = q
    vb, vx, vr, vp, vq = (HipVec(W, i) for i in range(5))
    lib = W._lib
    vb.set(np.asarray(b, dtype=np.float64))
    if x0 is None:
        vx.zero_()
    else:
        vx.set(np.asarray(x0, dtype=np.float64))
    op.apply(vx, vq)                      # y0 = apply(operator, x0)   :3
    vr.scale_from_(vb, 1.0)
    if a0 != 0:
        vr.add_(vx, -a0)
    vr.add_(vq, -a1)
    normr = vr.norm()
    numops, numiter = 1, 0
    if normr < tol:
        return vx.get(), ConvergenceInfo(1, vr.get(), normr, numiter, numops)

    def iterate(beta, first, rho):
        """[p = r + beta p]; q = (a0 + a1 A) p; alpha = rho/<p,q>; x += alpha p; r -= alpha q -> |r|  (one host sync)"""
        pq, nr = C.c_double(), C.c_double()
        check(lib.kk_cg_iterate(op.handle, W.handle, 1, 2, 3, 4, a0, a1, beta, int(first), rho, C.byref(pq), C.byref(nr)))
        return nr.value

    rho = normr ** 2
    vp.scale_from_(vr, 1.0)               # :33-34
    normr = iterate(0.0, True, rho)
    rho_old, rho = rho, normr ** 2
    beta = rho / rho_old
    numops += 1
    numiter += 1
    if normr < tol:
        return vx.get(), ConvergenceInfo(1, vr.get(), normr, numiter, numops)
    while True:                           # :60-101
        normr = iterate(beta, False, rho)  # p = add!!(p, r, 1, beta); q = apply; alpha = rho/inner(p,q); x, r updates
        if normr < tol:                   # recompute explicitly   :67-72
            vr.scale_from_(vb, 1.0)
            op.apply_affine(vx, vq, a0, a1)
            vr.add_(vq, -1.0)
            normr = vr.norm()
            rho = normr ** 2
            beta = 0.0
        else:
            rho_old, rho = rho, normr ** 2
            beta = rho / rho_old
        numops += 1
        numiter += 1
        if normr < tol:
            return vx.get(), ConvergenceInfo(1, vr.get(), normr, numiter, numops)
        if numiter >= maxiter:
            return vx.get(), ConvergenceInfo(0, vr.get(), normr, numiter, numops)


# -------------------------------------------------------------------- linsolve (BiCGStab)
@dataclass
class BiCGStab:  # algorithms.jl:469-481
    maxiter: int = KrylovDefaults.maxiter
    tol: float = KrylovDefaults.tol
    verbosity: int = 0


def linsolve_bicgstab(A, b, x0=None, alg: Optional[BiCGStab] = None, a0: float = 0.0, a1: float = 1.0, **kw):
    """linsolve(operator, b, x0, alg::BiCGStab, a0, a1) (src/linsolve/bicgstab.jl:1-203) for a general a0 + a1*A.
    Per iteration: two SpMVs (each with its inner products fused) and three fused vector kernels
    (kk_bicgstab_half / kk_bicgstab_full); rho, sigma, alpha, omega stay on the device, the host reads the two
    norms the reference compares with tol."""
    import ctypes as C
    from ._lib import check
    op = _as_operator(A)
    n = op.shape[0]
    alg = alg or BiCGStab(**kw)
    maxiter, tol = alg.maxiter, alg.tol
    # 0 = b, 1 = x, 2 = r, 3 = r_shadow, 4/9 = p (double buffer), 5/10 = v (double buffer), 6 = s, 7 = t, 8 = xhalf
    W = DeviceBasis(n, 11, op.ctx)
    vb, vx, vr, vrs, _, _, vs, vt, vh, _, _ = (HipVec(W, i) for i in range(11))
    lib = W._lib
    cur, alt = (4, 5), (9, 10)

    def colarr(pv, prev):
        return (C.c_int * 9)(1, 2, 3, pv[0], pv[1], 6, 7, prev[0], prev[1])

    vb.set(np.asarray(b, dtype=np.float64))
    if x0 is None:
        vx.zero_()
    else:
        vx.set(np.asarray(x0, dtype=np.float64))
    op.apply(vx, vt)                      # y0 = apply(operator, x0)   :3
    vr.scale_from_(vb, 1.0)
    if a0 != 0:
        vr.add_(vx, -a0)
    vr.add_(vt, -a1)
    normr = vr.norm()
    numops, numiter = 1, 0
    if normr < tol:                       # :22-28
        return vx.get(), ConvergenceInfo(1, vr.get(), normr, numiter, numops)
    numiter += 1
    vrs.scale_from_(vr, 1.0)              # shadow residual   :35
    rho = vrs.inner(vr)
    if np.isclose(rho, 0.0):              # :39-46
        return vx.get(), ConvergenceInfo(0, vr.get(), normr, numiter, numops)
    HipVec(W, cur[0]).scale_from_(vr, 1.0)   # p = r
    first = True
    mode = 1       # 1: first iteration; 0: rho on the device; 2: rho handed over again; 3: half already enqueued
    cols = colarr(cur, cur)
    snorm, alpha, rnorm, rho_c, omega = (C.c_double() for _ in range(5))
    while True:
        if not first:
            numiter += 1
        # BiCG half: p update, v = A p, alpha, s = r - alpha v (and, run ahead of the host, t = A s)
        check(lib.kk_bicgstab_half(op.handle, W.handle, cols, a0, a1, mode, rho, C.byref(snorm), C.byref(alpha)))
        numops += 1
        normr = snorm.value
        redo_t = 0
        if normr < tol:                   # explicit residual at the half step   :65-80 / :142-157
            vh.scale_from_(vx, 1.0)
            vh.add_(HipVec(W, cols[3]), alpha.value)      # xhalf = x + alpha p
            op.apply_affine(vh, vt, a0, a1)
            vs.scale_from_(vb, 1.0)
            vs.add_(vt, -1.0)
            numops += 1
            normr_act = vs.norm()
            if normr_act < tol:
                return vh.get(), ConvergenceInfo(1, vs.get(), normr_act, numiter, numops)
            redo_t = 1                    # s was replaced: t = A s has to be recomputed
        numops += 1                       # t = apply(operator, s, a0, a1)   :83 / :163
        last = (not first) and numiter >= maxiter
        nxt = None if last else colarr(alt, cur)     # next half into the other p/v buffers, reading the current ones
        check(lib.kk_bicgstab_full(op.handle, W.handle, cols, a0, a1, redo_t, nxt, C.byref(rnorm), C.byref(rho_c),
                                   C.byref(omega)))
        normr = rnorm.value
        rho = rho_c.value
        mode = 3
        if normr < tol:                   # explicit residual at the full step   :94-110 / :175-190
            op.apply_affine(vx, vt, a0, a1)
            vr.scale_from_(vb, 1.0)
            vr.add_(vt, -1.0)
            numops += 1
            normr_act = vr.norm()
            if normr_act < tol:
                return vx.get(), ConvergenceInfo(1, vr.get(), normr_act, numiter, numops)
            rho = vrs.inner(vr)           # r was replaced: the next rho = <r_shadow, r> is that of the NEW r   :120
            mode = 2                      # ... and the run-ahead half (old r, old rho) is discarded and redone
        if last:                          # :191-198
            return vx.get(), ConvergenceInfo(0, vr.get(), normr, numiter, numops)
        first = False
        cols = nxt
        cur, alt = alt, cur


# -------------------------------------------------------------------- lssolve (LSMR)
@dataclass
class LSMR:  # algorithms.jl:506-521
    orth: Orthogonalizer = field(default_factory=ModifiedGramSchmidt)
    maxiter: int = KrylovDefaults.maxiter
    krylovdim: int = KrylovDefaults.krylovdim
    tol: float = KrylovDefaults.tol
    verbosity: int = 0


def lssolve(A, b, alg: Optional[LSMR] = None, lam: float = 0.0, *, rtol: float = KrylovDefaults.tol,
            atol: float = KrylovDefaults.tol, **kw):
    """lssolve(operator, b, alg::LSMR, lambda) (src/lssolve/lsmr.jl:1-151; defaults lssolve.jl:101-110): minimise
    |A x - b|^2 + |lambda x|^2.  Device data: (u, r, Ah, Ahbar, Av) in the row space of A, the circular buffer of the
    `krylovdim` most recent v's plus (x, h, hbar) in its domain.  Per iteration: A v, A' u, one re-orthogonalisation
    against the buffer, and three fused vector updates (kk_lsmr_step_u, 2 x kk_lsmr_update)."""
    import ctypes as C
    from ._lib import check
    op = _as_operator(A)
    nu, nv = op.shape
    b = np.asarray(b, dtype=np.float64)
    if alg is None:
        kw.setdefault("tol", max(atol, rtol * float(np.linalg.norm(b))))
        alg = LSMR(**kw)
    K, maxiter, tol = alg.krylovdim, alg.maxiter, alg.tol
    BU = DeviceBasis(nu, 5, op.ctx)            # 0 = u, 1 = r, 2 = Ah, 3 = Ahbar, 4 = Av
    BV = DeviceBasis(nv, K + 5, op.ctx)        # 0..K-1 = V (circular), K / K+4 = work vectors, K+1 = x, K+2 = h, K+3 = hbar
    lib = BU._lib
    u, r, Ah, Ahbar, Av = (HipVec(BU, i) for i in range(5))
    w, x, h, hbar = (HipVec(BV, K + i) for i in range(4))
    u.set(b)
    op.apply(u, w, transpose=True)             # v0 = apply_adjoint(operator, u0)   :4
    beta = u.norm()
    u.scale_(1 / beta)
    w.scale_(1 / beta)
    alpha = w.norm()
    v = HipVec(BV, 0).scale_from_(w, 1 / alpha)
    nV = 1                                     # length(V)
    alphabar, zetabar, rho, theta, rhobar, cbar, sbar = alpha, alpha * beta, 1.0, 0.0, 1.0, 1.0, 0.0
    abszetabar = abs(zetabar)
    x.zero_(); hbar.zero_(); Ah.zero_(); Ahbar.zero_()
    h.scale_from_(v, 1.0)
    r.scale_from_(u, beta)
    numiter, numops = 0, 1

    def result(conv):
        return x.get(), ConvergenceInfo(conv, r.get(), abszetabar, numiter, numops)

    if abszetabar < tol:                       # :48-58
        return result(1)
    bt = C.c_double()
    while True:
        numiter += 1
        op.apply(v, Av)                        # :63
        numops += 1
        # Ah = Av - (theta/rho) Ah ; u = Av - alpha u ; beta = |u|     :64-68
        check(lib.kk_lsmr_step_u(BU.handle, 4, 2, 0, theta / rho, alpha, C.byref(bt)))
        beta = bt.value
        if beta > tol:
            u.scale_(1 / beta)
            # v_new = A' u - beta v   :73  (after an alpha <= tol step v itself lives in a work column: use the other)
            w = HipVec(BV, K + 4) if v.col == K else HipVec(BV, K)
            op.apply(u, w, transpose=True)
            w.add_(v, -beta)
            numops += 1
            if K > 1:
                _, alpha, _ = BV.orthogonalize(w, alg.orth, 0, nV)      # :76-78 (+ the norm of :80 fused)
            else:
                alpha = w.norm()
            if alpha > tol:
                slot = nV if numiter < K else (numiter % K)             # mod1(numiter + 1, K) - 1
                v = HipVec(BV, slot).scale_from_(w, 1 / alpha)
                if numiter < K:
                    nV += 1
            else:
                v = w
        alphahat = float(np.hypot(alphabar, lam))      # :92-94
        rhoold = rho                                   # :97-102
        rho = float(np.hypot(alphahat, beta))
        c = alphahat / rho
        s_ = beta / rho
        theta = s_ * alpha
        alphabar = c * alpha
        rhobarold = rhobar                             # :105-112
        thetabar = sbar * rho
        cbarrho = cbar * rho
        rhobar = float(np.hypot(cbarrho, theta))
        cbar = cbarrho / rhobar
        sbar = theta / rhobar
        zeta = cbar * zetabar
        zetabar = -sbar * zetabar
        c1 = thetabar * rho / (rhoold * rhobarold)
        c2 = zeta / (rho * rhobar)
        # hbar = h - c1 hbar ; x += c2 hbar ; h = v - (theta/rho) h      :115-121
        check(lib.kk_lsmr_update(BV.handle, K + 2, K + 3, K + 1, BV.handle, v.col, c1, c2, theta / rho))
        # Ahbar = Ah - c1 Ahbar ; r -= c2 Ahbar                           :116,119
        check(lib.kk_lsmr_update(BU.handle, 2, 3, 1, None, -1, c1, -c2, 0.0))
        abszetabar = abs(zetabar)
        if abszetabar <= tol:
            return result(1)
        if numiter >= maxiter:
            return result(0)



# -------------------------------------------------------------------- geneigsolve (Golub-Ye)
@dataclass
class GolubYe:  # algorithms.jl:310-325
    orth: Orthogonalizer = KrylovDefaults.orth
    krylovdim: int = KrylovDefaults.krylovdim
    maxiter: int = KrylovDefaults.maxiter
    tol: float = KrylovDefaults.tol
    verbosity: int = 0


def _checkposdef(z: float) -> float:  # KrylovKit.jl:143-148
    if not z > 0:
        raise ValueError(f"operator does not appear to be positive definite: diagonal element {z}")
    return z


def geneigsolve(AB, x0, howmany: int = 1, which: str = "SR", alg: Optional[GolubYe] = None, **kw):
    """geneigsolve((A, B), x0, howmany, which, alg::GolubYe) (src/eigsolve/golubye.jl:1-180) for a real symmetric A and a
    symmetric positive definite B, both device sparse operators.  The inner iteration is the Lanczos recurrence of
    A - rho B on the device basis (two SpMVs + the same orthogonalisation passes per step, golubye.jl:182-281); the
    projected K x K pencil is solved on the host (LAPACK sygvd through SciPy, as the reference does)."""
    from .factorizations import Block, block_inner, lanczos_recurrence_unfused
    alg = alg or GolubYe(**kw)
    if which in ("LI", "SI"):
        raise ValueError(f"Eigenvalue selector which = {which} invalid: real eigenvalues expected with Lanczos algorithm")
    A, B = (_as_operator(M) for M in AB)
    krylovdim, maxiter, tol, orth = alg.krylovdim, alg.maxiter, alg.tol, alg.orth
    if howmany > krylovdim:
        raise ValueError(f"krylov dimension {krylovdim} too small to compute {howmany} eigenvalues")
    n, ctx = A.shape[0], A.ctx
    cap = krylovdim + 2
    V, BV = DeviceBasis(n, cap, ctx), DeviceBasis(n, cap, ctx)
    Xv, Xr = DeviceBasis(n, cap, ctx), DeviceBasis(n, cap, ctx)     # Ritz vectors / residuals of the last process step
    S = DeviceBasis(n, 5, ctx)                                      # scratch: av, bv, vold, tmp, r
    av, bv, vold, tmp, rs = (HipVec(S, i) for i in range(5))

    v = HipVec(V, 0).set(np.asarray(x0, dtype=np.float64))
    A.apply(v, av); B.apply(v, bv)                                  # genapply  :7
    numops = 1
    beta0 = v.norm()
    if beta0 == 0:
        raise ValueError("initial vector should not have norm zero")
    xax, xbx = v.inner(av) / beta0 ** 2, v.inner(bv) / beta0 ** 2
    v.scale_(1 / beta0); av.scale_(1 / beta0); bv.scale_(1 / beta0)
    rho = xax / _checkposdef(xbx)
    r = rs.scale_from_(av, 1.0).add_(bv, -rho)                      # r = av - rho bv   :21
    HipVec(BV, 0).scale_from_(bv, 1.0)
    vold.scale_from_(v, 1.0)
    V.length = BV.length = 1
    HHA = np.zeros((krylovdim + 1, krylovdim + 1))
    numiter = 1
    alpha, beta = r.orthogonalize_against_(v, orth)                 # :44-45
    converged = 0
    values: List[float] = []
    nvec = 0
    normres: List[float] = []
    K = 1
    HHA[0, 0] = alpha
    by, rev = dense.eigsort(which)
    cur_v = cur_bv = None                                           # (v, bv) of the last Ritz pair looked at

    def recurrence(Kc, beta_old):
        """golubyerecurrence (:182-281) for V[Kc-1] = v; leaves w in `rs`, B v in BV[Kc-1]; returns (alpha, beta)."""
        nonlocal numops
        vK, bvK = HipVec(V, Kc - 1), HipVec(BV, Kc - 1)
        A.apply(vK, rs); B.apply(vK, bvK)
        numops += 1
        w = rs.add_(bvK, -rho)
        return lanczos_recurrence_unfused(V, 0, Kc, w, beta_old, orth)   # the six variants of :182-281

    def extend(vn: HipVec):
        """push a new (already orthonormalised) vector and its row / column of HHA   (:66-81 / :85-95)"""
        nonlocal K, numops
        A.apply(vn, av); B.apply(vn, bv)
        numops += 1
        av.add_(bv, -rho)
        h = V.project(av, 0, K)
        HHA[:K, K] = h
        HHA[K, :K] = h
        HHA[K, K] = vn.inner(av)
        HipVec(V, K).scale_from_(vn, 1.0)
        HipVec(BV, K).scale_from_(bv, 1.0)
        K += 1
        V.length = BV.length = K

    while True:
        beta = r.norm()
        if beta <= tol and K < howmany:                             # :59-67
            howmany = K
        if K == krylovdim - converged or beta <= tol:               # process  :68
            if numiter > 1:
                V.orthonormalize(vold, orth, 0, K)                  # orthonormalize!!(vold, V)   :64
                extend(vold)
            for i in range(converged):                              # re-add the converged Ritz vectors   :83-96
                tmp.scale_from_(HipVec(Xv, i), 1.0)
                V.orthonormalize(tmp, orth, 0, K)
                extend(tmp)
            M = block_inner(Block(V, 0, K), Block(BV, 0, K))        # buildHB!  :284-295
            HB = np.tril(M) + np.tril(M, -1).T
            for j in range(K):
                _checkposdef(HB[j, j])
            HA = HHA[:K, :K] + rho * HB
            import scipy.linalg as sla
            D, Z = sla.eigh(HA, HB)                                  # geneigh!  :102
            key = by(D)
            perm = np.argsort(-key if rev else key, kind="stable")
            converged = 0
            values, normres, nvec = [], [], 0
            for k in range(K):                                      # :110-133
                z = np.ascontiguousarray(Z[:, perm[k]])
                cur_v = V.times(z, HipVec(Xv, nvec), 0, K)          # v = unproject!!(zerovector, V, z)
                r = HipVec(Xr, nvec)
                A.apply(cur_v, r); B.apply(cur_v, bv)
                numops += 1
                rho = cur_v.inner(r) / _checkposdef(cur_v.inner(bv))
                r.add_(bv, -rho)
                beta = r.norm()
                if beta < tol * float(np.linalg.norm(z)):
                    converged += 1
                elif numiter < maxiter:
                    break
                values.append(rho)
                normres.append(beta)
                nvec += 1
                if k + 1 == howmany and numiter == maxiter:
                    break
            if converged >= howmany:
                howmany = converged
                break
        if K < krylovdim - converged:                               # expand  :143-157
            HipVec(V, K).scale_from_(r, 1 / beta)
            V.length = K + 1
            HHA[K, K - 1] = HHA[K - 1, K] = beta
            K += 1
            BV.length = K
            alpha, beta = recurrence(K, beta)
            r = rs
            HHA[K - 1, K - 1] = alpha
        else:                                                       # restart  :158-177
            if numiter == maxiter:
                break
            HHA[:] = 0.0
            K = 1
            invb = 1 / cur_v.norm()
            v = HipVec(V, 0).scale_from_(cur_v, invb)
            HipVec(BV, 0).scale_from_(bv, invb)
            r = rs.scale_from_(r, invb)
            V.length = BV.length = 1
            alpha, beta = r.orthogonalize_against_(v, orth)
            HHA[0, 0] = alpha
            numiter += 1
    vectors = [Xv.download(i) for i in range(nvec)]
    residuals = [Xr.download(i) for i in range(nvec)]
    return np.array(values), vectors, ConvergenceInfo(converged, residuals, np.array(normres), numiter, numops)
