"""eigsolve family on the SURVEY section-8 path (src/eigsolve/): Lanczos with thick restarts, GKL svdsolve and
BlockLanczos -- the reference's host control flow re-stated over the device factorizations.  (The drivers SURVEY section 2
marks out of scope -- Arnoldi / Krylov-Schur eigsolve and schursolve, bieigsolve, geneigsolve -- are not part of the product
package; their round-1 mirrors live on as test infrastructure in tests/hostmirror_extras.py.)  In a Julia deployment these loops are KrylovKit's own, unchanged; this mirror exists
because the image has no Julia toolchain and issues the identical call sequence into libkrylov_hip.so.  Small dense
work (k x k) is done on the host exactly as in the reference."""
from __future__ import annotations

import math
from typing import List, Optional

import numpy as np

from . import dense
from .algorithms import *  # noqa: F401,F403  (algorithm structs + ConvergenceInfo)
from .algorithms import ConvergenceInfo
from .core import DeviceBasis, HipVec, KrylovDefaults, Orthogonalizer, SparseOperator
from .factorizations import (ArnoldiIterator, GKLIterator, LanczosIterator, _as_operator, expand_, initialize,
                             initialize_, shrink_)


# -------------------------------------------------------------------- eigsolve (Lanczos)
def eigsolve(A, x0, howmany: int = 1, which: str = "LM", alg: Optional[Lanczos] = None, *, return_device: bool = False,
             iterator=None, **kw):
    """eigsolve(A, x0, howmany, which, alg::Lanczos) (src/eigsolve/lanczos.jl:1-155).

    A: SparseOperator or scipy.sparse matrix (must be symmetric).  Returns
    (values, vectors, ConvergenceInfo); vectors are numpy arrays unless return_device."""
    if isinstance(alg, Arnoldi):
        from .scope import OutOfScopeError
        raise OutOfScopeError("eigsolve with alg::Arnoldi (src/eigsolve/arnoldi.jl) is outside this package's scope (SURVEY section 2); "
                              "the Arnoldi FACTORIZATION (expand!) is what linsolve(GMRES) / exponentiate use.  Host transliteration with "
                              "the tests: tests/hostmirror_extras.py::eigsolve_arnoldi")
    alg = alg or Lanczos(**kw)
    krylovdim, maxiter = alg.krylovdim, alg.maxiter
    if howmany > krylovdim:
        raise ValueError(f"krylov dimension {krylovdim} too small to compute {howmany} eigenvalues")
    if iterator is not None:  # a caller-built iterator with the same initialize / expand interface (e.g. a split-phase sharded one)
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


