"""eigsolve family (src/eigsolve/): Lanczos with thick restarts, Arnoldi / Krylov-Schur (eigsolve, schursolve), GKL
svdsolve, BlockLanczos and Golub-Ye geneigsolve -- the reference's host control flow re-stated over the device
factorizations.  In a Julia deployment these loops are KrylovKit's own, unchanged; this mirror exists
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
    (values, vectors, ConvergenceInfo); vectors are numpy arrays unless return_device.
    With alg::Arnoldi the general (non-symmetric) method of src/eigsolve/arnoldi.jl is used."""
    if isinstance(alg, Arnoldi):
        return _eigsolve_arnoldi(A, x0, howmany, which, alg)
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


# -------------------------------------------------------------------- bieigsolve (BiArnoldi)
def _bischursolve(A, v0, w0, howmany: int, which: str, alg):
    """_bischursolve (src/eigsolve/biarnoldi.jl:196-393): two-sided Krylov-Schur.  Two Arnoldi factorizations on the
    device -- V for A (fused kk_arnoldi_expand), W for A' (the adjoint as a function operator, un-fused sequence);
    their coupling M = W'V, the oblique corrections of the residuals and the K x K Schur algebra follow the reference
    (host LAPACK for the small matrices, kk_project / kk_unproject / kk_basistransform for everything N-long)."""
    import scipy.linalg as sla
    from .core import FunctionOperator
    krylovdim, maxiter, tol = alg.krylovdim, alg.maxiter, alg.tol
    if howmany > krylovdim:
        raise ValueError(f"krylov dimension {krylovdim} too small to compute {howmany} eigenvalues")
    op = _as_operator(A)
    n = op.shape[0]
    opT = FunctionOperator(lambda x, y: op.apply(x, y, True), n, op.ctx)      # apply_adjoint (apply.jl:15)
    numiter = 1
    itV = ArnoldiIterator(op, v0, alg.orth, capacity=krylovdim + 2)             # BiArnoldiIterator (factorizations/biarnoldi.jl:24-41)
    itW = ArnoldiIterator(opT, w0, alg.orth, capacity=krylovdim + 2)
    fV, fW = initialize(itV), initialize(itW)
    numops = 1
    R = DeviceBasis(n, 2, op.ctx)                                               # the two corrected residuals
    rV, rW = HipVec(R, 0), HipVec(R, 1)
    MM = np.zeros((krylovdim, krylovdim))
    MM[0, 0] = HipVec(fW.V, 0).inner(HipVec(fV.V, 0))
    converged = 0
    S = T = Q = Z = h = k = M = None
    MinvWv = MinvVw = None
    while True:
        bv, bw = fV.normres, fW.normres
        Lk = len(fV)
        if Lk == krylovdim or (bv <= tol and bw <= tol) or (alg.eager and Lk >= howmany):   # process  :232
            H, K = fV.rayleighquotient(), fW.rayleighquotient()
            M = MM[:Lk, :Lk]
            rV.scale_from_(fV.r, 1 / bv)                                        # v_{l+1}, w_{l+1}   :248-254
            rW.scale_from_(fW.r, 1 / bw)
            V, W = fV.basis(), fW.basis()
            Wv = W.project(rV, 0, Lk)                                           # W' v, V' w   :256-259
            Vw = V.project(rW, 0, Lk)
            lu = sla.lu_factor(M)                                               # :260-262
            MinvWv = sla.lu_solve(lu, Wv)
            MinvVw = sla.lu_solve(lu, Vw, trans=1)
            H[:, Lk - 1] += bv * MinvWv                                         # :263-264
            K[:, Lk - 1] += bw * MinvVw
            V.unproject(rV, MinvWv, 0, Lk, -1.0, 1.0)                           # oblique corrections   :265-268
            W.unproject(rW, MinvVw, 0, Lk, -1.0, 1.0)
            brV, brW = rV.norm(), rW.norm()
            S, Q, valsH = dense.hschur(H)                                       # :271-278
            T, Z, valsK = dense.hschur(K)
            S, Q, _ = dense.permuteschur(S, Q, dense.sortperm_general(valsH, which))
            T, Z, _ = dense.permuteschur(T, Z, dense.sortperm_general(np.conj(valsK), which))
            h = Q[Lk - 1, :] * bv                                               # :280-281
            k = Z[Lk - 1, :] * bw
            converged = 0
            while converged < Lk and max(brV * abs(h[converged]), brW * abs(k[converged])) <= tol:
                converged += 1
            if 0 < converged < Lk and S[converged, converged - 1] != 0:
                converged -= 1
            if converged >= howmany or (bv <= tol and bw <= tol):
                break
        if Lk < krylovdim:                                                      # expand  :303-312
            fV, fW = expand_(itV, fV), expand_(itW, fW)
            V, W = fV.basis(), fW.basis()
            vL, wL = HipVec(V, Lk), HipVec(W, Lk)
            MM[:Lk, Lk] = W.project(vL, 0, Lk)                                  # <W_i, V_{L+1}>
            MM[Lk, :Lk] = V.project(wL, 0, Lk)                                  # <W_{L+1}, V_i>
            MM[Lk, Lk] = wL.inner(vL)
            numops += 2
        else:                                                                   # shrink  :313-358
            if numiter == maxiter:
                break
            keep = (3 * krylovdim + 2 * converged) // 5
            stuck = False
            while S[keep, keep - 1] != 0 or T[keep, keep - 1] != 0:
                if keep > 1:
                    keep -= 1
                else:
                    keep += 1
                    if krylovdim == 2:
                        stuck = True
                        break
            if stuck:
                break
            H, K = np.array(S), np.array(T)
            VQv = -Q[:, :keep].T @ MinvWv
            WZw = -Z[:, :keep].T @ MinvVw
            H[:keep, :keep] += np.outer(VQv, h[:keep])
            K[:keep, :keep] += np.outer(WZw, k[:keep])
            V, W = fV.basis(), fW.basis()
            V.unproject(rV, Q[:, :keep] @ VQv, 0, Lk, -1.0, 1.0)
            W.unproject(rW, Z[:, :keep] @ WZw, 0, Lk, -1.0, 1.0)
            brV, brW = rV.norm(), rW.norm()
            rV.scale_(1 / brV)
            rW.scale_(1 / brW)
            h, k = h * brV, k * brW
            dense.restorearnoldiform(Q, H, h, keep)
            dense.restorearnoldiform(Z, K, k, keep)
            for fact, Hm, Um, rnew in ((fV, H, Q, rV), (fW, K, Z, rW)):
                _set_packed_hessenberg(fact, Hm, Lk)
                B = fact.basis()
                B.basistransform(np.ascontiguousarray(Um[:, :keep]))
                HipVec(B, keep).scale_from_(rnew, 1.0)                          # V[keep+1] = v_hat
                shrink_(fact, keep)
            MM[:keep, :keep] = Z[:, :keep].T @ (M @ Q[:, :keep])                # :352-356
            numiter += 1
    return (S, T), (Q, Z), (fV, fW), (rV, rW), (h, k), M, converged, numiter, numops


def bieigsolve(A, v0, w0, howmany: int = 1, which: str = "LM", alg=None, **kw):
    """bieigsolve(f, v0, w0, howmany, which, alg::BiArnoldi) (src/eigsolve/biarnoldi.jl:127-194): eigenvalues with right
    and left eigenvectors (W'V = I) of a general operator.  Returns (values, (vectorsV, vectorsW), (infoV, infoW))."""
    from .algorithms import BiArnoldi
    alg = alg or BiArnoldi(**kw)
    (S, T), (Q, Z), (fV, fW), (rV, rW), (h, k), M, converged, numiter, numops = _bischursolve(A, v0, w0, howmany, which, alg)
    hm = howmany
    if howmany < T.shape[0] and T[howmany, howmany - 1] != 0:
        hm += 1
    elif T.shape[0] < howmany:
        hm = T.shape[0]
    if converged > howmany:
        hm = converged
    SS = S[:hm, :hm]
    valuesS = dense.schur2eigvals(SS)
    vecsS = dense.schur2eigvecs(SS)
    ZMQ = Z[:, :hm].T @ M @ Q[:, :hm]
    vecsT = np.linalg.inv((ZMQ @ vecsS).conj().T)
    Lk = len(fV)
    vectorsS = _times_complex(fV.basis(), Lk, Q[:, :hm] @ vecsS)
    vectorsT = _times_complex(fW.basis(), Lk, Z[:, :hm] @ vecsT)
    hVS = [h[:hm] @ vecsS[:, i] for i in range(hm)]
    kVT = [k[:hm] @ vecsT[:, i] for i in range(hm)]
    rVh, rWh = rV.get(), rW.get()
    nV, nW = rV.norm(), rW.norm()   # device norms: all-reduced under a communicator
    infoS = ConvergenceInfo(converged, [rVh * s for s in hVS], np.array([nV * abs(s) for s in hVS]), numiter, numops)
    infoT = ConvergenceInfo(converged, [rWh * s for s in kVT], np.array([nW * abs(s) for s in kVT]), numiter, numops)
    return valuesS, (vectorsS, vectorsT), (infoS, infoT)


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


# -------------------------------------------------------------------- geneigsolve (Golub-Ye)


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
