"""TEST INFRASTRUCTURE, not product: host-driver mirrors of reference solvers that SURVEY.md section 2 marks OUT OF SCOPE --
Arnoldi / Krylov-Schur `eigsolve` and `schursolve` (src/eigsolve/arnoldi.jl), `bieigsolve` (src/eigsolve/biarnoldi.jl),
`geneigsolve` (src/eigsolve/golubye.jl) -- with the Schur helpers of src/dense/linalg.jl they need.  Written in round 1 inside
the package; moved here in round 4 so that `krylovkit_hip/` holds the section-8 path and nothing else.  They keep exercising
the device factorizations (kk_arnoldi_expand, kk_basistransform, ...) from more call patterns, which is their only purpose."""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import List, Optional

import numpy as np
import scipy.linalg as sla

from krylovkit_hip import dense
from krylovkit_hip.dense import *  # noqa: F401,F403  (householder, givens, eigsort, ... used unqualified by the moved code)
from krylovkit_hip.dense import householder, lmul_householder, rmul_householder
from krylovkit_hip.algorithms import *  # noqa: F401,F403
from krylovkit_hip.algorithms import Arnoldi, ConvergenceInfo
from krylovkit_hip.core import DeviceBasis, HipVec, KrylovDefaults, Orthogonalizer, SparseOperator
from krylovkit_hip.factorizations import (ArnoldiIterator, GKLIterator, LanczosIterator, _as_operator, expand_, initialize,
                                          initialize_, shrink_)


@dataclass
class BiArnoldi:  # algorithms.jl:274-291
    orth: Orthogonalizer = KrylovDefaults.orth
    krylovdim: int = KrylovDefaults.krylovdim
    maxiter: int = KrylovDefaults.maxiter
    tol: float = KrylovDefaults.tol
    eager: bool = False
    verbosity: int = 0




@dataclass
class GolubYe:  # algorithms.jl:310-325
    orth: Orthogonalizer = KrylovDefaults.orth
    krylovdim: int = KrylovDefaults.krylovdim
    maxiter: int = KrylovDefaults.maxiter
    tol: float = KrylovDefaults.tol
    verbosity: int = 0


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


# -------------------------------------------------------------------- eigsolve / schursolve (Arnoldi)
def _set_packed_hessenberg(fact, H: np.ndarray, K: int):
    """copy!(rayleighquotient(fact), H) (eigsolve/arnoldi.jl:443): write the K x K dense H back into the packed
    Hessenberg storage (dense/packedhessenberg.jl:32-48)."""
    from krylovkit_hip.factorizations import packed_index
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
            T, U, values = hschur(fact.rayleighquotient())
            T, U, values = permuteschur(T, U, sortperm_general(values, which))
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
            restorearnoldiform(U, H, f, keep)                          # :441
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
    values = schur2eigvals(TT)
    K = len(fact)
    vectors = _times_complex(fact.basis(), K, U[:, :hm])
    r = fact.r.get()
    residuals = [r * U[K - 1, i] for i in range(hm)]
    normres = np.array([fact.normres * abs(U[K - 1, i]) for i in range(hm)])
    return TT, vectors, values, ConvergenceInfo(converged, residuals, normres, numiter, numops)


def eigsolve_arnoldi(A, x0, howmany: int = 1, which: str = "LM", alg: Optional[Arnoldi] = None):
    """eigsolve(A, x0, howmany, which, alg::Arnoldi) (src/eigsolve/arnoldi.jl:277-316)."""
    T, U, fact, converged, numiter, numops = _schursolve(A, x0, howmany, which, alg)
    hm = _arnoldi_howmany(T, fact, howmany, converged)
    TT = T[:hm, :hm]
    values = schur2eigvals(TT)
    V = U[:, :hm] @ schur2eigvecs(TT)                                   # :297
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
    from krylovkit_hip.core import FunctionOperator
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
            S, Q, valsH = hschur(H)                                       # :271-278
            T, Z, valsK = hschur(K)
            S, Q, _ = permuteschur(S, Q, sortperm_general(valsH, which))
            T, Z, _ = permuteschur(T, Z, sortperm_general(np.conj(valsK), which))
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
            restorearnoldiform(Q, H, h, keep)
            restorearnoldiform(Z, K, k, keep)
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
    pass  # BiArnoldi is defined in this module
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
    valuesS = schur2eigvals(SS)
    vecsS = schur2eigvecs(SS)
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
    from krylovkit_hip.factorizations import Block, block_inner, lanczos_recurrence_unfused
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
