"""CPU oracle: NumPy/SciPy (float64) restatement of KrylovKit.jl's Krylov `expand!` path.

TEST INFRASTRUCTURE ONLY.  Nothing in the shipped product (`krylovkit.jl_amd/`) may import
this module; only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg use
it, and only as the checker.

PARITY STATUS: **bit-level parity unpinned** -- the reference is Julia and no `julia` binary
exists in this image, its test inputs come from Julia's seeded RNG, and every comparison in the
reference test-suite is tolerance based (`test/testsetup.jl:14`).  The oracle is therefore pinned
against what the reference's tests *do* pin: the per-`expand!` invariants of
`test/factorize.jl:140-148,185-193,285-296`, the orthogonaliser identities of
`test/linalg.jl:4-25`, dense LAPACK spectra (`test/eigsolve.jl:74,122-123`,
`test/svdsolve.jl:14,89`, `test/linsolve.jl:135,230`) and the two known-answer fixtures
(toric code: four eigenvalues at -16, `test/eigsolve.jl:471-549`; issue #143 71x71 matrix,
`test/issues.jl:39-129`).  See tests/test_oracle_*.py.

Every function cites the reference file:line (relative to /root/reference) it restates.  The
arithmetic that is NOT in the reference tree (VectorInterface.jl compat 0.5/0.6 -- unpinned, no
Manifest; Julia stdlib LinearAlgebra / SparseArrays) is restated from the call sites' semantics
(SURVEY.md Appendix B): inner = conj-linear dot, norm = 2-norm, add!!(y,x,a,b) = b*y + a*x.

Indices are 0-based here; comments give the Julia 1-based line they mirror.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Callable, List, Optional, Sequence, Tuple

import numpy as np
import scipy.linalg as sla
import scipy.sparse as sp

EPS = np.finfo(np.float64).eps


# --------------------------------------------------------------------------------------
# Orthogonalizers -- src/algorithms.jl:17-80
# --------------------------------------------------------------------------------------
@dataclass(frozen=True)
class Orthogonalizer:
    name: str
    eta: float = 1.0 / math.sqrt(2.0)  # DGKS default, algorithms.jl:66,80


CGS = Orthogonalizer("cgs")
MGS = Orthogonalizer("mgs")
CGS2 = Orthogonalizer("cgs2")
MGS2 = Orthogonalizer("mgs2")


def CGSIR(eta: float = 1.0 / math.sqrt(2.0)) -> Orthogonalizer:
    return Orthogonalizer("cgsir", eta)


def MGSIR(eta: float = 1.0 / math.sqrt(2.0)) -> Orthogonalizer:
    return Orthogonalizer("mgsir", eta)


ALL_ORTHS = (CGS, MGS, CGS2, MGS2, CGSIR(), MGSIR())


# --------------------------------------------------------------------------------------
# L1 vector verbs (VectorInterface.jl semantics) on float64 ndarrays
# --------------------------------------------------------------------------------------
def inner(x: np.ndarray, y: np.ndarray) -> float:
    return float(np.dot(x, y))  # real case of conj-linear <x,y>


def norm(x: np.ndarray) -> float:
    return float(np.linalg.norm(x))


def add(y: np.ndarray, x: np.ndarray, a: float = 1.0, b: float = 1.0) -> np.ndarray:
    """add!!(y, x, a, b): y <- b*y + a*x, in place, returns y."""
    if b != 1.0:
        y *= b
    y += a * x
    return y


def scale(x: np.ndarray, a: float) -> np.ndarray:
    return x * a  # fresh vector


def scale_(y: np.ndarray, a: float) -> np.ndarray:
    y *= a
    return y


# --------------------------------------------------------------------------------------
# Operator protocol -- src/apply.jl:1-19
# --------------------------------------------------------------------------------------
def apply(op, x: np.ndarray) -> np.ndarray:
    """apply(A::AbstractMatrix, x) = A*x ; apply(f, x) = f(x)   (apply.jl:1-2)"""
    if callable(op):
        return np.asarray(op(x), dtype=np.float64)
    return np.asarray(op @ x, dtype=np.float64).ravel()


def apply_normal(op, x):
    """apply.jl:14,16: matrix -> A*x ; tuple (f, fadj) -> f(x)"""
    if isinstance(op, tuple):
        return np.asarray(op[0](x), dtype=np.float64)
    return apply(op, x)


def apply_adjoint(op, x):
    """apply.jl:15,17: matrix -> A'*x ; tuple (f, fadj) -> fadj(x)"""
    if isinstance(op, tuple):
        return np.asarray(op[1](x), dtype=np.float64)
    return np.asarray(op.T @ x, dtype=np.float64).ravel()


# --------------------------------------------------------------------------------------
# OrthonormalBasis helpers -- src/orthonormal.jl.  A basis is a python list of ndarrays.
# --------------------------------------------------------------------------------------
def project(b: Sequence[np.ndarray], x: np.ndarray, y: Optional[np.ndarray] = None,
            alpha: float = 1.0, beta: float = 0.0, r: Optional[Sequence[int]] = None) -> np.ndarray:
    """project!! (orthonormal.jl:88-118): y[j] = beta*y[j] + alpha*inner(b[r[j]], x)."""
    r = range(len(b)) if r is None else r
    if y is None:
        y = np.zeros(len(r))
    for j, rj in enumerate(r):
        if beta == 0:
            y[j] = alpha * inner(b[rj], x)
        else:
            y[j] = beta * y[j] + alpha * inner(b[rj], x)
    return y


def unproject(y: np.ndarray, b: Sequence[np.ndarray], x: Sequence[float],
              alpha: float = 1.0, beta: float = 0.0, r: Optional[Sequence[int]] = None) -> np.ndarray:
    """unproject!! generic path (orthonormal.jl:132-150): y = beta*y + alpha*sum b[r[i]]*x[i]."""
    r = range(len(b)) if r is None else r
    if beta == 0:
        y[:] = 0.0
    elif beta != 1:
        y *= beta
    for i, ri in enumerate(r):
        y = add(y, b[ri], alpha * x[i])
    return y


def basis_times(b: Sequence[np.ndarray], x: Sequence[float]) -> np.ndarray:
    """Base.:*(b::OrthonormalBasis, x) (orthonormal.jl:57-60)."""
    return unproject(np.zeros_like(b[0]), b, x)


def rank1update(b: List[np.ndarray], y: np.ndarray, x: Sequence[float],
                alpha: float = 1.0, beta: float = 1.0, r: Optional[Sequence[int]] = None):
    """rank1update! (orthonormal.jl:210-229): b[r[i]] = beta*b[r[i]] + alpha*y*conj(x[i])."""
    r = range(len(b)) if r is None else r
    for i, ri in enumerate(r):
        if beta == 1:
            b[ri] = add(b[ri], y, alpha * x[i])
        elif beta == 0:
            b[ri] = y * (alpha * x[i])
        else:
            b[ri] = add(b[ri], y, alpha * x[i], beta)
    return b


def basistransform(b: List[np.ndarray], U: np.ndarray) -> List[np.ndarray]:
    """basistransform! (orthonormal.jl:291-321): b[j] <- sum_i b[i]*U[i,j]; old vectors dropped
    for j < n, vectors j >= n are left untouched (the reference only overwrites b[1:n], :316-318)."""
    m, n = U.shape
    assert m == len(b)
    b2 = []
    for j in range(n):
        v = b[0] * U[0, j]  # scale!!(b2[j], b[1], U[1,j])  :310
        for i in range(1, m):
            v = add(v, b[i], U[i, j])  # :312
        b2.append(v)
    for j in range(n):
        b[j] = b2[j]
    return b


def givens_rmul(b: List[np.ndarray], i1: int, i2: int, c: float, s: float):
    """rmul!(b::OrthonormalBasis, G::Givens) generic path (dense/givens.jl:30-36), real case:
    q1' = c*q1 - s*q2 ; q2' = s*q1 + c*q2."""
    q1, q2 = b[i1], b[i2]
    q1n = c * q1 - s * q2
    q2n = s * q1 + c * q2
    b[i1], b[i2] = q1n, q2n
    return b


def householder_rmul(b: List[np.ndarray], beta: float, v: Sequence[float], r: Sequence[int]):
    """rmul!(b::OrthonormalBasis, H::Householder) (dense/reflector.jl:143-154):
    w = sum_{i in r} b[i] v[i]; b[i] <- b[i] - beta*w*conj(v[i])."""
    if beta == 0:
        return b
    w = unproject(np.zeros_like(b[r[0]]), b, v, 1.0, 0.0, r)
    return rank1update(b, w, v, -beta, 1.0, r)


# --------------------------------------------------------------------------------------
# orthogonalize!! family -- src/orthonormal.jl:370-527
# --------------------------------------------------------------------------------------
def _cgs(v, b, x):  # :378-384
    x = project(b, v, x)
    v = unproject(v, b, x, -1.0, 1.0)
    return v, x


def _recgs(v, b, x):  # reorthogonalize!! CGS :385-393
    s = project(b, v, np.empty_like(x))
    v = unproject(v, b, s, -1.0, 1.0)
    x += s
    return v, x


def _mgs(v, b, x):  # :414-423
    for i, q in enumerate(b):
        s = inner(q, v)
        v = add(v, q, -s)
        x[i] = s
    return v, x


def _remgs(v, b, x):  # :424-433
    for i, q in enumerate(b):
        s = inner(q, v)
        v = add(v, q, -s)
        x[i] += s
    return v, x


def orthogonalize(v: np.ndarray, b: Sequence[np.ndarray], alg: Orthogonalizer,
                  x: Optional[np.ndarray] = None, stats: Optional[dict] = None):
    """orthogonalize!!(v, b, x, alg) (orthonormal.jl:372-452). Mutates v, returns (v, x).
    `stats['passes']` (optional) counts full passes over the basis (for byte accounting)."""
    if x is None:
        x = np.empty(len(b))
    passes = 1
    n = alg.name
    if n == "cgs":
        v, x = _cgs(v, b, x)
    elif n == "cgs2":  # :394-399
        v, x = _cgs(v, b, x)
        v, x = _recgs(v, b, x)
        passes = 2
    elif n == "cgsir":  # :400-412
        nold = norm(v)
        v, x = _cgs(v, b, x)
        nnew = norm(v)
        while EPS < nnew < alg.eta * nold:
            nold = nnew
            v, x = _recgs(v, b, x)
            nnew = norm(v)
            passes += 1
    elif n == "mgs":
        v, x = _mgs(v, b, x)
    elif n == "mgs2":  # :434-439
        v, x = _mgs(v, b, x)
        v, x = _remgs(v, b, x)
        passes = 2
    elif n == "mgsir":  # :440-452
        nold = norm(v)
        v, x = _mgs(v, b, x)
        nnew = norm(v)
        while EPS < nnew < alg.eta * nold:
            nold = nnew
            v, x = _remgs(v, b, x)
            nnew = norm(v)
            passes += 1
    else:
        raise ValueError(n)
    if stats is not None:
        stats["passes"] = stats.get("passes", 0) + passes
    return v, x


def orthogonalize_vec(v: np.ndarray, q: np.ndarray, alg: Orthogonalizer):
    """_orthogonalize!!(v, q, alg) vector-vs-vector variants (orthonormal.jl:455-489)."""
    n = alg.name
    if n in ("cgs", "mgs"):  # :458-464
        s = inner(q, v)
        v = add(v, q, -s)
        return v, s
    if n in ("cgs2", "mgs2"):  # :465-473
        s = inner(q, v)
        v = add(v, q, -s)
        ds = inner(q, v)
        v = add(v, q, -ds)
        return v, s + ds
    # IR :474-489
    nold = norm(v)
    s = inner(q, v)
    v = add(v, q, -s)
    nnew = norm(v)
    while EPS < nnew < alg.eta * nold:
        nold = nnew
        ds = inner(q, v)
        v = add(v, q, -ds)
        s += ds
        nnew = norm(v)
    return v, s


def orthonormalize(v, b, alg, x=None):
    """orthonormalize!! (orthonormal.jl:522-527): returns (v/|v|, beta, x)."""
    v, x = orthogonalize(v, b, alg, x)
    beta = norm(v)
    v = scale_(v, 1.0 / beta)
    return v, beta, x


# --------------------------------------------------------------------------------------
# Lanczos factorization -- src/factorizations/lanczos.jl
# --------------------------------------------------------------------------------------
@dataclass
class LanczosFactorization:  # lanczos.jl:31-37
    k: int
    V: List[np.ndarray]
    alphas: List[float]
    betas: List[float]
    r: np.ndarray

    def __len__(self):
        return self.k

    @property
    def normres(self):
        return self.betas[-1]

    def rayleighquotient(self):
        """SymTridiagonal(alphas, betas) (lanczos.jl:44-46) -> (diag, offdiag[:k-1])."""
        return np.array(self.alphas[: self.k]), np.array(self.betas[: self.k - 1])


@dataclass
class LanczosIterator:  # lanczos.jl:129-153
    operator: object
    x0: np.ndarray
    orth: Orthogonalizer = MGS2
    keepvecs: bool = True

    def __post_init__(self):
        if not self.keepvecs and self.orth.name not in ("cgs", "mgs"):
            raise ValueError("Cannot use reorthogonalization without keeping all Krylov vectors")


def lanczos_initialize(it: LanczosIterator) -> LanczosFactorization:
    """initialize(iter::LanczosIterator) (lanczos.jl:180-222)."""
    x0 = it.x0
    beta0 = norm(x0)
    if beta0 == 0:
        raise ValueError("initial vector should not have norm zero")
    Ax0 = apply(it.operator, x0)
    alpha = inner(x0, Ax0) / (beta0 * beta0)
    v = x0 * (1.0 / beta0)  # add!!(scale(Ax0, 0), x0, 1/beta0)  :190
    r = scale_(Ax0, 1.0 / beta0)  # :194
    beta_old = norm(r)
    r = add(r, v, -alpha)
    beta = norm(r)
    n = it.orth.name
    if n in ("cgs2", "mgs2"):  # :200-204
        dalpha = inner(v, r)
        alpha += dalpha
        r = add(r, v, -dalpha)
        beta = norm(r)
    elif n in ("cgsir", "mgsir"):  # :205-213
        while EPS < beta < it.orth.eta * beta_old:
            beta_old = beta
            dalpha = inner(v, r)
            alpha += dalpha
            r = add(r, v, -dalpha)
            beta = norm(r)
    return LanczosFactorization(1, [v], [alpha], [beta], r)


def lanczos_initialize_(it: LanczosIterator, st: LanczosFactorization) -> LanczosFactorization:
    """initialize!(iter, state) (lanczos.jl:223-249)."""
    V = st.V
    while len(V) > 1:
        V.pop()
    st.alphas.clear()
    st.betas.clear()
    V[0] = it.x0 * (1.0 / norm(it.x0))
    w = apply(it.operator, V[0])
    r, alpha = orthogonalize_vec(w, V[0], it.orth)
    beta = norm(r)
    st.k = 1
    st.alphas.append(alpha)
    st.betas.append(beta)
    st.r = r
    return st


def lanczosrecurrence(operator, V: List[np.ndarray], beta: float, orth: Orthogonalizer,
                      stats: Optional[dict] = None):
    """The six lanczosrecurrence methods (lanczos.jl:295-376). Returns (w, alpha, beta)."""
    n = orth.name
    v = V[-1]
    w = apply(operator, v)
    passes = 0
    if n == "cgs":  # :295-303
        alpha = inner(v, w)
        w = add(w, V[-2], -beta)
        w = add(w, v, -alpha)
        beta = norm(w)
    elif n == "mgs":  # :304-312
        w = add(w, V[-2], -beta)
        alpha = inner(v, w)
        w = add(w, v, -alpha)
        beta = norm(w)
    elif n == "cgs2":  # :313-324
        alpha = inner(v, w)
        w = add(w, V[-2], -beta)
        w = add(w, v, -alpha)
        w, s = orthogonalize(w, V, CGS)
        alpha += s[-1]
        beta = norm(w)
        passes = 1
    elif n == "mgs2":  # :325-338
        w = add(w, V[-2], -beta)
        w, alpha = orthogonalize_vec(w, v, MGS)
        s = alpha
        for q in V:
            w, s = orthogonalize_vec(w, q, MGS)
        alpha += s
        beta = norm(w)
        passes = 1
    elif n == "cgsir":  # :339-356
        alpha = inner(v, w)
        w = add(w, V[-2], -beta)
        w = add(w, v, -alpha)
        ab2 = alpha * alpha + beta * beta
        beta = norm(w)
        nold = math.sqrt(beta * beta + ab2)
        while EPS < beta < orth.eta * nold:
            nold = beta
            w, s = orthogonalize(w, V, CGS)
            alpha += s[-1]
            beta = norm(w)
            passes += 1
    elif n == "mgsir":  # :357-376
        w = add(w, V[-2], -beta)
        w, alpha = orthogonalize_vec(w, v, MGS)
        ab2 = alpha * alpha + beta * beta
        beta = norm(w)
        nold = math.sqrt(beta * beta + ab2)
        while EPS < beta < orth.eta * nold:
            nold = beta
            s = 0.0
            for q in V:
                w, s = orthogonalize_vec(w, q, MGS)
            alpha += s
            beta = norm(w)
            passes += 1
    else:
        raise ValueError(n)
    if stats is not None:
        stats["passes"] = stats.get("passes", 0) + passes
    return w, alpha, beta


def lanczos_expand(it: LanczosIterator, st: LanczosFactorization,
                   stats: Optional[dict] = None) -> LanczosFactorization:
    """expand!(iter::LanczosIterator, state) (lanczos.jl:250-272)."""
    beta_old = st.normres
    st.V.append(scale_(st.r, 1.0 / beta_old))  # :257
    r, alpha, beta = lanczosrecurrence(it.operator, st.V, beta_old, it.orth, stats)  # :258
    st.alphas.append(alpha)
    st.betas.append(beta)
    if not it.keepvecs:
        st.V.pop(0)  # :264
    st.k += 1
    st.r = r
    return st


def lanczos_shrink(st: LanczosFactorization, k: int) -> LanczosFactorization:
    """shrink!(state::LanczosFactorization, k) (lanczos.jl:273-291)."""
    assert len(st) == len(st.V)
    if len(st) <= k:
        return st
    V = st.V
    while len(V) > k + 1:
        V.pop()
    r = V.pop()
    del st.alphas[k:]
    del st.betas[k:]
    st.k = k
    st.r = scale_(r, st.normres)
    return st


# --------------------------------------------------------------------------------------
# Arnoldi factorization -- src/factorizations/arnoldi.jl, dense/packedhessenberg.jl
# --------------------------------------------------------------------------------------
def packed_index(i: int, j: int) -> int:
    """0-based offset of H[i,j] (1-based i<=j+1) in packed storage (packedhessenberg.jl:32-39):
    data[((j*j + j - 2) >> 1) + i]  (1-based)  ->  minus one for 0-based."""
    return ((j * j + j - 2) >> 1) + i - 1


@dataclass
class ArnoldiFactorization:  # arnoldi.jl:31-36
    k: int
    V: List[np.ndarray]
    H: List[float]  # packed Hessenberg
    r: np.ndarray

    def __len__(self):
        return self.k

    @property
    def normres(self):
        return abs(self.H[-1])  # arnoldi.jl:50

    def rayleighquotient(self) -> np.ndarray:
        """Dense k x k upper Hessenberg view of the packed data (arnoldi.jl:46-48)."""
        k = self.k
        Hd = np.zeros((k, k))
        for j in range(1, k + 1):
            for i in range(1, min(j + 1, k) + 1):
                Hd[i - 1, j - 1] = self.H[packed_index(i, j)]
        return Hd


@dataclass
class ArnoldiIterator:  # arnoldi.jl:98-106
    operator: object
    x0: np.ndarray
    orth: Orthogonalizer = MGS2


def arnoldi_initialize(it: ArnoldiIterator) -> ArnoldiFactorization:
    """initialize(iter::ArnoldiIterator) (arnoldi.jl:135-175)."""
    x0 = it.x0
    beta0 = norm(x0)
    if beta0 == 0:
        raise ValueError("initial vector should not have norm zero")
    Ax0 = apply(it.operator, x0)
    alpha = inner(x0, Ax0) / (beta0 * beta0)
    v = x0 * (1.0 / beta0)
    r = scale_(Ax0, 1.0 / beta0)
    beta_old = norm(r)
    r = add(r, v, -alpha)
    beta = norm(r)
    n = it.orth.name
    if n in ("cgs2", "mgs2"):
        dalpha = inner(v, r)
        alpha += dalpha
        r = add(r, v, -dalpha)
        beta = norm(r)
    elif n in ("cgsir", "mgsir"):
        while EPS < beta < it.orth.eta * beta_old:
            beta_old = beta
            dalpha = inner(v, r)
            alpha += dalpha
            r = add(r, v, -dalpha)
            beta = norm(r)
    return ArnoldiFactorization(1, [v], [alpha, beta], r)


def arnoldi_initialize_(it: ArnoldiIterator, st: ArnoldiFactorization) -> ArnoldiFactorization:
    """initialize!(iter, state) (arnoldi.jl:176-198)."""
    V = st.V
    while len(V) > 1:
        V.pop()
    st.H.clear()
    V[0] = it.x0 * (1.0 / norm(it.x0))
    w = apply(it.operator, V[0])
    r, alpha = orthogonalize_vec(w, V[0], it.orth)
    beta = norm(r)
    st.k = 1
    st.H.extend([alpha, beta])
    st.r = r
    return st


def arnoldi_expand(it: ArnoldiIterator, st: ArnoldiFactorization,
                   stats: Optional[dict] = None) -> ArnoldiFactorization:
    """expand!(iter::ArnoldiIterator, state) (arnoldi.jl:199-219) with
    arnoldirecurrence!! (arnoldi.jl:239-245)."""
    st.k += 1
    k = st.k
    beta = st.normres
    st.V.append(scale(st.r, 1.0 / beta))  # :209 (non-mutating)
    w = apply(it.operator, st.V[-1])  # :242
    h = np.empty(k)
    r, h = orthogonalize(w, st.V, it.orth, h, stats)  # :243
    beta = norm(r)  # :244
    st.H.extend(float(t) for t in h)
    st.H.append(beta)  # :213
    st.r = r
    return st


def arnoldi_shrink(st: ArnoldiFactorization, k: int) -> ArnoldiFactorization:
    """shrink!(state::ArnoldiFactorization, k) (arnoldi.jl:220-236)."""
    if len(st) <= k:
        return st
    V = st.V
    while len(V) > k + 1:
        V.pop()
    r = V.pop()
    del st.H[(k * k + 3 * k) >> 1:]
    st.k = k
    st.r = scale_(r, st.normres)
    return st


# --------------------------------------------------------------------------------------
# GKL factorization -- src/factorizations/gkl.jl
# --------------------------------------------------------------------------------------
@dataclass
class GKLFactorization:  # gkl.jl:31-38
    k: int
    U: List[np.ndarray]
    V: List[np.ndarray]
    alphas: List[float]
    betas: List[float]
    r: np.ndarray

    def __len__(self):
        return self.k

    @property
    def normres(self):
        return self.betas[-1]


@dataclass
class GKLIterator:  # gkl.jl:137-152
    operator: object
    u0: np.ndarray
    orth: Orthogonalizer = MGS2


def gkl_initialize(it: GKLIterator) -> GKLFactorization:
    """initialize(iter::GKLIterator) (gkl.jl:183-215)."""
    u0 = it.u0
    beta0 = norm(u0)
    if beta0 == 0:
        raise ValueError("initial vector should not have norm zero")
    v0 = apply_adjoint(it.operator, u0)
    alpha = norm(v0) / beta0
    Av0 = apply_normal(it.operator, v0)
    alpha2 = inner(u0, Av0) / beta0 ** 2
    if not np.isclose(alpha2, alpha * alpha, rtol=math.sqrt(EPS)):  # `≈`  :192
        raise ValueError("operator and its adjoint are not compatible")
    u = scale(u0, 1.0 / beta0)
    v = scale(v0, 1.0 / (alpha * beta0))
    r = scale_(Av0, 1.0 / (alpha * beta0))
    r = add(r, u, -alpha)
    beta = norm(r)
    return GKLFactorization(1, [u], [v], [alpha], [beta], r)


def gklrecurrence(operator, U, V, beta, orth: Orthogonalizer, stats: Optional[dict] = None):
    """The gklrecurrence methods (gkl.jl:294-404). Returns (v, r, alpha, beta)."""
    n = orth.name
    u = U[-1]
    v = apply_adjoint(operator, u)
    v = add(v, V[-1], -beta)
    pv = pu = 0
    if n in ("cgs", "mgs"):  # :294-307
        alpha = norm(v)
        v = scale_(v, 1.0 / alpha)
        r = apply_normal(operator, v)
        r = add(r, u, -alpha)
        beta = norm(r)
    elif n == "cgs2":  # :308-323
        alpha = norm(v)
        v = scale_(v, 1.0 / alpha)
        r = apply_normal(operator, v)
        r = add(r, u, -alpha)
        r, _ = orthogonalize(r, U, CGS)
        beta = norm(r)
        pu = 1
    elif n == "mgs2":  # :324-346
        for q in V:
            v, _ = orthogonalize_vec(v, q, MGS)
        alpha = norm(v)
        v = scale_(v, 1.0 / alpha)
        r = apply_normal(operator, v)
        r = add(r, u, -alpha)
        for q in U:
            r, _ = orthogonalize_vec(r, q, MGS)
        beta = norm(r)
        pv = pu = 1
    elif n == "cgsir":  # :347-373
        alpha = norm(v)
        nold = math.sqrt(alpha * alpha + beta * beta)
        while alpha < orth.eta * nold:  # note: no eps guard in the reference  :355
            nold = alpha
            v, _ = orthogonalize(v, V, CGS)
            alpha = norm(v)
            pv += 1
        v = scale_(v, 1.0 / alpha)
        r = apply_normal(operator, v)
        r = add(r, u, -alpha)
        beta = norm(r)
        nold = math.sqrt(alpha * alpha + beta * beta)
        while EPS < beta < orth.eta * nold:
            nold = beta
            r, _ = orthogonalize(r, U, CGS)
            beta = norm(r)
            pu += 1
    elif n == "mgsir":  # :374-404
        alpha = norm(v)
        nold = math.sqrt(alpha * alpha + beta * beta)
        while EPS < alpha < orth.eta * nold:
            nold = alpha
            for q in V:
                v, _ = orthogonalize_vec(v, q, MGS)
            alpha = norm(v)
            pv += 1
        v = scale_(v, 1.0 / alpha)
        r = apply_normal(operator, v)
        r = add(r, u, -alpha)
        beta = norm(r)
        nold = math.sqrt(alpha * alpha + beta * beta)
        while EPS < beta < orth.eta * nold:
            nold = beta
            for q in U:
                r, _ = orthogonalize_vec(r, q, MGS)
            beta = norm(r)
            pu += 1
    else:
        raise ValueError(n)
    if stats is not None:
        stats["passes_v"] = stats.get("passes_v", 0) + pv
        stats["passes_u"] = stats.get("passes_u", 0) + pu
    return v, r, alpha, beta


def gkl_expand(it: GKLIterator, st: GKLFactorization, stats: Optional[dict] = None):
    """expand!(iter::GKLIterator, state) (gkl.jl:246-269)."""
    beta_old = st.normres
    st.U.append(scale_(st.r, 1.0 / beta_old))
    v, r, alpha, beta = gklrecurrence(it.operator, st.U, st.V, beta_old, it.orth, stats)
    st.V.append(v)
    st.alphas.append(alpha)
    st.betas.append(beta)
    st.k += 1
    st.r = r
    return st


def gkl_shrink(st: GKLFactorization, k: int):
    """shrink!(state::GKLFactorization, k) (gkl.jl:270-291)."""
    assert len(st) == len(st.V)
    if len(st) <= k:
        return st
    U, V = st.U, st.V
    while len(V) > k + 1:
        U.pop()
        V.pop()
    V.pop()
    r = U.pop()
    del st.alphas[k:]
    del st.betas[k:]
    st.k = k
    st.r = scale_(r, st.normres)
    return st


# --------------------------------------------------------------------------------------
# Small dense helpers -- src/dense/*.jl, Julia LinearAlgebra.givens
# --------------------------------------------------------------------------------------
def householder_vec(x: np.ndarray, i: int):
    """_householder!(v, i) (dense/reflector.jl:40-69), real case. `i` 0-based.
    Returns (beta, v, nu) with v[i] = 1 such that (I - beta v v') x = nu e_i, nu = |x| >= 0."""
    v = np.array(x, dtype=np.float64)
    sigma = float(np.dot(v[:i], v[:i]) + np.dot(v[i + 1:], v[i + 1:]))
    vi = v[i]
    nu = math.sqrt(vi * vi + sigma)
    if sigma == 0 and vi == nu:
        beta = 0.0
    else:
        if vi < 0:
            vi = vi - nu
        else:
            vi = (-sigma) / (vi + nu)  # ((vi - conj(vi))*nu - sigma)/(conj(vi)+nu), real
        v[:i] /= vi
        v[i + 1:] /= vi
        v[i] = 1.0
        beta = -vi / nu
    return beta, v, nu


def householder_lmul(beta, v, r, A: np.ndarray):
    """lmul!(H, A) (dense/reflector.jl:89-112): A[r,:] -= beta * v (v' A[r,:])."""
    if beta == 0:
        return A
    mu = beta * (v @ A[r, :])
    A[r, :] -= np.outer(v, mu)
    return A


def householder_rmul_mat(A: np.ndarray, beta, v, r, rows=None):
    """rmul!(A, H) (dense/reflector.jl:113-142): A[rows,r] -= (A[rows,r] v) beta v'.
    For H' (adjoint) beta -> conj(beta): identical in the real case."""
    if beta == 0:
        return A
    rows = slice(None) if rows is None else rows
    w = A[rows, r] @ v
    A[rows, r] -= np.outer(w, beta * v)
    return A


def givens(f: float, g: float):
    """LinearAlgebra.givensAlgorithm (LAPACK dlartg semantics), real case: returns (c, s, r)
    with [c s; -s c] [f; g] = [r; 0]."""
    if g == 0:
        return 1.0, 0.0, f
    if f == 0:
        return 0.0, 1.0, g
    r = math.hypot(f, g)
    c, s = f / r, g / r
    if abs(f) > abs(g) and c < 0:
        c, s, r = -c, -s, -r
    return c, s, r


def eigsort_key(which: str):
    """eigsort (eigsolve/eigsolve.jl:334-355), real spectrum."""
    if which == "LM":
        return (lambda d: np.abs(d)), True
    if which == "LR":
        return (lambda d: d), True
    if which == "SR":
        return (lambda d: d), False
    raise ValueError(f"invalid specification of which eigenvalues to target: which = {which}")


def sortperm(D: np.ndarray, which: str) -> np.ndarray:
    by, rev = eigsort_key(which)
    key = by(np.asarray(D))
    # Julia's sortperm is stable; rev=true keeps stability w.r.t. reversed comparison
    return np.argsort(-key if rev else key, kind="stable")


def tridiageigh(alphas: np.ndarray, betas: np.ndarray):
    """tridiageigh! -> LAPACK stegr (dense/linalg.jl:109-116,396-458): ascending eigenvalues."""
    if len(alphas) == 1:
        return np.array([alphas[0]]), np.ones((1, 1))
    return sla.eigh_tridiagonal(alphas, betas, lapack_driver="stemr")


# --------------------------------------------------------------------------------------
# Drivers
# --------------------------------------------------------------------------------------
@dataclass
class ConvergenceInfo:  # KrylovKit.jl:212-218
    converged: int
    residual: object
    normres: object
    numiter: int
    numops: int


def eigsolve_lanczos(A, x0: np.ndarray, howmany: int = 1, which: str = "LM", *, krylovdim: int = 30,
                     maxiter: int = 100, tol: float = 1e-12, orth: Orthogonalizer = MGS2,
                     eager: bool = False, trace: Optional[list] = None):
    """eigsolve(A, x0, howmany, which, alg::Lanczos) (eigsolve/lanczos.jl:1-155)."""
    if howmany > krylovdim:
        raise ValueError("krylov dimension too small")
    it = LanczosIterator(A, np.array(x0, dtype=np.float64), orth)
    fact = lanczos_initialize(it)
    numops = 1
    numiter = 1
    HH = np.zeros((krylovdim + 1, krylovdim))
    converged = 0
    D = U = f = None
    while True:
        beta = fact.normres
        K = len(fact)
        if K == krylovdim or beta <= tol or (eager and K >= howmany):  # :45
            a, b = fact.rayleighquotient()
            if K == 1:
                D = np.array([a[0]])
                U = np.ones((1, 1))
                f = np.array([beta])
                converged = int(beta <= tol)
            else:
                D, U = tridiageigh(a, b)
                p = sortperm(D, which)
                D, U = D[p], U[:, p]
                f = U[K - 1, :] * beta  # :61
                converged = 0
                while converged < K and abs(f[converged]) <= tol:
                    converged += 1
            if trace is not None:
                trace.append((numiter, K, D.copy(), np.abs(f)))
            if converged >= howmany or beta <= tol:
                break
        if K < krylovdim:  # :77-79
            fact = lanczos_expand(it, fact)
            numops += 1
        else:  # :80-116
            if numiter == maxiter:
                break
            keep = (3 * krylovdim + 2 * converged) // 5
            H = HH[: keep + 1, :keep]
            H[:] = 0.0
            for j in range(keep):
                H[j, j] = D[j]
                H[keep, j] = f[j]
            U = np.array(U)
            for j in range(keep, 0, -1):  # j = keep:-1:1 (1-based)  :94-101
                hb, hv, nu = householder_vec(H[j, :j], j - 1)  # householder(H, j+1, 1:j, j)
                H[j, j - 1] = nu
                H[j, : j - 1] = 0.0
                rr = np.arange(j)
                householder_lmul(hb, hv, rr, H)
                householder_rmul_mat(H, hb, hv, rr, rows=slice(0, j))
                householder_rmul_mat(U, hb, hv, rr)
            for j in range(keep):
                fact.alphas[j] = H[j, j]
                fact.betas[j] = H[j + 1, j]
            B = fact.V
            B = basistransform(B, U[:, :keep])  # :109
            r = fact.r
            B[keep] = scale_(r, 1.0 / beta)  # :111
            fact = lanczos_shrink(fact, keep)
            numiter += 1
    hm = howmany
    if converged > howmany:
        hm = converged
    elif len(D) < howmany:
        hm = len(D)
    values = D[:hm]
    Vc = U[:, :hm]
    vectors = [basis_times(fact.V, Vc[:, i]) for i in range(hm)]
    residuals = [scale(fact.r, Vc[-1, i]) for i in range(hm)]
    normres = np.abs(f[:hm])
    return values, vectors, ConvergenceInfo(converged, residuals, normres, numiter, numops)


def gmres(operator, b: np.ndarray, x0: Optional[np.ndarray] = None, a0: float = 0.0, a1: float = 1.0, *,
          krylovdim: int = 30, maxiter: int = 100, tol: float = 1e-12,
          orth: Orthogonalizer = MGS2, trace: Optional[list] = None):
    """linsolve(operator, b, x0, alg::GMRES, a0, a1) (linsolve/gmres.jl:1-151).
    `tol` is the absolute tolerance (the front-end computes max(atol, rtol*|b|),
    linsolve/linsolve.jl:135-140)."""
    b = np.asarray(b, dtype=np.float64)
    x0 = np.zeros_like(b) if x0 is None else np.asarray(x0, dtype=np.float64)
    y0 = apply(operator, x0)
    r = scale(b, 1.0)
    if a0 != 0:
        r = add(r, x0, -a0)
    r = add(r, y0, -a1)
    x = x0.copy()
    beta = norm(r)
    if beta < tol:
        return x, ConvergenceInfo(1, r, beta, 0, 1)
    y = np.zeros(krylovdim + 1)
    gs = [None] * krylovdim
    R = np.zeros((krylovdim, krylovdim))
    numiter = 0
    numops = 1
    it = ArnoldiIterator(operator, r, orth)
    fact = arnoldi_initialize(it)
    numops += 1
    while True:
        numiter += 1
        y[0] = beta
        k = 1
        H = fact.rayleighquotient()
        R[0, 0] = a0 + a1 * H[0, 0]
        c, s, R[0, 0] = givens(R[0, 0], a1 * fact.normres)
        gs[0] = (0, 1, c, s)  # Givens(i1, i2, c, s), 0-based
        y[1] = 0.0
        y[0], y[1] = c * y[0] + s * y[1], -s * y[0] + c * y[1]
        beta = abs(y[1])
        if trace is not None:
            trace.append((numiter, 1, beta))
        while R[k - 1, k - 1] != 0 and beta > tol and len(fact) < krylovdim:  # :55
            fact = arnoldi_expand(it, fact)
            numops += 1
            k = len(fact)
            H = fact.rayleighquotient()
            for i in range(k - 1):
                R[i, k - 1] = a1 * H[i, k - 1]
            R[k - 1, k - 1] = a0 + a1 * H[k - 1, k - 1]
            Rk = R[:, k - 1]
            for i in range(k - 1):  # apply old Givens: lmul!(gs[i], Rk)  :72-75
                i1, i2, c, s = gs[i]
                Rk[i1], Rk[i2] = c * Rk[i1] + s * Rk[i2], -s * Rk[i1] + c * Rk[i2]
            if math.hypot(R[k - 1, k - 1], a1 * fact.normres) < tol:  # :78-85
                c, s, y[k] = givens(0.0, y[k - 1])  # givens(0, y[k], k+1, k): i1=k+1, i2=k
                gs[k - 1] = (k, k - 1, c, s)
                y[k - 1] = 0.0
                R[k - 1, k - 1] = 0.0
            else:
                c, s, R[k - 1, k - 1] = givens(R[k - 1, k - 1], a1 * fact.normres)
                gs[k - 1] = (k - 1, k, c, s)
                y[k] = 0.0
                y[k - 1], y[k] = c * y[k - 1] + s * y[k], -s * y[k - 1] + c * y[k]
            beta = abs(y[k])
            if trace is not None:
                trace.append((numiter, k, beta))
        # triangular solve :98-102 (ldiv!, dense/linalg.jl:96-106)
        kk = k - 1 if (R[k - 1, k - 1] == 0 and y[k - 1] == 0) else k
        for j in range(kk - 1, -1, -1):
            y[j] = y[j] / R[j, j]
            y[:j] -= R[:j, j] * y[j]
        V = fact.V
        for i in range(k):  # :105-108
            x = add(x, V[i], y[i])
        if beta > tol and numiter < maxiter:  # :110-117
            w = fact.r
            V.append(scale_(w, 1.0 / fact.normres))
            for i in range(k):
                i1, i2, c, s = gs[i]
                givens_rmul(V, i1, i2, c, -s)  # rmul!(V, gs[i]')  adjoint: s -> -s
            r = V[k] * y[k]
        else:  # :119-132
            r = scale(b, 1.0)
            Ax = apply(operator, x)
            r = add(r, a0 * x + a1 * Ax, -1.0)
            numops += 1
            beta = norm(r)
            if beta < tol:
                return x, ConvergenceInfo(1, r, beta, numiter, numops)
        if numiter >= maxiter:
            return x, ConvergenceInfo(0, r, beta, numiter, numops)
        it = ArnoldiIterator(operator, r, orth)
        fact = arnoldi_initialize_(it, fact)


def svdsolve_gkl(A, x0: np.ndarray, howmany: int = 1, which: str = "LR", *, krylovdim: int = 30,
                 maxiter: int = 100, tol: float = 1e-12, orth: Orthogonalizer = MGS2,
                 eager: bool = False):
    """svdsolve(A, x0, howmany, which, alg::GKL) (eigsolve/svdsolve.jl:144-314)."""
    if howmany > krylovdim:
        raise ValueError("krylov dimension too small")
    numiter = 1
    it = GKLIterator(A, np.array(x0, dtype=np.float64), orth)
    fact = gkl_initialize(it)
    numops = 2
    HH = np.zeros((krylovdim + 1, krylovdim))
    converged = 0
    P = Q = f = S = None
    while True:
        beta = fact.normres
        K = len(fact)
        if K == krylovdim or beta <= tol or (eager and K >= howmany):
            # rayleighquotient: lower Bidiagonal(alphas, betas[:K-1])  gkl.jl:50-55
            Bm = np.diag(np.array(fact.alphas[:K]))
            if K > 1:
                Bm += np.diag(np.array(fact.betas[: K - 1]), -1)
            P, S, Q = np.linalg.svd(Bm)  # B = P*diag(S)*Q ; descending S  (bidiagsvd!, linalg.jl:123-130)
            if which == "SR":
                P = P[:, ::-1]
                S = S[::-1]
                Q = Q[::-1, :]
            elif which != "LR":
                raise ValueError(which)
            f = Q.T[K - 1, :] * beta  # mul!(f, view(Q', K, :), beta)  :208
            converged = 0
            while converged < K and abs(f[converged]) < tol:
                converged += 1
            if converged >= howmany or beta <= tol:
                break
        if K < krylovdim:
            fact = gkl_expand(it, fact)
            numops += 2
        else:
            if numiter == maxiter:
                break
            keep = (3 * krylovdim + 2 * converged) // 5
            U = fact.U
            basistransform(U, P[:, :keep])  # :233
            V = fact.V
            basistransform(V, Q.T[:, :keep])  # :240
            r = fact.r
            U[keep] = scale_(r, 1.0 / fact.normres)  # :249
            H = HH[: keep + 1, :keep]
            H[:] = 0.0
            for j in range(keep):
                H[j, j] = S[j]
                H[keep, j] = f[j]
            for j in range(keep, 0, -1):  # :257-269
                hb, hv, nu = householder_vec(H[j, :j], j - 1)  # householder(H, j+1, 1:j, j)
                H[j, j - 1] = nu
                H[j, : j - 1] = 0.0
                rr = np.arange(j)
                householder_rmul_mat(H, hb, hv, rr, rows=slice(0, j))
                householder_rmul(V, hb, hv, rr)  # rmul!(V, h')
                hb, hv, nu = householder_vec(H[:j, j - 1], j - 1)  # householder(H, 1:j, j, j)
                H[j - 1, j - 1] = nu
                H[: j - 1, j - 1] = 0.0
                householder_lmul(hb, hv, rr, H[:, : j - 1])
                householder_rmul(U, hb, hv, rr)  # rmul!(U, h')
            for j in range(keep):
                fact.alphas[j] = H[j, j]
                fact.betas[j] = H[j + 1, j]
            fact = gkl_shrink(fact, keep)
            numiter += 1
    if converged > howmany:
        howmany = converged
    values = S[:howmany]
    Pv = P[:, :howmany]
    Qv = Q[:howmany, :]
    left = [basis_times(fact.U, Pv[:, i]) for i in range(howmany)]
    right = [basis_times(fact.V, Qv[i, :]) for i in range(howmany)]
    residuals = [scale(fact.r, Qv[i, -1]) for i in range(howmany)]
    normres = np.abs(f[:howmany])
    return values, left, right, ConvergenceInfo(converged, residuals, normres, numiter, numops)


def cg(operator, b: np.ndarray, x0: Optional[np.ndarray] = None, a0: float = 0.0, a1: float = 1.0, *,
       maxiter: int = 100, tol: float = 1e-12):
    """linsolve(operator, b, x0, alg::CG, a0, a1) (linsolve/cg.jl:1-103)."""
    b = np.asarray(b, dtype=np.float64)
    x0 = np.zeros_like(b) if x0 is None else np.asarray(x0, dtype=np.float64)
    y0 = apply(operator, x0)
    r = scale(b, 1.0)
    if a0 != 0:
        r = add(r, x0, -a0)
    r = add(r, y0, -a1)
    x = x0.copy()
    normr = norm(r)
    numops, numiter = 1, 0
    if normr < tol:
        return x, ConvergenceInfo(1, r, normr, numiter, numops)
    rho = normr ** 2
    p = r.copy()
    q = a0 * p + a1 * apply(operator, p)
    alpha = rho / inner(p, q)
    x = add(x, p, alpha)
    r = add(r, q, -alpha)
    normr = norm(r)
    rho_old, rho = rho, normr ** 2
    beta = rho / rho_old
    numops += 1
    numiter += 1
    if normr < tol:
        return x, ConvergenceInfo(1, r, normr, numiter, numops)
    while True:
        p = add(p, r, 1.0, beta)
        q = a0 * p + a1 * apply(operator, p)
        alpha = rho / inner(p, q)
        x = add(x, p, alpha)
        r = add(r, q, -alpha)
        normr = norm(r)
        if normr < tol:
            r = b - (a0 * x + a1 * apply(operator, x))
            normr = norm(r)
            rho = normr ** 2
            beta = 0.0
        else:
            rho_old, rho = rho, normr ** 2
            beta = rho / rho_old
        numops += 1
        numiter += 1
        if normr < tol:
            return x, ConvergenceInfo(1, r, normr, numiter, numops)
        if numiter >= maxiter:
            return x, ConvergenceInfo(0, r, normr, numiter, numops)


# --------------------------------------------------------------------------------------
# BlockLanczos -- src/factorizations/blocklanczos.jl, src/eigsolve/blocklanczos.jl
# A Block is a python list of ndarrays.
# --------------------------------------------------------------------------------------
def block_inner(B1: Sequence[np.ndarray], B2: Sequence[np.ndarray]) -> np.ndarray:
    """block_inner (blocklanczos.jl:43-52): M[i,j] = inner(B1[i], B2[j]), p*q scalar calls."""
    M = np.empty((len(B1), len(B2)))
    for j in range(len(B2)):
        for i in range(len(B1)):
            M[i, j] = inner(B1[i], B2[j])
    return M


def block_norm(B: Sequence[np.ndarray]) -> float:
    """LinearAlgebra.norm(b::Block) = norm(b.vec) (blocklanczos.jl:37): Frobenius norm."""
    return math.sqrt(sum(inner(b, b) for b in B))


def block_reorthogonalize(R: List[np.ndarray], V: Sequence[np.ndarray]) -> List[np.ndarray]:
    """block_reorthogonalize! (blocklanczos.jl:277-284): one MGS sweep of every R[i] against V."""
    for i in range(len(R)):
        for q in V:
            R[i], _ = orthogonalize_vec(R[i], q, MGS)
    return R


def block_qr(block: List[np.ndarray], tol: float):
    """block_qr! (blocklanczos.jl:312-353): in-place MGS QR with rank detection (beta < tol ->
    zero vector, dropped from good_idx) and one DGKS correction when tol < beta < 100 tol.
    Returns (R[good_idx, :], good_idx, is_drift)."""
    n = len(block)
    is_drift = False
    idx = [True] * n
    R = np.zeros((n, n))
    beta = math.sqrt(inner(block[0], block[0]))
    if beta > tol:
        R[0, 0] = beta
        block[0] = scale_(block[0], 1.0 / beta)
    else:
        block[0][:] = 0.0
        idx[0] = False
    for j in range(1, n):
        for i in range(j):  # first MGS  :328-331
            R[i, j] = inner(block[i], block[j])
            block[j] = add(block[j], block[i], -R[i, j])
        beta = norm(block[j])
        if tol < beta < 100 * tol:  # DGKS  :334-342
            is_drift = True
            for i in range(j):
                d = inner(block[i], block[j])
                R[i, j] += d
                block[j] = add(block[j], block[i], -d)
            beta = norm(block[j])
        if beta < tol:
            block[j][:] = 0.0
            idx[j] = False
        else:
            R[j, j] = beta
            block[j] = scale_(block[j], 1.0 / beta)
    good = [i for i in range(n) if idx[i]]
    return R[good, :], good, is_drift


@dataclass
class BlockLanczosFactorization:  # blocklanczos.jl:89-96
    k: int
    V: List[np.ndarray]
    H: np.ndarray
    R: List[np.ndarray]
    R_size: int
    norm_R: float

    def __len__(self):
        return self.k

    @property
    def normres(self):
        return self.norm_R


@dataclass
class BlockLanczosIterator:  # blocklanczos.jl:133-157
    operator: object
    x0: List[np.ndarray]
    maxdim: int
    orth: Orthogonalizer = MGS2
    qr_tol: float = 1e-12

    def __post_init__(self):
        if self.orth.name != "mgs2":
            raise ValueError("BlockLanczosIterator only supports ModifiedGramSchmidt2 orthogonalizer")


def blocklanczos_initialize(it: BlockLanczosIterator) -> BlockLanczosFactorization:
    """initialize(iter::BlockLanczosIterator) (blocklanczos.jl:159-198)."""
    X0 = it.x0
    beta0 = block_norm(X0)
    if beta0 == 0:
        raise ValueError("initial vector should not have norm zero")
    X1 = [np.array(x, dtype=np.float64) for x in X0]
    _, good, _ = block_qr(X1, it.qr_tol)
    X1 = [X1[i] for i in good]
    V = list(X1)
    bs = len(X1)
    AX1 = [apply(it.operator, x) for x in X1]
    M1 = block_inner(X1, AX1)
    BTD = np.zeros((it.maxdim, it.maxdim))
    BTD[:bs, :bs] = M1
    for j in range(bs):
        for i in range(bs):
            AX1[j] = add(AX1[j], X1[i], -M1[i, j])
    return BlockLanczosFactorization(bs, V, BTD, AX1, bs, block_norm(AX1))


def block_lanczosrecurrence(operator, V: List[np.ndarray], B: np.ndarray):
    """block_lanczosrecurrence (blocklanczos.jl:242-263)."""
    bs, bs_prev = B.shape
    k = len(V)
    X = V[k - bs:k]
    AX = [apply(operator, x) for x in X]
    M = block_inner(X, AX)
    Xprev = V[k - bs_prev - bs:k - bs]
    for j in range(len(X)):
        for i in range(len(X)):
            AX[j] = add(AX[j], X[i], -M[i, j])
        for i in range(len(Xprev)):
            AX[j] = add(AX[j], Xprev[i], -B[j, i])
    block_reorthogonalize(AX, V)
    return AX, M


def blocklanczos_expand(it: BlockLanczosIterator, st: BlockLanczosFactorization) -> BlockLanczosFactorization:
    """expand!(iter::BlockLanczosIterator, state) (blocklanczos.jl:200-240)."""
    k = st.k
    R = st.R[: st.R_size]
    bs = len(R)
    V = st.V
    Rcopy = [r.copy() for r in R]
    B, good, is_drift = block_qr(R, it.qr_tol)
    if is_drift:  # :212-216
        block_reorthogonalize(R, V)
        _, good, is_drift = block_qr(R, it.qr_tol)
        B = block_inner([R[i] for i in good], Rcopy)
    bs_next = len(good)
    V.extend(R[i] for i in good)
    st.H[k:k + bs_next, k - bs:k] = B[:bs_next, :bs]
    st.H[k - bs:k, k:k + bs_next] = B[:bs_next, :bs].T
    Rnext, Mnext = block_lanczosrecurrence(it.operator, V, B)
    st.H[k:k + bs_next, k:k + bs_next] = Mnext[:bs_next, :bs_next]
    st.R[:bs_next] = Rnext
    st.norm_R = block_norm(Rnext)
    st.k += bs_next
    st.R_size = bs_next
    return st


def eigsolve_blocklanczos(A, x0: List[np.ndarray], howmany: int = 1, which: str = "SR", *, krylovdim: int = 100,
                          maxiter: int = 100, tol: float = 1e-12, qr_tol: float = 1e-12, eager: bool = False):
    """eigsolve(A, x0::Block, howmany, which, alg::BlockLanczos) (src/eigsolve/blocklanczos.jl:1-144)."""
    if howmany > krylovdim:
        raise ValueError("krylov dimension too small")
    bs = len(x0)
    it = BlockLanczosIterator(A, x0, krylovdim + bs, MGS2, qr_tol)
    fact = blocklanczos_initialize(it)
    numops = bs + 1
    numiter = 1
    converged = 0
    normresiduals = D = U = None
    while True:
        K = len(fact)
        beta = fact.normres
        if K >= krylovdim or beta <= tol or (eager and K >= howmany):  # :39
            BTD = fact.H[:K, :K]
            D, U = np.linalg.eigh((BTD + BTD.T) / 2)  # eigen(Hermitian(BTD))
            p = sortperm(D, which)
            D, U = D[p], U[:, p]
            bs_R = fact.R_size
            r = fact.R[:bs_R]
            UU = U[K - bs_R:K, :]
            Rm = block_inner(r, r)
            normresiduals = np.array([math.sqrt(max(float(UU[:, i] @ Rm @ UU[:, i]), 0.0)) for i in range(K)])
            converged = 0
            while converged < K and normresiduals[converged] <= tol:
                converged += 1
            if converged >= howmany or beta <= tol:
                break
        if K < krylovdim:
            fact = blocklanczos_expand(it, fact)
            numops += fact.R_size
        else:  # :68-104
            if numiter >= maxiter:
                break
            keep = max((3 * krylovdim + 2 * converged) // (5 * bs), 1) * bs
            H = np.zeros((keep + bs, keep))
            for j in range(keep):
                H[j, j] = D[j]
                H[keep:, j] = U[K - bs:K, j]
            U = np.array(U)
            for j in range(keep, 0, -1):  # :80-87
                hb, hv, nu = householder_vec(H[j + bs - 1, :j], j - 1)  # householder(H, j+bs, 1:j, j)
                H[j + bs - 1, j - 1] = nu
                H[j + bs - 1, : j - 1] = 0.0
                rr = np.arange(j)
                householder_lmul(hb, hv, rr, H)
                householder_rmul_mat(H, hb, hv, rr, rows=slice(0, j + bs - 1))
                householder_rmul_mat(U, hb, hv, rr)
            fact.H[:] = 0.0
            Hk = H[:keep, :keep]
            fact.H[:keep, :keep] = (Hk + Hk.T) / 2
            basistransform(fact.V, U[:, :keep])  # :92
            Rnew = list(fact.R[:bs_R])
            view_H = H[keep + bs - bs_R:keep + bs, keep - bs_R:keep]
            basistransform(Rnew, view_H)  # :96
            fact.R[:bs_R] = Rnew[:bs_R]
            while len(fact.V) > keep:  # the reference pops while length(fact) > keep
                fact.V.pop()
            fact.k = keep
            numiter += 1
    hm = howmany
    if converged > howmany:
        hm = converged
    elif len(D) < howmany:
        hm = len(D)
    values = D[:hm]
    K = len(fact)
    vectors = [basis_times(fact.V[:K], U[:, i]) for i in range(hm)]
    return values, vectors, ConvergenceInfo(converged, None, normresiduals[:hm], numiter, numops)


# --------------------------------------------------------------------------------------
# Synthetic operators of SURVEY.md §8(d) (shared by tests and bench; pure functions of shape/seed)
# --------------------------------------------------------------------------------------
def laplacian_2d(nx: int, ny: int, shift_diag: Optional[np.ndarray] = None) -> sp.csr_matrix:
    """5-point Dirichlet Laplacian on an nx x ny grid, diag 4, off-diag -1 (cfg 2).
    Row index = ix + nx*iy."""
    ex, ey = np.ones(nx), np.ones(ny)
    Tx = sp.diags([-ex[:-1], 2 * ex, -ex[:-1]], [-1, 0, 1])
    Ty = sp.diags([-ey[:-1], 2 * ey, -ey[:-1]], [-1, 0, 1])
    A = sp.kron(sp.identity(ny), Tx) + sp.kron(Ty, sp.identity(nx))
    if shift_diag is not None:
        A = A + sp.diags(shift_diag)
    return sp.csr_matrix(A)


def laplacian_2d_eigs(nx: int, ny: int) -> np.ndarray:
    i = np.arange(1, nx + 1)
    j = np.arange(1, ny + 1)
    lx = 2 - 2 * np.cos(i * np.pi / (nx + 1))
    ly = 2 - 2 * np.cos(j * np.pi / (ny + 1))
    return np.sort((lx[:, None] + ly[None, :]).ravel())


def convection_diffusion_2d(nx: int, ny: int, px: float = 0.5, py: float = 0.25) -> sp.csr_matrix:
    """Nonsymmetric 5-point convection-diffusion (cfg 3), central differences, cell Peclet
    numbers px, py: stencil  W=-(1+px) E=-(1-px) S=-(1+py) N=-(1-py) C=4."""
    ex, ey = np.ones(nx), np.ones(ny)
    Tx = sp.diags([-(1 + px) * ex[:-1], 2 * ex, -(1 - px) * ex[:-1]], [-1, 0, 1])
    Ty = sp.diags([-(1 + py) * ey[:-1], 2 * ey, -(1 - py) * ey[:-1]], [-1, 0, 1])
    return sp.csr_matrix(sp.kron(sp.identity(ny), Tx) + sp.kron(Ty, sp.identity(nx)))


def sparse_random(nrows: int, ncols: int, nnz_per_row: int, seed: int) -> sp.csr_matrix:
    """cfg 4: random sparse rectangular, duplicates summed."""
    rng = np.random.default_rng(seed)
    cols = rng.integers(0, ncols, size=(nrows, nnz_per_row))
    vals = rng.standard_normal((nrows, nnz_per_row))
    rows = np.repeat(np.arange(nrows), nnz_per_row)
    A = sp.coo_matrix((vals.ravel(), (rows, cols.ravel())), shape=(nrows, ncols))
    return sp.csr_matrix(A)


def bicgstab(operator, b: np.ndarray, x0: Optional[np.ndarray] = None, a0: float = 0.0, a1: float = 1.0, *,
             maxiter: int = 100, tol: float = 1e-12):
    """linsolve(operator, b, x0, alg::BiCGStab, a0, a1) (linsolve/bicgstab.jl:1-203)."""
    b = np.asarray(b, dtype=np.float64)
    x0 = np.zeros_like(b) if x0 is None else np.asarray(x0, dtype=np.float64)

    def aff(z):  # apply(operator, z, a0, a1)  (apply.jl:4-11)
        return a0 * z + a1 * apply(operator, z)

    y0 = apply(operator, x0)                                   # :3
    r = scale(b, 1.0)
    if a0 != 0:
        r = add(r, x0, -a0)
    r = add(r, y0, -a1)
    x = x0.copy()
    normr = norm(r)
    numops, numiter = 1, 0
    if normr < tol:                                            # :22-28
        return x, ConvergenceInfo(1, r, normr, numiter, numops)
    numiter += 1
    r_shadow = r.copy()                                        # :35
    rho = inner(r_shadow, r)
    if rho == 0.0:   # `ρ ≈ 0.0` = isapprox with atol = 0: true only for an exact zero   :39-46
        return x, ConvergenceInfo(0, r, normr, numiter, numops)
    p = r.copy()
    v = None
    omega = alpha = 1.0
    first = True
    while True:
        if not first:                                          # :118-125
            numiter += 1
            rho_old = rho
            rho = inner(r_shadow, r)
            beta = (rho / rho_old) * (alpha / omega)
            p = add(p, v, -omega)
            p = add(p, r, 1.0, beta)
        v = aff(p)                                             # :50 / :127
        numops += 1
        sigma = inner(r_shadow, v)
        alpha = rho / sigma
        s = add(r.copy(), v, -alpha)                           # half step residual
        xhalf = add(x.copy(), p, alpha)
        normr = norm(s)
        if normr < tol:                                        # :65-80 / :142-157
            s = add(b.copy(), aff(xhalf), -1.0)
            numops += 1
            normr_act = norm(s)
            if normr_act < tol:
                return xhalf, ConvergenceInfo(1, s, normr_act, numiter, numops)
        t = aff(s)                                             # :83 / :163
        numops += 1
        omega = inner(t, s) / inner(t, t)
        x = add(xhalf, s, omega)
        r = add(s.copy(), t, -omega)
        normr = norm(r)
        if normr < tol:                                        # :94-110 / :175-190
            r = add(b.copy(), aff(x), -1.0)
            numops += 1
            normr_act = norm(r)
            if normr_act < tol:
                return x, ConvergenceInfo(1, r, normr_act, numiter, numops)
        if not first and numiter >= maxiter:                   # :191-198 (only inside the while loop)
            return x, ConvergenceInfo(0, r, normr, numiter, numops)
        first = False


def lsmr(operator, b: np.ndarray, lam: float = 0.0, *, krylovdim: int = 30, maxiter: int = 100, tol: float = 1e-12,
         orth=MGS):
    """lssolve(operator, b, alg::LSMR, lambda) (lssolve/lsmr.jl:1-151).  `operator` is a matrix (A @ x, A.T @ x) or a
    pair of callables (apply_normal, apply_adjoint) (apply.jl:14-19)."""
    if isinstance(operator, tuple):
        f_normal, f_adjoint = operator
    else:
        f_normal, f_adjoint = (lambda z: operator @ z), (lambda z: operator.T @ z)
    u = np.asarray(b, dtype=np.float64).copy()
    v = np.asarray(f_adjoint(u), dtype=np.float64).copy()      # :4
    beta = norm(u)
    u = scale_(u, 1 / beta)
    v = scale_(v, 1 / beta)
    alpha = norm(v)
    v = scale_(v, 1 / alpha)
    V = [v]                                                    # OrthonormalBasis([v])   :15
    K = krylovdim
    alphabar, zetabar, rho, theta, rhobar, cbar, sbar = alpha, alpha * beta, 1.0, 0.0, 1.0, 1.0, 0.0
    abszetabar = abs(zetabar)
    x = np.zeros_like(v)
    h = v.copy()
    hbar = np.zeros_like(v)
    r = scale(u, beta)
    Ah = np.zeros_like(u)
    Ahbar = np.zeros_like(u)
    numiter, numops = 0, 1
    if abszetabar < tol:                                       # :48-58
        return x, ConvergenceInfo(1, r, abszetabar, numiter, numops)
    while True:
        numiter += 1
        Av = np.asarray(f_normal(v), dtype=np.float64).copy()  # :63
        numops += 1
        Ah = add(Ah, Av, 1.0, -theta / rho)
        u = add(Av, u, -alpha)                                 # :68
        beta = norm(u)
        if beta > tol:
            u = scale_(u, 1 / beta)
            v = add(np.asarray(f_adjoint(u), dtype=np.float64).copy(), v, -beta)   # :73
            numops += 1
            if K > 1:
                v, _ = orthogonalize(v, V, orth)               # :76-78
            alpha = norm(v)
            if alpha > tol:
                v = scale_(v, 1 / alpha)
                if numiter < K:
                    V.append(v)
                else:
                    V[(numiter + 1 - 1) % K] = v               # mod1(numiter + 1, K), 0-based   :86
        alphahat = float(np.hypot(alphabar, lam))              # :92-94
        chat = alphabar / alphahat
        shat = lam / alphahat
        rhoold = rho                                           # :97-102
        rho = float(np.hypot(alphahat, beta))
        c = alphahat / rho
        s_ = beta / rho
        theta = s_ * alpha
        alphabar = c * alpha
        rhobarold = rhobar                                     # :105-112
        thetabar = sbar * rho
        cbarrho = cbar * rho
        rhobar = float(np.hypot(cbarrho, theta))
        cbar = cbarrho / rhobar
        sbar = theta / rhobar
        zeta = cbar * zetabar
        zetabar = -sbar * zetabar
        c1 = thetabar * rho / (rhoold * rhobarold)
        hbar = add(hbar, h, 1.0, -c1)                          # :115-116
        Ahbar = add(Ahbar, Ah, 1.0, -c1)
        c2 = zeta / (rho * rhobar)
        x = add(x, hbar, c2)                                   # :118-119
        r = add(r, Ahbar, -c2)
        h = add(h, v, 1.0, -theta / rho)                       # :121
        abszetabar = abs(zetabar)
        if abszetabar <= tol:
            return x, ConvergenceInfo(1, r, abszetabar, numiter, numops)
        if numiter >= maxiter:
            return x, ConvergenceInfo(0, r, abszetabar, numiter, numops)


def expintegrator(A, t: float, u: Sequence[np.ndarray], *, krylovdim: int = 30, maxiter: int = 100, tol: float = 1e-12,
                  orth: Orthogonalizer = MGS2, method: str = "lanczos", eager: bool = False):
    """expintegrator(A, t, u::Tuple, alg::Union{Lanczos,Arnoldi}) (matrixfun/expintegrator.jl:100-323), real t.
    exponentiate(A, t, v) = expintegrator(A, t, (v,)) (matrixfun/exponentiate.jl:83-84)."""
    import scipy.linalg as sla
    u = [np.asarray(z, dtype=np.float64) for z in u]
    if len(u) == 1:                                            # :101
        u = [u[0], np.zeros_like(u[0])]
    p = len(u) - 1
    u0 = u[0]
    Au0 = apply(A, u0)
    numops = 1
    w0 = scale(u0, 1.0)
    K = krylovdim
    HH = np.zeros((K + p + 1, K + p + 1))
    eta = tol                                                  # :120-134
    totalerr = 0.0
    sgn = float(np.sign(t))
    tau = abs(t)
    if np.isfinite(tau):
        dtau = tau
        dtaumin = tau / maxiter
        maxerr = tau * eta
    else:
        dtau = 1.0
        dtaumin = 0.0
        maxerr = eta
    gamma = 0.8
    tau0 = 0.0
    w: List[Optional[np.ndarray]] = [None] * (p + 1)
    w[0] = w0
    w[1] = scale(Au0, 1.0)

    def stage_vectors(first: bool):
        nonlocal numops
        for j in range(1, p + 1):                              # :146-158 / :293-301
            if j > 1 or not first:
                w[j] = apply(A, w[j - 1])
                numops += 1
            lfac = 1
            for l in range(0, p - j + 1):
                w[j] = add(w[j], u[j + l], (sgn * tau0) ** l / lfac)
                lfac *= l + 1

    def small_exp(fact, step):
        Kc = len(fact)
        H = np.zeros((Kc + p + 1, Kc + p + 1))
        rq = fact.rayleighquotient()
        if isinstance(rq, tuple):
            d, e = rq
            rq = np.diag(d) + np.diag(e, 1) + np.diag(e, -1)
        H[:Kc, :Kc] = rq * (sgn * step)
        H[0, Kc] = 1.0
        for i in range(1, p + 1):
            H[Kc + i - 1, Kc + i] = 1.0
        return sla.expm(H)

    def take_step(fact, expH, step):
        nonlocal w0
        Kc = len(fact)
        jfac = 1
        for j in range(1, p):
            w0 = add(w0, w[j], (sgn * step) ** j / jfac)
            jfac *= j + 1
        w[p] = unproject(w[p], fact.V[:Kc], expH[:Kc, Kc + p - 1])      # :233 / :263
        w[p] = add(w[p], fact.r, expH[Kc - 1, Kc + p])
        w0 = add(w0, w[p], beta * (sgn * step) ** p)
        w[0] = w0

    stage_vectors(True)
    beta = norm(w[p])
    if beta < eta and p == 1:                                  # :161-166
        return w0, ConvergenceInfo(1, None, beta, 0, numops)
    mk_iter = LanczosIterator if method == "lanczos" else ArnoldiIterator
    init, init_, expand = ((lanczos_initialize, lanczos_initialize_, lanczos_expand) if method == "lanczos"
                           else (arnoldi_initialize, arnoldi_initialize_, arnoldi_expand))
    it = mk_iter(A, w[p], orth)
    fact = init(it)
    numops += 1
    numiter = 1
    while True:
        Kc = len(fact)
        if Kc == krylovdim:                                    # :184-241
            if numiter < maxiter:
                dtau = min(dtau, tau - tau0)
                if np.isfinite(tau):
                    dtaumin = (tau - tau0) / (maxiter - numiter + 1)
            else:
                dtau = tau - tau0
            expH = small_exp(fact, dtau)
            eps_ = abs(dtau ** p * beta * fact.normres * expH[Kc - 1, Kc + p])
            omega = eps_ / (dtau * eta)
            q = Kc / 2
            while numiter < maxiter and omega >= 1.0 and dtau > dtaumin:
                eps_prev, dtau_prev = eps_, dtau
                dtau = max(dtau * (gamma / omega) ** (1 / (q + 1)), dtaumin)
                expH = small_exp(fact, dtau)
                eps_ = abs(dtau ** p * beta * fact.normres * expH[Kc - 1, Kc + p])
                omega = eps_ / (dtau * eta)
                q = max(0.0, math.log(eps_ / eps_prev) / math.log(dtau / dtau_prev) - 1)
            tau0 = tau0 + dtau if numiter < maxiter else tau
            totalerr += eps_
            take_step(fact, expH, dtau)
            if omega < gamma:
                dtau *= (gamma / omega) ** (1 / (q + 1))
        elif fact.normres <= (tau - tau0) * eta or eager:      # :242-268
            step = tau - tau0
            expH = small_exp(fact, step)
            eps_ = abs(step ** p * beta * fact.normres * expH[Kc - 1, Kc + p])
            omega = eps_ / (step * eta)
            if omega < 1.0:
                totalerr += eps_
                take_step(fact, expH, step)
                tau0 = tau
        if tau0 >= tau:                                        # :269-285
            return w0, ConvergenceInfo(1 if totalerr <= maxerr else 0, None, totalerr, numiter, numops)
        if Kc < krylovdim:
            fact = expand(it, fact)
            numops += 1
        else:
            stage_vectors(False)
            beta = norm(w[p])
            if beta < eta and p == 1:                          # :302-307
                return w0, ConvergenceInfo(1, None, beta, numiter, numops)
            it = mk_iter(A, w[p], orth)
            fact = init_(it, fact)
            numops += 1
            numiter += 1


# --------------------------------------------------------------------------------------
# eigsolve / schursolve with Arnoldi (Krylov-Schur) -- src/eigsolve/arnoldi.jl, dense/linalg.jl:152-383
# --------------------------------------------------------------------------------------
def _eigsort_complex(which: str):
    """eigsort (eigsolve/eigsolve.jl:334-355)."""
    if which == "LM":
        return np.abs, True
    if which == "LR":
        return np.real, True
    if which == "SR":
        return np.real, False
    if which == "LI":
        return np.imag, True
    if which == "SI":
        return np.imag, False
    raise ValueError(f"invalid specification of which eigenvalues to target: which = {which}")


def _schur_values(T: np.ndarray) -> np.ndarray:
    """schur2eigvals for real quasi-triangular T (dense/linalg.jl:166-189)."""
    n = T.shape[0]
    out = np.zeros(n, dtype=complex)
    i = 0
    while i < n:
        if i + 1 < n and T[i + 1, i] != 0:
            tr2 = (T[i, i] + T[i + 1, i + 1]) / 2
            df = (T[i, i] - T[i + 1, i + 1]) / 2
            disc = df * df + T[i, i + 1] * T[i + 1, i]
            out[i] = tr2 + 1j * math.sqrt(-disc)
            out[i + 1] = tr2 - 1j * math.sqrt(-disc)
            i += 2
        else:
            out[i] = T[i, i]
            i += 1
    return out


def _permute_schur(T: np.ndarray, Q: np.ndarray, order: Sequence[int]):
    """permuteschur!(T, Q, order) for real T (dense/linalg.jl:356-383) through LAPACK trexc."""
    from scipy.linalg import lapack
    n = T.shape[0]
    p = [k + 1 for k in order]
    T, Q = np.array(T, order="F"), np.array(Q, order="F")
    i = 1
    while i <= len(p):
        ifirst, ilast = p[i - 1], i
        if ifirst == n or T[ifirst, ifirst - 1] == 0:
            T, Q, info = lapack.dtrexc(T, Q, ifirst, ilast)
            assert info == 0
            for k in range(i, len(p)):
                if p[k] < p[i - 1]:
                    p[k] += 1
            i += 1
        else:
            if p[i] != ifirst + 1:
                raise RuntimeError("cannot split 2x2 blocks when permuting schur decomposition")
            T, Q, info = lapack.dtrexc(T, Q, ifirst, ilast)
            assert info == 0
            for k in range(i + 1, len(p)):
                if p[k] < p[i - 1]:
                    p[k] += 2
            i += 2
    return T, Q, _schur_values(T)


def _schur_eigvecs(T: np.ndarray) -> np.ndarray:
    """schur2eigvecs for real T (dense/linalg.jl:223-246): normalised eigenvectors in diagonal order.  LAPACK trevc is
    not exposed by SciPy; the eigenvectors of the small quasi-triangular matrix come from numpy.linalg.eig, matched
    to the diagonal order of T (they are unique up to a phase when the eigenvalues are distinct)."""
    lam = _schur_values(T)
    w, X = np.linalg.eig(T)
    n = T.shape[0]
    out = np.zeros((n, n), dtype=complex)
    free = list(range(n))
    for j in range(n):
        k = min(free, key=lambda q: abs(w[q] - lam[j]))
        free.remove(k)
        out[:, j] = X[:, k] / np.linalg.norm(X[:, k])
    return out


def _schursolve_arnoldi(A, x0, howmany, which, krylovdim, maxiter, tol, orth, eager):
    """_schursolve (eigsolve/arnoldi.jl:361-464)."""
    import scipy.linalg as sla
    if howmany > krylovdim:
        raise ValueError(f"krylov dimension {krylovdim} too small to compute {howmany} eigenvalues")
    numiter = 1
    it = ArnoldiIterator(A, np.asarray(x0, dtype=np.float64), orth)
    fact = arnoldi_initialize(it)
    numops = 1
    converged = 0
    T = U = f = None
    while True:
        beta = fact.normres
        K = len(fact)
        if K == krylovdim or beta <= tol or (eager and K >= howmany):          # :389
            T, U = sla.schur(fact.rayleighquotient(), output="real")            # hschur!  :396
            vals = _schur_values(T)
            by, rev = _eigsort_complex(which)
            key = by(vals)
            perm = np.argsort(-key if rev else key, kind="stable")
            T, U, vals = _permute_schur(T, U, list(perm))
            f = U[K - 1, :] * beta                                              # :400
            converged = 0
            while converged < K and abs(f[converged]) <= tol:
                converged += 1
            if 0 < converged < K and T[converged, converged - 1] != 0:
                converged -= 1
            if converged >= howmany or beta <= tol:
                break
        if K < krylovdim:
            fact = arnoldi_expand(it, fact)
            numops += 1
        else:
            if numiter == maxiter:
                break
            keep = (3 * krylovdim + 2 * converged) // 5
            if T[keep, keep - 1] != 0:                                          # :424-436
                if keep > 1:
                    keep -= 1
                else:
                    keep += 1
                    if krylovdim == 2:
                        break
            H = np.array(T)
            for j in range(keep):                                               # _restorearnoldiform!  :466-480
                H[keep, j] = f[j]
            for j in range(keep, 0, -1):
                hb, hv, nu = householder_vec(H[j, :j], j - 1)
                H[j, j - 1] = nu
                H[j, : j - 1] = 0.0
                r = np.arange(j)
                householder_lmul(hb, hv, r, H)
                householder_rmul_mat(H, hb, hv, r, rows=slice(0, j))
                householder_rmul_mat(U, hb, hv, r)
            for j in range(1, K + 1):                                           # copy!(rayleighquotient(fact), H)  :443
                for i in range(1, min(j + 1, K) + 1):
                    fact.H[packed_index(i, j)] = float(H[i - 1, j - 1])
            B = basistransform(list(fact.V[:K]), U[:, :keep])                   # :444
            for j in range(keep):
                fact.V[j] = B[j]
            fact.V[keep] = scale(fact.r, 1.0 / beta)                            # B[keep+1] = scale!!(residual, 1/beta)
            fact = arnoldi_shrink(fact, keep)
            numiter += 1
    return T, U, fact, converged, numiter, numops


def eigsolve_arnoldi(A, x0, howmany: int = 1, which: str = "LM", *, krylovdim: int = 30, maxiter: int = 100,
                     tol: float = 1e-12, orth: Orthogonalizer = MGS2, eager: bool = False):
    """eigsolve(A, x0, howmany, which, alg::Arnoldi) (eigsolve/arnoldi.jl:277-316): complex eigenvalues, eigenvectors
    (numpy arrays), ConvergenceInfo."""
    T, U, fact, converged, numiter, numops = _schursolve_arnoldi(A, x0, howmany, which, krylovdim, maxiter, tol, orth, eager)
    hm = howmany
    if howmany < len(fact) and T[howmany, howmany - 1] != 0:
        hm += 1
    elif T.shape[0] < howmany:
        hm = T.shape[0]
    if converged > howmany:
        hm = converged
    TT = T[:hm, :hm]
    values = _schur_values(TT)
    V = U[:, :hm] @ _schur_eigvecs(TT)
    K = len(fact)
    Bm = np.stack(fact.V[:K], axis=1)
    vectors = [Bm @ V[:, i] for i in range(hm)]
    residuals = [fact.r * V[K - 1, i] for i in range(hm)]
    normres = np.array([fact.normres * abs(V[K - 1, i]) for i in range(hm)])
    return values, vectors, ConvergenceInfo(converged, residuals, normres, numiter, numops)


# --------------------------------------------------------------------------------------
# geneigsolve with Golub-Ye -- src/eigsolve/golubye.jl
# --------------------------------------------------------------------------------------
def _checkposdef(z: float) -> float:
    if not z > 0:                                              # KrylovKit.jl:143-148
        raise ValueError(f"operator does not appear to be positive definite: diagonal element {z}")
    return z


def golubyerecurrence(A, B, rho, V: List[np.ndarray], beta, orth: Orthogonalizer):
    """golubyerecurrence (golubye.jl:182-281): Lanczos recurrence of A - rho B; returns (w, alpha, beta, B v)."""
    v = V[-1]
    av, bv = apply(A, v), apply(B, v)                          # genapply (apply.jl:22)
    w = add(av, bv, -rho)
    n = orth.name
    if n in ("cgs", "cgs2", "cgsir"):
        alpha = inner(v, w)
        w = add(w, V[-2], -beta)
        w = add(w, v, -alpha)
        if n == "cgs2":
            w, s = orthogonalize(w, V, CGS)
            alpha += s[-1]
            beta = norm(w)
        elif n == "cgsir":
            ab2 = alpha * alpha + beta * beta
            beta = norm(w)
            nold = math.sqrt(beta * beta + ab2)
            while EPS < beta < orth.eta * nold:
                nold = beta
                w, s = orthogonalize(w, V, CGS)
                alpha += s[-1]
                beta = norm(w)
        else:
            beta = norm(w)
        return w, alpha, beta, bv
    w = add(w, V[-2], -beta)
    w, alpha = orthogonalize_vec(w, v, MGS)
    if n == "mgs2":
        s = alpha
        for q in V:
            w, s = orthogonalize_vec(w, q, MGS)
        alpha += s
        beta = norm(w)
    elif n == "mgsir":
        ab2 = alpha * alpha + beta * beta
        beta = norm(w)
        nold = math.sqrt(beta * beta + ab2)
        while EPS < beta < orth.eta * nold:
            nold = beta
            s = 0.0
            for q in V:
                w, s = orthogonalize_vec(w, q, MGS)
            alpha += s
            beta = norm(w)
    else:
        beta = norm(w)
    return w, alpha, beta, bv


def geneigsolve_golubye(A, B, x0, howmany: int = 1, which: str = "SR", *, krylovdim: int = 30, maxiter: int = 100,
                        tol: float = 1e-12, orth: Orthogonalizer = MGS2):
    """geneigsolve((A, B), x0, howmany, which, alg::GolubYe) (eigsolve/golubye.jl:1-180), real symmetric A, s.p.d. B."""
    import scipy.linalg as sla
    if howmany > krylovdim:
        raise ValueError(f"krylov dimension {krylovdim} too small to compute {howmany} eigenvalues")
    x0 = np.asarray(x0, dtype=np.float64)
    ax0, bx0 = apply(A, x0), apply(B, x0)
    numops = 1
    beta0 = norm(x0)
    if beta0 == 0:
        raise ValueError("initial vector should not have norm zero")
    xax = inner(x0, ax0) / beta0 ** 2
    xbx = inner(x0, bx0) / beta0 ** 2
    v = scale(x0, 1 / beta0)
    av = scale(ax0, 1 / beta0)
    bv = scale(bx0, 1 / beta0)
    rho = xax / _checkposdef(xbx)
    r = add(av, bv, -rho)
    HHA = np.zeros((krylovdim + 1, krylovdim + 1))
    numiter = 1
    vold = v
    V: List[np.ndarray] = [v]
    BV: List[np.ndarray] = [bv]
    r, alpha = orthogonalize_vec(r, v, orth)                   # :44
    beta = norm(r)
    converged = 0
    values: List[float] = []
    vectors: List[np.ndarray] = []
    residuals: List[np.ndarray] = []
    normres: List[float] = []
    K = 1
    HHA[0, 0] = alpha
    by, rev = eigsort_key(which)
    while True:
        beta = norm(r)
        if beta <= tol and K < howmany:                        # :59-67
            howmany = K
        if K == krylovdim - converged or beta <= tol:          # process  :68
            def extend(vnew):
                nonlocal K, numops
                avn, bvn = apply(A, vnew), apply(B, vnew)
                numops += 1
                avn = add(avn, bvn, -rho)
                for i in range(K):
                    HHA[i, K] = inner(V[i], avn)
                    HHA[K, i] = HHA[i, K]
                K += 1
                HHA[K - 1, K - 1] = inner(vnew, avn)
                V.append(vnew)
                BV.append(bvn)

            if numiter > 1:                                    # :69-82
                vn, _, _ = orthonormalize(vold, V, orth)
                extend(vn)
            for i in range(converged):                         # :83-96
                vn, _, _ = orthonormalize(vectors[i].copy(), V, orth)
                extend(vn)
            HA = HHA[:K, :K].copy()
            HB = np.zeros((K, K))
            for j in range(K):                                 # buildHB!  :284-295
                HB[j, j] = _checkposdef(inner(V[j], BV[j]))
                for i in range(j + 1, K):
                    HB[i, j] = inner(V[i], BV[j])
                    HB[j, i] = HB[i, j]
            HA += rho * HB
            D, Z = sla.eigh(HA, HB)                            # geneigh! -> sygvd  :102
            key = by(D)
            perm = np.argsort(-key if rev else key, kind="stable")
            converged = 0
            values, vectors, residuals, normres = [], [], [], []
            for k in range(K):                                 # :110-133
                z = Z[:, perm[k]]
                v = unproject(np.zeros_like(vold), V, z)
                av, bv = apply(A, v), apply(B, v)
                numops += 1
                rho = inner(v, av) / _checkposdef(inner(v, bv))
                r = add(av, bv, -rho)
                beta = norm(r)
                if beta < tol * norm(z):
                    converged += 1
                elif numiter < maxiter:
                    break
                values.append(rho)
                vectors.append(v)
                residuals.append(r)
                normres.append(beta)
                if k + 1 == howmany and numiter == maxiter:
                    break
            if converged >= howmany:
                howmany = converged
                break
        if K < krylovdim - converged:                          # expand  :143-157
            v = scale_(r, 1 / beta)
            V.append(v)
            HHA[K, K - 1] = beta
            HHA[K - 1, K] = beta
            beta_old = beta
            r, alpha, beta, bv = golubyerecurrence(A, B, rho, V, beta_old, orth)
            numops += 1
            K += 1
            HHA[K - 1, K - 1] = alpha
            BV.append(bv)
        else:                                                  # restart  :158-177
            if numiter == maxiter:
                break
            V.clear()
            BV.clear()
            HHA[:] = 0.0
            K = 1
            invb = 1 / norm(v)
            v = scale_(v, invb)
            bv = scale_(bv, invb)
            r = scale_(r, invb)
            r, alpha = orthogonalize_vec(r, v, orth)
            beta = norm(r)
            V.append(v)
            HHA[0, 0] = alpha
            BV.append(bv)
            numiter += 1
    return np.array(values), vectors, ConvergenceInfo(converged, residuals, np.array(normres), numiter, numops)


# --------------------------------------------------------------------------------------
# bieigsolve with BiArnoldi (two-sided Krylov-Schur) -- src/eigsolve/biarnoldi.jl, factorizations/biarnoldi.jl
# --------------------------------------------------------------------------------------
def _restore_arnoldi_form(U, H, f, keep):
    """_restorearnoldiform!(U, H, f, keep) (eigsolve/arnoldi.jl:466-480)."""
    for j in range(keep):
        H[keep, j] = f[j]
    for j in range(keep, 0, -1):
        hb, hv, nu = householder_vec(H[j, :j], j - 1)
        H[j, j - 1] = nu
        H[j, : j - 1] = 0.0
        r = np.arange(j)
        householder_lmul(hb, hv, r, H)
        householder_rmul_mat(H, hb, hv, r, rows=slice(0, j))
        householder_rmul_mat(U, hb, hv, r)


def _bischursolve(A, v0, w0, howmany, which, krylovdim, maxiter, tol, orth, eager):
    """_bischursolve (eigsolve/biarnoldi.jl:196-393), real Float64."""
    import scipy.linalg as sla
    if howmany > krylovdim:
        raise ValueError(f"krylov dimension {krylovdim} too small to compute {howmany} eigenvalues")
    if isinstance(A, tuple):
        fn, fa = A
    else:
        fn, fa = (lambda z: A @ z), (lambda z: A.T @ z)
    numiter = 1
    itV = ArnoldiIterator(fn, np.asarray(v0, dtype=np.float64), orth)       # BiArnoldiIterator (factorizations/biarnoldi.jl:24-41)
    itW = ArnoldiIterator(fa, np.asarray(w0, dtype=np.float64), orth)
    fV, fW = arnoldi_initialize(itV), arnoldi_initialize(itW)
    numops = 1
    MM = np.zeros((krylovdim, krylovdim))
    MM[0, 0] = inner(fW.V[0], fV.V[0])
    converged = 0
    by, rev = _eigsort_complex(which)
    S = T = Q = Z = rV = rW = h = k = M = None
    MinvWv = MinvVw = None
    while True:
        bv, bw = fV.normres, fW.normres
        Lk = len(fV)
        if Lk == krylovdim or (bv <= tol and bw <= tol) or (eager and Lk >= howmany):     # process  :232
            H = fV.rayleighquotient()
            K = fW.rayleighquotient()
            M = MM[:Lk, :Lk]
            rV = scale(fV.r, 1 / bv)                                                   # :248-254
            rW = scale(fW.r, 1 / bw)
            V, W = fV.V[:Lk], fW.V[:Lk]
            Wv = np.array([inner(W[i], rV) for i in range(Lk)])
            Vw = np.array([inner(V[i], rW) for i in range(Lk)])
            lu = sla.lu_factor(M)                                                      # :260-262
            MinvWv = sla.lu_solve(lu, Wv)
            MinvVw = sla.lu_solve(lu, Vw, trans=1)
            H[:, Lk - 1] += bv * MinvWv                                                # :263-264
            K[:, Lk - 1] += bw * MinvVw
            for i in range(Lk):
                rV = add(rV, V[i], -MinvWv[i])
                rW = add(rW, W[i], -MinvVw[i])
            brV, brW = norm(rV), norm(rW)
            S, Q = sla.schur(H, output="real")                                         # hschur!  :271-272
            T, Z = sla.schur(K, output="real")
            valsH, valsK = _schur_values(S), _schur_values(T)
            kH = by(valsH)
            kK = by(np.conj(valsK))
            pH = np.argsort(-kH if rev else kH, kind="stable")
            pK = np.argsort(-kK if rev else kK, kind="stable")
            S, Q, _ = _permute_schur(S, Q, list(pH))
            T, Z, _ = _permute_schur(T, Z, list(pK))
            h = Q[Lk - 1, :] * bv                                                      # :280-281
            k = Z[Lk - 1, :] * bw
            converged = 0
            while converged < Lk:
                if max(brV * abs(h[converged]), brW * abs(k[converged])) <= tol:
                    converged += 1
                else:
                    break
            if 0 < converged < Lk and S[converged, converged - 1] != 0:
                converged -= 1
            if converged >= howmany or (bv <= tol and bw <= tol):
                break
        if Lk < krylovdim:                                                             # expand  :303-312
            fV = arnoldi_expand(itV, fV)
            fW = arnoldi_expand(itW, fW)
            V, W = fV.V, fW.V                                                          # V[L+1], W[L+1] are the vectors just pushed
            for i in range(Lk):
                MM[i, Lk] = inner(W[i], V[Lk])
                MM[Lk, i] = inner(W[Lk], V[i])
            MM[Lk, Lk] = inner(W[Lk], V[Lk])
            numops += 2
        else:                                                                          # shrink  :313-358
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
            QVQv = Q[:, :keep] @ VQv
            ZWZw = Z[:, :keep] @ WZw
            V, W = fV.V[:Lk], fW.V[:Lk]
            for i in range(Lk):
                rV = add(rV, V[i], -QVQv[i])
                rW = add(rW, W[i], -ZWZw[i])
            brV, brW = norm(rV), norm(rW)
            rV = scale_(rV, 1 / brV)
            rW = scale_(rW, 1 / brW)
            h = h * brV
            k = k * brW
            _restore_arnoldi_form(Q, H, h, keep)
            _restore_arnoldi_form(Z, K, k, keep)
            for fact, Hm, Um, rnew in ((fV, H, Q, rV), (fW, K, Z, rW)):
                for j in range(1, Lk + 1):                                             # copy!(rayleighquotient(fact), H)
                    for i in range(1, min(j + 1, Lk) + 1):
                        fact.H[packed_index(i, j)] = float(Hm[i - 1, j - 1])
                B = basistransform(list(fact.V[:Lk]), Um[:, :keep])
                for j in range(keep):
                    fact.V[j] = B[j]
                fact.V[keep] = rnew
                arnoldi_shrink(fact, keep)
            MM[:keep, :keep] = Z[:, :keep].T @ (M @ Q[:, :keep])                       # :352-356
            numiter += 1
    return (S, T), (Q, Z), (fV, fW), (rV, rW), (h, k), M, converged, numiter, numops


def bieigsolve_biarnoldi(A, v0, w0, howmany: int = 1, which: str = "LM", *, krylovdim: int = 30, maxiter: int = 100,
                         tol: float = 1e-12, orth: Orthogonalizer = MGS2, eager: bool = False):
    """bieigsolve(f, v0, w0, howmany, which, alg::BiArnoldi) (eigsolve/biarnoldi.jl:127-194): eigenvalues, right and left
    eigenvectors (biorthonormal: W' V = I), two ConvergenceInfo."""
    (S, T), (Q, Z), (fV, fW), (rV, rW), (h, k), M, converged, numiter, numops = _bischursolve(
        A, v0, w0, howmany, which, krylovdim, maxiter, tol, orth, eager)
    hm = howmany
    if howmany < T.shape[0] and T[howmany, howmany - 1] != 0:
        hm += 1
    elif T.shape[0] < howmany:
        hm = T.shape[0]
    if converged > howmany:
        hm = converged
    SS = S[:hm, :hm]
    valuesS = _schur_values(SS)
    vecsS = _schur_eigvecs(SS)
    ZMQ = Z[:, :hm].T @ M @ Q[:, :hm]
    vecsT = np.linalg.inv((ZMQ @ vecsS).conj().T)
    VS = Q[:, :hm] @ vecsS
    VT = Z[:, :hm] @ vecsT
    Lk = len(fV)
    Vm, Wm = np.stack(fV.V[:Lk], axis=1), np.stack(fW.V[:Lk], axis=1)
    vectorsS = [Vm @ VS[:, i] for i in range(hm)]
    vectorsT = [Wm @ VT[:, i] for i in range(hm)]
    hVS = [h[:hm] @ vecsS[:, i] for i in range(hm)]
    kVT = [k[:hm] @ vecsT[:, i] for i in range(hm)]
    infoS = ConvergenceInfo(converged, [rV * s for s in hVS], np.array([norm(rV) * abs(s) for s in hVS]), numiter, numops)
    infoT = ConvergenceInfo(converged, [rW * s for s in kVT], np.array([norm(rW) * abs(s) for s in kVT]), numiter, numops)
    return valuesS, (vectorsS, vectorsT), (infoS, infoT)

