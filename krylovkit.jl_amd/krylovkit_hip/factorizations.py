"""Krylov factorizations on the device: host-side mirror of src/factorizations/{lanczos,arnoldi,gkl}.jl.

Same names and meaning as the reference (`LanczosIterator`, `initialize`, `expand!` -> `expand_`,
`shrink!` -> `shrink_`, `initialize!` -> `initialize_`, `basis`, `rayleighquotient`, `residual`,
`normres`); the state lives in a DeviceBasis slab:

    columns 0..k-1  = V (the orthonormal Krylov basis)
    column  k       = r (the residual), |r| = normres

so `push!(V, scale!!(r, 1/beta))` (lanczos.jl:257) is an in-place scale plus `length += 1`, and
each `expand_` is ONE call into libkrylov_hip.so (kk_lanczos_expand / kk_arnoldi_expand /
kk_gkl_expand) with one host synchronisation.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass, field
from typing import List, Optional

import numpy as np

from . import _lib
from ._lib import check
from .core import (ClassicalGramSchmidt, Context, DeviceBasis, FunctionOperator, HipVec, KrylovDefaults,
                   ModifiedGramSchmidt, Orthogonalizer, SparseOperator, default_context)


def _as_operator(A, ctx=None) -> SparseOperator:
    if isinstance(A, (SparseOperator, FunctionOperator)):
        return A
    return SparseOperator(A, ctx)


_EPS = float(np.finfo(np.float64).eps)


def lanczos_recurrence_unfused(V: DeviceBasis, c0: int, K: int, w: HipVec, beta_old: float, orth: Orthogonalizer,
                               have_prev: bool = True):
    """lanczosrecurrence x6 (factorizations/lanczos.jl:295-376) AFTER the operator application, for w = A V[K-1] already
    in `w`: the three-term part through the L1 verbs, the re-orthogonalisation through the fused pass entry points.
    Basis vectors are columns c0 .. c0+K-1 of V.  Returns (alpha, beta); w holds the new residual."""
    v = HipVec(V, c0 + K - 1)
    vprev = HipVec(V, c0 + K - 2) if have_prev else None
    name = orth.name
    if name in ("cgs", "cgs2", "cgsir"):                        # :297-301, :313-323, :341-356
        a = v.inner(w)
        if vprev is not None:
            w.add_(vprev, -beta_old)
        w.add_(v, -a)
        if name == "cgs":
            return a, w.norm()
        if name == "cgs2":
            s, b, _ = V.orthogonalize(w, ClassicalGramSchmidt(), c0, K)
            return a + s[-1], b
        ab2 = a * a + beta_old * beta_old
        b = w.norm()
        nold = float(np.sqrt(b * b + ab2))
        while _EPS < b < orth.eta * nold:
            nold = b
            s, b, _ = V.orthogonalize(w, ClassicalGramSchmidt(), c0, K)
            a += s[-1]
        return a, b
    if vprev is not None:                                       # :306-310, :326-338, :359-375
        w.add_(vprev, -beta_old)
    a, b = w.orthogonalize_against_(v, ModifiedGramSchmidt())
    if name == "mgs":
        return a, b
    if name == "mgs2":
        s, b, _ = V.orthogonalize(w, ModifiedGramSchmidt(), c0, K)   # for q in V: orthogonalize!!(w, q, MGS); s = last
        return a + s[-1], b
    ab2 = a * a + beta_old * beta_old
    nold = float(np.sqrt(b * b + ab2))
    while _EPS < b < orth.eta * nold:
        nold = b
        s, b, _ = V.orthogonalize(w, ModifiedGramSchmidt(), c0, K)
        a += s[-1]
    return a, b


def _krylov_initialize_unfused(op, V: DeviceBasis, orth: Orthogonalizer):
    """initialize(iter) for a function operator (lanczos.jl:180-222 == arnoldi.jl:135-175): returns (alpha, beta)."""
    x0, r = HipVec(V, 0), HipVec(V, 1)
    beta0 = x0.norm()
    if beta0 == 0:
        raise _lib.KrylovHipError(_lib.KK_ERR_ZERO_NORM, "initial vector should not have norm zero")
    op.apply(x0, r)                                              # Ax0   :186
    alpha = x0.inner(r) / (beta0 * beta0)
    x0.scale_(1.0 / beta0)                                       # v
    r.scale_(1.0 / beta0)                                        # :194
    beta_old = r.norm()
    r.add_(x0, -alpha)
    beta = r.norm()
    if orth.name in ("cgs2", "mgs2"):                            # :200-204
        da = x0.inner(r)
        alpha += da
        r.add_(x0, -da)
        beta = r.norm()
    elif orth.name in ("cgsir", "mgsir"):                        # :205-213
        while _EPS < beta < orth.eta * beta_old:
            beta_old = beta
            da = x0.inner(r)
            alpha += da
            r.add_(x0, -da)
            beta = r.norm()
    return alpha, beta


# =========================================================================================
# Lanczos -- src/factorizations/lanczos.jl
# =========================================================================================
@dataclass
class LanczosFactorization:  # lanczos.jl:31-37
    k: int
    V: DeviceBasis
    alphas: List[float]
    betas: List[float]
    window: int = 0  # first live column when keepvecs=false (sliding window of 2 vectors)

    def __len__(self):
        return self.k

    @property
    def normres(self) -> float:  # lanczos.jl:57
        return self.betas[self.k - 1]

    @property
    def r(self) -> HipVec:  # residual(F)
        return HipVec(self.V, self.window + len(self.V))

    def basis(self) -> DeviceBasis:  # lanczos.jl:49-52
        if len(self.V) != self.k:
            raise RuntimeError("Not keeping vectors during Lanczos factorization")
        return self.V

    def rayleighquotient(self):
        """SymTridiagonal(alphas[1:k], betas[1:k-1]) (lanczos.jl:53-55) as (diag, offdiag)."""
        return np.array(self.alphas[: self.k]), np.array(self.betas[: self.k - 1])


@dataclass
class LanczosIterator:  # lanczos.jl:129-153
    operator: SparseOperator
    x0: np.ndarray
    orth: Orthogonalizer = KrylovDefaults.orth
    keepvecs: bool = True
    capacity: int = KrylovDefaults.krylovdim + 2  # columns of the slab: krylovdim + residual + 1 work

    def __post_init__(self):
        if not self.keepvecs and self.orth.is_reorth:
            raise ValueError("Cannot use reorthogonalization without keeping all Krylov vectors")  # lanczos.jl:140-142
        _check_capacity(self.capacity, self.keepvecs)


def _check_capacity(capacity: int, keepvecs: bool = True):
    """No limit, as in the reference: beyond KK_MAX_M basis vectors the library's expand! entry points orthogonalise panel
    by panel (csrc/kk_orth.hip::orth_run_wide) instead of through the single-panel fused kernels."""
    return None


def initialize(it, V: Optional[DeviceBasis] = None):
    """initialize(iter) for LanczosIterator / ArnoldiIterator / GKLIterator (and any iterator
    object that brings its own `initialize` / `expand`, e.g. the row-sharded ones of dist.py)."""
    if hasattr(it, "initialize"):
        return it.initialize(V)
    if isinstance(it, LanczosIterator):
        return _lanczos_initialize(it, V)
    if isinstance(it, ArnoldiIterator):
        return _arnoldi_initialize(it, V)
    if isinstance(it, GKLIterator):
        return _gkl_initialize(it)
    raise TypeError(type(it))


def expand_(it, fact):
    if hasattr(it, "expand"):
        return it.expand(fact)
    if isinstance(it, LanczosIterator):
        return _lanczos_expand(it, fact)
    if isinstance(it, ArnoldiIterator):
        return _arnoldi_expand(it, fact)
    if isinstance(it, GKLIterator):
        return _gkl_expand(it, fact)
    raise TypeError(type(it))


def shrink_(fact, k: int):
    if isinstance(fact, LanczosFactorization):
        return _lanczos_shrink(fact, k)
    if isinstance(fact, ArnoldiFactorization):
        return _arnoldi_shrink(fact, k)
    if isinstance(fact, GKLFactorization):
        return _gkl_shrink(fact, k)
    raise TypeError(type(fact))


def initialize_(it, fact):
    if isinstance(it, LanczosIterator):
        return _lanczos_initialize_(it, fact)
    if isinstance(it, ArnoldiIterator):
        return _arnoldi_initialize_(it, fact)
    raise TypeError(type(it))


def _upload_x0(V: DeviceBasis, x0, col: int = 0):
    if isinstance(x0, HipVec):
        HipVec(V, col).scale_from_(x0, 1.0)
    else:
        V.upload(col, np.asarray(x0, dtype=np.float64))


def _lanczos_initialize(it: LanczosIterator, V: Optional[DeviceBasis] = None) -> LanczosFactorization:
    """initialize(iter::LanczosIterator) (lanczos.jl:180-222)"""
    op = it.operator
    n = op.shape[0]
    if V is None:
        V = DeviceBasis(n, it.capacity, op.ctx)
    _upload_x0(V, it.x0, 0)
    if isinstance(op, FunctionOperator):
        alpha, beta = _krylov_initialize_unfused(op, V, it.orth)
        V.length = 1
        return LanczosFactorization(1, V, [alpha], [beta])
    a, b = C.c_double(), C.c_double()
    check(V._lib.kk_lanczos_initialize(op.handle, V.handle, 0, it.orth.code, it.orth.eta, C.byref(a), C.byref(b)))
    V.length = 1
    return LanczosFactorization(1, V, [a.value], [b.value])


def _lanczos_initialize_(it: LanczosIterator, st: LanczosFactorization) -> LanczosFactorization:
    """initialize!(iter, state) (lanczos.jl:223-249): reuse the slab."""
    V = st.V
    st.window = 0
    _upload_x0(V, it.x0, 0)
    v1, w = HipVec(V, 0), HipVec(V, 1)
    v1.scale_(1.0 / v1.norm())                       # V[1] = scale!!(V[1], x0, 1/norm(x0))   :235
    it.operator.apply(v1, w)                         # w = apply(operator, V[1])              :236
    alpha, beta = w.orthogonalize_against_(v1, it.orth)  # r, alpha = orthogonalize!!(w, V[1], orth); beta = norm(r)
    V.length = 1
    st.k = 1
    st.alphas[:] = [alpha]
    st.betas[:] = [beta]
    return st


def _lanczos_expand(it: LanczosIterator, st: LanczosFactorization) -> LanczosFactorization:
    """expand!(iter::LanczosIterator, state) (lanczos.jl:250-272) -> kk_lanczos_expand."""
    V = st.V
    beta_old = st.normres
    nv = len(V)  # vectors currently held (k, or <= 1... 2 when keepvecs=false)
    c0 = st.window
    if c0 + nv + 2 > V.capacity:
        if it.keepvecs:
            raise RuntimeError(f"Lanczos slab of capacity {V.capacity} is full at k={st.k}")
        # keepvecs=false: slide the 2-vector window back to the front of the slab
        for j in range(nv + 1):  # live vectors + residual
            HipVec(V, j).scale_from_(HipVec(V, c0 + j), 1.0)
        c0 = st.window = 0
    a, b, npass = C.c_double(), C.c_double(), C.c_int()
    if isinstance(it.operator, FunctionOperator):   # un-fused: V[k+1] = r / beta ; w = f(v) ; recurrence   :256-259
        vnew, w = HipVec(V, c0 + nv), HipVec(V, c0 + nv + 1)
        vnew.scale_(1.0 / beta_old)
        it.operator.apply(vnew, w)
        a.value, b.value = lanczos_recurrence_unfused(V, c0, nv + 1, w, beta_old, it.orth, have_prev=nv >= 1)
    else:
        check(V._lib.kk_lanczos_expand(it.operator.handle, V.handle, c0, nv, it.orth.code, it.orth.eta, beta_old,
                                       C.byref(a), C.byref(b), C.byref(npass)))
    st.alphas.append(a.value)  # push!(alphas, real(alpha))   :261
    st.betas.append(b.value)
    V.length = nv + 1
    if not it.keepvecs and V.length > 1:  # popfirst!(state.V)   :264
        st.window += 1
        V.length -= 1
    st.k += 1
    st.last_passes = npass.value
    return st


def _lanczos_shrink(st: LanczosFactorization, k: int) -> LanczosFactorization:
    """shrink!(state::LanczosFactorization, k) (lanczos.jl:273-291)"""
    if len(st) != len(st.V):
        raise RuntimeError("we cannot shrink LanczosFactorization without keeping Lanczos vectors")
    if len(st) <= k:
        return st
    V = st.V
    V.length = k  # pop! down to k+1 vectors, then r = pop!(V): the residual is column k again
    del st.alphas[k:]
    del st.betas[k:]
    st.k = k
    HipVec(V, k).scale_(st.normres)  # state.r = scale!!(r, beta)   :289
    return st


# =========================================================================================
# Arnoldi -- src/factorizations/arnoldi.jl
# =========================================================================================
def packed_index(i: int, j: int) -> int:
    """0-based offset of H[i,j] (1-based, i <= j+1) in the packed Hessenberg vector
    (dense/packedhessenberg.jl:32-39)."""
    return ((j * j + j - 2) >> 1) + i - 1


@dataclass
class ArnoldiFactorization:  # arnoldi.jl:31-36
    k: int
    V: DeviceBasis
    H: List[float]  # packed Hessenberg

    def __len__(self):
        return self.k

    @property
    def normres(self) -> float:  # arnoldi.jl:50
        return abs(self.H[-1])

    @property
    def r(self) -> HipVec:
        return HipVec(self.V, len(self.V))

    def basis(self) -> DeviceBasis:
        return self.V

    def rayleighquotient(self) -> np.ndarray:
        """PackedHessenberg(H, k) (arnoldi.jl:48) as a dense k x k array."""
        k = self.k
        Hd = np.zeros((k, k))
        for j in range(1, k + 1):
            for i in range(1, min(j + 1, k) + 1):
                Hd[i - 1, j - 1] = self.H[packed_index(i, j)]
        return Hd


@dataclass
class ArnoldiIterator:  # arnoldi.jl:98-106
    operator: SparseOperator
    x0: object
    orth: Orthogonalizer = KrylovDefaults.orth
    capacity: int = KrylovDefaults.krylovdim + 2

    def __post_init__(self):
        _check_capacity(self.capacity)


def _arnoldi_initialize(it: ArnoldiIterator, V: Optional[DeviceBasis] = None) -> ArnoldiFactorization:
    """initialize(iter::ArnoldiIterator) (arnoldi.jl:135-175)"""
    op = it.operator
    if V is None:
        V = DeviceBasis(op.shape[0], it.capacity, op.ctx)
    _upload_x0(V, it.x0, 0)
    if isinstance(op, FunctionOperator):
        alpha, beta = _krylov_initialize_unfused(op, V, it.orth)
        V.length = 1
        return ArnoldiFactorization(1, V, [alpha, beta])
    a, b = C.c_double(), C.c_double()
    check(V._lib.kk_arnoldi_initialize(op.handle, V.handle, 0, it.orth.code, it.orth.eta, C.byref(a), C.byref(b)))
    V.length = 1
    return ArnoldiFactorization(1, V, [a.value, b.value])


def _arnoldi_initialize_(it: ArnoldiIterator, st: ArnoldiFactorization) -> ArnoldiFactorization:
    """initialize!(iter, state) (arnoldi.jl:176-198)"""
    V = st.V
    _upload_x0(V, it.x0, 0)
    v1, w = HipVec(V, 0), HipVec(V, 1)
    v1.scale_(1.0 / v1.norm())
    it.operator.apply(v1, w)
    alpha, beta = w.orthogonalize_against_(v1, it.orth)
    V.length = 1
    st.k = 1
    st.H[:] = [alpha, beta]
    return st


def _arnoldi_expand(it: ArnoldiIterator, st: ArnoldiFactorization) -> ArnoldiFactorization:
    """expand!(iter::ArnoldiIterator, state) (arnoldi.jl:199-219) -> kk_arnoldi_expand."""
    V = st.V
    k = len(V)
    if k + 2 > V.capacity:
        raise RuntimeError(f"Arnoldi slab of capacity {V.capacity} is full at k={k}")
    beta_old = st.normres
    h = np.zeros(k + 1)
    b, npass = C.c_double(), C.c_int()
    if isinstance(it.operator, FunctionOperator):   # un-fused arnoldirecurrence!! (arnoldi.jl:239-245)
        vnew, w = HipVec(V, k), HipVec(V, k + 1)
        vnew.scale_(1.0 / beta_old)                  # V[k+1] = scale(r, 1/beta)   :209
        it.operator.apply(vnew, w)
        V.length = k + 1
        hx, b.value, npass.value = V.orthogonalize(w, it.orth, 0, k + 1)
        h[:] = hx
    else:
        check(V._lib.kk_arnoldi_expand(it.operator.handle, V.handle, 0, k, it.orth.code, it.orth.eta, beta_old,
                                       h.ctypes.data_as(_lib.c_dp), C.byref(b), C.byref(npass)))
    st.k += 1
    V.length = k + 1
    st.H.extend(float(t) for t in h)  # H[m+1 : m+k]   :211-212
    st.H.append(b.value)              # H[m+k+1] = beta   :213
    st.last_passes = npass.value
    return st


def _arnoldi_shrink(st: ArnoldiFactorization, k: int) -> ArnoldiFactorization:
    """shrink!(state::ArnoldiFactorization, k) (arnoldi.jl:220-236)"""
    if len(st) <= k:
        return st
    V = st.V
    V.length = k
    del st.H[(k * k + 3 * k) >> 1:]
    st.k = k
    HipVec(V, k).scale_(st.normres)
    return st


# =========================================================================================
# GKL -- src/factorizations/gkl.jl
# =========================================================================================
@dataclass
class GKLFactorization:  # gkl.jl:31-38
    k: int
    U: DeviceBasis
    V: DeviceBasis
    alphas: List[float]
    betas: List[float]

    def __len__(self):
        return self.k

    @property
    def normres(self) -> float:
        return self.betas[self.k - 1]

    @property
    def r(self) -> HipVec:
        return HipVec(self.U, len(self.U))

    def basis(self, which: str) -> DeviceBasis:
        return self.U if which == "U" else self.V

    def rayleighquotient(self) -> np.ndarray:
        """Bidiagonal(alphas, betas[1:k-1], :L) (gkl.jl:77-79) as a dense array."""
        k = self.k
        B = np.diag(np.array(self.alphas[:k]))
        if k > 1:
            B += np.diag(np.array(self.betas[: k - 1]), -1)
        return B


@dataclass
class GKLIterator:  # gkl.jl:137-152
    operator: SparseOperator
    u0: object
    orth: Orthogonalizer = KrylovDefaults.orth
    capacity: int = KrylovDefaults.krylovdim + 2

    def __post_init__(self):
        _check_capacity(self.capacity)


def _gkl_initialize(it: GKLIterator) -> GKLFactorization:
    """initialize(iter::GKLIterator) (gkl.jl:183-215)"""
    op = it.operator
    U = DeviceBasis(op.shape[0], it.capacity, op.ctx)
    V = DeviceBasis(op.shape[1], it.capacity, op.ctx)
    _upload_x0(U, it.u0, 0)
    a, b = C.c_double(), C.c_double()
    check(U._lib.kk_gkl_initialize(op.handle, U.handle, V.handle, C.byref(a), C.byref(b)))
    U.length = 1
    V.length = 1
    return GKLFactorization(1, U, V, [a.value], [b.value])


def _gkl_expand(it: GKLIterator, st: GKLFactorization) -> GKLFactorization:
    """expand!(iter::GKLIterator, state) (gkl.jl:246-269) -> kk_gkl_expand."""
    k = len(st.U)
    if k + 2 > st.U.capacity or k + 1 > st.V.capacity:
        raise RuntimeError(f"GKL slabs are full at k={k}")
    a, b, pv, pu = C.c_double(), C.c_double(), C.c_int(), C.c_int()
    check(st.U._lib.kk_gkl_expand(it.operator.handle, st.U.handle, st.V.handle, k, it.orth.code, it.orth.eta,
                                  st.normres, C.byref(a), C.byref(b), C.byref(pv), C.byref(pu)))
    st.U.length = k + 1
    st.V.length = k + 1
    st.alphas.append(a.value)
    st.betas.append(b.value)
    st.k += 1
    st.last_passes = (pv.value, pu.value)
    return st


def _gkl_shrink(st: GKLFactorization, k: int) -> GKLFactorization:
    """shrink!(state::GKLFactorization, k) (gkl.jl:270-291)"""
    if len(st) != len(st.V):
        raise RuntimeError("we cannot shrink GKLFactorization without keeping vectors")
    if len(st) <= k:
        return st
    st.U.length = k  # pops U down to k+1, r = pop!(U): residual is column k of U again
    st.V.length = k  # V popped down to k
    del st.alphas[k:]
    del st.betas[k:]
    st.k = k
    HipVec(st.U, k).scale_(st.normres)
    return st


# =========================================================================================
# BlockLanczos -- src/factorizations/blocklanczos.jl
# =========================================================================================
@dataclass
class Block:
    """Block{T} (blocklanczos.jl:10-17): `size` consecutive columns of a slab."""
    basis: DeviceBasis
    col: int
    size: int

    def __len__(self):
        return self.size

    def __getitem__(self, i: int) -> HipVec:
        return HipVec(self.basis, self.col + i)

    def norm(self) -> float:
        """LinearAlgebra.norm(b::Block) (blocklanczos.jl:37): Frobenius norm."""
        return float(np.sqrt(sum(self[i].norm() ** 2 for i in range(self.size))))


def block_inner(B1: Block, B2: Block) -> np.ndarray:
    """block_inner (blocklanczos.jl:43-52) -> kk_block_inner."""
    M = np.zeros((len(B1), len(B2)), order="F")
    if M.size:
        check(B1.basis._lib.kk_block_inner(B1.basis.handle, B1.col, len(B1), B2.basis.handle, B2.col, len(B2),
                                           M.ctypes.data_as(_lib.c_dp), max(len(B1), 1)))
    return M


def block_qr_(block: Block, tol: float, out_col: Optional[int] = None):
    """block_qr! (blocklanczos.jl:312-353) -> (R[good_idx, :], good_idx, is_drift); the orthonormal
    vectors end up compacted at columns out_col.. (default: in place)."""
    p = len(block)
    out_col = block.col if out_col is None else out_col
    R = np.zeros((p, p), order="F")
    good = (C.c_int * p)()
    ng, drift = C.c_int(), C.c_int()
    check(block.basis._lib.kk_block_qr(block.basis.handle, block.col, p, out_col, tol, R.ctypes.data_as(_lib.c_dp), p,
                                       good, C.byref(ng), C.byref(drift)))
    return R[: ng.value, :].copy(), list(good[: ng.value]), bool(drift.value)


def block_reorthogonalize_(R: Block, V: DeviceBasis, m: Optional[int] = None):
    """block_reorthogonalize! (blocklanczos.jl:277-284)."""
    m = len(V) if m is None else m
    check(V._lib.kk_block_reorthogonalize(V.handle, 0, m, R.col, len(R)))
    return R


@dataclass
class BlockLanczosFactorization:  # blocklanczos.jl:89-96
    k: int
    V: DeviceBasis
    H: np.ndarray
    r_col: int     # first column of the residual block
    R_size: int
    norm_R: float
    area: tuple = (0, 0)  # the two residual areas of the slab (used alternately)

    def __len__(self):
        return self.k

    @property
    def normres(self) -> float:
        return self.norm_R

    def basis(self) -> DeviceBasis:
        return self.V

    def residual(self) -> Block:
        return Block(self.V, self.r_col, self.R_size)


@dataclass
class BlockLanczosIterator:  # blocklanczos.jl:133-157
    operator: SparseOperator
    x0: object            # list of numpy vectors (the start block)
    maxdim: int
    orth: Orthogonalizer = KrylovDefaults.orth
    qr_tol: float = KrylovDefaults.tol

    def __post_init__(self):
        if self.orth.name != "mgs2":
            raise ValueError("BlockLanczosIterator only supports ModifiedGramSchmidt2 orthogonalizer")  # :154-155

    def initialize(self, V: Optional[DeviceBasis] = None) -> BlockLanczosFactorization:
        """initialize(iter::BlockLanczosIterator) (blocklanczos.jl:159-198) -> kk_blocklanczos_initialize."""
        op = self.operator
        bs0 = len(self.x0)
        area_a, area_b = self.maxdim, self.maxdim + bs0
        if V is None:
            V = DeviceBasis(op.shape[0], self.maxdim + 2 * bs0, op.ctx)
        for j, x in enumerate(self.x0):
            _upload_x0(V, x, area_b + j)
        bs, nr = C.c_int(), C.c_double()
        M1 = np.zeros((bs0, bs0), order="F")
        check(V._lib.kk_blocklanczos_initialize(op.handle, V.handle, area_b, bs0, area_a, self.qr_tol, C.byref(bs),
                                                M1.ctypes.data_as(_lib.c_dp), bs0, C.byref(nr)))
        b = bs.value
        H = np.zeros((self.maxdim, self.maxdim))
        H[:b, :b] = M1[:b, :b]
        V.length = b
        return BlockLanczosFactorization(b, V, H, area_a, b, nr.value, (area_a, area_b))

    def expand(self, st: BlockLanczosFactorization) -> BlockLanczosFactorization:
        """expand!(iter::BlockLanczosIterator, state) (blocklanczos.jl:200-240) -> kk_blocklanczos_expand."""
        V, k, bs = st.V, st.k, st.R_size
        c_next = st.area[1] if st.r_col == st.area[0] else st.area[0]
        B = np.zeros((bs, bs), order="F")
        M = np.zeros((bs, bs), order="F")
        bsn, nr, drift = C.c_int(), C.c_double(), C.c_int()
        check(V._lib.kk_blocklanczos_expand(self.operator.handle, V.handle, k, bs, st.r_col, c_next, self.qr_tol,
                                            C.byref(bsn), B.ctypes.data_as(_lib.c_dp), bs, M.ctypes.data_as(_lib.c_dp), bs,
                                            C.byref(nr), C.byref(drift)))
        n = bsn.value
        st.H[k:k + n, k - bs:k] = B[:n, :bs]                 # :220
        st.H[k - bs:k, k:k + n] = B[:n, :bs].T               # :221
        st.H[k:k + n, k:k + n] = M[:n, :n]                   # :229
        st.r_col, st.R_size, st.norm_R = c_next, n, nr.value
        st.k += n
        V.length = st.k
        st.last_drift = bool(drift.value)
        return st
