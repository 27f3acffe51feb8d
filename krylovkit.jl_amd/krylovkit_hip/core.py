"""Device objects: Context, DeviceBasis (the HBM slab replacing OrthonormalBasis{T}),
HipVec (one column of a slab), SparseOperator, and the Orthogonalizer hierarchy.

Mirrors the reference's L1/L2 surface: `src/orthonormal.jl`, `src/apply.jl`,
`src/algorithms.jl:17-80`.  Every numerical operation is a call into libkrylov_hip.so.
"""
from __future__ import annotations

import ctypes as C
import math
from dataclasses import dataclass
from typing import Optional, Sequence

import numpy as np

from . import _lib
from ._lib import check


# ------------------------------------------------------------------ orthogonalizers
@dataclass(frozen=True)
class Orthogonalizer:
    """src/algorithms.jl:17-80"""
    name: str
    eta: float = 1.0 / math.sqrt(2.0)

    @property
    def code(self) -> int:
        return _lib.ORTH_CODES[self.name]

    @property
    def is_reorth(self) -> bool:  # Reorthogonalizer subtypes, algorithms.jl:18
        return self.name not in ("cgs", "mgs")


def ClassicalGramSchmidt():
    return Orthogonalizer("cgs")


def ModifiedGramSchmidt():
    return Orthogonalizer("mgs")


def ClassicalGramSchmidt2():
    return Orthogonalizer("cgs2")


def ModifiedGramSchmidt2():
    return Orthogonalizer("mgs2")


def ClassicalGramSchmidtIR(eta: float = 1.0 / math.sqrt(2.0)):
    return Orthogonalizer("cgsir", eta)


def ModifiedGramSchmidtIR(eta: float = 1.0 / math.sqrt(2.0)):
    return Orthogonalizer("mgsir", eta)


class KrylovDefaults:  # src/algorithms.jl:556-564
    orth = ModifiedGramSchmidt2()
    krylovdim = 30
    maxiter = 100
    tol = 1e-12


# ------------------------------------------------------------------ context
class Context:
    """One GPU, one HIP stream (kk_ctx)."""

    def __init__(self, device: int = 0):
        self._lib = _lib.load()
        h = C.c_void_p()
        check(self._lib.kk_ctx_create(device, C.byref(h)))
        self.handle = h
        self.device = device

    def close(self):
        if getattr(self, "handle", None):
            self._lib.kk_ctx_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def sync(self):
        check(self._lib.kk_ctx_sync(self.handle))

    def set_option(self, key: str, value: float):
        check(self._lib.kk_ctx_set_option(self.handle, key.encode(), float(value)))

    def get_option(self, key: str) -> float:
        v = C.c_double()
        check(self._lib.kk_ctx_get_option(self.handle, key.encode(), C.byref(v)))
        return v.value

    def set_stream(self, stream_ptr: Optional[int]):
        check(self._lib.kk_ctx_set_stream(self.handle, C.c_void_p(stream_ptr or 0)))

    def timer_start(self):
        check(self._lib.kk_ctx_timer_start(self.handle))

    def timer_stop(self) -> float:
        ms = C.c_double()
        check(self._lib.kk_ctx_timer_stop(self.handle, C.byref(ms)))
        return ms.value

    def prof_enable(self, on=True):
        """0/False off, 1/True every kernel class, 2 only the basis-streaming kernels (low overhead)."""
        check(self._lib.kk_ctx_prof_enable(self.handle, int(on)))

    def prof_reset(self):
        check(self._lib.kk_ctx_prof_reset(self.handle))

    def prof_get(self, kernel_class: str):
        ms, n = C.c_double(), C.c_int64()
        check(self._lib.kk_ctx_prof_get(self.handle, kernel_class.encode(), C.byref(ms), C.byref(n)))
        return ms.value, n.value


_default_ctx: Optional[Context] = None


def default_context() -> Context:
    global _default_ctx
    if _default_ctx is None or _default_ctx.handle is None:
        _default_ctx = Context(0)
    return _default_ctx


def device_count() -> int:
    n = C.c_int()
    check(_lib.load().kk_device_count(C.byref(n)))
    return n.value


def _dp(a: np.ndarray):
    return a.ctypes.data_as(_lib.c_dp)


# ------------------------------------------------------------------ basis slab
class DeviceBasis:
    """Contiguous HBM slab of `capacity` column vectors of length n (kk_basis).

    Replaces `OrthonormalBasis{T}` (src/orthonormal.jl:26-54): `length` is host-side
    bookkeeping (`push!`/`pop!`/`resize!` are O(1)); vector j is column j."""

    def __init__(self, n: int, capacity: int, ctx: Optional[Context] = None):
        self.ctx = ctx or default_context()
        self._lib = self.ctx._lib
        h = C.c_void_p()
        check(self._lib.kk_basis_create(self.ctx.handle, int(n), int(capacity), C.byref(h)))
        self.handle = h
        self.n = int(n)
        self.capacity = int(capacity)
        self.length = 0  # number of basis vectors currently held (columns 0..length-1)

    def free(self):
        if getattr(self, "handle", None):
            self._lib.kk_basis_free(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass

    def __len__(self):
        return self.length

    def info(self):
        n, ld, cap, p = C.c_int64(), C.c_int64(), C.c_int(), C.c_void_p()
        check(self._lib.kk_basis_info(self.handle, C.byref(n), C.byref(ld), C.byref(cap), C.byref(p)))
        return n.value, ld.value, cap.value, p.value

    def __getitem__(self, col: int) -> "HipVec":
        if col < 0:
            col += self.length
        return HipVec(self, col)

    # --- host <-> device
    def upload(self, col: int, x: np.ndarray):
        x = np.ascontiguousarray(x, dtype=np.float64)
        if x.shape != (self.n,):
            raise _lib.DimensionMismatch(_lib.KK_ERR_DIM, f"upload: expected shape ({self.n},), got {x.shape}")
        check(self._lib.kk_basis_upload(self.handle, col, _dp(x)))

    def download(self, col: int) -> np.ndarray:
        out = np.empty(self.n)
        check(self._lib.kk_basis_download(self.handle, col, _dp(out)))
        return out

    def to_numpy(self, ncols: Optional[int] = None) -> np.ndarray:
        ncols = self.length if ncols is None else ncols
        return np.stack([self.download(j) for j in range(ncols)], axis=1) if ncols else np.zeros((self.n, 0))

    def invalidate_gram(self):
        check(self._lib.kk_basis_invalidate_gram(self.handle))

    # --- L2 operations on columns [c0, c0+m)   (src/orthonormal.jl)
    def project(self, x: "HipVec", c0: int = 0, m: Optional[int] = None, alpha: float = 1.0, beta: float = 0.0,
                y: Optional[np.ndarray] = None) -> np.ndarray:
        """project!! (orthonormal.jl:88-118)"""
        m = self.length - c0 if m is None else m
        if y is None:
            y = np.zeros(m)
        if y.shape != (m,):
            raise _lib.DimensionMismatch(_lib.KK_ERR_DIM, "project: length(y) != length(r)")
        for j0 in range(0, m, _lib.KK_MAX_M):     # the library takes at most KK_MAX_M basis vectors per call
            mm = min(_lib.KK_MAX_M, m - j0)
            yj = np.ascontiguousarray(y[j0:j0 + mm])
            check(self._lib.kk_project(self.handle, c0 + j0, mm, x.basis.handle, x.col, alpha, beta, _dp(yj)))
            y[j0:j0 + mm] = yj
        return y

    def unproject(self, y: "HipVec", x: Sequence[float], c0: int = 0, m: Optional[int] = None, alpha: float = 1.0,
                  beta: float = 0.0) -> "HipVec":
        """unproject!! (orthonormal.jl:132-196)"""
        m = self.length - c0 if m is None else m
        xa = np.ascontiguousarray(x, dtype=np.float64)
        if xa.shape != (m,):
            raise _lib.DimensionMismatch(_lib.KK_ERR_DIM, "unproject: length(x) != length(r)")
        if m == 0:
            check(self._lib.kk_unproject(y.basis.handle, y.col, self.handle, c0, 0, _dp(xa), alpha, beta))
        for j0 in range(0, m, _lib.KK_MAX_M):     # panels of KK_MAX_M vectors; beta applies to the first one only
            mm = min(_lib.KK_MAX_M, m - j0)
            xj = np.ascontiguousarray(xa[j0:j0 + mm])
            check(self._lib.kk_unproject(y.basis.handle, y.col, self.handle, c0 + j0, mm, _dp(xj), alpha, beta if j0 == 0 else 1.0))
        return y

    def rank1update(self, y: "HipVec", x: Sequence[float], c0: int = 0, m: Optional[int] = None, alpha: float = 1.0,
                    beta: float = 1.0):
        """rank1update! (orthonormal.jl:210-275)"""
        m = self.length - c0 if m is None else m
        xa = np.ascontiguousarray(x, dtype=np.float64)
        if xa.shape != (m,):
            raise _lib.DimensionMismatch(_lib.KK_ERR_DIM, "rank1update: length(x) != length(r)")
        check(self._lib.kk_rank1update(self.handle, c0, m, y.basis.handle, y.col, _dp(xa), alpha, beta))
        return self

    def basistransform(self, U: np.ndarray, c0: int = 0):
        """basistransform! (orthonormal.jl:291-354): b[j] <- sum_i b[i] U[i,j]."""
        Uf = np.asfortranarray(U, dtype=np.float64)
        m, n = Uf.shape
        if m != self.length - c0:
            raise _lib.DimensionMismatch(_lib.KK_ERR_DIM, f"basistransform: size(U,1)={m} != length(b)={self.length - c0}")
        check(self._lib.kk_basistransform(self.handle, c0, m, n, _dp(Uf), m))
        return self

    def rmul_givens(self, i1: int, i2: int, c: float, s: float):
        """rmul!(b, G::Givens) (dense/givens.jl:12-36)"""
        check(self._lib.kk_givens_rmul(self.handle, i1, i2, c, s))
        return self

    def rmul_householder(self, beta: float, v: Sequence[float], c0: int, m: int):
        """rmul!(b, H::Householder) over the contiguous range c0:c0+m (dense/reflector.jl:143-154)"""
        va = np.ascontiguousarray(v, dtype=np.float64)
        check(self._lib.kk_householder_rmul(self.handle, c0, m, _dp(va), beta))
        return self

    def times(self, x: Sequence[float], out: "HipVec", c0: int = 0, m: Optional[int] = None) -> "HipVec":
        """Base.:*(b, x) (orthonormal.jl:57-60) into a caller-provided vector."""
        return self.unproject(out, x, c0, m, 1.0, 0.0)

    def orthogonalize(self, w: "HipVec", alg: Orthogonalizer, c0: int = 0, m: Optional[int] = None,
                      want_norm: bool = True):
        """orthogonalize!!(w, b, x, alg) (orthonormal.jl:378-452) -> (x, |w|, passes)"""
        m = self.length - c0 if m is None else m
        x = np.zeros(max(m, 1))
        nrm, npass = C.c_double(), C.c_int()
        check(self._lib.kk_orthogonalize(self.handle, c0, m, w.basis.handle, w.col, alg.code, alg.eta, _dp(x),
                                         C.byref(nrm) if want_norm else None, C.byref(npass)))
        return x[:m], nrm.value, npass.value

    def orthonormalize(self, w: "HipVec", alg: Orthogonalizer, c0: int = 0, m: Optional[int] = None):
        """orthonormalize!! (orthonormal.jl:522-527) -> (x, beta, passes)"""
        m = self.length - c0 if m is None else m
        x = np.zeros(max(m, 1))
        nrm, npass = C.c_double(), C.c_int()
        check(self._lib.kk_orthonormalize(self.handle, c0, m, w.basis.handle, w.col, alg.code, alg.eta, _dp(x),
                                          C.byref(nrm), C.byref(npass)))
        return x[:m], nrm.value, npass.value


@dataclass
class HipVec:
    """One column of a DeviceBasis: the device vector type T behind the L1 verbs
    (VectorInterface.jl semantics).  `!!` verbs mutate and return self."""
    basis: DeviceBasis
    col: int

    @property
    def n(self):
        return self.basis.n

    def _l(self):
        return self.basis._lib

    def inner(self, other: "HipVec") -> float:
        out = C.c_double()
        check(self._l().kk_vec_dot(self.basis.handle, self.col, other.basis.handle, other.col, C.byref(out)))
        return out.value

    def norm(self) -> float:
        out = C.c_double()
        check(self._l().kk_vec_nrm2(self.basis.handle, self.col, C.byref(out)))
        return out.value

    def add_(self, x: "HipVec", a: float = 1.0, b: float = 1.0) -> "HipVec":
        """add!!(y, x, a, b): y = b*y + a*x"""
        check(self._l().kk_vec_axpby(self.basis.handle, self.col, x.basis.handle, x.col, a, b))
        return self

    def scale_(self, a: float) -> "HipVec":
        check(self._l().kk_vec_scal(self.basis.handle, self.col, a))
        return self

    def scale_from_(self, x: "HipVec", a: float) -> "HipVec":
        """scale!!(y, x, a): y = a*x"""
        check(self._l().kk_vec_copy_scal(self.basis.handle, self.col, x.basis.handle, x.col, a))
        return self

    def zero_(self) -> "HipVec":
        check(self._l().kk_vec_zero(self.basis.handle, self.col))
        return self

    def rand_(self, seed: int) -> "HipVec":
        check(self._l().kk_vec_fill_random(self.basis.handle, self.col, C.c_uint64(seed)))
        return self

    def set(self, x: np.ndarray) -> "HipVec":
        self.basis.upload(self.col, x)
        return self

    def get(self) -> np.ndarray:
        return self.basis.download(self.col)

    def orthogonalize_against_(self, q: "HipVec", alg: Orthogonalizer, want_norm: bool = True):
        """orthogonalize!!(v, q, alg) (orthonormal.jl:455-489) -> (s, |v|)"""
        s, nrm = C.c_double(), C.c_double()
        check(self._l().kk_orthogonalize_vec(q.basis.handle, q.col, self.basis.handle, self.col, alg.code, alg.eta,
                                             C.byref(s), C.byref(nrm) if want_norm else None))
        return s.value, nrm.value


# ------------------------------------------------------------------ operators
class SparseOperator:
    """Device sparse linear map (kk_op): `apply` / `apply_normal` / `apply_adjoint`
    of src/apply.jl:1-19 for a sparse matrix.  Accepts any scipy.sparse matrix; CSC input is
    handed over as Julia would hand a SparseMatrixCSC (kk_csc_create)."""

    def __init__(self, A, ctx: Optional[Context] = None, symmetric: Optional[bool] = None, via_csc: bool = False):
        import scipy.sparse as sp

        self.ctx = ctx or default_context()
        self._lib = self.ctx._lib
        if not sp.issparse(A):
            raise TypeError("SparseOperator needs a scipy.sparse matrix")
        self.shape = A.shape
        self.symmetric = bool(symmetric)
        flags = _lib.KK_OP_SYMMETRIC if symmetric else 0
        h = C.c_void_p()
        if via_csc or A.format == "csc":
            Ac = A.tocsc()
            Ac.sort_indices()
            colptr = np.ascontiguousarray(Ac.indptr, dtype=np.int64) + 1  # Julia: 1-based Int64
            rowval = np.ascontiguousarray(Ac.indices, dtype=np.int64) + 1
            nz = np.ascontiguousarray(Ac.data, dtype=np.float64)
            check(self._lib.kk_csc_create(self.ctx.handle, A.shape[0], A.shape[1], Ac.nnz,
                                          colptr.ctypes.data_as(_lib.c_i64p), rowval.ctypes.data_as(_lib.c_i64p),
                                          _dp(nz), 1, flags, C.byref(h)))
        else:
            Ar = A.tocsr()
            rowptr = np.ascontiguousarray(Ar.indptr, dtype=np.int64)
            col = np.ascontiguousarray(Ar.indices, dtype=np.int32)
            val = np.ascontiguousarray(Ar.data, dtype=np.float64)
            check(self._lib.kk_csr_create(self.ctx.handle, A.shape[0], A.shape[1], Ar.nnz,
                                          rowptr.ctypes.data_as(_lib.c_i64p), col.ctypes.data_as(_lib.c_i32p),
                                          _dp(val), 0, flags, C.byref(h)))
        self.handle = h
        self.nnz = int(A.nnz)

    @classmethod
    def from_csr_arrays(cls, shape, rowptr, col, val, ctx: Optional[Context] = None, symmetric: bool = False):
        """kk_csr_create on caller-built arrays (0-based int64 row pointers, int32 columns, f64 values) -- no scipy object in
        between; what a host with more than 2^31 stored entries uses"""
        self = cls.__new__(cls)
        self.ctx = ctx or default_context()
        self._lib = self.ctx._lib
        self.shape = tuple(shape)
        self.symmetric = bool(symmetric)
        rowptr = np.ascontiguousarray(rowptr, dtype=np.int64)
        col = np.ascontiguousarray(col, dtype=np.int32)
        val = np.ascontiguousarray(val, dtype=np.float64)
        h = C.c_void_p()
        check(self._lib.kk_csr_create(self.ctx.handle, shape[0], shape[1], int(rowptr[-1]), rowptr.ctypes.data_as(_lib.c_i64p),
                                      col.ctypes.data_as(_lib.c_i32p), _dp(val), 0, _lib.KK_OP_SYMMETRIC if symmetric else 0, C.byref(h)))
        self.handle = h
        self.nnz = int(rowptr[-1])
        return self

    def free(self):
        if getattr(self, "handle", None):
            self._lib.kk_op_free(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass

    def info(self):
        nr, nc, nnz, fmt, nb = C.c_int64(), C.c_int64(), C.c_int64(), C.c_int(), C.c_int64()
        check(self._lib.kk_op_info(self.handle, C.byref(nr), C.byref(nc), C.byref(nnz), C.byref(fmt), C.byref(nb)))
        return dict(nrows=nr.value, ncols=nc.value, nnz=nnz.value, format=("ELL", "CSR", "SELL", "SELL-tiled", "ELL+DIA", "ELL+DIA const")[fmt.value] if fmt.value in (0, 1, 2, 3, 4, 5) else "?",
                    device_bytes=nb.value)

    def apply(self, x: HipVec, y: HipVec, transpose: bool = False) -> HipVec:
        check(self._lib.kk_spmv(self.handle, int(transpose), x.basis.handle, x.col, y.basis.handle, y.col))
        return y

    def apply_adjoint(self, x: HipVec, y: HipVec) -> HipVec:
        return self.apply(x, y, True)

    def apply_affine(self, x: HipVec, y: HipVec, a0: float, a1: float) -> HipVec:
        """apply(op, x, a0, a1) = a0*x + a1*A*x (apply.jl:4-11)"""
        check(self._lib.kk_spmv_affine(self.handle, x.basis.handle, x.col, y.basis.handle, y.col, a0, a1))
        return y


class FunctionOperator:
    """apply(f, x) = f(x) (src/apply.jl:2): a linear map given as a callable instead of a matrix.  `f(x, y)` receives two
    HipVec and must leave y = A x using device-side work only (compositions / polynomials of SparseOperators, a
    shift-and-invert through an inner solver, ...).  The fused kk_*_expand entry points need a sparse kk_op, so with a
    function the iterators issue the reference's un-fused sequence instead: one call of f per `expand!`, then the
    recurrence through the L1 verbs and the fused orthogonalisation passes (kk_orthogonalize, kk_orthogonalize_vec).
    `adjoint` is the callable for apply_adjoint (apply.jl:15), `symmetric` the promise needed by Lanczos."""

    def __init__(self, f, n: int, ctx: Optional[Context] = None, symmetric: bool = False, adjoint=None, n_out: Optional[int] = None):
        self.f, self.adjoint = f, adjoint
        self.shape = (n if n_out is None else n_out, n)
        self.ctx = ctx or default_context()
        self.symmetric = bool(symmetric)
        self.handle = None      # no kk_op behind a function

    def apply(self, x: HipVec, y: HipVec, transpose: bool = False) -> HipVec:
        if transpose:
            if self.adjoint is None:
                if not self.symmetric:
                    raise ValueError("FunctionOperator: no adjoint callable was given")
                self.f(x, y)
            else:
                self.adjoint(x, y)
        else:
            self.f(x, y)
        return y

    def apply_adjoint(self, x: HipVec, y: HipVec) -> HipVec:
        return self.apply(x, y, True)

    def apply_affine(self, x: HipVec, y: HipVec, a0: float, a1: float) -> HipVec:
        """apply(f, x, a0, a1) = a0 x + a1 f(x) (apply.jl:4-11)"""
        self.f(x, y)
        return y.add_(x, a0, a1)

