"""ctypes access to oracle/libcpu_ref.so (TEST INFRASTRUCTURE): the C restatement of the reference's CPU path used as the
checker at BASELINE.json's full sizes (10M-row Lanczos sweep, 2M-row GMRES cycle) where the NumPy oracle is too slow."""
import ctypes as C
import os
from pathlib import Path

import numpy as np

LIB = Path(__file__).resolve().parent.parent / "oracle" / "libcpu_ref.so"
_dp, _ip = C.POINTER(C.c_double), C.POINTER(C.c_int64)


def load():
    if not LIB.exists():
        import pytest
        pytest.skip("oracle/libcpu_ref.so not built (run __graft_entry__.build())")
    os.environ.setdefault("OMP_WAIT_POLICY", "passive")
    lib = C.CDLL(str(LIB))
    lib.kkref_lanczos.argtypes = [C.c_int64, _ip, _ip, _dp, _dp, C.c_int, C.c_int, C.c_double, C.c_int, _dp, _dp,
                                  C.POINTER(C.c_int), _dp]
    lib.kkref_lanczos.restype = C.c_int
    lib.kkref_gmres.argtypes = [C.c_int64, _ip, _ip, _dp, _dp, _dp, C.c_double, C.c_double, C.c_int, C.c_int, C.c_double,
                                C.c_int, C.c_double, C.c_int, _dp, C.POINTER(C.c_int), _dp, _dp, C.c_int, C.POINTER(C.c_int)]
    lib.kkref_gmres.restype = C.c_int
    lib.kkref_gkl.argtypes = [C.c_int64, C.c_int64, _ip, _ip, _dp, _ip, _ip, _dp, _dp, C.c_int, C.c_int, C.c_double, C.c_int,
                              _dp, _dp, _dp, _dp]
    lib.kkref_gkl.restype = C.c_int
    lib.kkref_blocklanczos.argtypes = [C.c_int64, _ip, _ip, _dp, _dp, C.c_int, C.c_int, C.c_int, C.c_double, C.c_int, _dp, C.c_int,
                                       C.POINTER(C.c_int), C.POINTER(C.c_int), _dp, _dp]
    lib.kkref_blocklanczos.restype = C.c_int
    return lib


def julia_csc(A):
    """scipy sparse -> the arrays of a Julia SparseMatrixCSC{Float64,Int64} (1-based)"""
    A = A.tocsc()
    A.sort_indices()
    return (np.ascontiguousarray(A.indptr, dtype=np.int64) + 1, np.ascontiguousarray(A.indices, dtype=np.int64) + 1,
            np.ascontiguousarray(A.data, dtype=np.float64))


def usable_threads() -> int:
    try:
        n = len(os.sched_getaffinity(0))
    except Exception:
        n = os.cpu_count() or 1
    try:
        txt = Path("/sys/fs/cgroup/cpu.max").read_text().split()
        if txt[0] != "max":
            n = min(n, max(1, int(float(txt[0]) / float(txt[1]) + 0.5)))
    except Exception:
        pass
    return max(1, min(n, 32))


def run_lanczos(lib, A, x0, steps, orth, eta=0.75, nthreads=2, want_basis=False):
    colptr, rowval, nz = julia_csc(A)
    n = A.shape[0]
    x0 = np.ascontiguousarray(x0, dtype=np.float64)
    al, be = np.zeros(steps + 1), np.zeros(steps + 1)
    passes = C.c_int()
    basis = np.zeros((steps + 2) * n) if want_basis else None
    rc = lib.kkref_lanczos(n, colptr.ctypes.data_as(_ip), rowval.ctypes.data_as(_ip), nz.ctypes.data_as(_dp),
                           x0.ctypes.data_as(_dp), steps, orth, eta, nthreads, al.ctypes.data_as(_dp), be.ctypes.data_as(_dp),
                           C.byref(passes), basis.ctypes.data_as(_dp) if want_basis else None)
    assert rc == 0
    return al, be, passes.value, (basis.reshape(steps + 2, n).T if want_basis else None)


def run_gmres(lib, A, b, x0, a0, a1, krylovdim, maxiter, tol, orth, eta=0.75, nthreads=2):
    """linsolve(A, b, x0, GMRES(...), a0, a1) by oracle/cpu_ref.c::kkref_gmres -> (x, info dict, residual-estimate trace)"""
    colptr, rowval, nz = julia_csc(A)
    n = A.shape[0]
    b = np.ascontiguousarray(b, dtype=np.float64)
    x = np.zeros(n)
    info = (C.c_int * 3)()
    normres = C.c_double()
    cap = maxiter * krylovdim + 8
    trace = np.zeros(cap)
    tl = C.c_int()
    x0p = None if x0 is None else np.ascontiguousarray(x0, dtype=np.float64).ctypes.data_as(_dp)
    rc = lib.kkref_gmres(n, colptr.ctypes.data_as(_ip), rowval.ctypes.data_as(_ip), nz.ctypes.data_as(_dp), b.ctypes.data_as(_dp),
                         x0p, a0, a1, krylovdim, maxiter, tol, orth, eta, nthreads, x.ctypes.data_as(_dp), info,
                         C.byref(normres), trace.ctypes.data_as(_dp), cap, C.byref(tl))
    assert rc == 0
    return x, dict(converged=info[0], numiter=info[1], numops=info[2], normres=normres.value), trace[:tl.value]


def julia_csr(A):
    """scipy sparse -> 1-based Int64 CSR image with ascending column indices (what a row-wise evaluation of Julia's
    column-scatter product needs to give the same bits)"""
    A = A.tocsr()
    A.sort_indices()
    return (np.ascontiguousarray(A.indptr, dtype=np.int64) + 1, np.ascontiguousarray(A.indices, dtype=np.int64) + 1,
            np.ascontiguousarray(A.data, dtype=np.float64))


def run_gkl(lib, A, u0, steps, orth, eta=0.75, nthreads=2, want_bases=False):
    """initialize + `steps` expand! of the GKL factorization by oracle/cpu_ref.c::kkref_gkl -> (alphas, betas, U, V)"""
    colptr, rowval, nz = julia_csc(A)
    rowptr, colval, rv = julia_csr(A)
    nr, nc = A.shape
    u0 = np.ascontiguousarray(u0, dtype=np.float64)
    al, be = np.zeros(steps + 1), np.zeros(steps + 1)
    U = np.zeros((steps + 2) * nr) if want_bases else None
    V = np.zeros((steps + 1) * nc) if want_bases else None
    rc = lib.kkref_gkl(nr, nc, colptr.ctypes.data_as(_ip), rowval.ctypes.data_as(_ip), nz.ctypes.data_as(_dp),
                       rowptr.ctypes.data_as(_ip), colval.ctypes.data_as(_ip), rv.ctypes.data_as(_dp), u0.ctypes.data_as(_dp),
                       steps, orth, eta, nthreads, al.ctypes.data_as(_dp), be.ctypes.data_as(_dp),
                       U.ctypes.data_as(_dp) if want_bases else None, V.ctypes.data_as(_dp) if want_bases else None)
    assert rc == 0, rc
    return al, be, (U.reshape(steps + 2, nr).T if want_bases else None), (V.reshape(steps + 1, nc).T if want_bases else None)


def run_blocklanczos(lib, A, X0, target_dim, max_steps, qr_tol=1e-12, nthreads=2, want_basis=False):
    """initialize + expand! of the BlockLanczos factorization by oracle/cpu_ref.c::kkref_blocklanczos on a symmetric A.
    X0: n x bs0 (columns = start block) -> dict(H, k, R_size, steps, numops, sizes, norm_R, V)"""
    colptr, rowval, nz = julia_csc(A)
    n = A.shape[0]
    X0 = np.asfortranarray(X0, dtype=np.float64)
    bs0 = X0.shape[1]
    ldh = target_dim + 2 * bs0
    H = np.zeros((ldh, ldh), order="F")
    info = (C.c_int * 4)()
    sizes = (C.c_int * (max_steps + 2))()
    nR = C.c_double()
    Vb = np.zeros((n, ldh), order="F") if want_basis else None
    rc = lib.kkref_blocklanczos(n, colptr.ctypes.data_as(_ip), rowval.ctypes.data_as(_ip), nz.ctypes.data_as(_dp),
                                X0.ctypes.data_as(_dp), bs0, target_dim, max_steps, qr_tol, nthreads, H.ctypes.data_as(_dp), ldh,
                                info, sizes, C.byref(nR), Vb.ctypes.data_as(_dp) if want_basis else None)
    assert rc == 0, rc
    k = info[0]
    return dict(H=H[:k, :k].copy(), k=k, R_size=info[1], steps=info[2], numops=info[3], sizes=list(sizes[:info[2] + 1]),
                norm_R=nR.value, V=(Vb[:, :k + info[1]] if want_basis else None))
