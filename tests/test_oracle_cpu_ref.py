"""The C timing twin (oracle/cpu_ref.c, used for bench.py's cpu_baseline) against the NumPy oracle."""
import ctypes as C
from pathlib import Path

import numpy as np
import pytest

LIB = Path(__file__).resolve().parent.parent / "oracle" / "libcpu_ref.so"


def load():
    if not LIB.exists():
        pytest.skip("oracle/libcpu_ref.so not built (run __graft_entry__.build())")
    lib = C.CDLL(str(LIB))
    dp, ip = C.POINTER(C.c_double), C.POINTER(C.c_int64)
    lib.kkref_lanczos.argtypes = [C.c_int64, ip, ip, dp, dp, C.c_int, C.c_int, C.c_double, C.c_int, dp, dp,
                                  C.POINTER(C.c_int), dp]
    lib.kkref_lanczos.restype = C.c_int
    return lib


@pytest.mark.parametrize("orth", [0, 1, 2, 3, 4, 5])
def test_cpu_ref_matches_numpy_oracle(ko, orth):
    lib = load()
    nx, ny, steps = 31, 17, 20
    n = nx * ny
    A = ko.laplacian_2d(nx, ny, shift_diag=10 * np.linspace(0, 1, n) ** 2).tocsc()
    A.sort_indices()
    colptr = np.ascontiguousarray(A.indptr, dtype=np.int64) + 1
    rowval = np.ascontiguousarray(A.indices, dtype=np.int64) + 1
    nz = np.ascontiguousarray(A.data)
    x0 = np.random.default_rng(3).random(n)
    al, be = np.zeros(steps + 1), np.zeros(steps + 1)
    passes = C.c_int()
    basis = np.zeros((steps + 2) * n)
    dp, ip = C.POINTER(C.c_double), C.POINTER(C.c_int64)
    rc = lib.kkref_lanczos(n, colptr.ctypes.data_as(ip), rowval.ctypes.data_as(ip), nz.ctypes.data_as(dp),
                           x0.ctypes.data_as(dp), steps, orth, 0.75, 2, al.ctypes.data_as(dp), be.ctypes.data_as(dp),
                           C.byref(passes), basis.ctypes.data_as(dp))
    assert rc == 0
    alg = [ko.CGS, ko.MGS, ko.CGS2, ko.MGS2, ko.CGSIR(0.75), ko.MGSIR(0.75)][orth]
    it = ko.LanczosIterator(A, x0.copy(), alg)
    f = ko.lanczos_initialize(it)
    stats = {}
    for _ in range(steps):
        f = ko.lanczos_expand(it, f, stats)
    tol = 1e-10 if orth >= 2 else 1e-6
    np.testing.assert_allclose(al, f.alphas, rtol=tol)
    np.testing.assert_allclose(be, f.betas, rtol=tol)
    assert passes.value == stats.get("passes", 0)
    V = basis.reshape(steps + 2, n).T
    if orth >= 2:
        assert np.max(np.abs(V[:, :-1].T @ V[:, :-1] - np.eye(steps + 1))) < 1e-12


from cpu_ref_lib import load as load_gmres, run_gmres as run_kkref_gmres  # noqa: E402


@pytest.mark.parametrize("orth", [0, 1, 2, 3, 4, 5])
@pytest.mark.parametrize("case", ["restarts", "converges", "shifted"])
def test_cpu_ref_gmres_matches_numpy_oracle(ko, orth, case):
    """kkref_gmres (restated from src/linsolve/gmres.jl:1-151) against the NumPy oracle: identical
    numiter / numops / converged, residual-estimate trace and solution to tolerance."""
    lib = load_gmres()
    nx, ny = 23, 19
    n = nx * ny
    A = ko.convection_diffusion_2d(nx, ny)
    rng = np.random.default_rng(4)
    b = rng.random(n)
    alg = [ko.CGS, ko.MGS, ko.CGS2, ko.MGS2, ko.CGSIR(0.75), ko.MGSIR(0.75)][orth]
    a0, a1, x0 = 0.0, 1.0, None
    if case == "restarts":
        kd, mi, tol = 12, 4, 1e-12 * np.linalg.norm(b)
    elif case == "converges":
        kd, mi, tol = 40, 30, 1e-9 * np.linalg.norm(b)
    else:
        kd, mi, tol = 15, 6, 1e-10 * np.linalg.norm(b)
        a0, a1, x0 = 0.7, -0.4, rng.random(n)
    tr = []
    xo, io = ko.gmres(A, b, x0, a0, a1, krylovdim=kd, maxiter=mi, tol=tol, orth=alg, trace=tr)
    xc, ic, tc = run_kkref_gmres(lib, A, b, x0, a0, a1, kd, mi, tol, orth)
    assert (ic["converged"], ic["numiter"], ic["numops"]) == (io.converged, io.numiter, io.numops)
    rt = 1e-8 if orth >= 2 else 1e-5
    np.testing.assert_allclose(tc, [t[2] for t in tr], rtol=rt)
    np.testing.assert_allclose(ic["normres"], io.normres, rtol=rt, atol=1e-3 * tol)
    np.testing.assert_allclose(xc, xo, rtol=0, atol=1e-9 * np.linalg.norm(xo))
