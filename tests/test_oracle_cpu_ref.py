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


from cpu_ref_lib import run_gkl as run_kkref_gkl, run_blocklanczos as run_kkref_block  # noqa: E402


@pytest.mark.parametrize("orth", [0, 1, 2, 3, 4, 5])
def test_cpu_ref_gkl_matches_numpy_oracle(ko, orth):
    """kkref_gkl (src/factorizations/gkl.jl:183-215, 246-269, 294-404) against the NumPy oracle: alpha / beta trajectories,
    orthonormal bases, the two GKL relations (test/factorize.jl:387-401)."""
    lib = load_gmres()
    A = ko.sparse_random(310, 120, 7, 5)
    u0 = np.random.default_rng(6).random(310)
    steps = 18
    alg = [ko.CGS, ko.MGS, ko.CGS2, ko.MGS2, ko.CGSIR(0.75), ko.MGSIR(0.75)][orth]
    al, be, U, V = run_kkref_gkl(lib, A, u0, steps, orth, want_bases=True)
    it = ko.GKLIterator(A, u0.copy(), alg)
    f = ko.gkl_initialize(it)
    for _ in range(steps):
        f = ko.gkl_expand(it, f)
    tol = 1e-10 if orth >= 2 else 1e-6
    np.testing.assert_allclose(al, f.alphas, rtol=tol)
    np.testing.assert_allclose(be, f.betas, rtol=tol)
    if orth >= 2:
        k = steps + 1
        Uk, r = U[:, :k], U[:, k]
        if orth != 2:     # CGS2 re-orthogonalises only r against U (gkl.jl:308-323): V loses orthogonality by design
            assert np.max(np.abs(V.T @ V - np.eye(k))) < 1e-12
        assert np.max(np.abs(Uk.T @ Uk - np.eye(k))) < 1e-12
        B = np.diag(al) + np.diag(be[:-1], -1)                 # lower bidiagonal: A V = U B + r e_k', A'U = V B'
        ek = np.zeros(k); ek[-1] = 1
        assert np.max(np.abs(A @ V - Uk @ B - np.outer(r, ek))) < 1e-11
        assert np.max(np.abs(A.T @ Uk - V @ B.T)) < 1e-10


def _sym(ko, n, seed):
    import scipy.sparse as sp
    R = sp.random(n, n, density=6.0 / n, random_state=np.random.default_rng(seed), format="csr")
    return (R + R.T + sp.diags(np.linspace(1.0, 9.0, n))).tocsr()


@pytest.mark.parametrize("case", ["regular", "rank_drop_issue143", "drift"])
def test_cpu_ref_blocklanczos_matches_numpy_oracle(ko, case):
    """kkref_blocklanczos (src/factorizations/blocklanczos.jl:159-263, 312-353) against the NumPy oracle: block sizes,
    H, residual norm; the issue-#143 fixture exercises the rank drop (20 + 20 + 20 + 11), nearly dependent start vectors
    the DGKS / drift branch."""
    import scipy.sparse as sp
    lib = load_gmres()
    rng = np.random.default_rng(12)
    if case == "regular":
        A, bs, steps, tol = _sym(ko, 400, 3), 4, 6, 1e-12
        X0 = rng.random((400, bs))
    elif case == "rank_drop_issue143":
        A = sp.csr_matrix(np.load(Path(__file__).parent / "golden" / "issue143_A.npy"))
        bs, steps, tol = 20, 3, 1e-8
        X0 = np.random.default_rng(143).standard_normal((A.shape[0], bs))
    else:
        A, bs, steps, tol = _sym(ko, 300, 4), 3, 4, 1e-6
        X0 = rng.random((300, bs))
        X0[:, 2] = X0[:, 0] + 5e-6 * rng.random(300)           # |r_3| after the first MGS lands between tol and 100 tol
        assert ko.block_qr([X0[:, j].copy() for j in range(bs)], tol)[2], "the case must take the DGKS / drift branch"
    n = A.shape[0]
    it = ko.BlockLanczosIterator(A, [X0[:, j].copy() for j in range(bs)], n + bs, qr_tol=tol)
    f = ko.blocklanczos_initialize(it)
    sizes = [len(f)]
    for _ in range(steps):
        k0 = len(f)
        f = ko.blocklanczos_expand(it, f)
        sizes.append(len(f) - k0)
    out = run_kkref_block(lib, A, X0, target_dim=n, max_steps=steps, qr_tol=tol, want_basis=True)
    assert out["sizes"] == sizes and out["k"] == len(f), (out["sizes"], sizes)
    if case == "rank_drop_issue143":
        assert sizes == [20, 20, 20, 11]
    k = len(f)
    scale = np.max(np.abs(f.H[:k, :k]))
    np.testing.assert_allclose(out["H"], f.H[:k, :k], rtol=0, atol=1e-9 * scale)
    np.testing.assert_allclose(np.linalg.eigvalsh(out["H"]), np.linalg.eigvalsh(f.H[:k, :k]), rtol=0,
                               atol=(1e-9 if case == "drift" else 1e-11) * scale)   # drift: start block of condition 1e5
    assert abs(out["norm_R"] - f.norm_R) <= 1e-8 * max(f.norm_R, 1e-3 * scale)
    V = out["V"][:, :k]
    assert np.max(np.abs(V.T @ V - np.eye(k))) < 1e-11
