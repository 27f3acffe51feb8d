"""Pins the CPU oracle (oracle/krylov_oracle.py) against everything the reference's own tests
pin at this boundary: the per-expand! invariants of test/factorize.jl, the
orthogonaliser identities of test/linalg.jl, dense-LAPACK spectra (test/eigsolve.jl,
test/svdsolve.jl, test/linsolve.jl), the issue-#143 matrix fixture and the toric-code known
answer.  CPU only."""
import numpy as np
import pytest
import scipy.sparse as sp
from pathlib import Path

GOLD = Path(__file__).resolve().parent / "golden"
TOL = np.finfo(float).eps ** (2 / 3)  # tolerance(Float64), test/testsetup.jl:14


def rand_sym(n, seed):
    R = np.random.default_rng(seed).random((n, n))
    return (R + R.T) / 2


@pytest.mark.parametrize("n", [10, 100])
def test_orthogonalizer_identities(ko, n):
    """test/linalg.jl:4-25 for all six algorithms (eta0 = 0.75, test/runtests.jl:18-24)."""
    rng = np.random.default_rng(n)
    A = rng.standard_normal((n, n))
    for alg in (ko.CGS, ko.MGS, ko.CGS2, ko.MGS2, ko.CGSIR(0.75), ko.MGSIR(0.75)):
        b = []
        v, beta, _ = ko.orthonormalize(A[:, 0].copy(), b, alg)
        b.append(v)
        for i in range(1, n):
            x = np.zeros(i)
            r, x = ko.orthogonalize(A[:, i].copy(), b, alg, x)
            assert abs(np.hypot(np.linalg.norm(r), np.linalg.norm(x)) - np.linalg.norm(A[:, i])) < 1e-10 * n
            b.append(r / np.linalg.norm(r))
        U = np.stack(b, 1)
        ortho_tol = 1e-10 if alg.name not in ("cgs", "mgs") else 1e-3
        assert np.max(np.abs(U.T @ U - np.eye(n))) < ortho_tol
        v = rng.standard_normal(n)
        np.testing.assert_allclose(U @ v, ko.basis_times(b, v), rtol=1e-12, atol=1e-12)


@pytest.mark.parametrize("n", [10, 100])
def test_lanczos_complete_and_incomplete(ko, n):
    """test/factorize.jl:18-43 (complete) and :120-150 (incomplete): invariants after every expand!."""
    A = rand_sym(n, 1)
    x0 = np.random.default_rng(2).random(n)
    for alg in (ko.CGS2, ko.MGS2, ko.CGSIR(0.75), ko.MGSIR(0.75)):
        it = ko.LanczosIterator(A, x0.copy(), alg)
        f = ko.lanczos_initialize(it)
        while len(f) < n:
            f = ko.lanczos_expand(it, f)
            k = len(f)
            V = np.stack(f.V, 1)
            T = np.diag(f.alphas) + np.diag(f.betas[:-1], 1) + np.diag(f.betas[:-1], -1)
            ek = np.zeros(k); ek[-1] = 1
            assert np.max(np.abs(V.T @ V - np.eye(k))) < 1e-11
            assert np.max(np.abs(A @ V - V @ T - np.outer(f.r, ek))) < 1e-10
            assert abs(np.linalg.norm(f.r) - f.normres) < 1e-12
        assert f.normres < 10 * n * np.finfo(float).eps * np.linalg.norm(A)  # :28
        np.testing.assert_allclose(np.sort(np.linalg.eigvalsh(T)), np.linalg.eigvalsh(A), rtol=0, atol=1e-10 * n)
        f = ko.lanczos_shrink(f, n // 2)
        assert len(f) == n // 2 and len(f.V) == n // 2
        f2 = ko.lanczos_initialize_(it, f)
        assert len(f2) == 1 and abs(np.linalg.norm(f2.V[0]) - 1) < 1e-14


def test_issue143_matrix_fixture(ko):
    """The 71x71 matrix of test/issues.jl:40-112: D ~ eigvals(A) from a complete factorization."""
    A = np.load(GOLD / "issue143_A.npy")
    ev = np.linalg.eigvalsh(A)
    x0 = np.random.default_rng(143).standard_normal(71)
    it = ko.LanczosIterator(A, x0, ko.MGS2)
    f = ko.lanczos_initialize(it)
    while len(f) < 71 and f.normres > 1e-9 * np.linalg.norm(A):
        f = ko.lanczos_expand(it, f)
    T = np.diag(f.alphas) + np.diag(f.betas[:-1], 1) + np.diag(f.betas[:-1], -1)
    D = np.linalg.eigvalsh(T)
    # every Ritz value of the (possibly early-terminated) complete factorization is an eigenvalue
    for d in D:
        assert np.min(np.abs(ev - d)) < TOL * np.max(np.abs(ev))
    vals, vecs, info = ko.eigsolve_lanczos(A, x0, 4, "SR", krylovdim=30, tol=1e-8, maxiter=300)
    assert info.converged >= 4
    # the lowest eigenvalue (0) is 3-fold degenerate: single-vector Lanczos resolves one copy per
    # invariant subspace (the reason the reference's regression test uses BlockLanczos), so compare
    # as a set: every returned value is an eigenvalue, the first one is the smallest.
    scale = np.max(np.abs(ev))
    for d in vals[:4]:
        assert np.min(np.abs(ev - d)) < 1e-7 * scale
    assert abs(vals[0] - ev[0]) < 1e-7 * scale
    for lam, v in zip(vals[:4], vecs[:4]):
        assert np.linalg.norm(A @ v - lam * v) < 1e-6 * scale


def toric_code_hamiltonian(m, n):
    """Deterministic sparse Hamiltonian of test/eigsolve.jl:471-535 (dimension 2^(2mn))."""
    li = lambda i, j: ((j - 1) % n) * m + ((i - 1) % m) + 1  # LinearIndices((m,n))[mod1(i,m), mod1(j,n)]
    bottom = lambda i, j: li(i, j) + m * n
    right = lambda i, j: li(i, j)
    xs, zs = [], []
    for j in range(1, n + 1):          # `for i in 1:m, j in 1:n` -> j outer... order is irrelevant except which
        for i in range(1, m + 1):      # string is dropped by [1:end-1]: Julia iterates j fastest for `i in, j in`
            pass
    for i in range(1, m + 1):
        for j in range(1, n + 1):
            xs.append((bottom(i, j + 1), right(i, j), bottom(i, j), right(i - 1, j)))
            zs.append((right(i, j), bottom(i, j), right(i, j - 1), bottom(i + 1, j)))
    N = 2 * m * n
    dim = 2 ** N
    idx = np.arange(dim, dtype=np.int64)
    rows, cols, vals = [], [], []
    diag = np.zeros(dim)
    for s in xs[:-1]:
        mask = 0
        for pos in s:
            mask |= 1 << (N - pos)
        rows.append(idx); cols.append(idx ^ mask); vals.append(np.ones(dim))
    for s in zs[:-1]:
        par = np.zeros(dim, dtype=np.int64)
        for pos in s:
            par ^= (idx >> (N - pos)) & 1
        diag += 1.0 - 2.0 * par
    H = sp.coo_matrix((np.concatenate(vals), (np.concatenate(rows), np.concatenate(cols))), shape=(dim, dim)).tocsr()
    return H + sp.diags(diag)


def test_toric_code_known_answer(ko):
    """test/eigsolve.jl:537-548: the lowest eigenvalue of -H (3x3 toric code, dim 2^18) is -16.
    (The reference resolves its 4-fold degeneracy with BlockLanczos; single-vector Lanczos must
    find the value itself.)"""
    H = toric_code_hamiltonian(3, 3)
    x0 = np.random.default_rng(7).random(H.shape[0])
    vals, vecs, info = ko.eigsolve_lanczos(-H, x0, 1, "SR", krylovdim=30, tol=1e-8, maxiter=50)
    assert info.converged >= 1
    assert abs(vals[0] + 16.0) < 1e-8
    assert np.linalg.norm(-H @ vecs[0] - vals[0] * vecs[0]) < 1e-6


@pytest.mark.parametrize("n", [10, 100])
def test_arnoldi_invariants(ko, n):
    """test/factorize.jl:66-117,155-195."""
    A = np.random.default_rng(3).random((n, n))
    x0 = np.random.default_rng(4).random(n)
    for alg in (ko.CGS2, ko.MGS2, ko.CGSIR(0.75), ko.MGSIR(0.75)):
        it = ko.ArnoldiIterator(A, x0.copy(), alg)
        f = ko.arnoldi_initialize(it)
        while len(f) < n:
            f = ko.arnoldi_expand(it, f)
            k = len(f)
            V, H = np.stack(f.V, 1), f.rayleighquotient()
            ek = np.zeros(k); ek[-1] = 1
            assert np.max(np.abs(V.T @ V - np.eye(k))) < 1e-11
            assert np.max(np.abs(A @ V - V @ H - np.outer(f.r, ek))) < 1e-10 * n
            assert abs(np.linalg.norm(f.r) - f.normres) < 1e-12
        assert f.normres < 10 * n * np.finfo(float).eps * np.linalg.norm(A)
        f = ko.arnoldi_shrink(f, n // 2)
        assert len(f.H) == ((n // 2) ** 2 + 3 * (n // 2)) // 2


def test_gkl_invariants(ko):
    """test/factorize.jl:239-243,285-296."""
    A = np.random.default_rng(5).random((60, 40))
    u0 = np.random.default_rng(6).random(60)
    for alg in (ko.MGS2, ko.CGSIR(0.75), ko.MGSIR(0.75)):
        it = ko.GKLIterator(A, u0.copy(), alg)
        f = ko.gkl_initialize(it)
        while len(f) < 30:
            f = ko.gkl_expand(it, f)
            k = len(f)
            U, V = np.stack(f.U, 1), np.stack(f.V, 1)
            B = np.diag(f.alphas) + np.diag(f.betas[:-1], -1)
            ek = np.zeros(k); ek[-1] = 1
            assert np.max(np.abs(U.T @ U - np.eye(k))) < 1e-11 and np.max(np.abs(V.T @ V - np.eye(k))) < 1e-11
            assert np.max(np.abs(A @ V - U @ B - np.outer(f.r, ek))) < 1e-11
            assert np.max(np.abs(A.T @ U - V @ B.T)) < 1e-11
    with pytest.raises(ValueError):
        ko.gkl_initialize(ko.GKLIterator((lambda x: A @ x, lambda x: 2 * (A.T @ x)), u0))  # gkl.jl:192


@pytest.mark.parametrize("which", ["LM", "LR", "SR"])
def test_eigsolve_vs_dense(ko, which):
    """test/eigsolve.jl:74,122-123."""
    n = 100
    A = rand_sym(n, 11)
    ev = np.linalg.eigvalsh(A)
    x0 = np.random.default_rng(12).random(n)
    for alg in (ko.CGS2, ko.MGS2, ko.CGSIR(0.75), ko.MGSIR(0.75)):
        vals, vecs, info = ko.eigsolve_lanczos(A, x0, 5, which, krylovdim=3 * 10, tol=1e-12, maxiter=200, orth=alg)
        assert info.converged >= 5
        key = {"LM": lambda d: -np.abs(d), "SR": lambda d: d, "LR": lambda d: -d}[which]
        expect = ev[np.argsort(key(ev), kind="stable")][: len(vals)]
        np.testing.assert_allclose(vals, expect, rtol=0, atol=TOL)
        for lam, v in zip(vals, vecs):
            assert np.linalg.norm(A @ v - lam * v) < 1e-9
    # full factorization, krylovdim = n (test/eigsolve.jl:60-83)
    vals, vecs, info = ko.eigsolve_lanczos(A, x0, n, "SR", krylovdim=n, tol=1e-12, maxiter=1)
    np.testing.assert_allclose(np.sort(vals), ev, rtol=0, atol=TOL)


def test_gmres_vs_dense(ko):
    """test/linsolve.jl:120-137,215-232."""
    n = 100
    rng = np.random.default_rng(13)
    A = rng.random((n, n)) - 0.5
    A = A @ A.T / n + np.eye(n) * 0.5 + (rng.random((n, n)) - 0.5) * 0.05
    b = rng.random(n)
    for a0, a1 in ((0.0, 1.0), (0.4, 1.3)):
        for alg in (ko.CGS, ko.MGS, ko.CGS2, ko.MGS2, ko.CGSIR(0.75), ko.MGSIR(0.75)):
            tol = 1e-11 * np.linalg.norm(b)
            x, info = ko.gmres(A, b, None, a0, a1, krylovdim=20, maxiter=50, tol=tol, orth=alg)
            assert info.converged == 1
            assert np.linalg.norm(b - (a0 * x + a1 * (A @ x))) <= 1.01 * tol
    xs = rng.random(n)
    x, info = ko.gmres(A, A @ xs, xs, krylovdim=10, tol=1e-8)
    assert info.numops == 1  # test/linsolve.jl:167


def test_svdsolve_vs_dense(ko):
    """test/svdsolve.jl:14,61-63,89,97-98."""
    rng = np.random.default_rng(14)
    A = rng.random((80, 50))
    sv = np.linalg.svd(A, compute_uv=False)
    for alg in (ko.CGS2, ko.MGS2, ko.CGSIR(0.75), ko.MGSIR(0.75)):
        S, L, R, info = ko.svdsolve_gkl(A, rng.random(80), 6, "LR", krylovdim=20, tol=1e-11, maxiter=200, orth=alg)
        assert info.converged >= 6
        np.testing.assert_allclose(S[:6], sv[:6], rtol=0, atol=TOL * sv[0])
        Lm, Rm = np.stack(L, 1), np.stack(R, 1)
        assert np.max(np.abs(Lm.T @ Lm - np.eye(len(S)))) < 1e-9
        assert np.max(np.abs(A @ Rm - Lm * S)) < 1e-8 and np.max(np.abs(A.T @ Lm - Rm * S)) < 1e-8
    # :SR on the wide matrix A' (50 x 80): the start vector lives in the 50-dim codomain, A'A'' is
    # full rank, so the smallest Ritz values are the smallest singular values
    S, L, R, info = ko.svdsolve_gkl(A.T.copy(), rng.random(50), 3, "SR", krylovdim=50, tol=1e-10, maxiter=300)
    np.testing.assert_allclose(S[:3], sv[::-1][:3], rtol=0, atol=1e-8)


def test_laplacian_closed_form(ko):
    """SURVEY 8(d) cfg 2: closed-form spectrum of the synthetic 5-point Laplacian."""
    nx, ny = 13, 9
    A = ko.laplacian_2d(nx, ny)
    assert A.nnz == 5 * nx * ny - 2 * (nx + ny)
    np.testing.assert_allclose(np.linalg.eigvalsh(A.toarray()), ko.laplacian_2d_eigs(nx, ny), atol=1e-12)
    C = ko.convection_diffusion_2d(7, 5)
    assert np.max(np.abs((C - C.T).toarray())) > 0.1  # nonsymmetric


def test_small_dense_helpers(ko):
    rng = np.random.default_rng(15)
    x = rng.standard_normal(7)
    for i in (0, 3, 6):
        beta, v, nu = ko.householder_vec(x, i)
        Hx = x - beta * v * (v @ x)
        e = np.zeros(7); e[i] = nu
        np.testing.assert_allclose(Hx, e, atol=1e-13)
        assert nu >= 0 and abs(nu - np.linalg.norm(x)) < 1e-13
    for f, g in ((3.0, 4.0), (-3.0, 1.0), (0.0, 2.0), (2.0, 0.0), (1.0, -5.0)):
        c, s, r = ko.givens(f, g)
        np.testing.assert_allclose([c * f + s * g, -s * f + c * g], [r, 0.0], atol=1e-14)
        assert abs(c * c + s * s - 1) < 1e-14
    # packed Hessenberg layout (dense/packedhessenberg.jl:32-39)
    k, off = 5, 0
    for j in range(1, k + 1):
        for i in range(1, min(j + 1, k) + 1):
            assert ko.packed_index(i, j) == off
            off += 1


def test_blocklanczos_issue143_known_answer(ko):
    """test/issues.jl:114-128 verbatim: eigsolve(A, Block of 20 randn(71), 4, :SR, BlockLanczos(tol=1e-8))
    returns ALL 71 eigenvalues with numiter == 1 and numops == length(D) + 1."""
    A = np.load(GOLD / "issue143_A.npy")
    rng = np.random.default_rng(143)
    x0 = [rng.standard_normal(71) for _ in range(20)]
    D, V, info = ko.eigsolve_blocklanczos(A, x0, 4, "SR", tol=1e-8)
    ev = np.linalg.eigvalsh(A)
    assert len(D) == len(ev)
    np.testing.assert_allclose(np.sort(D), ev, rtol=0, atol=TOL * np.max(np.abs(ev)))
    U = np.stack(V, 1)
    assert np.max(np.abs(A @ U - U * D)) < TOL * np.max(np.abs(ev)) * 10
    assert info.converged == len(D)
    assert info.numiter == 1
    assert info.numops == len(D) + 1
    assert np.max(info.normres) < TOL


def test_blocklanczos_toric_code_known_answer(ko):
    """test/eigsolve.jl:537-543: exactly 4 of the 10 lowest eigenvalues of -H equal -16 (block size 5)."""
    H = toric_code_hamiltonian(3, 3)
    rng = np.random.default_rng(5)
    x0 = [rng.random(H.shape[0]) for _ in range(5)]
    tol = 1e-6
    D, U, info = ko.eigsolve_blocklanczos(-H, x0, 10, "SR", tol=tol, maxiter=1)
    assert np.sum(np.abs(D[:10] + 16.0) < 2.0 - tol) == 4
    assert np.sum(np.abs(D[:10] + 16.0) < tol) == 4


def test_blocklanczos_bs1_equals_lanczos(ko):
    """test/eigsolve.jl:685-715: BlockLanczos with block size 1 == Lanczos (values, numops + 1)."""
    n = 100
    A = rand_sym(n, 21)
    x0 = np.random.default_rng(22).random(n)
    D1, V1, i1 = ko.eigsolve_lanczos(A, x0, 3, "SR", krylovdim=30, tol=1e-10, maxiter=50, orth=ko.MGS2)
    D2, V2, i2 = ko.eigsolve_blocklanczos(A, [x0.copy()], 3, "SR", krylovdim=30, tol=1e-10, maxiter=50)
    assert i1.converged >= 3 and i2.converged >= 3
    np.testing.assert_allclose(D1[:3], D2[:3], rtol=0, atol=TOL)
    assert i1.numiter == i2.numiter
    assert i1.numops + 1 == i2.numops


def test_block_primitives(ko):
    """test/block.jl:74-86,105-124,142-161."""
    rng = np.random.default_rng(31)
    n, p = 100, 6
    A = [rng.standard_normal(n) for _ in range(p)]
    B = [rng.standard_normal(n) for _ in range(p)]
    np.testing.assert_allclose(ko.block_inner(A, B), np.stack(A, 1).T @ np.stack(B, 1), atol=1e-12)
    Q, _ = np.linalg.qr(rng.standard_normal((n, 10)))
    b0 = [Q[:, j].copy() for j in range(10)]
    b1 = ko.block_reorthogonalize([a.copy() for a in A], b0)
    assert np.linalg.norm(ko.block_inner(b0, b1)) < TOL
    # rank-deficient block: Q*R ~ A, Q'Q ~ I, fewer good columns
    C = [a.copy() for a in A] + [A[0] + 2 * A[1], A[2] - A[3]]
    Cm = np.stack(C, 1)
    R, good, drift = ko.block_qr(C, 1e-10)
    assert len(good) == p and R.shape == (p, p + 2)
    Qm = np.stack([C[i] for i in good], 1)
    np.testing.assert_allclose(Qm @ R, Cm, atol=1e-10)
    assert np.max(np.abs(Qm.T @ Qm - np.eye(p))) < 1e-12


def test_cg_vs_dense(ko):
    """test/linsolve.jl:4-62: CG on a positive definite matrix, b ~ (a0 + a1 A) x, numops == 1 from the solution."""
    n = 100
    rng = np.random.default_rng(41)
    M = rng.random((n, n)) - 0.5
    A = M @ M.T / n + np.eye(n)
    b = rng.random(n)
    for a0, a1 in ((0.0, 1.0), (0.5, 1.2)):
        x, info = ko.cg(A, b, None, a0, a1, maxiter=300, tol=1e-11 * np.linalg.norm(b))
        assert info.converged == 1
        assert np.linalg.norm(b - (a0 * x + a1 * (A @ x))) <= 1.01e-11 * np.linalg.norm(b)
    xs = rng.random(n)
    x, info = ko.cg(A, A @ xs, xs, tol=1e-9)
    assert info.numops == 1 and info.numiter == 0


def test_bicgstab_reference_cases(ko):
    """test/linsolve.jl:288-350 (BiCGStab small problem) and :352-395 (large problem), real Float64 case."""
    rng = np.random.default_rng(31)
    n = 10
    A = rng.random((n, n)) - 0.5
    A = np.eye(n) - 0.9 * A / np.max(np.abs(np.linalg.eigvals(A)))
    b = rng.random(n)
    tol = 1e-10 * np.linalg.norm(b)
    x, info = ko.bicgstab(A, b, np.zeros(n), maxiter=4 * n, tol=tol)
    assert info.converged > 0
    np.testing.assert_allclose(A @ x, b, rtol=0, atol=10 * tol)
    x2, info2 = ko.bicgstab(A, b, x, maxiter=4 * n, tol=tol)      # restart from the solution: one operator application
    assert info2.numops == 1 and info2.converged == 1
    a0, a1 = rng.random() + 1, rng.random()
    x, info = ko.bicgstab(A, b, np.zeros(n), a0, a1, maxiter=4 * n, tol=tol)
    assert info.converged > 0
    np.testing.assert_allclose(a0 * x + a1 * (A @ x), b, rtol=0, atol=10 * tol)
    # large problem: maxiter = 2 stops early with b = (a0 + a1 A) x + residual exactly   (:360-366)
    N = 60
    A = rng.random((N, N)) - 0.5
    b = rng.random(N)
    a0 = np.max(np.abs(np.linalg.eigvals(A)))
    a1 = -0.9 * rng.random()
    x, info = ko.bicgstab(A, b, np.zeros(N), a0, a1, maxiter=2, tol=1e-10 * np.linalg.norm(b))
    np.testing.assert_allclose(a0 * x + a1 * (A @ x) + info.residual, b, rtol=0, atol=1e-12 * np.linalg.norm(b))
    if info.converged == 0:
        assert info.numiter == 2
    x, info = ko.bicgstab(A, b, x, a0, a1, maxiter=10 * N, tol=1e-10 * np.linalg.norm(b))
    assert info.converged > 0
    np.testing.assert_allclose(a0 * x + a1 * (A @ x), b, rtol=0, atol=1e-9 * np.linalg.norm(b))


def test_lsmr_reference_cases(ko):
    """test/issues.jl:22-29 (issue #133, exact known answer) and test/lssolve.jl:2-60 (rank-deficient small problem)."""
    x, info = ko.lsmr(np.eye(2), np.array([1.0, 0.0]), tol=1e-12)
    assert np.array_equal(x, [1.0, 0.0])
    assert (info.converged, info.numiter, info.numops, info.normres) == (1, 1, 2, 0.0)
    rng = np.random.default_rng(17)
    n = 10
    A = rng.random((2 * n, n))
    U, S, Vt = np.linalg.svd(A, full_matrices=False)
    invS = 1 / S
    S[-1] = 0.0
    invS[-1] = 0.0
    A = U @ np.diag(S) @ Vt
    b = rng.random(2 * n)
    tol = 10 * n * np.finfo(float).eps
    x, info = ko.lsmr(A, b, maxiter=3, krylovdim=1, tol=1e-12 * np.linalg.norm(b))   # no reorthogonalisation
    r = b - A @ x
    np.testing.assert_allclose(info.residual, r, atol=1e-12)
    np.testing.assert_allclose(info.normres, np.linalg.norm(A.T @ r), rtol=1e-8)
    assert info.converged == 0
    # reorthogonalisation is essential to converge in exactly n iterations   (:38-44)
    x, info = ko.lsmr(A, b, maxiter=n, tol=tol, krylovdim=n)
    assert info.converged > 0
    assert abs(Vt[-1] @ x) < tol
    np.testing.assert_allclose(x, Vt.T @ (invS * (U.T @ b)), atol=1e-9)
    lam = rng.random()
    x, info = ko.lsmr(A, b, lam, maxiter=n, tol=tol, krylovdim=n)
    assert info.converged > 0
    np.testing.assert_allclose(A.T @ (b - A @ x), lam ** 2 * x, atol=2 * tol * 10)


def _phi(A, v, p):
    """test/expintegrator.jl:1-13: phi_p(A) v through the augmented matrix exponential."""
    import scipy.linalg as sla
    m = A.shape[0]
    Ap = np.zeros((m + p, m + p))
    Ap[:m, :m] = A
    Ap[:m, m] = v
    for k in range(1, p):
        Ap[m + k - 1, m + k] = 1.0
    return sla.expm(Ap)[:m, -1]


@pytest.mark.parametrize("method", ["lanczos", "arnoldi"])
def test_expintegrator_reference_cases(ko, method):
    """test/expintegrator.jl:15-118 (full Krylov space) and :120-191 (iterative, krylovdim < n), real Float64."""
    import scipy.linalg as sla
    rng = np.random.default_rng(23)
    n = 10
    A = rng.random((n, n)) - 0.5
    if method == "lanczos":
        A = (A + A.T) / 2
    for orth in (ko.CGS2, ko.MGS2, ko.CGSIR(), ko.MGSIR()):
        W = np.zeros((n, n))
        for k in range(n):
            W[:, k], _ = ko.expintegrator(A, 1.0, (np.eye(n)[:, k],), krylovdim=n, maxiter=2, tol=1e-12, orth=orth, method=method)
        np.testing.assert_allclose(W, sla.expm(A), atol=1e-10)
        for t in (rng.random(), -rng.random()):
            for p in range(1, 6):
                u = [rng.random(n) for _ in range(p + 1)]
                w, info = ko.expintegrator(A, t, u, krylovdim=n, maxiter=2, tol=1e-12, orth=orth, method=method)
                w2 = sla.expm(t * A) @ u[0]
                for j in range(1, p + 1):
                    w2 = w2 + t ** j * _phi(t * A, u[j], j)
                assert info.converged > 0
                np.testing.assert_allclose(w, w2, atol=1e-9)
    # iterative: N = 100, krylovdim 20   (:120-191)
    N = 100
    A = rng.random((N, N)) - 0.5
    if method == "lanczos":
        A = (A + A.T) / 2
    s = np.max(np.abs(np.linalg.eigvals(A)))
    A = A / s
    for p in (1, 3):
        u = [rng.random(N) for _ in range(p + 1)]
        w1, info = ko.expintegrator(A, 1.0, u, krylovdim=20, maxiter=100, tol=1e-10, method=method)
        assert info.converged > 0
        w2 = sla.expm(A) @ u[0]
        for j in range(1, p + 1):
            w2 = w2 + _phi(A, u[j], j)
        np.testing.assert_allclose(w1, w2, atol=1e-8)
        w1e, infoe = ko.expintegrator(A, 1.0, u, krylovdim=20, maxiter=100, tol=1e-10, method=method, eager=True)
        assert infoe.converged > 0
        np.testing.assert_allclose(w1e, w2, atol=1e-8)


def test_expintegrator_fixed_point_branch(ko):
    """Analogue of test/expintegrator.jl:193-214 (fixed-point branch, :161-166 / :302-307) with a real negative
    definite A instead of the reference's shifted complex one: t = 1000 converges to the fixed point of the ODE."""
    rng = np.random.default_rng(29)
    n = 10
    A = rng.random((n, n)) - 0.5
    A = -(A @ A.T) - np.eye(n) * 0.1
    v0, v1 = rng.random(n), rng.random(n)
    w1, info1 = ko.expintegrator(A, 1000.0, (v0,), krylovdim=n, maxiter=100, tol=1e-12, method="arnoldi")
    assert info1.converged > 0
    np.testing.assert_allclose(w1, np.zeros(n), atol=1e-9)
    w2, info2 = ko.expintegrator(A, 1000.0, (v0, v1), krylovdim=n, maxiter=100, tol=1e-12, method="arnoldi")
    assert info2.converged > 0
    np.testing.assert_allclose(A @ w2 + v1, np.zeros(n), atol=1e-8)


def _sorted_eigs(D):
    """test/eigsolve.jl:208: sort by imag (rev) then stably by real."""
    D = np.asarray(D)
    D = D[np.argsort(-D.imag, kind="stable")]
    return D[np.argsort(D.real, kind="stable")]


@pytest.mark.parametrize("orth_name", ["CGS2", "MGS2", "CGSIR", "MGSIR"])
def test_arnoldi_eigsolve_full(ko, orth_name):
    """test/eigsolve.jl:137-214 (Arnoldi - eigsolve full): n1 smallest-real + n2 largest-real values = spectrum."""
    orth = getattr(ko, orth_name)
    orth = orth() if callable(orth) else orth
    rng = np.random.default_rng(41)
    n = 10
    A = rng.random((n, n)) - 0.5
    v = rng.random(n)
    n1 = n // 2
    D1, V1, info1 = ko.eigsolve_arnoldi(A, v, n1, "SR", krylovdim=n, maxiter=1, tol=1e-12, orth=orth)
    n2 = n - n1
    D2, V2, info2 = ko.eigsolve_arnoldi(A, v, n2, "LR", krylovdim=2 * n, maxiter=1, tol=1e-12, orth=orth)
    D = _sorted_eigs(np.linalg.eigvals(A))
    D2s = _sorted_eigs(D2)
    np.testing.assert_allclose(np.concatenate([D1[:n1], D2s[len(D2s) - n2:]]), D, atol=1e-9)
    for Dk, Vk in ((D1, V1), (D2, V2)):
        Uk = np.stack(Vk, axis=1)
        np.testing.assert_allclose(A @ Uk, Uk * Dk[None, :], atol=1e-9)


def test_arnoldi_eigsolve_iteratively(ko):
    """test/eigsolve.jl:262-330 (Arnoldi - eigsolve iteratively): N = 100, krylovdim = 3n, eager, restarts."""
    rng = np.random.default_rng(43)
    N, n = 100, 10
    A = rng.random((N, N)) - 0.5
    v = rng.random(N)
    ev = np.linalg.eigvals(A)
    ev = ev[np.argsort(-ev.imag, kind="stable")]
    kw = dict(krylovdim=3 * n, maxiter=20, tol=1e-12, eager=True)
    D1, V1, info1 = ko.eigsolve_arnoldi(A, v, n, "SR", **kw)
    D2, V2, info2 = ko.eigsolve_arnoldi(A, v, n, "LR", **kw)
    D3, V3, info3 = ko.eigsolve_arnoldi(A, v, n, "LM", **kw)
    l1, l2, l3 = info1.converged, info2.converged, info3.converged
    assert l1 > 0 and l2 > 0 and l3 > 0

    def close_as_sets(a, b):
        a, b = list(a), list(b)
        for z in a:
            k = int(np.argmin([abs(z - w) for w in b]))
            assert abs(z - b[k]) < 1e-8 * max(1.0, abs(z))
            b.pop(k)

    close_as_sets(D1[:l1], ev[np.argsort(ev.real, kind="stable")][:l1])
    close_as_sets(D2[:l2], ev[np.argsort(-ev.real, kind="stable")][:l2])
    close_as_sets(D3[:l3], ev[np.argsort(-np.abs(ev), kind="stable")][:l3])
    for Dk, Vk, infok in ((D1, V1, info1), (D2, V2, info2), (D3, V3, info3)):
        Uk = np.stack(Vk, axis=1)
        Rk = np.stack(infok.residual, axis=1)
        np.testing.assert_allclose(A @ Uk, Uk * Dk[None, :] + Rk, atol=1e-9)     # :318-320
        assert np.all(infok.normres[:infok.converged] <= 1e-12 * 1.0001)


@pytest.mark.parametrize("orth_name", ["CGS2", "MGS2", "CGSIR", "MGSIR"])
def test_golubye_geneigsolve_full(ko, orth_name):
    """test/geneigsolve.jl:1-113 (GolubYe - geneigsolve full), real Float64."""
    import scipy.linalg as sla
    orth = getattr(ko, orth_name)
    orth = orth() if callable(orth) else orth
    rng = np.random.default_rng(51)
    n = 10
    A = rng.random((n, n)) - 0.5
    A = (A + A.T) / 2
    B = rng.random((n, n)) - 0.5
    B = np.real(sla.sqrtm(B @ B.T))
    v = rng.random(n)
    n1 = n // 2
    D1, V1, info = ko.geneigsolve_golubye(A, B, v, n1, "SR", krylovdim=n, maxiter=1, tol=1e-12, orth=orth)
    n2 = n - n1
    D2, V2, info2 = ko.geneigsolve_golubye(A, B, v, n2, "LR", krylovdim=n, maxiter=1, tol=1e-12, orth=orth)
    ref = sla.eigh(A, B, eigvals_only=True)
    np.testing.assert_allclose(np.concatenate([D1[:n1], D2[:n2][::-1]]), ref, atol=1e-8)
    for Dk, Vk in ((D1, V1), (D2, V2)):
        U = np.stack(Vk, axis=1)
        np.testing.assert_allclose(U.T @ B @ U, np.eye(U.shape[1]), atol=1e-8)
        np.testing.assert_allclose(A @ U, B @ U * Dk[None, :], atol=1e-7)


def test_golubye_geneigsolve_iteratively(ko):
    """test/geneigsolve.jl:115-172 (GolubYe - geneigsolve iteratively): N = 100, krylovdim = 3n, restarts."""
    import scipy.linalg as sla
    rng = np.random.default_rng(53)
    N, n = 100, 10
    A = rng.random((N, N)) - 0.5
    A = (A + A.T) / 2
    B = rng.random((N, N)) - 0.5
    B = np.real(sla.sqrtm(B @ B.T))
    v = rng.random(N)
    tol = np.linalg.cond(B) * 1e-12
    D1, V1, info1 = ko.geneigsolve_golubye(A, B, v, n, "SR", krylovdim=3 * n, maxiter=100, tol=tol)
    D2, V2, info2 = ko.geneigsolve_golubye(A, B, v, n, "LR", krylovdim=3 * n, maxiter=100, tol=tol)
    l1, l2 = info1.converged, info2.converged
    assert l1 > 0 and l2 > 0
    ref = sla.eigh(A, B, eigvals_only=True)
    np.testing.assert_allclose(D1[:l1], ref[:l1], rtol=1e-7, atol=1e-7)
    np.testing.assert_allclose(D2[:l2], ref[::-1][:l2], rtol=1e-7, atol=1e-7)
    for Dk, Vk, infok in ((D1, V1, info1), (D2, V2, info2)):
        U = np.stack(Vk, axis=1)
        R = np.stack(infok.residual, axis=1)
        np.testing.assert_allclose(U.T @ B @ U, np.eye(U.shape[1]), atol=1e-7)
        np.testing.assert_allclose(A @ U, B @ U * Dk[None, :] + R, atol=1e-8)


@pytest.mark.parametrize("orth_name", ["CGS2", "MGS2", "CGSIR", "MGSIR"])
def test_biarnoldi_bieigsolve_full(ko, orth_name):
    """test/bieigsolve.jl:1-135 (BiArnoldi - eigsolve full), real Float64: spectrum, right / left eigenvectors,
    biorthogonality."""
    orth = getattr(ko, orth_name)
    orth = orth() if callable(orth) else orth
    rng = np.random.default_rng(61)
    n = 10
    A = rng.random((n, n)) - 0.5
    v, w = rng.random(n), rng.random(n)
    n1 = n // 2
    D1, (V1, W1), (i1, _) = ko.bieigsolve_biarnoldi(A, v, w, n1, "SR", krylovdim=n, maxiter=1, tol=1e-12, orth=orth)
    n2 = n - n1
    D2, (V2, W2), (i2, _) = ko.bieigsolve_biarnoldi(A, v, w, n2, "LR", krylovdim=2 * n, maxiter=1, tol=1e-12, orth=orth)
    D = _sorted_eigs(np.linalg.eigvals(A))
    D2s = _sorted_eigs(D2)
    np.testing.assert_allclose(np.concatenate([D1[:n1], D2s[len(D2s) - n2:]]), D, atol=1e-8)
    for Dk, Vk, Wk in ((D1, V1, W1), (D2, V2, W2)):
        UV, UW = np.stack(Vk, axis=1), np.stack(Wk, axis=1)
        np.testing.assert_allclose(A @ UV, UV * Dk[None, :], atol=1e-8)
        np.testing.assert_allclose(A.T @ UW, UW * np.conj(Dk)[None, :], atol=1e-8)
        np.testing.assert_allclose(UW.conj().T @ UV, np.eye(UV.shape[1]), atol=1e-8)


@pytest.mark.parametrize("eager", [True, False])
def test_biarnoldi_bieigsolve_iteratively(ko, eager):
    """test/bieigsolve.jl:137-230 (BiArnoldi - eigsolve iteratively): N = 100, krylovdim = 3n, restarts."""
    rng = np.random.default_rng(63)
    N, n = 100, 10
    A = rng.random((N, N)) - 0.5
    v, w = rng.random(N), rng.random(N)
    ev = np.linalg.eigvals(A)
    for which, order in (("SR", np.argsort(ev.real, kind="stable")), ("LR", np.argsort(-ev.real, kind="stable")),
                         ("LM", np.argsort(-np.abs(ev), kind="stable"))):
        D, (Vr, Wl), (iV, iW) = ko.bieigsolve_biarnoldi(A, v, w, n, which, krylovdim=3 * n, maxiter=30, tol=1e-11, eager=eager)
        l = iV.converged
        assert l > 0
        ref = list(ev[order][:l + 2])
        for z in D[:l]:
            j = int(np.argmin([abs(z - r) for r in ref]))
            assert abs(z - ref[j]) < 1e-7 * max(1.0, abs(z))
            ref.pop(j)
        UV, UW = np.stack(Vr, axis=1), np.stack(Wl, axis=1)
        RV, RW = np.stack(iV.residual, axis=1), np.stack(iW.residual, axis=1)
        np.testing.assert_allclose(A @ UV, UV * D[None, :] + RV, atol=1e-8)
        np.testing.assert_allclose(A.T @ UW, UW * np.conj(D)[None, :] + RW, atol=1e-8)
        np.testing.assert_allclose((UW.conj().T @ UV)[:l, :l], np.eye(l), atol=1e-6)

