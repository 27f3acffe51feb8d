"""More basis vectors than one kernel panel (KK_MAX_M = 256): the reference has no limit on krylovdim, so every
`orthogonalize!!` / `expand!` entry point goes panel by panel beyond it (csrc/kk_orth.hip::orth_run_wide) and must keep
matching the oracle -- src/orthonormal.jl:378-452, src/factorizations/{lanczos,arnoldi,gkl}.jl."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def relerr(a, b):
    a, b = np.asarray(a, float), np.asarray(b, float)
    return float(np.max(np.abs(a - b) / np.maximum(np.abs(b), 1e-300)))


def pairs(kk, ko):
    return [(kk.ClassicalGramSchmidt(), ko.CGS), (kk.ModifiedGramSchmidt(), ko.MGS), (kk.ClassicalGramSchmidt2(), ko.CGS2),
            (kk.ModifiedGramSchmidt2(), ko.MGS2), (kk.ClassicalGramSchmidtIR(0.99), ko.CGSIR(0.99)),
            (kk.ModifiedGramSchmidtIR(0.99), ko.MGSIR(0.99))]


@pytest.mark.parametrize("mgs_mode", [0, 1])
def test_orthogonalize_against_300_vectors(kk, ko, ctx, mgs_mode):
    ctx.set_option("mgs_mode", mgs_mode)
    rng = np.random.default_rng(7)
    n, m = 2500, 300
    Q, _ = np.linalg.qr(rng.standard_normal((n, m)))
    B = kk.DeviceBasis(n, m + 2, ctx)
    for j in range(m):
        B.upload(j, Q[:, j])
    B.length = m
    w = rng.standard_normal(n)
    for dev, ref in pairs(kk, ko):
        x, nrm, npass = B.orthogonalize(B[m].set(w), dev)
        wo, xo = ko.orthogonalize(w.copy(), [Q[:, j].copy() for j in range(m)], ref)
        np.testing.assert_allclose(x, xo, rtol=0, atol=1e-12 * np.linalg.norm(w), err_msg=dev.name)
        np.testing.assert_allclose(B[m].get(), wo, rtol=0, atol=1e-12 * np.linalg.norm(w), err_msg=dev.name)
        assert abs(nrm - np.linalg.norm(wo)) < 1e-12 * np.linalg.norm(w)
    ctx.set_option("mgs_mode", 2)


def test_lanczos_krylovdim_300(kk, ko, ctx):
    """factorizations/lanczos.jl:250-376 with the basis growing to 300 vectors: fused expand below the panel limit, the
    panel-by-panel route above it, one continuous (alpha, beta) trajectory"""
    nx, ny, steps = 60, 50, 299
    n = nx * ny
    A = ko.laplacian_2d(nx, ny, shift_diag=10 * np.linspace(0, 1, n) ** 2)
    x0 = np.random.default_rng(3).random(n)
    op = kk.SparseOperator(A, ctx, symmetric=True)
    for dev, ref in pairs(kk, ko)[2:]:
        it = kk.LanczosIterator(op, x0, dev, capacity=steps + 3)
        f = kk.initialize(it)
        oit = ko.LanczosIterator(A, x0.copy(), ref)
        of = ko.lanczos_initialize(oit)
        for _ in range(steps):
            f = kk.expand_(it, f)
            of = ko.lanczos_expand(oit, of)
        assert len(f) == 300
        assert relerr(f.alphas, of.alphas) < 1e-9 and relerr(f.betas, of.betas) < 1e-9, dev.name
        V = f.V.to_numpy()
        assert np.max(np.abs(V.T @ V - np.eye(300))) < 1e-11, dev.name
        T = np.diag(f.alphas) + np.diag(f.betas[:-1], 1) + np.diag(f.betas[:-1], -1)
        ek = np.zeros(300); ek[-1] = 1
        assert np.max(np.abs(A @ V - V @ T - np.outer(f.r.get(), ek))) < 1e-10, dev.name


def test_arnoldi_and_gkl_beyond_the_panel_limit(kk, ko, ctx):
    A = ko.convection_diffusion_2d(45, 40)
    n = A.shape[0]
    x0 = np.random.default_rng(4).random(n)
    steps = 270
    it = kk.ArnoldiIterator(kk.SparseOperator(A, ctx), x0, kk.ModifiedGramSchmidt2(), capacity=steps + 3)
    f = kk.initialize(it)
    for _ in range(steps):
        f = kk.expand_(it, f)
    k = len(f)
    V, r, H = f.V.to_numpy(), f.r.get(), f.rayleighquotient()
    ek = np.zeros(k); ek[-1] = 1
    assert k == steps + 1 and np.max(np.abs(V.T @ V - np.eye(k))) < 1e-11
    assert np.max(np.abs(A @ V - V @ H - np.outer(r, ek))) < 1e-10
    # GKL: 260 steps on a rectangular map with MGS2 (both bases swept; CGS2 re-orthogonalises only r against U, gkl.jl:308-323,
    # and loses the relation below once V has lost its orthogonality -- in the reference as well)
    Ar = ko.sparse_random(900, 700, 9, 3)
    u0 = np.random.default_rng(6).random(900)
    for dev in (kk.ModifiedGramSchmidt2(),):
        git = kk.GKLIterator(kk.SparseOperator(Ar, ctx), u0, dev, capacity=265)
        gf = kk.initialize(git)
        for _ in range(260):
            gf = kk.expand_(git, gf)
        U, Vv, B = gf.U.to_numpy(), gf.V.to_numpy(), gf.rayleighquotient()
        kq = len(gf)
        assert np.max(np.abs(U.T @ U - np.eye(kq))) < 1e-11 and np.max(np.abs(Vv.T @ Vv - np.eye(kq))) < 1e-11, dev.name
        ekq = np.zeros(kq); ekq[-1] = 1
        assert np.max(np.abs(Ar @ Vv - U @ B - np.outer(gf.r.get(), ekq))) < 1e-10, dev.name


def test_eigsolve_krylovdim_300_with_thick_restarts(kk, ko, ctx):
    """the restarting drivers hand the WHOLE basis to basistransform! (eigsolve/lanczos.jl:109) -- with
    krylovdim = 300 that is more than one kernel panel.  The run must restart at least once and agree with the oracle in
    values AND in the restart / operation counts; kk_basistransform / kk_householder_rmul / kk_rank1update take m > 256."""
    nx, ny = 64, 60
    n = nx * ny
    A = ko.laplacian_2d(nx, ny)                          # clustered low end: 300 steps do not converge 6 values to 1e-12
    x0 = np.random.default_rng(2).random(n)
    op = kk.SparseOperator(A, ctx, symmetric=True)
    vals, vecs, info = kk.eigsolve(op, x0, 6, "SR", krylovdim=300, maxiter=6, tol=1e-12, orth=kk.ModifiedGramSchmidt2())
    ovals, ovecs, oinfo = ko.eigsolve_lanczos(A, x0, 6, "SR", krylovdim=300, maxiter=6, tol=1e-12, orth=ko.MGS2)
    assert info.numiter > 1, "the test is meant to restart"
    assert (info.numiter, info.numops, info.converged) == (oinfo.numiter, oinfo.numops, oinfo.converged)
    assert relerr(vals[:6], ovals[:6]) < 1e-9
    expect = np.sort(ko.laplacian_2d_eigs(nx, ny))[:6]
    for lam, v, nr in zip(vals[:6], vecs[:6], info.normres[:6]):
        assert np.linalg.norm(A @ v - lam * v) <= max(10 * nr, 1e-9)
    if info.converged >= 6:
        assert relerr(vals[:6], expect) < 1e-9


def test_wide_basis_restart_primitives(kk, ko, ctx):
    """basistransform! (orthonormal.jl:291-354), rmul!(b, Householder) (dense/reflector.jl:143-154) and rank1update!
    (orthonormal.jl:210-275) on 300 columns against NumPy"""
    rng = np.random.default_rng(17)
    n, m, keep = 3000, 300, 180
    X = rng.standard_normal((n, m))
    B = kk.DeviceBasis(n, m + 2, ctx)
    for j in range(m):
        B.upload(j, X[:, j])
    B.length = m
    U, _ = np.linalg.qr(rng.standard_normal((m, m)))
    B.basistransform(U[:, :keep])
    got = np.stack([B[j].get() for j in range(keep)], 1)
    np.testing.assert_allclose(got, X @ U[:, :keep], rtol=0, atol=1e-11 * np.abs(X).max() * np.sqrt(m))
    # Householder over all 300 columns
    for j in range(m):
        B.upload(j, X[:, j])
    B.length = m
    v = rng.standard_normal(m); v /= np.linalg.norm(v)
    beta = 2.0
    B.rmul_householder(beta, v, 0, m)
    got = np.stack([B[j].get() for j in range(m)], 1)
    np.testing.assert_allclose(got, X - beta * np.outer(X @ v, v), rtol=0, atol=1e-11 * np.abs(X).max() * np.sqrt(m))
    # rank-1 update of 300 columns
    y = rng.standard_normal(n)
    xs = rng.standard_normal(m)
    B[m].set(y)
    B.rank1update(B[m], xs, 0, m, 0.7, 1.3)
    got2 = np.stack([B[j].get() for j in range(m)], 1)
    np.testing.assert_allclose(got2, 1.3 * got + 0.7 * np.outer(y, xs), rtol=0, atol=1e-11 * np.abs(got).max() * 10)
