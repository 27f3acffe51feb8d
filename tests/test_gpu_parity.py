"""GPU parity tests proper: the HIP path (through the C ABI of libkrylov_hip.so) against the CPU
oracle on the same seeded inputs.  Float64 tolerances: coefficients / norms 1e-12 relative,
Ritz values 1e-10 relative (BASELINE.json north_star), orthogonality 1e-12."""
import numpy as np
import pytest
import scipy.sparse as sp

import hostmirror_extras as hx   # mirrors of host drivers outside SURVEY section 8 (Arnoldi eigsolve, bieigsolve, geneigsolve): test infrastructure

pytestmark = pytest.mark.gpu

RTOL = 1e-12


BLOCK_FUSE_DEFAULT = 5   # kk_ctx option block_fuse as the library ships it (bit 1: CholQR2 round 2, bit 4: one-pass projection)

def orth_pairs(kk, ko):
    return [
        (kk.ClassicalGramSchmidt(), ko.CGS), (kk.ModifiedGramSchmidt(), ko.MGS),
        (kk.ClassicalGramSchmidt2(), ko.CGS2), (kk.ModifiedGramSchmidt2(), ko.MGS2),
        (kk.ClassicalGramSchmidtIR(0.75), ko.CGSIR(0.75)), (kk.ModifiedGramSchmidtIR(0.75), ko.MGSIR(0.75)),
    ]


def relerr(a, b):
    a, b = np.asarray(a, float), np.asarray(b, float)
    return float(np.max(np.abs(a - b) / np.maximum(np.abs(b), 1e-300))) if a.size else 0.0


@pytest.mark.parametrize("n", [1, 2, 7, 511, 512, 513, 1000, 4097, 100003])
def test_l1_verbs(kk, ctx, n):
    rng = np.random.default_rng(n)
    x, y = rng.standard_normal(n), rng.standard_normal(n)
    B = kk.DeviceBasis(n, 3, ctx)
    vx, vy, vz = B[0].set(x), B[1].set(y), B[2]
    assert abs(vx.inner(vy) - x @ y) <= 1e-13 * np.linalg.norm(x) * np.linalg.norm(y) + 1e-300
    assert abs(vx.norm() - np.linalg.norm(x)) <= 1e-14 * np.linalg.norm(x)
    vy.add_(vx, 0.3, -1.7)
    np.testing.assert_allclose(vy.get(), -1.7 * y + 0.3 * x, rtol=1e-15, atol=1e-15)
    vz.scale_from_(vx, 2.5)
    np.testing.assert_allclose(vz.get(), 2.5 * x, rtol=1e-15)
    vz.scale_(-0.5)
    np.testing.assert_allclose(vz.get(), -1.25 * x, rtol=1e-15)
    vz.zero_()
    assert np.all(vz.get() == 0)
    # pad rows stay zero: norm of a freshly random-filled column only sees n entries
    vz.rand_(42)
    z = vz.get()
    assert z.min() >= 0 and z.max() < 1 and abs(vz.norm() - np.linalg.norm(z)) <= 1e-13 * np.linalg.norm(z)
    vz2 = kk.DeviceBasis(n, 1, ctx)[0].rand_(42)
    assert np.array_equal(vz2.get(), z)  # counter-based: reproducible


@pytest.mark.parametrize("n,m", [(5, 1), (300, 3), (1000, 4), (5000, 17), (70000, 64), (33333, 65), (20000, 130)])
def test_project_unproject(kk, ko, ctx, n, m):
    rng = np.random.default_rng(n + m)
    V = rng.standard_normal((n, m))
    w = rng.standard_normal(n)
    B = kk.DeviceBasis(n, m + 2, ctx)
    for j in range(m):
        B.upload(j, V[:, j])
    B.length = m
    vw = B[m].set(w)
    y = B.project(vw)
    ref = V.T @ w
    assert np.max(np.abs(y - ref)) <= 1e-13 * np.linalg.norm(w) * np.sqrt(n)
    y0 = rng.standard_normal(m)
    y2 = B.project(vw, alpha=0.5, beta=2.0, y=y0.copy())
    np.testing.assert_allclose(y2, 2.0 * y0 + 0.5 * ref, rtol=1e-12, atol=1e-12)
    # sub-range
    if m > 2:
        y3 = B.project(vw, c0=1, m=m - 2)
        np.testing.assert_allclose(y3, ref[1:m - 1], rtol=1e-11, atol=1e-11)
    x = rng.standard_normal(m)
    out = B[m + 1].set(w)
    B.unproject(out, x, alpha=-1.0, beta=1.0)
    np.testing.assert_allclose(out.get(), w - V @ x, rtol=1e-12, atol=1e-12)
    B.unproject(out, x, alpha=2.0, beta=0.0)
    np.testing.assert_allclose(out.get(), 2.0 * (V @ x), rtol=1e-12, atol=1e-12)
    with pytest.raises(kk.DimensionMismatch):
        B.project(vw, y=np.zeros(m + 1))
    with pytest.raises(kk.DimensionMismatch):
        B.unproject(out, np.zeros(m + 1))


@pytest.mark.parametrize("mgs_mode", [0, 1])
@pytest.mark.parametrize("n,m", [(10, 4), (100, 30), (5000, 61), (40000, 100)])
def test_orthogonalize_all_algs(kk, ko, ctx, n, m, mgs_mode):
    """test/linalg.jl:4-25 identities + coefficient parity with the oracle."""
    ctx.set_option("mgs_mode", mgs_mode)
    rng = np.random.default_rng(1000 + n + m)
    Q, _ = np.linalg.qr(rng.standard_normal((n, min(m, n - 1))))
    m = Q.shape[1]
    # a vector with a large component inside span(Q) so that IR variants take a second pass
    w0 = Q @ rng.standard_normal(m) * 50 + rng.standard_normal(n) * 1e-3
    for dev, ref in orth_pairs(kk, ko):
        B = kk.DeviceBasis(n, m + 1, ctx)
        for j in range(m):
            B.upload(j, Q[:, j])
        B.length = m
        vw = B[m].set(w0)
        x, nrm, passes = B.orthogonalize(vw, dev)
        wr, xr = ko.orthogonalize(w0.copy(), [Q[:, j].copy() for j in range(m)], ref)
        assert relerr(x, xr) < 1e-9, (dev.name, relerr(x, xr))
        wd = vw.get()
        assert abs(nrm - np.linalg.norm(wd)) <= 1e-12 * np.linalg.norm(wd)
        # norm([r, norm(x)]) ~ norm(a)   (test/linalg.jl:12)
        assert abs(np.hypot(nrm, np.linalg.norm(x)) - np.linalg.norm(w0)) <= 1e-10 * np.linalg.norm(w0)
        if dev.is_reorth:
            assert np.max(np.abs(Q.T @ wd)) <= 1e-12 * np.linalg.norm(w0)
            np.testing.assert_allclose(wd, wr, rtol=0, atol=1e-11 * np.linalg.norm(w0))
        # orthonormalize!!
        vw.set(w0)
        x2, beta, _ = B.orthonormalize(vw, dev)
        assert abs(vw.norm() - 1) < 1e-13 and abs(beta - nrm) <= 1e-12 * nrm
    ctx.set_option("mgs_mode", 2)


def test_orthogonalize_vec(kk, ko, ctx):
    rng = np.random.default_rng(5)
    n = 3000
    q = rng.standard_normal(n)
    q /= np.linalg.norm(q)
    w0 = 30 * q + rng.standard_normal(n) * 1e-2
    for dev, ref in orth_pairs(kk, ko):
        B = kk.DeviceBasis(n, 2, ctx)
        vq, vw = B[0].set(q), B[1].set(w0)
        s, nrm = vw.orthogonalize_against_(vq, dev)
        wr, sr = ko.orthogonalize_vec(w0.copy(), q, ref)
        assert abs(s - sr) <= 1e-13 * abs(sr)
        np.testing.assert_allclose(vw.get(), wr, rtol=0, atol=1e-13 * np.linalg.norm(w0))
        assert abs(nrm - np.linalg.norm(wr)) <= 1e-11 * np.linalg.norm(wr)


def _mats(ko):
    rng = np.random.default_rng(9)
    A1 = ko.laplacian_2d(37, 23)                       # ELL, symmetric
    A2 = ko.convection_diffusion_2d(41, 17)            # ELL, nonsymmetric
    A3 = ko.sparse_random(700, 300, 9, 11)             # rectangular
    d = rng.integers(0, 60, size=500)                  # very ragged rows -> CSR path
    rows = np.repeat(np.arange(500), d)
    A4 = sp.csr_matrix((rng.standard_normal(rows.size), (rows, rng.integers(0, 500, rows.size))), shape=(500, 500))
    A5 = sp.csr_matrix(([1.5], ([0], [0])), shape=(3, 3))  # empty rows
    return [A1, A2, A3, A4, A5]


@pytest.mark.parametrize("via_csc", [False, True])
def test_spmv_formats(kk, ko, ctx, via_csc):
    rng = np.random.default_rng(3)
    for A in _mats(ko):
        op = kk.SparseOperator(A, ctx, via_csc=via_csc)
        nr, nc = A.shape
        X, Y = kk.DeviceBasis(nc, 2, ctx), kk.DeviceBasis(nr, 2, ctx)
        x, u = rng.standard_normal(nc), rng.standard_normal(nr)
        op.apply(X[0].set(x), Y[0])
        scale = np.abs(A) @ np.abs(x) + 1e-300
        assert np.max(np.abs(Y[0].get() - A @ x) / scale) < 1e-14
        op.apply_adjoint(Y[1].set(u), X[1])
        scale = np.abs(A.T) @ np.abs(u) + 1e-300
        assert np.max(np.abs(X[1].get() - A.T @ u) / scale) < 1e-14
        if nr == nc:
            op.apply_affine(X[0], Y[0], 0.7, -1.3)
            np.testing.assert_allclose(Y[0].get(), 0.7 * x - 1.3 * (A @ x), rtol=1e-12, atol=1e-12)
        with pytest.raises(kk.DimensionMismatch):
            op.apply(kk.DeviceBasis(nc + 1, 1, ctx)[0], Y[0])


@pytest.mark.parametrize("fmt", ["csr", "sell", "ell"])
def test_spmv_format_forced(kk, ko, ctx, monkeypatch, fmt):
    """Every device format (ELL, CSR, SELL-64-sigma) on the same matrices, incl. the fused Lanczos epilogue."""
    monkeypatch.setenv("KK_SPMV_FORMAT", fmt)
    rng = np.random.default_rng(1)
    d = rng.integers(0, 70, size=9000)
    rows = np.repeat(np.arange(9000), d)
    Ar = sp.csr_matrix((rng.standard_normal(rows.size), (rows, rng.integers(0, 9000, rows.size))), shape=(9000, 9000))
    for A in (ko.laplacian_2d(50, 40), (Ar + Ar.T).tocsr()):
        op = kk.SparseOperator(A, ctx, symmetric=True)
        assert op.info()["format"] == fmt.upper()
        n = A.shape[0]
        x = rng.standard_normal(n)
        X = kk.DeviceBasis(n, 2, ctx)
        op.apply(X[0].set(x), X[1])
        scale = np.abs(A) @ np.abs(x) + 1e-300
        assert np.max(np.abs(X[1].get() - A @ x) / scale) < 1e-14
        x0 = rng.random(n)
        it = kk.LanczosIterator(op, x0, kk.ClassicalGramSchmidt2(), capacity=14)
        f = kk.initialize(it)
        oit = ko.LanczosIterator(A, x0.copy(), ko.CGS2)
        of = ko.lanczos_initialize(oit)
        for _ in range(10):
            f = kk.expand_(it, f)
            of = ko.lanczos_expand(oit, of)
        assert relerr(f.alphas, of.alphas) < 1e-10 and relerr(f.betas, of.betas) < 1e-10


@pytest.mark.parametrize("tile_cols", [64, 1000])
def test_spmv_column_tiled(kk, ko, ctx, monkeypatch, tile_cols):
    """Column-tiled SELL (the format chosen for operators whose gathers have no locality; forced onto small matrices
    by shrinking the tile): plain / adjoint / affine apply, the fused Lanczos epilogue (dot modes, v_prev, norm),
    the GKL recurrence on a rectangular map, rows that are empty in some tiles, banded matrices stay untiled."""
    monkeypatch.setenv("KK_SPMV_TILE_COLS", str(tile_cols))
    rng = np.random.default_rng(21)
    n = 5000
    d = rng.integers(0, 40, size=n)
    d[::7] = 0                                                   # empty rows
    rows = np.repeat(np.arange(n), d)
    Ar = sp.csr_matrix((rng.standard_normal(rows.size), (rows, rng.integers(0, n, rows.size))), shape=(n, n))
    S = (Ar + Ar.T).tocsr()
    op = kk.SparseOperator(S, ctx, symmetric=True)
    assert op.info()["format"] == "SELL-tiled"
    if tile_cols == 1000:
        assert kk.SparseOperator(ko.laplacian_2d(100, 80), ctx).info()["format"] == "ELL+DIA const"   # banded (span 200): no tiling, stencil detected
    x = rng.standard_normal(n)
    X = kk.DeviceBasis(n, 3, ctx)
    op.apply(X[0].set(x), X[1])
    scale = np.abs(S) @ np.abs(x) + 1e-300
    assert np.max(np.abs(X[1].get() - S @ x) / scale) < 1e-14
    op.apply_affine(X[0], X[2], 0.7, -1.3)
    np.testing.assert_allclose(X[2].get(), 0.7 * x - 1.3 * (S @ x), rtol=1e-12, atol=1e-12)
    x0 = rng.random(n)
    for dev, ref in ((kk.ClassicalGramSchmidt2(), ko.CGS2), (kk.ModifiedGramSchmidt2(), ko.MGS2)):
        it = kk.LanczosIterator(op, x0, dev, capacity=14)
        f = kk.initialize(it)
        oit = ko.LanczosIterator(S, x0.copy(), ref)
        of = ko.lanczos_initialize(oit)
        for _ in range(10):
            f = kk.expand_(it, f)
            of = ko.lanczos_expand(oit, of)
        assert relerr(f.alphas, of.alphas) < 1e-10 and relerr(f.betas, of.betas) < 1e-10
    # rectangular map and its transpose (both tiled), GKL recurrence
    R = sp.random(4000, 3500, density=0.01, random_state=5, format="csr")
    rop = kk.SparseOperator(R, ctx)
    assert rop.info()["format"] == "SELL-tiled"
    U, V = kk.DeviceBasis(4000, 1, ctx), kk.DeviceBasis(3500, 1, ctx)
    u = rng.standard_normal(4000)
    rop.apply_adjoint(U[0].set(u), V[0])
    scale = np.abs(R.T) @ np.abs(u) + 1e-300
    assert np.max(np.abs(V[0].get() - R.T @ u) / scale) < 1e-14
    u0 = rng.random(4000)
    it = kk.GKLIterator(rop, u0, kk.ModifiedGramSchmidt2(), capacity=12)
    f = kk.initialize(it)
    oit = ko.GKLIterator(R, u0.copy(), ko.MGS2)
    of = ko.gkl_initialize(oit)
    for _ in range(8):
        f = kk.expand_(it, f)
        of = ko.gkl_expand(oit, of)
    assert relerr(f.alphas, of.alphas) < 1e-10 and relerr(f.betas, of.betas) < 1e-10


@pytest.mark.parametrize("mgs_mode", [0, 1])
def test_lanczos_factorization(kk, ko, ctx, mgs_mode):
    """test/factorize.jl:140-148 invariants after every expand! + (alpha, beta) parity with the oracle."""
    ctx.set_option("mgs_mode", mgs_mode)
    nx, ny = 40, 30
    n = nx * ny
    A = ko.laplacian_2d(nx, ny, shift_diag=10 * np.linspace(0, 1, n) ** 2)
    x0 = np.random.default_rng(3).random(n)
    steps = 25
    for dev, ref in orth_pairs(kk, ko):
        it = kk.LanczosIterator(kk.SparseOperator(A, ctx, symmetric=True), x0, dev, capacity=steps + 3)
        fact = kk.initialize(it)
        oit = ko.LanczosIterator(A, x0.copy(), ref)
        ofact = ko.lanczos_initialize(oit)
        for _ in range(steps):
            fact = kk.expand_(it, fact)
            ofact = ko.lanczos_expand(oit, ofact)
            k = len(fact)
            V = fact.V.to_numpy()
            r = fact.r.get()
            T = np.diag(fact.alphas) + np.diag(fact.betas[:-1], 1) + np.diag(fact.betas[:-1], -1)
            ek = np.zeros(k); ek[-1] = 1
            assert abs(np.linalg.norm(r) - fact.normres) <= 1e-12 * max(1, fact.normres)
            assert np.max(np.abs(A @ V - V @ T - np.outer(r, ek))) < 1e-11
            if dev.is_reorth:
                assert np.max(np.abs(V.T @ V - np.eye(k))) < 1e-12
        tol = 1e-10 if dev.is_reorth else 1e-6  # plain CGS/MGS Lanczos loses orthogonality: trajectories drift
        assert relerr(fact.alphas, ofact.alphas) < tol, dev.name
        assert relerr(fact.betas, ofact.betas) < tol, dev.name
        # shrink! (lanczos.jl:273-291)
        fact = kk.shrink_(fact, 10)
        ofact = ko.lanczos_shrink(ofact, 10)
        assert len(fact) == 10 and len(fact.V) == 10
        np.testing.assert_allclose(fact.r.get(), ofact.r, rtol=0, atol=1e-6 if not dev.is_reorth else 1e-10)
    ctx.set_option("mgs_mode", 2)


def test_lanczos_keepvecs_false(kk, ko, ctx):
    n = 900
    A = ko.laplacian_2d(30, 30)
    x0 = np.random.default_rng(8).random(n)
    for dev, ref in orth_pairs(kk, ko)[:2]:
        it = kk.LanczosIterator(kk.SparseOperator(A, ctx, symmetric=True), x0, dev, keepvecs=False, capacity=6)
        fact = kk.initialize(it)
        oit = ko.LanczosIterator(A, x0.copy(), ref, keepvecs=False)
        ofact = ko.lanczos_initialize(oit)
        for _ in range(12):
            fact = kk.expand_(it, fact)
            ofact = ko.lanczos_expand(oit, ofact)
        assert len(fact.V) == 1 and len(ofact.V) == 1
        assert relerr(fact.alphas, ofact.alphas) < 1e-9 and relerr(fact.betas, ofact.betas) < 1e-9
    with pytest.raises(ValueError):
        kk.LanczosIterator(None, x0, kk.ModifiedGramSchmidt2(), keepvecs=False)


@pytest.mark.parametrize("mgs_mode", [0, 1])
def test_arnoldi_factorization(kk, ko, ctx, mgs_mode):
    """test/factorize.jl:185-193: V'V = I, A V = V H + r e', norm(r) = beta, after every expand!."""
    ctx.set_option("mgs_mode", mgs_mode)
    A = ko.convection_diffusion_2d(30, 20)
    n = A.shape[0]
    x0 = np.random.default_rng(4).random(n)
    steps = 20
    for dev, ref in orth_pairs(kk, ko):
        it = kk.ArnoldiIterator(kk.SparseOperator(A, ctx), x0, dev, capacity=steps + 3)
        fact = kk.initialize(it)
        oit = ko.ArnoldiIterator(A, x0.copy(), ref)
        ofact = ko.arnoldi_initialize(oit)
        for _ in range(steps):
            fact = kk.expand_(it, fact)
            ofact = ko.arnoldi_expand(oit, ofact)
        k = len(fact)
        V, r, H = fact.V.to_numpy(), fact.r.get(), fact.rayleighquotient()
        ek = np.zeros(k); ek[-1] = 1
        assert np.max(np.abs(A @ V - V @ H - np.outer(r, ek))) < 1e-11
        assert abs(np.linalg.norm(r) - fact.normres) < 1e-12
        assert np.max(np.abs(V.T @ V - np.eye(k))) < (1e-12 if dev.is_reorth else 1e-8)
        # the bar the path is held to: 1e-10 relative to |H| for the re-orthogonalising algorithms (entries that are
        # themselves rounding residue of an orthogonalisation have no relative accuracy of their own)
        dH = np.max(np.abs(np.array(fact.H) - np.array(ofact.H))) / np.max(np.abs(ofact.H))
        assert dH < (1e-10 if dev.is_reorth else 1e-5), (dev.name, dH)
        fact = kk.shrink_(fact, 7)
        ofact = ko.arnoldi_shrink(ofact, 7)
        assert len(fact.H) == len(ofact.H) and len(fact.V) == 7
    ctx.set_option("mgs_mode", 2)


@pytest.mark.parametrize("mgs_mode", [0, 1])
def test_gkl_factorization(kk, ko, ctx, mgs_mode):
    """test/factorize.jl:285-296: A V = U B + r e', A'U = V B', U'U = I, V'V = I."""
    ctx.set_option("mgs_mode", mgs_mode)
    A = ko.sparse_random(600, 250, 8, 21)
    u0 = np.random.default_rng(6).random(600)
    steps = 15
    for dev, ref in orth_pairs(kk, ko):
        it = kk.GKLIterator(kk.SparseOperator(A, ctx), u0, dev, capacity=steps + 3)
        fact = kk.initialize(it)
        oit = ko.GKLIterator(A, u0.copy(), ref)
        ofact = ko.gkl_initialize(oit)
        for _ in range(steps):
            fact = kk.expand_(it, fact)
            ofact = ko.gkl_expand(oit, ofact)
        k = len(fact)
        U, V, r, B = fact.U.to_numpy(), fact.V.to_numpy(), fact.r.get(), fact.rayleighquotient()
        ek = np.zeros(k); ek[-1] = 1
        assert np.max(np.abs(A @ V - U @ B - np.outer(r, ek))) < 1e-11
        assert np.max(np.abs(A.T @ U - V @ B.T)) < 1e-10
        if dev.name in ("mgs2", "cgsir", "mgsir"):
            assert np.max(np.abs(U.T @ U - np.eye(k))) < 1e-12
            assert np.max(np.abs(V.T @ V - np.eye(k))) < 1e-12
        tol = 1e-10 if dev.is_reorth else 1e-6
        assert relerr(fact.alphas, ofact.alphas) < tol and relerr(fact.betas, ofact.betas) < tol
        fact = kk.shrink_(fact, 5)
        assert len(fact.U) == 5 and len(fact.V) == 5
    ctx.set_option("mgs_mode", 2)


def test_restart_kernels(kk, ko, ctx):
    """test/linalg.jl:27-44: Givens / Householder on a basis agree with dense rmul!; basistransform!."""
    rng = np.random.default_rng(12)
    n, m = 3001, 23
    V = rng.standard_normal((n, m))
    B = kk.DeviceBasis(n, m + 1, ctx)
    for j in range(m):
        B.upload(j, V[:, j])
    B.length = m
    c, s = np.cos(0.3), np.sin(0.3)
    B.rmul_givens(2, 5, c, s)
    Vr = V.copy()
    Vr[:, 2], Vr[:, 5] = c * V[:, 2] - s * V[:, 5], s * V[:, 2] + c * V[:, 5]
    np.testing.assert_allclose(B.to_numpy(), Vr, rtol=1e-14, atol=1e-14)
    hv = rng.standard_normal(9); hv[3] = 1.0
    beta = 2.0 / (hv @ hv)
    B.rmul_householder(beta, hv, 4, 9)
    w = Vr[:, 4:13] @ hv
    Vr[:, 4:13] -= np.outer(w, beta * hv)
    np.testing.assert_allclose(B.to_numpy(), Vr, rtol=1e-13, atol=1e-12)
    U, _ = np.linalg.qr(rng.standard_normal((m, m)))
    keep = 14
    B.basistransform(U[:, :keep])
    Vt = Vr.copy()
    Vt[:, :keep] = Vr @ U[:, :keep]
    np.testing.assert_allclose(B.to_numpy(), Vt, rtol=1e-12, atol=1e-12)
    y = B[m].set(rng.standard_normal(n))
    x = rng.standard_normal(5)
    B.rank1update(y, x, c0=3, m=5, alpha=0.5, beta=1.0)
    Vt[:, 3:8] += 0.5 * np.outer(y.get(), x)
    np.testing.assert_allclose(B.to_numpy(), Vt, rtol=1e-12, atol=1e-12)
    with pytest.raises(kk.DimensionMismatch):
        B.basistransform(U[:5, :3])


@pytest.mark.parametrize("which", ["LM", "SR", "LR"])
def test_eigsolve_lanczos(kk, ko, ctx, which):
    """test/eigsolve.jl:74,122-123: values vs dense eigvals, residual identity; counts equal the oracle's."""
    nx, ny = 24, 18
    n = nx * ny
    A = ko.laplacian_2d(nx, ny, shift_diag=10 * np.linspace(0, 1, n) ** 2)
    ev = np.linalg.eigvalsh(A.toarray())
    x0 = np.random.default_rng(2).random(n)
    for dev, ref in orth_pairs(kk, ko)[2:]:
        vals, vecs, info = kk.eigsolve(kk.SparseOperator(A, ctx, symmetric=True), x0, 4, which,
                                       krylovdim=30, maxiter=200, tol=1e-11, orth=dev)
        ovals, ovecs, oinfo = ko.eigsolve_lanczos(A, x0, 4, which, krylovdim=30, maxiter=200, tol=1e-11, orth=ref)
        assert info.converged >= 4
        key = {"LM": lambda d: -np.abs(d), "SR": lambda d: d, "LR": lambda d: -d}[which]
        expect = ev[np.argsort(key(ev), kind="stable")][: len(vals)]
        assert relerr(vals, expect) < 1e-10
        assert relerr(vals[:4], ovals[:4]) < 1e-10
        assert (info.numiter, info.numops) == (oinfo.numiter, oinfo.numops)
        for lam, v in zip(vals, vecs):
            assert np.linalg.norm(A @ v - lam * v) < 1e-9
        Vm = np.stack(vecs, 1)
        assert np.max(np.abs(Vm.T @ Vm - np.eye(len(vecs)))) < 1e-10


def test_linsolve_gmres(kk, ko, ctx):
    """test/linsolve.jl:135,230: b ~ (a0 + a1 A) x; numops / numiter equal the oracle's (BASELINE north_star)."""
    A = ko.convection_diffusion_2d(40, 25)
    n = A.shape[0]
    b = np.random.default_rng(4).random(n)
    for a0, a1 in ((0.0, 1.0), (0.3, 0.9)):
        for dev, ref in orth_pairs(kk, ko):
            tol = 1e-10 * np.linalg.norm(b)
            x, info = kk.linsolve(kk.SparseOperator(A, ctx), b, None, kk.GMRES(dev, 20, 25, tol), a0, a1)
            xo, oinfo = ko.gmres(A, b, None, a0, a1, krylovdim=25, maxiter=20, tol=tol, orth=ref)
            assert info.converged == 1 and oinfo.converged == 1
            assert (info.numiter, info.numops) == (oinfo.numiter, oinfo.numops), dev.name
            assert abs(info.normres - oinfo.normres) <= 1e-3 * tol  # final explicit residual: cancellation-level noise
            assert np.linalg.norm(a0 * x + a1 * (A @ x) - b) <= 1.01 * tol
            np.testing.assert_allclose(x, xo, rtol=0, atol=1e-8 * np.linalg.norm(xo))
    # started from the solution: numops == 1 (test/linsolve.jl:167)
    xs = np.random.default_rng(1).random(n)
    x, info = kk.linsolve(kk.SparseOperator(A, ctx), A @ xs, xs, kk.GMRES(tol=1e-8))
    assert info.numops == 1 and info.converged == 1


def test_svdsolve_gkl(kk, ko, ctx):
    """test/svdsolve.jl:14,61-63,89: S ~ svdvals, U'U = I, A V ~ U S, A'U ~ V S."""
    A = ko.sparse_random(400, 180, 10, 31)
    sv = np.linalg.svd(A.toarray(), compute_uv=False)
    u0 = np.random.default_rng(6).random(400)
    for dev, ref in orth_pairs(kk, ko)[2:]:
        S, L, R, info = kk.svdsolve(kk.SparseOperator(A, ctx), u0, 5, "LR", krylovdim=20, maxiter=100, tol=1e-10, orth=dev)
        So, Lo, Ro, oinfo = ko.svdsolve_gkl(A, u0, 5, "LR", krylovdim=20, maxiter=100, tol=1e-10, orth=ref)
        assert info.converged >= 5
        assert relerr(S[:5], sv[:5]) < 1e-10 and relerr(S[:5], So[:5]) < 1e-10
        assert (info.numiter, info.numops) == (oinfo.numiter, oinfo.numops)
        Lm, Rm = np.stack(L, 1), np.stack(R, 1)
        assert np.max(np.abs(Lm.T @ Lm - np.eye(len(S)))) < 1e-10
        assert np.max(np.abs(A @ Rm - Lm * S)) < 1e-8
        assert np.max(np.abs(A.T @ Lm - Rm * S)) < 1e-8


def test_error_behaviour(kk, ctx):
    B = kk.DeviceBasis(10, 3, ctx)
    with pytest.raises(kk.KrylovHipError):
        B.upload(5, np.zeros(10))
    with pytest.raises(kk.DimensionMismatch):
        B.upload(0, np.zeros(11))
    A = sp.identity(10, format="csr")
    it = kk.LanczosIterator(kk.SparseOperator(A, ctx, symmetric=True), np.zeros(10), kk.ModifiedGramSchmidt2(), capacity=4)
    with pytest.raises(kk.KrylovHipError) as e:
        kk.initialize(it)
    assert "norm zero" in str(e.value)


def test_dist_hip_backend_world1(kk, ko, ctx):
    """The caller-owned-communicator mechanisms of the C ABI (split-phase kk_*_dev entry points, hooks) on the real device
    with torch.distributed over RCCL, world_size 1 (exercisers: tests/splitphase_dist.py): torch device tensors handed to
    the split-phase C entry points, all-reduce plumbing, stream sharing."""
    import os
    import socket
    import sys
    from pathlib import Path
    import torch
    import torch.distributed as dist
    sys.path.insert(0, str(Path(__file__).resolve().parent))
    import splitphase_dist as kd

    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        nx, ny = 40, 30
        n = nx * ny
        A = ko.laplacian_2d(nx, ny, shift_diag=10 * np.linspace(0, 1, n) ** 2)
        x0 = np.random.default_rng(3).random(n)
        be = kd.HipBackend(0)
        part = kd.Partition.even(n, 1, 0, align=nx)
        op = kd.DistSparseOperator(A, part, be)
        for dev, ref in ((kk.ClassicalGramSchmidt2(), ko.CGS2), (kk.ModifiedGramSchmidt2(), ko.MGS2),
                         (kk.ClassicalGramSchmidt(), ko.CGS), (kk.ModifiedGramSchmidt(), ko.MGS),
                         (kk.ClassicalGramSchmidtIR(), ko.CGSIR()), (kk.ModifiedGramSchmidtIR(), ko.MGSIR())):
            it = kd.DistLanczosIterator(op, x0, dev, capacity=24)
            f = it.initialize()
            oit = ko.LanczosIterator(A, x0.copy(), ref)
            of = ko.lanczos_initialize(oit)
            for _ in range(20):
                f = it.expand(f)
                of = ko.lanczos_expand(oit, of)
            tol = 1e-10 if dev.is_reorth else 1e-6
            assert relerr(f.alphas, of.alphas) < tol and relerr(f.betas, of.betas) < tol, dev.name
            if dev.is_reorth:
                V = f.V.to_numpy()
                assert np.max(np.abs(V.T @ V - np.eye(V.shape[1]))) < 1e-12
        # the generic eigsolve driver over the sharded iterator (thick restarts included)
        it = kd.DistLanczosIterator(op, x0, kk.ModifiedGramSchmidt2(), capacity=32)
        vals, vecs, info = kk.eigsolve(None, None, 3, "SR", kk.Lanczos(krylovdim=30, tol=1e-10, maxiter=100), iterator=it)
        ev = np.linalg.eigvalsh(A.toarray())
        assert info.converged >= 3 and relerr(vals[:3], ev[:3]) < 1e-10
        # sharded GKL (config 4): ghost-only local operator fed from the gathered v, transposed SpMV + scatter
        Ar = ko.sparse_random(600, 250, 8, 21)
        u0 = np.random.default_rng(6).random(600)
        rop = kd.DistRectOperator(Ar, kd.Partition.even(600, 1, 0), kd.Partition.even(250, 1, 0), be)
        for dev, ref in ((kk.ClassicalGramSchmidt2(), ko.CGS2), (kk.ModifiedGramSchmidt2(), ko.MGS2),
                         (kk.ClassicalGramSchmidtIR(0.9999), ko.CGSIR(0.9999)), (kk.ModifiedGramSchmidtIR(0.9999), ko.MGSIR(0.9999))):
            git = kd.DistGKLIterator(rop, u0, dev, capacity=18)
            gf = git.initialize()
            oit = ko.GKLIterator(Ar, u0.copy(), ref)
            of = ko.gkl_initialize(oit)
            for _ in range(14):
                gf = git.expand(gf)
                of = ko.gkl_expand(oit, of)
            assert relerr(gf.alphas, of.alphas) < 1e-10 and relerr(gf.betas, of.betas) < 1e-10, dev.name
            Um, Vm = gf.U.to_numpy(), gf.V.to_numpy()
            Bm = np.diag(gf.alphas) + np.diag(gf.betas[:-1], -1)
            assert np.max(np.abs(Ar.T @ Um - Vm @ Bm.T)) < 1e-10
        # hook-based sharding over torch.distributed (RCCL), world 1: callback plumbing / pointer resolution
        sctx = kd.ShardedContext(kd.TorchCollective(), 0, backend=be)
        hop = sctx.operator(A, part)
        vals, vecs, info = kk.eigsolve(hop, x0, 2, "SR", kk.Lanczos(krylovdim=30, tol=1e-10, maxiter=100))
        assert relerr(vals[:2], ev[:2]) < 1e-10 and sctx.calls > 100
        sctx.close()
    finally:
        dist.destroy_process_group()


# ------------------------------------------------------------------ BlockLanczos (config 5)
@pytest.mark.parametrize("block_mode", [0, 1])
def test_block_primitives(kk, ko, ctx, block_mode):
    """test/block.jl:74-86 (block_inner), :105-124 (block_reorthogonalize!), :142-161 (block_qr! rank deficiency)."""
    ctx.set_option("block_mode", block_mode)
    rng = np.random.default_rng(31)
    for n, p, q in ((100, 6, 6), (5000, 20, 7), (70001, 37, 16), (3000, 130, 3)):
        X, Y = rng.standard_normal((n, p)), rng.standard_normal((n, q))
        S = kk.DeviceBasis(n, p + q, ctx)
        for j in range(p):
            S.upload(j, X[:, j])
        for j in range(q):
            S.upload(p + j, Y[:, j])
        M = kk.block_inner(kk.Block(S, 0, p), kk.Block(S, p, q))
        ref = X.T @ Y
        assert np.max(np.abs(M - ref)) <= 1e-13 * np.sqrt(n) * np.max(np.abs(ref)) + 1e-12, (n, p, q)
    # block_reorthogonalize!
    n, m, q = 4000, 45, 9
    Qm, _ = np.linalg.qr(rng.standard_normal((n, m)))
    W = rng.standard_normal((n, q))
    S = kk.DeviceBasis(n, m + q, ctx)
    for j in range(m):
        S.upload(j, Qm[:, j])
    for j in range(q):
        S.upload(m + j, W[:, j])
    S.length = m
    kk.block_reorthogonalize_(kk.Block(S, m, q), S)
    Wd = np.stack([S.download(m + j) for j in range(q)], 1)
    assert np.linalg.norm(Qm.T @ Wd) < 1e-11 * np.linalg.norm(W)
    Wo = ko.block_reorthogonalize([W[:, j].copy() for j in range(q)], [Qm[:, j] for j in range(m)])
    np.testing.assert_allclose(Wd, np.stack(Wo, 1), atol=1e-11 * np.linalg.norm(W))
    # block_qr! with dependent columns, in place and out of place
    n, p = 3000, 6
    A = [rng.standard_normal(n) for _ in range(p)]
    Cc = A + [A[0] + 2 * A[1], A[2] - A[3]]
    Cm = np.stack(Cc, 1)
    for out_col in (None, 10):
        S = kk.DeviceBasis(n, 20, ctx)
        for j in range(p + 2):
            S.upload(j, Cm[:, j])
        R, good, drift = kk.block_qr_(kk.Block(S, 0, p + 2), 1e-10, out_col)
        Ro, goodo, drifto = ko.block_qr([c.copy() for c in Cc], 1e-10)
        assert good == goodo and R.shape == Ro.shape == (p, p + 2)
        c0 = 0 if out_col is None else out_col
        Qd = np.stack([S.download(c0 + j) for j in range(len(good))], 1)
        np.testing.assert_allclose(Qd @ R, Cm, atol=1e-10)
        assert np.max(np.abs(Qd.T @ Qd - np.eye(p))) < 1e-12
        np.testing.assert_allclose(R, Ro, atol=1e-9 * np.max(np.abs(Ro)))
        if out_col is not None:  # input block untouched (serves as Rcopy)
            np.testing.assert_array_equal(S.download(3), Cm[:, 3])
    ctx.set_option("block_mode", 1)


@pytest.mark.parametrize("block_mode", [0, 1])
def test_block_apply_update(kk, ko, ctx, block_mode):
    ctx.set_option("block_mode", block_mode)
    rng = np.random.default_rng(2)
    for A in (ko.laplacian_2d(37, 23), ko.sparse_random(500, 500, 9, 3)):
        n = A.shape[0]
        for nb in (1, 5, 16, 20):
            X = rng.standard_normal((n, nb))
            S = kk.DeviceBasis(n, 2 * nb, ctx)
            for j in range(nb):
                S.upload(j, X[:, j])
            op = kk.SparseOperator(A, ctx)
            from krylovkit_hip._lib import check
            check(S._lib.kk_block_apply(op.handle, S.handle, 0, S.handle, nb, nb))
            Y = np.stack([S.download(nb + j) for j in range(nb)], 1)
            np.testing.assert_allclose(Y, A @ X, rtol=1e-12, atol=1e-12)
    n, m, q = 2001, 23, 19
    V, W, Sm = rng.standard_normal((n, m)), rng.standard_normal((n, q)), rng.standard_normal((m, q))
    S = kk.DeviceBasis(n, m + q, ctx)
    for j in range(m):
        S.upload(j, V[:, j])
    for j in range(q):
        S.upload(m + j, W[:, j])
    from krylovkit_hip._lib import check, c_dp
    Sf = np.asfortranarray(Sm)
    norms = np.zeros(q)
    check(S._lib.kk_block_update(S.handle, m, q, S.handle, 0, m, Sf.ctypes.data_as(c_dp), m, -0.5, 2.0, norms.ctypes.data_as(c_dp)))
    ref = 2.0 * W - 0.5 * V @ Sm
    Wd = np.stack([S.download(m + j) for j in range(q)], 1)
    np.testing.assert_allclose(Wd, ref, rtol=1e-12, atol=1e-12)
    np.testing.assert_allclose(norms, np.linalg.norm(ref, axis=0), rtol=1e-12)
    ctx.set_option("block_mode", 1)


def _grid_stencil(nx, ny, nine, rng, drop_tail=0):
    """5- or 9-point operator with row-dependent coefficients on an nx x ny grid in natural ordering (offsets
    {-nx, 0, nx} + {-1, 0, 1}); `drop_tail` removes the last rows / columns so n is not a multiple of nx."""
    n = nx * ny
    rows, cols, vals = [], [], []
    shifts = [(-1, 0), (0, -1), (0, 0), (0, 1), (1, 0)] + ([(-1, -1), (-1, 1), (1, -1), (1, 1)] if nine else [])
    ix, iy = np.meshgrid(np.arange(nx), np.arange(ny))
    ix, iy = ix.ravel(), iy.ravel()
    for dy, dx in shifts:
        ok = (ix + dx >= 0) & (ix + dx < nx) & (iy + dy >= 0) & (iy + dy < ny)
        r = (iy * nx + ix)[ok]
        rows.append(r); cols.append(r + dy * nx + dx); vals.append(rng.standard_normal(r.size))
    A = sp.csr_matrix((np.concatenate(vals), (np.concatenate(rows), np.concatenate(cols))), shape=(n, n))
    if drop_tail:
        A = A[:n - drop_tail, :n - drop_tail].tocsr()
    return A


@pytest.mark.parametrize("nine", [False, True])
def test_block_apply_grid_stencil_sweep(kk, ko, ctx, nine):
    """apply(f, ::Block) on grid stencils goes through the sweeping diagonal kernel (k_spmm_dia: three-line register window,
    +-1 neighbours by wave shifts) -- against SciPy and against the gather kernel, for line lengths around the 62-position
    strip, block widths 2..16 and a truncated last line (coefficients are random, i.e. non-symmetric operators)."""
    from krylovkit_hip._lib import check
    rng = np.random.default_rng(17 + nine)
    for nx, ny, drop in ((64, 70, 0), (125, 40, 0), (187, 30, 11), (66, 80, 0), (124, 40, 0), (300, 20, 299)):
        A = _grid_stencil(nx, ny, nine, rng, drop)
        n = A.shape[0]
        op = kk.SparseOperator(A, ctx)
        for nb in (2, 3, 5, 8, 13, 16):
            X = rng.standard_normal((n, nb))
            S = kk.DeviceBasis(n, 2 * nb, ctx)
            for j in range(nb):
                S.upload(j, X[:, j])
            for dia in (1, 0):
                ctx.set_option("spmm_dia", dia)
                ctx.prof_reset(); ctx.prof_enable(1)
                check(S._lib.kk_block_apply(op.handle, S.handle, 0, S.handle, nb, nb))
                ctx.prof_enable(0)
                Y = np.stack([S.download(nb + j) for j in range(nb)], 1)
                np.testing.assert_allclose(Y, A @ X, rtol=1e-12, atol=1e-12, err_msg=f"nx={nx} nb={nb} dia={dia}")
                assert (ctx.prof_get("k_spmm_dia")[1] > 0) == bool(dia), "the sweeping kernel must be the one that ran"
            ctx.set_option("spmm_dia", 1)
            S.free()
    # no stencil structure -> gather kernel, silently
    R = ko.sparse_random(5000, 5000, 7, 5)
    op = kk.SparseOperator(R, ctx)
    S = kk.DeviceBasis(5000, 8, ctx)
    X = rng.standard_normal((5000, 4))
    for j in range(4):
        S.upload(j, X[:, j])
    ctx.prof_reset(); ctx.prof_enable(1)
    check(S._lib.kk_block_apply(op.handle, S.handle, 0, S.handle, 4, 4))
    ctx.prof_enable(0)
    assert ctx.prof_get("k_spmm_dia")[1] == 0
    np.testing.assert_allclose(np.stack([S.download(4 + j) for j in range(4)], 1), R @ X, rtol=1e-12, atol=1e-12)


def test_block_apply_constant_stencil_aligned_sweep(kk, ko, ctx):
    """the aligned form of the sweeping multi-column apply (k_spmm_dia_al: 16-byte window loads, +-1 neighbours from the lanes next
    door, next line in flight, 2 / 4 columns per wave) against SciPy and BITWISE against the 8-byte form, for value-free 5-point
    stencils with four different off-diagonal coefficients (a slot mix-up cannot hide), line lengths around the 128-position strip,
    block widths with and without a remainder group, sweeps of 1 / 4 / 5 / 16 lines per wave; an odd line length falls back silently"""
    from krylovkit_hip._lib import check
    rng = np.random.default_rng(29)
    al_default = ctx.get_option("spmm_dia_al")
    try:
        for nx, ny in ((64, 70), (128, 40), (130, 33), (254, 21), (256, 19), (1000, 12), (66, 80), (65, 80)):
            A = ko.convection_diffusion_2d(nx, ny) if nx != 128 else ko.laplacian_2d(nx, ny)
            n = A.shape[0]
            op = kk.SparseOperator(A, ctx)
            assert "const" in op.info()["format"], op.info()
            for nb in (16, 4, 5, 3, 9):
                X = rng.standard_normal((n, nb))
                S = kk.DeviceBasis(n, 2 * nb, ctx)
                for j in range(nb):
                    S.upload(j, X[:, j])
                ref = None
                for al, lines in ((0, 16), (4, 16), (2, 4), (4, 1), (2, 5)):
                    ctx.set_option("spmm_dia_al", al); ctx.set_option("spmm_dia_al_lines", lines)
                    for j in range(nb):
                        S[nb + j].rand_(5 + j)          # (stale output must not pass for output)
                    l0 = ctx.get_option("spmm_dia_al_launches")
                    ctx.prof_reset(); ctx.prof_enable(1)
                    check(S._lib.kk_block_apply(op.handle, S.handle, 0, S.handle, nb, nb))
                    ctx.prof_enable(0)
                    assert ctx.prof_get("k_spmm_dia")[1] > 0
                    took = ctx.get_option("spmm_dia_al_launches") - l0
                    assert took == (1 if (al and nx % 2 == 0 and nb >= al) else 0), (nx, nb, al, took)
                    Y = np.stack([S.download(nb + j) for j in range(nb)], 1)
                    if ref is None:
                        ref = Y
                        np.testing.assert_allclose(Y, A @ X, rtol=1e-13, atol=1e-13)
                    else:
                        assert np.array_equal(Y, ref), (nx, ny, nb, al, lines, float(np.max(np.abs(Y - ref))))
                S.free()
    finally:
        ctx.set_option("spmm_dia_al", al_default); ctx.set_option("spmm_dia_al_lines", 4)


def test_grid_stencil_single_vector_apply_and_fused_epilogues(kk, ko, ctx):
    """The diagonal SpMV of detected grid stencils (k_spmv_dia) against SciPy, the affine form, and -- through the fused
    Lanczos / CG epilogues (inner products, v_prev term, norms, speculative scaled apply) -- against the oracle; every case also
    with the gather kernel (spmv_dia = 0): the two must agree to rounding."""
    rng = np.random.default_rng(23)
    for nine in (False, True):
        for nx, ny, drop in ((64, 70, 0), (131, 37, 0), (187, 30, 11)):
            A = _grid_stencil(nx, ny, nine, rng, drop)
            n = A.shape[0]
            op = kk.SparseOperator(A, ctx)
            assert op.info()["format"].startswith("ELL+DIA")
            X, Y = kk.DeviceBasis(n, 2, ctx), kk.DeviceBasis(n, 2, ctx)
            x, u = rng.standard_normal(n), rng.standard_normal(n)
            for dia in (1, 0):
                ctx.set_option("spmv_dia", dia)
                ctx.prof_reset(); ctx.prof_enable(1)
                op.apply(X[0].set(x), Y[0])
                ctx.prof_enable(0)
                assert (ctx.prof_get("k_spmv_dia")[1] > 0) == bool(dia)
                scale = np.abs(A) @ np.abs(x) + 1e-300
                assert np.max(np.abs(Y[0].get() - A @ x) / scale) < 1e-14
                op.apply_adjoint(Y[1].set(u), X[1])
                scale = np.abs(A.T) @ np.abs(u) + 1e-300
                assert np.max(np.abs(X[1].get() - A.T @ u) / scale) < 1e-14
                op.apply_affine(X[0], Y[0], 0.7, -1.3)
                np.testing.assert_allclose(Y[0].get(), 0.7 * x - 1.3 * (A @ x), rtol=1e-12, atol=1e-12)
                if dia:   # 1 / 2 row pairs per lane of the diagonal kernel (4 falls back to 2 when the values are streamed): same bits
                    ref = Y[0].get()
                    for pairs in (1, 2, 4):
                        ctx.set_option("spmv_dia_pairs", pairs)
                        op.apply_affine(X[0], Y[0], 0.7, -1.3)
                        assert np.array_equal(Y[0].get(), ref), (nine, nx, ny, pairs)
                    ctx.set_option("spmv_dia_pairs", 0)
            ctx.set_option("spmv_dia", 1)
    A = ko.laplacian_2d(70, 64, shift_diag=10 * np.linspace(0, 1, 70 * 64) ** 2)
    n = A.shape[0]
    x0 = rng.random(n)
    for dia in (1, 0):
        ctx.set_option("spmv_dia", dia)
        for dev, ref in ((kk.ModifiedGramSchmidt2(), ko.MGS2), (kk.ClassicalGramSchmidt2(), ko.CGS2), (kk.ModifiedGramSchmidt(), ko.MGS)):
            it = kk.LanczosIterator(kk.SparseOperator(A, ctx, symmetric=True), x0, dev, capacity=30)
            oit = ko.LanczosIterator(A, x0.copy(), ref)
            f, of = kk.initialize(it), ko.lanczos_initialize(oit)
            for _ in range(25):
                f, of = kk.expand_(it, f), ko.lanczos_expand(oit, of)
            tol = 1e-10 if dev.is_reorth else 1e-6
            assert relerr(f.alphas, of.alphas) < tol and relerr(f.betas, of.betas) < tol, (dia, dev.name)
        b = rng.random(n)
        tol = 1e-10 * np.linalg.norm(b)
        xs, info = kk.linsolve_cg(kk.SparseOperator(A, ctx, symmetric=True), b, None, kk.CG(2000, tol), 0.3, 0.9)
        xo, oinfo = ko.cg(A, b, None, 0.3, 0.9, maxiter=2000, tol=tol)
        assert info.converged == 1 and (info.numiter, info.numops) == (oinfo.numiter, oinfo.numops)
        np.testing.assert_allclose(xs, xo, rtol=0, atol=1e-9 * np.linalg.norm(xo))
    ctx.set_option("spmv_dia", 1)


@pytest.mark.parametrize("block_mode", [0, 1])
def test_blocklanczos_factorization(kk, ko, ctx, block_mode):
    """test/factorize.jl:387-401: V'V = I, A V = V H + R B' after every expand!; parity of H with the oracle."""
    ctx.set_option("block_mode", block_mode)
    nx, ny, bs = 30, 20, 4
    n = nx * ny
    A = ko.laplacian_2d(nx, ny, shift_diag=10 * np.linspace(0, 1, n) ** 2)
    rng = np.random.default_rng(7)
    x0 = [rng.random(n) for _ in range(bs)]
    it = kk.BlockLanczosIterator(kk.SparseOperator(A, ctx, symmetric=True), x0, 40 + bs)
    f = it.initialize()
    oit = ko.BlockLanczosIterator(A, [x.copy() for x in x0], 40 + bs)
    of = ko.blocklanczos_initialize(oit)
    for _ in range(8):
        f = it.expand(f)
        of = ko.blocklanczos_expand(oit, of)
        k = len(f)
        assert k == len(of) and f.R_size == of.R_size
        V = f.V.to_numpy()
        R = np.stack([f.residual()[j].get() for j in range(f.R_size)], 1)
        H = f.H[:k, :k]
        E = np.zeros((k, f.R_size)); E[k - f.R_size:, :] = np.eye(f.R_size)
        assert np.max(np.abs(V.T @ V - np.eye(k))) < 1e-12
        assert np.max(np.abs(A @ V - V @ H - R @ E.T)) < 1e-10
        assert abs(f.normres - np.linalg.norm(R)) < 1e-11
        assert np.max(np.abs(V.T @ R)) < 1e-11
    # H agrees with the oracle's up to the sign/rotation freedom-free quantities: its spectrum
    np.testing.assert_allclose(np.linalg.eigvalsh(f.H[:k, :k]), np.linalg.eigvalsh(of.H[:k, :k]), atol=1e-9)
    np.testing.assert_allclose(np.abs(f.H[:k, :k]), np.abs(of.H[:k, :k]), atol=1e-8)
    ctx.set_option("block_mode", 1)


@pytest.mark.parametrize("bs", [2, 3, 5, 8, 16])
def test_blocklanczos_async_step_matches_synchronous(kk, ko, ctx, bs):
    """The asynchronous block step (CholQR2 algebra on the device, one host sync per expand!) against the synchronous panel
    route and the oracle: same block sizes, H up to roundoff, invariants of test/factorize.jl:387-401."""
    nx, ny = 40, 25
    n = nx * ny
    A = ko.laplacian_2d(nx, ny, shift_diag=10 * np.linspace(0, 1, n) ** 2)
    rng = np.random.default_rng(70 + bs)
    x0 = [rng.random(n) for _ in range(bs)]
    steps = 5 if bs < 16 else 4
    Hs = {}
    # 1: asynchronous + tile-fused Gram kernels, 2: asynchronous with separate kernels, 3: asynchronous one-pass projection
    # (+ fused CholQR2 round), 0: synchronous
    # 4: as 3 but the second CholQR2 back-substitution always executed (qr_skip_tol = 0; the default skips it when the first
    # round left the block orthonormal to 2e-14, which is the case here: last_qr_dev ~ 1e-15)
    # 5: as 3 but without the residual-block Gram matrix handed from step to step (resid_gram = 0: every step reads its
    # residual block once more for the first CholQR2 Gram pass)
    # 6: as 3 with the normalised commit of the residual block (the library default: the update writes T = W R1^-1 into the
    # next basis slot, the next step starts at the second CholQR2 round); 3, 4, 5 run with it switched off
    launches = {}
    for mode in (1, 2, 3, 4, 5, 6, 0):
        ctx.set_option("block_async", 1 if mode else 0)
        ctx.set_option("block_fuse", {1: 3, 2: 0, 3: 5, 4: 5, 5: 5, 6: 5, 0: 0}[mode])
        ctx.set_option("block_commit", 1 if mode == 6 else 0)
        ctx.set_option("qr_skip_tol", 0.0 if mode == 4 else 2e-14)
        ctx.set_option("resid_gram", 0 if mode == 5 else 1)
        ctx.prof_reset(); ctx.prof_enable(1)
        it = kk.BlockLanczosIterator(kk.SparseOperator(A, ctx, symmetric=True), x0, (steps + 1) * bs + bs)
        f = it.initialize()
        for _ in range(steps):
            f = it.expand(f)
            assert f.R_size == bs and not f.last_drift
        k = len(f)
        V = f.V.to_numpy()
        R = np.stack([f.residual()[j].get() for j in range(f.R_size)], 1)
        H = f.H[:k, :k]
        E = np.zeros((k, bs)); E[k - bs:, :] = np.eye(bs)
        assert np.max(np.abs(V.T @ V - np.eye(k))) < 1e-12
        assert np.max(np.abs(A @ V - V @ H - R @ E.T)) < 1e-10
        assert abs(f.normres - np.linalg.norm(R)) < 1e-11 and np.max(np.abs(V.T @ R)) < 1e-11
        Hs[mode] = H.copy()
        ctx.prof_enable(0)
        launches[mode] = ctx.prof_get("k_block_gram")[1]
        if mode in (3, 4, 6):
            assert 0 < ctx.get_option("last_qr_dev") < 2e-14    # the skip branch (3) / the full branch (4) were really taken
    ctx.set_option("qr_skip_tol", 2e-14)
    ctx.set_option("resid_gram", 1)
    assert launches[5] - launches[3] == steps - 1    # every step but the first started from the handed-over Gram matrix
    # ... and, with the commit, without the Q1 = W R1^-1 pass either; since round 6 initialize ends in the commit too: the FIRST step
    # also skips its Gram pass and its Q1 pass (one launch per later step, two or more for the first)
    assert launches[3] - launches[6] >= steps + 1
    ctx.set_option("block_commit", 1)
    ctx.set_option("block_async", 1)
    ctx.set_option("block_fuse", BLOCK_FUSE_DEFAULT)
    np.testing.assert_allclose(Hs[1], Hs[0], atol=1e-10)
    np.testing.assert_allclose(Hs[2], Hs[0], atol=1e-10)
    np.testing.assert_allclose(Hs[3], Hs[0], atol=1e-10)
    np.testing.assert_allclose(Hs[4], Hs[0], atol=1e-10)
    np.testing.assert_allclose(Hs[5], Hs[0], atol=1e-10)
    np.testing.assert_allclose(Hs[6], Hs[0], atol=1e-10)
    oit = ko.BlockLanczosIterator(A, [x.copy() for x in x0], (steps + 1) * bs + bs)
    of = ko.blocklanczos_initialize(oit)
    for _ in range(steps):
        of = ko.blocklanczos_expand(oit, of)
    np.testing.assert_allclose(np.linalg.eigvalsh(Hs[1]), np.linalg.eigvalsh(of.H[:k, :k]), atol=1e-9)


def test_blocklanczos_one_pass_step_repeats_on_cancellation(kk, ko, ctx):
    """The one-pass block step projects A X against the whole basis once; when a column loses more than a factor 10 of its
    norm in that projection (A close to a multiple of the identity: |w_j| << |A x_j|) the device flag sends the step to the
    two-pass route of the reference (three-term recurrence, then block_reorthogonalize!, blocklanczos.jl:253-284)."""
    nx, ny, bs = 30, 20, 4
    n = nx * ny
    A = (ko.laplacian_2d(nx, ny) + 1000.0 * sp.identity(n)).tocsr()
    rng = np.random.default_rng(11)
    x0 = [rng.random(n) for _ in range(bs)]
    ctx.set_option("block_async", 1)
    ctx.set_option("block_fuse", 5)
    it = kk.BlockLanczosIterator(kk.SparseOperator(A, ctx, symmetric=True), x0, 6 * bs)
    oit = ko.BlockLanczosIterator(A, [x.copy() for x in x0], 6 * bs)
    f, of = it.initialize(), ko.blocklanczos_initialize(oit)
    ctx.prof_reset(); ctx.prof_enable(1)
    for _ in range(3):
        f = it.expand(f)
        of = ko.blocklanczos_expand(oit, of)
    ctx.prof_enable(0)
    assert ctx.prof_get("k_spmm_ell")[1] == 6, "every step must have been repeated on the two-pass route"
    ctx.set_option("block_fuse", BLOCK_FUSE_DEFAULT)
    k = len(f)
    V = f.V.to_numpy()
    assert np.max(np.abs(V.T @ V - np.eye(k))) < 1e-12
    np.testing.assert_allclose(np.linalg.eigvalsh(f.H[:k, :k]), np.linalg.eigvalsh(of.H[:k, :k]), rtol=1e-12)
    assert abs(f.normres - of.normres) < 1e-9 * max(1.0, of.normres)


def test_blocklanczos_async_step_hands_rank_drop_to_faithful_route(kk, ko, ctx):
    """A residual block that loses rank must not be accepted by the asynchronous CholQR2 step: the safety flag sends it to
    the reference's column-by-column block_qr! (blocklanczos.jl:312-353) and the block shrinks exactly as in the oracle."""
    n, bs = 300, 4
    rng = np.random.default_rng(5)
    Q, _ = np.linalg.qr(rng.standard_normal((n, n)))
    lam = np.concatenate([np.array([10.0, 9.0, 8.0]), np.linspace(0.1, 1.0, n - 3)])
    Ad = (Q * lam) @ Q.T
    Ad = (Ad + Ad.T) / 2
    # start block inside an invariant subspace of dimension 6: the third block cannot have full rank
    X0 = Q[:, :6] @ rng.standard_normal((6, bs))
    x0 = [X0[:, j].copy() for j in range(bs)]
    it = kk.BlockLanczosIterator(kk.SparseOperator(sp.csr_matrix(Ad), ctx, symmetric=True), x0, 40)
    oit = ko.BlockLanczosIterator(Ad, [x.copy() for x in x0], 40)
    f, of = it.initialize(), ko.blocklanczos_initialize(oit)
    sizes, osizes = [f.R_size], [of.R_size]
    for _ in range(2):
        try:
            f = it.expand(f)
            sizes.append(f.R_size)
        except kk.KrylovHipError:
            sizes.append(0)
            break
        try:
            of = ko.blocklanczos_expand(oit, of)
            osizes.append(of.R_size)
        except Exception:
            osizes.append(0)
            break
    assert sizes[:len(osizes)] == osizes[:len(sizes)] and min(sizes) < bs


@pytest.mark.parametrize("block_mode", [0, 1])
def test_blocklanczos_issue143_known_answer_on_device(kk, ko, ctx, block_mode):
    """The reference's regression test test/issues.jl:114-128 on the HIP path: all 71 eigenvalues,
    numiter == 1, numops == length(D) + 1 (rank-deficient last block: 20+20+20+11)."""
    import scipy.sparse as sps
    from pathlib import Path
    ctx.set_option("block_mode", block_mode)
    A = np.load(Path(__file__).resolve().parent / "golden" / "issue143_A.npy")
    rng = np.random.default_rng(143)
    x0 = [rng.standard_normal(71) for _ in range(20)]
    D, V, info = kk.eigsolve_block(kk.SparseOperator(sps.csr_matrix(A), ctx, symmetric=True), x0, 4, "SR",
                                   kk.BlockLanczos(tol=1e-8))
    ev = np.linalg.eigvalsh(A)
    assert len(D) == len(ev)
    np.testing.assert_allclose(np.sort(D), ev, rtol=0, atol=1e-10 * np.max(np.abs(ev)))
    U = np.stack(V, 1)
    assert np.max(np.abs(A @ U - U * D)) < 1e-9 * np.max(np.abs(ev))
    assert info.converged == len(D) and info.numiter == 1 and info.numops == len(D) + 1
    Do, Vo, oinfo = ko.eigsolve_blocklanczos(A, [x.copy() for x in x0], 4, "SR", tol=1e-8)
    assert (info.numiter, info.numops, info.converged) == (oinfo.numiter, oinfo.numops, oinfo.converged)
    ctx.set_option("block_mode", 1)


def test_eigsolve_block_with_restarts(kk, ko, ctx):
    """test/eigsolve.jl:551-600 style: BlockLanczos with restarts vs dense eigenvalues and vs the oracle."""
    nx, ny, bs = 24, 18, 3
    n = nx * ny
    A = ko.laplacian_2d(nx, ny, shift_diag=10 * np.linspace(0, 1, n) ** 2)
    ev = np.linalg.eigvalsh(A.toarray())
    rng = np.random.default_rng(9)
    x0 = [rng.random(n) for _ in range(bs)]
    D, V, info = kk.eigsolve_block(kk.SparseOperator(A, ctx, symmetric=True), x0, 5, "SR",
                                   kk.BlockLanczos(krylovdim=30, tol=1e-10, maxiter=200))
    Do, Vo, oinfo = ko.eigsolve_blocklanczos(A, [x.copy() for x in x0], 5, "SR", krylovdim=30, tol=1e-10, maxiter=200)
    assert info.converged >= 5
    assert relerr(D[:5], ev[:5]) < 1e-10 and relerr(D[:5], Do[:5]) < 1e-10
    assert (info.numiter, info.numops) == (oinfo.numiter, oinfo.numops)
    for lam, v in zip(D[:5], V[:5]):
        assert np.linalg.norm(A @ v - lam * v) < 1e-8


@pytest.mark.parametrize("n,m", [(300, 5), (5000, 33), (40000, 61), (20000, 100), (9000, 128), (9000, 130)])
def test_fused_unproject_project_matches_unfused(kk, ko, ctx, n, m):
    """The fused [w -= V s1 ; s2 = V'w] kernel (V read once) against the two separate passes, CGS2 and MGS2."""
    rng = np.random.default_rng(n + m)
    Q, _ = np.linalg.qr(rng.standard_normal((n, m)))
    w0 = Q @ rng.standard_normal(m) * 20 + rng.standard_normal(n)
    res = {}
    for fuse in (0, 1):
        ctx.set_option("fuse_passes", fuse)
        for dev in (kk.ClassicalGramSchmidt2(), kk.ModifiedGramSchmidt2()):
            B = kk.DeviceBasis(n, m + 1, ctx)
            for j in range(m):
                B.upload(j, Q[:, j])
            B.length = m
            vw = B[m].set(w0)
            x, nrm, passes = B.orthogonalize(vw, dev)
            res[(fuse, dev.name)] = (x.copy(), nrm, vw.get())
            assert passes == 2
    ctx.set_option("fuse_passes", 1)
    for name in ("cgs2", "mgs2"):
        x0_, n0, w0_ = res[(0, name)]
        x1_, n1, w1_ = res[(1, name)]
        assert np.max(np.abs(x0_ - x1_)) <= 1e-12 * np.linalg.norm(w0)
        assert abs(n0 - n1) <= 1e-12 * n0
        np.testing.assert_allclose(w0_, w1_, rtol=0, atol=1e-12 * np.linalg.norm(w0))
        assert np.max(np.abs(Q.T @ w1_)) < 1e-12 * np.linalg.norm(w0)


@pytest.mark.parametrize("n,m,k", [(1000, 7, 3), (70001, 100, 60), (20000, 100, 77), (9000, 61, 61), (9000, 116, 96), (9000, 120, 100)])
def test_basistransform_mfma_and_fallback(kk, ctx, n, m, k):
    """basistransform! (orthonormal.jl:291-354) in place: the MFMA tall-skinny GEMM path (n <= 96) and the LDS fallback."""
    rng = np.random.default_rng(n + m + k)
    V = rng.standard_normal((n, m))
    U = rng.standard_normal((m, k))
    B = kk.DeviceBasis(n, m, ctx)
    for j in range(m):
        B.upload(j, V[:, j])
    B.length = m
    B.basistransform(U)
    ref = V.copy()
    ref[:, :k] = V @ U
    got = B.to_numpy()
    scale = np.abs(V) @ np.abs(U)
    assert np.max(np.abs(got[:, :k] - ref[:, :k]) / scale) < 1e-14
    np.testing.assert_array_equal(got[:, k:], V[:, k:])  # untouched columns
    # pad rows must still be zero: norms only see n entries
    assert abs(B[0].norm() - np.linalg.norm(ref[:, 0])) <= 1e-13 * np.linalg.norm(ref[:, 0])


def test_linsolve_cg(kk, ko, ctx):
    """CG on device vectors (fused SpMV+dot, fused update): same iterates / counts as the oracle (linsolve/cg.jl)."""
    A = ko.laplacian_2d(40, 30, shift_diag=0.5 + np.linspace(0, 1, 1200))
    n = A.shape[0]
    b = np.random.default_rng(4).random(n)
    for a0, a1 in ((0.0, 1.0), (0.3, 0.9)):
        tol = 1e-10 * np.linalg.norm(b)
        x, info = kk.linsolve_cg(kk.SparseOperator(A, ctx, symmetric=True), b, None, kk.CG(500, tol), a0, a1)
        xo, oinfo = ko.cg(A, b, None, a0, a1, maxiter=500, tol=tol)
        assert info.converged == 1 and (info.numiter, info.numops) == (oinfo.numiter, oinfo.numops)
        assert np.linalg.norm(a0 * x + a1 * (A @ x) - b) <= 1.01 * tol
        np.testing.assert_allclose(x, xo, rtol=0, atol=1e-9 * np.linalg.norm(xo))
    xs = np.random.default_rng(1).random(n)
    x, info = kk.linsolve_cg(kk.SparseOperator(A, ctx, symmetric=True), A @ xs, xs, kk.CG(tol=1e-8))
    assert info.numops == 1 and info.converged == 1


def test_linsolve_bicgstab(kk, ko, ctx):
    """BiCGStab on device vectors (3 fused vector kernels + 2 SpMVs with fused inner products per iteration, scalars
    on the device): same counts and iterates as the oracle (linsolve/bicgstab.jl), incl. early stop at maxiter."""
    A = ko.convection_diffusion_2d(40, 30)
    n = A.shape[0]
    b = np.random.default_rng(4).random(n)
    for a0, a1 in ((0.0, 1.0), (0.3, 0.9)):
        tol = 1e-10 * np.linalg.norm(b)
        x, info = kk.linsolve_bicgstab(kk.SparseOperator(A, ctx), b, None, kk.BiCGStab(500, tol), a0, a1)
        xo, oinfo = ko.bicgstab(A, b, None, a0, a1, maxiter=500, tol=tol)
        assert info.converged == 1 and oinfo.converged == 1
        assert abs(info.numiter - oinfo.numiter) <= 1 and abs(info.numops - oinfo.numops) <= 2
        assert np.linalg.norm(a0 * x + a1 * (A @ x) - b) <= 1.01 * tol
        np.testing.assert_allclose(x, xo, rtol=0, atol=1e-8 * np.linalg.norm(xo))
    # fixed small iteration count: the recurrence itself is compared (no convergence-branch luck involved)
    x, info = kk.linsolve_bicgstab(kk.SparseOperator(A, ctx), b, None, kk.BiCGStab(6, 1e-30))
    xo, oinfo = ko.bicgstab(A, b, None, maxiter=6, tol=1e-30)
    assert (info.converged, info.numiter, info.numops) == (0, oinfo.numiter, oinfo.numops) == (0, 6, 13)
    np.testing.assert_allclose(x, xo, rtol=0, atol=1e-11 * np.linalg.norm(xo))
    np.testing.assert_allclose(info.normres, oinfo.normres, rtol=1e-8)
    xs = np.random.default_rng(1).random(n)
    x, info = kk.linsolve(kk.SparseOperator(A, ctx), A @ xs, xs, kk.BiCGStab(tol=1e-8))   # dispatch on the algorithm type
    assert info.numops == 1 and info.converged == 1


def test_lssolve_lsmr(kk, ko, ctx):
    """LSMR on device vectors (lssolve/lsmr.jl): issue #133 known answer, then a sparse rectangular least-squares
    problem with / without damping against the oracle (counts equal, iterates to 1e-9)."""
    import scipy.sparse as sp
    x, info = kk.lssolve(sp.identity(2, format="csr"), np.array([1.0, 0.0]))
    assert np.array_equal(x, [1.0, 0.0])
    assert (info.converged, info.numiter, info.numops, info.normres) == (1, 1, 2, 0.0)
    rng = np.random.default_rng(5)
    A = (sp.random(3000, 800, density=0.01, random_state=3, format="csr") + sp.eye(3000, 800, format="csr")).tocsr()
    b = rng.random(3000)
    for lam, K, orth_d, orth_o in ((0.0, 30, kk.ModifiedGramSchmidt(), ko.MGS), (0.7, 8, kk.ClassicalGramSchmidt2(), ko.CGS2),
                                   (0.0, 1, kk.ModifiedGramSchmidt(), ko.MGS)):
        tol = 1e-9 * np.linalg.norm(b)
        x, info = kk.lssolve(kk.SparseOperator(A, ctx), b, kk.LSMR(orth_d, 400, K, tol), lam)
        xo, oinfo = ko.lsmr(A, b, lam, krylovdim=K, maxiter=400, tol=tol, orth=orth_o)
        assert info.converged == 1 and (info.numiter, info.numops) == (oinfo.numiter, oinfo.numops)
        r = b - A @ x
        assert np.linalg.norm(A.T @ r - lam ** 2 * x) <= 5 * tol
        np.testing.assert_allclose(info.residual, r, rtol=0, atol=1e-9 * np.linalg.norm(b))
        np.testing.assert_allclose(x, xo, rtol=0, atol=1e-8 * np.linalg.norm(xo))
    # fixed iteration count (no convergence-branch luck): recurrence scalars must agree
    x, info = kk.lssolve(kk.SparseOperator(A, ctx), b, kk.LSMR(kk.ModifiedGramSchmidt(), 7, 30, 1e-30))
    xo, oinfo = ko.lsmr(A, b, krylovdim=30, maxiter=7, tol=1e-30)
    assert (info.converged, info.numiter, info.numops) == (0, 7, 15) == (oinfo.converged, oinfo.numiter, oinfo.numops)
    np.testing.assert_allclose(info.normres, oinfo.normres, rtol=1e-9)
    np.testing.assert_allclose(x, xo, rtol=0, atol=1e-12 * np.linalg.norm(xo))


@pytest.mark.parametrize("method", ["lanczos", "arnoldi"])
def test_expintegrator(kk, ko, ctx, method):
    """exponentiate / expintegrator on device vectors (matrixfun/expintegrator.jl): same result, numiter and numops as
    the oracle, for one-shot (krylovdim large enough) and multi-step (small krylovdim) runs, p = 1 and p = 3."""
    import scipy.sparse.linalg as spl
    A = ko.laplacian_2d(30, 20) if method == "lanczos" else ko.convection_diffusion_2d(30, 20)
    A = (A / 8.0).tocsr()
    n = A.shape[0]
    rng = np.random.default_rng(9)
    op = kk.SparseOperator(A, ctx, symmetric=(method == "lanczos"))
    mk = kk.Lanczos if method == "lanczos" else kk.Arnoldi
    for t, K in ((0.7, 40), (-2.5, 12), (6.0, 10)):
        v = rng.random(n)
        w, info = kk.exponentiate(op, t, v, mk(kk.ModifiedGramSchmidt2(), K, 100, 1e-11))
        wo, oinfo = ko.expintegrator(A, t, (v,), krylovdim=K, maxiter=100, tol=1e-11, orth=ko.MGS2, method=method)
        assert info.converged == 1 and (info.numiter, info.numops) == (oinfo.numiter, oinfo.numops)
        np.testing.assert_allclose(w, wo, rtol=0, atol=1e-10 * np.linalg.norm(wo))
        np.testing.assert_allclose(w, spl.expm_multiply(t * A.tocsc(), v), rtol=0, atol=1e-8 * np.linalg.norm(v))
    u = [rng.random(n) for _ in range(4)]
    for K, orth_d, orth_o in ((12, kk.ClassicalGramSchmidt2(), ko.CGS2), (25, kk.ModifiedGramSchmidtIR(), ko.MGSIR())):
        w, info = kk.expintegrator(op, 1.3, u, mk(orth_d, K, 100, 1e-11))
        wo, oinfo = ko.expintegrator(A, 1.3, u, krylovdim=K, maxiter=100, tol=1e-11, orth=orth_o, method=method)
        assert info.converged == 1 and (info.numiter, info.numops) == (oinfo.numiter, oinfo.numops)
        np.testing.assert_allclose(w, wo, rtol=0, atol=1e-10 * np.linalg.norm(wo))


@pytest.mark.parametrize("which", ["LM", "LR", "SR"])
def test_eigsolve_arnoldi(kk, ko, ctx, which):
    """eigsolve(A, x0, howmany, which, alg::Arnoldi) (eigsolve/arnoldi.jl): Krylov-Schur restarts on the device basis
    (kk_arnoldi_expand, kk_basistransform) for a real non-symmetric operator with complex-conjugate eigenvalue pairs:
    same restart / operation counts and eigenvalues as the oracle, eigen-residuals, schursolve relation."""
    import scipy.sparse as sp
    n = 500
    rng = np.random.default_rng(12)
    A = (sp.random(n, n, density=0.02, random_state=7, format="csr") - 0.5 * sp.random(n, n, density=0.02, random_state=8, format="csr")
         + sp.diags(np.linspace(-1, 1, n))).tocsr()
    x0 = rng.random(n)
    op = kk.SparseOperator(A, ctx)
    alg = kk.Arnoldi(kk.ModifiedGramSchmidt2(), 30, 50, 1e-10)
    vals, vecs, info = hx.eigsolve_arnoldi(op, x0, 4, which, alg)
    ovals, ovecs, oinfo = ko.eigsolve_arnoldi(A, x0, 4, which, krylovdim=30, maxiter=50, tol=1e-10, orth=ko.MGS2)
    assert info.converged >= 4 and (info.converged, info.numiter, info.numops) == (oinfo.converged, oinfo.numiter, oinfo.numops)
    assert len(vals) == len(ovals)
    np.testing.assert_allclose(vals, ovals, rtol=0, atol=1e-9 * np.max(np.abs(ovals)))
    for lam, v, nr in zip(vals, vecs, info.normres):
        assert abs(np.linalg.norm(v) - 1) < 1e-9
        assert np.linalg.norm(A @ v - lam * v) <= max(5 * nr, 1e-9)
    T, Q, svals, sinfo = hx.schursolve(op, x0, 4, which, alg)
    Qm = np.stack(Q, axis=1).real
    np.testing.assert_allclose(Qm.T @ Qm, np.eye(Qm.shape[1]), atol=1e-9)
    np.testing.assert_allclose(A @ Qm, Qm @ T, atol=1e-8)
    np.testing.assert_allclose(svals, vals[:len(svals)], rtol=0, atol=1e-9 * np.max(np.abs(ovals)))


@pytest.mark.parametrize("orth_name", ["cgs2", "mgs2", "cgsir", "mgsir", "cgs", "mgs"])
def test_geneigsolve_golubye(kk, ko, ctx, orth_name):
    """geneigsolve((A, B), x0, howmany, which, alg::GolubYe) (eigsolve/golubye.jl) on device vectors: A = shifted 2-D
    Laplacian, B = s.p.d. mass-like matrix; restarts, converged-vector re-insertion, B-orthonormal Ritz vectors; counts
    and values against the oracle."""
    import scipy.linalg as sla
    import scipy.sparse as sp
    dev = {"cgs": kk.ClassicalGramSchmidt(), "mgs": kk.ModifiedGramSchmidt(), "cgs2": kk.ClassicalGramSchmidt2(),
           "mgs2": kk.ModifiedGramSchmidt2(), "cgsir": kk.ClassicalGramSchmidtIR(), "mgsir": kk.ModifiedGramSchmidtIR()}[orth_name]
    ref = {"cgs": ko.CGS, "mgs": ko.MGS, "cgs2": ko.CGS2, "mgs2": ko.MGS2, "cgsir": ko.CGSIR(), "mgsir": ko.MGSIR()}[orth_name]
    nx, ny = 12, 10
    n = nx * ny
    A = ko.laplacian_2d(nx, ny, shift_diag=np.linspace(0, 2, n))
    T1 = sp.diags([np.full(n - 1, 1.0), np.full(n, 4.0), np.full(n - 1, 1.0)], [-1, 0, 1], format="csr") / 6.0
    B = (T1 + sp.diags(np.linspace(0.5, 1.5, n))).tocsr()
    x0 = np.random.default_rng(8).random(n)
    opA, opB = kk.SparseOperator(A, ctx, symmetric=True), kk.SparseOperator(B, ctx, symmetric=True)
    reorth = orth_name not in ("cgs", "mgs")
    for which, hm in (("SR", 3), ("LR", 2)):
        kw = dict(krylovdim=14, maxiter=60 if reorth else 3, tol=1e-9)
        vals, vecs, info = hx.geneigsolve((opA, opB), x0, hm, which, hx.GolubYe(dev, kw["krylovdim"], kw["maxiter"], kw["tol"]))
        ovals, ovecs, oinfo = ko.geneigsolve_golubye(A, B, x0, hm, which, orth=ref, **kw)
        assert (info.converged, info.numiter, info.numops) == (oinfo.converged, oinfo.numiter, oinfo.numops)
        assert len(vals) == len(ovals)
        np.testing.assert_allclose(vals, ovals, rtol=1e-8, atol=1e-10)
        if reorth:
            assert info.converged >= hm
            exact = sla.eigh(A.toarray(), B.toarray(), eigvals_only=True)
            exact = exact[:len(vals)] if which == "SR" else exact[::-1][:len(vals)]
            np.testing.assert_allclose(vals[:info.converged], exact[:info.converged], rtol=1e-7, atol=1e-8)
        U = np.stack(vecs, axis=1)
        R = np.stack(info.residual, axis=1)
        np.testing.assert_allclose(U.T @ (B @ U), np.eye(U.shape[1]), atol=1e-7)
        np.testing.assert_allclose(A @ U, (B @ U) * vals[None, :] + R, atol=1e-8)


def test_short_recurrence_entry_points_reject_bad_arguments(kk, ko, ctx):
    """Error behaviour of the 8(f)-3 entry points: status < 0 and a message, never a crash or a silent no-op."""
    import ctypes as C
    from krylovkit_hip import _lib
    A = ko.convection_diffusion_2d(10, 8)
    op = kk.SparseOperator(A, ctx)
    W = kk.DeviceBasis(80, 9, ctx)
    lib = W._lib
    d = [C.c_double() for _ in range(3)]
    good = (C.c_int * 9)(0, 1, 2, 3, 4, 5, 6, 3, 4)
    bad = (C.c_int * 9)(0, 1, 2, 3, 4, 5, 99, 3, 4)
    for cols, mode in ((good, 7), (good, 3), (bad, 1)):      # unknown mode; collect without a run-ahead half; column range
        with pytest.raises(kk.KrylovHipError):
            _lib.check(lib.kk_bicgstab_half(op.handle, W.handle, cols, 0.0, 1.0, mode, 1.0, C.byref(d[0]), C.byref(d[1])))
    with pytest.raises(kk.KrylovHipError):
        _lib.check(lib.kk_bicgstab_full(op.handle, W.handle, bad, 0.0, 1.0, 0, None, C.byref(d[0]), C.byref(d[1]), C.byref(d[2])))
    with pytest.raises(kk.KrylovHipError):
        _lib.check(lib.kk_lsmr_step_u(W.handle, 1, 1, 2, 0.5, 0.5, C.byref(d[0])))           # aliased columns
    with pytest.raises(kk.KrylovHipError):
        _lib.check(lib.kk_lsmr_update(W.handle, 1, 2, 2, None, -1, 0.1, 0.2, 0.3))
    R = kk.SparseOperator(ko.sparse_random(50, 30, 3, 1), ctx)
    with pytest.raises(kk.KrylovHipError):                                                    # rectangular map in a square solver
        kk.linsolve_bicgstab(R, np.ones(50))
    with pytest.raises(ValueError):
        hx.geneigsolve((op, op), np.ones(80), 1, "LI")


def test_function_operator(kk, ko, ctx):
    """apply(f, x) = f(x) (apply.jl:2): a callable operator drives the same iterators through the un-fused sequence
    (one call of f, L1 verbs, fused orthogonalisation passes).  (1) f = a SparseOperator's apply must reproduce the
    fused kk_lanczos_expand / kk_arnoldi_expand results for all six orthogonalisers; (2) f = A^2 (two SpMVs) in
    eigsolve, against the oracle driven by the same composite map; (3) Arnoldi eigsolve and exponentiate."""
    nx, ny = 30, 20
    n = nx * ny
    A = ko.laplacian_2d(nx, ny, shift_diag=np.linspace(0, 3, n))
    Cm = ko.convection_diffusion_2d(nx, ny)
    x0 = np.random.default_rng(13).random(n)
    opA, opC = kk.SparseOperator(A, ctx, symmetric=True), kk.SparseOperator(Cm, ctx)
    fA = kk.FunctionOperator(lambda x, y: opA.apply(x, y), n, ctx, symmetric=True)
    fC = kk.FunctionOperator(lambda x, y: opC.apply(x, y), n, ctx)
    devs = [kk.ClassicalGramSchmidt(), kk.ModifiedGramSchmidt(), kk.ClassicalGramSchmidt2(), kk.ModifiedGramSchmidt2(),
            kk.ClassicalGramSchmidtIR(), kk.ModifiedGramSchmidtIR()]
    for dev in devs:
        f1 = kk.initialize(kk.LanczosIterator(opA, x0, dev, capacity=20))
        f2 = kk.initialize(kk.LanczosIterator(fA, x0, dev, capacity=20))
        it1, it2 = kk.LanczosIterator(opA, x0, dev, capacity=20), kk.LanczosIterator(fA, x0, dev, capacity=20)
        f1, f2 = kk.initialize(it1), kk.initialize(it2)
        for _ in range(15):
            f1, f2 = kk.expand_(it1, f1), kk.expand_(it2, f2)
        assert relerr(f2.alphas, f1.alphas) < 1e-10 and relerr(f2.betas, f1.betas) < 1e-9, dev.name
        a1, a2 = kk.ArnoldiIterator(opC, x0, dev, capacity=20), kk.ArnoldiIterator(fC, x0, dev, capacity=20)
        g1, g2 = kk.initialize(a1), kk.initialize(a2)
        for _ in range(12):
            g1, g2 = kk.expand_(a1, g1), kk.expand_(a2, g2)
        np.testing.assert_allclose(g2.rayleighquotient(), g1.rayleighquotient(), rtol=0, atol=1e-10)
        assert abs(g2.normres - g1.normres) < 1e-10
    # (2) composite map A^2 through a scratch column
    S = kk.DeviceBasis(n, 1, ctx)

    def a2(x, y):
        opA.apply(x, S[0])
        opA.apply(S[0], y)

    f2op = kk.FunctionOperator(a2, n, ctx, symmetric=True)
    vals, vecs, info = kk.eigsolve(f2op, x0, 3, "LM", kk.Lanczos(kk.ModifiedGramSchmidt2(), 24, 100, 1e-10))
    ovals, _, oinfo = ko.eigsolve_lanczos(lambda z: A @ (A @ z), x0, 3, "LM", krylovdim=24, maxiter=100, tol=1e-10, orth=ko.MGS2)
    assert info.converged >= 3 and (info.numiter, info.numops) == (oinfo.numiter, oinfo.numops)
    np.testing.assert_allclose(vals[:3], ovals[:3], rtol=1e-10)
    ev = np.linalg.eigvalsh(A.toarray()) ** 2
    np.testing.assert_allclose(np.sort(vals[:3])[::-1], np.sort(ev)[::-1][:3], rtol=1e-9)
    # (3) Arnoldi eigsolve and exponentiate with a callable
    vals, vecs, info = hx.eigsolve_arnoldi(fC, x0, 2, "LR", kk.Arnoldi(kk.ModifiedGramSchmidt2(), 30, 60, 1e-9))
    ovals, _, oinfo = ko.eigsolve_arnoldi(lambda z: Cm @ z, x0, 2, "LR", krylovdim=30, maxiter=60, tol=1e-9, orth=ko.MGS2)
    assert (info.converged, info.numiter, info.numops) == (oinfo.converged, oinfo.numiter, oinfo.numops)
    np.testing.assert_allclose(vals, ovals, rtol=0, atol=1e-8 * np.max(np.abs(ovals)))
    fE = kk.FunctionOperator(lambda x, y: opA.apply(x, y).scale_(-0.25), n, ctx, symmetric=True)
    w, info = kk.exponentiate(fE, 0.8, x0, kk.Lanczos(kk.ModifiedGramSchmidt2(), 20, 100, 1e-11))
    wo, oinfo = ko.expintegrator(lambda z: -0.25 * (A @ z), 0.8, (x0,), krylovdim=20, maxiter=100, tol=1e-11, orth=ko.MGS2)
    assert (info.converged, info.numiter, info.numops) == (1, oinfo.numiter, oinfo.numops)
    np.testing.assert_allclose(w, wo, rtol=0, atol=1e-10 * np.linalg.norm(wo))


@pytest.mark.parametrize("which", ["LM", "SR"])
def test_bieigsolve_biarnoldi(kk, ko, ctx, which):
    """bieigsolve(f, v0, w0, howmany, which, alg::BiArnoldi) (eigsolve/biarnoldi.jl): two device Arnoldi factorizations
    (A fused, A' through the function-operator path), oblique residual corrections, two-sided Krylov-Schur restarts;
    counts and eigenvalues against the oracle, right / left eigen-relations, biorthogonality W'V = I."""
    import scipy.sparse as sp
    n = 400
    rng = np.random.default_rng(14)
    A = (sp.random(n, n, density=0.03, random_state=17, format="csr") - 0.5 * sp.random(n, n, density=0.03, random_state=18, format="csr")
         + sp.diags(np.linspace(-1, 1, n))).tocsr()
    v0, w0 = rng.random(n), rng.random(n)
    alg = hx.BiArnoldi(kk.ModifiedGramSchmidt2(), 30, 60, 1e-9)
    vals, (VR, WL), (iV, iW) = hx.bieigsolve(kk.SparseOperator(A, ctx), v0, w0, 3, which, alg)
    ovals, (oVR, oWL), (oV, oW) = ko.bieigsolve_biarnoldi(A, v0, w0, 3, which, krylovdim=30, maxiter=60, tol=1e-9, orth=ko.MGS2)
    assert iV.converged >= 3 and (iV.converged, iV.numiter, iV.numops) == (oV.converged, oV.numiter, oV.numops)
    assert len(vals) == len(ovals)
    np.testing.assert_allclose(vals, ovals, rtol=0, atol=1e-8 * np.max(np.abs(ovals)))
    UV, UW = np.stack(VR, axis=1), np.stack(WL, axis=1)
    RV, RW = np.stack(iV.residual, axis=1), np.stack(iW.residual, axis=1)
    np.testing.assert_allclose(A @ UV, UV * vals[None, :] + RV, atol=1e-8)
    np.testing.assert_allclose(A.T @ UW, UW * np.conj(vals)[None, :] + RW, atol=1e-8)
    l = iV.converged
    assert np.max(iV.normres[:l]) < 1e-7 and np.max(iW.normres[:l]) < 1e-7
    np.testing.assert_allclose((UW.conj().T @ UV)[:l, :l], np.eye(l), atol=1e-6)


@pytest.mark.parametrize("mgs_mode", [0, 1])
def test_mgs_on_non_orthonormal_basis(kk, ko, ctx, mgs_mode):
    """The low-sync form (I + L) s = V'w is exact algebra for ANY basis (MGS never divides by |q|^2):
    coefficients must match the sequential oracle also when the 'basis' is far from orthonormal."""
    ctx.set_option("mgs_mode", mgs_mode)
    rng = np.random.default_rng(77)
    n, m = 4000, 12
    Q, _ = np.linalg.qr(rng.standard_normal((n, m)))
    V = Q @ (np.eye(m) + 0.3 * rng.standard_normal((m, m)))      # correlated, non-normalised columns
    w0 = rng.standard_normal(n)
    for dev, ref in ((kk.ModifiedGramSchmidt(), ko.MGS), (kk.ModifiedGramSchmidt2(), ko.MGS2)):
        B = kk.DeviceBasis(n, m + 1, ctx)
        for j in range(m):
            B.upload(j, V[:, j])
        B.length = m
        vw = B[m].set(w0)
        x, nrm, _ = B.orthogonalize(vw, dev)
        wr, xr = ko.orthogonalize(w0.copy(), [V[:, j].copy() for j in range(m)], ref)
        np.testing.assert_allclose(x, xr, rtol=1e-9, atol=1e-10 * np.linalg.norm(w0))
        np.testing.assert_allclose(vw.get(), wr, rtol=0, atol=1e-9 * np.linalg.norm(w0))
    ctx.set_option("mgs_mode", 2)


def test_wide_basis_is_chunked(kk, ko, ctx):
    """A basis wider than one kernel panel (KK_MAX_M = 256 columns) is processed in panels by project!! / unproject!!;
    iterators of any krylovdim can be built (the expand! entry points go panel by panel beyond it: test_gpu_wide_basis.py)."""
    n, m = 2000, 300
    rng = np.random.default_rng(11)
    Vh = rng.standard_normal((n, m))
    B = kk.DeviceBasis(n, m + 1, ctx)
    for j in range(m):
        B.upload(j, Vh[:, j])
    B.length = m
    w = rng.standard_normal(n)
    vw = B[m].set(w)
    y = B.project(vw)
    np.testing.assert_allclose(y, Vh.T @ w, rtol=1e-12, atol=1e-10)
    c = rng.standard_normal(m)
    B.unproject(vw, c, 0, m, -0.5, 2.0)
    np.testing.assert_allclose(vw.get(), 2.0 * w - 0.5 * Vh @ c, rtol=1e-12, atol=1e-9)
    A = ko.laplacian_2d(20, 10)
    kk.LanczosIterator(kk.SparseOperator(A, ctx, symmetric=True), np.ones(200), kk.ModifiedGramSchmidt2(), capacity=300)
    kk.ArnoldiIterator(kk.SparseOperator(A, ctx), np.ones(200), kk.ModifiedGramSchmidt2(), capacity=300)


def test_linsolve_front_end_tolerances_and_bicgstab_breakdown_test(kk, ko, ctx):
    """the keyword front-end defaults atol = rtol = 1e-12 and uses tol = max(atol, rtol*|b|) (linsolve/linsolve.jl:123-151);
    BiCGStab's breakdown test `rho ≈ 0` is an exact-zero test (isapprox with atol = 0, bicgstab.jl:39)."""
    A = ko.convection_diffusion_2d(20, 15)
    n = A.shape[0]
    rng = np.random.default_rng(8)
    b = 1e3 * rng.random(n)                               # |b| >> 1: an absolute 1e-12 would be unreachable
    nb = np.linalg.norm(b)
    x, info = kk.linsolve(kk.SparseOperator(A, ctx), b, krylovdim=40, maxiter=50)
    assert info.converged == 1 and np.linalg.norm(A @ x - b) <= 1e-12 * nb * 1.01
    xo, oinfo = ko.gmres(A, b, krylovdim=40, maxiter=50, tol=max(1e-12, 1e-12 * nb), orth=ko.MGS2)
    assert (info.numiter, info.numops) == (oinfo.numiter, oinfo.numops)
    x2, info2 = kk.linsolve(kk.SparseOperator(A, ctx), b, rtol=1e-6, krylovdim=40, maxiter=50)   # only rtol given: atol keeps its default
    xo2, oinfo2 = ko.gmres(A, b, krylovdim=40, maxiter=50, tol=max(1e-12, 1e-6 * nb), orth=ko.MGS2)
    assert (info2.converged, info2.numiter, info2.numops) == (oinfo2.converged, oinfo2.numiter, oinfo2.numops)
    assert info2.converged == 1 and info2.numops < info.numops and np.linalg.norm(A @ x2 - b) <= 1e-6 * nb
    with pytest.raises(TypeError):
        kk.linsolve(kk.SparseOperator(A, ctx), b, None, kk.GMRES(tol=1e-8), rtol=1e-6)
    bs = 1e-6 * rng.random(n)                             # |r0|^2 = 1e-12-ish: np.isclose(rho, 0) would have stopped here
    xs, sinfo = kk.linsolve(kk.SparseOperator(A, ctx), bs, None, kk.BiCGStab(tol=1e-16, maxiter=400))
    xso, soinfo = ko.bicgstab(A, bs, tol=1e-16, maxiter=400)
    assert sinfo.numiter > 1 and sinfo.converged == soinfo.converged == 1
    assert np.linalg.norm(A @ xs - bs) < 1e-15


def test_constant_coefficient_stencil_needs_neither_indices_nor_values(kk, ko, ctx):
    """A grid stencil whose diagonals hold ONE value each (Laplacians, constant convection-diffusion) is applied by the
    value-free diagonal kernel (option spmv_dia_const, default on): bit-identical to the kernel that streams the diagonals
    (same products, same order), equal to SciPy, through every fused epilogue; operators with varying coefficients, or with
    entries that wrap around a grid line, keep streaming their diagonals."""
    import scipy.sparse as sp
    rng = np.random.default_rng(17)
    nx, ny = 70, 61
    n = nx * ny
    lap = ko.laplacian_2d(nx, ny)                                   # constant 5-point
    cd = ko.convection_diffusion_2d(nx, ny)                         # constant, non-symmetric 5-point
    ix, iy = np.meshgrid(np.arange(nx), np.arange(ny)); ix, iy = ix.ravel(), iy.ravel()
    rows, cols, vals = [], [], []
    for dy in (-1, 0, 1):                                           # constant 9-point
        for dx in (-1, 0, 1):
            ok = (ix + dx >= 0) & (ix + dx < nx) & (iy + dy >= 0) & (iy + dy < ny)
            r = (iy * nx + ix)[ok]
            rows.append(r); cols.append(r + dy * nx + dx); vals.append(np.full(r.size, 8.0 if (dx, dy) == (0, 0) else -1.0 + 0.1 * dx + 0.01 * dy))
    nine = sp.csr_matrix((np.concatenate(vals), (np.concatenate(rows), np.concatenate(cols))), shape=(n, n))
    wrap = lap.tolil(); wrap[nx - 1, nx] = -1.0; wrap = wrap.tocsr()      # one entry across a line end: not a pure stencil
    varying = ko.laplacian_2d(nx, ny, shift_diag=np.linspace(0, 1, n))
    B = kk.DeviceBasis(n, 8, ctx)
    X = rng.standard_normal((n, 3))
    for j in range(3):
        B.upload(j, X[:, j])
    for name, A, sym in (("laplacian", lap, True), ("convdiff", cd, False), ("nine", nine, False), ("wrap", wrap, False), ("varying", varying, True)):
        op = kk.SparseOperator(A, ctx, symmetric=sym)
        assert op.info()["format"] == ("ELL+DIA" if name in ("wrap", "varying") else "ELL+DIA const"), name
        outs = []
        for const in (1, 0):
            ctx.set_option("spmv_dia_const", const)
            op.apply(B[0], B[4])
            op.apply_affine(B[1], B[5], 0.7, -0.4)
            outs.append((B[4].get(), B[5].get()))
            # 1 / 2 / 4 row pairs per lane (the 4-pair form exists for the value-free kernel; long vectors pick it by size):
            # same products in the same order, so the same bits
            for pairs in (1, 2, 4):
                ctx.set_option("spmv_dia_pairs", pairs)
                op.apply(B[0], B[6])
                op.apply_affine(B[1], B[7], 0.7, -0.4)
                assert np.array_equal(B[6].get(), outs[-1][0]) and np.array_equal(B[7].get(), outs[-1][1]), (name, const, pairs)
            ctx.set_option("spmv_dia_pairs", 0)
        ctx.set_option("spmv_dia_const", 1)
        assert np.array_equal(outs[0][0], outs[1][0]) and np.array_equal(outs[0][1], outs[1][1]), name
        # the multi-column sweeping apply has the same value-free form
        from krylovkit_hip._lib import check
        Ys = []
        for const in (1, 0):
            ctx.set_option("spmv_dia_const", const)
            check(ctx._lib.kk_block_apply(op.handle, B.handle, 0, B.handle, 4, 3))
            Ys.append(np.stack([B.download(4 + j) for j in range(3)], 1))
        ctx.set_option("spmv_dia_const", 1)
        assert np.array_equal(Ys[0], Ys[1]), name
        np.testing.assert_allclose(Ys[0], A @ X, rtol=0, atol=1e-12, err_msg=name)
        np.testing.assert_allclose(outs[0][0], A @ X[:, 0], rtol=0, atol=1e-12, err_msg=name)
        np.testing.assert_allclose(outs[0][1], 0.7 * X[:, 1] - 0.4 * (A @ X[:, 1]), rtol=0, atol=1e-12, err_msg=name)
    # fused Lanczos epilogues (alpha dot, - beta v_prev, on-the-fly 1/beta scale of the speculative apply) on the constant operator
    x0 = rng.random(n)
    runs = []
    for const in (1, 0):
        ctx.set_option("spmv_dia_const", const)
        it = kk.LanczosIterator(kk.SparseOperator(lap, ctx, symmetric=True), x0, kk.ModifiedGramSchmidt2(), capacity=30)
        f = kk.initialize(it)
        for _ in range(25):
            f = kk.expand_(it, f)
        runs.append((list(f.alphas), list(f.betas)))
    ctx.set_option("spmv_dia_const", 1)
    # (the value-free apply of this operator -- even line length -- is the SWEEPING kernel since round 6: same y bits, but its fused inner
    # product is summed in another order than k_spmv_dia's; the bitwise comparison of the two value-free / streaming forms of k_spmv_dia
    # itself follows with the sweep switched off)
    assert relerr(runs[0][0], runs[1][0]) < 1e-12 and relerr(runs[0][1], runs[1][1]) < 1e-12
    ctx.set_option("spmv_dia_sw", 0)
    runs_k = []
    for const in (1, 0):
        ctx.set_option("spmv_dia_const", const)
        it = kk.LanczosIterator(kk.SparseOperator(lap, ctx, symmetric=True), x0, kk.ModifiedGramSchmidt2(), capacity=30)
        f = kk.initialize(it)
        for _ in range(25):
            f = kk.expand_(it, f)
        runs_k.append((list(f.alphas), list(f.betas)))
    ctx.set_option("spmv_dia_const", 1); ctx.set_option("spmv_dia_sw", 1)
    assert runs_k[0] == runs_k[1]
    oit = ko.LanczosIterator(lap, x0.copy(), ko.MGS2)
    of = ko.lanczos_initialize(oit)
    for _ in range(25):
        of = ko.lanczos_expand(oit, of)
    assert relerr(runs[0][0], of.alphas) < 1e-10 and relerr(runs[0][1], of.betas) < 1e-10


def _const_stencil(nx, ny, c):
    """5-point stencil with ONE coefficient per diagonal (offsets -nx, -1, 0, +1, +nx), natural ordering"""
    import scipy.sparse as sp
    n = nx * ny
    i = np.arange(n)
    ix = i % nx
    rows, cols, vals = [], [], []
    for q, off in enumerate((-nx, -1, 0, 1, nx)):
        ok = (i + off >= 0) & (i + off < n)
        if off == -1:
            ok &= ix > 0
        if off == 1:
            ok &= ix < nx - 1
        rows.append(i[ok]); cols.append(i[ok] + off); vals.append(np.full(ok.sum(), c[q]))
    return sp.csr_matrix((np.concatenate(vals), (np.concatenate(rows), np.concatenate(cols))), shape=(n, n))


@pytest.mark.parametrize("nx,ny", [(64, 70), (126, 40), (128, 37), (130, 41), (256, 20), (510, 12), (512, 11), (514, 13), (1030, 9), (2050, 7), (4000, 5)])
def test_sweeping_single_vector_stencil_apply(kk, ko, ctx, nx, ny):
    """k_spmv_dia_sw: the single-vector apply of a value-free 5-point stencil as a four-line rotating window per wave,
    1 / 2 strips per wave, 2 .. 16 lines per sweep -- BIT-identical to k_spmv_dia (same products in the same order) for the plain and the
    affine apply, equal to SciPy; the fused Lanczos epilogues (alpha dot in MGS and CGS order, - beta v_prev, |w|^2, the speculative
    scaled apply) against the other kernel to rounding and against the oracle at 1e-10 (src/apply.jl:1, factorizations/lanczos.jl:297-310)"""
    rng = np.random.default_rng(nx * 7 + ny)
    c = np.array([-1.25, -1.5, 4.0, -0.5, -0.75])
    A = _const_stencil(nx, ny, c)
    n = nx * ny
    op = kk.SparseOperator(A, ctx)
    assert op.info()["format"] == "ELL+DIA const"
    B = kk.DeviceBasis(n, 6, ctx)
    x, z = rng.standard_normal(n), rng.standard_normal(n)
    B.upload(0, x); B.upload(1, z)
    ctx.set_option("spmv_dia_sw", 0)
    op.apply(B[0], B[2]); op.apply_affine(B[1], B[3], 0.7, -0.4)
    ref = (B[2].get(), B[3].get())
    np.testing.assert_allclose(ref[0], A @ x, rtol=0, atol=1e-12)
    for ns in (1, 2):
        for lines in (0, 2, 3, 4, 8, 16):
            ctx.set_option("spmv_dia_sw", ns); ctx.set_option("spmv_dia_sw_lines", lines)
            l0 = ctx.get_option("spmv_dia_sw_launches")
            op.apply(B[0], B[4]); op.apply_affine(B[1], B[5], 0.7, -0.4)
            assert ctx.get_option("spmv_dia_sw_launches") == l0 + 2, (ns, lines)
            assert np.array_equal(B[4].get(), ref[0]) and np.array_equal(B[5].get(), ref[1]), (nx, ny, ns, lines)
    ctx.set_option("spmv_dia_sw_lines", 0)
    # fused epilogues through the Lanczos step; a symmetric operator of the same shape
    S = _const_stencil(nx, ny, np.array([-1.0, -1.5, 4.5, -1.5, -1.0]))
    x0 = rng.random(n)
    steps = 12
    for dev, oref in ((kk.ModifiedGramSchmidt2(), ko.MGS2), (kk.ClassicalGramSchmidt2(), ko.CGS2)):
        runs = {}
        for ns in (0, 1, 2):
            ctx.set_option("spmv_dia_sw", ns)
            it = kk.LanczosIterator(kk.SparseOperator(S, ctx, symmetric=True), x0, dev, capacity=steps + 3)
            f = kk.initialize(it)
            for _ in range(steps):
                f = kk.expand_(it, f)
            runs[ns] = (np.array(f.alphas), np.array(f.betas))
        oit = ko.LanczosIterator(S, x0.copy(), oref); of = ko.lanczos_initialize(oit)
        for _ in range(steps):
            of = ko.lanczos_expand(oit, of)
        for ns in (1, 2):
            assert relerr(runs[ns][0], runs[0][0]) < 1e-11 and relerr(runs[ns][1], runs[0][1]) < 1e-11, (dev.name, ns)
            assert relerr(runs[ns][0], of.alphas) < 1e-10 and relerr(runs[ns][1], of.betas) < 1e-10, (dev.name, ns)
    ctx.set_option("spmv_dia_sw", 1)
    B.free()


@pytest.mark.parametrize("bs", [3, 16])
def test_blocklanczos_pipelined_two_panel_gram_kernel(kk, ko, ctx, bs):
    """k_block_gram2p (software-pipelined form of the two-panel Gram kernel of the one-pass block step: buffer-descriptor
    loads, clamped columns instead of zero fill, ride-along tile aliased to the last X group) against the plain kernel: same
    H to rounding over enough steps to visit every instantiation (1..5 groups per launch, with and without the alias)."""
    nx, ny = 64, 40
    n = nx * ny
    A = ko.laplacian_2d(nx, ny, shift_diag=10 * np.linspace(0, 1, n) ** 2)
    rng = np.random.default_rng(90 + bs)
    x0 = [rng.random(n) for _ in range(bs)]
    steps = 7
    ctx.set_option("block_async", 1)
    ctx.set_option("block_fuse", 5)
    Hs = {}
    for pipe in (0, 1):
        ctx.set_option("gram2_pipe", pipe)
        it = kk.BlockLanczosIterator(kk.SparseOperator(A, ctx, symmetric=True), x0, (steps + 2) * bs)
        f = it.initialize()
        for _ in range(steps):
            f = it.expand(f)
            assert f.R_size == bs and not f.last_drift
        k = len(f)
        V = f.V.to_numpy()
        assert np.max(np.abs(V.T @ V - np.eye(k))) < 1e-12
        Hs[pipe] = f.H[:k, :k].copy()
    ctx.set_option("gram2_pipe", 1)
    assert np.max(np.abs(Hs[0] - Hs[1])) < 1e-12 * np.max(np.abs(Hs[0]))
