"""Randomised (hypothesis) shapes for the streaming kernels: every (n, m) combination around the
tile boundaries (512-row sub-steps, 2048-row groups, 4-column batches, 64-column accumulator groups,
16-column MFMA tiles) must agree with NumPy."""
import numpy as np
import pytest
from hypothesis import given, settings, strategies as st, HealthCheck

pytestmark = pytest.mark.gpu
SET = dict(max_examples=30, deadline=None, suppress_health_check=[HealthCheck.function_scoped_fixture])

sizes = st.one_of(st.integers(1, 40), st.integers(500, 530), st.integers(2040, 2060), st.integers(4090, 4110),
                  st.integers(1, 20000))
cols = st.one_of(st.integers(1, 9), st.integers(60, 70), st.integers(125, 132), st.integers(1, 140))


def _fill(kk, ctx, V):
    n, m = V.shape
    B = kk.DeviceBasis(n, m + 2, ctx)
    for j in range(m):
        B.upload(j, V[:, j])
    B.length = m
    return B


@settings(**SET)
@given(n=sizes, m=cols, seed=st.integers(0, 2**31 - 1))
def test_project_unproject_random(kk, ctx, n, m, seed):
    rng = np.random.default_rng(seed)
    V, w, x = rng.standard_normal((n, m)), rng.standard_normal(n), rng.standard_normal(m)
    B = _fill(kk, ctx, V)
    vw = B[m].set(w)
    y = B.project(vw)
    tol = 1e-13 * (np.linalg.norm(w) * np.sqrt(n) + 1)
    assert np.max(np.abs(y - V.T @ w)) <= tol * np.max(np.abs(V))
    B.unproject(vw, x, alpha=-1.0, beta=1.0)
    np.testing.assert_allclose(vw.get(), w - V @ x, rtol=1e-12, atol=1e-12 * (1 + np.abs(V) @ np.abs(x)).max())


@settings(**SET)
@given(n=sizes, m=st.integers(1, 128), seed=st.integers(0, 2**31 - 1), alg=st.sampled_from(["cgs2", "mgs2", "cgs", "mgs"]))
def test_orthogonalize_random(kk, ctx, n, m, seed, alg):
    m = min(m, n)
    rng = np.random.default_rng(seed)
    Q, _ = np.linalg.qr(rng.standard_normal((n, m)))
    w = rng.standard_normal(n)
    B = _fill(kk, ctx, Q)
    vw = B[m].set(w)
    x, nrm, _ = B.orthogonalize(vw, kk.Orthogonalizer(alg))
    wd = vw.get()
    nw = np.linalg.norm(w)
    assert np.max(np.abs(Q.T @ wd)) <= 1e-12 * nw
    np.testing.assert_allclose(wd + Q @ x, w, rtol=0, atol=1e-12 * nw)
    assert abs(nrm - np.linalg.norm(wd)) <= 1e-12 * nw


@settings(**SET)
@given(n=sizes, p=st.integers(1, 40), q=st.integers(1, 20), seed=st.integers(0, 2**31 - 1))
def test_block_gram_update_random(kk, ctx, n, p, q, seed):
    rng = np.random.default_rng(seed)
    X, Y = rng.standard_normal((n, p)), rng.standard_normal((n, q))
    S = kk.DeviceBasis(n, p + q, ctx)
    for j in range(p):
        S.upload(j, X[:, j])
    for j in range(q):
        S.upload(p + j, Y[:, j])
    M = kk.block_inner(kk.Block(S, 0, p), kk.Block(S, p, q))
    ref = X.T @ Y
    assert np.max(np.abs(M - ref)) <= 1e-13 * (np.sqrt(n) + 1) * (np.abs(X).T @ np.abs(Y)).max() + 1e-300
    from krylovkit_hip._lib import check, c_dp
    Sm = np.asfortranarray(rng.standard_normal((p, q)))
    norms = np.zeros(q)
    check(S._lib.kk_block_update(S.handle, p, q, S.handle, 0, p, Sm.ctypes.data_as(c_dp), p, -1.0, 1.0, norms.ctypes.data_as(c_dp)))
    refW = Y - X @ Sm
    W = np.stack([S.download(p + j) for j in range(q)], 1)
    np.testing.assert_allclose(W, refW, rtol=1e-12, atol=1e-12 * (1 + (np.abs(X) @ np.abs(Sm)).max()))
    np.testing.assert_allclose(norms, np.linalg.norm(refW, axis=0), rtol=1e-12, atol=1e-300)


@settings(**SET)
@given(n=sizes, m=st.integers(1, 120), frac=st.floats(0.1, 1.0), seed=st.integers(0, 2**31 - 1))
def test_basistransform_random(kk, ctx, n, m, frac, seed):
    k = max(1, int(m * frac))
    rng = np.random.default_rng(seed)
    V, U = rng.standard_normal((n, m)), rng.standard_normal((m, k))
    B = _fill(kk, ctx, V)
    B.basistransform(U)
    got = B.to_numpy()
    scale = (np.abs(V) @ np.abs(U)).max() + 1
    assert np.max(np.abs(got[:, :k] - V @ U)) <= 1e-13 * scale
    np.testing.assert_array_equal(got[:, k:], V[:, k:])


@settings(**SET)
@given(nrows=st.integers(1, 3000), ncols=st.integers(1, 3000), dens=st.floats(0.0005, 0.05), seed=st.integers(0, 2**31 - 1),
       fmt=st.sampled_from(["auto", "csr", "sell", "tiled17", "tiled300"]))
def test_spmv_random(kk, ctx, monkeypatch, nrows, ncols, dens, seed, fmt):
    import scipy.sparse as sp
    monkeypatch.delenv("KK_SPMV_FORMAT", raising=False)
    monkeypatch.delenv("KK_SPMV_TILE_COLS", raising=False)
    if fmt.startswith("tiled"):                      # column-tiled SELL forced onto small shapes by a tiny tile width
        monkeypatch.setenv("KK_SPMV_TILE_COLS", fmt[5:])
    elif fmt != "auto":
        monkeypatch.setenv("KK_SPMV_FORMAT", fmt)
    rng = np.random.default_rng(seed)
    A = sp.random(nrows, ncols, density=dens, random_state=rng.integers(1 << 30), format="csr")
    A.data[:] = rng.standard_normal(A.nnz)
    op = kk.SparseOperator(A, ctx)
    X, Y = kk.DeviceBasis(ncols, 2, ctx), kk.DeviceBasis(nrows, 2, ctx)
    x, u = rng.standard_normal(ncols), rng.standard_normal(nrows)
    op.apply(X[0].set(x), Y[0])
    np.testing.assert_allclose(Y[0].get(), A @ x, rtol=1e-12, atol=1e-12 * (1 + (abs(A) @ abs(x)).max()))
    op.apply_adjoint(Y[1].set(u), X[1])
    np.testing.assert_allclose(X[1].get(), A.T @ u, rtol=1e-12, atol=1e-12 * (1 + (abs(A.T) @ abs(u)).max()))


@settings(max_examples=20, deadline=None, suppress_health_check=[HealthCheck.function_scoped_fixture])
@given(nx=st.one_of(st.integers(64, 70), st.integers(120, 130), st.integers(64, 400)), ny=st.integers(12, 70),
       nine=st.booleans(), nb=st.integers(1, 16), drop=st.integers(0, 63), seed=st.integers(0, 2**31 - 1))
def test_grid_stencil_kernels_random(kk, ctx, nx, ny, nine, nb, drop, seed):
    """Diagonal SpMV / sweeping SpMM of detected grid stencils for random line lengths (around the 62-position wave strip),
    line counts, block widths and a truncated last line, against SciPy and against the gather kernels."""
    import scipy.sparse as sp
    from krylovkit_hip._lib import check
    if nx * ny - drop < 4200:       # the detector leaves small operators alone
        ny = 4200 // nx + 2
    rng = np.random.default_rng(seed)
    n = nx * ny
    ix, iy = np.meshgrid(np.arange(nx), np.arange(ny))
    ix, iy = ix.ravel(), iy.ravel()
    rows, cols_, vals = [], [], []
    for dy, dx in [(-1, 0), (0, -1), (0, 0), (0, 1), (1, 0)] + ([(-1, -1), (-1, 1), (1, -1), (1, 1)] if nine else []):
        ok = (ix + dx >= 0) & (ix + dx < nx) & (iy + dy >= 0) & (iy + dy < ny)
        r = (iy * nx + ix)[ok]
        rows.append(r); cols_.append(r + dy * nx + dx); vals.append(rng.standard_normal(r.size))
    A = sp.csr_matrix((np.concatenate(vals), (np.concatenate(rows), np.concatenate(cols_))), shape=(n, n))
    if drop:
        A = A[:n - drop, :n - drop].tocsr()
        n -= drop
    op = kk.SparseOperator(A, ctx)
    assert op.info()["format"].startswith("ELL+DIA")
    X = rng.standard_normal((n, nb))
    S = kk.DeviceBasis(n, 2 * nb + 1, ctx)
    for j in range(nb):
        S.upload(j, X[:, j])
    ref = A @ X
    scale = (np.abs(A) @ np.abs(X)).max() + 1e-300
    for dia in (1, 0):
        ctx.set_option("spmv_dia", dia); ctx.set_option("spmm_dia", dia)
        check(S._lib.kk_block_apply(op.handle, S.handle, 0, S.handle, nb, nb))
        Y = np.stack([S.download(nb + j) for j in range(nb)], 1)
        assert np.max(np.abs(Y - ref)) <= 1e-13 * scale, (nx, ny, nine, nb, drop, dia)
        op.apply(S[0], S[2 * nb])
        assert np.max(np.abs(S[2 * nb].get() - ref[:, 0])) <= 1e-13 * scale
    ctx.set_option("spmv_dia", 1); ctx.set_option("spmm_dia", 1)
    S.free()


@settings(max_examples=25, deadline=None, suppress_health_check=[HealthCheck.function_scoped_fixture])
@given(half_nx=st.one_of(st.integers(32, 35), st.integers(60, 68), st.integers(250, 262), st.integers(32, 1100)), ny=st.integers(3, 60),
       ns=st.integers(1, 2), lines=st.sampled_from([0, 2, 3, 4, 5, 8, 16]), seed=st.integers(0, 2**31 - 1), epi=st.integers(0, 3))
def test_sweeping_constant_stencil_apply_random(kk, ctx, half_nx, ny, ns, lines, seed, epi):
    """k_spmv_dia_sw for random EVEN line lengths (around the 128-position strip and the 512 / 1024-position block chunk), line counts,
    strips per wave and sweep lengths, random coefficients: bit-identical to k_spmv_dia, equal to SciPy; the fused inner product and
    norm of kk_spmv_affine_dot to rounding (reference: src/apply.jl:1, factorizations/lanczos.jl:306-310)"""
    import scipy.sparse as sp
    nx = 2 * half_nx
    if nx * ny < 4200:       # the detector leaves small operators alone
        ny = 4200 // nx + 2
    rng = np.random.default_rng(seed)
    n = nx * ny
    c = rng.standard_normal(5)
    i = np.arange(n); ix = i % nx
    rows, cols_, vals = [], [], []
    for q, off in enumerate((-nx, -1, 0, 1, nx)):
        ok = (i + off >= 0) & (i + off < n)
        if off == -1:
            ok &= ix > 0
        if off == 1:
            ok &= ix < nx - 1
        rows.append(i[ok]); cols_.append(i[ok] + off); vals.append(np.full(ok.sum(), c[q]))
    A = sp.csr_matrix((np.concatenate(vals), (np.concatenate(rows), np.concatenate(cols_))), shape=(n, n))
    op = kk.SparseOperator(A, ctx)
    assert op.info()["format"] == "ELL+DIA const"
    S = kk.DeviceBasis(n, 4, ctx)
    x = rng.standard_normal(n)
    S.upload(0, x)
    a0, a1 = (0.0, 1.0) if epi == 0 else (float(rng.standard_normal()), float(rng.standard_normal()))
    outs = {}
    for sw in (0, ns):
        ctx.set_option("spmv_dia_sw", sw); ctx.set_option("spmv_dia_sw_lines", lines)
        if epi >= 2:   # <x, a0 x + a1 A x> fused into the apply (kk_spmv_affine_dot: the CG / Lanczos epilogue)
            import ctypes as C
            from krylovkit_hip._lib import check
            dd = C.c_double()
            check(ctx._lib.kk_spmv_affine_dot(op.handle, S.handle, 0, S.handle, 1 + (sw > 0), a0, a1, C.byref(dd)))
            d = dd.value
        else:
            op.apply_affine(S[0], S[1 + (sw > 0)], a0, a1); d = 0.0
        outs[sw] = (S[1 + (sw > 0)].get(), d)
    ctx.set_option("spmv_dia_sw", 1); ctx.set_option("spmv_dia_sw_lines", 0)
    assert np.array_equal(outs[0][0], outs[ns][0]), (nx, ny, ns, lines)
    ref = a0 * x + a1 * (A @ x)
    scale = np.abs(a0 * x).max() + abs(a1) * (np.abs(A) @ np.abs(x)).max() + 1e-300
    assert np.max(np.abs(outs[ns][0] - ref)) <= 1e-13 * scale
    if epi >= 2:
        assert abs(outs[ns][1] - outs[0][1]) <= 1e-12 * (np.abs(x) @ np.abs(ref) + 1e-300)
    S.free()
