"""The library's own multi-GPU layer (kk_comm_* / kk_csr_create_sharded*, RCCL inside libkrylov_hip.so) on the ONE GPU
of the box: a communicator of world size 1 with KK_COMM_FORCE_COLLECTIVES issues every collective for real
(ncclAllReduce / ncclAllGather / ncclReduceScatter on the context stream, no torch anywhere), so the sharded code paths --
the two-all-reduce Lanczos step, the gather / scatter GKL step, the generic finalize sites -- run exactly as they do on
8 GPUs and must reproduce the serial oracle.  (Two logical ranks are covered by tests/test_gpu_sharded_hooks.py through
the hook mechanism, which enters the same fused code; real multi-GPU runs are the driver's.)"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def relerr(a, b):
    a, b = np.asarray(a, float), np.asarray(b, float)
    return float(np.max(np.abs(a - b) / np.maximum(np.abs(b), 1e-300)))


@pytest.fixture()
def comm(kk):
    from krylovkit_hip import dist as kd
    c = kk.Context(0)
    cm = kd.NativeComm.single(c, force_collectives=True)
    yield cm
    cm.close()
    c.close()


def orth_pairs(kk, ko):
    return [(kk.ClassicalGramSchmidt(), ko.CGS), (kk.ModifiedGramSchmidt(), ko.MGS), (kk.ClassicalGramSchmidt2(), ko.CGS2),
            (kk.ModifiedGramSchmidt2(), ko.MGS2), (kk.ClassicalGramSchmidtIR(0.75), ko.CGSIR(0.75)),
            (kk.ModifiedGramSchmidtIR(0.75), ko.MGSIR(0.75))]


def test_comm_info_and_scalar_allreduce(kk, comm):
    info = comm.info()
    assert info["rank"] == 0 and info["world"] == 1 and info["rccl_version"] >= 20000
    assert comm.allreduce_scalar(3.25, "max") == 3.25 and comm.allreduce_scalar(-1.5, "sum") == -1.5
    comm.barrier()
    assert comm.stats()["allreduce"] >= 2   # forced: the collectives really ran


@pytest.mark.parametrize("mgs_mode", [0, 1])
def test_native_sharded_lanczos_matches_oracle(kk, ko, comm, mgs_mode):
    """factorizations/lanczos.jl:250-376 through kk_lanczos_expand with the library's communicator active."""
    from krylovkit_hip import dist as kd
    ctx = comm.ctx
    ctx.set_option("mgs_mode", mgs_mode)
    nx, ny, steps = 40, 30, 25
    n = nx * ny
    A = ko.laplacian_2d(nx, ny, shift_diag=10 * np.linspace(0, 1, n) ** 2)
    x0 = np.random.default_rng(3).random(n)
    part = kd.Partition.even(n, 1, 0, align=nx)
    op = kd.NativeShardedOperator(A, part, ctx, symmetric=True)
    for dev, ref in orth_pairs(kk, ko):
        it = kk.LanczosIterator(op, x0, dev, capacity=steps + 3)
        f = kk.initialize(it)
        oit = ko.LanczosIterator(A, x0.copy(), ref)
        of = ko.lanczos_initialize(oit)
        before = comm.stats()["allreduce"]
        for _ in range(steps):
            f = kk.expand_(it, f)
            of = ko.lanczos_expand(oit, of)
        per_step = (comm.stats()["allreduce"] - before) / steps
        tol = 1e-10 if dev.is_reorth else 1e-6
        assert relerr(f.alphas, of.alphas) < tol and relerr(f.betas, of.betas) < tol, dev.name
        V = f.V.to_numpy()
        if dev.is_reorth:
            assert np.max(np.abs(V.T @ V - np.eye(V.shape[1]))) < 1e-12
        if dev.name == "cgs2" or (dev.name == "mgs2" and mgs_mode == 1):
            assert per_step == 2.0, (dev.name, per_step)   # [alpha0 | V'w | V'v] and |w|^2: SURVEY.md 8(e)
        # thick-restart style mutation + continue: the Gram bookkeeping of the low-sync path must recover
        if dev.name == "mgs2":
            k = len(f)
            Q, _ = np.linalg.qr(np.random.default_rng(1).standard_normal((k, k)))
            f.V.basistransform(Q)
            Vn = f.V.to_numpy()
            w = kk.DeviceBasis(n, 1, ctx)[0].set(np.random.default_rng(2).random(n))
            x, nrm, _ = f.V.orthogonalize(w, dev)
            assert np.max(np.abs(Vn.T @ w.get())) < 1e-12 * max(1.0, nrm)
    ctx.set_option("mgs_mode", 2)


def test_native_sharded_eigsolve_and_gmres(kk, ko, comm):
    from krylovkit_hip import dist as kd
    ctx = comm.ctx
    nx, ny = 32, 24
    n = nx * ny
    A = ko.laplacian_2d(nx, ny, shift_diag=10 * np.linspace(0, 1, n) ** 2)
    x0 = np.random.default_rng(3).random(n)
    part = kd.Partition.even(n, 1, 0)
    op = kd.NativeShardedOperator(A, part, ctx, symmetric=True)
    vals, vecs, info = kk.eigsolve(op, x0, 3, "LM", krylovdim=20, tol=1e-10, maxiter=50, orth=kk.ModifiedGramSchmidt2())
    ovals, ovecs, oinfo = ko.eigsolve_lanczos(A, x0.copy(), 3, "LM", krylovdim=20, tol=1e-10, maxiter=50, orth=ko.MGS2)
    assert info.converged >= 3 and (info.numiter, info.numops) == (oinfo.numiter, oinfo.numops)
    assert relerr(vals[:3], ovals[:3]) < 1e-10
    B = ko.convection_diffusion_2d(nx, ny)
    b = np.random.default_rng(4).random(n)
    tol = 1e-9 * np.linalg.norm(b)
    opB = kd.NativeShardedOperator(B, part, ctx)
    tr, otr = [], []
    x, ginfo = kk.linsolve(opB, b, None, kk.GMRES(kk.ModifiedGramSchmidt2(), 20, 25, tol), trace=tr)
    ox, oinfo = ko.gmres(B, b, krylovdim=25, maxiter=20, tol=tol, orth=ko.MGS2, trace=otr)
    assert (ginfo.converged, ginfo.numiter, ginfo.numops) == (oinfo.converged, oinfo.numiter, oinfo.numops)
    assert len(tr) == len(otr) and relerr([t[2] for t in tr], [t[2] for t in otr]) < 1e-6
    assert np.linalg.norm(B @ x - b) < 2 * tol


@pytest.mark.parametrize("mgs_mode", [0, 1])
def test_native_sharded_gkl_matches_oracle(kk, ko, comm, mgs_mode):
    """factorizations/gkl.jl:183-404 on a kk_csr_create_sharded_rect map: all-gather before A v, reduce-scatter after A'u."""
    from krylovkit_hip import dist as kd
    ctx = comm.ctx
    ctx.set_option("mgs_mode", mgs_mode)
    A = ko.sparse_random(600, 250, 8, 21)
    u0 = np.random.default_rng(6).random(600)
    steps = 15
    op = kd.NativeShardedRectOperator(A, 250, ctx)
    assert op.shape == (600, 250)
    for dev, ref in orth_pairs(kk, ko):
        it = kk.GKLIterator(op, u0, dev, capacity=steps + 3)
        f = kk.initialize(it)
        oit = ko.GKLIterator(A, u0.copy(), ref)
        of = ko.gkl_initialize(oit)
        g0 = comm.stats()["gather"]
        for _ in range(steps):
            f = kk.expand_(it, f)
            of = ko.gkl_expand(oit, of)
        assert comm.stats()["gather"] - g0 == 2 * steps   # one all-gather + one reduce-scatter per expand!
        tol = 1e-10 if dev.is_reorth else 1e-6
        assert relerr(f.alphas, of.alphas) < tol and relerr(f.betas, of.betas) < tol, dev.name
        U, V, r, Bm = f.U.to_numpy(), f.V.to_numpy(), f.r.get(), f.rayleighquotient()
        ek = np.zeros(len(f)); ek[-1] = 1
        assert np.max(np.abs(A @ V - U @ Bm - np.outer(r, ek))) < 1e-11
        assert np.max(np.abs(A.T @ U - V @ Bm.T)) < 1e-10
    # apply / apply_adjoint through the public kk_spmv
    xb, yb = kk.DeviceBasis(250, 1, ctx), kk.DeviceBasis(600, 1, ctx)
    xv = np.random.default_rng(9).standard_normal(250)
    op.apply(xb[0].set(xv), yb[0])
    np.testing.assert_allclose(yb[0].get(), A @ xv, rtol=0, atol=1e-12)
    yv = np.random.default_rng(10).standard_normal(600)
    op.apply_adjoint(yb[0].set(yv), xb[0])
    np.testing.assert_allclose(xb[0].get(), A.T @ yv, rtol=0, atol=1e-12)
    ctx.set_option("mgs_mode", 2)


def test_native_sharded_blocklanczos_issue143(kk, ko, comm):
    """test/issues.jl:39-129 known answer through the sharded finalize sites (block Gram panels all-reduced)."""
    from pathlib import Path
    from krylovkit_hip import dist as kd
    import scipy.sparse as sp
    ctx = comm.ctx
    A = np.load(Path(__file__).parent / "golden" / "issue143_A.npy")
    n = A.shape[0]
    part = kd.Partition.even(n, 1, 0)
    op = kd.NativeShardedOperator(sp.csr_matrix(A), part, ctx, symmetric=True)
    rng = np.random.default_rng(143)
    x0 = [rng.standard_normal(n) for _ in range(20)]
    D, V, info = kk.eigsolve_block(op, x0, 4, "SR", kk.BlockLanczos(tol=1e-8))
    ev = np.linalg.eigvalsh(A)
    assert len(D) == len(ev)
    np.testing.assert_allclose(np.sort(D), ev, rtol=0, atol=1e-10 * np.max(np.abs(ev)))
    assert info.converged == len(D) and info.numiter == 1 and info.numops == len(D) + 1


def test_sharded_create_rejects_bad_input(kk, comm):
    import ctypes as C
    from krylovkit_hip import _lib
    lib, ctx = comm.ctx._lib, comm.ctx
    h = C.c_void_p()
    offs = (C.c_int64 * 2)(0, 4)
    rowptr = (C.c_int64 * 5)(0, 1, 2, 3, 4)
    col = (C.c_int64 * 4)(0, 1, 2, 9)     # column 9 outside the 4 x 4 operator
    val = (C.c_double * 4)(1, 1, 1, 1)
    assert lib.kk_csr_create_sharded(ctx.handle, 4, offs, 4, rowptr, col, val, 0, 0, C.byref(h)) == _lib.KK_ERR_DIM
    bad_ptr = (C.c_int64 * 5)(0, 2, 1, 3, 4)   # not monotone
    col2 = (C.c_int64 * 4)(0, 1, 2, 3)
    assert lib.kk_csr_create_sharded(ctx.handle, 4, offs, 4, bad_ptr, col2, val, 0, 0, C.byref(h)) == _lib.KK_ERR_DIM
    assert lib.kk_csr_create_sharded(ctx.handle, 3, offs, 4, rowptr, col2, val, 0, 0, C.byref(h)) == _lib.KK_ERR_DIM


def test_loopback_ghost_plan_runs_the_whole_exchange_path(kk, ko, comm, monkeypatch):
    """One-GPU coverage of the multi-GPU data path short of the wire (SURVEY.md 8(e) "loopback shard mode"): with
    KK_LOOPBACK_GHOST_FROM the columns >= r of a world-1 operator go through the native ghost plan -- request list, gather
    into the send buffer, exchange (device copy to self), ghost-indexed reads of k_spmv_* and of the block apply k_spmm_ell.
    Results must equal those of the plain operator / the oracle."""
    from krylovkit_hip import dist as kd
    ctx = comm.ctx
    nx, ny = 36, 28
    n = nx * ny
    A = ko.laplacian_2d(nx, ny, shift_diag=10 * np.linspace(0, 1, n) ** 2)
    part = kd.Partition.even(n, 1, 0)
    monkeypatch.setenv("KK_LOOPBACK_GHOST_FROM", str(n // 2 + 5))
    op = kd.NativeShardedOperator(A, part, ctx, symmetric=True)
    monkeypatch.delenv("KK_LOOPBACK_GHOST_FROM")
    assert op.info()["ncols"] > n            # ghost columns were appended
    rng = np.random.default_rng(12)
    B = kk.DeviceBasis(n, 40, ctx)
    X = rng.standard_normal((n, 16))
    for j in range(16):
        B.upload(j, X[:, j])
    # single vector, affine form, fused dot
    op.apply(B[3], B[20])
    np.testing.assert_allclose(B[20].get(), A @ X[:, 3], rtol=0, atol=1e-12)
    op.apply_affine(B[3], B[21], 0.7, -0.4)
    np.testing.assert_allclose(B[21].get(), 0.7 * X[:, 3] - 0.4 * (A @ X[:, 3]), rtol=0, atol=1e-12)
    # block apply: ONE exchange for 16 vectors, ghost-aware SpMM
    p0 = comm.stats()
    import ctypes as C
    from krylovkit_hip._lib import check
    check(ctx._lib.kk_block_apply(op.handle, B.handle, 0, B.handle, 20, 16))
    Y = np.stack([B.download(20 + j) for j in range(16)], 1)
    np.testing.assert_allclose(Y, A @ X, rtol=0, atol=1e-11)
    for nb in (3, 8, 11):
        check(ctx._lib.kk_block_apply(op.handle, B.handle, 1, B.handle, 20, nb))
        Y = np.stack([B.download(20 + j) for j in range(nb)], 1)
        np.testing.assert_allclose(Y, A @ X[:, 1:1 + nb], rtol=0, atol=1e-11)
    # a Lanczos run and a BlockLanczos run on the looped-back operator against the oracle
    x0 = rng.random(n)
    it = kk.LanczosIterator(op, x0, kk.ModifiedGramSchmidt2(), capacity=24)
    f = kk.initialize(it)
    oit = ko.LanczosIterator(A, x0.copy(), ko.MGS2)
    of = ko.lanczos_initialize(oit)
    for _ in range(20):
        f = kk.expand_(it, f)
        of = ko.lanczos_expand(oit, of)
    assert relerr(f.alphas, of.alphas) < 1e-10 and relerr(f.betas, of.betas) < 1e-10
    xb = [rng.random(n) for _ in range(4)]
    bit = kk.BlockLanczosIterator(op, xb, 28)
    bf = bit.initialize()
    obit = ko.BlockLanczosIterator(A, [x.copy() for x in xb], 28)
    obf = ko.blocklanczos_initialize(obit)
    for _ in range(4):
        bf = bit.expand(bf)
        obf = ko.blocklanczos_expand(obit, obf)
    k = len(bf)
    np.testing.assert_allclose(np.linalg.eigvalsh(bf.H[:k, :k]), np.linalg.eigvalsh(obf.H[:k, :k]), atol=1e-9)


@pytest.mark.parametrize("nine", [False, True])
def test_loopback_sharded_stencil_uses_diagonal_kernels_on_the_interior(kk, ko, comm, monkeypatch, nine):
    """A row-sharded grid stencil runs the diagonal kernels on its ghost-free interior rows and the gather kernels (ghost
    buffer) on the boundary strips.  One-GPU coverage through the loop-back plan: the last grid line(s) of a world-1 operator
    are routed through the ghost machinery, so the rows that reference them form the boundary strip."""
    import scipy.sparse as sp
    from krylovkit_hip import dist as kd
    from krylovkit_hip._lib import check
    ctx = comm.ctx
    rng = np.random.default_rng(31 + nine)
    nx, ny = 70, 66
    n = nx * ny
    if nine:   # symmetric 9-point operator with row-dependent coefficients
        ix, iy = np.meshgrid(np.arange(nx), np.arange(ny)); ix, iy = ix.ravel(), iy.ravel()
        rows, cols, vals = [], [], []
        for dy, dx in ((0, 1), (1, -1), (1, 0), (1, 1)):
            ok = (ix + dx >= 0) & (ix + dx < nx) & (iy + dy < ny)
            r = (iy * nx + ix)[ok]
            rows.append(r); cols.append(r + dy * nx + dx); vals.append(rng.standard_normal(r.size))
        U = sp.csr_matrix((np.concatenate(vals), (np.concatenate(rows), np.concatenate(cols))), shape=(n, n))
        A = (U + U.T + sp.diags(4.0 + rng.random(n))).tocsr()
    else:
        A = ko.laplacian_2d(nx, ny, shift_diag=10 * np.linspace(0, 1, n) ** 2)
    part = kd.Partition.even(n, 1, 0)
    ghost_from = n - nx - 3                      # a bit more than the last grid line goes through the ghost buffer ...
    monkeypatch.setenv("KK_LOOPBACK_GHOST_FROM", str(ghost_from))
    monkeypatch.setenv("KK_LOOPBACK_GHOST_BELOW", str(nx + 5))   # ... and the first one: a middle rank, two boundary strips
    op = kd.NativeShardedOperator(A, part, ctx, symmetric=True)
    monkeypatch.delenv("KK_LOOPBACK_GHOST_FROM")
    monkeypatch.delenv("KK_LOOPBACK_GHOST_BELOW")
    assert op.info()["ncols"] > n and op.info()["format"].startswith("ELL+DIA")
    B = kk.DeviceBasis(n, 40, ctx)
    X = rng.standard_normal((n, 16))
    for j in range(16):
        B.upload(j, X[:, j])
    for dia in (1, 0):
        ctx.set_option("spmv_dia", dia); ctx.set_option("spmm_dia", dia)
        ctx.prof_reset(); ctx.prof_enable(1)
        op.apply(B[3], B[20])
        ctx.prof_enable(0)
        assert (ctx.prof_get("k_spmv_dia")[1] > 0) == bool(dia)
        np.testing.assert_allclose(B[20].get(), A @ X[:, 3], rtol=0, atol=1e-12)
        op.apply_affine(B[3], B[21], 0.7, -0.4)
        np.testing.assert_allclose(B[21].get(), 0.7 * X[:, 3] - 0.4 * (A @ X[:, 3]), rtol=0, atol=1e-12)
        for nb in (2, 5, 8, 16):
            ctx.prof_reset(); ctx.prof_enable(1)
            check(ctx._lib.kk_block_apply(op.handle, B.handle, 0, B.handle, 20, nb))
            ctx.prof_enable(0)
            assert (ctx.prof_get("k_spmm_dia")[1] > 0) == bool(dia) and ctx.prof_get("k_spmm_ell")[1] > 0   # interior + strips
            Y = np.stack([B.download(20 + j) for j in range(nb)], 1)
            np.testing.assert_allclose(Y, A @ X[:, :nb], rtol=0, atol=1e-11)
        # fused Lanczos epilogues (inner products and norms summed over the three launches of one apply)
        x0 = np.random.default_rng(5).random(n)
        it = kk.LanczosIterator(op, x0, kk.ModifiedGramSchmidt2(), capacity=24)
        f = kk.initialize(it)
        oit = ko.LanczosIterator(A, x0.copy(), ko.MGS2)
        of = ko.lanczos_initialize(oit)
        for _ in range(20):
            f = kk.expand_(it, f)
            of = ko.lanczos_expand(oit, of)
        assert relerr(f.alphas, of.alphas) < 1e-10 and relerr(f.betas, of.betas) < 1e-10
    ctx.set_option("spmv_dia", 1); ctx.set_option("spmm_dia", 1)
