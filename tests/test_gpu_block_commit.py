"""Normalised commit of the residual block in the one-pass BlockLanczos step (csrc/kk_block.hip, k_block_update_commit): the
residual update writes T = W R1^-1 -- R1 from the Gram matrix the panel predicts -- straight into the next basis slot and
accumulates T'T, so the next expand! starts at the second CholQR2 round; the residual block W itself exists only on demand
(W = T R1, formed when anything looks at the slab).  Reference semantics being kept: blocklanczos.jl:209-216 (block_qr! of
the residual block), :242-263 (recurrence), :277-284 (re-orthogonalisation); test invariants of test/factorize.jl:387-401."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def problem(ko, nx=48, ny=30):
    n = nx * ny
    return ko.laplacian_2d(nx, ny, shift_diag=10 * np.linspace(0, 1, n) ** 2), n


def run(kk, ctx, A, x0, steps, commit, look=False):
    bs = len(x0)
    ctx.set_option("block_async", 1)
    ctx.set_option("block_fuse", 5)
    ctx.set_option("block_commit", commit)
    c0 = ctx.get_option("block_commits")
    it = kk.BlockLanczosIterator(kk.SparseOperator(A, ctx, symmetric=True), x0, (steps + 2) * bs)
    f = it.initialize()
    nrm = []
    for _ in range(steps):
        f = it.expand(f)
        assert f.R_size == bs and not f.last_drift
        nrm.append(f.normres)
        if look:   # looking at the residual block settles a pending commit: W = T R1 is written to its area
            R = np.stack([f.residual()[j].get() for j in range(bs)], 1)
            assert abs(np.linalg.norm(R) - f.normres) < 1e-11 * max(1.0, f.normres)
    k = len(f)
    V = f.V.to_numpy()
    R = np.stack([f.residual()[j].get() for j in range(bs)], 1)
    return dict(H=f.H[:k, :k].copy(), V=V, R=R, nrm=np.array(nrm), commits=ctx.get_option("block_commits") - c0, k=k)


@pytest.mark.parametrize("bs", [2, 5, 8, 16])
def test_commit_path_gives_the_same_factorization(kk, ko, ctx, bs):
    A, n = problem(ko)
    rng = np.random.default_rng(300 + bs)
    x0 = [rng.random(n) for _ in range(bs)]
    steps = 6
    a = run(kk, ctx, A, x0, steps, commit=1)
    b = run(kk, ctx, A, x0, steps, commit=0)
    ctx.set_option("block_commit", 1)
    assert a["commits"] == steps and b["commits"] == 0              # every step consumed a commit (the first one that of initialize: round 6)
    k = a["k"]
    E = np.zeros((k, bs)); E[k - bs:, :] = np.eye(bs)
    for r in (a, b):
        assert np.max(np.abs(r["V"].T @ r["V"] - np.eye(k))) < 1e-12
        assert np.max(np.abs(A @ r["V"] - r["V"] @ r["H"] - r["R"] @ E.T)) < 1e-10
        assert np.max(np.abs(r["V"].T @ r["R"])) < 1e-11
    scale = np.max(np.abs(b["H"]))
    assert np.max(np.abs(a["H"] - b["H"])) < 1e-11 * scale
    np.testing.assert_allclose(a["nrm"], b["nrm"], rtol=1e-11)
    assert np.max(np.abs(a["R"] - b["R"])) < 1e-10 * np.max(np.abs(b["R"]))
    assert np.max(np.abs(a["V"] - b["V"])) < 1e-9


def test_looking_at_the_residual_block_settles_the_commit(kk, ko, ctx):
    A, n = problem(ko)
    rng = np.random.default_rng(41)
    x0 = [rng.random(n) for _ in range(4)]
    # 12 steps: after two settled commits in a row the library pauses committing (2 << streak steps), so this also runs the
    # plain step in between and the re-armed commit afterwards
    a = run(kk, ctx, A, x0, 12, commit=1, look=True)
    b = run(kk, ctx, A, x0, 12, commit=0, look=True)
    ctx.set_option("block_commit", 1)
    assert a["commits"] == 1                 # every commit of a STEP was flushed before the next expand! could take it (initialize's own was consumed)
    assert np.max(np.abs(a["H"] - b["H"])) < 1e-11 * np.max(np.abs(b["H"]))
    assert np.max(np.abs(a["R"] - b["R"])) < 1e-10 * np.max(np.abs(b["R"]))


def test_repeated_step_after_a_commit_recovers_the_block(kk, ko, ctx):
    """A + 1000 I: the one-pass projection loses more than a factor 10 of a column's norm, the device flag sends the step to
    the two-pass route (blocklanczos.jl:253-284) -- which needs the residual block the commit never wrote"""
    import scipy.sparse as sp
    nx, ny, bs = 30, 20, 4
    n = nx * ny
    A = (ko.laplacian_2d(nx, ny) + 1000.0 * sp.identity(n)).tocsr()
    rng = np.random.default_rng(11)
    x0 = [rng.random(n) for _ in range(bs)]
    ctx.set_option("block_async", 1)
    ctx.set_option("block_fuse", 5)
    ctx.set_option("block_commit", 1)
    it = kk.BlockLanczosIterator(kk.SparseOperator(A, ctx, symmetric=True), x0, 7 * bs)
    oit = ko.BlockLanczosIterator(A, [x.copy() for x in x0], 7 * bs)
    f, of = it.initialize(), ko.blocklanczos_initialize(oit)
    for _ in range(4):
        f = it.expand(f)
        of = ko.blocklanczos_expand(oit, of)
    k = len(f)
    V = f.V.to_numpy()
    R = np.stack([f.residual()[j].get() for j in range(bs)], 1)
    E = np.zeros((k, bs)); E[k - bs:, :] = np.eye(bs)
    assert np.max(np.abs(V.T @ V - np.eye(k))) < 1e-12
    assert np.max(np.abs(A @ V - V @ f.H[:k, :k] - R @ E.T)) < 1e-9
    np.testing.assert_allclose(np.linalg.eigvalsh(f.H[:k, :k]), np.linalg.eigvalsh(of.H[:k, :k]), rtol=1e-12)


def test_eigsolve_blocklanczos_with_restarts(kk, ko, ctx):
    """thick restarts transform the basis and keep the residual block: every restart settles the pending commit"""
    A, n = problem(ko, 40, 25)
    rng = np.random.default_rng(5)
    x0 = [rng.random(n) for _ in range(4)]
    vals = {}
    for commit in (1, 0):
        ctx.set_option("block_commit", commit)
        v, vecs, info = kk.eigsolve_block(kk.SparseOperator(A, ctx, symmetric=True), x0, 4, "SR",
                                          kk.BlockLanczos(krylovdim=24, tol=1e-10, maxiter=200))
        assert info.converged >= 4
        vals[commit] = np.asarray(v[:4])
    ctx.set_option("block_commit", 1)
    np.testing.assert_allclose(vals[1], vals[0], rtol=1e-10)
    w = np.linalg.eigvalsh(A.toarray())[:4]
    np.testing.assert_allclose(vals[1], w, rtol=1e-8)
