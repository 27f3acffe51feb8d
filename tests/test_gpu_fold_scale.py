"""The scale!!(r, 1/beta) of every expand! (factorizations/lanczos.jl:257, arnoldi.jl:209) and of orthonormalize!!
(orthonormal.jl:522-527) folded into the commit of the persistent strict-MGS kernel: the kernel holds |w| before
it writes w back and stores w / |w|; the slab remembers (column, beta), the next expand! of the same factorization takes the
column as its new basis vector, and ANY other access multiplies it back first, so residual(F), shrink! and restarts see r.

The stored bits are those of the separate scale pass (the same product w[i] * (1/|w|)), hence alpha / beta trajectories are
bitwise the same with the option on and off."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def relerr(a, b):
    a, b = np.asarray(a, float), np.asarray(b, float)
    return float(np.max(np.abs(a - b) / np.maximum(np.abs(b), 1e-300)))


@pytest.fixture()
def sctx(kk):
    c = kk.Context(0)
    c.set_option("mgs_mode", 0)          # strict order: the persistent kernel runs at every size
    c.set_option("mgs_panel", 0)         # these tests are about k_mgs_persist (the panel kernel has its own: tests/test_gpu_panel.py)
    if c.get_option("mgs_persist") == 0:
        pytest.skip("no cooperative launch on this device: the persistent route is off")
    yield c
    c.close()


def lanczos_run(kk, ctx, A, x0, steps, fold, poke=None, speculate=1):
    ctx.set_option("fold_scale", fold)
    ctx.set_option("speculate", speculate)
    op = kk.SparseOperator(A, ctx, symmetric=True)
    it = kk.LanczosIterator(op, x0, kk.ModifiedGramSchmidt2(), capacity=steps + 3)
    f = kk.initialize(it)
    seen = []
    for i in range(steps):
        f = kk.expand_(it, f)
        if poke is not None and i in poke:
            seen.append((i, f.r.get().copy(), f.normres))   # residual(F) in the middle of the run
    return f, seen


def test_lanczos_same_bits_with_and_without_the_fold(kk, ko, sctx):
    nx, ny, steps = 56, 44, 30
    n = nx * ny
    A = ko.laplacian_2d(nx, ny, shift_diag=10 * np.linspace(0, 1, n) ** 2)
    x0 = np.random.default_rng(3).random(n)
    # (speculation off for the bitwise comparison: the speculative apply of an un-normalised residual scales AFTER the row
    # sums, xs * (A r), where the plain order -- and the folded one -- applies A to the scaled vector)
    sctx.prof_reset(); sctx.prof_enable(1)
    f1, _ = lanczos_run(kk, sctx, A, x0, steps, 1, speculate=0)
    sctx.prof_enable(0)
    n_scal_fold = sctx.prof_get("k_scal")[1]
    assert sctx.prof_get("k_mgs_persist")[1] == steps
    a1, b1, V1, r1 = np.array(f1.alphas), np.array(f1.betas), f1.V.to_numpy().copy(), f1.r.get().copy()
    sctx.prof_reset(); sctx.prof_enable(1)
    f0, _ = lanczos_run(kk, sctx, A, x0, steps, 0, speculate=0)
    sctx.prof_enable(0)
    n_scal_plain = sctx.prof_get("k_scal")[1]
    sctx.set_option("fold_scale", 1)
    sctx.set_option("speculate", 1)
    # every expand! after the first one found its vector normalised already: steps - 1 scale launches fewer
    assert n_scal_plain - n_scal_fold == steps - 1
    assert np.array_equal(a1, np.array(f0.alphas)) and np.array_equal(b1, np.array(f0.betas))
    assert np.array_equal(V1, f0.V.to_numpy())
    # residual(F): handed back as beta * stored -> equal to the un-folded residual to an ulp or two, norm = beta
    r0 = f0.r.get()
    assert np.max(np.abs(r1 - r0)) <= 4 * np.finfo(float).eps * np.max(np.abs(r0))
    assert abs(np.linalg.norm(r1) - f1.normres) < 1e-13 * f1.normres
    # and against the oracle
    oit = ko.LanczosIterator(A, x0.copy(), ko.MGS2)
    of = ko.lanczos_initialize(oit)
    for _ in range(steps):
        of = ko.lanczos_expand(oit, of)
    assert relerr(a1, of.alphas) < 1e-10 and relerr(b1, of.betas) < 1e-10
    # with the speculative next-step apply on (the default) the folded run agrees to rounding
    f2, _ = lanczos_run(kk, sctx, A, x0, steps, 1)
    assert relerr(f2.alphas, a1) < 1e-13 and relerr(f2.betas, b1) < 1e-13


def test_residual_read_in_the_middle_of_a_run(kk, ko, sctx):
    """looking at residual(F) between two expand! calls settles the pending normalisation (one scale by beta) and the run
    goes on: every invariant of test/factorize.jl:140-148 holds, the trajectory matches the oracle"""
    nx, ny, steps = 40, 36, 20
    n = nx * ny
    A = ko.laplacian_2d(nx, ny, shift_diag=10 * np.linspace(0, 1, n) ** 2)
    x0 = np.random.default_rng(5).random(n)
    f, seen = lanczos_run(kk, sctx, A, x0, steps, 1, poke={3, 4, 11})
    oit = ko.LanczosIterator(A, x0.copy(), ko.MGS2)
    of = ko.lanczos_initialize(oit)
    for i in range(steps):
        of = ko.lanczos_expand(oit, of)
        for (j, r, nr) in seen:
            if j == i:
                assert abs(np.linalg.norm(r) - nr) < 1e-13 * nr
                assert np.max(np.abs(r - of.r)) < 1e-10 * np.linalg.norm(of.r)
    assert relerr(f.alphas, of.alphas) < 1e-10 and relerr(f.betas, of.betas) < 1e-10
    V = f.V.to_numpy()
    assert np.max(np.abs(V.T @ V - np.eye(V.shape[1]))) < 1e-12
    T = np.diag(f.alphas) + np.diag(f.betas[:-1], 1) + np.diag(f.betas[:-1], -1)
    R = A @ V - V @ T
    R[:, -1] -= f.r.get()
    assert np.max(np.abs(R)) < 1e-11 * abs(A).max()


def test_arnoldi_expand_and_shrink(kk, ko, sctx):
    """arnoldi.jl:199-219 with the fold, then shrink! (arnoldi.jl:220-238): the vector that becomes the residual again is
    rescaled through the ordinary entry points, which settle the slab first"""
    nx, ny, steps = 36, 30, 14
    A = ko.convection_diffusion_2d(nx, ny)
    x0 = np.random.default_rng(8).random(nx * ny)
    it = kk.ArnoldiIterator(kk.SparseOperator(A, sctx), x0, kk.ModifiedGramSchmidt2(), capacity=steps + 3)
    f = kk.initialize(it)
    oit = ko.ArnoldiIterator(A, x0.copy(), ko.MGS2)
    of = ko.arnoldi_initialize(oit)
    for _ in range(steps):
        f = kk.expand_(it, f)
        of = ko.arnoldi_expand(oit, of)
    assert np.max(np.abs(np.asarray(f.H) - np.asarray(of.H))) < 1e-10 * np.max(np.abs(of.H))
    r = f.r.get()
    assert abs(np.linalg.norm(r) - f.normres) < 1e-13 * f.normres
    assert np.max(np.abs(r - of.r)) < 1e-10 * np.linalg.norm(of.r)
    f = kk.shrink_(f, 9)
    of = ko.arnoldi_shrink(of, 9)
    assert abs(f.normres - of.normres) < 1e-10 * of.normres
    assert np.max(np.abs(f.r.get() - of.r)) < 1e-10 * np.linalg.norm(of.r)
    for _ in range(3):
        f = kk.expand_(it, f)
        of = ko.arnoldi_expand(oit, of)
    assert np.max(np.abs(np.asarray(f.H) - np.asarray(of.H))) < 1e-10 * np.max(np.abs(of.H))


@pytest.mark.parametrize("alg", ["mgs", "mgs2"])
def test_orthonormalize_writes_the_normalised_vector_at_the_commit(kk, ko, sctx, alg):
    rng = np.random.default_rng(11)
    n, m = 6000, 10
    Q, _ = np.linalg.qr(rng.standard_normal((n, m)))
    w = rng.standard_normal(n)
    B = kk.DeviceBasis(n, m + 2, sctx)
    for j in range(m):
        B.upload(j, Q[:, j])
    B.length = m
    dev = {"mgs": kk.ModifiedGramSchmidt(), "mgs2": kk.ModifiedGramSchmidt2()}[alg]
    ref = {"mgs": ko.MGS, "mgs2": ko.MGS2}[alg]
    out = {}
    for fold in (1, 0):
        sctx.set_option("fold_scale", fold)
        sctx.prof_reset(); sctx.prof_enable(1)
        x, beta, _ = B.orthonormalize(B[m].set(w), dev)
        sctx.prof_enable(0)
        out[fold] = (x.copy(), beta, B[m].get().copy(), sctx.prof_get("k_scal")[1])
    sctx.set_option("fold_scale", 1)
    assert out[1][3] == 0 and out[0][3] == 1                      # no scale launch with the fold
    assert np.array_equal(out[1][2], out[0][2]) and out[1][1] == out[0][1]
    wo, xo = ko.orthogonalize(w.copy(), [Q[:, j].copy() for j in range(m)], ref)
    np.testing.assert_allclose(out[1][0], xo, rtol=0, atol=1e-12 * np.linalg.norm(w))
    np.testing.assert_allclose(out[1][2], wo / np.linalg.norm(wo), rtol=0, atol=1e-12)
