"""A grid-barrier timeout of the persistent strict-MGS kernel (GPU shared with another job) is recovered INSIDE the
library: no block writes the work vector back, the sweep is repeated on the launch-per-vector route, the speculative
next-step apply that consumed the failed launch's norm is dropped, and the caller sees an ordinary result.  The option
"persist_fault" makes the next N persistent launches behave like a timed-out one (block 0 raises the flag and leaves).
Reference order: src/orthonormal.jl:414-439, src/factorizations/lanczos.jl:325-338."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def relerr(a, b):
    a, b = np.asarray(a, float), np.asarray(b, float)
    return float(np.max(np.abs(a - b) / np.maximum(np.abs(b), 1e-300)))


@pytest.fixture()
def sctx(kk):
    c = kk.Context(0)
    c.set_option("mgs_mode", 0)
    c.set_option("mgs_panel", 0)         # these tests are about k_mgs_persist (the panel kernel has its own: tests/test_gpu_panel.py)
    c.set_option("lookahead", 0)         # call-by-call route: the launch counts below are those of one sweep per call (run-ahead: tests/test_gpu_lookahead.py)
    if c.get_option("mgs_persist") == 0:
        pytest.skip("no cooperative launch on this device: the persistent route is off anyway")
    yield c
    c.close()


def test_lanczos_expand_recovers_from_a_barrier_timeout(kk, ko, sctx):
    nx, ny, steps = 48, 40, 24
    n = nx * ny
    A = ko.laplacian_2d(nx, ny, shift_diag=10 * np.linspace(0, 1, n) ** 2)
    x0 = np.random.default_rng(3).random(n)
    op = kk.SparseOperator(A, sctx, symmetric=True)
    it = kk.LanczosIterator(op, x0, kk.ModifiedGramSchmidt2(), capacity=steps + 3)
    f = kk.initialize(it)
    oit = ko.LanczosIterator(A, x0.copy(), ko.MGS2)
    of = ko.lanczos_initialize(oit)
    for i in range(steps):
        if i == 7:
            sctx.set_option("persist_fault", 1)
        f = kk.expand_(it, f)
        of = ko.lanczos_expand(oit, of)
        if i == 6:
            sctx.prof_reset(); sctx.prof_enable(1)
    sctx.prof_enable(0)
    # one failed launch, its repeat + the next 3 sweeps on the launch-per-vector route, then the persistent route again
    # (the route is suspended, not switched off: an auto-mode context keeps the reference's strict order)
    assert sctx.get_option("persist_timeouts") == 1 and sctx.get_option("mgs_persist") == 1 and sctx.get_option("persist_skip") == 0
    assert sctx.prof_get("k_mgs_persist")[1] == 1 + (steps - 1 - 7 - 3) and sctx.prof_get("k_mgs_step")[1] > 0
    assert relerr(f.alphas, of.alphas) < 1e-10 and relerr(f.betas, of.betas) < 1e-10
    V = f.V.to_numpy()
    assert np.max(np.abs(V.T @ V - np.eye(V.shape[1]))) < 1e-12


@pytest.mark.parametrize("alg", ["mgs", "mgs2", "mgsir"])
def test_orthogonalize_recovers_from_a_barrier_timeout(kk, ko, sctx, alg):
    rng = np.random.default_rng(5)
    n, m = 5000, 12
    Q, _ = np.linalg.qr(rng.standard_normal((n, m)))
    w = rng.standard_normal(n)
    B = kk.DeviceBasis(n, m + 2, sctx)
    for j in range(m):
        B.upload(j, Q[:, j])
    B.length = m
    dev = {"mgs": kk.ModifiedGramSchmidt(), "mgs2": kk.ModifiedGramSchmidt2(), "mgsir": kk.ModifiedGramSchmidtIR(0.99)}[alg]
    ref = {"mgs": ko.MGS, "mgs2": ko.MGS2, "mgsir": ko.MGSIR(0.99)}[alg]
    sctx.set_option("persist_fault", 1)
    x, nrm, _ = B.orthogonalize(B[m].set(w), dev)
    assert sctx.get_option("persist_timeouts") == 1
    wo, xo = ko.orthogonalize(w.copy(), [Q[:, j].copy() for j in range(m)], ref)
    np.testing.assert_allclose(x, xo, rtol=0, atol=1e-12 * np.linalg.norm(w))
    np.testing.assert_allclose(B[m].get(), wo, rtol=0, atol=1e-12 * np.linalg.norm(w))
    assert abs(nrm - np.linalg.norm(wo)) < 1e-12 * np.linalg.norm(w)


def test_arnoldi_expand_recovers_and_respeculates(kk, ko, sctx):
    """the Arnoldi step requests its speculative next apply through final_sync: after the recovery it must be re-issued
    with the norm of the repeated sweep (same H as a run on the launch-per-vector route throughout)"""
    nx, ny, steps = 40, 32, 16
    A = ko.convection_diffusion_2d(nx, ny)
    x0 = np.random.default_rng(8).random(nx * ny)
    runs = []
    for fault_at in (None, 5):
        c = sctx if fault_at is not None else None
        ctx = kk.Context(0) if c is None else c
        ctx.set_option("mgs_mode", 0)
        if fault_at is None:
            ctx.set_option("mgs_persist", 0)          # reference run: the launch-per-vector route throughout
        it = kk.ArnoldiIterator(kk.SparseOperator(A, ctx), x0, kk.ModifiedGramSchmidt2(), capacity=steps + 3)
        f = kk.initialize(it)
        for i in range(steps):
            if i == fault_at:
                ctx.set_option("persist_fault", 1)
            f = kk.expand_(it, f)
        runs.append((np.array(f.H, dtype=float).copy(), f.normres))
        if c is None:
            ctx.close()
    assert sctx.get_option("persist_timeouts") == 1
    oit = ko.ArnoldiIterator(A, x0.copy(), ko.MGS2)
    of = ko.arnoldi_initialize(oit)
    for _ in range(steps):
        of = ko.arnoldi_expand(oit, of)
    assert len(runs[1][0]) == len(of.H) == len(runs[0][0])
    assert np.max(np.abs(runs[0][0] - runs[1][0])) < 1e-12 * np.max(np.abs(of.H))   # recovered run vs launch-per-vector run (summation orders differ)
    assert np.max(np.abs(runs[1][0] - np.asarray(of.H))) < 1e-10 * np.max(np.abs(of.H))
    assert abs(runs[1][1] - of.normres) < 1e-10 * abs(of.normres)
