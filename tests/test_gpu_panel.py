"""The persistent PANEL kernel for modified Gram-Schmidt sweeps on short vectors (csrc/kk_kernels_panel.hip): w resident in
registers, P basis vectors per grid reduction with the exact in-panel triangular correction from on-the-fly Gram entries,
every basis vector read once.  Reference order being restated: src/orthonormal.jl:414-452 (MGS / MGS2 / MGSIR),
src/factorizations/arnoldi.jl:239-245, lanczos.jl:325-338.  P = 1 (mgs_mode 0) is the strict order itself."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def relerr(a, b):
    a, b = np.asarray(a, float), np.asarray(b, float)
    return float(np.max(np.abs(a - b) / np.maximum(np.abs(b), 1e-300)))


@pytest.fixture(params=["lag1", "in_panel"])
def pctx(kk, request):
    """outside the strict order the sweeps go through the cross-panel lag-1 kernel (k_mgs_panel_lag: three register-resident panels,
    reductions off the critical path; vectors of <= 2.75 M rows) or -- option panel_lag = 0, and beyond that length -- through
    k_mgs_panel with its in-panel correction only: both against the oracle"""
    c = kk.Context(0)
    if c.get_option("mgs_persist") == 0:
        pytest.skip("no cooperative launch on this device: the persistent routes are off")
    c.set_option("panel_min_rows", 0)     # auto mode takes the panel kernel at every size it can hold
    c.set_option("panel_lag", 1 if request.param == "lag1" else 0)
    yield c
    c.close()


def algs(kk, ko):
    return [(kk.ModifiedGramSchmidt(), ko.MGS), (kk.ModifiedGramSchmidt2(), ko.MGS2), (kk.ModifiedGramSchmidtIR(0.99), ko.MGSIR(0.99))]


# vector lengths chosen to hit every instantiation on 256 CUs (k_mgs_panel, grid-row = 262144 rows: NV = 4 / 8 / 16; k_mgs_panel_lag,
# grid-row = 229376 rows: NV = 4 / 6 / 9 / 12, beyond 2.75 M rows the former)
@pytest.mark.parametrize("n,m", [(777, 5), (900000, 7), (1200000, 5), (1900000, 4), (2500000, 4), (3600000, 3)])
@pytest.mark.parametrize("mode,width", [(0, 0), (2, 0), (2, 3)])
def test_orthogonalize_through_the_panel_kernel(kk, ko, pctx, n, m, mode, width):
    pctx.set_option("mgs_mode", mode)
    pctx.set_option("panel_width", width)
    rng = np.random.default_rng(n + m)
    Q, _ = np.linalg.qr(rng.standard_normal((n, m)))
    w = Q @ rng.standard_normal(m) * 3 + rng.standard_normal(n)
    B = kk.DeviceBasis(n, m + 2, pctx)
    for j in range(m):
        B.upload(j, Q[:, j])
    B.length = m
    cols = [Q[:, j].copy() for j in range(m)]
    for dev, ref in algs(kk, ko):
        pctx.prof_reset(); pctx.prof_enable(1)
        x, nrm, npass = B.orthogonalize(B[m].set(w), dev)
        pctx.prof_enable(0)
        assert pctx.prof_get("k_mgs_panel")[1] >= 1 and pctx.prof_get("k_project")[1] == 0 and pctx.prof_get("k_mgs_persist")[1] == 0, dev.name
        wo, xo = ko.orthogonalize(w.copy(), cols, ref)
        np.testing.assert_allclose(x, xo, rtol=0, atol=1e-12 * np.linalg.norm(w), err_msg=dev.name)
        np.testing.assert_allclose(B[m].get(), wo, rtol=0, atol=1e-12 * np.linalg.norm(w), err_msg=dev.name)
        assert abs(nrm - np.linalg.norm(wo)) < 1e-12 * np.linalg.norm(w)
    B.free()


@pytest.mark.parametrize("width", [1, 2, 3])
def test_non_orthonormal_basis_gets_the_exact_mgs_coefficients(kk, ko, pctx, width):
    """the in-panel correction uses the Gram entries of the vectors as they are: a basis that is far from orthonormal must
    give the sequential MGS coefficients (orthonormal.jl:417-421), not those of a projection"""
    pctx.set_option("mgs_mode", 2)
    pctx.set_option("panel_width", width)
    rng = np.random.default_rng(21)
    n, m = 9000, 7
    X = rng.standard_normal((n, m))
    X[:, 1] += 0.8 * X[:, 0]
    X[:, 2] += 0.5 * X[:, 1] - 0.3 * X[:, 0]
    X[:, 5] += 0.9 * X[:, 4]
    X /= np.linalg.norm(X, axis=0)          # unit vectors, far from orthogonal
    w = rng.standard_normal(n) + X @ rng.standard_normal(m)
    B = kk.DeviceBasis(n, m + 2, pctx)
    for j in range(m):
        B.upload(j, X[:, j])
    B.length = m
    x, nrm, _ = B.orthogonalize(B[m].set(w), kk.ModifiedGramSchmidt())
    wo, xo = ko.orthogonalize(w.copy(), [X[:, j].copy() for j in range(m)], ko.MGS)
    np.testing.assert_allclose(x, xo, rtol=0, atol=1e-11 * np.linalg.norm(w))
    np.testing.assert_allclose(B[m].get(), wo, rtol=0, atol=1e-10 * np.linalg.norm(w))


@pytest.mark.parametrize("width", [0, 1])
def test_lanczos_and_arnoldi_factorizations(kk, ko, pctx, width):
    """per-expand! invariants of test/factorize.jl:140-148 / 185-193 and the oracle's trajectory, sweeps through the panel
    kernel (pending `w -= alpha v` in front, fused norm, normalised commit)"""
    pctx.set_option("mgs_mode", 2)
    pctx.set_option("panel_width", width)
    nx, ny, steps = 50, 44, 26
    n = nx * ny
    A = ko.laplacian_2d(nx, ny, shift_diag=10 * np.linspace(0, 1, n) ** 2)
    x0 = np.random.default_rng(3).random(n)
    pctx.prof_reset(); pctx.prof_enable(1)
    it = kk.LanczosIterator(kk.SparseOperator(A, pctx, symmetric=True), x0, kk.ModifiedGramSchmidt2(), capacity=steps + 3)
    f = kk.initialize(it)
    oit = ko.LanczosIterator(A, x0.copy(), ko.MGS2)
    of = ko.lanczos_initialize(oit)
    for _ in range(steps):
        f = kk.expand_(it, f)
        of = ko.lanczos_expand(oit, of)
    pctx.prof_enable(0)
    assert pctx.prof_get("k_mgs_panel")[1] == steps and pctx.prof_get("k_scal")[1] <= 3   # scale pass folded into the commit
    assert relerr(f.alphas, of.alphas) < 1e-10 and relerr(f.betas, of.betas) < 1e-10
    V = f.V.to_numpy()
    assert np.max(np.abs(V.T @ V - np.eye(V.shape[1]))) < 1e-12
    T = np.diag(f.alphas) + np.diag(f.betas[:-1], 1) + np.diag(f.betas[:-1], -1)
    R = A @ V - V @ T
    R[:, -1] -= f.r.get()
    assert np.max(np.abs(R)) < 1e-11 * abs(A).max()
    # Arnoldi (two sweeps in one launch)
    C = ko.convection_diffusion_2d(40, 36)
    x0 = np.random.default_rng(8).random(C.shape[0])
    ita = kk.ArnoldiIterator(kk.SparseOperator(C, pctx), x0, kk.ModifiedGramSchmidt2(), capacity=20)
    fa = kk.initialize(ita)
    oita = ko.ArnoldiIterator(C, x0.copy(), ko.MGS2)
    ofa = ko.arnoldi_initialize(oita)
    for _ in range(16):
        fa = kk.expand_(ita, fa)
        ofa = ko.arnoldi_expand(oita, ofa)
    assert np.max(np.abs(np.asarray(fa.H) - np.asarray(ofa.H))) < 1e-10 * np.max(np.abs(ofa.H))
    Va = fa.V.to_numpy()
    assert np.max(np.abs(Va.T @ Va - np.eye(Va.shape[1]))) < 1e-12


def test_gmres_counts_equal_the_oracle(kk, ko, pctx):
    """north_star: GMRES residual norm matching iteration count -- linsolve through the panel kernel (auto mode)"""
    pctx.set_option("mgs_mode", 2)
    pctx.set_option("panel_width", 0)
    A = ko.convection_diffusion_2d(60, 40)
    b = np.random.default_rng(4).random(A.shape[0])
    tol = 1e-10 * np.linalg.norm(b)
    pctx.prof_reset(); pctx.prof_enable(1)
    x, info = kk.linsolve(kk.SparseOperator(A, pctx), b, None, kk.GMRES(kk.ModifiedGramSchmidt2(), 20, 25, tol), 0.2, 1.0)
    pctx.prof_enable(0)
    xo, oinfo = ko.gmres(A, b, None, 0.2, 1.0, krylovdim=25, maxiter=20, tol=tol, orth=ko.MGS2)
    assert pctx.prof_get("k_mgs_panel")[1] > 0
    assert (info.converged, info.numiter, info.numops) == (oinfo.converged, oinfo.numiter, oinfo.numops)
    assert abs(info.normres - oinfo.normres) <= 1e-3 * tol  # final explicit residual: cancellation-level noise
    assert np.linalg.norm(0.2 * x + A @ x - b) <= 1.01 * tol


def test_panel_kernel_recovers_from_a_barrier_timeout(kk, ko, pctx):
    pctx.set_option("mgs_mode", 2)
    pctx.set_option("panel_width", 0)
    rng = np.random.default_rng(5)
    n, m = 5000, 12
    Q, _ = np.linalg.qr(rng.standard_normal((n, m)))
    w = rng.standard_normal(n)
    B = kk.DeviceBasis(n, m + 2, pctx)
    for j in range(m):
        B.upload(j, Q[:, j])
    B.length = m
    pctx.set_option("persist_fault", 1)
    x, nrm, _ = B.orthogonalize(B[m].set(w), kk.ModifiedGramSchmidt2())
    assert pctx.get_option("persist_timeouts") == 1
    wo, xo = ko.orthogonalize(w.copy(), [Q[:, j].copy() for j in range(m)], ko.MGS2)
    np.testing.assert_allclose(x, xo, rtol=0, atol=1e-12 * np.linalg.norm(w))
    np.testing.assert_allclose(B[m].get(), wo, rtol=0, atol=1e-12 * np.linalg.norm(w))


def test_bitwise_reproducible_and_identical_on_every_block(kk, pctx):
    """the grid reduction sums in a fixed order: two runs give the same bits"""
    pctx.set_option("mgs_mode", 2)
    pctx.set_option("panel_width", 0)
    rng = np.random.default_rng(9)
    n, m = 700000, 9
    Q, _ = np.linalg.qr(rng.standard_normal((n, m)))
    w = rng.standard_normal(n)
    B = kk.DeviceBasis(n, m + 2, pctx)
    for j in range(m):
        B.upload(j, Q[:, j])
    B.length = m
    outs = []
    for _ in range(3):
        x, nrm, _ = B.orthogonalize(B[m].set(w), kk.ModifiedGramSchmidt2())
        outs.append((x.copy(), nrm, B[m].get().copy()))
    for o in outs[1:]:
        assert np.array_equal(o[0], outs[0][0]) and o[1] == outs[0][1] and np.array_equal(o[2], outs[0][2])


@pytest.mark.parametrize("shape", [(2000, 160), (1000, 700), (130, 2400), (4000, 500)])
@pytest.mark.parametrize("orth_name", ["mgs", "mgs2"])
def test_panel_kernel_applies_the_stencil_itself(kk, ko, shape, orth_name):
    """k_mgs_panel<.., APPLY> (round 6): the run-ahead of an Arnoldi step on a value-free 5-point stencil (even line length: BASELINE config 3's operator)
    forms w = A v inside the sweep launch -- no separate apply, no store and re-load of w.  Same products in the same order as k_spmv_dia: every entry of
    H and every basis vector BIT-identical to panel_apply = 0; the apply launches disappear from the steps that ran ahead (arnoldi.jl:242,
    orthonormal.jl:414-439)"""
    nx, ny = shape
    n = nx * ny
    A = ko.convection_diffusion_2d(nx, ny)
    x0 = np.random.default_rng(3).random(n)
    dev, ref = (kk.ModifiedGramSchmidt(), ko.MGS) if orth_name == "mgs" else (kk.ModifiedGramSchmidt2(), ko.MGS2)
    steps = 14
    c = kk.Context(0)
    try:
        if c.get_option("mgs_persist") == 0:
            pytest.skip("persistent routes off on this device")
        c.set_option("panel_min_rows", 0); c.set_option("fused_step", 0)
        out = {}
        for pa in (1, 0):
            c.set_option("panel_apply", pa)
            l0 = c.get_option("panel_apply_launches")
            c.prof_reset(); c.prof_enable(1)
            it = kk.ArnoldiIterator(kk.SparseOperator(A, c), x0, dev, capacity=steps + 3)
            f = kk.initialize(it)
            for _ in range(steps):
                f = kk.expand_(it, f)
            c.prof_enable(0)
            out[pa] = (np.asarray(f.H, float).copy(), f.normres, f.V.to_numpy().copy(), int(c.get_option("panel_apply_launches") - l0), c.prof_get("k_spmv_dia")[1],
                       c.prof_get("k_mgs_panel")[1])
        c.set_option("panel_apply", 1)
        assert out[1][3] >= steps - 2 and out[0][3] == 0, (out[1][3], out[0][3])         # every step that ran ahead applied the operator inside its sweep launch
        assert out[1][4] <= out[0][4] - (steps - 2), (out[1][4], out[0][4])               # ... and those apply launches are gone
        assert out[1][5] == out[0][5]
        assert np.array_equal(out[1][0], out[0][0]) and out[1][1] == out[0][1] and np.array_equal(out[1][2], out[0][2])
        oit = ko.ArnoldiIterator(A, x0.copy(), ref); of = ko.arnoldi_initialize(oit)
        for _ in range(steps):
            of = ko.arnoldi_expand(oit, of)
        Ho = np.asarray(of.H, float)
        tol = 1e-10 if orth_name == "mgs2" else 1e-7
        assert np.max(np.abs(out[1][0] - Ho)) < tol * np.max(np.abs(Ho))
        assert c.get_option("persist_timeouts") == 0
    finally:
        c.close()


def test_lost_launch_that_was_to_apply_the_stencil_itself_is_repeated_with_its_apply(kk, ko):
    """a run-ahead launch of k_mgs_panel<.., APPLY> that gives up (test hook) has written nothing -- not even w = A v, which no other launch formed:
    the recovery has to repeat the apply as well as the sweeps (kk_arnoldi_expand, `la_inside`)"""
    nx, ny = 600, 500
    A = ko.convection_diffusion_2d(nx, ny)
    x0 = np.random.default_rng(8).random(nx * ny)
    c = kk.Context(0)
    try:
        if c.get_option("mgs_persist") == 0:
            pytest.skip("persistent routes off on this device")
        c.set_option("panel_min_rows", 0); c.set_option("fused_step", 0); c.set_option("persist_timeout_ms", 20)
        it = kk.ArnoldiIterator(kk.SparseOperator(A, c), x0, kk.ModifiedGramSchmidt2(), capacity=22)
        f = kk.initialize(it)
        oit = ko.ArnoldiIterator(A, x0.copy(), ko.MGS2); of = ko.arnoldi_initialize(oit)
        for i in range(16):
            if i in (5, 11):
                c.set_option("persist_fault", 1)     # the NEXT persistent launch -- the one enqueued ahead during this call -- gives up
            f = kk.expand_(it, f); of = ko.arnoldi_expand(oit, of)
        assert c.get_option("persist_timeouts") == 2 and c.get_option("panel_apply_launches") >= 8
        assert np.max(np.abs(np.asarray(f.H) - np.asarray(of.H))) < 1e-10 * np.max(np.abs(of.H))
        V = f.V.to_numpy()
        assert np.max(np.abs(V.T @ V - np.eye(V.shape[1]))) < 1e-12
    finally:
        c.close()
