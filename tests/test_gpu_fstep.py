"""The whole Lanczos expand! of a short vector in ONE launch (csrc/kk_kernels_fstep.hip): apply, the grid reduction of
[alpha0 | V'w | V'v], the (low-sync) solve, the update, the norm and the normalised commit in one kernel, the scalars delivered through a
pinned slot the host polls, the next step's launch enqueued before the host looks.  Reference recurrences: src/factorizations/lanczos.jl
:297-322 (ClassicalGramSchmidt2) and :325-336 (ModifiedGramSchmidt2), expand! :250-291; the oracle restates them (oracle/krylov_oracle.py).
Not bitwise equal to the projection pair (another summation order of the inner products): equal to rounding, 1e-10 against the oracle."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def relerr(a, b):
    a, b = np.asarray(a, float), np.asarray(b, float)
    return float(np.max(np.abs(a - b) / np.maximum(np.abs(b), 1e-300)))


def run(kk, c, A, x0, dev, steps, fused, lookahead=1, sym=True):
    c.set_option("fused_step", fused); c.set_option("lookahead", lookahead)
    l0 = c.get_option("fstep_launches")
    it = kk.LanczosIterator(kk.SparseOperator(A, c, symmetric=sym), x0, dev, capacity=steps + 3)
    f = kk.initialize(it)
    for _ in range(steps):
        f = kk.expand_(it, f)
    return f, int(c.get_option("fstep_launches") - l0)


@pytest.mark.parametrize("shape", [(36, 30), (131, 77), (512, 200), (498, 500)])
@pytest.mark.parametrize("orth", ["mgs2", "cgs2"])
def test_one_launch_step_against_oracle_and_projection_pair(kk, ko, shape, orth):
    """1 080 .. 250 000 rows (1, 2, 4 and 8 row pairs per thread), odd and even row counts: alpha / beta within 1e-10 of the oracle and 1e-12 of the
    projection pair, V orthonormal, one launch per step (+ the one enqueued ahead), lookahead on and off bit-identical (same kernel either way)"""
    nx, ny = shape
    n = nx * ny
    A = ko.laplacian_2d(nx, ny, shift_diag=10 * np.linspace(0, 1, n) ** 2)
    x0 = np.random.default_rng(3).random(n)
    dev, ref = (kk.ModifiedGramSchmidt2(), ko.MGS2) if orth == "mgs2" else (kk.ClassicalGramSchmidt2(), ko.CGS2)
    steps = 24
    c = kk.Context(0)
    try:
        assert c.get_option("fused_step") == 1
        if n > c.get_option("fused_step_max_rows"):       # (the default route up to 131 072 rows; the kernel itself holds up to 262 144)
            c.set_option("fused_step_max_rows", 249500)
        c.set_option("fused_step_m_limit", -1)            # (by default long bases switch to the projection pair: test_switch_at_the_basis_length_limit)
        f1, n1 = run(kk, c, A, x0, dev, steps, 1, 1)
        f2, n2 = run(kk, c, A, x0, dev, steps, 1, 0)
        f0, n0 = run(kk, c, A, x0, dev, steps, 0, 1)
        assert n0 == 0 and n2 == steps and steps <= n1 <= steps + 1, (n0, n1, n2)
        assert np.array_equal(f1.alphas, f2.alphas) and np.array_equal(f1.betas, f2.betas)
        assert relerr(f1.alphas, f0.alphas) < 1e-12 and relerr(f1.betas, f0.betas) < 1e-12
        oit = ko.LanczosIterator(A, x0.copy(), ref); of = ko.lanczos_initialize(oit)
        for _ in range(steps):
            of = ko.lanczos_expand(oit, of)
        assert relerr(f1.alphas, of.alphas) < 1e-10 and relerr(f1.betas, of.betas) < 1e-10
        V = f1.V.to_numpy()
        assert np.max(np.abs(V.T @ V - np.eye(V.shape[1]))) < 1e-12
        r = f1.r.get()
        assert abs(np.linalg.norm(r) - f1.normres) < 1e-12 * f1.normres and np.max(np.abs(V.T @ r)) < 1e-11 * f1.normres
        assert c.get_option("fstep_failures") == 0
    finally:
        c.close()


@pytest.mark.parametrize("shape", [(36, 30), (300, 200)])
@pytest.mark.parametrize("orth", ["cgs", "mgs", "cgs2", "mgs2"])
def test_one_launch_arnoldi_step(kk, ko, shape, orth):
    """arnoldirecurrence!! + orthogonalize!! (src/factorizations/arnoldi.jl:199-245, orthonormal.jl:378-452) in one launch: w = A v, one (CGS, MGS) or
    two (CGS2, MGS2) passes -- MGS family in the library's low-synchronisation form --, norm, normalised commit; H and beta through the pinned slot.
    H within 1e-10 of the oracle and 1e-12 of the ordinary route, V orthonormal, lookahead on / off bit-identical"""
    nx, ny = shape
    n = nx * ny
    A = ko.convection_diffusion_2d(nx, ny)
    x0 = np.random.default_rng(3).random(n)
    dev, ref = {"cgs": (kk.ClassicalGramSchmidt(), ko.CGS), "mgs": (kk.ModifiedGramSchmidt(), ko.MGS), "cgs2": (kk.ClassicalGramSchmidt2(), ko.CGS2),
                "mgs2": (kk.ModifiedGramSchmidt2(), ko.MGS2)}[orth]
    steps = 20
    c = kk.Context(0)
    try:
        def arnoldi(fused, la):
            c.set_option("fused_step", fused); c.set_option("lookahead", la)
            l0 = c.get_option("fstep_launches")
            it = kk.ArnoldiIterator(kk.SparseOperator(A, c), x0, dev, capacity=steps + 3)
            f = kk.initialize(it)
            for _ in range(steps):
                f = kk.expand_(it, f)
            return f, int(c.get_option("fstep_launches") - l0)
        f1, n1 = arnoldi(1, 1)
        f2, n2 = arnoldi(1, 0)
        f0, n0 = arnoldi(0, 1)
        assert n0 == 0 and n2 == steps and steps <= n1 <= steps + 1, (n0, n1, n2)
        H1, H2, H0 = (np.asarray(f.H, float) for f in (f1, f2, f0))
        assert np.array_equal(H1, H2) and f1.normres == f2.normres
        tol_pair = 1e-12 if dev.is_reorth else 1e-8          # (one-pass orthogonalisers: the two routes differ by their loss of orthogonality)
        assert np.max(np.abs(H1 - H0)) < tol_pair * np.max(np.abs(H0))
        oit = ko.ArnoldiIterator(A, x0.copy(), ref); of = ko.arnoldi_initialize(oit)
        for _ in range(steps):
            of = ko.arnoldi_expand(oit, of)
        Ho = np.asarray(of.H, float)
        tol = 1e-10 if dev.is_reorth else 1e-6
        assert np.max(np.abs(H1 - Ho)) < tol * np.max(np.abs(Ho)) and abs(f1.normres - of.normres) < tol * abs(of.normres)
        if dev.is_reorth:
            V = f1.V.to_numpy()
            assert np.max(np.abs(V.T @ V - np.eye(V.shape[1]))) < 1e-12
        assert c.get_option("fstep_failures") == 0
    finally:
        c.close()


def test_gmres_on_the_one_launch_step(kk, ko):
    """linsolve(GMRES) with restarts (linsolve/gmres.jl:55-140) on a short vector: every Arnoldi step one launch; iteration / operation counts of the
    oracle, the same solution"""
    nx, ny = 60, 50
    n = nx * ny
    A = ko.convection_diffusion_2d(nx, ny)
    b = np.random.default_rng(4).random(n)
    c = kk.Context(0)
    try:
        l0 = c.get_option("fstep_launches")
        tol = 1e-10 * np.linalg.norm(b)
        x, info = kk.linsolve(kk.SparseOperator(A, c), b, None, kk.GMRES(kk.ModifiedGramSchmidt2(), 40, 20, tol))
        xo, oinfo = ko.gmres(A, b, None, krylovdim=20, maxiter=40, tol=tol, orth=ko.MGS2)
        assert c.get_option("fstep_launches") - l0 > 20
        assert info.converged == 1 and (info.numiter, info.numops) == (oinfo.numiter, oinfo.numops)
        assert np.linalg.norm(x - xo) < 1e-8 * np.linalg.norm(xo) and np.linalg.norm(A @ x - b) <= 1.01 * tol
    finally:
        c.close()


def test_general_sparsity_and_nonsymmetric_values(kk, ko, monkeypatch):
    """any operator in the ELL format: a random symmetric pattern (no stencil structure), 20 entries per row"""
    import scipy.sparse as sp
    monkeypatch.setenv("KK_SPMV_FORMAT", "ell")
    n = 30_000
    rng = np.random.default_rng(8)
    R = sp.random(n, n, density=10 / n, random_state=rng, format="csr")
    A = (R + R.T + sp.diags(np.linspace(1, 30, n))).tocsr()
    x0 = rng.random(n)
    c = kk.Context(0)
    try:
        op = kk.SparseOperator(A, c, symmetric=True)
        if not op.info()["format"].startswith("ELL"):
            pytest.skip("operator did not get the ELL format")
        f1, n1 = run(kk, c, A, x0, kk.ModifiedGramSchmidt2(), 20, 1)
        assert n1 >= 20
        oit = ko.LanczosIterator(A, x0.copy(), ko.MGS2); of = ko.lanczos_initialize(oit)
        for _ in range(20):
            of = ko.lanczos_expand(oit, of)
        assert relerr(f1.alphas, of.alphas) < 1e-10 and relerr(f1.betas, of.betas) < 1e-10
    finally:
        c.close()


def test_eigsolve_with_restarts_runs_on_the_one_launch_step(kk, ko):
    """thick restarts (eigsolve/lanczos.jl:33-120): shrink!, basistransform!, the restart's scale!!(r, 1 / beta) between cycles of one-launch
    steps -- numiter / numops and the eigenvalues of the oracle"""
    nx, ny = 60, 50
    n = nx * ny
    A = ko.laplacian_2d(nx, ny)
    x0 = np.random.default_rng(5).random(n)
    c = kk.Context(0)
    try:
        l0 = c.get_option("fstep_launches")
        vals, vecs, info = kk.eigsolve(kk.SparseOperator(A, c, symmetric=True), x0, 3, "SR", krylovdim=20, maxiter=60, tol=1e-10, orth=kk.ModifiedGramSchmidt2())
        ovals, ovecs, oinfo = ko.eigsolve_lanczos(A, x0.copy(), 3, "SR", krylovdim=20, maxiter=60, tol=1e-10, orth=ko.MGS2)
        assert c.get_option("fstep_launches") - l0 > 20
        assert info.converged >= 3 and (info.numiter, info.numops) == (oinfo.numiter, oinfo.numops)
        np.testing.assert_allclose(vals[:3], ovals[:3], rtol=1e-10)
    finally:
        c.close()


def test_launch_that_gives_up_is_repeated_on_the_ordinary_route(kk, ko):
    """test hook "fstep_fault": a launch whose block 0 leaves at once never stores its token -- the host notices, switches the route off for
    the context ("fstep_failures", "fused_step" -> 0) and repeats the step on the projection pair; same factorization"""
    nx, ny = 70, 64
    n = nx * ny
    A = ko.laplacian_2d(nx, ny, shift_diag=10 * np.linspace(0, 1, n) ** 2)
    x0 = np.random.default_rng(3).random(n)
    c = kk.Context(0)
    try:
        c.set_option("persist_timeout_ms", 20)
        it = kk.LanczosIterator(kk.SparseOperator(A, c, symmetric=True), x0, kk.ModifiedGramSchmidt2(), capacity=30)
        f = kk.initialize(it)
        for i in range(20):
            if i == 7:
                c.set_option("fstep_fault", 1)
            f = kk.expand_(it, f)
        assert c.get_option("fstep_failures") == 1 and c.get_option("fused_step") == 0
        oit = ko.LanczosIterator(A, x0.copy(), ko.MGS2); of = ko.lanczos_initialize(oit)
        for _ in range(20):
            of = ko.lanczos_expand(oit, of)
        assert relerr(f.alphas, of.alphas) < 1e-10 and relerr(f.betas, of.betas) < 1e-10
        V = f.V.to_numpy()
        assert np.max(np.abs(V.T @ V - np.eye(V.shape[1]))) < 1e-12
    finally:
        c.close()


def test_switch_at_the_basis_length_limit(kk, ko):
    """by default the one launch serves the SHORT bases (m <= 96 - 64 n / 1e5, at least 16) and the projection pair the steps beyond: the
    switch happens inside a factorization, on the same slab state (Gram rows, normalised residual column), with no repeated step"""
    nx, ny = 250, 240            # 60 000 rows: limit 57.6 -> steps with m = k + 1 <= 57 take the one launch
    n = nx * ny
    A = ko.laplacian_2d(nx, ny, shift_diag=10 * np.linspace(0, 1, n) ** 2)
    x0 = np.random.default_rng(3).random(n)
    steps = 70
    c = kk.Context(0)
    try:
        for dev, ref in ((kk.ModifiedGramSchmidt2(), ko.MGS2), (kk.ClassicalGramSchmidt2(), ko.CGS2)):
            c.prof_reset(); c.prof_enable(1)
            f, nl = run(kk, c, A, x0, dev, steps, 1)
            c.prof_enable(0)
            assert 56 <= nl <= 57, nl                                     # k = 1 .. 56 (m = 2 .. 57)
            assert c.prof_get("k_project")[1] == steps - 56, c.prof_get("k_project")   # ... and one projection step each for the rest: none repeated
            oit = ko.LanczosIterator(A, x0.copy(), ref); of = ko.lanczos_initialize(oit)
            for _ in range(steps):
                of = ko.lanczos_expand(oit, of)
            assert relerr(f.alphas, of.alphas) < 1e-10 and relerr(f.betas, of.betas) < 1e-10
            V = f.V.to_numpy()
            assert np.max(np.abs(V.T @ V - np.eye(V.shape[1]))) < 1e-12
    finally:
        c.close()


def test_routes_outside_its_scope_are_unchanged(kk, ko):
    """strict MGS2 (mgs_mode 0), more rows than fused_step_max_rows, a factorization that does not start at column 0: not this route"""
    nx, ny = 70, 64
    n = nx * ny
    A = ko.laplacian_2d(nx, ny)
    x0 = np.random.default_rng(3).random(n)
    c = kk.Context(0)
    try:
        c.set_option("mgs_mode", 0)
        _, n_strict = run(kk, c, A, x0, kk.ModifiedGramSchmidt2(), 6, 1)
        c.set_option("mgs_mode", 2); c.set_option("fused_step_max_rows", 1000)
        _, n_long = run(kk, c, A, x0, kk.ModifiedGramSchmidt2(), 6, 1)
        assert n_strict == 0 and n_long == 0
    finally:
        c.close()
