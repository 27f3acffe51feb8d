"""BASELINE.json's FULL sizes on the GPU, checked through size-independent properties (the
oracle cannot run 10M rows in test time): orthonormality of the basis (device Gram panels),
the Krylov relation A V = V T + r e' on sampled columns, residual identities of converged
Ritz pairs, GMRES true-residual identity, bitwise run-to-run reproducibility."""
import sys
from pathlib import Path

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))


def _gram_offdiag_max(kk, V, k):
    """max |V'V - I| over the first k columns, via the MFMA Gram panel (16 columns at a time)."""
    worst = 0.0
    for j0 in range(0, k, 16):
        q = min(16, k - j0)
        M = kk.block_inner(kk.Block(V, 0, k), kk.Block(V, j0, q))
        M[j0:j0 + q, :] -= np.eye(q)
        worst = max(worst, float(np.max(np.abs(M))))
    return worst


@pytest.mark.parametrize("orth_name", ["mgs2", "cgs2"])
def test_config2_lanczos_10M_properties(kk, ctx, orth_name):
    from bench import laplacian_rows, NX, NY
    N, K = NX * NY, 100
    A = laplacian_rows(NX, NY, 0, NY)
    op = kk.SparseOperator(A, ctx, symmetric=True, via_csc=True)
    assert op.info()["format"] == "ELL+DIA const" and op.info()["nnz"] == 5 * N - 2 * (NX + NY)   # grid stencil: diagonals next to ELL
    V = kk.DeviceBasis(N, K + 4, ctx)
    x0 = kk.DeviceBasis(N, 1, ctx)
    x0[0].rand_(3)
    it = kk.LanczosIterator(op, x0[0], kk.Orthogonalizer(orth_name), capacity=K + 2)
    runs = []
    for rep in range(2):
        f = kk.initialize(it, V)
        for _ in range(K - 1):
            f = kk.expand_(it, f)
        runs.append((list(f.alphas), list(f.betas)))
    assert runs[0] == runs[1]  # deterministic reductions + speculation: bitwise reproducible
    al, be = np.array(runs[0][0]), np.array(runs[0][1])
    assert np.all(be > 0) and np.all(np.abs(al - 4.0) < 4.0)  # Gershgorin: spectrum in (0, 8)
    assert _gram_offdiag_max(kk, V, K) < 1e-12
    # A v_j = beta_{j-1} v_{j-1} + alpha_j v_j + beta_j v_{j+1}  (v_{K+1} = r/beta_K) on sampled columns
    W = kk.DeviceBasis(N, 1, ctx)
    for j in (1, 37, K - 2, K - 1):
        op.apply(V[j], W[0])
        W[0].add_(V[j], -al[j]).add_(V[j - 1], -be[j - 1])
        if j < K - 1:
            W[0].add_(V[j + 1], -be[j])
        else:
            W[0].add_(f.r, -1.0)
        assert W[0].norm() < 1e-11, j
    # Ritz values of the 100x100 tridiagonal lie inside the closed-form spectrum bounds
    theta = np.linalg.eigvalsh(np.diag(al) + np.diag(be[:-1], 1) + np.diag(be[:-1], -1))
    lam_max = 4 + 2 * np.cos(np.pi / (NX + 1)) + 2 * np.cos(np.pi / (NY + 1))
    lam_min = 4 - 2 * np.cos(np.pi / (NX + 1)) - 2 * np.cos(np.pi / (NY + 1))
    assert lam_min - 1e-10 <= theta[0] and theta[-1] <= lam_max + 1e-10


def test_config2b_eigsolve_10M_converges_with_restarts(kk, ctx):
    """Variant 2b: + diag(10 linspace(0,1,N)^2); converged pairs satisfy |A v - lambda v| <= tol."""
    import scipy.sparse as sp
    from bench import laplacian_rows, NX, NY
    N = NX * NY
    A = (laplacian_rows(NX, NY, 0, NY) + sp.diags(10 * np.linspace(0, 1, N) ** 2)).tocsr()
    op = kk.SparseOperator(A, ctx, symmetric=True)
    x0 = np.random.default_rng(3).random(N)
    tol = 1e-6
    n0 = ctx.get_option("norm_commits_consumed")
    ctx.prof_reset(); ctx.prof_enable(1)
    vals, out, info = kk.eigsolve(op, x0, 1, "LM", kk.Lanczos(krylovdim=100, tol=tol, maxiter=60), return_device=True)
    ctx.prof_enable(0)
    assert info.converged >= 1, info
    # The thick restart's B[keep+1] = scale!!(r, 1 / beta) (eigsolve/lanczos.jl:111) meets a residual the persistent kernel left
    # normalised: the commit is CONSUMED (the stored bits are the reference's r * (1 / beta)) -- no pass that multiplies beta back
    # in and none that divides it out again.  What remains per restart is the reference's own pair: shrink!'s scale!!(r, beta)
    # (lanczos.jl:289) and the next expand!'s scale!!(r, 1 / beta) (:257).  VERDICT r4, next 3.
    restarts = info.numiter - 1
    assert restarts >= 1
    assert ctx.get_option("norm_commits_consumed") - n0 == restarts
    assert ctx.prof_get("k_scal")[1] <= 3 + 2 * restarts, (ctx.prof_get("k_scal"), restarts)
    assert ctx.prof_get("k_mgs_persist")[1] > 0
    W = kk.DeviceBasis(N, 1, ctx)
    op.apply(out[0], W[0])
    W[0].add_(out[0], -vals[0])
    # residual identity: the true residual equals the Lanczos estimate |f_1| (eigsolve/lanczos.jl:61-67)
    res = W[0].norm()
    assert res <= 2 * tol and abs(res - info.normres[0]) <= 1e-3 * tol + 1e-2 * res
    assert abs(out[0].norm() - 1) < 1e-12 and 17.0 < vals[0] < 18.0


def test_config3_gmres_2M_true_residual(kk, ctx):
    from tools.bench_configs import convdiff
    nx, ny = 2000, 1000
    N = nx * ny
    A = convdiff(nx, ny)
    op = kk.SparseOperator(A, ctx)
    b = np.random.default_rng(4).random(N)
    nb = np.linalg.norm(b)
    x, info = kk.linsolve(op, b, None, kk.GMRES(kk.ModifiedGramSchmidt2(), 3, 60, 1e-10 * nb))
    # numops (gmres.jl:37,42,60,123): 1 (r0) + 1 (initialize) + 59 expand! per cycle + 1 explicit residual in the last cycle
    assert info.numiter == 3 and info.numops == 2 + 3 * 59 + 1 and info.converged == 0
    r_true = b - A @ x
    # the reported residual norm is the explicitly recomputed one (gmres.jl:119-124)
    assert abs(np.linalg.norm(r_true) - info.normres) <= 1e-9 * nb
    assert np.linalg.norm(r_true) < nb  # made progress


def test_config5_blocklanczos_10M_properties(kk, ctx):
    from bench import laplacian_rows, NX, NY
    N, bs, K = NX * NY, 16, 100
    A = laplacian_rows(NX, NY, 0, NY)
    op = kk.SparseOperator(A, ctx, symmetric=True)
    S = kk.DeviceBasis(N, K + 3 * bs, ctx)
    it = kk.BlockLanczosIterator(op, [None] * bs, K + bs)
    # start block generated on the device (no 1.3 GB host upload)
    area_b = it.maxdim + bs
    for j in range(bs):
        S[area_b + j].rand_(100 + j)
    it.x0 = [S[area_b + j] for j in range(bs)]
    f = it.initialize(S)
    while len(f) < K:
        f = it.expand(f)
    k = len(f)
    assert k == 7 * bs and f.R_size == bs
    assert _gram_offdiag_max(kk, S, k) < 1e-12
    R = f.residual()
    P = kk.block_inner(kk.Block(S, 0, k), R)
    assert np.max(np.abs(P)) < 1e-10
    H = f.H[:k, :k]
    assert np.max(np.abs(H - H.T)) < 1e-12
    # A X_last = V H[:, last] + R  on the last block, column 0: check one column via device ops
    W = kk.DeviceBasis(N, 1, ctx)
    op.apply(S[k - bs], W[0])
    S.length = k
    S.unproject(W[0], H[:, k - bs], 0, k, -1.0, 1.0)
    W[0].add_(R[0], -1.0)
    assert W[0].norm() < 1e-10


def test_config4_gkl_5Mx1M_properties(kk, ctx):
    """svdsolve/GKL on the 5M x 1M random sparse map (nnz/row = 20), single GPU: bidiagonal relations
    on sampled columns, orthonormality of both bases via device Gram panels, sigma_max bound."""
    import scipy.sparse as sp
    m, n, per, K = 5_000_000, 1_000_000, 20, 30
    rng = np.random.default_rng(5)
    cols = rng.integers(0, n, size=m * per, dtype=np.int32)
    vals = rng.standard_normal(m * per)
    A = sp.csr_matrix((vals, cols, np.arange(0, m * per + 1, per, dtype=np.int64)), shape=(m, n))
    A.sum_duplicates()
    op = kk.SparseOperator(A, ctx)
    it = kk.GKLIterator(op, rng.random(m), kk.ModifiedGramSchmidt2(), capacity=K + 2)
    f = kk.initialize(it)
    for _ in range(K - 1):
        f = kk.expand_(it, f)
    al, be = np.array(f.alphas), np.array(f.betas)
    assert np.all(al > 0) and np.all(be > 0)
    assert _gram_offdiag_max(kk, f.U, K) < 1e-12 and _gram_offdiag_max(kk, f.V, K) < 1e-12
    WU, WV = kk.DeviceBasis(m, 1, ctx), kk.DeviceBasis(n, 1, ctx)
    for j in (0, 11, K - 1):
        # A v_j = alpha_j u_j + beta_j u_{j+1}   (u_{K+1} = r / beta_K)
        op.apply(f.V[j], WU[0])
        WU[0].add_(f.U[j], -al[j])
        if j < K - 1:
            WU[0].add_(f.U[j + 1], -be[j])
        else:
            WU[0].add_(f.r, -1.0)
        assert WU[0].norm() < 1e-10, j
        # A' u_j = alpha_j v_j + beta_{j-1} v_{j-1}
        op.apply_adjoint(f.U[j], WV[0])
        WV[0].add_(f.V[j], -al[j])
        if j > 0:
            WV[0].add_(f.V[j - 1], -be[j - 1])
        assert WV[0].norm() < 1e-10, j
    smax = np.linalg.svd(f.rayleighquotient(), compute_uv=False)[0]
    assert 0.9 * (np.sqrt(m * per / n) + np.sqrt(per)) < smax < 1.1 * (np.sqrt(m * per / n) + np.sqrt(per))  # Marchenko-Pastur edge


# ---------------------------------------------------------------------------------------------------------------------
# Parity at BASELINE.json's headline sizes against the CPU restatement of the reference path (oracle/cpu_ref.c):
# north_star -- "Ritz values / residuals within 1e-10 relative for Float64, GMRES residual norm bit-matching iteration
# count ... eigenvalues matching CPU reference to 1e-10" on the 10M-row run itself.
# ---------------------------------------------------------------------------------------------------------------------
def _tri_eigs(al, be):
    return np.linalg.eigvalsh(np.diag(al) + np.diag(be[:-1], 1) + np.diag(be[:-1], -1))


@pytest.fixture(scope="module")
def cfg2_cpu_reference():
    """(alpha, beta) of initialize + 99 expand! on the 4000 x 2500 Laplacian by oracle/cpu_ref.c, per orthogonaliser
    (computed once per orthogonaliser: ~15-30 s of host time each)."""
    import cpu_ref_lib as cr
    from bench import laplacian_rows, NX, NY
    lib = cr.load()
    A = laplacian_rows(NX, NY, 0, NY)
    cache = {}

    def get(orth_code, x0):
        if orth_code not in cache:
            al, be, _, _ = cr.run_lanczos(lib, A, x0, 99, orth_code, nthreads=cr.usable_threads())
            cache[orth_code] = (al, be)
        return cache[orth_code]

    return get


@pytest.mark.parametrize("seed", [11, 29])
def test_config2_lanczos_10M_parity_more_start_vectors(kk, ctx, seed):
    """The headline parity figure (Ritz values of the 100 x 100 T, 1e-10 relative) is the rounding noise of the smallest Ritz
    value amplified by |T| / theta_min ~ 2e4: one start vector is thin evidence, so two more (the bench line carries the
    maximum over its own three).  MGS2 in the library's default mode = the timed configuration (strict order, persistent kernel)."""
    import cpu_ref_lib as cr
    from bench import laplacian_rows, NX, NY
    N, K = NX * NY, 100
    A = laplacian_rows(NX, NY, 0, NY)
    op = kk.SparseOperator(A, ctx, symmetric=True, via_csc=True)
    x0b = kk.DeviceBasis(N, 1, ctx)
    x0b[0].rand_(seed)
    V = kk.DeviceBasis(N, K + 2, ctx)
    it = kk.LanczosIterator(op, x0b[0], kk.ModifiedGramSchmidt2(), capacity=K + 2)
    f = kk.initialize(it, V)
    for _ in range(K - 1):
        f = kk.expand_(it, f)
    al_g, be_g = np.array(f.alphas), np.array(f.betas)
    al_c, be_c, _, _ = cr.run_lanczos(cr.load(), A, x0b[0].get(), 99, 3, nthreads=cr.usable_threads())
    assert np.max(np.abs(al_g - al_c) / np.abs(al_c)) <= 1e-10 and np.max(np.abs(be_g - be_c) / np.abs(be_c)) <= 1e-10
    th_g, th_c = _tri_eigs(al_g, be_g), _tri_eigs(al_c, be_c)
    assert np.max(np.abs(th_g - th_c) / np.abs(th_c)) <= 1e-10
    assert np.max(np.abs(th_g - th_c)) <= 1e-14 * np.max(np.abs(th_c)) * 10       # absolute agreement: a few ulp of |T|
    V.free(); x0b.free(); op.free()


@pytest.mark.parametrize("orth_name,mgs_mode", [("mgs2", 2), ("mgs2", 1), ("mgs2", 0), ("cgs2", 1)])
def test_config2_lanczos_10M_parity_with_cpu_reference(kk, ctx, cfg2_cpu_reference, orth_name, mgs_mode):
    """src/factorizations/lanczos.jl:250-272 + :313-338 at N = 10^7, krylovdim = 100: alpha / beta trajectories and the Ritz
    values of the 100 x 100 tridiagonal, GPU (MGS2 in the library's default mode = the reference's sequential order through
    the persistent kernel with the basis vector parked on chip, low-sync MGS2, strict MGS2 forced, CGS2) vs the CPU
    reference path, <= 1e-10 relative."""
    from bench import laplacian_rows, NX, NY
    N, K = NX * NY, 100
    ctx.set_option("mgs_mode", mgs_mode)
    try:
        op = kk.SparseOperator(laplacian_rows(NX, NY, 0, NY), ctx, symmetric=True, via_csc=True)
        x0b = kk.DeviceBasis(N, 1, ctx)
        x0b[0].rand_(3)
        x0 = x0b[0].get()
        V = kk.DeviceBasis(N, K + 2, ctx)
        orth = kk.Orthogonalizer(orth_name)
        it = kk.LanczosIterator(op, x0b[0], orth, capacity=K + 2)
        f = kk.initialize(it, V)
        for _ in range(K - 1):
            f = kk.expand_(it, f)
        al_g, be_g = np.array(f.alphas), np.array(f.betas)
        if orth_name == "mgs2":   # which kernels ran: the persistent one in strict / auto mode, the projection pair otherwise
            ctx.prof_reset(); ctx.prof_enable(1)
            kk.expand_(it, f)
            ctx.prof_enable(0)
            assert (ctx.prof_get("k_mgs_persist")[1] > 0) == (mgs_mode != 1) and (ctx.prof_get("k_project")[1] > 0) == (mgs_mode == 1)
    finally:
        ctx.set_option("mgs_mode", 2)
    al_c, be_c = cfg2_cpu_reference(orth.code, x0)
    assert np.max(np.abs(al_g - al_c) / np.abs(al_c)) <= 1e-10
    assert np.max(np.abs(be_g - be_c) / np.abs(be_c)) <= 1e-10
    th_g, th_c = _tri_eigs(al_g, be_g), _tri_eigs(al_c, be_c)
    assert np.max(np.abs(th_g - th_c) / np.abs(th_c)) <= 1e-10
    V.free(); x0b.free(); op.free()


@pytest.mark.parametrize("ny,expect_persist", [(2600, True), (2650, False)])
def test_lanczos_parity_on_both_sides_of_the_register_file_limit(kk, ctx, ny, expect_persist):
    """The register-resident strict-MGS kernel holds work vectors of at most 10.48 M rows (`persist_capacity_rows`); the
    headline vector uses 95 % of that.  4000 x 2600 = 10.4 M rows still runs it, 4000 x 2650 = 10.6 M rows takes the
    low-synchronisation pair in auto mode (the line's `mgs2_lowsync` leg): both must match the CPU reference path
    (lanczos.jl:250-338) to 1e-10 over 40 steps, and each must run the kernels it is supposed to."""
    import cpu_ref_lib as cr
    from bench import laplacian_rows, NX
    N, K = NX * ny, 41
    assert (N <= ctx.get_option("persist_capacity_rows")) == expect_persist
    A = laplacian_rows(NX, ny, 0, ny)
    op = kk.SparseOperator(A, ctx, symmetric=True, via_csc=True)
    x0b = kk.DeviceBasis(N, 1, ctx)
    x0b[0].rand_(17)
    V = kk.DeviceBasis(N, K + 2, ctx)
    it = kk.LanczosIterator(op, x0b[0], kk.ModifiedGramSchmidt2(), capacity=K + 2)
    ctx.prof_reset(); ctx.prof_enable(1)
    f = kk.initialize(it, V)
    for _ in range(K - 1):
        f = kk.expand_(it, f)
    ctx.prof_enable(0)
    assert (ctx.prof_get("k_mgs_persist")[1] > 0) == expect_persist and (ctx.prof_get("k_project")[1] > 0) == (not expect_persist)
    al_g, be_g = np.array(f.alphas), np.array(f.betas)
    al_c, be_c, _, _ = cr.run_lanczos(cr.load(), A, x0b[0].get(), K - 1, 3, nthreads=cr.usable_threads())
    assert np.max(np.abs(al_g - al_c) / np.abs(al_c)) <= 1e-10 and np.max(np.abs(be_g - be_c) / np.abs(be_c)) <= 1e-10
    assert _gram_offdiag_max(kk, f.V, K) < 1e-12
    V.free(); x0b.free(); op.free()


def test_config3_gmres_2M_parity_with_cpu_reference(kk, ctx):
    """src/linsolve/gmres.jl:44-149 on the 2M-row convection-diffusion operator, krylovdim = 60: one full restart cycle
    plus the start of the second.  numiter / numops / converged equal to the CPU reference path, the residual estimate
    after EVERY inner step (gmres.jl:53,94) and the final residual norm within 1e-10 relative."""
    import cpu_ref_lib as cr
    from tools.bench_configs import convdiff
    nx, ny = 2000, 1000
    N = nx * ny
    A = convdiff(nx, ny)
    b = np.random.default_rng(4).random(N)
    nb = np.linalg.norm(b)
    tol = 1e-10 * nb
    lib = cr.load()
    xc, ic, tc = cr.run_gmres(lib, A, b, None, 0.0, 1.0, 60, 2, tol, 3, nthreads=cr.usable_threads())
    tr = []
    ctx.prof_reset(); ctx.prof_enable(1)
    x, info = kk.linsolve(kk.SparseOperator(A, ctx), b, None, kk.GMRES(kk.ModifiedGramSchmidt2(), 2, 60, tol), trace=tr)
    ctx.prof_enable(0)
    # which kernel this parity figure belongs to: the default route of a 2M-row MGS2 sweep is the persistent
    # PANEL kernel with two vectors per grid reduction -- exact algebra, NOT the reference's association of the operations
    assert ctx.prof_get("k_mgs_panel")[1] >= 2 * 59 and ctx.prof_get("k_project")[1] == 0 and ctx.prof_get("k_mgs_persist")[1] == 0
    assert (info.converged, info.numiter, info.numops) == (ic["converged"], ic["numiter"], ic["numops"])
    tg = np.array([t[2] for t in tr])
    assert len(tg) == len(tc) == 120
    assert np.max(np.abs(tg - tc) / tc) <= 1e-10
    assert abs(info.normres - ic["normres"]) <= 1e-10 * ic["normres"]
    assert np.linalg.norm(x - xc) <= 1e-9 * np.linalg.norm(xc)
    # CGS2 as well: same counts, same trace to 1e-10
    xc2, ic2, tc2 = cr.run_gmres(lib, A, b, None, 0.0, 1.0, 60, 1, tol, 2, nthreads=cr.usable_threads())
    tr2 = []
    x2, info2 = kk.linsolve(kk.SparseOperator(A, ctx), b, None, kk.GMRES(kk.ClassicalGramSchmidt2(), 1, 60, tol), trace=tr2)
    assert (info2.converged, info2.numiter, info2.numops) == (ic2["converged"], ic2["numiter"], ic2["numops"])
    assert np.max(np.abs(np.array([t[2] for t in tr2]) - tc2) / tc2) <= 1e-10


def test_config3b_gmres_2M_converges_with_equal_iteration_count(kk, ctx):
    """north_star: "GMRES residual norm bit-matching iteration count" on a run that CONVERGES.  The 2M-row
    convection-diffusion operator shifted by a0 = 0.15 (linsolve's a0 + a1 A form, src/linsolve/gmres.jl:3-12) reaches
    rtol 1e-10 with GMRES(60) in the 4th restart cycle: converged, numiter, numops (the inner loop stops mid-cycle, :55)
    and the final residual norm must equal the CPU reference path, the whole residual-estimate trace to 1e-10."""
    import cpu_ref_lib as cr
    from tools.bench_configs import convdiff
    nx, ny = 2000, 1000
    N = nx * ny
    A = convdiff(nx, ny)
    b = np.random.default_rng(4).random(N)
    nb = np.linalg.norm(b)
    tol, a0 = 1e-10 * nb, 0.15
    lib = cr.load()
    for orth_code, orth in ((3, kk.ModifiedGramSchmidt2()), (2, kk.ClassicalGramSchmidt2())):
        xc, ic, tc = cr.run_gmres(lib, A, b, None, a0, 1.0, 60, 20, tol, orth_code, nthreads=cr.usable_threads())
        assert ic["converged"] == 1 and 3 <= ic["numiter"] <= 8, ic
        tr = []
        ctx.prof_reset(); ctx.prof_enable(1)
        x, info = kk.linsolve(kk.SparseOperator(A, ctx), b, None, kk.GMRES(orth, 20, 60, tol), a0, 1.0, trace=tr)
        ctx.prof_enable(0)
        if orth_code == 3:    # MGS2: every sweep through k_mgs_panel (two vectors per reduction), none through the projection pair
            assert ctx.prof_get("k_mgs_panel")[1] >= len(tc) - info.numiter and ctx.prof_get("k_project")[1] == 0
        assert (info.converged, info.numiter, info.numops) == (ic["converged"], ic["numiter"], ic["numops"]), (info, ic)
        tg = np.array([t[2] for t in tr])
        assert len(tg) == len(tc)
        assert np.max(np.abs(tg - tc) / tc) <= 1e-10
        # the final norm is the explicitly recomputed |b - (a0 + a1 A) x| ~ 1e-10 |b|: its own rounding is eps |b|, i.e.
        # ~1e-6 relative to itself -- compare on the scale of the tolerance it is tested against
        assert abs(info.normres - ic["normres"]) <= 1e-4 * tol and info.normres < tol
        assert np.linalg.norm(x - xc) <= 1e-9 * np.linalg.norm(xc)
        r_true = b - (a0 * x + A @ x)
        assert abs(np.linalg.norm(r_true) - info.normres) <= 1e-4 * tol


@pytest.mark.parametrize("orth_name", ["mgs2", "cgs2"])
def test_config4_gkl_5Mx1M_parity_with_cpu_reference(kk, ctx, orth_name):
    """src/factorizations/gkl.jl:183-215, 246-269, 308-346 at BASELINE size (5M x 1M, 20 nnz/row, 29 expand! steps):
    alpha / beta of every step and the 20 largest singular values of the 30 x 30 bidiagonal against oracle/cpu_ref.c
    (kkref_gkl), <= 1e-10 relative."""
    import cpu_ref_lib as cr
    from bench import gkl_rows
    m, n, per, K = 5_000_000, 1_000_000, 20, 30
    A = gkl_rows(m, n, per, 0, m)
    u0 = np.random.default_rng([6, 0]).random(m)
    orth = kk.Orthogonalizer(orth_name)
    op = kk.SparseOperator(A, ctx)
    it = kk.GKLIterator(op, u0, orth, capacity=K + 2)
    f = kk.initialize(it)
    for _ in range(K - 1):
        f = kk.expand_(it, f)
    al_g, be_g = np.array(f.alphas), np.array(f.betas)
    Bg = f.rayleighquotient()
    del f, it
    op.free()
    al_c, be_c, _, _ = cr.run_gkl(cr.load(), A, u0, K - 1, orth.code, nthreads=cr.usable_threads())
    assert np.max(np.abs(al_g - al_c) / np.abs(al_c)) <= 1e-10, np.max(np.abs(al_g - al_c) / np.abs(al_c))
    assert np.max(np.abs(be_g - be_c) / np.abs(be_c)) <= 1e-10, np.max(np.abs(be_g - be_c) / np.abs(be_c))
    Bc = np.diag(al_c) + np.diag(be_c[:-1], -1)
    assert np.max(np.abs(np.abs(Bg) - np.abs(Bc))) <= 1e-10 * np.max(np.abs(Bc))     # same bidiagonal (either triangle convention)
    sg, sc = np.linalg.svd(Bg, compute_uv=False)[:20], np.linalg.svd(Bc, compute_uv=False)[:20]
    assert np.max(np.abs(sg - sc) / sc) <= 1e-10


@pytest.mark.parametrize("block_mode", [1, 0])
def test_config5_blocklanczos_10M_parity_with_cpu_reference(kk, ctx, block_mode):
    """src/factorizations/blocklanczos.jl:159-263, 312-353 at BASELINE size (10M rows, block size 16, 16 -> 112 basis
    vectors): the 112 x 112 block-tridiagonal matrix of the GPU step -- default mode (CholQR2 + one-pass projection with
    Gram correction, NOT the reference's operation order) and strict mode (the reference's own order) -- against
    oracle/cpu_ref.c (kkref_blocklanczos: MGS block_qr!, three-term recurrence, MGS re-orthogonalisation).  QR factors
    with positive diagonals are unique, so the blocks must agree entry by entry; eigenvalues <= 1e-10 relative to |H|."""
    import cpu_ref_lib as cr
    from bench import laplacian_rows, NX, NY
    N, bs, K = NX * NY, 16, 100
    A = laplacian_rows(NX, NY, 0, NY)
    ctx.set_option("block_mode", block_mode)
    try:
        op = kk.SparseOperator(A, ctx, symmetric=True)
        S = kk.DeviceBasis(N, K + 3 * bs, ctx)
        it = kk.BlockLanczosIterator(op, [None] * bs, K + bs)
        area_b = it.maxdim + bs
        X0 = np.empty((N, bs), order="F")
        for j in range(bs):
            S[area_b + j].rand_(100 + j)
            X0[:, j] = S[area_b + j].get()
        it.x0 = [S[area_b + j] for j in range(bs)]
        f = it.initialize(S)
        while len(f) < K:
            f = it.expand(f)
        k = len(f)
        Hg = f.H[:k, :k].copy()
        nR_g = f.normres
    finally:
        ctx.set_option("block_mode", 1)
    S.free(); op.free()
    out = cr.run_blocklanczos(cr.load(), A, X0, target_dim=K, max_steps=6, qr_tol=it.qr_tol, nthreads=cr.usable_threads())
    assert out["k"] == k == 7 * bs and out["sizes"] == [bs] * 7
    Hc = out["H"]
    scale = np.max(np.abs(Hc))
    eg, ec = np.linalg.eigvalsh((Hg + Hg.T) / 2), np.linalg.eigvalsh((Hc + Hc.T) / 2)
    assert np.max(np.abs(eg - ec)) <= 1e-10 * scale, np.max(np.abs(eg - ec)) / scale
    # diagonal blocks M_j and sub-diagonal blocks B_j, entry by entry
    assert np.max(np.abs(Hg - Hc)) <= 1e-9 * scale, np.max(np.abs(Hg - Hc)) / scale
    assert abs(nR_g - out["norm_R"]) <= 1e-9 * out["norm_R"]


def test_config2_lanczos_10M_apply_inside_the_persistent_kernel(kk, ctx):
    """k_mgs_persist<.., APPLY> (round 6): at BASELINE config 2's size the strict MGS2 Lanczos step forms w = A v - beta v_prev and alpha0 = <v, w>
    INSIDE the sweep launch (value-free stencil) -- no apply launch, no store and re-load of w.  Same w bits; alpha0 summed in another order: the
    factorization equals the one with persist_apply = 0 to rounding.  And a launch that gives up (test hook) takes its apply with it: the
    recovery repeats both (factorizations/lanczos.jl:250-336)"""
    from bench import laplacian_rows, NX, NY
    N, steps = NX * NY, 16
    op = kk.SparseOperator(laplacian_rows(NX, NY, 0, NY), ctx, symmetric=True, via_csc=True)
    V = kk.DeviceBasis(N, steps + 4, ctx)
    x0 = kk.DeviceBasis(N, 1, ctx)
    x0[0].rand_(3)
    it = kk.LanczosIterator(op, x0[0], kk.ModifiedGramSchmidt2(), capacity=steps + 3)
    out = {}
    for mode in ("inside", "separate", "inside_fault"):
        ctx.set_option("persist_apply", 0 if mode == "separate" else 1)
        l0, t0 = ctx.get_option("persist_apply_launches"), ctx.get_option("persist_timeouts")
        ctx.prof_reset(); ctx.prof_enable(1)
        f = kk.initialize(it, V)
        for i in range(steps):
            if mode == "inside_fault" and i in (4, 9):
                ctx.set_option("persist_fault", 1)
            f = kk.expand_(it, f)
        ctx.prof_enable(0)
        out[mode] = (np.array(f.alphas), np.array(f.betas), int(ctx.get_option("persist_apply_launches") - l0), ctx.prof_get("k_spmv_dia")[1],
                     int(ctx.get_option("persist_timeouts") - t0), _gram_offdiag_max(kk, V, steps))
    ctx.set_option("persist_apply", 1)
    a, s, ft = out["inside"], out["separate"], out["inside_fault"]
    assert a[2] >= steps - 1 and s[2] == 0, (a[2], s[2])                         # every step's sweep launch applied the operator itself ...
    assert a[3] <= s[3] - (steps - 1), (a[3], s[3])                              # ... and the apply launches are gone (initialize keeps its own)
    rel = lambda x, y: float(np.max(np.abs(x - y) / np.abs(y)))
    assert rel(a[0], s[0]) < 1e-12 and rel(a[1], s[1]) < 1e-12
    assert ft[4] == 2 and rel(ft[0], s[0]) < 1e-12 and rel(ft[1], s[1]) < 1e-12  # two lost launches, each repeated WITH its apply
    assert max(a[5], s[5], ft[5]) < 1e-12
