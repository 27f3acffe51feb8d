"""BASELINE.json configs[0]: eigsolve(Lanczos) on a 10 000 x 10 000 dense Symmetric(rand) Float64 matrix, krylovdim = 30,
CPU reference path (plumbing, no GPU).  Inputs as SURVEY.md 8(d) cfg 1 fixes them: R = rng(1).random((n, n)),
A = (R + R') / 2, x0 = rng(2).random(n), howmany = 1, :LM, tol = 1e-12; the oracle's restatement of
/root/reference/src/eigsolve/lanczos.jl:11-154 must reproduce the extreme eigenvalue of the dense LAPACK solve to 1e-10
(the check the reference's own tests make against `eigen`, test/eigsolve.jl:74,122-123) and satisfy the convergence
record the driver returns (`converged`, `normres`, `numops`).  Runs at the stated size (about a minute on 8 cores: the
LAPACK tridiagonalisation is the expensive half)."""
import numpy as np
import scipy.linalg as sla


def test_config1_dense_symmetric_10000_krylovdim30(ko):
    n, krylovdim, tol = 10_000, 30, 1e-12
    R = np.random.default_rng(1).random((n, n))
    A = R + R.T
    A *= 0.5
    del R
    x0 = np.random.default_rng(2).random(n)
    trace = []
    vals, vecs, info = ko.eigsolve_lanczos(A, x0, 1, "LM", krylovdim=krylovdim, maxiter=100, tol=tol, orth=ko.MGS2, trace=trace)
    assert info.converged >= 1 and info.normres[0] <= tol
    assert info.numops <= 100 * krylovdim and info.numiter >= 1
    lam, x = vals[0], vecs[0]
    # eigenpair quality as the reference's tests state it: A x = lambda x to the tolerance, |x| = 1
    assert abs(np.linalg.norm(x) - 1.0) < 1e-12
    assert np.linalg.norm(A @ x - lam * x) <= 10 * tol * max(1.0, abs(lam))
    # residual record = what the driver computed from the factorization (eigsolve/lanczos.jl:128-146)
    assert abs(np.linalg.norm(info.residual[0]) - info.normres[0]) <= 1e-12
    # dense LAPACK on the same matrix (one tridiagonalisation, all eigenvalues): :LM picks the end of larger magnitude
    ev = sla.eigh(A, eigvals_only=True, overwrite_a=False, check_finite=False)
    lo, hi = ev[0], ev[-1]
    exact = hi if abs(hi) >= abs(lo) else lo
    assert abs(lam - exact) <= 1e-10 * abs(exact)
    # Symmetric(rand) has ONE far eigenvalue near n / 2 (Perron) and a bulk of radius ~ sqrt(n / 3): without `eager` the
    # driver looks at convergence for the first time when the factorization is full (eigsolve/lanczos.jl:45) -> 30 operations
    assert abs(exact - n / 2) < 0.01 * n and info.numiter == 1 and info.numops == krylovdim
    assert len(trace) == 1 and trace[0][1] == krylovdim
    # the other end (:SR) through the same driver: inside the bulk the gaps are tiny, so thick restarts are exercised
    # (eigsolve/lanczos.jl:80-116); ten of them do not converge, but every Ritz value is a Rayleigh quotient (>= lambda_min)
    # and the restarts may only improve it
    tr2 = []
    vals2, _, info2 = ko.eigsolve_lanczos(A, x0, 1, "SR", krylovdim=krylovdim, maxiter=10, tol=1e-12, orth=ko.MGS2, trace=tr2)
    assert info2.numiter == 10 and info2.converged == 0 and len(tr2) == 10
    keep = (3 * krylovdim) // 5
    assert info2.numops == krylovdim + 9 * (krylovdim - keep)
    ritz = [t[2][0] for t in tr2]
    assert all(r >= lo - 1e-10 * abs(lo) for r in ritz)
    assert all(ritz[i + 1] <= ritz[i] + 1e-10 * abs(lo) for i in range(len(ritz) - 1))
    assert ritz[-1] - lo < 0.05 * abs(lo)
