"""Host-side dense machinery of the Arnoldi-family drivers (tests/hostmirror_extras.py -- test infrastructure since round 4: Schur form, reordering, eigenvectors,
restoring the Arnoldi form) and the algorithm structs -- no GPU involved; checked against NumPy / SciPy and against the
independent restatement in the oracle (dense/linalg.jl:152-383, eigsolve/arnoldi.jl:466-480, algorithms.jl)."""
import numpy as np
import pytest
import scipy.linalg as sla


@pytest.mark.parametrize("which", ["LM", "LR", "SR"])
def test_schur_reordering_and_eigenvectors(kk, ko, which):
    import hostmirror_extras as dense
    rng = np.random.default_rng(5)
    for n in (1, 2, 7, 12):
        H = sla.hessenberg(rng.standard_normal((n, n)))
        T, U, vals = dense.hschur(H.copy())
        np.testing.assert_allclose(U @ T @ U.T, H, atol=1e-12)
        np.testing.assert_allclose(np.sort_complex(vals), np.sort_complex(np.linalg.eigvals(H)), atol=1e-10)
        p = dense.sortperm_general(vals, which)
        T2, U2, v2 = dense.permuteschur(T, U, p)
        np.testing.assert_allclose(U2 @ T2 @ U2.T, H, atol=1e-11)                 # still a Schur decomposition
        np.testing.assert_allclose(U2.T @ U2, np.eye(n), atol=1e-12)
        assert np.allclose(np.tril(T2, -2), 0)
        by, rev = dense.eigsort_general(which)
        key = by(v2)
        assert np.all(np.diff(key) <= 1e-9) if rev else np.all(np.diff(key) >= -1e-9)   # sorted as requested
        np.testing.assert_allclose(v2, ko._schur_values(T2), atol=1e-13)           # same diagonal-order eigenvalues as the oracle
        X = dense.schur2eigvecs(T2)
        for j in range(n):
            assert abs(np.linalg.norm(X[:, j]) - 1) < 1e-12
            assert np.linalg.norm(T2 @ X[:, j] - v2[j] * X[:, j]) < 1e-9 * max(1.0, abs(v2[j]))


def test_permuteschur_refuses_to_split_a_block(kk):
    import hostmirror_extras as dense
    T = np.array([[1.0, 2.0, 0.3], [-2.0, 1.0, 0.1], [0.0, 0.0, 5.0]])              # 2x2 block (1 +- 2i), then 5
    with pytest.raises(RuntimeError):
        dense.permuteschur(T, np.eye(3), [0, 2, 1])
    T2, Q2, v = dense.permuteschur(T, np.eye(3), [2, 0, 1])
    np.testing.assert_allclose(v[0], 5.0, atol=1e-12)
    np.testing.assert_allclose(Q2 @ T2 @ Q2.T, T, atol=1e-12)


def test_restorearnoldiform_matches_oracle_and_keeps_the_krylov_relation(kk, ko):
    import hostmirror_extras as dense
    rng = np.random.default_rng(9)
    K, keep = 9, 5
    T = np.triu(rng.standard_normal((K, K)))
    f = rng.standard_normal(K)
    U1, H1 = np.eye(K), T.copy()
    dense.restorearnoldiform(U1, H1, f, keep)
    U2, H2 = np.eye(K), T.copy()
    ko._restore_arnoldi_form(U2, H2, f, keep)
    np.testing.assert_allclose(H1, H2, atol=1e-13)
    np.testing.assert_allclose(U1, U2, atol=1e-13)
    Hk = H1[: keep + 1, :keep]
    assert np.allclose(np.tril(Hk, -2), 0)                                         # upper Hessenberg again
    Uk = U1[:keep, :keep]
    np.testing.assert_allclose(Uk.T @ Uk, np.eye(keep), atol=1e-12)
    # [T_kk ; f_k'] = [U 0; 0 1] [H_k ; nu e_k'] U'  restricted to the kept block
    np.testing.assert_allclose(Uk @ Hk[:keep] @ Uk.T, T[:keep, :keep], atol=1e-11)
    np.testing.assert_allclose(Hk[keep, keep - 1] * Uk[:, keep - 1], f[:keep], atol=1e-11)


def test_algorithm_struct_defaults_follow_the_reference(kk):
    """algorithms.jl:555-562 KrylovDefaults and the keyword constructors."""
    import hostmirror_extras as hx
    for cls in (kk.Lanczos, kk.Arnoldi, hx.BiArnoldi, kk.GKL, hx.GolubYe, kk.GMRES, kk.LSMR):
        a = cls()
        assert (a.krylovdim, a.maxiter, a.tol) == (30, 100, 1e-12)
    assert kk.BlockLanczos().krylovdim == 100                                      # algorithms.jl:561
    for cls in (kk.CG, kk.BiCGStab):
        a = cls()
        assert (a.maxiter, a.tol) == (100, 1e-12)
    assert kk.LSMR().orth.name == "mgs" and kk.Lanczos().orth.name == "mgs2"      # algorithms.jl:517, 556
    assert kk.ModifiedGramSchmidtIR().eta == pytest.approx(1 / np.sqrt(2))         # algorithms.jl:66,80
