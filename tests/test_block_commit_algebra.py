"""CPU check of the algebra behind the normalised commit of the one-pass BlockLanczos step (csrc/kk_block.hip,
k_blk_panel_correct / k_blk_resid_gram / k_blk_commit_prep).  No GPU, no library: plain NumPy.

One-pass step: P = V'(A X) against the whole basis, coefficients corrected to first order for the basis not being exactly
orthonormal, Pc = (I - E) P with E = V'V - I, residual block W = A X - V Pc (reference: blocklanczos.jl:253-260 three-term
recurrence + :277-284 re-orthogonalisation, which project twice instead).  Claims checked here:
  1. with the FULL E (off-diagonal part and the diagonal |v_i|^2 - 1) V'W = O(E^2 |P|) and the Gram matrix of the residual
     block is predicted by (A X)'(A X) - P'Pc to O(E^2);
  2. with the off-diagonal part only -- what the library did until round 4 -- V'W keeps -D P and the prediction is off by
     -P'D P: the term that made the committed first CholQR2 factor miss the 2e-14 skip threshold;
  3. T = W R1^-1 with R1 = chol(predicted Gram), then R2 = chol(T'T): Q = T R2^-1 is orthonormal to rounding, W = Q (R2 R1),
     and W = T R1 recovers the residual block (what blk_commit_flush forms on demand)."""
import numpy as np


def setup(scale_diag, scale_off, seed=3, n=4000, k=48, p=8):
    rng = np.random.default_rng(seed)
    Q, _ = np.linalg.qr(rng.standard_normal((n, k)))
    # a basis that is NOT exactly orthonormal: column norms off by ~scale_diag, small off-diagonal overlaps ~scale_off
    d = 1.0 + scale_diag * rng.uniform(0.5, 1.0, k)
    S = np.eye(k) + scale_off * np.triu(rng.standard_normal((k, k)), 1)
    V = (Q * d) @ S
    AX = V @ rng.standard_normal((k, p)) * 3.0 + rng.standard_normal((n, p))      # mostly inside span(V), like A X of a Krylov step
    return V, AX


def step(V, AX, full):
    k = V.shape[1]
    G = V.T @ V
    E = G - np.eye(k)
    if not full:
        E = E - np.diag(np.diag(E))
    P = V.T @ AX
    Pc = P - E @ P
    W = AX - V @ Pc
    pred = AX.T @ AX - P.T @ Pc
    pred = 0.5 * (pred + pred.T)
    return P, Pc, W, pred, np.diag(G) - 1.0


def test_full_correction_leaves_second_order_terms_only():
    V, AX = setup(scale_diag=1e-7, scale_off=1e-7)
    P, Pc, W, pred, D = step(V, AX, full=True)
    e = 1e-7 * np.sqrt(V.shape[1])
    assert np.max(np.abs(V.T @ W)) < 50 * e * e * np.max(np.abs(P))            # O(E^2 |P|)
    actual = W.T @ W
    assert np.max(np.abs(pred - actual)) < 50 * e * e * np.max(np.abs(P)) ** 2


def test_off_diagonal_correction_alone_keeps_the_norm_term():
    V, AX = setup(scale_diag=1e-7, scale_off=1e-7)
    P, Pc, W, pred, D = step(V, AX, full=False)
    # V'W = -D P (+ second order): the component the off-diagonal correction does not remove
    np.testing.assert_allclose(V.T @ W, -D[:, None] * P, atol=1e-11 * np.max(np.abs(P)))
    # predicted - actual Gram matrix = - P'DP (+ second order)
    actual = W.T @ W
    np.testing.assert_allclose(pred - actual, -(P.T * D) @ P, atol=1e-11 * np.max(np.abs(P)) ** 2)
    assert np.max(np.abs(pred - actual)) > 1e3 * 1e-14 * np.max(np.abs(actual))   # far above what the full correction leaves
    Pf, Pcf, Wf, predf, _ = step(V, AX, full=True)
    assert np.max(np.abs(predf - Wf.T @ Wf)) < 1e-3 * np.max(np.abs(pred - actual))


def test_committed_block_is_a_cholqr2_first_round():
    V, AX = setup(scale_diag=2e-14, scale_off=1e-15)       # the magnitudes of the library: norms off by <= the 2e-14 skip threshold
    P, Pc, W, pred, D = step(V, AX, full=True)
    R1 = np.linalg.cholesky(pred).T                         # upper factor of the PREDICTED Gram matrix
    T = W @ np.linalg.inv(R1)                               # what k_block_update_commit writes into the next basis slot
    G2 = T.T @ T
    assert np.max(np.abs(G2 - np.eye(G2.shape[0]))) < 2e-14  # ... and it passes the skip test of the second round
    R2 = np.linalg.cholesky(G2).T
    Q = T @ np.linalg.inv(R2)
    assert np.max(np.abs(Q.T @ Q - np.eye(Q.shape[1]))) < 5e-15
    B = R2 @ R1
    np.testing.assert_allclose(Q @ B, W, atol=1e-13 * np.max(np.abs(W)))
    np.testing.assert_allclose(T @ R1, W, atol=1e-13 * np.max(np.abs(W)))      # blk_commit_flush
    # with the off-diagonal correction only, the same norms push the first round over the threshold
    P0, Pc0, W0, pred0, _ = step(*setup(scale_diag=2e-14, scale_off=1e-15), full=False)
    T0 = W0 @ np.linalg.inv(np.linalg.cholesky(pred0).T)
    assert np.max(np.abs(T0.T @ T0 - np.eye(T0.shape[1]))) > np.max(np.abs(G2 - np.eye(G2.shape[0])))
