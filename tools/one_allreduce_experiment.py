"""SURVEY 8(e) / VERDICT round 3 item 6: can the second all-reduce of the sharded Lanczos step (|w'|^2 after the update) be
folded into the first one?  The first reduction already carries d = V'w (and alpha0); with ww = <w, w> added to it,
    |w - V x|^2 = ww - 2 x'd + x'G x          (G = V'V, known)
needs no second reduction.  The price is cancellation: the result is formed from quantities of size |A v|^2.  This script
runs the BASELINE config-2 recurrence (5-point Laplacian, krylovdim 100, full re-orthogonalisation in the low-sync form) on a
reduced grid in float64, forms beta both ways and reports the relative difference, the amplification |Av|^2 / beta^2, and
what it does to the Ritz values of T.  CPU only; usage: python tools/one_allreduce_experiment.py [nx ny [variant]]"""
import json
import sys

import numpy as np
import scipy.sparse as sp


def laplacian(nx, ny, shift=None):
    ex, ey = np.ones(nx), np.ones(ny)
    Tx = sp.diags([-ex[:-1], 2 * ex, -ex[:-1]], [-1, 0, 1])
    Ty = sp.diags([-ey[:-1], 2 * ey, -ey[:-1]], [-1, 0, 1])
    A = (sp.kron(sp.identity(ny), Tx) + sp.kron(Ty, sp.identity(nx))).tocsr()
    if shift is not None:
        A = A + sp.diags(shift)
    return A.tocsr()


def run(A, x0, K, fold):
    n = A.shape[0]
    V = np.zeros((n, K + 1))
    beta0 = np.linalg.norm(x0)
    V[:, 0] = x0 / beta0
    alphas, betas, amp, rel = [], [], [], []
    G = np.zeros((K + 1, K + 1))
    bprev = 0.0
    for k in range(K):
        v = V[:, k]
        w = A @ v
        if k > 0:
            w -= bprev * V[:, k - 1]
        m = k + 1
        Vm = V[:, :m]
        # first (and, folded, only) reduction: alpha0 = <v, w>, V'w, V'v -- with ww = <w, w> riding along
        dw = Vm.T @ w
        dv = Vm.T @ v
        a0 = float(v @ w)
        ww = float(w @ w)
        G[:m, m - 1] = dv
        G[m - 1, :m] = dv
        Gm = G[:m, :m]
        # lanczos.jl:325-338 (MGS2): w -= alpha0 v, then one MGS sweep over all of V -- in the library's low-synchronisation
        # form: rhs = V'(w - alpha0 v), s_i = rhs_i - sum_{k<i} G_ik s_k (k_lowsync_solve), alpha0 folded into the last coefficient
        x = np.linalg.solve(np.eye(m) + np.tril(Gm, -1), dw - a0 * dv)
        x[-1] += a0
        d = dw
        wn = w - Vm @ x
        b_direct = float(np.linalg.norm(wn))
        b2_fold = ww - 2.0 * float(x @ d) + float(x @ (Gm @ x))
        b_fold = float(np.sqrt(max(b2_fold, 0.0)))
        amp.append(ww / b_direct ** 2)
        rel.append(abs(b_fold - b_direct) / b_direct)
        beta = b_fold if fold else b_direct
        alphas.append(float(x[-1]))
        betas.append(beta)
        V[:, k + 1] = wn / beta
        bprev = beta
    T = np.diag(alphas) + np.diag(betas[:-1], 1) + np.diag(betas[:-1], -1)
    return np.array(alphas), np.array(betas), np.linalg.eigvalsh(T), np.array(amp), np.array(rel), V


def main():
    nx, ny = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (400, 250)
    variant = sys.argv[3] if len(sys.argv) > 3 else "2"
    n = nx * ny
    shift = 10 * np.linspace(0, 1, n) ** 2 if variant == "2b" else None
    A = laplacian(nx, ny, shift)
    K = 100
    out = []
    for seed in (0, 1, 2):
        x0 = np.random.default_rng(seed).random(n)
        a0, b0, r0, amp, rel, V0 = run(A, x0, K - 1, fold=False)
        a1, b1, r1, _, _, V1 = run(A, x0, K - 1, fold=True)
        out.append({"seed": seed,
                    "amplification_ww_over_beta2": {"median": float(np.median(amp)), "max": float(amp.max())},
                    "beta_folded_vs_norm_same_vectors": {"median": float(np.median(rel)), "max": float(rel.max())},
                    "trajectory_with_folded_beta_vs_reference": {
                        "alpha_relerr": float(np.max(np.abs(a1 - a0) / np.abs(a0))),
                        "beta_relerr": float(np.max(np.abs(b1 - b0) / np.abs(b0))),
                        "ritz_relerr": float(np.max(np.abs(r1 - r0) / np.abs(r0))),
                        "orthogonality": float(np.max(np.abs(V1[:, :K].T @ V1[:, :K] - np.eye(K))))}})
    print(json.dumps({"grid": [nx, ny], "variant": variant, "krylovdim": K, "runs": out}, indent=1))


if __name__ == "__main__":
    main()
