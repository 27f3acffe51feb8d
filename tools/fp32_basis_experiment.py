"""SURVEY.md 8(f)-4 experiment, accuracy side: Lanczos with the reference's MGS2 recurrence (factorizations/lanczos.jl:325-338)
when the stored Krylov vectors are rounded to fp32 (all arithmetic in f64), against the all-f64 run -- (alpha, beta)
trajectories, Ritz values of T, orthogonality of the basis.  CPU / NumPy, runs anywhere:  python tools/fp32_basis_experiment.py
(the speed side is tools/fp32_basis.hip; results of both: profiles/r02_fp32_basis_experiment.txt)"""
import numpy as np
import scipy.sparse as sp


def laplacian_2d(nx, ny):
    ex, ey = np.ones(nx), np.ones(ny)
    Tx = sp.diags([-ex[:-1], 2 * ex, -ex[:-1]], [-1, 0, 1])
    Ty = sp.diags([-ey[:-1], 2 * ey, -ey[:-1]], [-1, 0, 1])
    return (sp.kron(sp.identity(ny), Tx) + sp.kron(Ty, sp.identity(nx))).tocsr()


def lanczos_mgs2(A, x0, K, store):
    """store(v) -> the vector as it sits in the basis (identity or a round trip through fp32)"""
    v = store(x0 / np.linalg.norm(x0))
    V = [v]
    w = A @ v
    alpha = v @ w
    w = w - alpha * v
    d = v @ w
    alpha += d
    w = w - d * v
    al, be = [alpha], [np.linalg.norm(w)]
    for _ in range(K - 1):
        beta = be[-1]
        v = store(w / beta)            # push!(V, scale!!(r, 1/beta))  -- the stored copy is what every later pass reads
        V.append(v)
        w = A @ v
        w = w - beta * V[-2]
        a = v @ w
        w = w - a * v
        s = 0.0
        for q in V:                    # second pass, sequential MGS
            s = q @ w
            w = w - s * q
        al.append(a + s)
        be.append(np.linalg.norm(w))
    return np.array(al), np.array(be), np.stack(V, 1)


def main():
    nx, ny, K = 400, 250, 100
    A = laplacian_2d(nx, ny)
    x0 = np.random.default_rng(3).random(nx * ny)
    tri = lambda a, b: np.linalg.eigvalsh(np.diag(a) + np.diag(b[:-1], 1) + np.diag(b[:-1], -1))
    a64, b64, V64 = lanczos_mgs2(A, x0, K, lambda v: v)
    a32, b32, V32 = lanczos_mgs2(A, x0, K, lambda v: v.astype(np.float32).astype(np.float64))
    t64, t32 = tri(a64, b64), tri(a32, b32)
    print(f"{nx}x{ny} 5-point Laplacian, krylovdim {K}, MGS2; basis stored in fp32 (f64 arithmetic) vs all-f64:")
    print(f"  max |alpha32 - alpha64| / |alpha64| = {np.max(np.abs(a32 - a64) / np.abs(a64)):.2e}")
    print(f"  max |beta32  - beta64 | / |beta64 | = {np.max(np.abs(b32 - b64) / np.abs(b64)):.2e}")
    print(f"  max relative Ritz-value difference   = {np.max(np.abs(t32 - t64) / np.abs(t64)):.2e}   (parity bar of the path: 1e-10)")
    print(f"  max |V'V - I|: f64 basis {np.max(np.abs(V64.T @ V64 - np.eye(K))):.2e}, fp32-stored basis {np.max(np.abs(V32.T @ V32 - np.eye(K))):.2e}")
    R = A @ V32[:, :-1] - V32 @ (np.diag(a32) + np.diag(b32[:-1], 1) + np.diag(b32[:-1], -1))[:, :-1]
    print(f"  Krylov relation residual |A V - V T| with the fp32-stored basis: {np.max(np.abs(R)):.2e} (f64: "
          f"{np.max(np.abs(A @ V64[:, :-1] - V64 @ (np.diag(a64) + np.diag(b64[:-1], 1) + np.diag(b64[:-1], -1))[:, :-1])):.2e})")


if __name__ == "__main__":
    main()
