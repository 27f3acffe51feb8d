"""CPU emulation of the algebra of k_lanczos_fstep (csrc/kk_kernels_fstep.hip): the whole Lanczos / Arnoldi step of a short vector in one launch.
Emulated in the kernel's own terms -- rows partitioned into blocks, every inner product formed as per-block partials added in block order (the grid
reductions), ONE reduction of [alpha0 | V'w | V'v] (Lanczos) / [V'w | V'v] then [V'w] (Arnoldi, two passes), rhs = p - alpha0 g, the exact
low-synchronisation solve (I + L) s = rhs with the strictly-lower Gram rows L of the basis (row m - 1 = g, just measured), alpha0 folded into the last
coefficient, w -= V coef, the norm, the normalised commit -- against the oracle's sequential recurrences (src/factorizations/lanczos.jl:297-336,
arnoldi.jl:199-245, orthonormal.jl:378-452) over whole factorizations: alpha / beta / H to 1e-10, the basis orthonormal.  The Gram rows are carried from
step to step exactly as the kernel carries them (device mirror row m - 1 written by the step that measured it)."""
import numpy as np
import pytest

import krylov_oracle as ko


def blocks_dot(a, b, nblk):
    """sum of per-block partials in block order (what the granule reduction computes)"""
    n = a.shape[0]
    edges = np.linspace(0, n, nblk + 1).astype(int)
    return float(sum(float(a[lo:hi] @ b[lo:hi]) for lo, hi in zip(edges[:-1], edges[1:])))


def lowsync_solve(rhs, L, g, m):
    """(I + L) s = rhs, column-oriented forward substitution as the kernel runs it; row i of L for i < m - 1, g for i = m - 1"""
    s = rhs.copy()
    for j in range(m - 1):
        sj = s[j]
        for i in range(j + 1, m):
            lij = g[j] if i == m - 1 else L[i, j]
            s[i] -= lij * sj
    return s


def fstep_lanczos(A, V, L, beta_prev, lowsync, cgs_order, nblk):
    """one launch: V = [v_0 .. v_(m-1)] normalised columns; returns (alpha, beta, v_next, g)"""
    m = len(V)
    v, vprev = V[-1], V[-2] if m >= 2 else None
    w = A @ v
    a0 = blocks_dot(v, w, nblk) if cgs_order else None
    if vprev is not None:
        w = w - beta_prev * vprev
    if not cgs_order:
        a0 = blocks_dot(v, w, nblk)
    p = np.array([blocks_dot(q, w, nblk) for q in V])
    g = np.array([blocks_dot(q, v, nblk) for q in V])
    rhs = p - a0 * g
    s = lowsync_solve(rhs, L, g, m) if lowsync else rhs
    if lowsync:
        L[m - 1, : m - 1] = g[: m - 1]
    s_last = s[m - 1]
    coef = s.copy(); coef[m - 1] += a0
    for q, cf in zip(V, coef):
        w = w - cf * q
    beta = np.sqrt(blocks_dot(w, w, nblk))
    return a0 + s_last, beta, w / beta, g


def fstep_arnoldi(A, V, L, lowsync, npass, nblk):
    m = len(V)
    v = V[-1]
    w = A @ v
    h = np.zeros(m)
    g = None
    for ps in range(npass):
        p = np.array([blocks_dot(q, w, nblk) for q in V])
        if ps == 0:
            g = np.array([blocks_dot(q, v, nblk) for q in V])
            if lowsync:
                L[m - 1, : m - 1] = g[: m - 1]
        s = lowsync_solve(p, L, g, m) if lowsync else p
        h += s
        for q, cf in zip(V, s):
            w = w - cf * q
    beta = np.sqrt(blocks_dot(w, w, nblk))
    return h, beta, w / beta


@pytest.mark.parametrize("orth,lowsync,cgs_order", [("mgs2", True, False), ("cgs2", False, True)])
@pytest.mark.parametrize("nblk", [1, 7, 64])
def test_lanczos_step_algebra_matches_the_sequential_recurrence(orth, lowsync, cgs_order, nblk):
    nx, ny, steps = 40, 33, 25
    n = nx * ny
    A = ko.laplacian_2d(nx, ny, shift_diag=10 * np.linspace(0, 1, n) ** 2)
    x0 = np.random.default_rng(3).random(n)
    ref = ko.MGS2 if orth == "mgs2" else ko.CGS2
    oit = ko.LanczosIterator(A, x0.copy(), ref); of = ko.lanczos_initialize(oit)
    # initialize as the library does it (two host-side steps), then the fused steps
    V = [of.V[0].copy()]
    alphas, betas = [of.alphas[0]], [of.betas[0]]
    r = of.r.copy()
    L = np.zeros((steps + 3, steps + 3))
    for _ in range(steps):
        of = ko.lanczos_expand(oit, of)
    for k in range(1, steps + 1):
        V.append(r / betas[-1])
        a, b, vn, _ = fstep_lanczos(A, V, L, betas[-1], lowsync, cgs_order, nblk)
        alphas.append(a); betas.append(b)
        r = vn * b
    rel = lambda x, y: float(np.max(np.abs(np.array(x) - np.array(y)) / np.abs(np.array(y))))
    assert rel(alphas, of.alphas) < 1e-10 and rel(betas, of.betas) < 1e-10
    Vm = np.stack(V, 1)
    assert np.max(np.abs(Vm.T @ Vm - np.eye(Vm.shape[1]))) < 1e-12


@pytest.mark.parametrize("orth,lowsync,npass", [("mgs2", True, 2), ("cgs2", False, 2), ("mgs", True, 1), ("cgs", False, 1)])
def test_arnoldi_step_algebra_matches_the_sequential_recurrence(orth, lowsync, npass):
    nx, ny, steps, nblk = 36, 30, 20, 13
    n = nx * ny
    A = ko.convection_diffusion_2d(nx, ny)
    x0 = np.random.default_rng(3).random(n)
    ref = {"mgs2": ko.MGS2, "cgs2": ko.CGS2, "mgs": ko.MGS, "cgs": ko.CGS}[orth]
    oit = ko.ArnoldiIterator(A, x0.copy(), ref); of = ko.arnoldi_initialize(oit)
    V = [np.asarray(of.V[0], float).copy()]
    r, beta = np.asarray(of.r, float).copy(), float(of.normres)
    Hp = [float(of.H[0]), beta]          # packed Hessenberg as the factorization stores it: per column h[0 .. m - 1], then beta (arnoldi.jl:31-50)
    L = np.zeros((steps + 3, steps + 3))
    for _ in range(steps):
        of = ko.arnoldi_expand(oit, of)
    for k in range(1, steps + 1):
        V.append(r / beta)
        h, beta, vn = fstep_arnoldi(A, V, L, lowsync, npass, nblk)
        Hp.extend(float(x) for x in h); Hp.append(beta)
        r = vn * beta
    Ho = np.asarray(of.H, float)
    Hp = np.asarray(Hp, float)
    tol = 1e-10 if npass == 2 else 1e-6          # one-pass orthogonalisers: the two orders differ by their loss of orthogonality
    assert Hp.shape == Ho.shape and np.max(np.abs(Hp - Ho)) < tol * np.max(np.abs(Ho))
    assert abs(beta - of.normres) < tol * abs(of.normres)
    if npass == 2:
        Vm = np.stack(V, 1)
        assert np.max(np.abs(Vm.T @ Vm - np.eye(Vm.shape[1]))) < 1e-12
