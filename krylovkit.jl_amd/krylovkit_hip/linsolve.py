"""linsolve family (src/linsolve/): GMRES on the Arnoldi factorization, CG and BiCGStab on device vectors with fused
iteration bodies -- same host control flow as the reference."""
from __future__ import annotations

import math
from typing import List, Optional

import numpy as np

from . import dense
from .algorithms import *  # noqa: F401,F403  (algorithm structs + ConvergenceInfo)
from .algorithms import ConvergenceInfo
from .core import DeviceBasis, HipVec, KrylovDefaults, Orthogonalizer, SparseOperator
from .factorizations import (ArnoldiIterator, GKLIterator, LanczosIterator, _as_operator, expand_, initialize,
                             initialize_, shrink_)


# -------------------------------------------------------------------- linsolve (GMRES)
def linsolve(A, b, x0=None, alg: Optional[GMRES] = None, a0: float = 0.0, a1: float = 1.0, *, atol: Optional[float] = None,
             rtol: Optional[float] = None, tol: Optional[float] = None, return_device: bool = False,
             trace: Optional[list] = None, **kw):
    """linsolve(operator, b, x0, alg::GMRES, a0, a1) (src/linsolve/gmres.jl:1-151).
    Without an algorithm struct the keyword front-end of the reference applies (`linselector`,
    linsolve/linsolve.jl:123-151): atol and rtol both default to KrylovDefaults.tol and
    `tol = max(atol, rtol*norm(b))`; GMRES is selected.  With an explicit `alg` its own `tol` is used, exactly as
    `linsolve(f, b, x0, alg, a0, a1)` does; atol / rtol / tol then make no sense and raise.
    Like the reference's method table, `alg::CG` and `alg::BiCGStab` select those solvers (linsolve/cg.jl, bicgstab.jl).
    `trace` (optional list) receives (numiter, k, beta) after every inner step (the residual estimates of gmres.jl:53,94)."""
    if alg is not None and (atol is not None or rtol is not None or tol is not None):
        raise TypeError("linsolve: atol / rtol / tol belong to the keyword front-end; with an explicit algorithm struct "
                        "the tolerance is alg.tol (linsolve/linsolve.jl:112-121)")
    if isinstance(alg, CG):
        return linsolve_cg(A, b, x0, alg, a0, a1)
    if isinstance(alg, BiCGStab):
        return linsolve_bicgstab(A, b, x0, alg, a0, a1)
    op = _as_operator(A)
    ctx = op.ctx
    n = op.shape[0]
    b = np.asarray(b, dtype=np.float64)
    # work vectors: 0 = b, 1 = x, 2 = r, 3 = tmp
    W = DeviceBasis(n, 4, ctx)
    vb, vx, vr, vt = HipVec(W, 0), HipVec(W, 1), HipVec(W, 2), HipVec(W, 3)
    vb.set(b)
    if alg is None:
        if tol is None:   # tol::Real = max(atol, rtol * norm(b))   linsolve.jl:131-133
            atol = KrylovDefaults.tol if atol is None else atol
            rtol = KrylovDefaults.tol if rtol is None else rtol
            # norm(b) on the device: under a communicator `b` is this rank's block and the norm is all-reduced, so every
            # rank forms the same tolerance (a host norm of the local block would let the ranks stop at different steps)
            tol = max(atol, rtol * vb.norm())
        alg = GMRES(kw.get("orth", KrylovDefaults.orth), kw.get("maxiter", KrylovDefaults.maxiter),
                    kw.get("krylovdim", KrylovDefaults.krylovdim), tol)
    krylovdim, maxiter, tol = alg.krylovdim, alg.maxiter, alg.tol
    if x0 is None:
        vx.zero_()
    else:
        vx.set(np.asarray(x0, dtype=np.float64))
    # r = b - a0 x0 - a1 A x0   :3-12
    op.apply(vx, vt)
    vr.scale_from_(vb, 1.0)
    if a0 != 0:
        vr.add_(vx, -a0)
    vr.add_(vt, -a1)
    beta = vr.norm()
    if beta < tol:
        x = W if return_device else vx.get()
        return x, ConvergenceInfo(1, vr.get(), beta, 0, 1)
    y = np.zeros(krylovdim + 1)
    gs: List[Optional[tuple]] = [None] * krylovdim
    R = np.zeros((krylovdim, krylovdim))
    numiter = 0
    numops = 1
    it = ArnoldiIterator(op, vr, alg.orth, capacity=krylovdim + 2)
    fact = initialize(it)
    numops += 1
    while True:
        numiter += 1
        y[0] = beta
        k = 1
        H = fact.rayleighquotient()
        R[0, 0] = a0 + a1 * H[0, 0]
        c, s, R[0, 0] = dense.givens(R[0, 0], a1 * fact.normres)
        gs[0] = (0, 1, c, s)
        y[1] = 0.0
        y[0], y[1] = c * y[0] + s * y[1], -s * y[0] + c * y[1]
        beta = abs(y[1])
        if trace is not None:
            trace.append((numiter, 1, beta))
        while R[k - 1, k - 1] != 0 and beta > tol and len(fact) < krylovdim:  # :55
            fact = expand_(it, fact)
            numops += 1
            k = len(fact)
            # new Hessenberg column straight from the packed storage
            base = ((k * k + k - 2) >> 1)
            hcol = fact.H[base: base + k]
            for i in range(k - 1):
                R[i, k - 1] = a1 * hcol[i]
            R[k - 1, k - 1] = a0 + a1 * hcol[k - 1]
            Rk = R[:, k - 1]
            for i in range(k - 1):  # :72-75
                i1, i2, c, s = gs[i]
                Rk[i1], Rk[i2] = c * Rk[i1] + s * Rk[i2], -s * Rk[i1] + c * Rk[i2]
            if math.hypot(R[k - 1, k - 1], a1 * fact.normres) < tol:  # :78-85
                c, s, y[k] = dense.givens(0.0, y[k - 1])
                gs[k - 1] = (k, k - 1, c, s)
                y[k - 1] = 0.0
                R[k - 1, k - 1] = 0.0
            else:
                c, s, R[k - 1, k - 1] = dense.givens(R[k - 1, k - 1], a1 * fact.normres)
                gs[k - 1] = (k - 1, k, c, s)
                y[k] = 0.0
                y[k - 1], y[k] = c * y[k - 1] + s * y[k], -s * y[k - 1] + c * y[k]
            beta = abs(y[k])
            if trace is not None:
                trace.append((numiter, k, beta))
        kk = k - 1 if (R[k - 1, k - 1] == 0 and y[k - 1] == 0) else k  # :98-102
        dense.ldiv_upper(R, y, kk)
        V = fact.basis()
        V.unproject(vx, y[:k], 0, k, 1.0, 1.0)  # x += sum V[i] y[i]   :105-108
        if beta > tol and numiter < maxiter:  # :110-117
            fact.r.scale_(1.0 / fact.normres)  # push!(V, scale!!(w, 1/normres))
            V.length = k + 1
            # rmul!(V, gs[i]') for i = 1..k, then r = y[k+1] V[k+1] (gmres.jl:113-117): only the rotated V[k+1] is used
            # afterwards (the slab is re-initialised), so the k rotations are accumulated on the (k+1)-square host
            # identity and applied as ONE pass over the basis instead of k two-vector launches
            Zr = np.eye(k + 1)
            for i in range(k):
                i1, i2, c, s = gs[i]
                z1, z2 = Zr[:, i1].copy(), Zr[:, i2].copy()
                Zr[:, i1], Zr[:, i2] = c * z1 + s * z2, -s * z1 + c * z2
            V.unproject(vr, Zr[:, k], 0, k + 1, y[k], 0.0)
            V.length = k
        else:  # :119-132
            vr.scale_from_(vb, 1.0)
            op.apply_affine(vx, vt, a0, a1)
            vr.add_(vt, -1.0)
            numops += 1
            beta = vr.norm()
            if beta < tol:
                x = W if return_device else vx.get()
                return x, ConvergenceInfo(1, vr.get(), beta, numiter, numops)
        if numiter >= maxiter:
            x = W if return_device else vx.get()
            return x, ConvergenceInfo(0, vr.get(), beta, numiter, numops)
        it = ArnoldiIterator(op, vr, alg.orth, capacity=krylovdim + 2)  # :147-148
        fact = initialize_(it, fact)


# -------------------------------------------------------------------- linsolve (CG)


def linsolve_cg(A, b, x0=None, alg: Optional[CG] = None, a0: float = 0.0, a1: float = 1.0, **kw):
    """linsolve(operator, b, x0, alg::CG, a0, a1) (src/linsolve/cg.jl:1-103) for a symmetric positive
    definite a0 + a1*A.  Per iteration: one SpMV with the fused <p, q>, one fused update
    (x += alpha p; r -= alpha q; |r|), one axpby (p = r + beta p)."""
    import ctypes as C
    from ._lib import check
    op = _as_operator(A)
    n = op.shape[0]
    alg = alg or CG(**kw)
    maxiter, tol = alg.maxiter, alg.tol
    W = DeviceBasis(n, 5, op.ctx)  # 0 = b, 1 = x, 2 = r, 3 = p, 4 = q
    vb, vx, vr, vp, vq = (HipVec(W, i) for i in range(5))
    lib = W._lib
    vb.set(np.asarray(b, dtype=np.float64))
    if x0 is None:
        vx.zero_()
    else:
        vx.set(np.asarray(x0, dtype=np.float64))
    op.apply(vx, vq)                      # y0 = apply(operator, x0)   :3
    vr.scale_from_(vb, 1.0)
    if a0 != 0:
        vr.add_(vx, -a0)
    vr.add_(vq, -a1)
    normr = vr.norm()
    numops, numiter = 1, 0
    if normr < tol:
        return vx.get(), ConvergenceInfo(1, vr.get(), normr, numiter, numops)

    def iterate(beta, first, rho):
        """[p = r + beta p]; q = (a0 + a1 A) p; alpha = rho/<p,q>; x += alpha p; r -= alpha q -> |r|  (one host sync)"""
        pq, nr = C.c_double(), C.c_double()
        check(lib.kk_cg_iterate(op.handle, W.handle, 1, 2, 3, 4, a0, a1, beta, int(first), rho, C.byref(pq), C.byref(nr)))
        return nr.value

    rho = normr ** 2
    vp.scale_from_(vr, 1.0)               # :33-34
    normr = iterate(0.0, True, rho)
    rho_old, rho = rho, normr ** 2
    beta = rho / rho_old
    numops += 1
    numiter += 1
    if normr < tol:
        return vx.get(), ConvergenceInfo(1, vr.get(), normr, numiter, numops)
    while True:                           # :60-101
        normr = iterate(beta, False, rho)  # p = add!!(p, r, 1, beta); q = apply; alpha = rho/inner(p,q); x, r updates
        if normr < tol:                   # recompute explicitly   :67-72
            vr.scale_from_(vb, 1.0)
            op.apply_affine(vx, vq, a0, a1)
            vr.add_(vq, -1.0)
            normr = vr.norm()
            rho = normr ** 2
            beta = 0.0
        else:
            rho_old, rho = rho, normr ** 2
            beta = rho / rho_old
        numops += 1
        numiter += 1
        if normr < tol:
            return vx.get(), ConvergenceInfo(1, vr.get(), normr, numiter, numops)
        if numiter >= maxiter:
            return vx.get(), ConvergenceInfo(0, vr.get(), normr, numiter, numops)


# -------------------------------------------------------------------- linsolve (BiCGStab)


def linsolve_bicgstab(A, b, x0=None, alg: Optional[BiCGStab] = None, a0: float = 0.0, a1: float = 1.0, **kw):
    """linsolve(operator, b, x0, alg::BiCGStab, a0, a1) (src/linsolve/bicgstab.jl:1-203) for a general a0 + a1*A.
    Per iteration: two SpMVs (each with its inner products fused) and three fused vector kernels
    (kk_bicgstab_half / kk_bicgstab_full); rho, sigma, alpha, omega stay on the device, the host reads the two
    norms the reference compares with tol."""
    import ctypes as C
    from ._lib import check
    op = _as_operator(A)
    n = op.shape[0]
    alg = alg or BiCGStab(**kw)
    maxiter, tol = alg.maxiter, alg.tol
    # 0 = b, 1 = x, 2 = r, 3 = r_shadow, 4/9 = p (double buffer), 5/10 = v (double buffer), 6 = s, 7 = t, 8 = xhalf
    W = DeviceBasis(n, 11, op.ctx)
    vb, vx, vr, vrs, _, _, vs, vt, vh, _, _ = (HipVec(W, i) for i in range(11))
    lib = W._lib
    cur, alt = (4, 5), (9, 10)

    def colarr(pv, prev):
        return (C.c_int * 9)(1, 2, 3, pv[0], pv[1], 6, 7, prev[0], prev[1])

    vb.set(np.asarray(b, dtype=np.float64))
    if x0 is None:
        vx.zero_()
    else:
        vx.set(np.asarray(x0, dtype=np.float64))
    op.apply(vx, vt)                      # y0 = apply(operator, x0)   :3
    vr.scale_from_(vb, 1.0)
    if a0 != 0:
        vr.add_(vx, -a0)
    vr.add_(vt, -a1)
    normr = vr.norm()
    numops, numiter = 1, 0
    if normr < tol:                       # :22-28
        return vx.get(), ConvergenceInfo(1, vr.get(), normr, numiter, numops)
    numiter += 1
    vrs.scale_from_(vr, 1.0)              # shadow residual   :35
    rho = vrs.inner(vr)
    if rho == 0.0:                        # `rho ≈ 0.0` with isapprox's atol = 0: only an exact zero   :39-46
        return vx.get(), ConvergenceInfo(0, vr.get(), normr, numiter, numops)
    HipVec(W, cur[0]).scale_from_(vr, 1.0)   # p = r
    first = True
    mode = 1       # 1: first iteration; 0: rho on the device; 2: rho handed over again; 3: half already enqueued
    cols = colarr(cur, cur)
    snorm, alpha, rnorm, rho_c, omega = (C.c_double() for _ in range(5))
    while True:
        if not first:
            numiter += 1
        # BiCG half: p update, v = A p, alpha, s = r - alpha v (and, run ahead of the host, t = A s)
        check(lib.kk_bicgstab_half(op.handle, W.handle, cols, a0, a1, mode, rho, C.byref(snorm), C.byref(alpha)))
        numops += 1
        normr = snorm.value
        redo_t = 0
        if normr < tol:                   # explicit residual at the half step   :65-80 / :142-157
            vh.scale_from_(vx, 1.0)
            vh.add_(HipVec(W, cols[3]), alpha.value)      # xhalf = x + alpha p
            op.apply_affine(vh, vt, a0, a1)
            vs.scale_from_(vb, 1.0)
            vs.add_(vt, -1.0)
            numops += 1
            normr_act = vs.norm()
            if normr_act < tol:
                return vh.get(), ConvergenceInfo(1, vs.get(), normr_act, numiter, numops)
            redo_t = 1                    # s was replaced: t = A s has to be recomputed
        numops += 1                       # t = apply(operator, s, a0, a1)   :83 / :163
        last = (not first) and numiter >= maxiter
        nxt = None if last else colarr(alt, cur)     # next half into the other p/v buffers, reading the current ones
        check(lib.kk_bicgstab_full(op.handle, W.handle, cols, a0, a1, redo_t, nxt, C.byref(rnorm), C.byref(rho_c),
                                   C.byref(omega)))
        normr = rnorm.value
        rho = rho_c.value
        mode = 3
        if normr < tol:                   # explicit residual at the full step   :94-110 / :175-190
            op.apply_affine(vx, vt, a0, a1)
            vr.scale_from_(vb, 1.0)
            vr.add_(vt, -1.0)
            numops += 1
            normr_act = vr.norm()
            if normr_act < tol:
                return vx.get(), ConvergenceInfo(1, vr.get(), normr_act, numiter, numops)
            rho = vrs.inner(vr)           # r was replaced: the next rho = <r_shadow, r> is that of the NEW r   :120
            mode = 2                      # ... and the run-ahead half (old r, old rho) is discarded and redone
        if last:                          # :191-198
            return vx.get(), ConvergenceInfo(0, vr.get(), normr, numiter, numops)
        first = False
        cols = nxt
        cur, alt = alt, cur
