"""lssolve (src/lssolve/lsmr.jl): LSMR on device vectors."""
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


# -------------------------------------------------------------------- lssolve (LSMR)


def lssolve(A, b, alg: Optional[LSMR] = None, lam: float = 0.0, *, rtol: float = KrylovDefaults.tol,
            atol: float = KrylovDefaults.tol, **kw):
    """lssolve(operator, b, alg::LSMR, lambda) (src/lssolve/lsmr.jl:1-151; defaults lssolve.jl:101-110): minimise
    |A x - b|^2 + |lambda x|^2.  Device data: (u, r, Ah, Ahbar, Av) in the row space of A, the circular buffer of the
    `krylovdim` most recent v's plus (x, h, hbar) in its domain.  Per iteration: A v, A' u, one re-orthogonalisation
    against the buffer, and three fused vector updates (kk_lsmr_step_u, 2 x kk_lsmr_update)."""
    import ctypes as C
    from ._lib import check
    op = _as_operator(A)
    nu, nv = op.shape
    b = np.asarray(b, dtype=np.float64)
    BU = DeviceBasis(nu, 5, op.ctx)            # 0 = u, 1 = r, 2 = Ah, 3 = Ahbar, 4 = Av
    if alg is None:
        # norm(b) on the device (all-reduced under a communicator: identical tolerance on every rank)
        kw.setdefault("tol", max(atol, rtol * HipVec(BU, 0).set(b).norm()))
        alg = LSMR(**kw)
    K, maxiter, tol = alg.krylovdim, alg.maxiter, alg.tol
    BV = DeviceBasis(nv, K + 5, op.ctx)        # 0..K-1 = V (circular), K / K+4 = work vectors, K+1 = x, K+2 = h, K+3 = hbar
    lib = BU._lib
    u, r, Ah, Ahbar, Av = (HipVec(BU, i) for i in range(5))
    w, x, h, hbar = (HipVec(BV, K + i) for i in range(4))
    u.set(b)
    op.apply(u, w, transpose=True)             # v0 = apply_adjoint(operator, u0)   :4
    beta = u.norm()
    u.scale_(1 / beta)
    w.scale_(1 / beta)
    alpha = w.norm()
    v = HipVec(BV, 0).scale_from_(w, 1 / alpha)
    nV = 1                                     # length(V)
    alphabar, zetabar, rho, theta, rhobar, cbar, sbar = alpha, alpha * beta, 1.0, 0.0, 1.0, 1.0, 0.0
    abszetabar = abs(zetabar)
    x.zero_(); hbar.zero_(); Ah.zero_(); Ahbar.zero_()
    h.scale_from_(v, 1.0)
    r.scale_from_(u, beta)
    numiter, numops = 0, 1

    def result(conv):
        return x.get(), ConvergenceInfo(conv, r.get(), abszetabar, numiter, numops)

    if abszetabar < tol:                       # :48-58
        return result(1)
    bt = C.c_double()
    while True:
        numiter += 1
        op.apply(v, Av)                        # :63
        numops += 1
        # Ah = Av - (theta/rho) Ah ; u = Av - alpha u ; beta = |u|     :64-68
        check(lib.kk_lsmr_step_u(BU.handle, 4, 2, 0, theta / rho, alpha, C.byref(bt)))
        beta = bt.value
        if beta > tol:
            u.scale_(1 / beta)
            # v_new = A' u - beta v   :73  (after an alpha <= tol step v itself lives in a work column: use the other)
            w = HipVec(BV, K + 4) if v.col == K else HipVec(BV, K)
            op.apply(u, w, transpose=True)
            w.add_(v, -beta)
            numops += 1
            if K > 1:
                _, alpha, _ = BV.orthogonalize(w, alg.orth, 0, nV)      # :76-78 (+ the norm of :80 fused)
            else:
                alpha = w.norm()
            if alpha > tol:
                slot = nV if numiter < K else (numiter % K)             # mod1(numiter + 1, K) - 1
                v = HipVec(BV, slot).scale_from_(w, 1 / alpha)
                if numiter < K:
                    nV += 1
            else:
                v = w
        alphahat = float(np.hypot(alphabar, lam))      # :92-94
        rhoold = rho                                   # :97-102
        rho = float(np.hypot(alphahat, beta))
        c = alphahat / rho
        s_ = beta / rho
        theta = s_ * alpha
        alphabar = c * alpha
        rhobarold = rhobar                             # :105-112
        thetabar = sbar * rho
        cbarrho = cbar * rho
        rhobar = float(np.hypot(cbarrho, theta))
        cbar = cbarrho / rhobar
        sbar = theta / rhobar
        zeta = cbar * zetabar
        zetabar = -sbar * zetabar
        c1 = thetabar * rho / (rhoold * rhobarold)
        c2 = zeta / (rho * rhobar)
        # hbar = h - c1 hbar ; x += c2 hbar ; h = v - (theta/rho) h      :115-121
        check(lib.kk_lsmr_update(BV.handle, K + 2, K + 3, K + 1, BV.handle, v.col, c1, c2, theta / rho))
        # Ahbar = Ah - c1 Ahbar ; r -= c2 Ahbar                           :116,119
        check(lib.kk_lsmr_update(BU.handle, 2, 3, 1, None, -1, c1, -c2, 0.0))
        abszetabar = abs(zetabar)
        if abszetabar <= tol:
            return result(1)
        if numiter >= maxiter:
            return result(0)
