"""exponentiate / expintegrator on device vectors (src/matrixfun/exponentiate.jl:83-84,
src/matrixfun/expintegrator.jl:93-323).  The Krylov factorisation is built by the same device `initialize` /
`expand!` / `initialize!` as eigsolve; the (K+p+1)-square matrix exponential is host work (scipy, the role
LinearAlgebra.exp plays in the reference); the time-step assembly is kk_unproject + axpys on device columns.
Real time steps only: device vectors are Float64."""
from __future__ import annotations

import math
from typing import Optional, Sequence, Union

import numpy as np
import scipy.linalg as sla

from .core import DeviceBasis, HipVec
from .factorizations import ArnoldiIterator, LanczosIterator, _as_operator, expand_, initialize, initialize_
from .algorithms import Arnoldi, ConvergenceInfo, Lanczos


def exponentiate(A, t: float, v, alg: Optional[Union[Lanczos, Arnoldi]] = None, **kw):
    """exponentiate(A, t, v, alg) = expintegrator(A, t, (v,), alg)   (matrixfun/exponentiate.jl:83-84)"""
    return expintegrator(A, t, (v,), alg, **kw)


def expintegrator(A, t: float, u: Sequence[np.ndarray], alg: Optional[Union[Lanczos, Arnoldi]] = None, **kw):
    """expintegrator(A, t, u::Tuple, alg::Union{Lanczos,Arnoldi}) (matrixfun/expintegrator.jl:100-323):
    w = exp(tA) u0 + sum_j t^j phi_j(tA) u_j.  Returns (w, ConvergenceInfo)."""
    op = _as_operator(A)
    if alg is None:
        alg = (Lanczos if op.symmetric else Arnoldi)(**kw)
    if isinstance(t, complex):
        raise TypeError("expintegrator on device vectors supports real time steps only (Float64 vectors)")
    n = op.shape[0]
    u = [np.asarray(z, dtype=np.float64) for z in u]
    if len(u) == 1:                                            # :101
        u = [u[0], np.zeros_like(u[0])]
    p = len(u) - 1
    maxiter, krylovdim = alg.maxiter, alg.krylovdim
    assert maxiter >= 1
    W = DeviceBasis(n, 2 * (p + 1), op.ctx)                    # columns 0..p = w[0..p], p+1..2p+1 = u[0..p]
    w = [HipVec(W, j) for j in range(p + 1)]
    ud = [HipVec(W, p + 1 + j).set(u[j]) for j in range(p + 1)]
    w0 = w[0]
    w0.scale_from_(ud[0], 1.0)
    op.apply(ud[0], w[1])                                      # Au0, reused as w[2] of the reference   :107,143
    numops = 1
    eta = alg.tol                                              # :120-134
    totalerr = 0.0
    sgn = float(np.sign(t))
    tau = abs(t)
    if math.isfinite(tau):
        dtau, dtaumin, maxerr = tau, tau / maxiter, tau * eta
    else:
        dtau, dtaumin, maxerr = 1.0, 0.0, eta
    gamma = 0.8
    tau0 = 0.0

    def stage_vectors(first: bool):
        nonlocal numops
        for j in range(1, p + 1):                              # :146-158 / :293-301
            if j > 1 or not first:
                op.apply(w[j - 1], w[j])
                numops += 1
            lfac = 1
            for l in range(0, p - j + 1):
                w[j].add_(ud[j + l], (sgn * tau0) ** l / lfac)
                lfac *= l + 1

    def small_exp(fact, step):
        Kc = len(fact)
        H = np.zeros((Kc + p + 1, Kc + p + 1))
        rq = fact.rayleighquotient()
        if isinstance(rq, tuple):
            d, e = rq
            rq = np.diag(d) + np.diag(e, 1) + np.diag(e, -1)
        H[:Kc, :Kc] = rq * (sgn * step)
        H[0, Kc] = 1.0
        for i in range(1, p + 1):
            H[Kc + i - 1, Kc + i] = 1.0
        return sla.expm(H)

    def take_step(fact, expH, step):
        Kc = len(fact)
        jfac = 1
        for j in range(1, p):
            w0.add_(w[j], (sgn * step) ** j / jfac)
            jfac *= j + 1
        fact.basis().times(np.ascontiguousarray(expH[:Kc, Kc + p - 1]), w[p], 0, Kc)   # unproject!!   :233 / :263
        w[p].add_(fact.r, expH[Kc - 1, Kc + p])                # first correction
        w0.add_(w[p], beta * (sgn * step) ** p)

    def done(conv, err, numiter):
        return w0.get(), ConvergenceInfo(conv, None, err, numiter, numops)

    stage_vectors(True)
    beta = w[p].norm()
    if beta < eta and p == 1:                                  # :161-166
        return done(1, beta, 0)
    mk_iter = LanczosIterator if isinstance(alg, Lanczos) else ArnoldiIterator
    it = mk_iter(op, w[p], alg.orth, capacity=krylovdim + 2)
    fact = initialize(it)
    numops += 1
    numiter = 1
    while True:
        Kc = len(fact)
        if Kc == krylovdim:                                    # :184-241
            if numiter < maxiter:
                dtau = min(dtau, tau - tau0)
                if math.isfinite(tau):
                    dtaumin = (tau - tau0) / (maxiter - numiter + 1)
            else:
                dtau = tau - tau0
            expH = small_exp(fact, dtau)
            eps_ = abs(dtau ** p * beta * fact.normres * expH[Kc - 1, Kc + p])
            omega = eps_ / (dtau * eta)
            q = Kc / 2
            while numiter < maxiter and omega >= 1.0 and dtau > dtaumin:
                eps_prev, dtau_prev = eps_, dtau
                dtau = max(dtau * (gamma / omega) ** (1 / (q + 1)), dtaumin)
                expH = small_exp(fact, dtau)
                eps_ = abs(dtau ** p * beta * fact.normres * expH[Kc - 1, Kc + p])
                omega = eps_ / (dtau * eta)
                q = max(0.0, math.log(eps_ / eps_prev) / math.log(dtau / dtau_prev) - 1)
            tau0 = tau0 + dtau if numiter < maxiter else tau
            totalerr += eps_
            take_step(fact, expH, dtau)
            if omega < gamma:
                dtau *= (gamma / omega) ** (1 / (q + 1))
        elif fact.normres <= (tau - tau0) * eta or getattr(alg, "eager", False):   # :242-268
            step = tau - tau0
            expH = small_exp(fact, step)
            eps_ = abs(step ** p * beta * fact.normres * expH[Kc - 1, Kc + p])
            omega = eps_ / (step * eta)
            if omega < 1.0:
                totalerr += eps_
                take_step(fact, expH, step)
                tau0 = tau
        if tau0 >= tau:                                        # :269-285
            return done(1 if totalerr <= maxerr else 0, totalerr, numiter)
        if Kc < krylovdim:
            fact = expand_(it, fact)
            numops += 1
        else:
            stage_vectors(False)
            beta = w[p].norm()
            if beta < eta and p == 1:                          # :302-307
                return done(1, beta, numiter)
            it = mk_iter(op, w[p], alg.orth, capacity=krylovdim + 2)
            fact = initialize_(it, fact)
            numops += 1
            numiter += 1
