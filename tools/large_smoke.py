"""Size sanity of the widened drivers at 2M rows (every N-length object on the device, small Krylov dimensions):
Arnoldi eigsolve, bieigsolve, geneigsolve, expintegrator, LSMR.  Prints residual checks.  usage: python tools/large_smoke.py"""
import sys
import time
from pathlib import Path

import numpy as np
import scipy.sparse as sp

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "krylovkit.jl_amd"))
sys.path.insert(0, str(ROOT))
import krylovkit_hip as kk  # noqa: E402
from bench import laplacian_rows  # noqa: E402
from tools.bench_configs import convdiff  # noqa: E402
sys.path.insert(0, str(ROOT / "tests"))
import hostmirror_extras as hx  # noqa: E402  (host drivers outside SURVEY section 8: test infrastructure)

nx, ny = 2000, 1000
n = nx * ny
ctx = kk.default_context()
L = laplacian_rows(nx, ny, 0, ny).tocsr()
Cm = convdiff(nx, ny)
rng = np.random.default_rng(0)
x0, w0 = rng.random(n), rng.random(n)
t0 = time.time()
opL, opC = kk.SparseOperator(L, ctx, symmetric=True), kk.SparseOperator(Cm, ctx)
vals, vecs, info = hx.eigsolve_arnoldi(opC, x0, 2, "LR", kk.Arnoldi(kk.ModifiedGramSchmidt2(), 20, 3, 1e-6))
print("arnoldi", vals[:2], info.numiter, info.numops, [float(np.linalg.norm(Cm @ v - l * v) - r) < 1e-8 for l, v, r in zip(vals, vecs, info.normres)], round(time.time() - t0, 1), flush=True)
vals, (VR, WL), (iV, iW) = hx.bieigsolve(opC, x0, w0, 1, "LR", hx.BiArnoldi(kk.ModifiedGramSchmidt2(), 16, 2, 1e-6))
print("biarnoldi", vals[:1], iV.numiter, iV.numops, float(np.linalg.norm(Cm @ VR[0] - vals[0] * VR[0] - iV.residual[0])) < 1e-8, round(time.time() - t0, 1), flush=True)
B = (sp.identity(n, format="csr") + 0.05 * L).tocsr()
opB = kk.SparseOperator(B, ctx, symmetric=True)
vals, vecs, info = hx.geneigsolve((opL, opB), x0, 1, "SR", hx.GolubYe(kk.ModifiedGramSchmidt2(), 12, 2, 1e-6))
print("golubye", vals, info.numiter, info.numops, float(np.linalg.norm(L @ vecs[0] - vals[0] * (B @ vecs[0]) - info.residual[0])) < 1e-8, round(time.time() - t0, 1), flush=True)
w, info = kk.exponentiate(kk.SparseOperator((-0.25 * L).tocsr(), ctx, symmetric=True), 0.3, x0, kk.Lanczos(kk.ModifiedGramSchmidt2(), 20, 20, 1e-9))
print("expm", info.converged, info.numops, float(np.linalg.norm(w)), round(time.time() - t0, 1), flush=True)
m2 = 500_000
R = sp.csr_matrix((rng.standard_normal(5 * n), (np.repeat(np.arange(n), 5), rng.integers(0, m2, 5 * n))), shape=(n, m2))
opR = kk.SparseOperator(R, ctx)
b = rng.random(n)
x, info = kk.lssolve(opR, b, kk.LSMR(kk.ModifiedGramSchmidt(), 60, 8, 1e-6 * np.linalg.norm(b)))
r = b - R @ x
print("lsmr", opR.info()["format"], info.converged, info.numiter, float(np.linalg.norm(R.T @ r)), float(info.normres), round(time.time() - t0, 1), flush=True)
