"""Thick-restart data path timing (basistransform) and a real multi-restart eigsolve."""
import sys, time
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "krylovkit.jl_amd")); sys.path.insert(0, str(ROOT))
import krylovkit_hip as kk
from bench import laplacian_rows
ctx = kk.default_context()
N = 10_000_000
m, n = 100, 60
B = kk.DeviceBasis(N, m + 2, ctx)
for j in range(m):
    B[j].rand_(j + 1)
B.length = m
U, _ = np.linalg.qr(np.random.default_rng(0).standard_normal((m, m)))
for rep in range(3):
    ctx.sync(); ctx.timer_start()
    B.basistransform(U[:, :n])
    ms = ctx.timer_stop()
print(f"basistransform N={N} m={m} n={n}: {ms:.2f} ms  ({8 * (m + n) * N / ms / 1e6:.0f} GB/s, {2 * m * n * N / ms / 1e9:.1f} TF/s)")
hv = np.random.default_rng(1).standard_normal(m)
for rep in range(2):
    ctx.sync(); ctx.timer_start()
    B.rmul_householder(2.0 / (hv @ hv), hv, 0, m)
    ms = ctx.timer_stop()
print(f"householder m={m}: {ms:.2f} ms ({16 * m * N / ms / 1e6:.0f} GB/s nominal)")
del B
A = laplacian_rows(4000, 2500, 0, 2500) + 0 * 0
import scipy.sparse as sp
A = A + sp.diags(10 * np.linspace(0, 1, N) ** 2)
op = kk.SparseOperator(A.tocsr(), ctx, symmetric=True)
x0 = np.random.default_rng(3).random(N)
t0 = time.perf_counter()
ctx.prof_reset(); ctx.prof_enable(1)
vals, vecs, info = kk.eigsolve(op, x0, 4, "LM", kk.Lanczos(krylovdim=100, tol=1e-8, maxiter=6), return_device=True)
ctx.sync(); dt = time.perf_counter() - t0
ctx.prof_enable(0)
print(f"eigsolve LM 4 values krylovdim=100: {dt:.3f}s numiter={info.numiter} numops={info.numops} converged={info.converged} vals={vals[:4]}")
print({k: round(ctx.prof_get(k)[0], 1) for k in ("k_project", "k_unproject", "k_spmv_ell", "k_basistransform", "k_scal", "k_axpby")})
