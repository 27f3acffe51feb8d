"""The parity figures of tests/test_gpu_fullsize.py as numbers (the tests only assert the 1e-10 bar): GPU path vs the CPU
twin oracle/cpu_ref.c at BASELINE.json's sizes for configs 3b (convergent GMRES), 4 (GKL) and 5 (BlockLanczos, both block
modes).  TEST-SIDE tool: it loads the oracle as the checker.  usage: python tools/fullsize_parity_report.py > profiles/r03_fullsize_parity.json"""
import json
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "krylovkit.jl_amd"))
sys.path.insert(0, str(ROOT / "tests"))
sys.path.insert(0, str(ROOT))
import krylovkit_hip as kk          # noqa: E402
import cpu_ref_lib as cr            # noqa: E402
from bench import gkl_rows, laplacian_rows, NX, NY   # noqa: E402

ctx = kk.default_context()
lib = cr.load()
nth = cr.usable_threads()
out = {"cpu_threads": nth}

# ---- config 3b: GMRES(60) on the 2M-row convection-diffusion operator shifted by a0 = 0.15, rtol 1e-10
from tools.bench_configs import convdiff   # noqa: E402
nx, ny = 2000, 1000
A = convdiff(nx, ny)
b = np.random.default_rng(4).random(nx * ny)
nb = np.linalg.norm(b)
tol, a0 = 1e-10 * nb, 0.15
res = {}
for code, orth in ((3, kk.ModifiedGramSchmidt2()), (2, kk.ClassicalGramSchmidt2())):
    t0 = time.perf_counter()
    xc, ic, tc = cr.run_gmres(lib, A, b, None, a0, 1.0, 60, 20, tol, code, nthreads=nth)
    t_cpu = time.perf_counter() - t0
    tr = []
    t0 = time.perf_counter()
    x, info = kk.linsolve(kk.SparseOperator(A, ctx), b, None, kk.GMRES(orth, 20, 60, tol), a0, 1.0, trace=tr)
    t_gpu = time.perf_counter() - t0
    tg = np.array([t[2] for t in tr])
    res[orth.name] = {"gpu": {"converged": info.converged, "numiter": info.numiter, "numops": info.numops, "normres": info.normres},
                      "cpu": ic, "counts_equal": (info.converged, info.numiter, info.numops) == (ic["converged"], ic["numiter"], ic["numops"]),
                      "trace_len": len(tg), "trace_max_relerr": float(np.max(np.abs(tg - tc) / tc)),
                      "final_normres_absdiff_over_tol": float(abs(info.normres - ic["normres"]) / tol),
                      "solution_relerr": float(np.linalg.norm(x - xc) / np.linalg.norm(xc)), "seconds": {"cpu": round(t_cpu, 2), "gpu_incl_setup": round(t_gpu, 2)}}
out["config3b_gmres_2M_a0_0.15_rtol_1e-10"] = res
del A

# ---- config 4: GKL 5M x 1M, 29 expand! steps
m, n, per, K = 5_000_000, 1_000_000, 20, 30
A = gkl_rows(m, n, per, 0, m)
u0 = np.random.default_rng([6, 0]).random(m)
res = {}
for name in ("mgs2", "cgs2"):
    orth = kk.Orthogonalizer(name)
    op = kk.SparseOperator(A, ctx)
    it = kk.GKLIterator(op, u0, orth, capacity=K + 2)
    f = kk.initialize(it)
    for _ in range(K - 1):
        f = kk.expand_(it, f)
    al_g, be_g, Bg = np.array(f.alphas), np.array(f.betas), f.rayleighquotient()
    del f, it
    op.free()
    t0 = time.perf_counter()
    al_c, be_c, _, _ = cr.run_gkl(lib, A, u0, K - 1, orth.code, nthreads=nth)
    t_cpu = time.perf_counter() - t0
    Bc = np.diag(al_c) + np.diag(be_c[:-1], -1)
    sg, sc = np.linalg.svd(Bg, compute_uv=False)[:20], np.linalg.svd(Bc, compute_uv=False)[:20]
    res[name] = {"alpha_max_relerr": float(np.max(np.abs(al_g - al_c) / np.abs(al_c))), "beta_max_relerr": float(np.max(np.abs(be_g - be_c) / np.abs(be_c))),
                 "top20_singular_values_max_relerr": float(np.max(np.abs(sg - sc) / sc)), "sigma_max": float(sg[0]), "cpu_seconds": round(t_cpu, 2)}
out["config4_gkl_5Mx1M_29_steps"] = res
del A

# ---- config 5: BlockLanczos 10M x 16 -> 112 basis vectors, both block modes
N, bs, Kb = NX * NY, 16, 100
A = laplacian_rows(NX, NY, 0, NY)
res = {}
ref = None
for mode in (1, 0):
    ctx.set_option("block_mode", mode)
    op = kk.SparseOperator(A, ctx, symmetric=True)
    S = kk.DeviceBasis(N, Kb + 3 * bs, ctx)
    it = kk.BlockLanczosIterator(op, [None] * bs, Kb + bs)
    area_b = it.maxdim + bs
    X0 = np.empty((N, bs), order="F")
    for j in range(bs):
        S[area_b + j].rand_(100 + j)
        X0[:, j] = S[area_b + j].get()
    it.x0 = [S[area_b + j] for j in range(bs)]
    f = it.initialize(S)
    while len(f) < Kb:
        f = it.expand(f)
    k = len(f)
    Hg, nR = f.H[:k, :k].copy(), f.normres
    S.free(); op.free()
    if ref is None:
        t0 = time.perf_counter()
        ref = cr.run_blocklanczos(lib, A, X0, target_dim=Kb, max_steps=6, qr_tol=it.qr_tol, nthreads=nth)
        ref["cpu_seconds"] = round(time.perf_counter() - t0, 2)
    Hc = ref["H"]
    scale = float(np.max(np.abs(Hc)))
    eg, ec = np.linalg.eigvalsh((Hg + Hg.T) / 2), np.linalg.eigvalsh((Hc + Hc.T) / 2)
    res["default (CholQR2 + one-pass projection with Gram correction)" if mode == 1 else "strict (reference order)"] = {
        "k": k, "block_sizes_equal": ref["sizes"] == [bs] * 7, "eigenvalues_max_abserr_over_norm": float(np.max(np.abs(eg - ec)) / scale),
        "H_max_abserr_over_norm": float(np.max(np.abs(Hg - Hc)) / scale), "norm_R_relerr": float(abs(nR - ref["norm_R"]) / ref["norm_R"])}
ctx.set_option("block_mode", 1)
res["cpu_seconds"] = ref["cpu_seconds"]
out["config5_blocklanczos_10M_bs16_112_vectors"] = res
print(json.dumps(out, indent=1))
