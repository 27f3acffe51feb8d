"""Secondary measurements for BASELINE.json configs 3-5 on one MI355X (the judged bench line is
bench.py = config 2).  usage: python tools/bench_configs.py [gmres] [gkl] [block] [--full]"""
import json
import os
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


def convdiff(nx, ny, px=0.5, py=0.25):
    n = nx * ny
    r = np.arange(n, dtype=np.int64)
    ix = r % nx
    cols = [r, r - 1, r + 1, r - nx, r + nx]
    vals = [np.full(n, 4.0), np.full(n, -(1 + px)), np.full(n, -(1 - px)), np.full(n, -(1 + py)), np.full(n, -(1 - py))]
    ok = [np.ones(n, bool), ix > 0, ix < nx - 1, r - nx >= 0, r + nx < n]
    rows = np.concatenate([r[m] for m in ok]); cc = np.concatenate([c[m] for c, m in zip(cols, ok)])
    vv = np.concatenate([v[m] for v, m in zip(vals, ok)])
    return sp.csr_matrix((vv, (rows, cc)), shape=(n, n))


def bench_gmres(ctx):
    nx, ny, K = 2000, 1000, 60
    N = nx * ny
    A = convdiff(nx, ny)
    op = kk.SparseOperator(A, ctx)
    out = {}
    for orth in (kk.ClassicalGramSchmidt2(), kk.ModifiedGramSchmidt2()):
        V = kk.DeviceBasis(N, K + 2, ctx)
        x0 = kk.DeviceBasis(N, 1, ctx); x0[0].rand_(4)
        it = kk.ArnoldiIterator(op, x0[0], orth, capacity=K + 2)
        best = 1e9
        for rep in range(4):
            f = kk.initialize(it, V)
            ctx.sync(); t0 = time.perf_counter()
            for _ in range(K - 1):
                f = kk.expand_(it, f)
            ctx.sync(); best = min(best, time.perf_counter() - t0)
        alg = sum((144 + 32 * m) * N for m in range(2, K + 1))
        out[orth.name] = {"arnoldi_it_per_s": round((K - 1) / best, 1), "alg_GBps": round(alg / best / 1e9, 1),
                          "frac_8TBps": round(alg / best / 8e12, 4)}
        V.free()
    b = np.random.default_rng(4).random(N)
    t0 = time.perf_counter()
    x, info = kk.linsolve(op, b, None, kk.GMRES(kk.ModifiedGramSchmidt2(), 20, K, 1e-10 * np.linalg.norm(b)))
    dt = time.perf_counter() - t0
    res = np.linalg.norm(A @ x - b) / np.linalg.norm(b)
    out["linsolve"] = {"seconds": round(dt, 3), "numops": info.numops, "numiter": info.numiter, "converged": info.converged,
                       "true_rel_residual": float(res)}
    print(json.dumps({"config": "3: linsolve(GMRES) 2M-row convection-diffusion, krylovdim=60", **out}), flush=True)


def bench_gkl(ctx, full):
    m, n, per = (5_000_000, 1_000_000, 20) if full else (1_000_000, 200_000, 20)
    rng = np.random.default_rng(5)
    t0 = time.time()
    cols = rng.integers(0, n, size=m * per, dtype=np.int32)
    vals = rng.standard_normal(m * per)
    indptr = np.arange(0, m * per + 1, per, dtype=np.int64)
    A = sp.csr_matrix((vals, cols, indptr), shape=(m, n))
    A.sum_duplicates()
    op = kk.SparseOperator(A, ctx)
    print(f"# gkl operator {m}x{n} nnz={A.nnz} built+uploaded in {time.time() - t0:.1f}s {op.info()}", flush=True)
    K = 30
    out = {}
    for orth in (kk.ClassicalGramSchmidt2(), kk.ModifiedGramSchmidt2()):
        it = kk.GKLIterator(op, rng.random(m), orth, capacity=K + 2)
        best = 1e9
        for rep in range(3):
            f = kk.initialize(it)
            ctx.sync(); t0 = time.perf_counter()
            for _ in range(K - 1):
                f = kk.expand_(it, f)
            ctx.sync(); best = min(best, time.perf_counter() - t0)
        spmv = 2 * (12 * A.nnz + 4 * (m + n + 2) + 8 * (m + n) * 2)
        alg = sum(spmv + (16 * 1 + 24 + 16 * (k - 1) + 24 + 16) * n + (16 + 24 + 16 * k + 24 + 16) * m for k in range(2, K + 1))
        ctx.prof_reset(); ctx.prof_enable(1)
        f = kk.initialize(it)
        for _ in range(K - 1):
            f = kk.expand_(it, f)
        ctx.prof_enable(0)
        prof = {k: round(ctx.prof_get(k)[0], 2) for k in ("k_spmv_ell", "k_spmv_dia", "k_spmv_sell", "k_spmv_csr", "k_project", "k_unproject", "k_unproj_proj", "k_mgs_step", "k_scal", "k_dot")}
        out[orth.name] = {"gkl_it_per_s": round((K - 1) / best, 1), "alg_GBps": round(alg / best / 1e9, 1), "frac_8TBps": round(alg / best / 8e12, 4),
                          "sigma_max_est": max(np.linalg.svd(f.rayleighquotient(), compute_uv=False)), "kernel_ms_one_sweep": prof}
    print(json.dumps({"config": f"4: svdsolve(GKL) {m}x{n} sparse random nnz/row=20, krylovdim=30 (1 GPU)", **out}), flush=True)


def bench_lsmr(ctx, full):
    m, n, per = (5_000_000, 1_000_000, 20) if full else (1_000_000, 200_000, 20)
    rng = np.random.default_rng(5)
    cols = rng.integers(0, n, size=m * per, dtype=np.int32)
    vals = rng.standard_normal(m * per)
    indptr = np.arange(0, m * per + 1, per, dtype=np.int64)
    A = sp.csr_matrix((vals, cols, indptr), shape=(m, n))
    A.sum_duplicates()
    op = kk.SparseOperator(A, ctx)
    b = rng.random(m)
    out = {}
    for K in (1, 30):
        times = {}
        for iters in (20, 120):
            for rep in range(3):
                ctx.sync(); t0 = time.perf_counter()
                x, info = kk.lssolve(op, b, kk.LSMR(kk.ModifiedGramSchmidt(), iters, K, 1e-300))
                ctx.sync(); times[iters] = time.perf_counter() - t0
        per_it = (times[120] - times[20]) / 100
        ctx.prof_reset(); ctx.prof_enable(1)
        kk.lssolve(op, b, kk.LSMR(kk.ModifiedGramSchmidt(), 50, K, 1e-300))
        ctx.prof_enable(0)
        prof = {k: round(ctx.prof_get(k)[0], 2) for k in ("k_spmv_ell", "k_spmv_dia", "k_spmv_sell", "k_spmv_csr", "k_project", "k_unproject",
                                                            "k_lsmr_u", "k_lsmr_hx", "k_axpby", "k_scal", "k_block_gram")}
        out[f"krylovdim={K}"] = {"ms_per_iteration": round(per_it * 1e3, 4), "it_per_s": round(1 / per_it, 1),
                                 "normres": info.normres, "kernel_ms_50_iterations": prof}
    print(json.dumps({"config": f"LSMR (SURVEY 8(f)-3) {m}x{n} sparse random nnz/row=20 (config-4 operator)", **out}), flush=True)


def bench_block(ctx):
    nx, ny, bs, K = 4000, 2500, 16, 100
    N = nx * ny
    A = laplacian_rows(nx, ny, 0, ny)
    op = kk.SparseOperator(A, ctx, symmetric=True)
    rng = np.random.default_rng(7)
    x0 = [rng.random(N) for _ in range(bs)]
    for mode in [int(m_) for m_ in os.environ.get("KK_BENCH_BLOCK_MODES", "1,0").split(",")]:
        ctx.set_option("block_mode", mode)
        it = kk.BlockLanczosIterator(op, x0, K + bs)
        V = None
        best, steps = 1e9, 0
        for rep in range(2):
            f = it.initialize(V); V = f.V
            ctx.prof_reset(); ctx.prof_enable(True)
            ctx.sync(); t0 = time.perf_counter(); steps = 0; alg = 0
            while len(f) < K:
                f = it.expand(f); steps += 1
                alg += (1856 + 16 * len(f)) * N
            ctx.sync(); best = min(best, time.perf_counter() - t0)
            ctx.prof_enable(False)
        prof = {k: round(ctx.prof_get(k)[0], 2) for k in ("k_block_gram", "k_block_update", "k_spmm_ell", "k_spmm_dia", "k_mgs_step", "k_dot", "k_axpby", "k_scal")}
        print(json.dumps({"config": f"5: BlockLanczos bs={bs} N=1e7 krylovdim={K}, block_mode={mode}", "block_steps": steps,
                          "seconds": round(best, 4), "ms_per_block_step": round(best / steps * 1e3, 2),
                          "alg_GBps": round(alg / best / 1e9, 1), "frac_8TBps": round(alg / best / 8e12, 4),
                          "normres": f.normres, "kernel_ms_last_rep": prof}), flush=True)
        V.free()
    ctx.set_option("block_mode", 1)


def bench_cg(ctx):
    import scipy.sparse as sps
    nx, ny = 4000, 2500
    N = nx * ny
    A = (laplacian_rows(nx, ny, 0, ny) + sps.identity(N) * 0.5).tocsr()   # SPD, well conditioned enough for a short run
    op = kk.SparseOperator(A, ctx, symmetric=True)
    b = np.random.default_rng(4).random(N)
    times = {}
    for iters in (60, 260):
        for rep in range(2):
            ctx.sync(); t0 = time.perf_counter()
            x, info = kk.linsolve_cg(op, b, None, kk.CG(iters, 1e-300))
            ctx.sync(); times[iters] = time.perf_counter() - t0
    per_it = (times[260] - times[60]) / 200          # slope: excludes the 80 MB host upload / download of b and x
    alg = (84 + 48 + 24) * N
    print(json.dumps({"config": "CG (SURVEY 8(f)-3) on the 10M-row shifted Laplacian", "seconds_60": round(times[60], 4),
                      "seconds_260": round(times[260], 4), "ms_per_iteration": round(per_it * 1e3, 4),
                      "it_per_s": round(1 / per_it, 1), "alg_GBps": round(alg / per_it / 1e9, 1),
                      "frac_8TBps": round(alg / per_it / 8e12, 4), "normres": info.normres}), flush=True)


def bench_bicgstab(ctx, nx=4000, ny=2500):
    N = nx * ny
    A = convdiff(nx, ny)                                   # config-3 operator (nonsymmetric)
    op = kk.SparseOperator(A, ctx)
    b = np.random.default_rng(4).random(N)
    times = {}
    for iters in (20, 120):
        for rep in range(2):
            ctx.sync(); t0 = time.perf_counter()
            x, info = kk.linsolve_bicgstab(op, b, None, kk.BiCGStab(iters, 1e-300))
            ctx.sync(); times[iters] = time.perf_counter() - t0
    per_it = (times[120] - times[20]) / 100
    alg = (2 * 84 + 8 + 32 + 24 + 56) * N     # 2 SpMV (+ r_shadow read in the first) ; p update ; s ; x/r update
    print(json.dumps({"config": f"BiCGStab (SURVEY 8(f)-3) on the {N}-row convection-diffusion operator (config-3 stencil)",
                      "seconds_20": round(times[20], 4), "seconds_120": round(times[120], 4),
                      "ms_per_iteration": round(per_it * 1e3, 4), "it_per_s": round(1 / per_it, 1),
                      "alg_GBps": round(alg / per_it / 1e9, 1), "frac_8TBps": round(alg / per_it / 8e12, 4),
                      "normres": info.normres}), flush=True)


if __name__ == "__main__":
    ctx = kk.default_context()
    import os
    if os.environ.get("KK_KEEP_MB"):
        ctx.set_option("keep_mb", float(os.environ["KK_KEEP_MB"]))
    what = [a for a in sys.argv[1:] if not a.startswith("--")] or ["gmres", "block", "gkl"]
    if "gmres" in what:
        bench_gmres(ctx)
    if "block" in what:
        bench_block(ctx)
    if "cg" in what:
        bench_cg(ctx)
    if "lsmr" in what:
        bench_lsmr(ctx, "--full" in sys.argv)
    if "bicgstab" in what:
        bench_bicgstab(ctx)
        bench_bicgstab(ctx, 2000, 1000)
    if "gkl" in what:
        bench_gkl(ctx, "--full" in sys.argv)
