"""BlockLanczos (BASELINE.json configs[4]: 10M rows, block size 16, krylovdim 100) step-time variants on one MI355X:
synchronous vs asynchronous block step, resident-block cap of the multi-column sparse apply.  usage: python tools/block_sweep.py"""
import json
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "krylovkit.jl_amd"))
sys.path.insert(0, str(ROOT))
import krylovkit_hip as kk  # noqa: E402
from bench import laplacian_rows  # noqa: E402

nx, ny, bs, K = 4000, 2500, 16, 100
N = nx * ny
ctx = kk.default_context()
op = kk.SparseOperator(laplacian_rows(nx, ny, 0, ny), ctx, symmetric=True)
S = kk.DeviceBasis(N, K + 3 * bs, ctx)
it = kk.BlockLanczosIterator(op, [None] * bs, K + bs)
area_b = it.maxdim + bs


def run():
    for j in range(bs):
        S[area_b + j].rand_(100 + j)
    it.x0 = [S[area_b + j] for j in range(bs)]
    f = it.initialize(S)
    ctx.sync()
    t0 = time.perf_counter()
    steps, alg = 0, 0
    while len(f) < K:
        f = it.expand(f)
        steps += 1
        alg += (1856 + 16 * len(f)) * N
    ctx.sync()
    return time.perf_counter() - t0, steps, alg, f


variants = [("two-pass (three-term, then panel), gather SpMM, both CholQR2 rounds", dict(block_fuse=1, spmm_dia=0, qr_skip_tol=0, resid_gram=0)),
            ("one-pass projection with Gram correction, sweeping SpMM, both CholQR2 rounds", dict(block_fuse=5, spmm_dia=1, qr_skip_tol=0, resid_gram=0)),
            ("+ second back-substitution skipped when |Q1'Q1 - I| <= 2e-14", dict(qr_skip_tol=2e-14)),
            ("shipped: + Gram matrix of the residual block from the panel pass (no first Gram pass)", dict(resid_gram=1))]
for name, opts in variants:
    for k_, v in opts.items():
        ctx.set_option(k_, v)
    run()
    best = 1e9
    for rep in range(3):
        dt, steps, alg, f = run()
        best = min(best, dt)
    ctx.prof_reset(); ctx.prof_enable(1)
    run()
    ctx.prof_enable(0)
    prof = {k_: round(ctx.prof_get(k_)[0], 2) for k_ in ("k_block_gram", "k_block_update", "k_spmm_ell", "k_spmm_dia", "k_block_qr_fused") if ctx.prof_get(k_)[1]}
    print(json.dumps({"variant": name, "ms_per_block_step": round(best / steps * 1e3, 3), "frac_8TBps": round(alg / best / 8e12, 4),
                      "normres": f.normres, "last_qr_dev": ctx.get_option("last_qr_dev") if hasattr(ctx, "get_option") else None, "kernel_ms_one_run": prof}), flush=True)
ctx.set_option("block_fuse", 5); ctx.set_option("qr_skip_tol", 2e-14); ctx.set_option("resid_gram", 1)
