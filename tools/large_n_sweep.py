"""Lanczos sweep (krylovdim 30) on vectors that are far beyond the Infinity Cache: 10^8 rows (0.8 GB per vector, 25 GB of
basis) against the 10^7-row headline size.  usage: python tools/large_n_sweep.py [out.jsonl]"""
import json
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "krylovkit.jl_amd"))
sys.path.insert(0, str(ROOT / "tools"))
sys.path.insert(0, str(ROOT))
import krylovkit_hip as kk  # noqa: E402
from stencil_shape_sweep import laplacian_csr_arrays  # noqa: E402
from bench import algorithmic_bytes_sweep  # noqa: E402

K = 30
ctx = kk.default_context()
out = open(sys.argv[1], "w") if len(sys.argv) > 1 else None
for nx, ny in [(4000, 2500), (4000, 25000)]:
    N = nx * ny
    rp, col, val = laplacian_csr_arrays(nx, ny)
    op = kk.SparseOperator.from_csr_arrays((N, N), rp, col, val, ctx, symmetric=True)
    del rp, col, val
    V = kk.DeviceBasis(N, K + 2, ctx)
    x0 = kk.DeviceBasis(N, 1, ctx)
    x0[0].rand_(3)
    for oname, orth in (("mgs2", kk.ModifiedGramSchmidt2()), ("cgs2", kk.ClassicalGramSchmidt2())):
        it = kk.LanczosIterator(op, x0[0], orth, capacity=K + 2)

        def sweep():
            f = kk.initialize(it, V)
            for _ in range(K - 1):
                f = kk.expand_(it, f)
            return f
        sweep(); ctx.sync()
        best = 1e9
        for rep in range(3):
            t0 = time.perf_counter(); sweep(); ctx.sync()
            best = min(best, time.perf_counter() - t0)
        ctx.prof_reset(); ctx.prof_enable(1); sweep(); ctx.prof_enable(0)
        prof = {k: round(ctx.prof_get(k)[0], 2) for k in ("k_mgs_persist", "k_project", "k_unproject", "k_spmv_dia", "k_scal", "k_lowsync_solve") if ctx.prof_get(k)[1]}
        row = {"rows": N, "orth": oname, "it_per_s": round((K - 1) / best, 1), "ms_per_sweep": round(best * 1e3, 2),
               "alg_GBps": round(algorithmic_bytes_sweep(N, K) / best / 1e9, 1), "kernel_ms": prof}
        print(json.dumps(row), flush=True)
        if out:
            out.write(json.dumps(row) + "\n")
    del V, x0, op
