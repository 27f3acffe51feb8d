"""Strict (reference-order) MGS2 Lanczos sweep on the 10M-row workload: persistent register-resident kernel vs the
launch-per-vector kernel, block size and load-policy variants.  usage: python tools/strict_sweep.py [ny]"""
import json
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "krylovkit.jl_amd"))
sys.path.insert(0, str(ROOT))
import krylovkit_hip as kk  # noqa: E402
from bench import laplacian_rows, NX, NY, KRYLOVDIM, algorithmic_bytes_sweep  # noqa: E402

ny = int(sys.argv[1]) if len(sys.argv) > 1 else NY
ctx = kk.default_context()
N = NX * ny
op = kk.SparseOperator(laplacian_rows(NX, ny, 0, ny), ctx, symmetric=True, via_csc=True)
V = kk.DeviceBasis(N, KRYLOVDIM + 2, ctx)
x0 = kk.DeviceBasis(N, 1, ctx)
x0[0].rand_(3)
it = kk.LanczosIterator(op, x0[0], kk.ModifiedGramSchmidt2(), capacity=KRYLOVDIM + 2)


def sweep():
    f = kk.initialize(it, V)
    for _ in range(KRYLOVDIM - 1):
        f = kk.expand_(it, f)
    return f


ref = None
for name, opts in [("lowsync", dict(mgs_mode=1, spmv_dia=1, spmv_dia_pairs=1)),
                   ("strict persistent 1024 nt, no LDS parking", dict(mgs_mode=0, mgs_persist=1, persist_threads=1024, persist_nt=1, persist_lds=0)),
                   ("strict persistent 512 nt, no LDS parking", dict(mgs_mode=0, mgs_persist=1, persist_threads=512, persist_nt=1, persist_lds=0)),
                   ("strict persistent 512 plain, no LDS parking", dict(mgs_mode=0, mgs_persist=1, persist_threads=512, persist_nt=0, persist_lds=0)),
                   ("strict persistent 1024 nt, LDS parking (9 of 20 grid-rows)", dict(mgs_mode=0, mgs_persist=1, persist_threads=1024, persist_nt=1, persist_lds=1)),
                   ("strict persistent 512 nt, LDS parking (19 of 39 grid-rows)", dict(mgs_mode=0, mgs_persist=1, persist_threads=512, persist_nt=1, persist_lds=1)),
                   ("strict persistent 512 plain, LDS parking", dict(mgs_mode=0, mgs_persist=1, persist_threads=512, persist_nt=0, persist_lds=1)),
                   ("strict persistent 512 nt, LDS (19) + register (8) parking of 39 grid-rows", dict(mgs_mode=0, mgs_persist=1, persist_threads=512, persist_nt=1, persist_lds=2)),
                   ("strict persistent 512 plain, LDS + register parking", dict(mgs_mode=0, mgs_persist=1, persist_threads=512, persist_nt=0, persist_lds=2)),
                   ("strict persistent 512 nt, LDS + register parking, relayed grid reduction", dict(mgs_mode=0, mgs_persist=1, persist_threads=512, persist_nt=1, persist_lds=2, persist_sync=1)),
                   ("lowsync again", dict(mgs_mode=1, persist_sync=0))]:
    for k, v in opts.items():
        ctx.set_option(k, v)
    try:
        sweep()
        ctx.sync()
        best = 1e9
        for rep in range(3):
            t0 = time.perf_counter()
            f = sweep()
            ctx.sync()
            best = min(best, time.perf_counter() - t0)
        ctx.prof_reset(); ctx.prof_enable(1)
        sweep()
        ctx.prof_enable(0)
        prof = {k: round(ctx.prof_get(k)[0], 2) for k in ("k_mgs_persist", "k_mgs_step", "k_project", "k_unproject", "k_spmv_ell", "k_spmv_dia", "k_scal")
                if ctx.prof_get(k)[1]}
        al = np.array(f.alphas)
        if ref is None:
            ref = al
        print(json.dumps({"variant": name, "it_per_s": round((KRYLOVDIM - 1) / best, 1), "ms_per_sweep": round(best * 1e3, 2),
                          "frac_8TBps_algorithmic": round(algorithmic_bytes_sweep(N, KRYLOVDIM) / best / 8e12, 4),
                          "max_alpha_reldiff_vs_lowsync": float(np.max(np.abs(al - ref) / np.abs(ref))), "kernel_ms": prof}), flush=True)
    except Exception as e:  # noqa: BLE001
        print(json.dumps({"variant": name, "error": str(e)}), flush=True)
ctx.set_option("mgs_mode", 1)
