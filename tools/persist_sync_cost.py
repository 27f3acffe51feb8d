"""What one grid reduction of the persistent strict-MGS kernel costs, in isolation: the same kernel on a vector of 1024
rows (every CU takes part in every reduction, the streams are empty) with m = 200 basis vectors -- time per launch / m is
the per-vector synchronisation cost that the 10M-row sweep pays on top of its streams.  Both reduction forms
(option persist_sync).  usage: python tools/persist_sync_cost.py"""
import json
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "krylovkit.jl_amd"))
import krylovkit_hip as kk  # noqa: E402

ctx = kk.default_context()
ctx.set_option("mgs_mode", 0)
out = []
for n in (1024, 262144, 2_000_000, 10_000_000):
    m = 200 if n <= 2_000_000 else 100
    rng = np.random.default_rng(1)
    B = kk.DeviceBasis(n, m + 2, ctx)
    for j in range(m + 1):
        B[j].rand_(10 + j)
    B.length = m
    for sync in (0, 1):
        ctx.set_option("persist_sync", sync)
        for threads in (512, 1024):
            ctx.set_option("persist_threads", threads)
            for _ in range(3):
                B.orthogonalize(B[m], kk.ModifiedGramSchmidt(), want_norm=True)
            ctx.sync()
            reps = 20
            t0 = time.perf_counter()
            for _ in range(reps):
                B.orthogonalize(B[m], kk.ModifiedGramSchmidt(), want_norm=True)
            ctx.sync()
            dt = (time.perf_counter() - t0) / reps
            ctx.prof_reset(); ctx.prof_enable(1)
            B.orthogonalize(B[m], kk.ModifiedGramSchmidt(), want_norm=True)
            ctx.prof_enable(0)
            kms, kn = ctx.prof_get("k_mgs_persist")
            out.append({"rows": n, "basis_vectors": m, "persist_sync": sync, "threads": threads, "kernel_ms": round(kms, 4), "launches": kn,
                        "us_per_vector_kernel": round(kms * 1e3 / (m + 1), 3), "us_per_vector_wall": round(dt * 1e6 / (m + 1), 3)})
            print(json.dumps(out[-1]), flush=True)
    B.free()
ctx.set_option("persist_sync", 0); ctx.set_option("persist_threads", 512); ctx.set_option("mgs_mode", 2)
