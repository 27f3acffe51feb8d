"""k_spmv_dia on config 2's operator (4000 x 2500 5-point Laplacian, value-free form) as the Lanczos step calls it (w = A v -
beta v_prev with the fused <v, w>): aligned 16-byte loads + lane-shift neighbours on / off, 1 / 2 / 4 row pairs per lane,
non-temporal store on / off.  One JSON line per variant -> profiles/r05_spmv_dia_sweep.jsonl"""
import ctypes as C
import json
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "krylovkit.jl_amd"))
sys.path.insert(0, str(ROOT))
import krylovkit_hip as kk          # noqa: E402
from krylovkit_hip._lib import check  # noqa: E402
from bench import laplacian_rows, NX, NY  # noqa: E402

ctx = kk.default_context()
A = laplacian_rows(NX, NY, 0, NY)
N = A.shape[0]
op = kk.SparseOperator(A, ctx, symmetric=True, via_csc=True)
B = kk.DeviceBasis(N, 4, ctx)
for j in range(3):
    B[j].rand_(3 + j)
dot = kk.DeviceBasis(8, 1, ctx)      # 8 doubles of device memory for the fused dot
_, _, _, dptr = (None, None, None, None)
info = B.info() if hasattr(B, "info") else None
n_, ld_, cap_, dp = C.c_int64(), C.c_int64(), C.c_int(), C.c_void_p()
check(ctx._lib.kk_basis_info(dot.handle, C.byref(n_), C.byref(ld_), C.byref(cap_), C.byref(dp)))
ref = None
for aligned in (0, 1):
    for pairs in (1, 2, 4):
        for nt in (1, 0):
            ctx.set_option("spmv_dia_aligned", aligned)
            ctx.set_option("spmv_dia_pairs", pairs)
            ctx.set_option("nt_store_rows", 4000000 if nt else 1e12)
            for _ in range(3):
                check(ctx._lib.kk_apply_fused_dev(op.handle, B.handle, 1, 0, 2, 0.7, 2, dp))
            ctx.sync()
            ctx.prof_reset(); ctx.prof_enable(1)
            for _ in range(40):
                check(ctx._lib.kk_apply_fused_dev(op.handle, B.handle, 1, 0, 2, 0.7, 2, dp))
            ctx.sync(); ctx.prof_enable(0)
            ms, n = ctx.prof_get("k_spmv_dia")
            y = B[2].get()
            if ref is None:
                ref = y.copy()
            us = ms / n * 1e3
            print(json.dumps({"aligned": aligned, "pairs": pairs, "nt_store": nt, "us": round(us, 2), "GBps_24N": round(24.0 * N / us / 1e3, 1),
                              "frac_of_8TBps": round(24.0 * N / us / 1e3 / 8000, 4), "bitwise_equal_to_first": bool(np.array_equal(y, ref))}), flush=True)
