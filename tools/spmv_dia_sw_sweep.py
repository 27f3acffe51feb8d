"""k_spmv_dia_sw against k_spmv_dia on the two operators whose single-vector apply sits on the measured paths:
config 2's 4000 x 2500 5-point Laplacian as the Lanczos step calls it (w = A v - beta v_prev with the fused <v, w>: 24 N bytes) and
config 3's 2000 x 1000 convection-diffusion operator as the Arnoldi step calls it (w = A v: 16 N bytes).  Strips per wave 1 / 2, lines per
sweep 2 .. 32, against the round-5 kernel.  HIP-event brackets per launch (they include the launch gap: short launches read high); the
rocprofv3 figures of the same kernels are in profiles/r06_*kernel_stats.csv.  One JSON line per variant -> profiles/r06_spmv_dia_sw_sweep.jsonl"""
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
from bench import laplacian_rows, convdiff_rows, NX, NY  # noqa: E402

ctx = kk.default_context()
dot = kk.DeviceBasis(8, 1, ctx)
n_, ld_, cap_, dp = C.c_int64(), C.c_int64(), C.c_int(), C.c_void_p()
check(ctx._lib.kk_basis_info(dot.handle, C.byref(n_), C.byref(ld_), C.byref(cap_), C.byref(dp)))

for name, A, sym, lanczos_form in (("config2_4000x2500_lanczos_step", laplacian_rows(NX, NY, 0, NY), True, True),
                                   ("config3_2000x1000_arnoldi_step", convdiff_rows(2000, 1000), False, False)):
    N = A.shape[0]
    op = kk.SparseOperator(A, ctx, symmetric=sym, via_csc=True) if sym else kk.SparseOperator(A, ctx)
    B = kk.DeviceBasis(N, 4, ctx)
    for j in range(3):
        B[j].rand_(3 + j)
    nbytes = (24.0 if lanczos_form else 16.0) * N
    ref = None

    def run(reps):
        for _ in range(reps):
            if lanczos_form:
                check(ctx._lib.kk_apply_fused_dev(op.handle, B.handle, 1, 0, 2, 0.7, 2, dp))
            else:
                op.apply(B[1], B[2])

    for sw, lines in [(0, 0)] + [(s_, l_) for s_ in (1, 2) for l_ in (2, 4, 8, 16, 32)]:
        ctx.set_option("spmv_dia_sw", sw); ctx.set_option("spmv_dia_sw_lines", lines)
        run(3); ctx.sync()
        ctx.prof_reset(); ctx.prof_enable(1)
        run(40)
        ctx.sync(); ctx.prof_enable(0)
        ms, n = ctx.prof_get("k_spmv_dia")
        y = B[2].get()
        if ref is None:
            ref = y.copy()
        us = ms / n * 1e3
        print(json.dumps({"operator": name, "rows": N, "kernel": "k_spmv_dia_sw" if sw else "k_spmv_dia", "strips_per_wave": sw, "lines": lines,
                          "us": round(us, 2), "GBps": round(nbytes / us / 1e3, 1), "frac_of_8TBps": round(nbytes / us / 1e3 / 8000, 4),
                          "bitwise_equal_to_k_spmv_dia": bool(np.array_equal(y, ref))}), flush=True)
    ctx.set_option("spmv_dia_sw", 1); ctx.set_option("spmv_dia_sw_lines", 0)
    B.free(); del op
