"""Single-column apply on the 10M-row 5-point Laplacian: ELL gather kernel, diagonal kernel, value-free constant-coefficient
kernel (1 / 2 row pairs per lane), plain and with the fused Lanczos epilogue.  usage: python tools/stencil_spmv_bench.py"""
import json
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "krylovkit.jl_amd"))
sys.path.insert(0, str(ROOT))
import krylovkit_hip as kk  # noqa: E402
from bench import laplacian_rows, NX, NY  # noqa: E402

ctx = kk.default_context()
N = NX * NY
op = kk.SparseOperator(laplacian_rows(NX, NY, 0, NY), ctx, symmetric=True, via_csc=True)
B = kk.DeviceBasis(N, 4, ctx)
B[0].rand_(1); B[1].rand_(2)
for name, opts, nbytes in (("ELL gather", dict(spmv_dia=0), 84), ("diagonals streamed", dict(spmv_dia=1, spmv_dia_const=0, spmv_dia_pairs=1), 64),
                           ("constant coefficients, 1 pair/lane", dict(spmv_dia=1, spmv_dia_const=1, spmv_dia_pairs=1), 24),
                           ("constant coefficients, 2 pairs/lane", dict(spmv_dia=1, spmv_dia_const=1, spmv_dia_pairs=2), 24)):
    for k, v in opts.items():
        ctx.set_option(k, v)
    for label, fn in (("y = A x", lambda: op.apply(B[0], B[2])), ("y = 0.5 x - A x", lambda: op.apply_affine(B[0], B[2], 0.5, -1.0))):
        for _ in range(5):
            fn()
        ctx.sync(); ctx.timer_start()
        for _ in range(50):
            fn()
        ms = ctx.timer_stop() / 50
        print(json.dumps({"kernel": name, "apply": label, "ms": round(ms, 4), "bytes_per_row": nbytes - (8 if label == "y = A x" else 8),
                          "GBps": round((nbytes - 8) * N / ms / 1e6, 1)}), flush=True)
ctx.set_option("spmv_dia", 1); ctx.set_option("spmv_dia_const", 1); ctx.set_option("spmv_dia_pairs", 0)
