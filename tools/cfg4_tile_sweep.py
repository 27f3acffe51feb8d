"""Config 4 (5M x 1M random sparse map): milliseconds per A x and per A'u as a function of the column-tile width of the
tiled SELL format (KK_SPMV_TILE_COLS, 0 = untiled SELL-64-sigma).  Smaller tiles keep the gathered slice of x closer to the
CU (L2 hit rate up) but add one read-modify-write pass over y per tile; the default (3 MB = 393216 columns) should sit at
the minimum.  usage: python tools/cfg4_tile_sweep.py > profiles/r03_cfg4_tile_sweep.json"""
import json
import os
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "krylovkit.jl_amd"))
sys.path.insert(0, str(ROOT))
import krylovkit_hip as kk  # noqa: E402
from bench import gkl_rows  # noqa: E402

m, n, per = 5_000_000, 1_000_000, 20
A = gkl_rows(m, n, per, 0, m)
ctx = kk.default_context()
xb, yb = kk.DeviceBasis(n, 2, ctx), kk.DeviceBasis(m, 2, ctx)
xb[0].rand_(1); yb[1].rand_(2)
alg_bytes = 12 * A.nnz + 4 * (m + 1) + 8 * (m + n)
rows = []
for tile in (0, 32768, 65536, 131072, 262144, 393216, 524288):
    os.environ["KK_SPMV_TILE_COLS"] = str(tile)
    op = kk.SparseOperator(A, ctx)
    res = {"tile_cols": tile, "x_slice_bytes": tile * 8, "format": None}
    for label, fn in (("A x", lambda: op.apply(xb[0], yb[0])), ("A'u", lambda: op.apply_adjoint(yb[1], xb[1]))):
        for _ in range(3):
            fn()
        ctx.sync(); ctx.timer_start()
        for _ in range(20):
            fn()
        ms = ctx.timer_stop() / 20
        res[label] = {"ms_per_apply": round(ms, 4), "frac_of_8TBps": round(alg_bytes / ms / 1e6 / 8000, 4)}
    res["format"] = op.info()["format"]
    res["device_bytes"] = op.info()["device_bytes"]
    rows.append(res)
    op.free()
    print(json.dumps(res), file=sys.stderr, flush=True)
print(json.dumps({"matrix": f"{m} x {n}, {A.nnz} stored entries", "algorithmic_bytes_per_apply": alg_bytes, "sweep": rows}, indent=1))
