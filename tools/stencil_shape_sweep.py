"""Constant-coefficient stencil apply (k_spmv_dia CONST) against grid shape: is the rate set by the number of rows or by the
line length D (reuse distance of the x[r +- D] reads)?  usage: python tools/stencil_shape_sweep.py [out.jsonl]"""
import json
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "krylovkit.jl_amd"))
import krylovkit_hip as kk  # noqa: E402


def laplacian_csr_arrays(nx, ny):
    N = nx * ny
    i = np.arange(N, dtype=np.int64)
    ix = i % nx
    has = np.empty((N, 5), dtype=bool)
    has[:, 0] = i >= nx; has[:, 1] = ix > 0; has[:, 2] = True; has[:, 3] = ix < nx - 1; has[:, 4] = i < N - nx
    rowptr = np.zeros(N + 1, dtype=np.int64)
    np.cumsum(has.sum(axis=1, dtype=np.int64), out=rowptr[1:])
    pos = np.flatnonzero(has.ravel())
    offs = np.array([-nx, -1, 0, 1, nx], dtype=np.int64)
    col = (pos // 5 + offs[pos % 5]).astype(np.int32)
    val = np.where(pos % 5 == 2, 4.0, -1.0)
    return rowptr, col, val


if __name__ == "__main__":
    ctx = kk.default_context()
    out = open(sys.argv[1], "w") if len(sys.argv) > 1 else None
    for nx, ny in [(4000, 2500), (1000, 10000), (4000, 12500), (100000, 500), (4000, 25000), (20000, 5000)]:
        N = nx * ny
        rp, col, val = laplacian_csr_arrays(nx, ny)
        op = kk.SparseOperator.from_csr_arrays((N, N), rp, col, val, ctx, symmetric=True)
        del rp, col, val
        B = kk.DeviceBasis(N, 3, ctx)
        B[0].rand_(1); B[1].rand_(2)
        row = {"nx": nx, "ny": ny, "rows": N, "format": op.info()["format"]}
        for name, opts, nbytes in (("const_1pair", dict(spmv_dia=1, spmv_dia_const=1, spmv_dia_pairs=1), 16), ("const_2pairs", dict(spmv_dia=1, spmv_dia_const=1, spmv_dia_pairs=2), 16),
                                   ("const_4pairs", dict(spmv_dia=1, spmv_dia_const=1, spmv_dia_pairs=4), 16), ("stored_diagonals_2pairs", dict(spmv_dia=1, spmv_dia_const=0, spmv_dia_pairs=2), 56),
                                   ("stored_diagonals", dict(spmv_dia=1, spmv_dia_const=0, spmv_dia_pairs=1), 56)):
            for k, v in opts.items():
                ctx.set_option(k, v)
            for _ in range(3):
                op.apply(B[0], B[2])
            ctx.sync(); ctx.timer_start()
            reps = 20
            for _ in range(reps):
                op.apply(B[0], B[2])
            ms = ctx.timer_stop() / reps
            row[name + "_ms"] = round(ms, 4)
            row[name + "_GBps"] = round(nbytes * N / ms / 1e6, 1)
        print(json.dumps(row), flush=True)
        if out:
            out.write(json.dumps(row) + "\n")
        del B, op
    ctx.set_option("spmv_dia", 1); ctx.set_option("spmv_dia_const", 1); ctx.set_option("spmv_dia_pairs", 0)
