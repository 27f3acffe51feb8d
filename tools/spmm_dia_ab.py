"""A/B of the sweeping multi-column stencil apply at BASELINE config-5 size (10M rows, 16 columns): the 8-byte form (k_spmm_dia) against
the aligned form (k_spmm_dia_al, 2 / 4 columns per wave) for several sweep lengths.  Prints us per apply, TB/s by the algorithmic
bytes (16 N read + 16 N written per column pair... = 16 * nb * N bytes) and whether the output bits equal the 8-byte form's.
usage: python tools/spmm_dia_ab.py [nb]"""
import json
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "krylovkit.jl_amd"))
sys.path.insert(0, str(ROOT))
import krylovkit_hip as kk  # noqa: E402
from krylovkit_hip._lib import check  # noqa: E402
from bench import laplacian_rows  # noqa: E402

nb = int(sys.argv[1]) if len(sys.argv) > 1 else 16
nx, ny = 4000, 2500
N = nx * ny
ctx = kk.default_context()
op = kk.SparseOperator(laplacian_rows(nx, ny, 0, ny), ctx, symmetric=True)
S = kk.DeviceBasis(N, 2 * nb, ctx)
for j in range(nb):
    S[j].rand_(11 + j)


def run(reps=20):
    check(S._lib.kk_block_apply(op.handle, S.handle, 0, S.handle, nb, nb)); ctx.sync()
    t0 = time.perf_counter()
    for _ in range(reps):
        check(S._lib.kk_block_apply(op.handle, S.handle, 0, S.handle, nb, nb))
    ctx.sync()
    return (time.perf_counter() - t0) / reps


ref = None
for rnd in range(2):
    for al, lines in ((0, 16), (0, 4), (4, 16), (4, 8), (4, 4), (4, 2), (2, 16), (2, 8), (2, 4), (2, 2), (2, 1)):
        ctx.set_option("spmm_dia_al", al); ctx.set_option("spmm_dia_al_lines" if al else "spmm_dia_lines", lines)
        dt = min(run() for _ in range(3))
        chk = [S.download(nb + j)[::997].copy() for j in (0, nb - 1)]
        if ref is None:
            ref = chk
        same = all(np.array_equal(a, b) for a, b in zip(chk, ref))
        print(json.dumps({"round": rnd, "spmm_dia_al": al, "spmm_dia_lines": lines, "us_per_apply": round(dt * 1e6, 1),
                          "TBps_algorithmic": round(16.0 * nb * N / dt / 1e12, 3), "bits_equal_8byte_form": bool(same)}), flush=True)
