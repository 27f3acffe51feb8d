"""Per-launch rates of the BlockLanczos kernels at N = 10M, 16 right-hand sides: block update (by basis width m and
in/out mode), Gram panel (by p), sparse multi-column apply.  usage: python tools/block_micro.py"""
import ctypes as C
import json
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "krylovkit.jl_amd"))
sys.path.insert(0, str(ROOT))
import krylovkit_hip as kk  # noqa: E402
from krylovkit_hip._lib import check, c_dp  # noqa: E402
from bench import laplacian_rows  # noqa: E402

nx, ny, bs = 4000, 2500, 16
N = nx * ny
ctx = kk.default_context()
lib = ctx._lib
S = kk.DeviceBasis(N, 160, ctx)
for j in range(160):
    S[j].rand_(j + 1)
op = kk.SparseOperator(laplacian_rows(nx, ny, 0, ny), ctx, symmetric=True)


def timeit(fn, reps=5):
    fn(); ctx.sync()
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        ctx.sync()
        best = min(best, (time.perf_counter() - t0) / reps)
    return best


out = []
ONLY_SPMM = "--spmm" in sys.argv
for opts in (() if ONLY_SPMM else (dict(bu_prefetch=0),)):
    for k_, v in opts.items():
        ctx.set_option(k_, v)
    for m in (16, 112):
        coef = np.asfortranarray(np.random.default_rng(m).standard_normal((m, bs)) * 1e-3)
        for beta, label in ((0.0, "out of place (beta=0)"), (1.0, "in place (beta=1)")):
            def f():
                check(lib.kk_block_update(S.handle, 130, bs, S.handle, 0, m, coef.ctypes.data_as(c_dp), m, 1.0, beta, None))
            dt = timeit(f)
            byts = (8 * m + (128 if beta == 0 else 256)) * N
            out.append({"kernel": "k_block_update", **opts, "m": m, "mode": label, "ms": round(dt * 1e3, 3), "TBps": round(byts / dt / 1e12, 2)})
            print(json.dumps(out[-1]), flush=True)
ctx.set_option("bu_prefetch", 1)
M = np.zeros((128, bs), order="F")
ctx.set_option("gram_nt", 0)
for bpc in (() if ONLY_SPMM else (2, 8)):
    ctx.set_option("gram_bpc", bpc)
    for p, same in ((16, True), (16, False), (32, False), (64, False), (112, False)):
        cy = 0 if same else 130
        def g():
            check(lib.kk_block_inner(S.handle, 0, p, S.handle, cy, bs, M.ctypes.data_as(c_dp), 128))
        dt = timeit(g)
        byts = (8 * p + (0 if same else 128)) * N
        out.append({"kernel": "k_block_gram", "gram_bpc": bpc, "p": p, "X_is_Y": same, "ms": round(dt * 1e3, 3), "TBps": round(byts / dt / 1e12, 2)})
        print(json.dumps(out[-1]), flush=True)
ctx.set_option("gram_bpc", 8)
for cols in (16, 8, 4):
    ctx.set_option("spmm_cols", cols)
    for bpc in (0, 2, 4, 8):
        ctx.set_option("spmm_bpc", bpc)
        def h():
            check(lib.kk_block_apply(op.handle, S.handle, 0, S.handle, 130, bs))
        dt = timeit(h)
        byts = (12 * 5 + 4 + 256) * N
        out.append({"kernel": "k_spmm_ell", "spmm_cols": cols, "spmm_bpc": bpc, "ms": round(dt * 1e3, 3), "TBps": round(byts / dt / 1e12, 2)})
        print(json.dumps(out[-1]), flush=True)
ctx.set_option("spmm_bpc", 4); ctx.set_option("spmm_cols", 16)
