"""SpMV micro-benchmark on the config-4 operator (5M x 1M, 20 nnz/row random) and its transpose:
ms per apply for several KK_SPMV_TILE_COLS / KK_SPMV_FORMAT settings (exploration tool).
usage: python tools/spmv_bench.py [--small] [tile_cols ...]"""
import os
import sys
import time
from pathlib import Path

import numpy as np
import scipy.sparse as sp

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "krylovkit.jl_amd"))
import krylovkit_hip as kk  # noqa: E402


def main():
    small = "--small" in sys.argv
    m, n, per = (1_000_000, 200_000, 20) if small else (5_000_000, 1_000_000, 20)
    rng = np.random.default_rng(5)
    cols = rng.integers(0, n, size=m * per, dtype=np.int32)
    vals = rng.standard_normal(m * per)
    indptr = np.arange(0, m * per + 1, per, dtype=np.int64)
    A = sp.csr_matrix((vals, cols, indptr), shape=(m, n))
    A.sum_duplicates()
    ctx = kk.default_context()
    settings = [a for a in sys.argv[1:] if not a.startswith("--")] or ["0", "262144", "524288"]
    U, V = kk.DeviceBasis(m, 2, ctx), kk.DeviceBasis(n, 2, ctx)
    U[0].rand_(1); V[0].rand_(2)
    for s in settings:
        if s in ("csr", "sell", "ell"):
            os.environ["KK_SPMV_FORMAT"] = s
            os.environ.pop("KK_SPMV_TILE_COLS", None)
        else:
            os.environ.pop("KK_SPMV_FORMAT", None)
            os.environ["KK_SPMV_TILE_COLS"] = s
        t0 = time.time()
        op = kk.SparseOperator(A, ctx)
        op.apply(V[0], U[1]); op.apply_adjoint(U[0], V[1]); ctx.sync()
        tb = time.time() - t0
        res = {}
        for name, fn in (("A", lambda: op.apply(V[0], U[1])), ("At", lambda: op.apply_adjoint(U[0], V[1]))):
            best = 1e9
            for rep in range(3):
                ctx.sync(); ctx.timer_start()
                for _ in range(10):
                    fn()
                best = min(best, ctx.timer_stop() / 10)
            res[name] = round(best, 4)
        print(f"setting={s:>8} format={op.info()['format']:<10} bytes={op.info()['device_bytes'] / 1e9:.2f}GB build={tb:.1f}s ms_per_apply={res}", flush=True)
        op.free()


if __name__ == "__main__":
    main()
