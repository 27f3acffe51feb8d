"""Kernel-level micro-benchmarks on the GPU box (exploration tool, not the judged bench).
usage: python tools/microbench.py [N] [--bpc 2,4,8] [--m 8,32,100]"""
import argparse
import sys
import time
from pathlib import Path

import numpy as np
import scipy.sparse as sp

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "krylovkit.jl_amd"))
import krylovkit_hip as kk  # noqa: E402
import ctypes as C  # noqa: E402


def laplacian(nx, ny):
    n = nx * ny
    main = np.full(n, 4.0)
    off1 = np.full(n - 1, -1.0)
    off1[np.arange(1, ny) * nx - 1] = 0.0
    offx = np.full(n - nx, -1.0)
    A = sp.diags([offx, off1, main, off1, offx], [-nx, -1, 0, 1, nx], format="csr")
    A.eliminate_zeros()
    return A


def timeit(ctx, fn, reps=5, warm=2):
    for _ in range(warm):
        fn()
    ctx.sync()
    best = 1e30
    for _ in range(reps):
        ctx.timer_start()
        fn()
        ms = ctx.timer_stop()
        best = min(best, ms)
    return best


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("N", nargs="?", type=int, default=10_000_000)
    ap.add_argument("--bpc", default="2,3,4,6,8,16")
    ap.add_argument("--m", default="8,32,64,100")
    ap.add_argument("--sweep", type=int, default=100)
    args = ap.parse_args()
    N = args.N
    ctx = kk.default_context()
    lib = ctx._lib
    ms_list = [int(x) for x in args.m.split(",")]
    mmax = max(ms_list)
    B = kk.DeviceBasis(N, mmax + 3, ctx)
    for j in range(mmax + 3):
        B[j].rand_(j + 1)
    B.length = mmax
    coef = np.random.default_rng(0).standard_normal(mmax) * 1e-3
    y = np.zeros(mmax)
    print(f"N={N} ld={B.info()[1]} CUs={ctx.get_option('num_cus')}")
    for bpc in [int(x) for x in args.bpc.split(",")]:
        ctx.set_option("blocks_per_cu", bpc)
        t = timeit(ctx, lambda: B[mmax].inner(B[mmax + 1]))
        print(f"bpc={bpc:2d} dot            {t:8.3f} ms  {16 * N / t / 1e6:8.1f} GB/s")
        t = timeit(ctx, lambda: B[mmax].add_(B[mmax + 1], 1e-9, 1.0))
        print(f"bpc={bpc:2d} axpby          {t:8.3f} ms  {24 * N / t / 1e6:8.1f} GB/s")
        for m in ms_list:
            t = timeit(ctx, lambda: B.project(B[mmax], 0, m, y=y[:m]))
            print(f"bpc={bpc:2d} project  m={m:3d} {t:8.3f} ms  {(8 * m + 8) * N / t / 1e6:8.1f} GB/s")
            t = timeit(ctx, lambda: B.unproject(B[mmax + 1], coef[:m], 0, m, -1.0, 1.0))
            print(f"bpc={bpc:2d} unproj   m={m:3d} {t:8.3f} ms  {(8 * m + 16) * N / t / 1e6:8.1f} GB/s")
    # SpMV + full Lanczos sweep
    nx = 4000
    ny = N // nx
    if nx * ny == N:
        t0 = time.time()
        A = laplacian(nx, ny)
        print(f"laplacian built in {time.time() - t0:.1f}s nnz={A.nnz}")
        t0 = time.time()
        op = kk.SparseOperator(A, ctx, symmetric=True)
        print(f"operator uploaded in {time.time() - t0:.1f}s", op.info())
        ctx.set_option("blocks_per_cu", 4)
        t = timeit(ctx, lambda: op.apply(B[0], B[1]))
        print(f"spmv ELL       {t:8.3f} ms  {(12 * 5 * N + 16 * N) / t / 1e6:8.1f} GB/s (ELL bytes) ")
        del B
        K = args.sweep
        for bpc in (4, 8):
            ctx.set_option("blocks_per_cu", bpc)
            for orth, mode in ((kk.ClassicalGramSchmidt2(), 1), (kk.ModifiedGramSchmidt2(), 1), (kk.ModifiedGramSchmidt2(), 0)):
                ctx.set_option("mgs_mode", mode)
                it = kk.LanczosIterator(op, np.random.default_rng(3).random(N), orth, capacity=K + 2)
                fact = kk.initialize(it)
                V = fact.V
                for rep in range(2):
                    fact = kk.initialize(it, V)
                    ctx.sync()
                    t0 = time.perf_counter()
                    for _ in range(K - 1):
                        fact = kk.expand_(it, fact)
                    ctx.sync()
                    dt = time.perf_counter() - t0
                alg_bytes = sum((176 + 16 * m) * N for m in range(2, K + 1))
                print(f"bpc={bpc} lanczos {orth.name} mgs_mode={mode}: {K - 1} expands in {dt:.4f}s = {(K - 1) / dt:.1f} it/s, "
                      f"{alg_bytes / dt / 1e9:.1f} GB/s algorithmic = {alg_bytes / dt / 8e12 * 100:.1f}% of 8 TB/s; "
                      f"alpha[-1]={fact.alphas[-1]:.12f} beta[-1]={fact.betas[-1]:.12f}")
                V.free()
                del fact, V, it


if __name__ == "__main__":
    main()
