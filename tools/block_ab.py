"""A/B of the BlockLanczos step (BASELINE.json configs[4]: 10M rows, block size 16, krylovdim 100) under option settings.
usage: python tools/block_ab.py key=v[,key=v] [key=v ...]   (each argument is one variant; 'base' = library defaults)
Prints ms per block step, the per-class kernel time of one profiled sweep and a checksum of the last R block (must agree
between variants to rounding)."""
import json
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "krylovkit.jl_amd"))
sys.path.insert(0, str(ROOT))
import krylovkit_hip as kk  # noqa: E402
from bench import laplacian_rows  # noqa: E402

nx, ny, bs, K = 4000, 2500, 16, 100
N = nx * ny
ctx = kk.default_context()
op = kk.SparseOperator(laplacian_rows(nx, ny, 0, ny), ctx, symmetric=True)
S = kk.DeviceBasis(N, K + 3 * bs, ctx)
it = kk.BlockLanczosIterator(op, [None] * bs, K + bs)
area_b = it.maxdim + bs


def run():
    for j in range(bs):
        S[area_b + j].rand_(100 + j)
    it.x0 = [S[area_b + j] for j in range(bs)]
    f = it.initialize(S)
    ctx.sync()
    t0 = time.perf_counter()
    steps = 0
    while len(f) < K:
        f = it.expand(f)
        steps += 1
    ctx.sync()
    return time.perf_counter() - t0, steps, f


variants = sys.argv[1:] or ["base"]
defaults = {}
for rnd in range(2):
    for var in variants:
        opts = {} if var == "base" else {kv.split("=")[0]: float(kv.split("=")[1]) for kv in var.split(",")}
        for k, v in opts.items():
            defaults.setdefault(k, ctx.get_option(k))
            ctx.set_option(k, v)
        run()
        best, steps, f = 1e9, 0, None
        for _ in range(3):
            dt, steps, f = run()
            best = min(best, dt)
        ctx.prof_reset(); ctx.prof_enable(1); run(); ctx.sync(); ctx.prof_enable(0)
        cls = {k: round(ctx.prof_get(k)[0], 3) for k in ("k_block_gram", "k_block_update", "k_spmm_dia")}
        T = np.asarray(f.H)[: len(f), : len(f)]
        print(json.dumps({"variant": var, "round": rnd, "ms_per_block_step": round(best / steps * 1e3, 4), "steps": steps,
                          "class_ms_per_sweep": cls, "T_fro": float(np.linalg.norm(T)), "T_last": float(T[-1, -1]), "normres": f.normres}), flush=True)
        for k, v in defaults.items():
            ctx.set_option(k, v)
