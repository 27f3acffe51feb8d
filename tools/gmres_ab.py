"""A/B of the config-3 Arnoldi cycle (2M-row convection-diffusion, krylovdim 60, MGS2) under option settings.
usage: python tools/gmres_ab.py key=v[,key=v] [key=v ...]   ('base' = library defaults)"""
import json
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "krylovkit.jl_amd"))
sys.path.insert(0, str(ROOT))
import krylovkit_hip as kk  # noqa: E402
from bench import convdiff_rows  # noqa: E402

import os
nx, ny, K = int(os.environ.get("KK_AB_NX", 2000)), int(os.environ.get("KK_AB_NY", 1000)), 60
N = nx * ny
ctx = kk.default_context()
op = kk.SparseOperator(convdiff_rows(nx, ny), ctx)
V = kk.DeviceBasis(N, K + 2, ctx)
x0 = kk.DeviceBasis(N, 1, ctx); x0[0].rand_(4)
it = kk.ArnoldiIterator(op, x0[0], kk.ModifiedGramSchmidt2(), capacity=K + 2)


def sweep():
    f = kk.initialize(it, V)
    for _ in range(K - 1):
        f = kk.expand_(it, f)
    return f


variants = sys.argv[1:] or ["base"]
defaults = {}
for rnd in range(2):
    for var in variants:
        opts = {} if var == "base" else {kv.split("=")[0]: float(kv.split("=")[1]) for kv in var.split(",")}
        for k, v in opts.items():
            defaults.setdefault(k, ctx.get_option(k))
            ctx.set_option(k, v)
        sweep(); ctx.sync()
        t0 = time.perf_counter()
        for _ in range(10):
            f = sweep()
        ctx.sync()
        dt = (time.perf_counter() - t0) / 10
        print(json.dumps({"rows": N, "variant": var, "round": rnd, "it_per_s": round((K - 1) / dt, 1), "ms_per_cycle": round(dt * 1e3, 3)}), flush=True)
        for k, v in defaults.items():
            ctx.set_option(k, v)
