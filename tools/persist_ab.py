"""A/B of the headline sweep (config 2: 10M-row Lanczos, krylovdim 100) under option settings of the persistent kernel.
usage: python tools/persist_ab.py key=v[,key=v] [key=v ...]   (each argument is one variant; 'base' = library defaults)"""
import json
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "krylovkit.jl_amd"))
sys.path.insert(0, str(ROOT))
import krylovkit_hip as kk  # noqa: E402
from bench import KRYLOVDIM, NX, NY, laplacian_rows  # noqa: E402

ctx = kk.default_context()
A = laplacian_rows(NX, NY, 0, NY)
op = kk.SparseOperator(A, ctx, symmetric=True, via_csc=True)
V = kk.DeviceBasis(NX * NY, KRYLOVDIM + 2, ctx)
x0 = kk.DeviceBasis(NX * NY, 1, ctx); x0[0].rand_(3)
it = kk.LanczosIterator(op, x0[0], kk.ModifiedGramSchmidt2(), capacity=KRYLOVDIM + 2)


def sweep():
    f = kk.initialize(it, V)
    for _ in range(KRYLOVDIM - 1):
        f = kk.expand_(it, f)
    return f


variants = sys.argv[1:] or ["base"]
defaults = {}
for rnd in range(2):              # two rounds: the order of the variants must not decide
    for var in variants:
        opts = {} if var == "base" else {kv.split("=")[0]: float(kv.split("=")[1]) for kv in var.split(",")}
        for k, v in opts.items():
            defaults.setdefault(k, ctx.get_option(k))
            ctx.set_option(k, v)
        sweep(); ctx.sync()
        t0 = time.perf_counter()
        for _ in range(5):
            f = sweep()
        ctx.sync()
        dt = (time.perf_counter() - t0) / 5
        ctx.prof_reset(); ctx.prof_enable(2); sweep(); ctx.sync(); ctx.prof_enable(0)
        ms, n = ctx.prof_get("k_mgs_persist")
        print(json.dumps({"variant": var, "round": rnd, "it_per_s": round((KRYLOVDIM - 1) / dt, 1), "ms_per_sweep": round(dt * 1e3, 3),
                          "persist_ms_per_sweep": round(ms, 3), "persist_launches": n, "last_beta": f.betas[-1]}), flush=True)
        for k, v in defaults.items():
            ctx.set_option(k, v)
