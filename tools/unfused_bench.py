"""How much the fused kk_lanczos_expand buys over the un-fused sequence on the same device primitives: the headline sweep
(10M-row Laplacian, krylovdim 100) once with the SparseOperator (fused expand!) and once with the same operator wrapped
as a FunctionOperator (one call of f, L1 verbs, fused orthogonalisation passes).  usage: python tools/unfused_bench.py"""
import json
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "krylovkit.jl_amd"))
sys.path.insert(0, str(ROOT))
import krylovkit_hip as kk  # noqa: E402
from bench import laplacian_rows  # noqa: E402

NX, NY, K = 4000, 2500, 100
ctx = kk.default_context()
A = laplacian_rows(NX, NY, 0, NY)
op = kk.SparseOperator(A, ctx, symmetric=True, via_csc=True)
fop = kk.FunctionOperator(lambda x, y: op.apply(x, y), NX * NY, ctx, symmetric=True)
x0 = kk.DeviceBasis(NX * NY, 1, ctx)
x0[0].rand_(3)
out = {}
for orth in (kk.ModifiedGramSchmidt2(), kk.ClassicalGramSchmidt2()):
    for name, o in (("fused", op), ("unfused", fop)):
        V = kk.DeviceBasis(NX * NY, K + 2, ctx)
        it = kk.LanczosIterator(o, x0[0], orth, capacity=K + 2)
        best = 1e9
        for rep in range(3):
            ctx.sync(); t0 = time.perf_counter()
            f = kk.initialize(it, V)
            for _ in range(K - 1):
                f = kk.expand_(it, f)
            ctx.sync(); best = min(best, time.perf_counter() - t0)
        out[f"{orth.name}_{name}_it_per_s"] = round((K - 1) / best, 1)
        del V
print(json.dumps(out))
