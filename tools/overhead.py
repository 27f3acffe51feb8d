"""Per-expand host/launch overhead: Lanczos sweep on a tiny problem (kernels ~ empty)."""
import sys, time
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "krylovkit.jl_amd")); sys.path.insert(0, str(ROOT))
import krylovkit_hip as kk
from bench import laplacian_rows
ctx = kk.default_context()
A = laplacian_rows(32, 32, 0, 32)
op = kk.SparseOperator(A, ctx, symmetric=True)
for orth in (kk.ClassicalGramSchmidt2(), kk.ModifiedGramSchmidt2(), kk.ClassicalGramSchmidt()):
    it = kk.LanczosIterator(op, np.random.default_rng(0).random(1024), orth, capacity=102)
    f = kk.initialize(it); V = f.V
    for rep in range(3):
        f = kk.initialize(it, V); ctx.sync(); t0 = time.perf_counter()
        for _ in range(99):
            f = kk.expand_(it, f)
        ctx.sync(); dt = time.perf_counter() - t0
    print(orth.name, f"{dt / 99 * 1e6:.1f} us per expand")
