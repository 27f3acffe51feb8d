"""How accurate are the MFMA Gram panels?  16 random columns of 10M rows (uniform in [0.5, 1.5): every product positive, like
the squares on the diagonal of (AX)'(AX)), G = X'X through kk_block_inner (k_block_gram), against the same sums in extended
precision on the host (math.fsum of float64 products is exact up to the final rounding; np.longdouble for the products).
usage: python tools/gram_accuracy.py [rows]"""
import json
import math
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "krylovkit.jl_amd"))
import krylovkit_hip as kk  # noqa: E402
from krylovkit_hip.factorizations import Block, block_inner  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
ctx = kk.default_context()
p = 16
B = kk.DeviceBasis(n, p, ctx)
rng = np.random.default_rng(5)
cols = []
for j in range(p):
    x = rng.random(n) + 0.5
    cols.append(x)
    B.upload(j, x)
B.length = p
G = block_inner(Block(B, 0, p), Block(B, 0, p))
rel = []
for (i, j) in [(0, 0), (3, 3), (7, 7), (15, 15), (0, 1), (5, 9)]:
    prod = cols[i].astype(np.longdouble) * cols[j].astype(np.longdouble)     # 64-bit mantissa products
    ref = float(np.sum(prod.reshape(-1, 1000).sum(axis=1)))                  # blocked long-double sum
    ref2 = math.fsum((cols[i] * cols[j]).tolist()) if n <= 2_000_000 else None
    rel.append({"entry": [i, j], "gpu": G[i, j], "ref_longdouble": ref, "relerr": abs(G[i, j] - ref) / abs(ref),
                "numpy_dot_relerr": abs(float(cols[i] @ cols[j]) - ref) / abs(ref), "fsum_relerr": None if ref2 is None else abs(ref2 - ref) / abs(ref)})
print(json.dumps({"rows": n, "entries": rel}, indent=1))
