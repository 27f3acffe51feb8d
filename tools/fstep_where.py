"""Where does the one-launch step spend its time?  Timing-only variants of k_lanczos_fstep (built by tools/fstep_where.sh with -DKK_FS_EXP=bits:
1 no wave reductions, 2 no loads of V, 4 no waiting for the other blocks' partials; 8 keeps the numbers finite) against the product build.
The variants compute garbage: only the clock is read.  usage: KRYLOV_HIP_LIB=<variant .so> python tools/fstep_where.py"""
import json
import os
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "krylovkit.jl_amd"))
sys.path.insert(0, str(ROOT))
import krylovkit_hip as kk  # noqa: E402
from bench import laplacian_rows  # noqa: E402

ctx = kk.default_context()
if os.environ.get("FSTEP_THREADS"):
    ctx.set_option("fstep_threads", int(os.environ["FSTEP_THREADS"]))
if os.environ.get("FSTEP_BLOCKS"):
    ctx.set_option("fstep_blocks", int(os.environ["FSTEP_BLOCKS"]))
out = {"lib": os.path.basename(os.environ.get("KRYLOV_HIP_LIB", "product")), "threads": os.environ.get("FSTEP_THREADS"), "blocks": os.environ.get("FSTEP_BLOCKS")}
for N, K in ((1024, 30), (102400, 30), (102400, 100)):
    nx = int(round(N ** 0.5)); ny = N // nx
    A = laplacian_rows(nx, ny, 0, ny)
    op = kk.SparseOperator(A, ctx, symmetric=True)
    x0 = np.random.default_rng(3).random(nx * ny)
    ctx.set_option("fused_step_m_limit", -1)
    it = kk.LanczosIterator(op, x0, kk.ModifiedGramSchmidt2(), capacity=K + 2)
    f = kk.initialize(it); V = f.V
    best = 1e9
    for rep in range(5):
        f = kk.initialize(it, V)
        ctx.sync()
        t0 = time.perf_counter()
        try:
            for _ in range(K - 1):
                f = kk.expand_(it, f)
        except Exception as e:          # (a variant's garbage may trip a zero-norm check: report what was measured)
            out[f"error_{N}_{K}"] = str(e)[:80]
            break
        ctx.sync()
        best = min(best, time.perf_counter() - t0)
    out[f"us_per_expand_rows{N}_k{K}"] = round(best / (K - 1) * 1e6, 1)
print(json.dumps(out), flush=True)
