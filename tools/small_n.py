"""Launch-bound regime: time per Lanczos expand! (krylovdim 30, sweeps of 29 expands) and per CG iteration on small grids,
GPU path vs the C twin of the reference's CPU path on the host cores.  usage: python tools/small_n.py [out.jsonl]"""
import json
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "krylovkit.jl_amd"))
sys.path.insert(0, str(ROOT / "tests"))
sys.path.insert(0, str(ROOT))
import krylovkit_hip as kk  # noqa: E402
import cpu_ref_lib  # noqa: E402
from bench import laplacian_rows  # noqa: E402

ctx = kk.default_context()
lib = cpu_ref_lib.load()
K = 30
out = open(sys.argv[1], "w") if len(sys.argv) > 1 else None
for nx, ny in [(32, 32), (100, 100), (320, 320), (1000, 1000), (2000, 2000)]:
    N = nx * ny
    A = laplacian_rows(nx, ny, 0, ny)
    op = kk.SparseOperator(A, ctx, symmetric=True)
    x0 = np.random.default_rng(3).random(N)
    row = {"rows": N}
    for oname, orth, ocode, fused in [("mgs2", kk.ModifiedGramSchmidt2(), 3, 1), ("cgs2", kk.ClassicalGramSchmidt2(), 2, 1),
                                      ("mgs2_projection_pair", kk.ModifiedGramSchmidt2(), 3, 0), ("cgs2_projection_pair", kk.ClassicalGramSchmidt2(), 2, 0)]:
        ctx.set_option("fused_step", fused)     # round 6: the one-launch step is the default below 250 k rows; the projection pair of rounds 1-5 next to it
        it = kk.LanczosIterator(op, x0, orth, capacity=K + 2)
        f = kk.initialize(it)
        V = f.V
        best = 1e9
        for rep in range(5):
            f = kk.initialize(it, V)
            ctx.sync()
            t0 = time.perf_counter()
            for _ in range(K - 1):
                f = kk.expand_(it, f)
            _ = f.normres
            ctx.sync()
            best = min(best, time.perf_counter() - t0)
        row[f"gpu_{oname}_us_per_expand"] = round(best / (K - 1) * 1e6, 1)
    row["fstep_launches"] = int(ctx.get_option("fstep_launches"))
    ctx.set_option("fused_step", 1)
    for nt in (1, cpu_ref_lib.usable_threads()):
        best = 1e9
        for rep in range(3):
            t0 = time.perf_counter()
            cpu_ref_lib.run_lanczos(lib, A, x0, K - 1, 3, nthreads=nt)
            best = min(best, time.perf_counter() - t0)
        row[f"cpu_mgs2_{nt}t_us_per_expand"] = round(best / K * 1e6, 1)
    b = np.random.default_rng(4).random(N)
    S = kk.SparseOperator(A, ctx, symmetric=True)
    for rep in range(3):
        ctx.sync()
        t0 = time.perf_counter()
        x, info = kk.linsolve_cg(S, b, None, kk.CG(60, 1e-300), 1.0, 1.0)
        ctx.sync()
        dt = time.perf_counter() - t0
    row["gpu_cg_us_per_iteration"] = round(dt / max(1, info.numiter) * 1e6, 1)
    row["cg_iterations"] = info.numiter
    print(json.dumps(row), flush=True)
    if out:
        out.write(json.dumps(row) + "\n")
