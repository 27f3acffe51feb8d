"""The one-launch Lanczos step (k_lanczos_fstep) by vector length: wall time per expand! with the route on / off and -- from the library's
HIP-event brackets -- the kernel's own stream time per launch (which contains the launch gap: an upper bound).  Run under
`rocprofv3 --kernel-trace --stats` for the kernel's true duration.  usage: python tools/fstep_probe.py [rows ...]"""
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

ctx = kk.default_context()
sizes = [int(a) for a in sys.argv[1:]] or [1024, 10000, 102400, 200000, 249000]
import itertools
for N, K in itertools.product(sizes, (30, 100)):
    nx = int(round(N ** 0.5)); ny = N // nx
    N = nx * ny
    A = laplacian_rows(nx, ny, 0, ny)
    op = kk.SparseOperator(A, ctx, symmetric=True)
    x0 = np.random.default_rng(3).random(N)
    row = {"rows": N, "krylovdim": K}
    ctx.set_option("fused_step_max_rows", 600000)   # (measure the kernel beyond its default range too)
    for oname, orth in (("mgs2", kk.ModifiedGramSchmidt2()), ("cgs2", kk.ClassicalGramSchmidt2())):
        for fused in (1, 0):
            ctx.set_option("fused_step", fused)
            it = kk.LanczosIterator(op, x0, orth, capacity=K + 2)
            f = kk.initialize(it)
            V = f.V
            best = 1e9
            for rep in range(6):
                f = kk.initialize(it, V)
                ctx.sync()
                t0 = time.perf_counter()
                for _ in range(K - 1):
                    f = kk.expand_(it, f)
                _ = f.normres
                ctx.sync()
                best = min(best, time.perf_counter() - t0)
            row[f"{oname}_{'one_launch' if fused else 'projection_pair'}_us"] = round(best / (K - 1) * 1e6, 1)
            if fused:
                ctx.prof_reset(); ctx.prof_enable(1)
                f = kk.initialize(it, V)
                for _ in range(K - 1):
                    f = kk.expand_(it, f)
                ctx.sync(); ctx.prof_enable(0)
                ms, n = ctx.prof_get("k_lanczos_fstep")
                row[f"{oname}_kernel_bracket_us"] = round(ms / max(n, 1) * 1e3, 2)
    # Arnoldi MGS2 (the GMRES step): two passes per step
    from bench import convdiff_rows
    opa = kk.SparseOperator(convdiff_rows(nx, ny), ctx)
    for fused in (1, 0):
        ctx.set_option("fused_step", fused)
        ita = kk.ArnoldiIterator(opa, x0, kk.ModifiedGramSchmidt2(), capacity=K + 2)
        fa = kk.initialize(ita); Va = fa.V
        best = 1e9
        for rep in range(4):
            fa = kk.initialize(ita, Va)
            ctx.sync()
            t0 = time.perf_counter()
            for _ in range(K - 1):
                fa = kk.expand_(ita, fa)
            _ = fa.normres
            ctx.sync()
            best = min(best, time.perf_counter() - t0)
        row[f"arnoldi_mgs2_{'one_launch' if fused else 'ordinary'}_us"] = round(best / (K - 1) * 1e6, 1)
    ctx.set_option("fused_step", 1)
    print(json.dumps(row), flush=True)
