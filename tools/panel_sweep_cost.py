"""Per-vector cost of an MGS sweep on short vectors: the persistent panel kernel (csrc/kk_kernels_panel.hip) at every panel
width it supports, the register-resident strict kernel and the low-synchronisation projection pair, event-timed per launch.
usage: python tools/panel_sweep_cost.py [rows ...]"""
import json
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "krylovkit.jl_amd"))
import krylovkit_hip as kk  # noqa: E402

rows = [int(a) for a in sys.argv[1:]] or [500_000, 1_000_000, 2_000_000, 3_600_000]
ctx = kk.Context(0)
for n in rows:
    for m in (16, 48):
        B = kk.DeviceBasis(n, m + 2, ctx)
        for j in range(m + 1):
            B[j].rand_(7 + j)
        B.length = m
        # make the basis orthonormal enough for the timings to be about streams, not values
        for j in range(m):
            B.orthonormalize(B[j], kk.ModifiedGramSchmidt2(), 0, j)
        variants = [("panel auto width", dict(mgs_mode=2, mgs_panel=1, panel_width=0, panel_min_rows=0)),
                    ("panel width 1 (strict order)", dict(mgs_mode=0, mgs_panel=1, panel_width=0)),
                    ("panel width 2", dict(mgs_mode=2, mgs_panel=1, panel_width=2, panel_min_rows=0)),
                    ("register-resident strict kernel (k_mgs_persist)", dict(mgs_mode=0, mgs_panel=0)),
                    ("low-sync pair (k_project + k_unproject)", dict(mgs_mode=1, mgs_panel=0))]
        for name, opts in variants:
            for k, v in opts.items():
                ctx.set_option(k, v)
            reps = 20
            for it in range(2):
                if it == 1:
                    ctx.prof_reset(); ctx.prof_enable(1)
                for _ in range(reps):
                    B[m].rand_(99)
                    B.orthogonalize(B[m], kk.ModifiedGramSchmidt())
            ctx.prof_enable(0)
            tot = sum(ctx.prof_get(c)[0] for c in ("k_mgs_panel", "k_mgs_persist", "k_project", "k_unproject", "k_mgs_step"))
            print(json.dumps({"rows": n, "basis_vectors": m, "variant": name, "kernel_us_per_sweep": round(tot / reps * 1e3, 2),
                              "us_per_vector": round(tot / reps * 1e3 / m, 3),
                              "stream_us_per_vector_at_8TBps": round(8.0 * n / 8e12 * 1e6, 3),
                              "launches": {c: ctx.prof_get(c)[1] for c in ("k_mgs_panel", "k_mgs_persist", "k_project", "k_unproject") if ctx.prof_get(c)[1]}}), flush=True)
        B.free()
ctx.set_option("mgs_mode", 2)
