"""k_block_update_mfma against the VALU kernel (k_block_update_lds) through kk_block_update: results to rounding, column norms, time
per launch at N = 10M for the basis widths of BASELINE.json configs[4].  usage: python tools/bu_mfma_check.py"""
import json
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "krylovkit.jl_amd"))
sys.path.insert(0, str(ROOT))
import krylovkit_hip as kk  # noqa: E402
from krylovkit_hip._lib import check, c_dp  # noqa: E402

N, bs = 4000 * 2500, 16
ctx = kk.default_context()
lib = ctx._lib
S = kk.DeviceBasis(N, 160, ctx)
for j in range(160):
    S[j].rand_(j + 1)


def timeit(fn, reps=5):
    fn(); ctx.sync()
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        ctx.sync()
        best = min(best, (time.perf_counter() - t0) / reps)
    return best


VARIANTS = [int(a) for a in sys.argv[1:]] or [1]
for m in (16, 48, 112, 37):
    for nb in (16, 7):
        coef = np.asfortranarray(np.random.default_rng(m).standard_normal((m, nb)) * 1e-2)
        for beta in (0.0, 1.0):
            res = {}
            for mf in [0] + VARIANTS:
                ctx.set_option("bu_mfma", mf)
                for j in range(nb):
                    S[130 + j].rand_(500 + j)
                nrm = np.zeros(nb)

                def f():
                    check(lib.kk_block_update(S.handle, 130, nb, S.handle, 0, m, coef.ctypes.data_as(c_dp), m, -1.0, beta, nrm.ctypes.data_as(c_dp)))
                f(); ctx.sync()
                cols = np.stack([S[130 + j].get()[::7] for j in range(nb)], 1)
                res[mf] = (cols, nrm.copy())
                if nb == 16 and beta == 1.0:
                    for j in range(nb):
                        S[130 + j].rand_(500 + j)
                    dt = timeit(f)
                    res[mf] += (dt,)
            d = max(float(np.max(np.abs(res[0][0] - res[v][0])) / np.max(np.abs(res[0][0]))) for v in VARIANTS)
            dn = max(float(np.max(np.abs(res[0][1] - res[v][1]) / np.abs(res[0][1]))) for v in VARIANTS)
            row = {"m": m, "nb": nb, "beta": beta, "max_rel_diff": d, "norm_rel_diff": dn}
            if len(res[0]) > 2:
                byts = (8 * m + 256) * N
                row.update({"valu_TBps": round(byts / res[0][2] / 1e12, 2)})
                for v in VARIANTS:
                    row[f"mfma{v}_TBps"] = round(byts / res[v][2] / 1e12, 2)
            print(json.dumps(row), flush=True)
            assert d < 1e-13 and dn < 1e-12, row
