"""An operator with MORE THAN 2^31 stored entries on one GPU (DESIGN.md limits): 5-point Laplacian on an nx x ny grid with
5 nx ny - 2 (nx + ny) >= 2^31 entries, handed over through kk_csr_create (int64 row pointers), applied by the constant-
coefficient diagonal kernel, the stored-diagonal kernel and the ELL gather kernel, then a short Lanczos run (vectors of
> 2^31 bytes).  Checks: A * ones against the closed form, the three kernels against each other, the Lanczos relation.
usage: python tools/large_nnz.py [nx ny] [out.json]     (needs ~130 GB of host memory and ~90 GB of HBM)"""
import json
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "krylovkit.jl_amd"))
import krylovkit_hip as kk  # noqa: E402

args = [a for a in sys.argv[1:] if not a.endswith(".json")]
outp = [a for a in sys.argv[1:] if a.endswith(".json")]
nx, ny = (int(args[0]), int(args[1])) if len(args) >= 2 else (20000, 21500)
N = nx * ny
need_gb = 5 * N * 60 / 1e9
avail_gb = 0.0
for ln in open("/proc/meminfo"):
    if ln.startswith("MemAvailable"):
        avail_gb = int(ln.split()[1]) / 1e6
try:
    cg = Path("/sys/fs/cgroup/memory.max").read_text().strip()
    if cg != "max":
        avail_gb = min(avail_gb, int(cg) / 1e9)
except Exception:
    pass
print(f"# grid {nx} x {ny}: {N} rows, host memory available {avail_gb:.0f} GB, estimated need {need_gb:.0f} GB", flush=True)
if avail_gb < 1.3 * need_gb:
    print(json.dumps({"skipped": f"host memory {avail_gb:.0f} GB < {1.3 * need_gb:.0f} GB"}))
    sys.exit(0)

t0 = time.time()
i = np.arange(N, dtype=np.int64)
ix = (i % nx).astype(np.int32)
has = np.empty((N, 5), dtype=bool)          # -nx, -1, 0, +1, +nx   (ascending column order inside a row)
has[:, 0] = i >= nx
has[:, 1] = ix > 0
has[:, 2] = True
has[:, 3] = ix < nx - 1
has[:, 4] = i < N - nx
del ix
cnt = has.sum(axis=1, dtype=np.int64)
rowptr = np.zeros(N + 1, dtype=np.int64)
np.cumsum(cnt, out=rowptr[1:])
nnz = int(rowptr[-1])
expected_ones = (4.0 - (cnt - 1)).astype(np.float64)       # A * ones: 4 - number of neighbours
del cnt
col = np.empty(nnz, dtype=np.int32)
val = np.empty(nnz, dtype=np.float64)
offs = np.array([-nx, -1, 0, 1, nx], dtype=np.int64)
flat = has.ravel()
pos = np.flatnonzero(flat)                                   # positions in the (N, 5) image, row-major = CSR order
del flat
col[:] = (pos // 5 + offs[pos % 5]).astype(np.int32)
val[:] = np.where(pos % 5 == 2, 4.0, -1.0)
del pos, has, i
print(f"# nnz = {nnz} (2^31 = {2**31}) built in {time.time() - t0:.0f} s", flush=True)
assert nnz >= 2 ** 31

ctx = kk.default_context()
t0 = time.time()
op = kk.SparseOperator.from_csr_arrays((N, N), rowptr, col, val, ctx, symmetric=True)
del col, val, rowptr
info = op.info()
print(f"# uploaded in {time.time() - t0:.0f} s: {info}", flush=True)
res = {"grid": [nx, ny], "rows": N, "nnz": nnz, "nnz_over_2_31": nnz / 2 ** 31, "vector_bytes": 8 * N, "op_info": info}

K = 6
W = kk.DeviceBasis(N, 4, ctx)
x, y0, y1, y2 = (W[j] for j in range(4))
x.set(np.ones(N))
variants = [("const_stencil", dict(spmv_dia=1, spmv_dia_const=1)), ("stored_diagonals", dict(spmv_dia=1, spmv_dia_const=0)),
            ("ell_gather", dict(spmv_dia=0, spmv_dia_const=0))]
ys = [y0, y1, y2]
for (name, opts), y in zip(variants, ys):
    for k, v in opts.items():
        ctx.set_option(k, v)
    op.apply(x, y)
    ctx.sync()
    t0 = time.perf_counter()
    for _ in range(3):
        op.apply(x, y)
    ctx.sync()
    res[f"apply_ms_{name}"] = round((time.perf_counter() - t0) / 3 * 1e3, 3)
got = y0.get()
res["ones_max_abs_err"] = float(np.max(np.abs(got - expected_ones)))
del got, expected_ones
x.rand_(11)
for (name, opts), y in zip(variants, ys):
    for k, v in opts.items():
        ctx.set_option(k, v)
    op.apply(x, y)
ref = y2.get()
nrm = float(np.linalg.norm(ref))
res["const_vs_gather_rel"] = float(np.linalg.norm(y0.get() - ref) / nrm)
res["diagonals_vs_gather_rel"] = float(np.linalg.norm(y1.get() - ref) / nrm)
del ref
del W, x, y0, y1, y2, ys
ctx.set_option("spmv_dia", 1)
ctx.set_option("spmv_dia_const", 1)

x0 = kk.DeviceBasis(N, 1, ctx)
x0[0].rand_(3)
for oname, orth in [("mgs2", kk.ModifiedGramSchmidt2()), ("cgs2", kk.ClassicalGramSchmidt2())]:
    V = kk.DeviceBasis(N, K + 2, ctx)
    it = kk.LanczosIterator(op, x0[0], orth, capacity=K + 2)
    f = kk.initialize(it, V)
    t0 = time.perf_counter()
    for _ in range(K - 1):
        f = kk.expand_(it, f)
    ctx.sync()
    dt = time.perf_counter() - t0
    al, be = np.array(f.alphas), np.array(f.betas)
    # V'V = I and the three-term relation A v_j = beta_{j-1} v_{j-1} + alpha_j v_j + beta_j v_{j+1}, on the device
    G = np.array([[V[a].inner(V[b]) for b in range(K)] for a in range(K)])
    T = kk.DeviceBasis(N, 1, ctx)
    rel = 0.0
    for j in range(K - 1):
        op.apply(V[j], T[0])
        T[0].add_(V[j], -al[j])
        if j > 0:
            T[0].add_(V[j - 1], -be[j - 1])
        T[0].add_(V[j + 1], -be[j])
        rel = max(rel, T[0].norm())
    res[oname] = {"expand_ms": round(dt / (K - 1) * 1e3, 2), "alphas": al.tolist(), "betas": be.tolist(),
                  "orthogonality": float(np.max(np.abs(G - np.eye(K)))), "lanczos_relation_residual": rel}
    del V, T, f, it
res["alphas_mgs2_vs_cgs2"] = float(np.max(np.abs(np.array(res["mgs2"]["alphas"]) - np.array(res["cgs2"]["alphas"]))))
ok = (res["ones_max_abs_err"] == 0.0 and res["const_vs_gather_rel"] < 1e-14 and res["diagonals_vs_gather_rel"] < 1e-14
      and all(res[o]["orthogonality"] < 1e-12 and res[o]["lanczos_relation_residual"] < 1e-11 for o in ("mgs2", "cgs2")))
res["ok"] = bool(ok)
print(json.dumps(res), flush=True)
if outp:
    Path(outp[0]).write_text(json.dumps(res, indent=1) + "\n")
sys.exit(0 if ok else 1)
