"""Vendor yardstick for the one kernel far below its roofline (config 4: y = A x and y = A'u on the 5M x 1M random
sparse map, 20 nnz/row): rocsparse_spmv on the IDENTICAL matrix, every CSR algorithm it offers, next to libkrylov_hip's
column-tiled SELL kernels.  Measurement tool only (SURVEY.md section 0 allows rocSPARSE "as cross-checks"): the product
never links or loads rocSPARSE.   usage: python tools/rocsparse_yardstick.py [--rows 5000000 --cols 1000000] > out.json"""
import argparse
import ctypes as C
import json
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "krylovkit.jl_amd"))
sys.path.insert(0, str(ROOT))

ap = argparse.ArgumentParser()
ap.add_argument("--rows", type=int, default=5_000_000)
ap.add_argument("--cols", type=int, default=1_000_000)
ap.add_argument("--reps", type=int, default=20)
args = ap.parse_args()

import torch                      # device buffers + events for the rocSPARSE side only
import krylovkit_hip as kk
from bench import gkl_rows

m, n, per = args.rows, args.cols, 20
A = gkl_rows(m, n, per, 0, m)
A.sort_indices()
At = A.T.tocsr()
At.sort_indices()
nnz = A.nnz
alg_bytes = 12 * nnz + 4 * (m + 1) + 8 * (m + n)       # CSR values + int32 columns + row pointers + both vectors once
out = {"matrix": f"{m} x {n}, {nnz} stored entries ({per} per row, duplicates summed), bench.gkl_rows seeds", "reps": args.reps,
       "algorithmic_bytes_per_apply": alg_bytes, "rocsparse": [], "libkrylov_hip": {}}

rs = C.CDLL("/opt/rocm/lib/librocsparse.so")
vp = C.c_void_p
handle = vp()
assert rs.rocsparse_create_handle(C.byref(handle)) == 0
rs.rocsparse_create_csr_descr.argtypes = [C.POINTER(vp), C.c_int64, C.c_int64, C.c_int64, vp, vp, vp, C.c_int, C.c_int, C.c_int, C.c_int]
rs.rocsparse_create_dnvec_descr.argtypes = [C.POINTER(vp), C.c_int64, vp, C.c_int]
rs.rocsparse_spmv.argtypes = [vp, C.c_int, vp, vp, vp, vp, vp, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_size_t), vp]
I32, F64, BASE0, OP_N, OP_T = 2, 152, 0, 111, 112
dev = torch.device("cuda", 0)


def csr_on_device(M):
    return (torch.from_numpy(M.indptr.astype(np.int32)).to(dev), torch.from_numpy(M.indices.astype(np.int32)).to(dev),
            torch.from_numpy(M.data.astype(np.float64)).to(dev))


def time_spmv(M, op, alg, xlen, ylen, label):
    rp, ci, va = csr_on_device(M)
    mat, xd, yd = vp(), vp(), vp()
    x = torch.randn(xlen, dtype=torch.float64, device=dev)
    y = torch.zeros(ylen, dtype=torch.float64, device=dev)
    assert rs.rocsparse_create_csr_descr(C.byref(mat), M.shape[0], M.shape[1], M.nnz, rp.data_ptr(), ci.data_ptr(), va.data_ptr(), I32, I32, BASE0, F64) == 0
    assert rs.rocsparse_create_dnvec_descr(C.byref(xd), xlen, x.data_ptr(), F64) == 0
    assert rs.rocsparse_create_dnvec_descr(C.byref(yd), ylen, y.data_ptr(), F64) == 0
    alpha, beta = C.c_double(1.0), C.c_double(0.0)
    bs = C.c_size_t(0)
    st = rs.rocsparse_spmv(handle, op, C.byref(alpha), mat, xd, C.byref(beta), yd, F64, alg, 1, C.byref(bs), None)
    if st != 0:
        return {"label": label, "status": st}
    buf = torch.empty(max(bs.value, 8), dtype=torch.uint8, device=dev)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    st = rs.rocsparse_spmv(handle, op, C.byref(alpha), mat, xd, C.byref(beta), yd, F64, alg, 2, C.byref(bs), buf.data_ptr())
    e1.record(); torch.cuda.synchronize()
    pre_ms = e0.elapsed_time(e1)
    if st != 0:
        return {"label": label, "status": st}
    for _ in range(3):
        rs.rocsparse_spmv(handle, op, C.byref(alpha), mat, xd, C.byref(beta), yd, F64, alg, 3, C.byref(bs), buf.data_ptr())
    torch.cuda.synchronize()
    e0.record()
    for _ in range(args.reps):
        st = rs.rocsparse_spmv(handle, op, C.byref(alpha), mat, xd, C.byref(beta), yd, F64, alg, 3, C.byref(bs), buf.data_ptr())
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / args.reps
    # correctness of what was timed
    xh = x.cpu().numpy()
    ref = (M @ xh) if op == OP_N else (M.T @ xh)
    err = float(np.max(np.abs(y.cpu().numpy() - ref)) / np.max(np.abs(ref)))
    return {"label": label, "status": st, "ms_per_apply": round(ms, 4), "preprocess_ms": round(pre_ms, 2), "buffer_bytes": bs.value,
            "GBps_algorithmic": round(alg_bytes / ms / 1e6, 1), "frac_of_8TBps": round(alg_bytes / ms / 1e6 / 8000, 4), "max_relerr": err}


ALGS = {"default": 0, "csr_adaptive": 2, "csr_rowsplit": 3, "csr_lrb": 7, "csr_nnzsplit": 8}
for name, alg in ALGS.items():
    out["rocsparse"].append(time_spmv(A, OP_N, alg, n, m, f"A x, CSR(A), {name}"))
for name, alg in ALGS.items():
    out["rocsparse"].append(time_spmv(At, OP_N, alg, m, n, f"A'u as (A')x on an explicit CSR(A'), {name}"))
out["rocsparse"].append(time_spmv(A, OP_T, 0, m, n, "A'u, CSR(A) with operation_transpose, default"))

# libkrylov_hip on the same matrix (kk_csr_create picks the column-tiled SELL format for it)
ctx = kk.default_context()
op = kk.SparseOperator(A, ctx)
xb, yb = kk.DeviceBasis(n, 2, ctx), kk.DeviceBasis(m, 2, ctx)
xh = np.random.default_rng(1).standard_normal(n)
xb.upload(0, xh)
op.apply(xb[0], yb[0])
err_n = float(np.max(np.abs(yb[0].get() - A @ xh)) / np.max(np.abs(A @ xh)))
uh = np.random.default_rng(2).standard_normal(m)
yb.upload(1, uh)
op.apply_adjoint(yb[1], xb[1])
err_t = float(np.max(np.abs(xb[1].get() - At @ uh)) / np.max(np.abs(At @ uh)))
for label, fn, err in (("A x", lambda: op.apply(xb[0], yb[0]), err_n), ("A'u", lambda: op.apply_adjoint(yb[1], xb[1]), err_t)):
    for _ in range(3):
        fn()
    ctx.sync(); ctx.timer_start()
    for _ in range(args.reps):
        fn()
    ms = ctx.timer_stop() / args.reps
    out["libkrylov_hip"][label] = {"ms_per_apply": round(ms, 4), "GBps_algorithmic": round(alg_bytes / ms / 1e6, 1),
                                   "frac_of_8TBps": round(alg_bytes / ms / 1e6 / 8000, 4), "max_relerr": err, "format": op.info()["format"],
                                   "device_bytes": op.info()["device_bytes"]}
print(json.dumps(out, indent=1))
