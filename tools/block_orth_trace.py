"""Orthogonality of the BlockLanczos basis step by step through thick restarts (eigsolve_block, 432 rows, block size 3,
krylovdim 30): two-pass route (block_fuse=1) vs one-pass projection with Gram correction (block_fuse=5).  Columns: basis
size before the step, max|V'V - I|, Krylov relation residual, max|V'R|, normres, SpMM launches of the step, |M - M_ref|.
usage: python tools/block_orth_trace.py      (results of round 2: profiles/r02_block_onepass_orthogonality.txt)"""
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT / "krylovkit.jl_amd")); sys.path.insert(0, str(ROOT / "oracle"))
import krylovkit_hip as kk  # noqa: E402
import krylov_oracle as ko  # noqa: E402
from krylovkit_hip import factorizations as F  # noqa: E402

ctx = kk.default_context()
nx, ny, bs = 24, 18, 3
n = nx * ny
A = ko.laplacian_2d(nx, ny, shift_diag=10 * np.linspace(0, 1, n) ** 2)
Ad = A.toarray()
ev = np.linalg.eigvalsh(Ad)
rng = np.random.default_rng(9)
x0 = [rng.random(n) for _ in range(bs)]
orig = F.BlockLanczosIterator.expand


def run(fuse, maxiter):
    ctx.set_option("block_fuse", fuse); ctx.set_option("block_async", 1)
    log = []

    def expand(self, st):
        k0 = len(st)
        ctx.prof_reset(); ctx.prof_enable(1)
        st = orig(self, st)
        ctx.prof_enable(0)
        k = len(st)
        V = st.V.to_numpy()[:, :k]
        R = np.stack([st.residual()[j].get() for j in range(st.R_size)], 1)
        H = st.H[:k, :k]
        E = np.zeros((k, st.R_size)); E[k - st.R_size:, :] = np.eye(st.R_size)
        log.append(dict(k0=k0, orth=np.max(np.abs(V.T @ V - np.eye(k))), kry=np.max(np.abs(Ad @ V - V @ H - R @ E.T)),
                        vr=np.max(np.abs(V.T @ R)), nr=st.normres, spmm=ctx.prof_get("k_spmm_ell")[1], M=H[k - bs:, k - bs:].copy()))
        return st

    F.BlockLanczosIterator.expand = expand
    try:
        D, V, info = kk.eigsolve_block(kk.SparseOperator(A, ctx, symmetric=True), [x.copy() for x in x0], 5, "SR",
                                       kk.BlockLanczos(krylovdim=30, tol=1e-10, maxiter=maxiter))
    finally:
        F.BlockLanczosIterator.expand = orig
    return log, D, info


l1, _, _ = run(1, 12)
l5, _, _ = run(5, 12)
for i, (a, b) in enumerate(zip(l1, l5)):
    print(i, "k0", a["k0"], b["k0"], "| orth %.1e %.1e | kry %.1e %.1e | V'R %.1e %.1e | nr %.6e %.6e | spmm %d %d | dM %.1e"
          % (a["orth"], b["orth"], a["kry"], b["kry"], a["vr"], b["vr"], a["nr"], b["nr"], a["spmm"], b["spmm"],
             np.max(np.abs(a["M"] - b["M"])) if a["M"].shape == b["M"].shape else -1), flush=True)
for fuse in (1, 5):
    _, D, info = run(fuse, 200)
    print("block_fuse", fuse, "converged", info.converged, "numiter", info.numiter, "numops", info.numops,
          "eigenvalue error", np.max(np.abs(D[:5] - ev[:5])), flush=True)
ctx.set_option("block_fuse", 5)
