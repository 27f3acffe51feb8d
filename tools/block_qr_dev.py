import sys, json
sys.path.insert(0, "krylovkit.jl_amd"); sys.path.insert(0, ".")
import numpy as np
import krylovkit_hip as kk
from bench import laplacian_rows
nx, ny, bs, K = 4000, 2500, 16, 100
N = nx * ny
ctx = kk.default_context()
op = kk.SparseOperator(laplacian_rows(nx, ny, 0, ny), ctx, symmetric=True)
S = kk.DeviceBasis(N, K + 3 * bs, ctx)
it = kk.BlockLanczosIterator(op, [None] * bs, K + bs)
area_b = it.maxdim + bs
for commit in (0, 1):
    ctx.set_option("block_commit", commit)
    for j in range(bs):
        S[area_b + j].rand_(100 + j)
    it.x0 = [S[area_b + j] for j in range(bs)]
    f = it.initialize(S)
    devs = []
    while len(f) < K:
        f = it.expand(f)
        devs.append(ctx.get_option("last_qr_dev"))
    print(json.dumps({"commit": commit, "last_qr_dev": devs, "commits": ctx.get_option("block_commits")}))
