"""One rank of a world-N run of libkrylov_hip's NATIVE multi-rank path (kk_comm_init, kk_csr_create_sharded*, the
all-reduces at every finalize site) with every rank on the box's single GPU: KK_RCCL_LIB points at the test-only RCCL
stand-in tests/fake_rccl/libfake_rccl.so, which moves the collectives through a mapped file.  Driven by
tests/test_gpu_world2.py:   python world2_worker.py <scenario> <rank> <world> <rendezvous dir>
Every scenario builds the same global problem on every rank (seeded), keeps its own row block, runs the ORDINARY
iterators / drivers of krylovkit_hip on it and compares with the serial oracle (oracle/krylov_oracle.py) run on the
global problem.  TEST INFRASTRUCTURE."""
import json
import sys
import time
from pathlib import Path

import numpy as np
import scipy.sparse as sp

HERE = Path(__file__).resolve().parent
ROOT = HERE.parent
sys.path.insert(0, str(ROOT / "krylovkit.jl_amd"))
sys.path.insert(0, str(ROOT / "oracle"))

import krylovkit_hip as kk            # noqa: E402
from krylovkit_hip import dist as kd  # noqa: E402
import krylov_oracle as ko            # noqa: E402

scenario, rank, world, rdv = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), Path(sys.argv[4])


def file_bcast(obj):
    """rank 0's 128-byte communicator id reaches the other ranks through the rendezvous directory"""
    f = rdv / "comm_id"
    if rank == 0:
        (rdv / "comm_id.tmp").write_bytes(obj)
        (rdv / "comm_id.tmp").rename(f)
        return obj
    t0 = time.time()
    while not f.exists():
        assert time.time() - t0 < 120, "rank 0 never published the communicator id"
        time.sleep(0.01)
    return f.read_bytes()


def relerr(a, b):
    a, b = np.asarray(a, float), np.asarray(b, float)
    return float(np.max(np.abs(a - b) / np.maximum(np.abs(b), 1e-300)))


def orth_pairs():
    return [(kk.ClassicalGramSchmidt(), ko.CGS), (kk.ModifiedGramSchmidt(), ko.MGS), (kk.ClassicalGramSchmidt2(), ko.CGS2),
            (kk.ModifiedGramSchmidt2(), ko.MGS2), (kk.ClassicalGramSchmidtIR(0.75), ko.CGSIR(0.75)),
            (kk.ModifiedGramSchmidtIR(0.75), ko.MGSIR(0.75))]


def uneven_offsets(n, seed):
    """row blocks of clearly different sizes whose boundaries fall anywhere (not on grid lines)"""
    cuts = np.sort(np.random.default_rng(seed).choice(np.arange(n // (4 * world), n - n // (4 * world)), world - 1, replace=False))
    return np.concatenate([[0], cuts, [n]]).astype(np.int64)


ctx = kk.Context(0)
comm = kd.NativeComm(ctx, rank, world, file_bcast)
info = comm.info()
assert info["rank"] == rank and info["world"] == world and info["rccl_version"] == 29999, info   # the stand-in, not RCCL
report = {"scenario": scenario, "rank": rank}


def gather_rows(name, local):
    """assemble a row-sharded array on every rank through the rendezvous directory (checks only, not the data path)"""
    np.save(rdv / f"{name}.{rank}.tmp.npy", local)
    (rdv / f"{name}.{rank}.tmp.npy").rename(rdv / f"{name}.{rank}.npy")
    parts = []
    for q in range(world):
        f = rdv / f"{name}.{q}.npy"
        t0 = time.time()
        while not f.exists():
            assert time.time() - t0 < 120, f"rank {q} never wrote {name}"
            time.sleep(0.01)
        parts.append(np.load(f))
    return np.concatenate(parts, axis=0)


def lanczos_against_oracle(A, part, x0, steps, tag, check_counts=True):
    op = kd.NativeShardedOperator(A[part.lo:part.hi], part, ctx, symmetric=True)
    assert comm.stats()["p2p_groups"] > 0, "the ghost plan negotiation must have used the grouped send / recv"
    for mgs_mode in (0, 1):
        ctx.set_option("mgs_mode", mgs_mode)
        for dev, ref in orth_pairs():
            it = kk.LanczosIterator(op, x0[part.lo:part.hi], dev, capacity=steps + 3)
            f = kk.initialize(it)
            oit = ko.LanczosIterator(A, x0.copy(), ref)
            of = ko.lanczos_initialize(oit)
            s0 = comm.stats()
            for _ in range(steps):
                f = kk.expand_(it, f)
                of = ko.lanczos_expand(oit, of)
            s1 = comm.stats()
            tol = 1e-10 if dev.is_reorth else 1e-6
            ea, eb = relerr(f.alphas, of.alphas), relerr(f.betas, of.betas)
            assert ea < tol and eb < tol, (tag, dev.name, mgs_mode, ea, eb)
            if check_counts:
                assert (s1["p2p_groups"] - s0["p2p_groups"]) >= steps, (tag, dev.name)      # one ghost exchange per apply (+ speculation)
                if dev.name == "cgs2" or (dev.name == "mgs2" and mgs_mode == 1):
                    assert (s1["allreduce"] - s0["allreduce"]) / steps == 2.0, (tag, dev.name, s0, s1)
            if dev.is_reorth:
                V = gather_rows(f"V_{tag}_{dev.name}_{mgs_mode}", f.V.to_numpy())
                assert np.max(np.abs(V.T @ V - np.eye(V.shape[1]))) < 1e-12, (tag, dev.name)
                Vo = np.stack(of.V, axis=1) if isinstance(of.V, list) else np.asarray(of.V)
                if Vo.shape == V.shape:
                    sgn = np.sign(np.sum(V * Vo, axis=0))
                    assert np.max(np.abs(V * sgn - Vo)) < 1e-8, (tag, dev.name)
            report[f"{tag}.{dev.name}.mgs{mgs_mode}"] = [ea, eb]
    ctx.set_option("mgs_mode", 2)
    return op


if scenario == "lanczos_grid":
    # config-2 shape: 5-point stencil split along grid lines -> diagonal kernels on the interior, ghost strips at the seams
    nx, ny = 70, 64 * world        # far offset >= 64 and >= 4096 rows per shard: the stencil detection gives the dense-diagonal format
    n = nx * ny
    A = ko.laplacian_2d(nx, ny, shift_diag=10 * np.linspace(0, 1, n) ** 2)
    x0 = np.random.default_rng(3).random(n)
    part = kd.Partition.even(n, world, rank, align=nx)
    op = lanczos_against_oracle(A, part, x0, 25, "grid")
    report["format"] = op.info()["format"]
    assert report["format"].startswith("ELL+DIA"), report["format"]
elif scenario == "lanczos_random":
    # random symmetric sparsity, uneven split: every rank needs scattered entries of every other one
    n = 3000
    R = sp.random(n, n, density=4.0 / n, random_state=np.random.default_rng(11), format="csr")
    A = (R + R.T + sp.diags(4.0 + np.arange(n) / n)).tocsr()
    x0 = np.random.default_rng(4).random(n)
    part = kd.Partition(uneven_offsets(n, 5), rank)
    op = lanczos_against_oracle(A, part, x0, 20, "random")
    # plain applies, affine form, block applies (ONE grouped exchange for all columns)
    nl = part.n_local
    B = kk.DeviceBasis(nl, 40, ctx)
    X = np.random.default_rng(12).standard_normal((n, 16))
    for j in range(16):
        B.upload(j, X[part.lo:part.hi, j])
    op.apply(B[3], B[20])
    np.testing.assert_allclose(B[20].get(), (A @ X[:, 3])[part.lo:part.hi], rtol=0, atol=1e-12)
    op.apply_affine(B[3], B[21], 0.7, -0.4)
    np.testing.assert_allclose(B[21].get(), (0.7 * X[:, 3] - 0.4 * (A @ X[:, 3]))[part.lo:part.hi], rtol=0, atol=1e-12)
    from krylovkit_hip._lib import check
    for nb in (16, 3, 11):
        p0 = comm.stats()["p2p_groups"]
        check(ctx._lib.kk_block_apply(op.handle, B.handle, 0, B.handle, 20, nb))
        assert comm.stats()["p2p_groups"] - p0 == 1   # ONE exchange for the block on every rank, whatever its local storage format
        Y = np.stack([B.download(20 + j) for j in range(nb)], 1)
        np.testing.assert_allclose(Y, (A @ X[:, :nb])[part.lo:part.hi], rtol=0, atol=1e-11)
    # inner products / norms through the L1 verbs are global
    assert abs(B[0].inner(B[1]) - X[:, 0] @ X[:, 1]) < 1e-10 and abs(B[2].norm() - np.linalg.norm(X[:, 2])) < 1e-11
elif scenario == "gkl":
    # config-4 shape: rectangular random map, rows split unevenly, V-vectors sharded evenly (odd column count: padded shard)
    for (m, nc, seed) in ((600, 250, 21), (700, 251, 22)):
        A = ko.sparse_random(m, nc, 8, seed)
        u0 = np.random.default_rng(6).random(m)
        offs = uneven_offsets(m, seed)
        r0, r1 = int(offs[rank]), int(offs[rank + 1])
        op = kd.NativeShardedRectOperator(A[r0:r1], nc, ctx)
        shard = -(-nc // world)
        c0, c1 = rank * shard, min((rank + 1) * shard, nc)
        assert op.shape == (r1 - r0, c1 - c0), op.shape
        steps = 15
        for mgs_mode in (0, 1):
            ctx.set_option("mgs_mode", mgs_mode)
            for dev, ref in orth_pairs():
                it = kk.GKLIterator(op, u0[r0:r1], dev, capacity=steps + 3)
                f = kk.initialize(it)
                oit = ko.GKLIterator(A, u0.copy(), ref)
                of = ko.gkl_initialize(oit)
                g0 = comm.stats()["gather"]
                for _ in range(steps):
                    f = kk.expand_(it, f)
                    of = ko.gkl_expand(oit, of)
                assert comm.stats()["gather"] - g0 == 2 * steps   # one all-gather + one reduce-scatter per expand!
                tol = 1e-10 if dev.is_reorth else 1e-6
                ea, eb = relerr(f.alphas, of.alphas), relerr(f.betas, of.betas)
                assert ea < tol and eb < tol, (dev.name, mgs_mode, ea, eb)
                report[f"gkl{nc}.{dev.name}.mgs{mgs_mode}"] = [ea, eb]
                if dev.name == "mgs2":
                    U = gather_rows(f"U_{nc}_{mgs_mode}", f.U.to_numpy())
                    V = gather_rows(f"Vg_{nc}_{mgs_mode}", f.V.to_numpy())
                    Bm = f.rayleighquotient()
                    assert np.max(np.abs(U.T @ U - np.eye(U.shape[1]))) < 1e-12 and np.max(np.abs(V.T @ V - np.eye(V.shape[1]))) < 1e-12
                    assert np.max(np.abs(A.T @ U - V @ Bm.T)) < 1e-10
        ctx.set_option("mgs_mode", 2)
        xb, yb = kk.DeviceBasis(c1 - c0, 4, ctx), kk.DeviceBasis(r1 - r0, 4, ctx)
        xv = np.random.default_rng(9).standard_normal((nc, 3))
        for j in range(3):
            xb.upload(j, xv[c0:c1, j])
        op.apply(xb[0], yb[0])
        np.testing.assert_allclose(yb[0].get(), (A @ xv[:, 0])[r0:r1], rtol=0, atol=1e-12)
        yv = np.random.default_rng(10).standard_normal(m)
        op.apply_adjoint(yb[3].set(yv[r0:r1]), xb[3])
        np.testing.assert_allclose(xb[3].get(), (A.T @ yv)[c0:c1], rtol=0, atol=1e-12)
        from krylovkit_hip._lib import check
        check(ctx._lib.kk_block_apply(op.handle, xb.handle, 0, yb.handle, 0, 3))     # gathers per column (was: stale buffer)
        Y = np.stack([yb.download(j) for j in range(3)], 1)
        np.testing.assert_allclose(Y, (A @ xv)[r0:r1], rtol=0, atol=1e-12)
    # svdsolve end to end on the sharded map
    A = ko.sparse_random(600, 250, 8, 21)
    offs = uneven_offsets(600, 21)
    r0, r1 = int(offs[rank]), int(offs[rank + 1])
    op = kd.NativeShardedRectOperator(A[r0:r1], 250, ctx)
    u0 = np.random.default_rng(6).random(600)
    S, _, _, sinfo = kk.svdsolve(op, u0[r0:r1], 4, "LR", krylovdim=20, tol=1e-10, maxiter=30)
    So, _, _, soinfo = ko.svdsolve_gkl(A, u0.copy(), 4, "LR", krylovdim=20, tol=1e-10, maxiter=30)
    assert (sinfo.converged, sinfo.numiter, sinfo.numops) == (soinfo.converged, soinfo.numiter, soinfo.numops)
    assert relerr(S[:4], So[:4]) < 1e-10
    report["svd"] = [float(s) for s in S[:4]]
elif scenario == "block":
    # config-5 shape: BlockLanczos on a row-sharded stencil, both block modes; the issue-#143 known answer with rank drop
    nx, ny, bs = 36, 28 + world, 4
    n = nx * ny
    A = ko.laplacian_2d(nx, ny, shift_diag=10 * np.linspace(0, 1, n) ** 2)
    part = kd.Partition.even(n, world, rank, align=nx)
    op = kd.NativeShardedOperator(A[part.lo:part.hi], part, ctx, symmetric=True)
    rng = np.random.default_rng(12)
    xb = [rng.random(n) for _ in range(bs)]
    obit = ko.BlockLanczosIterator(A, [x.copy() for x in xb], 28)
    obf = ko.blocklanczos_initialize(obit)
    for _ in range(5):
        obf = ko.blocklanczos_expand(obit, obf)
    ko_ev = np.linalg.eigvalsh(obf.H[:len(obf), :len(obf)])
    for block_mode in (1, 0):
        ctx.set_option("block_mode", block_mode)
        bit = kk.BlockLanczosIterator(op, [x[part.lo:part.hi] for x in xb], 28)
        bf = bit.initialize()
        for _ in range(5):
            bf = bit.expand(bf)
        k = len(bf)
        assert k == len(obf)
        ev = np.linalg.eigvalsh(bf.H[:k, :k])
        err = float(np.max(np.abs(ev - ko_ev)) / np.max(np.abs(ko_ev)))
        assert err < 1e-10, (block_mode, err)
        report[f"block.mode{block_mode}"] = err
    ctx.set_option("block_mode", 1)
    A143 = np.load(HERE / "golden" / "issue143_A.npy")
    n = A143.shape[0]
    part = kd.Partition(uneven_offsets(n, 3), rank)
    op = kd.NativeShardedOperator(sp.csr_matrix(A143)[part.lo:part.hi], part, ctx, symmetric=True)
    rng = np.random.default_rng(143)
    x0 = [rng.standard_normal(n) for _ in range(20)]
    D, V, binfo = kk.eigsolve_block(op, [x[part.lo:part.hi] for x in x0], 4, "SR", kk.BlockLanczos(tol=1e-8))
    ev = np.linalg.eigvalsh(A143)
    assert len(D) == len(ev) and binfo.converged == len(D) and binfo.numiter == 1 and binfo.numops == len(D) + 1
    np.testing.assert_allclose(np.sort(D), ev, rtol=0, atol=1e-10 * np.max(np.abs(ev)))
elif scenario == "solvers":
    # thick-restart eigsolve, GMRES through the keyword front door (tolerance from the ALL-REDUCED norm of b), CG
    nx, ny = 32, 24 + world
    n = nx * ny
    A = ko.laplacian_2d(nx, ny, shift_diag=10 * np.linspace(0, 1, n) ** 2)
    x0 = np.random.default_rng(3).random(n)
    part = kd.Partition(uneven_offsets(n, 8), rank)
    lo, hi = part.lo, part.hi
    op = kd.NativeShardedOperator(A[lo:hi], part, ctx, symmetric=True)
    for mgs_mode in (1, 0):
        ctx.set_option("mgs_mode", mgs_mode)
        vals, vecs, einfo = kk.eigsolve(op, x0[lo:hi], 3, "LM", krylovdim=20, tol=1e-10, maxiter=50, orth=kk.ModifiedGramSchmidt2())
        ovals, ovecs, oinfo = ko.eigsolve_lanczos(A, x0.copy(), 3, "LM", krylovdim=20, tol=1e-10, maxiter=50, orth=ko.MGS2)
        assert einfo.converged >= 3 and (einfo.numiter, einfo.numops) == (oinfo.numiter, oinfo.numops), (mgs_mode, einfo, oinfo)
        assert relerr(vals[:3], ovals[:3]) < 1e-10
        v0 = gather_rows(f"eigvec{mgs_mode}", np.asarray(vecs[0]))
        assert np.linalg.norm(A @ v0 - vals[0] * v0) < 1e-8
    ctx.set_option("mgs_mode", 2)
    Bm = ko.convection_diffusion_2d(nx, ny)
    b = np.random.default_rng(4).random(n) * np.linspace(0.1, 3.0, n)     # local norms differ a lot from rank to rank
    opB = kd.NativeShardedOperator(Bm[lo:hi], part, ctx)
    tr, otr = [], []
    x, ginfo = kk.linsolve(opB, b[lo:hi], None, None, krylovdim=25, maxiter=40, rtol=1e-9, atol=0.0, trace=tr)
    ox, goinfo = ko.gmres(Bm, b, krylovdim=25, maxiter=40, tol=1e-9 * np.linalg.norm(b), orth=ko.MGS2, trace=otr)
    assert ginfo.converged == 1 and (ginfo.converged, ginfo.numiter, ginfo.numops) == (goinfo.converged, goinfo.numiter, goinfo.numops)
    assert len(tr) == len(otr) and relerr([t[2] for t in tr], [t[2] for t in otr]) < 1e-6
    xg = gather_rows("gmres_x", np.asarray(x))
    assert np.linalg.norm(Bm @ xg - b) < 2e-9 * np.linalg.norm(b)
    report["gmres"] = [ginfo.numiter, ginfo.numops]
    As = (A + sp.identity(n)).tocsr()
    opS = kd.NativeShardedOperator(As[lo:hi], part, ctx, symmetric=True)
    xc, cinfo = kk.linsolve(opS, b[lo:hi], None, kk.CG(300, 1e-9 * np.linalg.norm(b)))
    xcg = gather_rows("cg_x", np.asarray(xc))
    assert cinfo.converged == 1 and np.linalg.norm(As @ xcg - b) < 2e-9 * np.linalg.norm(b)
    report["cg"] = [cinfo.numiter, cinfo.numops]
elif scenario == "solvers2":
    # SURVEY 8(f)-3/4 on local blocks: BiCGStab (device-resident rho / sigma / alpha / omega, two fused applies per iteration),
    # LSMR on the row-sharded rectangular map (all-gather / reduce-scatter per iteration), exponentiate (Lanczos expand! + restarts)
    nx, ny = 32, 24 + world
    n = nx * ny
    Bm = ko.convection_diffusion_2d(nx, ny)
    part = kd.Partition(uneven_offsets(n, 9), rank)
    lo, hi = part.lo, part.hi
    b = np.random.default_rng(4).random(n)
    tol = 1e-9 * np.linalg.norm(b)
    opB = kd.NativeShardedOperator(Bm[lo:hi], part, ctx)
    x, binfo = kk.linsolve_bicgstab(opB, b[lo:hi], None, kk.BiCGStab(500, tol), 0.3, 0.9)
    xo, boinfo = ko.bicgstab(Bm, b, None, 0.3, 0.9, maxiter=500, tol=tol)
    assert binfo.converged == 1 and (binfo.numiter, binfo.numops) == (boinfo.numiter, boinfo.numops), (binfo, boinfo)
    xg = gather_rows("bicg_x", np.asarray(x))
    assert np.linalg.norm(0.3 * xg + 0.9 * (Bm @ xg) - b) < 2 * tol
    report["bicgstab"] = [binfo.numiter, binfo.numops]
    Ar = ko.sparse_random(600, 250, 8, 21)
    offs = uneven_offsets(600, 21)
    r0, r1 = int(offs[rank]), int(offs[rank + 1])
    opR = kd.NativeShardedRectOperator(Ar[r0:r1], 250, ctx)
    br = np.random.default_rng(7).random(600)
    ltol = 1e-10 * np.linalg.norm(br)
    xl, linfo = kk.lssolve(opR, br[r0:r1], kk.LSMR(kk.ModifiedGramSchmidt2(), 400, 10, ltol), 0.1)
    xlo, loinfo = ko.lsmr(Ar, br, 0.1, krylovdim=10, maxiter=400, tol=ltol, orth=ko.MGS2)
    assert (linfo.converged, linfo.numiter, linfo.numops) == (loinfo.converged, loinfo.numiter, loinfo.numops), (linfo, loinfo)
    shard = -(-250 // world)
    xlg = gather_rows("lsmr_x", np.asarray(xl))
    assert xlg.shape == (250,) and np.linalg.norm(xlg - xlo) <= 1e-8 * np.linalg.norm(xlo)
    report["lsmr"] = [linfo.numiter, linfo.numops]
    A = ko.laplacian_2d(nx, ny, shift_diag=10 * np.linspace(0, 1, n) ** 2)
    opA = kd.NativeShardedOperator(A[lo:hi], part, ctx, symmetric=True)
    v = np.random.default_rng(8).random(n)
    w, einfo = kk.exponentiate(opA, -0.4, v[lo:hi], kk.Lanczos(kk.ModifiedGramSchmidt2(), 20, 100, 1e-11))
    wo, eoinfo = ko.expintegrator(A, -0.4, (v,), krylovdim=20, maxiter=100, tol=1e-11, orth=ko.MGS2, method="lanczos")
    wg = gather_rows("expm_w", np.asarray(w))
    assert einfo.converged == 1 and (einfo.numiter, einfo.numops) == (eoinfo.numiter, eoinfo.numops)
    assert np.linalg.norm(wg - wo) <= 1e-9 * np.linalg.norm(wo)
    report["exponentiate"] = [einfo.numiter, einfo.numops]
elif scenario in ("xsync", "xsync_fault", "xsync_late"):
    # The persistent MGS kernels on a row-sharded context: the sum over the ranks happens INSIDE the launch -- block 0 of a rank
    # stores the rank's partial into every peer's IPC-mapped sync area, every block adds the W partials in rank order
    # (csrc/kk_xsync.h; reference order src/orthonormal.jl:414-439, factorizations/lanczos.jl:325-338, arnoldi.jl:239-245).
    # Every rank owns num_cus = device / world CUs (KK_NUM_CUS): the launches of all ranks are resident side by side.
    assert ctx.get_option("xsync_active") == 1, "kk_comm_init did not establish the cross-rank sync areas"
    assert ctx.get_option("num_cus") * world <= ctx.get_option("device_cus") and ctx.get_option("ranks_on_this_gpu") == world   # (found by kk_comm_init)
    report["xsync_hop_us"] = ctx.get_option("xsync_hop_us"); report["comm_allreduce_us"] = ctx.get_option("comm_allreduce_us")
    report["num_cus"] = int(ctx.get_option("num_cus"))
    ctx.set_option("xsync", 2)   # this scenario tests the in-kernel route itself: not subject to the cost rule (kk_xs_pays)
    nx, ny = 70, 64 * world
    n = nx * ny
    A = ko.laplacian_2d(nx, ny, shift_diag=10 * np.linspace(0, 1, n) ** 2)
    Cd = ko.convection_diffusion_2d(nx, ny)
    x0 = np.random.default_rng(3).random(n)
    part = kd.Partition.even(n, world, rank, align=nx)
    lo, hi = part.lo, part.hi
    opA = kd.NativeShardedOperator(A[lo:hi], part, ctx, symmetric=True)
    opC = kd.NativeShardedOperator(Cd[lo:hi], part, ctx)
    import os
    steps = int(os.environ.get("KK_W2_STEPS", "24"))      # (the 8-rank rehearsal on ONE GPU runs a shorter factorization: eight processes share the chip's queues)
    fault_at = ({7, 15} if steps >= 20 else {3}) if scenario != "xsync" else set()   # (short run: one fault -- the back-off after it outlasts the factorization)
    # xsync_late: the faulting rank gives up at the LAST reduction of its launch with its partial already published -- to its peers it
    # looks like a rank whose wait ran out a moment before they arrived: they find every partial in their area and must NOT commit
    # (abort word re-read before the commit), or the ranks' collectives stop pairing up.  k_mgs_persist carries the hook.
    routes = (("persist", 1), ("persist", 0)) if scenario == "xsync_late" else (("persist", 1), ("panel", 1), ("panel_p", 1), ("persist", 0))
    if os.environ.get("KK_W2_ROUTES"):   # (the 8-rank rehearsal: a subset -- every route costs ~25 s there, the processes time-share the chip)
        keep = os.environ["KK_W2_ROUTES"].split(",")
        routes = tuple(r for r in routes if f"{r[0]}{r[1]}" in keep)
    for route, lookahead in routes:
        ctx.set_option("mgs_mode", 2 if route == "panel_p" else 0)      # panel_p: panels of 2-3 vectors per reduction (auto mode), else the strict order
        ctx.set_option("mgs_panel", 0 if route == "persist" else 1)
        ctx.set_option("panel_min_rows", 0); ctx.set_option("persist_min_rows", 0)
        ctx.set_option("lookahead", lookahead)
        for case in ("lanczos", "arnoldi_mgs", "arnoldi_mgs2"):
            x1 = ctx.get_option("xsync_launches"); t1 = ctx.get_option("persist_timeouts")
            ctx.prof_reset(); ctx.prof_enable(1)
            if case == "lanczos":
                it = kk.LanczosIterator(opA, x0[lo:hi], kk.ModifiedGramSchmidt2(), capacity=steps + 3)
            else:
                it = kk.ArnoldiIterator(opC, x0[lo:hi], kk.ModifiedGramSchmidt() if case == "arnoldi_mgs" else kk.ModifiedGramSchmidt2(), capacity=steps + 3)
            f = kk.initialize(it)
            s0 = comm.stats()
            for i in range(steps):
                if i in fault_at and rank == (i % world):
                    ctx.set_option("persist_fault_late" if scenario == "xsync_late" else "persist_fault", 1)     # ONE rank loses a launch: its peers must give up with it (abort word), all repeat the sweep
                f = kk.expand_(it, f)
            s1 = comm.stats()
            ctx.prof_enable(0)
            launches = ctx.prof_get("k_mgs_persist")[1] + ctx.prof_get("k_mgs_panel")[1]
            print(f"[xsync rank {rank}] {route} la={lookahead} {case}: launches {launches} timeouts {ctx.get_option('persist_timeouts') - t1}", flush=True)
            assert launches > 0 and ctx.get_option("xsync_launches") - x1 == launches, (route, case, launches)
            assert (ctx.prof_get("k_mgs_persist")[1] > 0) == (route == "persist"), (route, case)
            if fault_at:
                assert ctx.get_option("persist_timeouts") - t1 >= len(fault_at), (route, case, ctx.get_option("persist_timeouts") - t1)
            else:
                # ranks sharing ONE GPU occasionally lose a launch to each other (an artefact of the test box: on a node every rank owns
                # its chip).  Rounds 4-5 hid that behind a retry of the whole scenario; now the loss is COUNTED, bounded and reported
                #  -- a rising rate shows in the test's output instead of disappearing
                lost = int(ctx.get_option("persist_timeouts") - t1)
                report["lost_launches"] = report.get("lost_launches", 0) + lost
                assert lost <= 1, (route, case, lost)
                if case == "lanczos" and lost == 0:   # the only all-reduce left per expand! is alpha0 of the apply (speculative applies included)
                    assert (s1["allreduce"] - s0["allreduce"]) <= steps + 2, (route, s0, s1)
            # the oracle on the global problem, after the device run (a rank busy on the CPU would let its peers' kernels wait)
            if case == "lanczos":
                oit = ko.LanczosIterator(A, x0.copy(), ko.MGS2); of = ko.lanczos_initialize(oit)
                for _ in range(steps):
                    of = ko.lanczos_expand(oit, of)
                ea, eb = relerr(f.alphas, of.alphas), relerr(f.betas, of.betas)
                assert ea < 1e-10 and eb < 1e-10, (route, case, ea, eb)
                bits = [float(a).hex() for a in f.alphas] + [float(b).hex() for b in f.betas]
            else:
                oit = ko.ArnoldiIterator(Cd, x0.copy(), ko.MGS if case == "arnoldi_mgs" else ko.MGS2); of = ko.arnoldi_initialize(oit)
                for _ in range(steps):
                    of = ko.arnoldi_expand(oit, of)
                H = np.asarray(f.H, dtype=float)
                ea = float(np.max(np.abs(H - np.asarray(of.H))) / np.max(np.abs(of.H)))
                eb = abs(f.normres - of.normres) / abs(of.normres)
                assert ea < 1e-10 and eb < 1e-10, (route, case, ea, eb)
                bits = [float(h).hex() for h in H.ravel()] + [float(f.normres).hex()]
            V = gather_rows(f"V_{route}{lookahead}_{case}", f.V.to_numpy())
            assert np.max(np.abs(V.T @ V - np.eye(V.shape[1]))) < 1e-12, (route, case)
            report[f"{route}{lookahead}.{case}"] = bits      # compared rank against rank by the test: bit-identical scalars
    # orthonormalize!! through the same kernels: global coefficients and norm, the stored vector normalised
    ctx.set_option("mgs_mode", 0); ctx.set_option("mgs_panel", 1)
    rng = np.random.default_rng(5)
    m = 9
    Q, _ = np.linalg.qr(rng.standard_normal((n, m)))
    w = rng.standard_normal(n)
    B = kk.DeviceBasis(hi - lo, m + 2, ctx)
    for j in range(m):
        B.upload(j, Q[lo:hi, j])
    B.length = m
    x, nrm, _ = B.orthogonalize(B[m].set(w[lo:hi]), kk.ModifiedGramSchmidt2())
    wo, xo = ko.orthogonalize(w.copy(), [Q[:, j].copy() for j in range(m)], ko.MGS2)
    np.testing.assert_allclose(x, xo, rtol=0, atol=1e-12 * np.linalg.norm(w))
    assert abs(nrm - np.linalg.norm(wo)) < 1e-12 * np.linalg.norm(w)
    np.testing.assert_allclose(B[m].get(), wo[lo:hi], rtol=0, atol=1e-12 * np.linalg.norm(w))
    ctx.set_option("mgs_mode", 2); ctx.set_option("lookahead", 1)
elif scenario == "state_machine":
    # the deferred-state machinery (normalised commit, speculative apply, step enqueued ahead, Gram rows) under random
    # interleavings of the entry points ON A ROW-SHARDED CONTEXT with the in-kernel cross-rank reduction on -- the same seeded sequence on
    # every rank (SPMD), scalars bit-identical between the ranks, oracle <= 1e-10 (src/factorizations/lanczos.jl:250-291, arnoldi.jl:199-260)
    assert ctx.get_option("xsync_active") == 1
    ctx.set_option("xsync", 2)   # (the stand-in's all-reduce is slow enough that the rule would always say yes anyway; pinned for the record)
    nx, ny = 70, 32 * world
    n = nx * ny
    A = ko.laplacian_2d(nx, ny, shift_diag=10 * np.linspace(0, 1, n) ** 2)
    Cd = ko.convection_diffusion_2d(nx, ny)
    x0 = np.random.default_rng(3).random(n)
    part = kd.Partition.even(n, world, rank, align=nx)
    lo, hi = part.lo, part.hi
    opA = kd.NativeShardedOperator(A[lo:hi], part, ctx, symmetric=True)
    opC = kd.NativeShardedOperator(Cd[lo:hi], part, ctx)
    OPS = ["expand", "expand", "expand", "norm_r", "norm_v", "dot_vv", "dot_rv", "read_r", "read_v", "project_r", "orth_extra", "shrink", "restart_scale",
           "toggle_lookahead", "toggle_fold", "toggle_speculate", "switch_route", "sync"]
    rng = np.random.default_rng(2024)     # same stream on every rank
    nseq = 18
    t_before = ctx.get_option("persist_timeouts")
    for seq in range(nseq):
        kind = ("lanczos", "arnoldi_mgs", "arnoldi_mgs2")[seq % 3]
        route = ("persist", "panel", "panel_p", "lowsync")[(seq // 3) % 4]
        ops = [OPS[j] for j in rng.integers(0, len(OPS), size=int(rng.integers(10, 24)))]
        picks = [int(v) for v in rng.integers(0, 10 ** 6, size=len(ops))]
        ctx.set_option("mgs_mode", {"persist": 0, "panel": 0, "panel_p": 2, "lowsync": 1}[route])
        ctx.set_option("mgs_panel", 0 if route == "persist" else 1)
        ctx.set_option("panel_min_rows", 0); ctx.set_option("persist_min_rows", 0)
        opt = {"lookahead": 1, "fold_scale": 1, "speculate": 1}
        for k_, v_ in opt.items():
            ctx.set_option(k_, v_)
        max_k = 16
        cap = max_k + 6
        if kind == "lanczos":
            dev, ref = kk.ModifiedGramSchmidt2(), ko.MGS2
            it = kk.LanczosIterator(opA, x0[lo:hi], dev, capacity=cap)
            oit = ko.LanczosIterator(A, x0.copy(), ref); of = ko.lanczos_initialize(oit)
            oexp, oshrink = ko.lanczos_expand, ko.lanczos_shrink
        else:
            dev, ref = (kk.ModifiedGramSchmidt(), ko.MGS) if kind == "arnoldi_mgs" else (kk.ModifiedGramSchmidt2(), ko.MGS2)
            it = kk.ArnoldiIterator(opC, x0[lo:hi], dev, capacity=cap)
            oit = ko.ArnoldiIterator(Cd, x0.copy(), ref); of = ko.arnoldi_initialize(oit)
            oexp, oshrink = ko.arnoldi_expand, ko.arnoldi_shrink
        f = kk.initialize(it)
        spare = cap - 1
        cur_mode = 1 if route == "lowsync" else 0
        bits = []

        def check_scalars(where):
            if kind == "lanczos":
                assert relerr(f.alphas, of.alphas) < 1e-10 and relerr(f.betas, of.betas) < 1e-10, (seq, where, list(zip(ops, picks)))
                bits.extend(float(v).hex() for v in list(f.alphas) + list(f.betas))
            else:
                H, Ho = np.asarray(f.H, float), np.asarray(of.H, float)
                assert H.shape == Ho.shape and np.max(np.abs(H - Ho)) < 1e-10 * max(1.0, np.max(np.abs(Ho))), (seq, where, list(zip(ops, picks)))
                bits.extend(float(v).hex() for v in H.ravel())
            assert abs(f.normres - of.normres) < 1e-10 * max(abs(of.normres), 1e-300), (seq, where)

        for i, (op_, pk) in enumerate(zip(ops, picks)):
            k = len(f)
            V = f.V
            oV = of.V if isinstance(of.V, list) else list(np.asarray(of.V).T)
            rn = np.linalg.norm(of.r)
            if op_ == "expand":
                if k >= max_k:
                    continue
                f = kk.expand_(it, f); of = oexp(oit, of)
                check_scalars(f"expand at op {i}")
            elif op_ == "norm_r":
                v = f.r.norm(); bits.append(float(v).hex())
                assert abs(v - rn) < 1e-10 * rn, (seq, i, op_)
            elif op_ == "norm_v":
                v = V[pk % k].norm(); bits.append(float(v).hex())
                assert abs(v - 1.0) < 1e-12, (seq, i, op_)
            elif op_ == "dot_vv":
                j, j2 = pk % k, (pk // 7) % k
                v = V[j].inner(V[j2]); bits.append(float(v).hex())
                assert abs(v - float(oV[j] @ oV[j2])) < 1e-11, (seq, i, op_)
            elif op_ == "dot_rv":
                j = pk % k
                assert abs(f.r.inner(V[j]) - float(of.r @ oV[j])) < 1e-10 * rn, (seq, i, op_)
            elif op_ == "read_r":
                assert np.max(np.abs(f.r.get() - of.r[lo:hi])) < 1e-10 * rn, (seq, i, op_)
            elif op_ == "read_v":
                j = pk % k
                assert np.max(np.abs(V[j].get() - oV[j][lo:hi])) < 1e-9, (seq, i, op_)
            elif op_ == "project_r":
                sdev = V.project(f.r, 0, k)
                so = np.array([float(q @ of.r) for q in oV[:k]])
                assert np.max(np.abs(np.asarray(sdev) - so)) < 1e-10 * rn, (seq, i, op_)
            elif op_ == "orth_extra":
                w = np.random.default_rng(pk).standard_normal(n)
                x, nrm, _ = V.orthogonalize(V[spare].set(w[lo:hi]), dev, 0, k)
                wo, xo = ko.orthogonalize(w.copy(), [q.copy() for q in oV[:k]], ref)
                np.testing.assert_allclose(x, xo, rtol=0, atol=1e-10 * np.linalg.norm(w))
                assert abs(nrm - np.linalg.norm(wo)) < 1e-10 * np.linalg.norm(w)
                bits.append(float(nrm).hex())
            elif op_ == "shrink":
                if k < 4:
                    continue
                kn = 2 + pk % (k - 2)
                f = kk.shrink_(f, kn); of = oshrink(of, kn)
                check_scalars(f"shrink at op {i}")
            elif op_ == "restart_scale":
                out = V[spare - 1].scale_from_(f.r, 1.0 / f.normres).get()
                assert np.max(np.abs(out - of.r[lo:hi] / of.normres)) < 1e-10, (seq, i, op_)
                assert np.max(np.abs(f.r.get() - of.r[lo:hi])) < 1e-10 * rn, (seq, i, op_)
            elif op_.startswith("toggle_"):
                key = {"toggle_lookahead": "lookahead", "toggle_fold": "fold_scale", "toggle_speculate": "speculate"}[op_]
                opt[key] ^= 1
                ctx.set_option(key, opt[key])
            elif op_ == "switch_route":
                cur_mode = 1 - cur_mode
                ctx.set_option("mgs_mode", cur_mode)
            elif op_ == "sync":
                ctx.sync()
        check_scalars("end")
        Vg = gather_rows(f"V_sm{seq}", f.V.to_numpy(len(f)))
        assert np.max(np.abs(Vg.T @ Vg - np.eye(Vg.shape[1]))) < 1e-11, seq
        report[f"sm.{seq}"] = bits
        del it, f
    report["persist_timeouts"] = int(ctx.get_option("persist_timeouts") - t_before)
    report["xsync_launches"] = int(ctx.get_option("xsync_launches"))
    ctx.set_option("mgs_mode", 2); ctx.set_option("lookahead", 1); ctx.set_option("fold_scale", 1); ctx.set_option("speculate", 1)
elif scenario == "xsync_full":
    # config-2 shape at world x 5 M rows and config-3 shape at world x 1 M rows: the sizes where the persistent kernels ARE the
    # route of the auto mode (k_mgs_persist from 3.6 M rows per rank, k_mgs_panel for 1.4 .. 4.19 M) -- default options
    sys.path.insert(0, str(HERE))
    import cpu_ref_lib
    ref = cpu_ref_lib.load()
    assert ctx.get_option("xsync_active") == 1
    for shape, steps in (("lanczos", 20), ("gmres", 20)):
        nx = 2500 if shape == "lanczos" else 1000
        ny = (2000 if shape == "lanczos" else 1000) * world
        n = nx * ny
        A = ko.laplacian_2d(nx, ny) if shape == "lanczos" else ko.convection_diffusion_2d(nx, ny)
        x0 = np.random.default_rng(3).random(n)
        part = kd.Partition.even(n, world, rank, align=nx)
        lo, hi = part.lo, part.hi
        op = kd.NativeShardedOperator(A[lo:hi], part, ctx, symmetric=(shape == "lanczos"))
        ctx.prof_reset(); ctx.prof_enable(1)
        t1 = ctx.get_option("persist_timeouts")
        if shape == "lanczos":
            it = kk.LanczosIterator(op, x0[lo:hi], kk.ModifiedGramSchmidt2(), capacity=steps + 3)
        else:
            it = kk.ArnoldiIterator(op, x0[lo:hi], kk.ModifiedGramSchmidt2(), capacity=steps + 3)
        f = kk.initialize(it)
        t0 = time.time()
        for i in range(steps):
            f = kk.expand_(it, f)
        ctx.sync()
        dt = time.time() - t0
        ctx.prof_enable(0)
        kern = "k_mgs_persist" if shape == "lanczos" else "k_mgs_panel"
        assert ctx.prof_get(kern)[1] >= steps - 6, (shape, ctx.prof_get("k_mgs_persist"), ctx.prof_get("k_mgs_panel"))   # (a lost launch sends the next four sweeps down the launch-per-vector route)
        lost = int(ctx.get_option("persist_timeouts") - t1)
        report["lost_launches"] = report.get("lost_launches", 0) + lost
        assert lost <= 1, (shape, lost)
        comm.barrier()
        if shape == "lanczos":
            a_ref, b_ref = cpu_ref_lib.run_lanczos(ref, A, x0, steps, 3, nthreads=max(2, cpu_ref_lib.usable_threads() // world))[:2]   # 3 = MGS2
            ea, eb = relerr(f.alphas, a_ref), relerr(f.betas, b_ref)
            assert ea < 1e-10 and eb < 1e-10, (shape, ea, eb)
            report[f"full.{shape}"] = [float(a).hex() for a in f.alphas] + [float(b).hex() for b in f.betas]
        else:
            oit = ko.ArnoldiIterator(A, x0.copy(), ko.MGS2); of = ko.arnoldi_initialize(oit)
            for _ in range(steps):
                of = ko.arnoldi_expand(oit, of)
            H = np.asarray(f.H, dtype=float)
            ea = float(np.max(np.abs(H - np.asarray(of.H))) / np.max(np.abs(of.H)))
            assert ea < 1e-10, (shape, ea)
            report[f"full.{shape}"] = [float(h).hex() for h in H.ravel()]
        report[f"full.{shape}.ms_per_step"] = 1e3 * dt / steps
        del it, f, op
elif scenario == "bad_input":
    # a collective create call with bad input on ONE rank: every rank must come back with an error (nobody left waiting)
    import ctypes as C
    from krylovkit_hip import _lib
    n = 8
    offs = (C.c_int64 * (world + 1))(*[q * n for q in range(world + 1)])
    rowptr = (C.c_int64 * (n + 1))(*range(n + 1))
    cols = [rank * n + i for i in range(n)]
    if rank == world - 1:
        cols[-1] = world * n + 5          # out of range on the last rank only
    col = (C.c_int64 * n)(*cols)
    val = (C.c_double * n)(*([1.0] * n))
    h = C.c_void_p()
    st = ctx._lib.kk_csr_create_sharded(ctx.handle, n, offs, n, rowptr, col, val, 0, 0, C.byref(h))
    assert st == _lib.KK_ERR_DIM, st
    msg = ctx._lib.kk_last_error().decode()
    assert ("out of range" in msg) if rank == world - 1 else ("another rank" in msg), msg
    comm.barrier()                        # the communicator is still usable afterwards
    report["status"] = st
else:
    raise SystemExit(f"unknown scenario {scenario}")

report["stats"] = comm.stats()
comm.barrier()
comm.close()
ctx.close()
(rdv / f"report.{rank}.json").write_text(json.dumps(report))
print(f"world2 {scenario} rank {rank} OK", flush=True)
