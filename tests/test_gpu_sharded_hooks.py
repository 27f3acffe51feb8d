"""Hook-based row sharding (kk_ctx_set_allreduce / kk_op_set_halo_hook) with TWO logical ranks as
two threads on the one GPU of the box (tests/loopback_collective.py): the ordinary iterators and
solvers run unchanged on row shards and must reproduce the serial oracle -- Lanczos with all six
orthogonalisers, thick-restart eigsolve, GMRES (numops), BlockLanczos (known answer of issue #143)."""
import sys
import threading
import traceback
from pathlib import Path

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "tests"))


def run_ranks(world, fn):
    """fn(rank, sctx, coll) in `world` threads; returns the list of results (re-raises failures)."""
    import torch
    import splitphase_dist as kd
    from loopback_collective import LoopbackCollective, LoopbackWorld

    shared = LoopbackWorld(world)
    res, err = [None] * world, [None] * world

    def work(rank):
        try:
            torch.cuda.set_device(0)
            coll = LoopbackCollective(shared, rank)
            sctx = kd.ShardedContext(coll, 0)
            res[rank] = fn(rank, sctx, coll)
            sctx.backend.sync()
        except BaseException as e:  # noqa: BLE001
            err[rank] = (e, traceback.format_exc())
            shared.barrier.abort()

    ts = [threading.Thread(target=work, args=(r,)) for r in range(world)]
    for t in ts:
        t.start()
    for t in ts:
        t.join(timeout=600)
    for e in err:
        if e is not None:
            raise AssertionError(e[1])
    return res


def relerr(a, b):
    a, b = np.asarray(a, float), np.asarray(b, float)
    return float(np.max(np.abs(a - b) / np.maximum(np.abs(b), 1e-300)))


def test_sharded_lanczos_all_orthogonalizers(kk, ko, ctx):
    import splitphase_dist as kd
    nx, ny, steps, world = 40, 30, 20, 2
    n = nx * ny
    A = ko.laplacian_2d(nx, ny, shift_diag=10 * np.linspace(0, 1, n) ** 2)
    x0 = np.random.default_rng(3).random(n)
    pairs = [(kk.ClassicalGramSchmidt(), ko.CGS), (kk.ModifiedGramSchmidt(), ko.MGS), (kk.ClassicalGramSchmidt2(), ko.CGS2),
             (kk.ModifiedGramSchmidt2(), ko.MGS2), (kk.ClassicalGramSchmidtIR(0.75), ko.CGSIR(0.75)),
             (kk.ModifiedGramSchmidtIR(0.75), ko.MGSIR(0.75))]

    def fn(rank, sctx, coll):
        part = kd.Partition.even(n, world, rank, align=nx)
        op = sctx.operator(A[part.lo:part.hi, :], part)
        out = {}
        for mode in (0, 1):
            sctx.ctx.set_option("mgs_mode", mode)
            for dev, _ in pairs:
                it = kk.LanczosIterator(op, x0[part.lo:part.hi], dev, capacity=steps + 3)
                f = kk.initialize(it)
                for _ in range(steps):
                    f = kk.expand_(it, f)
                out[(mode, dev.name)] = (list(f.alphas), list(f.betas), f.V.to_numpy())
        out["calls"] = sctx.calls
        return out

    res = run_ranks(world, fn)
    assert res[0]["calls"] == res[1]["calls"] > 0
    for dev, ref in pairs:
        it = ko.LanczosIterator(A, x0.copy(), ref)
        f = ko.lanczos_initialize(it)
        for _ in range(steps):
            f = ko.lanczos_expand(it, f)
        for mode in (0, 1):
            a0, b0, V0 = res[0][(mode, dev.name)]
            a1, b1, V1 = res[1][(mode, dev.name)]
            assert a0 == a1 and b0 == b1                      # identical scalars on both ranks
            tol = 1e-10 if dev.is_reorth else 1e-6
            assert relerr(a0, f.alphas) < tol and relerr(b0, f.betas) < tol, (mode, dev.name)
            V = np.vstack([V0, V1])
            if dev.is_reorth:
                assert np.max(np.abs(V.T @ V - np.eye(V.shape[1]))) < 1e-12


def test_sharded_eigsolve_gmres_block(kk, ko, ctx):
    import splitphase_dist as kd
    import scipy.sparse as sps
    world = 2
    nx, ny = 24, 18
    n = nx * ny
    A = ko.laplacian_2d(nx, ny, shift_diag=10 * np.linspace(0, 1, n) ** 2)
    C = ko.convection_diffusion_2d(nx, ny)
    A143 = np.load(ROOT / "tests" / "golden" / "issue143_A.npy")
    x0 = np.random.default_rng(2).random(n)
    b = np.random.default_rng(4).random(n)
    tol = 1e-10 * np.linalg.norm(b)
    xb = [np.random.default_rng(143 + i).standard_normal(71) for i in range(20)]

    def fn(rank, sctx, coll):
        part = kd.Partition.even(n, world, rank, align=nx)
        sl = slice(part.lo, part.hi)
        out = {}
        op = sctx.operator(A[sl, :], part)
        vals, vecs, info = kk.eigsolve(op, x0[sl], 3, "SR", kk.Lanczos(krylovdim=30, tol=1e-10, maxiter=200))
        out["eig"] = (vals[:3], info.numiter, info.numops, np.stack(vecs[:3], 1))
        cop = sctx.operator(C[sl, :], part)
        x, ginfo = kk.linsolve(cop, b[sl], None, kk.GMRES(kk.ModifiedGramSchmidt2(), 20, 25, tol))
        out["gmres"] = (x, ginfo.numiter, ginfo.numops, ginfo.converged)
        p143 = kd.Partition.even(71, world, rank)
        bop = sctx.operator(sps.csr_matrix(A143)[p143.lo:p143.hi, :], p143)
        for mode in (0, 1):
            sctx.ctx.set_option("block_mode", mode)
            D, V, binfo = kk.eigsolve_block(bop, [v[p143.lo:p143.hi] for v in xb], 4, "SR", kk.BlockLanczos(tol=1e-8))
            out[("block", mode)] = (D, binfo.numiter, binfo.numops, binfo.converged)
        return out

    res = run_ranks(world, fn)
    ev = np.linalg.eigvalsh(A.toarray())
    ovals, _, oinfo = ko.eigsolve_lanczos(A, x0, 3, "SR", krylovdim=30, tol=1e-10, maxiter=200, orth=ko.MGS2)
    for r in res:
        vals, numiter, numops, _ = r["eig"]
        assert relerr(vals, ev[:3]) < 1e-10 and (numiter, numops) == (oinfo.numiter, oinfo.numops)
    Vg = np.vstack([res[0]["eig"][3], res[1]["eig"][3]])
    for i in range(3):
        assert np.linalg.norm(A @ Vg[:, i] - res[0]["eig"][0][i] * Vg[:, i]) < 1e-8
    xo, ginfo = ko.gmres(C, b, None, krylovdim=25, maxiter=20, tol=tol, orth=ko.MGS2)
    xg = np.concatenate([res[0]["gmres"][0], res[1]["gmres"][0]])
    assert res[0]["gmres"][1:] == res[1]["gmres"][1:] == (ginfo.numiter, ginfo.numops, 1)
    # NOTE: linsolve's host-side |b| would be a shard norm; the absolute tol is passed explicitly here
    assert np.linalg.norm(C @ xg - b) <= 1.01 * tol
    ev143 = np.linalg.eigvalsh(A143)
    for mode in (0, 1):
        D, numiter, numops, conv = res[0][("block", mode)]
        assert len(D) == 71 and numiter == 1 and numops == 72 and conv == 71
        np.testing.assert_allclose(np.sort(D), ev143, rtol=0, atol=1e-10 * np.max(np.abs(ev143)))


def test_sharded_short_recurrences_and_exponentiate(kk, ko, ctx):
    """SURVEY 8(f)-3/4 rows on row shards through the same hooks: CG, BiCGStab (device-resident rho/alpha/omega are
    all-reduced sums, so both ranks take the same branches) and exponentiate, against the serial oracle."""
    import splitphase_dist as kd
    world = 2
    nx, ny = 24, 18
    n = nx * ny
    A = ko.laplacian_2d(nx, ny, shift_diag=0.5 + np.linspace(0, 1, n))
    Cm = ko.convection_diffusion_2d(nx, ny)
    E = (ko.laplacian_2d(nx, ny) / 8.0).tocsr()
    b = np.random.default_rng(4).random(n)
    v = np.random.default_rng(6).random(n)
    tol = 1e-10 * np.linalg.norm(b)

    def fn(rank, sctx, coll):
        part = kd.Partition.even(n, world, rank, align=nx)
        sl = slice(part.lo, part.hi)
        out = {}
        x, info = kk.linsolve_cg(sctx.operator(A[sl, :], part), b[sl], None, kk.CG(500, tol), 0.3, 0.9)
        out["cg"] = (x, info.numiter, info.numops, info.converged)
        x, info = kk.linsolve_bicgstab(sctx.operator(Cm[sl, :], part), b[sl], None, kk.BiCGStab(7, 1e-30))
        out["bicg"] = (x, info.numiter, info.numops, info.converged, info.normres)
        w, info = kk.exponentiate(sctx.operator(E[sl, :], part), -1.5, v[sl], kk.Lanczos(kk.ModifiedGramSchmidt2(), 12, 100, 1e-11))
        out["exp"] = (w, info.numiter, info.numops, info.converged)
        return out

    res = run_ranks(world, fn)
    xo, oinfo = ko.cg(A, b, None, 0.3, 0.9, maxiter=500, tol=tol)
    xg = np.concatenate([res[0]["cg"][0], res[1]["cg"][0]])
    assert res[0]["cg"][1:] == res[1]["cg"][1:] == (oinfo.numiter, oinfo.numops, 1)
    np.testing.assert_allclose(xg, xo, rtol=0, atol=1e-9 * np.linalg.norm(xo))
    xo, oinfo = ko.bicgstab(Cm, b, None, maxiter=7, tol=1e-30)
    xg = np.concatenate([res[0]["bicg"][0], res[1]["bicg"][0]])
    assert res[0]["bicg"][1:4] == res[1]["bicg"][1:4] == (oinfo.numiter, oinfo.numops, 0)
    assert relerr(res[0]["bicg"][4], oinfo.normres) < 1e-8
    np.testing.assert_allclose(xg, xo, rtol=0, atol=1e-10 * np.linalg.norm(xo))
    wo, oinfo = ko.expintegrator(E, -1.5, (v,), krylovdim=12, maxiter=100, tol=1e-11, orth=ko.MGS2, method="lanczos")
    wg = np.concatenate([res[0]["exp"][0], res[1]["exp"][0]])
    assert res[0]["exp"][1:] == res[1]["exp"][1:] == (oinfo.numiter, oinfo.numops, 1)
    np.testing.assert_allclose(wg, wo, rtol=0, atol=1e-10 * np.linalg.norm(wo))
