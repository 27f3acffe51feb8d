"""world_size-2 gloo run of the row-sharded Lanczos path (krylovkit_hip.dist) on CPU.
The product's partition / ghost-exchange / all-reduce logic runs unchanged; only the local
compute engine is replaced by the NumPy checker backend (tests/dist_checker_backend.py).
Results are compared with the serial oracle."""
import os
import socket
import sys
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parent.parent


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, case, q):
    sys.path.insert(0, str(ROOT / "krylovkit.jl_amd"))
    sys.path.insert(0, str(ROOT / "oracle"))
    sys.path.insert(0, str(ROOT / "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch.distributed as dist
    import scipy.sparse as sp
    import krylov_oracle as ko
    import splitphase_dist as kd
    from krylovkit_hip.core import Orthogonalizer
    from dist_checker_backend import CheckerBackend

    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        if case == "laplacian":
            nx, ny = 12, 10
            A = ko.laplacian_2d(nx, ny, shift_diag=3 * np.linspace(0, 1, nx * ny) ** 2)
            part = kd.Partition.even(nx * ny, world, rank, align=nx)
        else:  # general sparse symmetric: ghosts scattered over the whole other rank
            n = 150
            R = sp.random(n, n, density=0.05, random_state=5, format="csr")
            A = (R + R.T + sp.identity(n) * 4).tocsr()
            part = kd.Partition.even(n, world, rank)
        n = A.shape[0]
        x0 = np.random.default_rng(3).random(n)
        be = CheckerBackend()
        op = kd.DistSparseOperator(A[part.lo:part.hi, :], part, be)
        res = {"rank": rank, "n_ghost": op.n_ghost, "send": op.send_counts, "recv": op.recv_counts}
        for name in ("cgs", "mgs", "cgs2", "mgs2", "cgsir", "mgsir"):
            it = kd.DistLanczosIterator(op, x0[part.lo:part.hi], Orthogonalizer(name), capacity=22)
            f = it.initialize()
            for _ in range(18):
                f = it.expand(f)
            Vloc = np.stack([be.download(f.V, j) for j in range(len(f.V))], 1)
            res[name] = (list(f.alphas), list(f.betas), Vloc)
        q.put(res)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("case", ["laplacian", "random"])
def test_row_sharded_lanczos_gloo_world2(case):
    import torch.multiprocessing as mp
    sys.path.insert(0, str(ROOT / "oracle"))
    import krylov_oracle as ko
    import scipy.sparse as sp

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, case, q)) for r in range(2)]
    for p in procs:
        p.start()
    out = [q.get(timeout=240) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    out.sort(key=lambda r: r["rank"])
    if case == "laplacian":
        nx, ny = 12, 10
        A = ko.laplacian_2d(nx, ny, shift_diag=3 * np.linspace(0, 1, nx * ny) ** 2)
        # one grid row of halo per neighbour, nothing else
        assert out[0]["n_ghost"] == nx and out[1]["n_ghost"] == nx
        assert out[0]["send"] == [0, nx] and out[1]["recv"] == [nx, 0]
    else:
        n = 150
        R = sp.random(n, n, density=0.05, random_state=5, format="csr")
        A = (R + R.T + sp.identity(n) * 4).tocsr()
        assert out[0]["n_ghost"] > 0 and out[0]["send"][1] == out[1]["recv"][0]
    x0 = np.random.default_rng(3).random(A.shape[0])
    for name, ref in (("cgs", ko.CGS), ("mgs", ko.MGS), ("cgs2", ko.CGS2), ("mgs2", ko.MGS2), ("cgsir", ko.CGSIR()),
                      ("mgsir", ko.MGSIR())):
        it = ko.LanczosIterator(A, x0.copy(), ref)
        f = ko.lanczos_initialize(it)
        stats = {}
        for _ in range(18):
            f = ko.lanczos_expand(it, f, stats)
        if name.endswith("ir"):
            assert stats["passes"] > 0, "the refinement loop must have been exercised"
        tol = 1e-7 if name in ("cgs", "mgs") else 1e-10
        for r in out:
            a, b, _ = r[name]
            np.testing.assert_allclose(a, f.alphas, rtol=tol, err_msg=name)
            np.testing.assert_allclose(b, f.betas, rtol=tol, err_msg=name)
        V = np.vstack([out[0][name][2], out[1][name][2]])  # re-assembled global basis
        if name not in ("cgs", "mgs"):
            assert np.max(np.abs(V.T @ V - np.eye(V.shape[1]))) < 1e-12
            np.testing.assert_allclose(np.abs(V), np.abs(np.stack(f.V, 1)), atol=1e-9)


def test_partition_even():
    sys.path.insert(0, str(ROOT / "krylovkit.jl_amd"))
    from krylovkit_hip.dist import Partition
    p = Partition.even(4000 * 10, 3, 1, align=4000)
    assert list(p.offsets) == [0, 16000, 28000, 40000] and p.n_local == 12000 and p.lo == 16000
    assert Partition.even(10, 1, 0).n_local == 10


def _gkl_worker(rank, world, port, q):
    sys.path.insert(0, str(ROOT / "krylovkit.jl_amd"))
    sys.path.insert(0, str(ROOT / "oracle"))
    sys.path.insert(0, str(ROOT / "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch.distributed as dist
    import krylov_oracle as ko
    import splitphase_dist as kd
    from krylovkit_hip.core import Orthogonalizer
    from dist_checker_backend import CheckerBackend

    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        m, n = 240, 100
        A = ko.sparse_random(m, n, 6, 41)
        u0 = np.random.default_rng(6).random(m)
        rp = kd.Partition.even(m, world, rank)
        cp = kd.Partition.even(n, world, rank)
        be = CheckerBackend()
        op = kd.DistRectOperator(A[rp.lo:rp.hi, :], rp, cp, be)
        res = {"rank": rank}
        for name in ("cgs", "cgs2", "mgs2", "cgsir", "mgsir"):
            # eta close to 1 so that the refinement loops of BOTH bases run (with 1/sqrt(2) the U side never needs one here)
            it = kd.DistGKLIterator(op, u0[rp.lo:rp.hi], Orthogonalizer(name, 0.9999), capacity=18)
            f = it.initialize()
            for _ in range(14):
                f = it.expand(f)
            Ul = np.stack([be.download(f.U, j) for j in range(len(f.U))], 1)
            Vl = np.stack([be.download(f.V, j) for j in range(len(f.V))], 1)
            res[name] = (list(f.alphas), list(f.betas), Ul, Vl)
        q.put(res)
    finally:
        dist.destroy_process_group()


def test_row_sharded_gkl_gloo_world2():
    """cfg 4 (svdsolve/GKL, basis row-sharded): all-gather of v, reduce-scatter of A'u, all-reduced
    inner products -- world_size 2 on gloo, checker backend, against the serial oracle."""
    import torch.multiprocessing as mp
    sys.path.insert(0, str(ROOT / "oracle"))
    import krylov_oracle as ko

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_gkl_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    out = [q.get(timeout=240) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    out.sort(key=lambda r: r["rank"])
    A = ko.sparse_random(240, 100, 6, 41)
    u0 = np.random.default_rng(6).random(240)
    for name, ref in (("cgs", ko.CGS), ("cgs2", ko.CGS2), ("mgs2", ko.MGS2), ("cgsir", ko.CGSIR(0.9999)),
                      ("mgsir", ko.MGSIR(0.9999))):
        it = ko.GKLIterator(A, u0.copy(), ref)
        f = ko.gkl_initialize(it)
        stats = {}
        for _ in range(14):
            f = ko.gkl_expand(it, f, stats)
        if name.endswith("ir"):   # the refinement loops of both bases must have been exercised
            assert stats["passes_v"] > 0 and stats["passes_u"] > 0
        tol = 1e-10 if name != "cgs" else 1e-6
        for r in out:
            np.testing.assert_allclose(r[name][0], f.alphas, rtol=tol, err_msg=name)
            np.testing.assert_allclose(r[name][1], f.betas, rtol=tol, err_msg=name)
        U = np.vstack([out[0][name][2], out[1][name][2]])
        V = np.vstack([out[0][name][3], out[1][name][3]])
        k = U.shape[1]
        B = np.diag(f.alphas) + np.diag(f.betas[:-1], -1)
        assert np.max(np.abs(A.T @ U - V @ B.T)) < 1e-9
        if name in ("mgs2", "mgsir"):
            assert np.max(np.abs(U.T @ U - np.eye(k))) < 1e-12 and np.max(np.abs(V.T @ V - np.eye(k))) < 1e-12
