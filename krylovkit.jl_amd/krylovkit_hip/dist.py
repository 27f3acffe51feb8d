"""Row-sharded (one process per GPU) operation over RCCL / xGMI  (SURVEY.md 8(e)).

kk_comm_init gives a context its own RCCL communicator: from then on every entry point of libkrylov_hip works on
row-sharded vectors and issues its collectives itself on the context stream -- ncclAllReduce at every finalize site
(kk_lanczos_expand with CGS2 / low-sync MGS2: exactly two per step, 2m+1 and 1 doubles; with the persistent MGS kernels, which since
0.3.1 sum their inner products over the ranks INSIDE the launch through IPC-mapped sync areas -- csrc/kk_xsync.h, set up by kk_comm_init,
`ctx.get_option("xsync_active")` -- only the one of alpha0), ONE grouped ncclSend / ncclRecv
of the ghost entries before a sparse apply (NativeShardedOperator; one group for all <= 16 columns of a block apply),
ncclAllGather / ncclReduceScatter for the rectangular map of the sharded GKL (NativeShardedRectOperator).  Every N-vector
(basis, w, r) is split into contiguous row blocks; axpy / scal / unproject / basistransform / Givens are purely local.
No torch tensor, no Python callback and no torch.distributed call sits on the hot path; the ORDINARY iterators and
drivers (LanczosIterator, ArnoldiIterator, GKLIterator, BlockLanczosIterator, eigsolve, linsolve, svdsolve, all six
orthogonalisers) are used unchanged on the local blocks and see bit-identical scalars on every rank.  Only the rendezvous
of the 128-byte communicator id needs an out-of-band channel (torch.distributed over gloo, MPI, a file ...).

This module is the ONE multi-GPU mechanism of the package.  (The C ABI additionally offers caller-owned collectives
through hooks -- kk_ctx_set_allreduce / kk_op_set_halo_hook -- and split-phase kk_*_dev entry points for hosts that
bring their own communicator; their Python exercisers live with the tests: tests/splitphase_dist.py.)
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass

import numpy as np

from . import _lib
from ._lib import check
from .core import Context, DeviceBasis, SparseOperator


# ------------------------------------------------------------------------------ partition
@dataclass
class Partition:
    """Contiguous row blocks: rank p owns global rows [offsets[p], offsets[p+1])."""
    offsets: np.ndarray
    rank: int

    @property
    def world(self) -> int:
        return len(self.offsets) - 1

    @property
    def lo(self) -> int:
        return int(self.offsets[self.rank])

    @property
    def hi(self) -> int:
        return int(self.offsets[self.rank + 1])

    @property
    def n_local(self) -> int:
        return self.hi - self.lo

    @property
    def n_global(self) -> int:
        return int(self.offsets[-1])

    @staticmethod
    def even(n_global: int, world: int, rank: int, align: int = 1) -> "Partition":
        """Near-equal blocks whose boundaries are multiples of `align` (e.g. nx for a stencil)."""
        units = n_global // align
        assert units * align == n_global, "n_global must be a multiple of align"
        base, rem = divmod(units, world)
        counts = np.array([(base + (1 if p < rem else 0)) * align for p in range(world)], dtype=np.int64)
        return Partition(np.concatenate([[0], np.cumsum(counts)]), rank)


# ------------------------------------------------------------------------------ communicator + sharded operators
class NativeComm:
    """One rank of a row-sharded run: Context + RCCL communicator owned by libkrylov_hip.

    `bcast(obj_or_None) -> obj` must return rank 0's object on every rank (torch.distributed over gloo, MPI, a file ...)."""

    def __init__(self, ctx: Context, rank: int, world: int, bcast, force_collectives: bool = False):
        self.ctx, self.rank, self.world = ctx, int(rank), int(world)
        lib = ctx._lib
        buf = C.create_string_buffer(_lib.KK_COMM_ID_BYTES)
        if self.rank == 0:
            check(lib.kk_comm_get_unique_id(buf))
        raw = bcast(bytes(buf.raw) if self.rank == 0 else None)
        if not isinstance(raw, (bytes, bytearray)) or len(raw) != _lib.KK_COMM_ID_BYTES:
            raise ValueError("NativeComm: the broadcast must deliver rank 0's 128-byte communicator id")
        idbuf = C.create_string_buffer(bytes(raw), _lib.KK_COMM_ID_BYTES)
        check(lib.kk_comm_init(ctx.handle, idbuf, self.rank, self.world,
                               _lib.KK_COMM_FORCE_COLLECTIVES if force_collectives else 0))
        self._open = True

    @classmethod
    def from_torch_distributed(cls, ctx: Context, group=None, force_collectives: bool = False) -> "NativeComm":
        """Rendezvous through an initialised torch.distributed group (any backend; gloo is enough: only 128 bytes travel)."""
        import torch.distributed as dist
        if not dist.is_initialized():
            return cls.single(ctx, force_collectives)
        rank, world = dist.get_rank(group), dist.get_world_size(group)

        def bcast(obj):
            box = [obj]
            dist.broadcast_object_list(box, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
            return box[0]

        return cls(ctx, rank, world, bcast, force_collectives)

    @classmethod
    def single(cls, ctx: Context, force_collectives: bool = False) -> "NativeComm":
        return cls(ctx, 0, 1, lambda obj: obj, force_collectives)

    def info(self):
        r, w, v = C.c_int(), C.c_int(), C.c_int()
        check(self.ctx._lib.kk_comm_info(self.ctx.handle, C.byref(r), C.byref(w), C.byref(v)))
        return dict(rank=r.value, world=w.value, rccl_version=v.value)

    def stats(self):
        a, p, g = C.c_int64(), C.c_int64(), C.c_int64()
        check(self.ctx._lib.kk_comm_stats(self.ctx.handle, C.byref(a), C.byref(p), C.byref(g)))
        return dict(allreduce=a.value, p2p_groups=p.value, gather=g.value)

    def barrier(self):
        check(self.ctx._lib.kk_comm_barrier(self.ctx.handle))

    def allreduce_scalar(self, value: float, op: str = "max") -> float:
        """all-reduce one host double through the communicator (e.g. the max of a per-rank time)"""
        tmp = DeviceBasis(1, 1, self.ctx)
        tmp.upload(0, np.array([float(value)]))
        n, ld, cap, ptr = tmp.info()
        check(self.ctx._lib.kk_comm_allreduce(self.ctx.handle, C.c_void_p(ptr), 1, {"sum": 0, "max": 1, "min": 2}[op]))
        out = float(tmp.download(0)[0])
        tmp.free()
        return out

    def close(self):
        if getattr(self, "_open", False) and self.ctx.handle:
            check(self.ctx._lib.kk_comm_destroy(self.ctx.handle))
        self._open = False


def _csr_i64(A):
    import scipy.sparse as sp
    A = sp.csr_matrix(A)
    return (A, np.ascontiguousarray(A.indptr, dtype=np.int64), np.ascontiguousarray(A.indices, dtype=np.int64),
            np.ascontiguousarray(A.data, dtype=np.float64))


class NativeShardedOperator(SparseOperator):
    """This rank's rows of a square global operator (scipy matrix n_local x n_global with GLOBAL column indices); the
    ghost-exchange plan is negotiated inside kk_csr_create_sharded (a collective call).  Usable wherever a SparseOperator
    is: applies take and produce local blocks."""

    def __init__(self, A_rows, part: Partition, ctx: Context, symmetric: bool = False):
        self.ctx, self._lib = ctx, ctx._lib
        A, rowptr, col, val = _csr_i64(A_rows)
        nl = part.n_local
        assert A.shape == (nl, part.n_global), (A.shape, nl, part.n_global)
        offs = np.ascontiguousarray(part.offsets, dtype=np.int64)
        h = C.c_void_p()
        check(self._lib.kk_csr_create_sharded(ctx.handle, nl, offs.ctypes.data_as(_lib.c_i64p), A.nnz,
                                              rowptr.ctypes.data_as(_lib.c_i64p), col.ctypes.data_as(_lib.c_i64p),
                                              val.ctypes.data_as(_lib.c_dp), 0, _lib.KK_OP_SYMMETRIC if symmetric else 0,
                                              C.byref(h)))
        self.handle = h
        self.part = part
        self.shape = (nl, nl)   # as seen by the iterators: local rows x local columns
        self.symmetric = bool(symmetric)
        self.nnz = int(A.nnz)


class NativeShardedRectOperator(SparseOperator):
    """This rank's rows of a rectangular global map for the sharded GKL / svdsolve (kk_csr_create_sharded_rect): U-vectors
    follow the row partition, V-vectors are sharded evenly (shape[1] = this rank's share)."""

    def __init__(self, A_rows, ncols_global: int, ctx: Context):
        self.ctx, self._lib = ctx, ctx._lib
        A, rowptr, col, val = _csr_i64(A_rows)
        assert A.shape[1] == ncols_global
        h, ncl = C.c_void_p(), C.c_int64()
        check(self._lib.kk_csr_create_sharded_rect(ctx.handle, A.shape[0], int(ncols_global), A.nnz,
                                                   rowptr.ctypes.data_as(_lib.c_i64p), col.ctypes.data_as(_lib.c_i64p),
                                                   val.ctypes.data_as(_lib.c_dp), 0, C.byref(h), C.byref(ncl)))
        self.handle = h
        self.shape = (A.shape[0], int(ncl.value))
        self.ncols_global = int(ncols_global)
        self.symmetric = False
        self.nnz = int(A.nnz)
