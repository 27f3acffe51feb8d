"""TEST INFRASTRUCTURE -- Python exercisers of the two caller-owned-communicator mechanisms of the C ABI:

  * the split-phase entry points (kk_apply_fused_dev2, kk_project_dev, kk_unproject_devcoef, kk_lanczos_coef_dev, ...):
    `DistLanczosIterator` / `DistGKLIterator` keep their communication buffers in torch tensors and call
    torch.distributed between the library's half-steps.  With the NumPy `CheckerBackend` (tests/dist_checker_backend.py)
    they run under gloo with world_size 2 on a box without a GPU and check the row partition, the ghost plan and the
    all-reduce placement of a sharded step (tests/test_dist_gloo.py, tests/bench_checker.py);
  * the hooks (kk_ctx_set_allreduce, kk_op_set_halo_hook, kk_ctx_set_workspace): `ShardedContext` / `ShardedOperator`
    (tests/test_gpu_sharded_hooks.py: two logical ranks as two threads on one GPU).

The product's multi-GPU mechanism is krylovkit_hip.dist (RCCL inside the library); nothing in the package imports this
file."""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass
from typing import List, Optional

import numpy as np

from krylovkit_hip import _lib
from krylovkit_hip._lib import check
from krylovkit_hip.core import Context, DeviceBasis, KrylovDefaults, Orthogonalizer, SparseOperator
from krylovkit_hip.dist import Partition
from krylovkit_hip.factorizations import LanczosFactorization


# ------------------------------------------------------------------------------ backend
class HipBackend:
    """Local compute engine = libkrylov_hip.so; communication buffers = torch device tensors
    whose raw pointers are handed to the split-phase C entry points (kk_*_dev)."""

    name = "hip"

    def __init__(self, device_index: int = 0):
        import torch

        if not torch.cuda.is_available():
            raise _lib.NoDeviceError(_lib.KK_ERR_NO_DEVICE, "HipBackend needs a GPU; there is no CPU fallback")
        self.torch = torch
        self.device = torch.device("cuda", device_index)
        torch.cuda.set_device(self.device)
        self.ctx = Context(device_index)
        # One explicit (non-default) HIP stream shared by libkrylov_hip's kernels, torch's copies and
        # the RCCL collectives, so everything is stream-ordered without host syncs.  (The legacy
        # default stream has handle 0, which kk_ctx_set_stream reads as "use the context's own
        # stream" -- and a non-blocking stream does not synchronise with the default one.)
        self.stream = torch.cuda.Stream(self.device)
        torch.cuda.set_stream(self.stream)
        assert self.stream.cuda_stream != 0
        self.ctx.set_stream(self.stream.cuda_stream)
        self._lib = self.ctx._lib

    # buffers
    def alloc(self, count: int, dtype="float64"):
        return self.torch.zeros(count, dtype=getattr(self.torch, dtype), device=self.device)

    def to_host(self, t) -> np.ndarray:
        return t.cpu().numpy()

    def fetch_begin(self, t):
        """Start the read-back of a few device scalars (pinned buffer + event on the shared stream); work enqueued
        afterwards -- the speculative next-step apply -- is NOT waited for by fetch_end."""
        n = t.numel()
        if getattr(self, "_pin", None) is None or self._pin.numel() < n:
            self._pin = self.torch.empty(max(n, 16), dtype=self.torch.float64, pin_memory=True)
            self._pin_ev = self.torch.cuda.Event()
        self._pin[:n].copy_(t, non_blocking=True)
        self._pin_ev.record(self.stream)
        return n

    def fetch_end(self, n) -> np.ndarray:
        self._pin_ev.synchronize()
        return self._pin[:n].numpy().copy()

    def from_host_i64(self, a: np.ndarray):
        return self.torch.as_tensor(np.ascontiguousarray(a, dtype=np.int64), device=self.device)

    def sync(self):
        self.ctx.sync()

    # objects
    def make_basis(self, n_local: int, capacity: int):
        return DeviceBasis(n_local, capacity, self.ctx)

    def make_operator(self, A_local, n_local: int, ghost):
        op = SparseOperator(A_local, self.ctx)
        n_ghost = A_local.shape[1] - n_local
        check(self._lib.kk_op_set_ghost(op.handle, n_local, n_ghost, C.c_void_p(ghost.data_ptr() if n_ghost else 0)))
        op._ghost_keepalive = ghost
        return op

    def upload(self, basis, col, x):
        basis.upload(col, x)

    def copy_vec(self, dst, dcol, src, scol):
        check(self._lib.kk_vec_copy_scal(dst.handle, dcol, src.handle, scol, 1.0))

    def download(self, basis, col):
        return basis.download(col)

    def download_device(self, basis, col, t):
        check(self._lib.kk_basis_download_device(basis.handle, col, C.c_void_p(t.data_ptr())))

    def upload_device(self, basis, col, t):
        check(self._lib.kk_basis_upload_device(basis.handle, col, C.c_void_p(t.data_ptr())))

    def spmv(self, op, transpose, bx, cx, by, cy):
        check(self._lib.kk_spmv(op.handle, int(transpose), bx.handle, cx, by.handle, cy))

    # split-phase compute (all stream-ordered, no host sync)
    def gather(self, basis, col, idx, out):
        check(self._lib.kk_gather(basis.handle, col, C.c_void_p(idx.data_ptr()), idx.numel(), C.c_void_p(out.data_ptr())))

    def scal(self, basis, col, a: float):
        check(self._lib.kk_vec_scal(basis.handle, col, a))

    def copy_scal(self, basis, cy, cx, a: float):
        check(self._lib.kk_vec_copy_scal(basis.handle, cy, basis.handle, cx, a))

    def apply_fused(self, op, basis, col_v, col_prev, col_w, beta_old, dot_mode, out, xscale=None, bprev=None):
        check(self._lib.kk_apply_fused_dev2(op.handle, basis.handle, col_v, col_prev, col_w,
                                            C.c_void_p(xscale.data_ptr() if xscale is not None else 0),
                                            C.c_void_p(bprev.data_ptr() if bprev is not None else 0), beta_old, dot_mode,
                                            C.c_void_p(out.data_ptr())))

    def unproject_dev(self, basis, col_y, c0, m, coef_t, alpha, beta, nrm_out):
        check(self._lib.kk_unproject_devcoef(basis.handle, col_y, basis.handle, c0, m, C.c_void_p(coef_t.data_ptr()), alpha,
                                             beta, C.c_void_p(nrm_out.data_ptr() if nrm_out is not None else 0)))

    def project(self, basis, c0, m, col_x, col_rhs2, out):
        check(self._lib.kk_project_dev(basis.handle, c0, m, basis.handle, col_x, col_rhs2, C.c_void_p(out.data_ptr())))

    def unproject(self, basis, col_y, c0, m, coef, alpha, beta, nrm_out):
        ca = np.ascontiguousarray(coef, dtype=np.float64)
        check(self._lib.kk_unproject_dev(basis.handle, col_y, basis.handle, c0, m, ca.ctypes.data_as(_lib.c_dp), alpha,
                                         beta, C.c_void_p(nrm_out.data_ptr() if nrm_out is not None else 0)))

    def lanczos_coef(self, buf, L, m: int, lowsync: bool, coef, res):
        """coefficient algebra of a sharded Lanczos step in one launch (kk_lanczos_coef_dev)"""
        check(self._lib.kk_lanczos_coef_dev(self.ctx.handle, C.c_void_p(buf.data_ptr()),
                                            C.c_void_p(L.data_ptr()) if L is not None else None,
                                            int(L.shape[1]) if L is not None else 0, m, int(bool(lowsync)),
                                            C.c_void_p(coef.data_ptr()), C.c_void_p(res.data_ptr())))

    def norm_scalars(self, nrm2, sc, res2):
        """sc = {1/sqrt(nrm2), sqrt(nrm2)}, res2 = nrm2 on the device (kk_norm_scalars_dev)"""
        check(self._lib.kk_norm_scalars_dev(self.ctx.handle, C.c_void_p(nrm2.data_ptr()), C.c_void_p(sc.data_ptr()),
                                            C.c_void_p(res2.data_ptr())))

    def dot(self, basis, cx, cy, out):
        check(self._lib.kk_dot_dev(basis.handle, cx, basis.handle, cy, C.c_void_p(out.data_ptr())))

    def nrm2(self, basis, cx, out3):
        check(self._lib.kk_nrm2_dev(basis.handle, cx, C.c_void_p(out3.data_ptr())))


# ------------------------------------------------------------------------------ operator
class DistSparseOperator:
    """Row block of a global sparse matrix with a ghost-column exchange plan.

    `A_rows`: scipy CSR holding this rank's rows with GLOBAL column indices
    (shape n_local x n_global).  Columns owned by other ranks become ghost columns
    n_local .. n_local+n_ghost-1 of the local operator; before every apply the owning ranks
    send exactly those entries (point-to-point `batch_isend_irecv`; for a 5-point stencil
    that is one grid row of nx doubles per neighbour)."""

    def __init__(self, A_rows, part: Partition, backend, group=None):
        import scipy.sparse as sp
        import torch.distributed as dist

        self.part, self.backend, self.group, self.dist = part, backend, group, dist
        A = sp.csr_matrix(A_rows)
        nl = part.n_local
        assert A.shape == (nl, part.n_global), (A.shape, nl, part.n_global)
        cols = A.indices.astype(np.int64)
        mine = (cols >= part.lo) & (cols < part.hi)
        needed = np.unique(cols[~mine])  # sorted global ids -> grouped by owner
        owner = np.searchsorted(part.offsets, needed, side="right") - 1
        # renumber columns: local -> col - lo ; ghost -> nl + position in `needed`
        newcols = np.where(mine, cols - part.lo, nl + np.searchsorted(needed, cols))
        A_loc = sp.csr_matrix((A.data, newcols.astype(np.int32), A.indptr), shape=(nl, nl + len(needed)))
        self.n_ghost = len(needed)
        # exchange of request lists (setup only)
        world = part.world
        requests: List[Optional[np.ndarray]] = [None] * world
        if world > 1:
            dist.all_gather_object(requests, needed, group=group)
        else:
            requests = [needed]
        self.recv_counts = [int(np.sum(owner == q)) for q in range(world)]
        send_lists = []
        for q in range(world):
            if q == part.rank:
                send_lists.append(np.zeros(0, dtype=np.int64))
                continue
            req = requests[q]
            sel = req[(req >= part.lo) & (req < part.hi)] - part.lo
            send_lists.append(sel.astype(np.int64))
        self.send_counts = [len(s) for s in send_lists]
        total_send = sum(self.send_counts)
        self.send_idx = backend.from_host_i64(np.concatenate(send_lists) if total_send else np.zeros(0, dtype=np.int64))
        self.sendbuf = backend.alloc(max(total_send, 1))
        self.ghost = backend.alloc(max(self.n_ghost, 1))
        self.local = backend.make_operator(A_loc, nl, self.ghost)
        self.nnz_local = int(A.nnz)

    def halo_exchange(self, basis, col):
        """Fill the ghost buffer with the off-rank entries of vector (basis, col)."""
        part, dist = self.part, self.dist
        if part.world == 1 or (self.n_ghost == 0 and sum(self.send_counts) == 0):
            return
        if sum(self.send_counts):
            self.backend.gather(basis, col, self.send_idx, self.sendbuf)
        ops, so, ro = [], 0, 0
        for q in range(part.world):
            if self.send_counts[q]:
                ops.append(dist.P2POp(dist.isend, self.sendbuf[so:so + self.send_counts[q]], q, group=self.group))
                so += self.send_counts[q]
            if self.recv_counts[q]:
                ops.append(dist.P2POp(dist.irecv, self.ghost[ro:ro + self.recv_counts[q]], q, group=self.group))
                ro += self.recv_counts[q]
        for w in dist.batch_isend_irecv(ops):
            w.wait()


# ------------------------------------------------------------------------------ Lanczos
@dataclass
class DistLanczosIterator:
    """Row-sharded LanczosIterator (factorizations/lanczos.jl:129-153).  `x0_local` is this
    rank's block of the start vector.  All six orthogonalisers; mgs2 / mgsir sweeps in the low-sync form.
    The pass count of cgsir / mgsir (lanczos.jl:339-376) is decided from all-reduced norms, so every rank
    takes the same number of passes."""
    operator: DistSparseOperator
    x0_local: np.ndarray
    orth: Orthogonalizer = KrylovDefaults.orth
    capacity: int = KrylovDefaults.krylovdim + 2
    keepvecs: bool = True

    def __post_init__(self):
        if self.orth.name not in ("cgs", "mgs", "cgs2", "mgs2", "cgsir", "mgsir"):
            raise ValueError(f"unknown orthogonalizer {self.orth.name}")
        be = self.operator.backend
        self.backend = be
        import torch
        self.torch = torch
        self.bufs = [be.alloc(2 * 256 + 8), be.alloc(2 * 256 + 8)]   # [alpha0 | p (m) | g (m)], used alternately: the
        self.buf = self.bufs[0]                                       # speculative SpMV writes alpha0 into the other one
        self.nbuf = be.alloc(4)            # [|w|^2, sqrt, 1/sqrt, spare]
        self.coef = be.alloc(256)          # coefficients of the update, formed on the device
        self.res = be.alloc(4)             # [alpha0, s_m, |w|^2]: the one read-back per expand!
        self.sc = be.alloc(2)              # [1/beta, beta] of the finished iteration (device scalars)
        self._spec = None
        self.Ldev = None                   # strictly-lower Gram matrix of the basis (low-sync MGS), on the device

    def _allreduce(self, t):
        part = self.operator.part
        if part.world > 1:
            self.operator.dist.all_reduce(t, group=self.operator.group)

    # initialize(iter) -- lanczos.jl:180-222 with every inner product all-reduced
    def initialize(self, V=None) -> LanczosFactorization:
        be, op = self.backend, self.operator
        nl = op.part.n_local
        if V is None:
            V = be.make_basis(nl, self.capacity)
        if isinstance(self.x0_local, tuple):   # (basis, column) already resident on the device
            be.copy_vec(V, 0, self.x0_local[0], self.x0_local[1])
        else:
            be.upload(V, 0, np.asarray(self.x0_local, dtype=np.float64))
        be.nrm2(V, 0, self.nbuf)
        self._allreduce(self.nbuf[0:1])
        op.halo_exchange(V, 0)
        be.apply_fused(op.local, V, 0, -1, 1, 0.0, 1, self.buf)
        self._allreduce(self.buf[0:1])
        beta0 = float(np.sqrt(be.to_host(self.nbuf[0:1])[0]))
        if beta0 == 0.0:
            raise _lib.KrylovHipError(_lib.KK_ERR_ZERO_NORM, "initial vector should not have norm zero")
        alpha = float(be.to_host(self.buf[0:1])[0]) / (beta0 * beta0)
        be.scal(V, 0, 1.0 / beta0)
        be.scal(V, 1, 1.0 / beta0)
        ir = self.orth.name in ("cgsir", "mgsir")
        if ir:                                                     # beta_old = norm(r) before the projection, :195
            be.nrm2(V, 1, self.nbuf)
            self._allreduce(self.nbuf[0:1])
            beta_old = float(np.sqrt(be.to_host(self.nbuf[0:1])[0]))
        be.unproject(V, 1, 0, 1, [alpha], -1.0, 1.0, self.nbuf)   # r -= alpha v ; |r|^2 partial
        self._allreduce(self.nbuf[0:1])
        beta = float(np.sqrt(be.to_host(self.nbuf[0:1])[0]))

        def again(alpha):
            be.dot(V, 0, 1, self.buf)
            self._allreduce(self.buf[0:1])
            da = float(be.to_host(self.buf[0:1])[0])
            be.unproject(V, 1, 0, 1, [da], -1.0, 1.0, self.nbuf)
            self._allreduce(self.nbuf[0:1])
            return alpha + da, float(np.sqrt(be.to_host(self.nbuf[0:1])[0]))

        if self.orth.name in ("cgs2", "mgs2"):  # :200-204
            alpha, beta = again(alpha)
        elif ir:                                # :205-213
            while float(np.finfo(np.float64).eps) < beta < self.orth.eta * beta_old:
                beta_old = beta
                alpha, beta = again(alpha)
        V.length = 1
        self.Ldev = be.alloc(self.capacity * self.capacity).reshape(self.capacity, self.capacity)
        self.gram_rows = 1
        self._spec = None
        return LanczosFactorization(1, V, [alpha], [beta])

    def _speculate(self, st, k_next: int, beta: float, dot_mode: int):
        """Enqueue the halo exchange + SpMV of iteration k_next (on the RAW residual, the 1/beta
        scale applied on the fly from the all-reduced device norm) before the host reads beta."""
        V = st.V
        self._spec = None
        if k_next + 2 > V.capacity:
            return
        be, op = self.backend, self.operator
        nxt = self.bufs[1] if self.buf is self.bufs[0] else self.bufs[0]
        op.halo_exchange(V, k_next)
        be.apply_fused(op.local, V, k_next, k_next - 1, k_next + 1, 0.0, dot_mode, nxt,
                       xscale=self.sc[0:1], bprev=self.sc[1:2])
        self._spec = [k_next, None, id(V)]

    # expand!(iter, state) -- lanczos.jl:250-272 + lanczosrecurrence :295-338
    def expand(self, st: LanczosFactorization) -> LanczosFactorization:
        be, op, V = self.backend, self.operator, st.V
        torch = self.torch
        k = len(V)
        if k + 2 > V.capacity:
            raise RuntimeError(f"Lanczos slab of capacity {V.capacity} is full at k={k}")
        m = k + 1
        beta_old = st.normres
        name = self.orth.name
        dot_mode = 1 if name in ("cgs", "cgs2", "cgsir") else 2
        ir = name in ("cgsir", "mgsir")
        hit = self._spec is not None and self._spec == [k, beta_old, id(V)]
        self._spec = None
        be.scal(V, k, 1.0 / beta_old)                       # V = push!(V, scale!!(r, 1/beta_old))
        if hit:                                              # the SpMV of this step is already done: its alpha0 sits
            self.buf = self.bufs[1] if self.buf is self.bufs[0] else self.bufs[0]   # in the other buffer
        else:
            op.halo_exchange(V, k)
            be.apply_fused(op.local, V, k, k - 1, k + 1, beta_old, dot_mode, self.buf)
        if name in ("cgs", "mgs", "cgsir", "mgsir"):
            self._allreduce(self.buf[0:1])
            self.res[0:1] = self.buf[0:1]
            self.res[1:2] = 0.0
            be.unproject_dev(V, k + 1, k, 1, self.buf[0:1], -1.0, 1.0, self.nbuf)
        else:
            be.project(V, 0, m, k + 1, k, self.buf[1:1 + 2 * m])
            self._allreduce(self.buf[0:1 + 2 * m])           # ONE all-reduce: alpha0, V'w, V'v
            lowsync = name == "mgs2"
            if lowsync:
                # low-sync MGS: (I + L) s = V'(w - alpha0 v), L = strictly lower Gram matrix of V; V'v is its newest row
                if self.gram_rows < k:
                    raise RuntimeError("Gram rows out of date (basis changed outside expand); call recompute_gram()")
                self.gram_rows = k + 1
            # s = V'w - alpha0 V'v [then the exact triangular solve], alpha0 folded into the last coefficient, and the two
            # scalars the host needs -- one launch between the two all-reduces
            be.lanczos_coef(self.buf, self.Ldev if lowsync else None, m, lowsync, self.coef, self.res)
            be.unproject_dev(V, k + 1, 0, m, self.coef, -1.0, 1.0, self.nbuf)
        self._allreduce(self.nbuf[0:1])
        be.norm_scalars(self.nbuf, self.sc, self.res[2:3])  # 1/beta, beta for the speculative apply; |w|^2 for the host
        if ir:
            alpha, beta = self._refine(st, k, beta_old)      # lanczos.jl:343-355 / :362-375, one host sync per extra pass
            self._speculate(st, k + 1, 0.0, dot_mode)
        else:
            tok = be.fetch_begin(self.res[0:3])              # read-back queued BEFORE the speculative work ...
            self._speculate(st, k + 1, 0.0, dot_mode)       # ... which keeps the GPU / links busy meanwhile
            h = be.fetch_end(tok)                            # the ONE host synchronisation of this expand!
            alpha = float(h[0] + h[1])
            beta = float(np.sqrt(h[2]))
        if self._spec is not None:
            self._spec[1] = beta
        st.alphas.append(alpha)
        st.betas.append(beta)
        V.length = m
        st.k += 1
        return st

    def _refine(self, st: LanczosFactorization, k: int, beta_old: float):
        """The `while eps < beta < eta*nold` loop of the IR recurrences (lanczos.jl:343-355, :362-375): every extra
        pass = project (one all-reduce) + coefficient kernel + unproject (one all-reduce of |w|^2); the loop condition
        is evaluated on all-reduced scalars, identical on every rank."""
        be, V = self.backend, st.V
        m = k + 1
        lowsync = self.orth.name == "mgsir"
        h = be.to_host(self.res[0:3])
        alpha, beta = float(h[0]), float(np.sqrt(h[2]))
        nold = float(np.sqrt(beta * beta + alpha * alpha + beta_old * beta_old))
        eps = float(np.finfo(np.float64).eps)
        while eps < beta < self.orth.eta * nold:
            nold = beta
            if lowsync:
                for i in range(max(self.gram_rows, 1), k):   # rows the refinement-free steps did not need
                    be.project(V, 0, i, i, -1, self.buf[0:i])
                    self._allreduce(self.buf[0:i])
                    self.Ldev[i, :i] = self.buf[0:i]
                self.gram_rows = k + 1
            self.buf[0:1] = 0.0                               # no alpha0 term in a refinement pass
            be.project(V, 0, m, k + 1, k, self.buf[1:1 + 2 * m])
            self._allreduce(self.buf[0:1 + 2 * m])
            be.lanczos_coef(self.buf, self.Ldev if lowsync else None, m, lowsync, self.coef, self.res)
            be.unproject_dev(V, k + 1, 0, m, self.coef, -1.0, 1.0, self.nbuf)
            self._allreduce(self.nbuf[0:1])
            be.norm_scalars(self.nbuf, self.sc, self.res[2:3])
            h = be.to_host(self.res[0:3])
            alpha += float(h[1])                               # alpha += s[end]
            beta = float(np.sqrt(h[2]))
        return alpha, beta

    def recompute_gram(self, st: LanczosFactorization):
        """After a restart transformed the basis: rebuild the strictly-lower Gram rows."""
        be, V = self.backend, st.V
        k = len(V)
        self._spec = None
        self.Ldev.zero_()
        for i in range(1, k):
            be.project(V, 0, i, i, -1, self.buf[0:i])
            self._allreduce(self.buf[0:i])
            self.Ldev[i, :i] = self.buf[0:i]
        self.gram_rows = max(k, 1)


# ------------------------------------------------------------------------------ GKL (config 4)
class _LowSyncGram:
    """Host copy of the strictly-lower Gram matrix of one sharded basis (low-sync MGS)."""

    def __init__(self, cap: int):
        self.L = np.zeros((cap, cap))
        self.rows = 1

    def solve(self, p: np.ndarray) -> np.ndarray:
        s = p.copy()
        for i in range(1, len(s)):
            s[i] -= self.L[i, :i] @ s[:i]
        return s


class DistRectOperator:
    """Row block of a rectangular sparse map A (m x n) for the sharded GKL (cfg 4):
    U-vectors (length m) are sharded by the rows of A, V-vectors (length n) by an even partition.
      A v   : all-gather of the short vector v (n doubles in total), local SpMV on the gathered buffer
      A' u  : local transposed SpMV (full-length partial), reduce-scatter (sum) onto the V shards."""

    def __init__(self, A_rows, row_part: Partition, col_part: Partition, backend, group=None):
        import scipy.sparse as sp
        import torch.distributed as dist

        self.row_part, self.col_part, self.backend, self.group, self.dist = row_part, col_part, backend, group, dist
        A = sp.csr_matrix(A_rows)
        assert A.shape == (row_part.n_local, col_part.n_global)
        counts = np.diff(col_part.offsets)
        assert np.all(counts == counts[0]), "V partition must be even (all-gather / reduce-scatter of equal shards)"
        self.n = col_part.n_global
        self.vfull = backend.alloc(self.n)            # gathered v (the operator's ghost buffer)
        self.zfull = backend.alloc(self.n)            # full-length partial of A'u
        self.vloc = backend.alloc(col_part.n_local)   # this rank's V shard staging
        self.local = backend.make_operator(A, 0, self.vfull)   # ghost-only: all columns read from vfull
        self.zbasis = backend.make_basis(self.n, 1)

    def apply_normal(self, Vb, cv, Ub, cu):
        """U[cu] = (A v) restricted to my rows;  v = V[cv] sharded."""
        be, dist = self.backend, self.dist
        be.download_device(Vb, cv, self.vloc)
        if self.col_part.world > 1:
            dist.all_gather_into_tensor(self.vfull, self.vloc, group=self.group)
        else:
            self.vfull.copy_(self.vloc)
        be.spmv(self.local, False, Vb, cv, Ub, cu)

    def apply_adjoint(self, Ub, cu, Vb, cv):
        """V[cv] = my shard of A' u;  u = U[cu] sharded by rows."""
        be, dist = self.backend, self.dist
        be.spmv(self.local, True, Ub, cu, self.zbasis, 0)
        be.download_device(self.zbasis, 0, self.zfull)
        if self.col_part.world > 1:
            try:
                dist.reduce_scatter_tensor(self.vloc, self.zfull, group=self.group)
            except (RuntimeError, NotImplementedError):   # backend without reduce_scatter (gloo): all-reduce + slice
                dist.all_reduce(self.zfull, group=self.group)
                self.vloc.copy_(self.zfull[self.col_part.lo:self.col_part.hi])
        else:
            self.vloc.copy_(self.zfull)
        be.upload_device(Vb, cv, self.vloc)


@dataclass
class DistGKLIterator:
    """Row-sharded GKLIterator (factorizations/gkl.jl:137-152).  Orthogonalisers: cgs, mgs (no
    re-orthogonalisation, gkl.jl:294-307), cgs2 (:308-323), mgs2 in its low-sync form (:324-346), cgsir / mgsir
    (:347-404; pass counts decided from all-reduced norms)."""
    operator: DistRectOperator
    u0_local: np.ndarray
    orth: Orthogonalizer = KrylovDefaults.orth
    capacity: int = KrylovDefaults.krylovdim + 2

    def __post_init__(self):
        if self.orth.name not in ("cgs", "mgs", "cgs2", "mgs2", "cgsir", "mgsir"):
            raise ValueError(f"unknown orthogonalizer {self.orth.name}")
        self.backend = self.operator.backend
        self.buf = self.backend.alloc(2 * 256 + 8)
        self.nbuf = self.backend.alloc(4)

    def _allreduce(self, t):
        if self.operator.row_part.world > 1:
            self.operator.dist.all_reduce(t, group=self.operator.group)

    def _norm(self, basis, col) -> float:
        be = self.backend
        be.nrm2(basis, col, self.nbuf)
        self._allreduce(self.nbuf[0:1])
        return float(np.sqrt(be.to_host(self.nbuf[0:1])[0]))

    def _sweep(self, basis, m, col, gram: Optional[_LowSyncGram]):
        """One orthogonalisation pass of (basis, col) against columns [0, m): classical (gram None)
        or low-sync modified; the Gram row of the newest basis vector rides along."""
        be = self.backend
        if gram is not None:
            for i in range(max(gram.rows, 1), m - 1):        # rows skipped while no refinement pass was needed (IR)
                be.project(basis, 0, i, i, -1, self.buf[0:i])
                self._allreduce(self.buf[0:i])
                gram.L[i, :i] = be.to_host(self.buf[0:i])
                gram.rows = i + 1
        ride = gram is not None and gram.rows == m - 1 and m >= 2
        be.project(basis, 0, m, col, (m - 1) if ride else -1, self.buf[0:2 * m])
        self._allreduce(self.buf[0:(2 * m if ride else m)])
        h = be.to_host(self.buf[0:2 * m])
        p = h[:m]
        if gram is not None:
            if ride:
                gram.L[m - 1, :m - 1] = h[m:2 * m - 1]
                gram.rows = m
            assert gram.rows >= m, "Gram rows out of date"
            p = gram.solve(p)
        be.unproject(basis, col, 0, m, p, -1.0, 1.0, self.nbuf)
        self._allreduce(self.nbuf[0:1])
        return float(np.sqrt(be.to_host(self.nbuf[0:1])[0]))

    def initialize(self, V=None):
        """initialize(iter::GKLIterator) (gkl.jl:183-215), every inner product all-reduced."""
        from krylovkit_hip.factorizations import GKLFactorization
        be, op = self.backend, self.operator
        U = be.make_basis(op.row_part.n_local, self.capacity)
        Vb = be.make_basis(op.col_part.n_local, self.capacity)
        be.upload(U, 0, np.asarray(self.u0_local, dtype=np.float64))
        beta0 = self._norm(U, 0)
        if beta0 == 0.0:
            raise _lib.KrylovHipError(_lib.KK_ERR_ZERO_NORM, "initial vector should not have norm zero")
        op.apply_adjoint(U, 0, Vb, 0)                    # v0 = A' u0
        alpha = self._norm(Vb, 0) / beta0
        op.apply_normal(Vb, 0, U, 1)                     # A v0
        be.dot(U, 0, 1, self.buf)
        self._allreduce(self.buf[0:1])
        a2 = float(be.to_host(self.buf[0:1])[0]) / beta0 ** 2
        if not abs(a2 - alpha * alpha) <= np.sqrt(np.finfo(float).eps) * max(abs(a2), alpha * alpha):
            raise ValueError("operator and its adjoint are not compatible")   # gkl.jl:192
        be.scal(U, 0, 1.0 / beta0)
        be.scal(Vb, 0, 1.0 / (alpha * beta0))
        be.scal(U, 1, 1.0 / (alpha * beta0))
        be.unproject(U, 1, 0, 1, [alpha], -1.0, 1.0, self.nbuf)   # r -= alpha u
        self._allreduce(self.nbuf[0:1])
        beta = float(np.sqrt(be.to_host(self.nbuf[0:1])[0]))
        U.length = Vb.length = 1
        self.gram_u, self.gram_v = _LowSyncGram(self.capacity), _LowSyncGram(self.capacity)
        return GKLFactorization(1, U, Vb, [alpha], [beta])

    def expand(self, st):
        """expand!(iter::GKLIterator, state) (gkl.jl:246-269) + gklrecurrence (:294-346)."""
        be, op, U, V = self.backend, self.operator, st.U, st.V
        k = len(U)
        if k + 2 > U.capacity or k + 1 > V.capacity:
            raise RuntimeError(f"GKL slabs are full at k={k}")
        name = self.orth.name
        beta_old = st.normres
        be.scal(U, k, 1.0 / beta_old)                    # U = push!(U, scale!!(r, 1/beta_old))
        op.apply_adjoint(U, k, V, k)                     # v = A' u
        be.unproject(V, k, k - 1, 1, [beta_old], -1.0, 1.0, self.nbuf)   # v -= beta_old V[end]; |v|^2 partial
        eps = float(np.finfo(np.float64).eps)
        if name == "mgs2":
            alpha = self._sweep(V, k, k, self.gram_v)    # for q in V: orthogonalize!!(v, q, MGS)   :330-335
        else:
            self._allreduce(self.nbuf[0:1])
            alpha = float(np.sqrt(be.to_host(self.nbuf[0:1])[0]))
            if name in ("cgsir", "mgsir"):                # :353-358 (no eps guard for cgsir, as the reference) / :380-386
                nold = float(np.sqrt(alpha * alpha + beta_old * beta_old))
                while (name == "cgsir" or eps < alpha) and alpha < self.orth.eta * nold:
                    nold = alpha
                    alpha = self._sweep(V, k, k, self.gram_v if name == "mgsir" else None)
        be.scal(V, k, 1.0 / alpha)
        op.apply_normal(V, k, U, k + 1)                  # r = A v
        be.unproject(U, k + 1, k, 1, [alpha], -1.0, 1.0, self.nbuf)      # r -= alpha u
        if name == "cgs2":
            beta = self._sweep(U, k + 1, k + 1, None)    # r, = orthogonalize!!(r, U, CGS)   :320
        elif name == "mgs2":
            beta = self._sweep(U, k + 1, k + 1, self.gram_u)   # :341-343
        else:
            self._allreduce(self.nbuf[0:1])
            beta = float(np.sqrt(be.to_host(self.nbuf[0:1])[0]))
            if name in ("cgsir", "mgsir"):                # :363-372 / :391-402
                nold = float(np.sqrt(alpha * alpha + beta * beta))
                while eps < beta < self.orth.eta * nold:
                    nold = beta
                    beta = self._sweep(U, k + 1, k + 1, self.gram_u if name == "mgsir" else None)
        st.alphas.append(alpha)
        st.betas.append(beta)
        U.length = V.length = k + 1
        st.k += 1
        return st


# ------------------------------------------------------------------------------ hook-based sharding
# libkrylov_hip calls an all-reduce hook after every finalize kernel and a halo hook before every
# sparse apply (include/krylov_hip.h, "row-sharded operation").  Installing them turns the ORDINARY
# objects -- SparseOperator on the local rows, DeviceBasis = local shard, Lanczos / Arnoldi /
# BlockLanczos iterators, eigsolve / linsolve / eigsolve_block and all six orthogonalisers -- into
# their row-sharded versions: the host control flow is unchanged and sees identical scalars on
# every rank.  (The split-phase iterators above remain the latency-optimised path for Lanczos / GKL.)
ALLREDUCE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_int64)
HALO_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p)


class TorchCollective:
    """Collectives over torch.distributed (backend "nccl" = RCCL over xGMI)."""

    def __init__(self, group=None):
        import torch.distributed as dist
        self.dist, self.group = dist, group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0

    def all_reduce(self, t):
        if self.world > 1:
            self.dist.all_reduce(t, group=self.group)

    def all_gather_object(self, obj):
        if self.world == 1:
            return [obj]
        out = [None] * self.world
        self.dist.all_gather_object(out, obj, group=self.group)
        return out

    def exchange(self, sendbuf, send_counts, recvbuf, recv_counts):
        dist = self.dist
        ops, so, ro = [], 0, 0
        for q in range(self.world):
            if send_counts[q]:
                ops.append(dist.P2POp(dist.isend, sendbuf[so:so + send_counts[q]], q, group=self.group))
                so += send_counts[q]
            if recv_counts[q]:
                ops.append(dist.P2POp(dist.irecv, recvbuf[ro:ro + recv_counts[q]], q, group=self.group))
                ro += recv_counts[q]
        if ops:
            for w in dist.batch_isend_irecv(ops):
                w.wait()


class ShardedContext:
    """One rank of a row-sharded run: a HipBackend context with the library's all-reduce hook
    installed; the two device scratch areas the reductions land in are torch tensors."""

    def __init__(self, coll, device_index: int = 0, backend: Optional[HipBackend] = None):
        self.coll = coll
        self.backend = backend or HipBackend(device_index)
        self.ctx = self.backend.ctx
        lib = self.ctx._lib
        nws, nblk = C.c_int64(), C.c_int64()
        check(lib.kk_ctx_workspace_size(self.ctx.handle, C.byref(nws), C.byref(nblk)))
        self.ws = self.backend.alloc(nws.value)
        self.blk = self.backend.alloc(nblk.value)
        check(lib.kk_ctx_set_workspace(self.ctx.handle, C.c_void_p(self.ws.data_ptr()), C.c_void_p(self.blk.data_ptr())))
        self._bufs = [(self.ws.data_ptr(), self.ws), (self.blk.data_ptr(), self.blk)]
        self.calls = 0

        def _cb(user, ptr, count):
            try:
                for base, t in self._bufs:
                    off = (ptr - base) // 8
                    if 0 <= off and off + count <= t.numel():
                        self.coll.all_reduce(t[off:off + count])
                        self.calls += 1
                        return 0
                return 2  # pointer outside the registered scratch areas
            except Exception:  # never let an exception cross the C boundary
                import traceback
                traceback.print_exc()
                return 1

        self._cb = ALLREDUCE_FN(_cb)
        check(lib.kk_ctx_set_allreduce(self.ctx.handle, self._cb, None))

    def close(self):
        check(self.ctx._lib.kk_ctx_set_allreduce(self.ctx.handle, None, None))
        check(self.ctx._lib.kk_ctx_set_workspace(self.ctx.handle, None, None))

    def operator(self, A_rows, part: Partition) -> "ShardedOperator":
        return ShardedOperator(A_rows, part, self)

    def basis(self, n_local: int, capacity: int) -> DeviceBasis:
        return DeviceBasis(n_local, capacity, self.ctx)


class ShardedOperator(SparseOperator):
    """SparseOperator on this rank's rows (global column indices) whose applies exchange the ghost
    entries through the library's halo hook -- usable wherever a SparseOperator is (iterators,
    eigsolve, linsolve, eigsolve_block)."""

    def __init__(self, A_rows, part: Partition, sctx: ShardedContext):
        import scipy.sparse as sp
        be, coll = sctx.backend, sctx.coll
        self.part, self.sctx = part, sctx
        A = sp.csr_matrix(A_rows)
        nl = part.n_local
        assert A.shape == (nl, part.n_global), (A.shape, nl, part.n_global)
        cols = A.indices.astype(np.int64)
        mine = (cols >= part.lo) & (cols < part.hi)
        needed = np.unique(cols[~mine])
        owner = np.searchsorted(part.offsets, needed, side="right") - 1
        newcols = np.where(mine, cols - part.lo, nl + np.searchsorted(needed, cols))
        A_loc = sp.csr_matrix((A.data, newcols.astype(np.int32), A.indptr), shape=(nl, nl + len(needed)))
        self.n_ghost = len(needed)
        requests = coll.all_gather_object(needed)
        self.recv_counts = [int(np.sum(owner == q)) for q in range(part.world)]
        send_lists = []
        for q in range(part.world):
            req = requests[q]
            sel = req[(req >= part.lo) & (req < part.hi)] - part.lo if q != part.rank else np.zeros(0, dtype=np.int64)
            send_lists.append(sel.astype(np.int64))
        self.send_counts = [len(s_) for s_ in send_lists]
        tot = sum(self.send_counts)
        self.send_idx = be.from_host_i64(np.concatenate(send_lists) if tot else np.zeros(0, dtype=np.int64))
        self.sendbuf = be.alloc(max(tot, 1))
        self.ghost = be.alloc(max(self.n_ghost, 1))
        super().__init__(A_loc, sctx.ctx)
        self.shape = (nl, nl)  # as seen by the iterators: local rows x local columns
        lib = self._lib
        check(lib.kk_op_set_ghost(self.handle, nl, self.n_ghost, C.c_void_p(self.ghost.data_ptr() if self.n_ghost else 0)))

        def _halo(user, xptr):
            try:
                if tot:
                    check(lib.kk_gather_ptr(sctx.ctx.handle, C.c_void_p(xptr), C.c_void_p(self.send_idx.data_ptr()), tot,
                                            C.c_void_p(self.sendbuf.data_ptr())))
                coll.exchange(self.sendbuf, self.send_counts, self.ghost, self.recv_counts)
                return 0
            except Exception:
                import traceback
                traceback.print_exc()
                return 1

        self._halo_cb = HALO_FN(_halo)
        if part.world > 1:
            check(lib.kk_op_set_halo_hook(self.handle, self._halo_cb, None))


