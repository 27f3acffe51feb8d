"""CPU checker backend for tests/splitphase_dist.py (TEST INFRASTRUCTURE): plain NumPy on torch CPU
tensors, same interface as its HipBackend, so the partition / halo-exchange / all-reduce logic of
the row-sharded path can run under gloo with world_size 2 on a machine without a GPU."""
import numpy as np
import torch


class _Basis:
    def __init__(self, n, cap):
        self.n, self.capacity, self.length = n, cap, 0
        self.cols = np.zeros((cap, n))

    def __len__(self):
        return self.length


class _Op:
    def __init__(self, A, n_local, ghost):
        self.A, self.n_local, self.ghost = A.tocsr(), n_local, ghost
        self.n_ghost = A.shape[1] - n_local


class CheckerBackend:
    name = "checker"

    def alloc(self, count, dtype="float64"):
        return torch.zeros(count, dtype=getattr(torch, dtype))

    def to_host(self, t):
        return t.numpy().copy()

    def fetch_begin(self, t):
        return t.numpy().copy()

    def fetch_end(self, tok):
        return tok

    def from_host_i64(self, a):
        return torch.as_tensor(np.ascontiguousarray(a, dtype=np.int64))

    def sync(self):
        pass

    def make_basis(self, n, cap):
        return _Basis(n, cap)

    def make_operator(self, A_local, n_local, ghost):
        return _Op(A_local, n_local, ghost)

    def upload(self, b, col, x):
        b.cols[col] = x

    def copy_vec(self, dst, dcol, src, scol):
        dst.cols[dcol] = src.cols[scol]

    def download(self, b, col):
        return b.cols[col].copy()

    def download_device(self, b, col, t):
        t[: b.n] = torch.from_numpy(b.cols[col])

    def upload_device(self, b, col, t):
        b.cols[col] = t.numpy()[: b.n].copy()

    def spmv(self, op, transpose, bx, cx, by, cy):
        if transpose:
            by.cols[cy] = op.A.T @ bx.cols[cx]
        else:
            x = np.concatenate([bx.cols[cx][: op.n_local], op.ghost.numpy()[: op.n_ghost]])
            by.cols[cy] = op.A @ x

    def gather(self, b, col, idx, out):
        out[: idx.numel()] = torch.from_numpy(b.cols[col][idx.numpy()])

    def scal(self, b, col, a):
        b.cols[col] *= a

    def copy_scal(self, b, cy, cx, a):
        b.cols[cy] = a * b.cols[cx]

    def apply_fused(self, op, b, cv, cprev, cw, beta_old, dot_mode, out, xscale=None, bprev=None):
        xs = float(xscale[0]) if xscale is not None else 1.0
        bp = float(bprev[0]) if bprev is not None else beta_old
        x = np.concatenate([b.cols[cv], op.ghost.numpy()[: op.n_ghost]])
        ax = (op.A @ x) * xs
        v = b.cols[cv] * xs
        w = ax.copy()
        if cprev >= 0:
            w -= bp * b.cols[cprev]
        if dot_mode == 1:
            out[0] = float(v @ ax)
        elif dot_mode == 2:
            out[0] = float(v @ w)
        b.cols[cw] = w

    def unproject_dev(self, b, cy, c0, m, coef_t, alpha, beta, nrm_out):
        self.unproject(b, cy, c0, m, coef_t.numpy()[:m], alpha, beta, nrm_out)

    def project(self, b, c0, m, cx, crhs2, out):
        V = b.cols[c0:c0 + m]
        out[:m] = torch.from_numpy(V @ b.cols[cx])
        if crhs2 >= 0:
            out[m:2 * m] = torch.from_numpy(V @ b.cols[crhs2])

    def unproject(self, b, cy, c0, m, coef, alpha, beta, nrm_out):
        y = beta * b.cols[cy] + alpha * (np.asarray(coef) @ b.cols[c0:c0 + m])
        b.cols[cy] = y
        if nrm_out is not None:
            nrm_out[0] = float(y @ y)

    def lanczos_coef(self, buf, L, m, lowsync, coef, res):
        h = buf.numpy()
        a0, p, g = h[0], h[1:1 + m].copy(), h[1 + m:1 + 2 * m].copy()
        s = p - a0 * g
        if lowsync:
            Lm = L.numpy()
            Lm[m - 1, :m - 1] = g[:m - 1]
            for j in range(m - 1):                      # exact forward substitution with I + L
                s[j + 1:] -= Lm[j + 1:m, j] * s[j]
        c = s.copy()
        c[m - 1] += a0
        coef[:m] = torch.from_numpy(c)
        res[0] = float(a0)
        res[1] = float(s[m - 1])

    def norm_scalars(self, nrm2, sc, res2):
        n2 = float(nrm2[0])
        sc[0] = 1.0 / np.sqrt(n2)
        sc[1] = np.sqrt(n2)
        res2[0] = n2

    def dot(self, b, cx, cy, out):
        out[0] = float(b.cols[cx] @ b.cols[cy])

    def nrm2(self, b, cx, out3):
        out3[0] = float(b.cols[cx] @ b.cols[cx])
