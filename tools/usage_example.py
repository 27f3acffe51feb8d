"""The README usage snippet as a runnable script (GPU box): python tools/usage_example.py"""
import sys; sys.path.insert(0, str(__import__("pathlib").Path(__file__).resolve().parent.parent / "krylovkit.jl_amd"))
import numpy as np, scipy.sparse as sp
import krylovkit_hip as kk
n  = 200_000; rng = np.random.default_rng(1)
A  = sp.csr_matrix((rng.standard_normal(10 * n), (rng.integers(0, n, 10 * n), rng.integers(0, n, 10 * n))), shape=(n, n))
A  = ((A + A.T) * 0.5).tocsr()
op = kk.SparseOperator(A, symmetric=True)
x0 = np.random.default_rng(0).random(A.shape[0])
vals, vecs, info = kk.eigsolve(op, x0, 4, "LM", kk.Lanczos(krylovdim=60, tol=1e-8, maxiter=30)); print(vals, info.converged)
x, info = kk.linsolve(op, x0, None, kk.GMRES(krylovdim=40, tol=1e-8), 12.0, 1.0); print(info.converged, np.linalg.norm(12*x + A@x - x0))
x, info = kk.linsolve(op, x0, None, kk.BiCGStab(tol=1e-8), 12.0, 1.0); print(info.converged, np.linalg.norm(12*x + A@x - x0))
w, info = kk.exponentiate(op, -0.5, x0, kk.Lanczos(krylovdim=30, tol=1e-10)); print(info.converged, np.linalg.norm(w))
f  = kk.FunctionOperator(lambda x, y: op.apply(x, y).add_(x, 3.0), A.shape[0], symmetric=True)
vals2, vecs, info = kk.eigsolve(f, x0, 2, "SR", kk.Lanczos(tol=1e-8, maxiter=30)); print(vals2, info.converged)
