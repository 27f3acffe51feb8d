"""krylovkit_hip -- host-side mirror of KrylovKit.jl's operator/vector/factorization interface
for the Krylov `expand!` hot path, over the C ABI of libkrylov_hip.so (MI355X / gfx950).

Import fails loudly if the shared library has not been built; creating a Context fails loudly
if no GPU is visible.  There is no CPU fallback in this package.
"""
from . import _lib
from ._lib import DimensionMismatch, KrylovHipError, NoDeviceError
from .core import (ClassicalGramSchmidt, ClassicalGramSchmidt2, ClassicalGramSchmidtIR, Context, DeviceBasis, HipVec,
                   KrylovDefaults, ModifiedGramSchmidt, ModifiedGramSchmidt2, ModifiedGramSchmidtIR, Orthogonalizer,
                   SparseOperator, FunctionOperator, default_context, device_count)
from .factorizations import (ArnoldiFactorization, ArnoldiIterator, Block, BlockLanczosFactorization, BlockLanczosIterator,
                             GKLFactorization, GKLIterator, block_inner, block_qr_, block_reorthogonalize_,
                             LanczosFactorization, LanczosIterator, expand_, initialize, initialize_, shrink_)
from . import dist
from .algorithms import CG, GKL, GMRES, LSMR, Arnoldi, BiCGStab, BlockLanczos, ConvergenceInfo, Lanczos
from .eigsolve import eigsolve, eigsolve_block, svdsolve
from .linsolve import linsolve, linsolve_bicgstab, linsolve_cg
from .lssolve import lssolve

from .matrixfun import expintegrator, exponentiate
from .scope import BiArnoldi, GolubYe, OutOfScopeError, bieigsolve, eigsolve_arnoldi, geneigsolve, schursolve   # import-compatible stubs (out of scope, see scope.py)

_lib.load()  # fail at import time if libkrylov_hip.so is missing

__all__ = [n for n in dir() if not n.startswith("_")]
