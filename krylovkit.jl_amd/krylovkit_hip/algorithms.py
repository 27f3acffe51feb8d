"""Algorithm structs (src/algorithms.jl) and ConvergenceInfo (src/KrylovKit.jl:212-218): the parameter records the
reference dispatches its solvers on."""
from __future__ import annotations

from dataclasses import dataclass, field

from .core import KrylovDefaults, ModifiedGramSchmidt, Orthogonalizer


@dataclass
class ConvergenceInfo:  # KrylovKit.jl:212-218
    converged: int
    residual: object
    normres: object
    numiter: int
    numops: int


# -------------------------------------------------------------------- algorithm structs
@dataclass
class Lanczos:  # algorithms.jl:110-127
    orth: Orthogonalizer = KrylovDefaults.orth
    krylovdim: int = KrylovDefaults.krylovdim
    maxiter: int = KrylovDefaults.maxiter
    tol: float = KrylovDefaults.tol
    eager: bool = False
    verbosity: int = 0


@dataclass
class Arnoldi:  # algorithms.jl:235-252
    orth: Orthogonalizer = KrylovDefaults.orth
    krylovdim: int = KrylovDefaults.krylovdim
    maxiter: int = KrylovDefaults.maxiter
    tol: float = KrylovDefaults.tol
    eager: bool = False
    verbosity: int = 0


@dataclass
class GMRES:  # algorithms.jl:373-390
    orth: Orthogonalizer = KrylovDefaults.orth
    maxiter: int = KrylovDefaults.maxiter
    krylovdim: int = KrylovDefaults.krylovdim
    tol: float = KrylovDefaults.tol
    verbosity: int = 0


@dataclass
class GKL:  # algorithms.jl:200-217
    orth: Orthogonalizer = KrylovDefaults.orth
    krylovdim: int = KrylovDefaults.krylovdim
    maxiter: int = KrylovDefaults.maxiter
    tol: float = KrylovDefaults.tol
    eager: bool = False
    verbosity: int = 0


@dataclass
class BlockLanczos:  # algorithms.jl:152-171 (blockkrylovdim default 100, algorithms.jl:561)
    orth: Orthogonalizer = KrylovDefaults.orth
    krylovdim: int = 100
    maxiter: int = KrylovDefaults.maxiter
    tol: float = KrylovDefaults.tol
    qr_tol: float = KrylovDefaults.tol
    eager: bool = False
    verbosity: int = 0


@dataclass
class CG:  # algorithms.jl:325-337
    maxiter: int = KrylovDefaults.maxiter
    tol: float = KrylovDefaults.tol
    verbosity: int = 0


@dataclass
class BiCGStab:  # algorithms.jl:469-481
    maxiter: int = KrylovDefaults.maxiter
    tol: float = KrylovDefaults.tol
    verbosity: int = 0


@dataclass
class LSMR:  # algorithms.jl:506-521
    orth: Orthogonalizer = field(default_factory=ModifiedGramSchmidt)
    maxiter: int = KrylovDefaults.maxiter
    krylovdim: int = KrylovDefaults.krylovdim
    tol: float = KrylovDefaults.tol
    verbosity: int = 0
