"""Backward-compatible aggregate of the solver modules (eigsolve / linsolve / lssolve / algorithms), laid out like
the reference's src/ directories."""
from .algorithms import *  # noqa: F401,F403
from .eigsolve import (_eigsolve_arnoldi, _schursolve, bieigsolve, eigsolve, eigsolve_block, geneigsolve, schursolve,  # noqa: F401
                       svdsolve)
from .linsolve import linsolve, linsolve_bicgstab, linsolve_cg  # noqa: F401
from .lssolve import lssolve  # noqa: F401
