"""Import-compatible names for the reference drivers this package deliberately does NOT ship.

SURVEY.md section 2 marks `eigsolve(alg::Arnoldi)` / `schursolve` (src/eigsolve/arnoldi.jl), `bieigsolve`
(src/eigsolve/biarnoldi.jl), `geneigsolve` (src/eigsolve/golubye.jl) and their algorithm types `BiArnoldi` / `GolubYe`
OUT OF SCOPE: they are host-side dense Schur / restart logic above the `expand!` hot path this library accelerates (the
Arnoldi FACTORIZATION itself -- initialize / expand! / shrink! -- is in scope and is what linsolve(GMRES) and exponentiate
use).  Round-2 code that imported these names from `krylovkit_hip` keeps importing; calling one raises `OutOfScopeError`
with the place where a transliteration lives as TEST INFRASTRUCTURE (`tests/hostmirror_extras.py`: it drives this
package's device factorizations and is compared with the oracle in tests/test_gpu_parity.py), not as product code.
The reduction is stated in README.md ("Scope") and INTEGRATION.md (section "What the drop-in does not cover")."""
class OutOfScopeError(NotImplementedError):
    """a reference driver that SURVEY.md section 2 places outside this package"""


def _stub(name: str, ref: str):
    def fn(*args, **kwargs):
        raise OutOfScopeError(f"{name} ({ref}) is outside the scope of krylovkit_hip (SURVEY.md section 2): only the Krylov expand! hot path "
                              f"and the drivers of BASELINE.json's configs are shipped.  A host-side transliteration over this package's device "
                              f"factorizations lives with the tests: tests/hostmirror_extras.py::{name}")
    fn.__name__ = name
    fn.__doc__ = f"out of scope ({ref}); see krylovkit_hip.scope"
    return fn


schursolve = _stub("schursolve", "src/eigsolve/arnoldi.jl:1-120")
eigsolve_arnoldi = _stub("eigsolve_arnoldi", "src/eigsolve/arnoldi.jl:122-215")
bieigsolve = _stub("bieigsolve", "src/eigsolve/biarnoldi.jl")
geneigsolve = _stub("geneigsolve", "src/eigsolve/golubye.jl")


class BiArnoldi:  # algorithms.jl:274-291
    def __init__(self, *a, **k):
        raise OutOfScopeError("BiArnoldi (src/algorithms.jl:274-291) is outside the scope of krylovkit_hip; see tests/hostmirror_extras.py")


class GolubYe:  # algorithms.jl:310-325
    def __init__(self, *a, **k):
        raise OutOfScopeError("GolubYe (src/algorithms.jl:310-325) is outside the scope of krylovkit_hip; see tests/hostmirror_extras.py")
