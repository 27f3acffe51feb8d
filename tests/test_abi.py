"""CPU-side checks of the drop-in boundary: the C-ABI library loads, exports every symbol
include/krylov_hip.h declares, the ctypes table matches the header, and WITHOUT a GPU the
product fails loudly instead of falling back to a CPU path."""
import ctypes
import re
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parent.parent
HEADER = ROOT / "include" / "krylov_hip.h"


def declared_symbols():
    txt = re.sub(r"/\*.*?\*/", "", HEADER.read_text(), flags=re.S)
    return sorted(set(re.findall(r"\b(kk_[a-z0-9_]+)\s*\(", txt)))


def test_header_symbols_exported(kk):
    lib = kk._lib.load()
    syms = declared_symbols()
    assert len(syms) >= 55
    for s in syms:
        assert hasattr(lib, s), f"{s} declared in include/krylov_hip.h but not exported"
    assert set(syms) == set(kk._lib.SIGNATURES), set(syms) ^ set(kk._lib.SIGNATURES)


def test_version_and_error_string(kk):
    lib = kk._lib.load()
    assert lib.kk_version() == 302
    assert isinstance(lib.kk_last_error(), bytes)


def test_no_silent_cpu_fallback(kk):
    """No GPU visible -> kk_ctx_create returns KK_ERR_NO_DEVICE and the mirror raises."""
    if kk.device_count() > 0:
        pytest.skip("GPU present")
    h = ctypes.c_void_p()
    rc = kk._lib.load().kk_ctx_create(0, ctypes.byref(h))
    assert rc == kk._lib.KK_ERR_NO_DEVICE and not h.value
    assert b"no CPU fallback" in kk._lib.load().kk_last_error()
    with pytest.raises(kk.NoDeviceError):
        kk.Context(0)
    with pytest.raises(kk.NoDeviceError):
        import scipy.sparse as sp
        kk.SparseOperator(sp.identity(4, format="csr"))


def test_product_does_not_import_oracle():
    """The shipped package must never reach into oracle/ (tier rule 3)."""
    pkg = ROOT / "krylovkit.jl_amd"
    for p in list(pkg.rglob("*.py")) + list(pkg.rglob("*.hip")) + list(pkg.rglob("*.h")):
        txt = p.read_text()
        assert "krylov_oracle" not in txt and "cpu_ref" not in txt, p


def test_orthogonalizer_codes_match_header(kk):
    txt = HEADER.read_text()
    for name, code in (("KK_CGS", 0), ("KK_MGS", 1), ("KK_CGS2", 2), ("KK_MGS2", 3), ("KK_CGSIR", 4), ("KK_MGSIR", 5)):
        assert re.search(rf"{name}\s*=\s*{code}\b", txt)
    assert kk.ModifiedGramSchmidt2().code == 3 and kk.ClassicalGramSchmidtIR().code == 4
    assert kk.KrylovDefaults.orth.name == "mgs2" and kk.KrylovDefaults.krylovdim == 30  # algorithms.jl:556-559


def test_host_dense_helpers(kk):
    from krylovkit_hip import dense
    rng = np.random.default_rng(1)
    x = rng.standard_normal(6)
    beta, v, nu = dense.householder(x, 2)
    e = np.zeros(6); e[2] = nu
    np.testing.assert_allclose(x - beta * v * (v @ x), e, atol=1e-13)
    c, s, r = dense.givens(-2.0, 1.0)
    np.testing.assert_allclose([c * -2 + s * 1, 2 * s + c * 1], [r, 0], atol=1e-14)
    D, U = dense.tridiageigh(np.array([2.0, 2.0, 2.0]), np.array([-1.0, -1.0]))
    np.testing.assert_allclose(D, 2 - 2 * np.cos(np.arange(1, 4) * np.pi / 4), atol=1e-13)
    assert list(dense.sortperm(np.array([-3.0, 1.0, 2.0]), "LM")) == [0, 2, 1]
    assert list(dense.sortperm(np.array([-3.0, 1.0, 2.0]), "SR")) == [0, 1, 2]
    from krylovkit_hip.factorizations import packed_index
    assert [packed_index(i, j) for j in (1, 2, 3) for i in range(1, min(j + 1, 3) + 1)] == list(range(8))


def test_out_of_scope_drivers_stay_importable_and_say_so():
    """eigsolve(alg=Arnoldi), schursolve, bieigsolve, geneigsolve, BiArnoldi, GolubYe are outside SURVEY section 2's
    scope; the names stay importable from the package (round-2 callers), a call raises OutOfScopeError (a NotImplementedError)
    that names tests/hostmirror_extras.py -- and README / INTEGRATION state the reduction"""
    import krylovkit_hip as kk
    for call in (lambda: kk.schursolve(None, None), lambda: kk.bieigsolve(None, None, None), lambda: kk.geneigsolve(None, None),
                 lambda: kk.eigsolve_arnoldi(None, None), lambda: kk.BiArnoldi(), lambda: kk.GolubYe(),
                 lambda: kk.eigsolve(None, None, 1, "LM", kk.Arnoldi())):
        with pytest.raises(kk.OutOfScopeError) as e:
            call()
        assert isinstance(e.value, NotImplementedError) and "hostmirror_extras" in str(e.value)
    root = Path(__file__).resolve().parent.parent
    assert "## Scope" in (root / "README.md").read_text() and "does not cover" in (root / "INTEGRATION.md").read_text()
