import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "krylovkit.jl_amd"))
sys.path.insert(0, str(ROOT / "oracle"))
sys.path.insert(0, str(ROOT))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _have_gpu():
    try:
        import krylovkit_hip as kk

        return kk.device_count() > 0
    except Exception:
        return False


@pytest.fixture(scope="session")
def kk():
    import krylovkit_hip

    return krylovkit_hip


@pytest.fixture(scope="session")
def ko():
    import krylov_oracle

    return krylov_oracle


@pytest.fixture(scope="session")
def ctx(kk):
    if not _have_gpu():
        pytest.fail("GPU test selected but no HIP device is visible (the product has no CPU fallback)")
    return kk.default_context()
