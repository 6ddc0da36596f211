import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parents[1]
for p in (ROOT, ROOT / "oracle", ROOT / "tests" / "hostemu"):
    if str(p) not in sys.path:
        sys.path.insert(0, str(p))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `-m gpu` on the GPU box)")


def _have_ref(cfg, imt, jmt, km):
    import refmodel
    return refmodel.available(cfg, imt, jmt, km)


@pytest.fixture(scope="session")
def have_ref():
    return _have_ref
