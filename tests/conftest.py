import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


def _gpu_available() -> bool:
    try:
        from romcomma_amd import _lib
        return _lib.device_count() > 0
    except Exception:
        return False


@pytest.fixture(scope='session')
def gpu():
    """Fails (does not skip) when -m gpu is requested on a box without a usable GPU/library: no silent fallback."""
    from romcomma_amd import _lib
    n = _lib.device_count()
    assert n > 0, 'no HIP device visible: GPU tests need the MI355X box'
    return _lib
