import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _gpu_present() -> bool:
    try:
        import coderag_amd  # noqa: F401
        from coderag_amd import ffi
        return ffi.device_count() > 0
    except Exception:
        return False


@pytest.fixture(scope="session")
def gpu():
    """GPU tests fail loudly (not skip) when selected with -m gpu on a box without a device."""
    if not _gpu_present():
        pytest.fail("this test needs an MI355X and libcoderag_hip.so (run under gpurun)")
    return 0
