import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


# The library nominates from its int8 copy only from 1M rows up (below that the bf16 scan is faster).  The tests' corpora are
# far smaller, and the copy is the newest code: let every index of the GPU tier use it, so that each parity test (filters,
# tombstones, ties, short batches, ...) also pins the int8 path.  The bf16 scans are pinned by the tests that switch the copy
# off (test_fused_scan_equals_the_three_kernel_form_and_the_oracle) and by test_search_fullsize_gpu.py's default-policy case.
os.environ.setdefault("CODERAG_HIP_I8_MIN_ROWS", "0")


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
