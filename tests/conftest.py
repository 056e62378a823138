import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _gpu_available():
    try:
        import ohm_amd
        return ohm_amd.device_count() > 0
    except Exception:
        return False


@pytest.fixture(scope="session")
def gpu():
    """GPU tests must never pass silently without the HIP library: fail (not skip) if it cannot be used."""
    import ohm_amd  # ImportError here == libohmhip.so missing: loud failure by design
    assert ohm_amd.device_count() > 0, "no HIP device visible: -m gpu tests need an MI355X"
    return ohm_amd.device_info(0)
