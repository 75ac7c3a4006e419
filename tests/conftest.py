import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (gfx950); run with -m gpu on the GPU box")


def _have_gpu():
    try:
        from mcptam_amd import chain_bundle
        return chain_bundle.device_count() > 0
    except Exception:
        return False


@pytest.fixture(scope="session")
def gpu_required():
    """GPU tests must run the HIP path: fail (not skip) if it cannot be loaded on a GPU box."""
    from mcptam_amd import chain_bundle
    n = chain_bundle.device_count()
    assert n > 0, "no gfx950 device visible: " + chain_bundle.last_error()
    return n
