import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    """The CPU oracle (test infrastructure), built on demand with gcc."""
    from tests import oracle_binding
    return oracle_binding.load()


@pytest.fixture(scope="session")
def grlx():
    """The product library; built on demand (hipcc cross-compiles without a GPU)."""
    import grl_amd
    from grl_amd import _build
    _build.build()
    grl_amd.capi.load()
    return grl_amd


def pytest_collection_modifyitems(config, items):
    """A plain `pytest` on a box without an MI355X skips the gpu-marked tests instead of failing them."""
    if not any("gpu" in it.keywords for it in items) or gpu_available():
        return
    skip = pytest.mark.skip(reason="no MI355X (HIP device) on this box")
    for it in items:
        if "gpu" in it.keywords:
            it.add_marker(skip)


def gpu_available() -> bool:
    try:
        from grl_amd import _build, capi
        _build.build()
        return capi.load().grlx_device_count() > 0
    except Exception:
        return False
