import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def orc():
    import oracle
    oracle.build()
    oracle.lib()
    return oracle


@pytest.fixture(scope="session")
def sdvgn_lib():
    """The product library; built in-tree if missing (hipcc cross-compiles without a GPU)."""
    from sdv_loam_amd import api
    if not os.path.exists(api.LIB_PATH):
        import subprocess
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "sdv-loam_amd", "csrc"), "-s"])
    return api.load_library()
