import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def native_lib():
    """libmirge_amd.so, built in-tree if missing (hipcc cross-compiles without a GPU)."""
    from mirge_amd import _native
    if not os.path.exists(_native.LIB_PATH):
        import subprocess
        subprocess.run(["make", "-C", os.path.join(ROOT, "mirge_amd", "csrc")], check=True)
    return _native.load()


@pytest.fixture(scope="session")
def oracle_lib():
    from oracle import model
    model.build()
    return model.lib()
