import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for _p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")):
    if _p not in sys.path:
        sys.path.insert(0, _p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


@pytest.fixture(scope="session")
def oracle():
    """The CPU oracle (test infrastructure).  Builds liboracle.so on demand."""
    import subprocess
    so = os.path.join(ROOT, "oracle", "liboracle.so")
    if not os.path.exists(so):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "oracle"])
    from oracle import pyoracle
    return pyoracle


@pytest.fixture(scope="session")
def hip():
    """The product library; GPU tests must run the HIP path, never a fallback."""
    import stereo_amd
    if stereo_amd.device_count() < 1:
        pytest.fail("no HIP device visible: -m gpu tests need a GPU")
    return stereo_amd
