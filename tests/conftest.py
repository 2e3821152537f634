import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle_bin():
    """Builds (if needed) and returns the path of the CPU oracle binary."""
    path = os.path.join(ROOT, "oracle", "_build", "oracle_readtape")
    subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "oracle"], check=True)
    assert os.path.exists(path)
    return path


@pytest.fixture(scope="session")
def ref_bin():
    """The unmodified reference built by oracle/Makefile (only where /root/reference is mounted)."""
    path = os.path.join(ROOT, "oracle", "_ref", "readtape_evt")
    if not os.path.exists(path):
        if os.path.isdir("/root/reference/src"):
            subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "ref"], check=True)
        else:
            pytest.skip("reference build not available here (oracle/_ref missing)")
    return path
