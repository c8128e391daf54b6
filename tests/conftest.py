import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_sessionstart(session):
    """Built artefacts are git-ignored: a checkout that never ran build() gets one here (the product itself still
    refuses to run without its HIP library -- this only spares a test session the manual step)."""
    pkg = os.path.join(ROOT, "parallel-data-compression-and-decompression_amd")
    needed = [os.path.join(pkg, "libzwz_hip.so"), os.path.join(pkg, "main"), os.path.join(ROOT, "oracle", "libzwz_oracle.so")]
    if all(os.path.exists(p) for p in needed):
        return
    import __graft_entry__
    __graft_entry__.build()


@pytest.fixture(scope="session")
def oracle():
    """ctypes handle on the CPU oracle (oracle/libzwz_oracle.so), built on demand."""
    import oracle_binding
    return oracle_binding.load()


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")
