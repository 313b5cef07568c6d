"""pytest configuration: the `gpu` marker and import paths.

`-m "not gpu"` tests run on a CPU-only machine (oracle vs golden vectors, host logic, that
the C-ABI library loads and exports its symbols); `-m gpu` tests are the parity tests and
call the HIP path through the C ABI.
"""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "single-view-3d-reconstruction_amd")
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")
