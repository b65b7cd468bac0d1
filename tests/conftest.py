import importlib
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def pm():
    """the product package (directory name is not an identifier)"""
    return importlib.import_module("pbrt-rs_amd")


@pytest.fixture(scope="session")
def ora():
    """the CPU oracle (test infrastructure)"""
    import oracle as o
    o.lib()
    return o
