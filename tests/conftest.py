import os
import sys
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import __graft_entry__ as graft  # noqa: E402


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def rt():
    """the product package (ctypes over librtx_hip.so); never falls back to anything else"""
    return graft.load_package()


@pytest.fixture(scope="session")
def orc():
    """the CPU oracle (test infrastructure)"""
    return graft.load_oracle()


@pytest.fixture(scope="session")
def cornell(rt):
    return rt.Scene.cornell()


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")
