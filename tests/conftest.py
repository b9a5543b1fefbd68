import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import __graft_entry__ as graft  # noqa: E402


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def rwr():
    """The product package (ctypes over librwr_hip.so); built on demand."""
    mod = graft.load_package()
    if not os.path.exists(mod.LIB_PATH):
        mod.build()
    return mod


@pytest.fixture(scope="session")
def orc():
    """The CPU oracle (test infrastructure)."""
    from oracle import oracle

    oracle.build()
    return oracle


@pytest.fixture(scope="session")
def ref_loader():
    from oracle import ref_loader as rl

    return rl


@pytest.fixture(scope="session")
def res_dir(rwr):
    return rwr.RES_DIR


@pytest.fixture(scope="session")
def suzanne(ref_loader, res_dir):
    return ref_loader.load_model_compute(res_dir, "suzanne_lowpoly.obj")


@pytest.fixture(scope="session")
def cube(ref_loader, res_dir):
    return ref_loader.load_model_compute(res_dir, "cube.obj")


@pytest.fixture(scope="session")
def gpu_ctx(rwr):
    """One context shared by the GPU tests (one process, one card)."""
    ctx = rwr.Context(0)
    yield ctx
    ctx.close()
