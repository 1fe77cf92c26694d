import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    from oracle_binding import Oracle
    return Oracle()


@pytest.fixture(scope="session")
def oracle_omp():
    from oracle_binding import Oracle
    return Oracle(omp=True)


def pytest_collection_finish(session):
    """GPU runs: bring torch's HIP context up BEFORE the library's first HIP call.  On this image torch cannot
    find the device once another HIP user in the process has initialised it first ("no ROCm-capable device"), so
    without this the tests that use torch tensors pass or fail with the order the files are run in."""
    if any(item.get_closest_marker("gpu") for item in session.items):
        import torch
        if torch.cuda.is_available():
            torch.cuda.init()
            torch.zeros(1, device="cuda")
