import os
import sys

import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_sessionstart(session):
    # A process that uses both torch's HIP runtime and the library's must initialise torch's first (bench.py does): the other
    # order leaves torch with "No HIP GPUs are available".  A few GPU tests use torch (hipFFT autocorrelations) after others
    # have created batches, so the session touches torch.cuda once before any test runs.
    try:
        import torch
        torch.cuda.is_available()
    except Exception:
        pass


@pytest.fixture(scope="session")
def oracle():
    import _oracle
    _oracle.build()
    return _oracle
