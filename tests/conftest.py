import os
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_sessionfinish(session, exitstatus):
    from tests import parity_log
    parity_log.flush()


def load_golden(name):
    return torch.load(os.path.join(GOLDEN, name + ".pt"), weights_only=False)


@pytest.fixture(scope="session", autouse=True)
def _build_oracle():
    from oracle import wkv6_oracle
    wkv6_oracle.build()


@pytest.fixture(scope="session")
def hip():
    """The C-ABI library on a GPU box; GPU tests must never silently fall back."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from paper_accurate_fast_cheap_amd import _lib
    return _lib.lib()
