import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)


@pytest.fixture(scope="session")
def oracle():
    from oracle.oracle import Oracle

    return Oracle()


@pytest.fixture(scope="session")
def ref_oracle():
    """The reference's own compiled CPU kernels (oracle/_ref), when present."""
    from oracle.oracle import load_ref

    r = load_ref()
    if r is None:
        pytest.skip("oracle/_ref not built (needs /root/reference; see oracle/build_ref.py)")
    return r


@pytest.fixture(scope="session")
def dev():
    import torch

    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")


def bits(a):
    """fp32 array -> int32 view for bit-exact comparisons."""
    return np.ascontiguousarray(a, dtype=np.float32).view(np.int32)
