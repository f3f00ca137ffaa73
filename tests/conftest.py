import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "street-gaussians-ns_amd")
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu under gpurun)")


@pytest.fixture(scope="session")
def c_oracle():
    from oracle import c_oracle as CO
    CO.build()
    return CO


@pytest.fixture(scope="session")
def torch_oracle():
    from oracle import torch_oracle as TO
    return TO
