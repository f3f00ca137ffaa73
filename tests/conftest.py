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


@pytest.fixture
def library_defaults():
    """Tests that assert WHICH mechanism served a call (the statistics of the one-call entries, the proofs, the caches, the
    exact kernel a result came from) describe the library's DEFAULTS.  Under `SGN_OPTIONS=...` (the whole suite is also run
    under non-default configurations, profiles/scripts/r06v.sh) they put every option back to its default for their
    duration; what they compare — results against the oracle — is unaffected."""
    from sgn_rast import config
    # (the kernel-selection options — thresholds, reduction form, exact exp — select kernels, not mechanisms, and several
    # test modules set them per test: they are left alone)
    with config.override(**{name: row[0] for name, row in config.OPTIONS.items()
                            if not any(mod == "opts" for mod, _attr in row[2])}):
        yield
