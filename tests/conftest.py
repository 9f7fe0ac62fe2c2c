import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `-m gpu` on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    """CPU oracle (test infrastructure): built on demand with gcc."""
    from oracle import antq_oracle as orc
    orc.build()
    orc.lib()
    return orc


@pytest.fixture(scope="session")
def antq_lib():
    """The product's C-ABI library; built by __graft_entry__.build() / make -C csrc."""
    from ant_quantization_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        import __graft_entry__
        __graft_entry__.build()
    return _lib


def golden(name):
    import numpy as np
    return np.load(os.path.join(GOLDEN, name), allow_pickle=False)
