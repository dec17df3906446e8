import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def pkg():
    import __graft_entry__ as ge
    return ge.load_package()


@pytest.fixture(scope="session")
def synth(pkg):
    return pkg.synth


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as O
    O.lib()
    return O


@pytest.fixture(scope="session")
def golden():
    import numpy as np
    d = os.path.join(ROOT, "tests", "golden")
    return {f[:-4]: np.load(os.path.join(d, f)) for f in os.listdir(d) if f.endswith(".npz")}


@pytest.fixture(scope="session")
def srukf(pkg):
    """The HIP path through the C-ABI.  Fails loudly (no fallback) if the library is missing."""
    pkg.srukf.load_library()
    return pkg.srukf
