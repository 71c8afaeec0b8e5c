import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def built():
    """Build the product library, the oracle (and oracle/_ref when the reference tree is present)."""
    import __graft_entry__ as ge
    ge.build()
    return True


@pytest.fixture(scope="session")
def oracle(built):
    from tests.support.oracle_binding import Oracle
    return Oracle()


@pytest.fixture(scope="session")
def lib(built):
    from aprilsam_amd import host
    return host.SolverLib()


@pytest.fixture(scope="session")
def reflib(built):
    """The unmodified reference (oracle/_ref), or skip when it was not built / did not travel."""
    from aprilsam_amd import host
    from tests.support.oracle_binding import REFLIB
    if not os.path.exists(REFLIB):
        pytest.skip("oracle/_ref/libaprilsam_ref.so not present")
    return host.SolverLib(REFLIB)


def golden(name):
    import numpy as np
    return np.load(os.path.join(GOLDEN, name))
