import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    """The C restatement (oracle/liboracle.so), built on demand.  Test infrastructure only."""
    from tests import _oracle
    return _oracle.load_oracle()


@pytest.fixture(scope="session")
def ref_kd():
    """oracle/_ref: the reference's nanoflann header compiled in place (strict-IEEE build)."""
    from tests import _oracle
    lib = _oracle.load_ref(strict=True)
    if lib is None:
        pytest.skip("oracle/_ref not built (needs /root/reference at build time)")
    return lib
