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
    from oracle.oracle import Oracle
    return Oracle()


@pytest.fixture(scope="session")
def reference():
    """The reference's own kernel objects (oracle/_ref), when built and runnable here."""
    from oracle.oracle import Reference
    if not Reference.available():
        pytest.skip("oracle/_ref not built (needs /root/reference at build time)")
    try:
        return Reference()
    except RuntimeError as e:  # no AVX on this host
        pytest.skip(str(e))


@pytest.fixture(scope="session")
def golden_cases():
    from tests.golden_io import load_testdata
    return load_testdata()
