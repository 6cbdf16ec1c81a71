import os
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "tests"))
sys.path.insert(0, os.path.join(REPO, "hypersonic-rle-kit_amd", "python"))
os.environ.setdefault("HSRLE_POISON_WORKSPACE", "1")   # workspaces the Python layer allocates for the tests start as garbage, not as fresh (zero) pages


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    from hsrle_testlib import Oracle

    return Oracle()


@pytest.fixture(scope="session")
def reference():
    from hsrle_testlib import Reference

    if not Reference.available():
        pytest.skip("oracle/_ref/libhsrle_ref.so not built (needs /root/reference)")
    return Reference()
