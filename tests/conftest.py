import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    import oracle_lib
    return oracle_lib.load_oracle()


@pytest.fixture(scope="session")
def ref():
    import oracle_lib
    r = oracle_lib.load_ref()
    if r is None:
        pytest.skip("oracle/_ref/libdafs_ref.so not built (needs /root/reference)")
    return r
