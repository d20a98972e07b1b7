import importlib
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def zkp():
    """the product package (directory name has a hyphen, hence importlib)"""
    return importlib.import_module("zk-paillier_amd")


@pytest.fixture(scope="session")
def oracle():
    import oracle_lib
    return oracle_lib.Oracle()


@pytest.fixture(scope="session")
def ctx(zkp):
    """one GPU context for the whole -m gpu session; fails loudly without GPU / built library"""
    c = zkp.Context(0)
    yield c
    c.close()
