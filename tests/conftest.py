import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


@pytest.fixture(scope="session")
def params():
    import configparser
    cfg = configparser.ConfigParser()
    cfg.read(os.path.join(ROOT, "config"))
    return dict(cfg["DEFAULT"])


@pytest.fixture(scope="session")
def oracle32(params):
    from oracle.oracle import Oracle
    return Oracle("f32", params)


@pytest.fixture(scope="session")
def oracle64(params):
    from oracle.oracle import Oracle
    return Oracle("f64", params)
