import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")
    # a hung test must not eat the GPU box's time limit: default per-test timeout when pytest-timeout is there
    if config.pluginmanager.hasplugin("timeout") and not getattr(config.option, "timeout", None):
        config.option.timeout = 600
    # KK_TEST_LIB=<flavour> runs the suite against a tools build of the library (libkokoro_hip_<flavour>.so: a --variant build under test
    # before it becomes the product default).  Test infrastructure only — the product reads no such variable.
    flavour = os.environ.get("KK_TEST_LIB")
    if flavour:
        from kokoro_ruslan_amd import lib
        lib.use_library(flavour)


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN

