import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def repo_root():
    return ROOT


@pytest.fixture(autouse=True)
def _clean_emulated_status_word():
    """the emulated kernels' health word is sticky like the real one: every test starts from a clean one"""
    from tests import _emul
    _emul._STATUS.zero_()
    yield
