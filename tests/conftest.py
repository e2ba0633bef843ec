import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import ief_amd  # noqa: E402,F401  (registers the hyphenated package dir as `ief_amd`)


import time as _time

_SESSION_T0 = _time.time()


def suite_seconds() -> float:
    """seconds since this pytest session started (the full-size oracle comparisons run last and skip themselves when a slow
    box has already used the suite's time budget, so the session always ends instead of being killed at the driver's limit)"""
    return _time.time() - _SESSION_T0


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")
