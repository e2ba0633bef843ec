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


# ---- the GPU suite's order and its time guard -------------------------------------------------------------------------------
# The round-end driver kills the `-m gpu` pytest step at 900 s (GPUTEST_r03.json: steps[0].timeout_s).  The suite takes ~600 s on a
# typical box; boxes differ by +-15 %.  So (1) GPU tests run in order of what they pin: full-size parity of the configurations
# BASELINE.json names first, kernel parity next, model-level parity, reverse passes, and the subprocess-heavy CLI / driver tests
# last; (2) once the session has run IEF_GPU_SUITE_BUDGET seconds (default 780) every remaining GPU test SKIPS itself with
# that reason: a pathologically slow box ends with skips at the least critical end, not with a kill that loses the record.
_GPU_ORDER = ["test_gpu_zz_fullsize", "test_gpu_x3p", "test_gpu_x3", "test_gpu_ops", "test_gpu_exact", "test_gpu_unet", "test_gpu_vae",
              "test_gpu_grad_f32", "test_gpu_grad", "test_gpu_pnp", "test_gpu_p2pzero", "test_gpu_sdxl", "test_gpu_cli"]


def pytest_collection_modifyitems(config, items):
    def rank(item):
        name = os.path.splitext(os.path.basename(str(item.fspath)))[0]
        return _GPU_ORDER.index(name) if name in _GPU_ORDER else -1          # CPU test files keep their place in front
    items.sort(key=rank)                                                     # stable: the order inside a file is kept


def pytest_runtest_setup(item):
    if item.get_closest_marker("gpu") is None:
        return
    budget = float(os.environ.get("IEF_GPU_SUITE_BUDGET", "780"))
    if suite_seconds() > budget:
        pytest.skip(f"GPU suite time budget ({budget:.0f} s of the driver's 900 s) used up: skipped, not killed")


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")
