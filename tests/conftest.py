import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")


def pytest_sessionfinish(session, exitstatus):
    """Per-quantity maxima of the strict relative / absolute errors of every assert_close of the session
    (tests/helpers.py), written where a GPU run's outputs are collected."""
    from tests import helpers
    if not helpers.PARITY_ERRORS:
        return
    import json
    out = os.path.join(ROOT, "gpurun_out")
    os.makedirs(out, exist_ok=True)
    with open(os.path.join(out, "parity_errors.json"), "w") as f:
        json.dump({"small": helpers.SMALL, "quantities": dict(sorted(helpers.PARITY_ERRORS.items()))}, f, indent=1)
