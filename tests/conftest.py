import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture
def hip_option():
    """set(name, value): frirl_hip_set_option for the duration of one test (experiment / variant switches of the HIP
    library: "no_uidx", "lanes_slices", "rollout_group", "rd_persist" ...); restored afterwards."""
    import frirl_amd
    saved = []

    def set_(name, value):
        saved.append((name, frirl_amd.set_option(name, value)))
    yield set_
    for name, old in reversed(saved):
        frirl_amd.set_option(name, old)
