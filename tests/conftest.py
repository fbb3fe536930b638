"""pytest configuration: registers the `gpu` marker and puts the product package / oracle on sys.path."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "semantic-slam-master_amd"), os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


import pytest


@pytest.fixture
def knob():
    """Set a TEST-ONLY load-time knob of libsslam_hip.so for one test (sslam_test_set_knob; the library reads the
    environment once at load, never per call) and restore its load-time value afterwards (what the environment said when the
    library was loaded, else the built-in default)."""
    from sslam_amd import lib
    touched = []

    def _set(name, value):
        lib._check(lib.lib().sslam_test_set_knob(name.encode(), int(value), 0), f"set_knob({name})")
        touched.append(name)

    yield _set
    for name in touched:
        lib.lib().sslam_test_set_knob(name.encode(), 0, 1)
