import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN_DIR = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def golden_names():
    if not os.path.isdir(GOLDEN_DIR):
        return []
    # (two_dots and coarse_* have their own tests: they carry parameters of a field, not a small sample array)
    return sorted(f[:-4] for f in os.listdir(GOLDEN_DIR) if f.endswith(".npz") and f != "two_dots.npz" and not f.startswith("coarse_")
                  and not f.startswith("crossing_"))


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN_DIR
