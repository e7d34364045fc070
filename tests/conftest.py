"""pytest configuration: the `gpu` marker and import paths.

-m "not gpu": oracle vs fixtures, host logic, C-ABI symbols (runs on CPU, no compute calls).
-m gpu      : parity of the HIP path against the oracle, through the C ABI (needs an MI355X).
"""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "pantr-cdc2023-experiments_amd"))
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN
