import os
import subprocess
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(REPO, "tests", "golden")
ORACLE_DIR = os.path.join(REPO, "oracle")
if REPO not in sys.path:
    sys.path.insert(0, REPO)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle_cli():
    """Path of the CPU restatement's CLI (oracle/gphocs_oracle); built on demand (gcc only)."""
    exe = os.path.join(ORACLE_DIR, "gphocs_oracle")
    subprocess.run(["make", "-C", ORACLE_DIR, "oracle"], check=True, capture_output=True, timeout=300)
    assert os.path.exists(exe)
    return exe


@pytest.fixture(scope="session")
def ref_cli():
    """Path of the prebuilt real-reference harness (oracle/_ref/gphocs_ref) or None."""
    exe = os.path.join(ORACLE_DIR, "_ref", "gphocs_ref")
    if os.path.isdir("/root/reference/src"):
        subprocess.run(["make", "-C", ORACLE_DIR, "ref"], check=True, capture_output=True, timeout=600)
    return exe if os.path.exists(exe) else None
