import os
import subprocess
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # the C oracle is test infrastructure: build it if the prebuilt library is not there
    lib = os.path.join(REPO, "oracle", "liboracle_knn.so")
    if not os.path.exists(lib):
        subprocess.check_call(["make", "-C", os.path.join(REPO, "oracle")])


@pytest.fixture(scope="session")
def golden():
    from helpers import load_golden
    return load_golden()
