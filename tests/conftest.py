import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")

# Multi-process GPU tests (test_gpu_sharded.py) fork their ranks from a fork server.  It is started
# here, before any test can touch the GPU, so that no process holding a HIP context ever forks or
# execs (on the GPU pool that is forbidden); the ranks are clean children of a clean server.
import multiprocessing as _mp  # noqa: E402
import multiprocessing.forkserver as _forkserver  # noqa: E402

_mp.set_forkserver_preload(["numpy"])
_forkserver.ensure_running()


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    def load(name):
        return np.load(os.path.join(GOLDEN, name))
    return load


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as orc
    orc.load()
    return orc
