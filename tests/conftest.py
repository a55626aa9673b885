import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


@pytest.fixture(scope='session', autouse=True)
def _oracle_thread_cap():
    """The CPU oracle is thousands of small torch ops: on a 128-thread GPU box torch's default pool spends its time waking workers
    (bench.py's cpu_baseline measured 0.16 frames/s with 128 threads against 9.7 with 16).  The GPU parity tests spend most of their
    wall time in the oracle, so the whole session runs it on at most 16 threads."""
    import torch
    torch.set_num_threads(min(16, torch.get_num_threads()))
    yield


@pytest.fixture(scope='session')
def hip_lib():
    from video_prediction_amd import lib
    return lib.get()
