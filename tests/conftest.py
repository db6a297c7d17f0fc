import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, 'tests', 'golden')

try:  # the CPU oracle: use the box's CPU share, not the host's core count (oversubscription crawls)
    import torch
    torch.set_num_threads(max(1, min(len(os.sched_getaffinity(0)), 16)))
except Exception:  # noqa: BLE001
    pass


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


@pytest.fixture(scope='session')
def golden_dir():
    return GOLDEN
