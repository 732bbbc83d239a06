import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


def _has_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return os.path.exists('/dev/kfd')


@pytest.fixture(scope='session')
def has_gpu():
    return _has_gpu()


@pytest.fixture()
def ctx():
    """A device context; GPU tests fail loudly (not skip) when the HIP library is missing."""
    if not _has_gpu():
        pytest.skip('no GPU visible')
    from opendrift_amd.device import Context
    c = Context(device=0, seed=0)
    yield c
    c.close()


def golden(name):
    return np.load(os.path.join(GOLDEN, name))
