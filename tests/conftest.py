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


def _probe_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return os.path.exists('/dev/kfd')


# Probed ONCE, when the session starts: torch brings its own HIP runtime, and asked for the first time AFTER
# libodrift_hip.so (system ROCm) has opened the device in this process it reports "no GPU" -- tests that ran behind a test
# creating its own Context were then skipped as if the box had none (seen with `pytest tests/test_gpu_movers.py
# tests/test_gpu_parity.py`).
_HAS_GPU = _probe_gpu()


def _has_gpu():
    return _HAS_GPU


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


@pytest.fixture(autouse=True)
def _loop_body_goldens_have_no_run_preamble():
    """The golden vectors were written by the reference's LOOP BODY (oracle/refdriver.py restates run()'s orchestration
    without its preamble), i.e. without seed:ocean_only moving elements seeded on land (reference default True): the models
    of the tests default to False; tests of the option set it themselves."""
    try:
        from opendrift_amd.oceandrift import OpenDriftSimulation
    except Exception:
        yield
        return
    old = OpenDriftSimulation.SEED_OCEAN_ONLY_DEFAULT
    OpenDriftSimulation.SEED_OCEAN_ONLY_DEFAULT = False
    yield
    OpenDriftSimulation.SEED_OCEAN_ONLY_DEFAULT = old


def golden(name):
    return np.load(os.path.join(GOLDEN, name))
