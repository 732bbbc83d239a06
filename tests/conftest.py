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


def _gpu_tests_selected(config):
    """True when the -m expression asks for the gpu tests (`-m gpu`, not `-m "not gpu"` and not an unmarked run)."""
    expr = (config.getoption('markexpr', '') or '').strip()
    return 'gpu' in expr.split() and 'not gpu' not in expr


@pytest.fixture()
def ctx(request):
    """A device context.  Under `-m gpu` a box without a visible device FAILS the test (a GPU record must not go green
    with every test skipped); an unmarked run on a CPU box skips.  A missing HIP library always fails (Context raises)."""
    if not _has_gpu():
        if _gpu_tests_selected(request.config):
            pytest.fail('-m gpu was asked for and no GPU is visible on this box')
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


# Two rows in the format of the reference's OBJECTPROP.DAT (three lines per class: key + number, description, nine
# coefficients; leeway.py:186-219).  The table itself is a data file of the reference and is not shipped; the numbers of
# class 1 are the ones the C5 goldens were generated with (oracle/gen_golden.py: object_type=1).
OBJECTPROP_EXCERPT = """ PIW-1                        1
 Person-in-water (PIW), unknown state (mean values)
       0.96     0.00     12.00      0.54      0.00      9.40     -0.54      0.00      9.40
 PIW-2                        2
 >PIW, vertical PFD type III conscious
       0.48     0.00      8.30      0.15      0.00      6.70     -0.15      0.00      6.70

"""


@pytest.fixture()
def objectprop_path(tmp_path):
    p = tmp_path / 'OBJECTPROP.DAT'
    p.write_text(OBJECTPROP_EXCERPT)
    return str(p)
