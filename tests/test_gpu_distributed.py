"""GPU: a model run sharded over two ranks gives the bits of the one-process run.

Two processes share the one GPU of the test box (ODR_DIST_BACKEND=gloo: RCCL wants one device per rank; the collectives
-- block broadcast from the rank that reads, all-reduce of the movers' global scalars, of OpenOil's means, of the element
counts and of the new deactivation reasons -- are the same calls with either backend).  Every rank runs the same
script, owns a contiguous range of element IDs and draws its random numbers from Philox streams keyed by element ID:
the concatenated final state must EQUAL the one-process run bit for bit, including the status numbers (OpenOil's z:
to 1e-6 m, the order of a float64 sum)."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _run(scenario, world, out):
    port = _free_port()
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK='0', WORLD_SIZE=str(world), MASTER_ADDR='127.0.0.1',
                   MASTER_PORT=str(port), ODR_DIST_BACKEND='gloo', HSA_ENABLE_IPC_MODE_LEGACY='0')
        procs.append(subprocess.Popen([sys.executable, os.path.join(HERE, 'dist_worker.py'), scenario, out], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    outs = [p.communicate(timeout=600)[0].decode() for p in procs]
    assert all(p.returncode == 0 for p in procs), '\n'.join('--- rank %d (rc %s)\n%s' % (r, p.returncode, o[-2500:])
                                                             for r, (p, o) in enumerate(zip(procs, outs)))
    parts = [np.load(out + '.rank%d.npz' % r) for r in range(world)]
    ID = np.concatenate([q['ID'] for q in parts])
    order = np.argsort(ID)
    res = {k: np.concatenate([q[k] for q in parts])[order] for k in ('ID', 'lon', 'lat', 'z', 'status')}
    cats = [list(q['categories']) for q in parts]
    assert all(c == cats[0] for c in cats), cats       # every rank numbers the deactivation reasons alike
    res['_parts'] = parts
    return res, cats[0], [tuple(q['shard']) for q in parts]


@pytest.mark.parametrize('scenario', ['oceandrift', 'openoil', 'ensemble'])
def test_two_ranks_equal_one_rank(tmp_path, scenario):
    one, cats1, _ = _run(scenario, 1, str(tmp_path / 'w1'))
    two, cats2, shards = _run(scenario, 2, str(tmp_path / 'w2'))
    assert shards[0][1] == shards[1][0] and shards[0][0] == 0 and shards[1][1] == len(one['ID'])
    assert cats1 == cats2
    for k in ('ID', 'status'):
        assert np.array_equal(one[k], two[k]), k
    for k in ('lon', 'lat'):
        assert np.array_equal(one[k], two[k]), (k, np.abs(one[k] - two[k]).max())
    if scenario == 'openoil':
        # np.mean(1.5 Hs) and np.mean(dV_50) are float64 sums whose order follows the memory layout (blocks of one
        # process, then ranks): the float32 intrusion depth scale may differ in its last bit -> z to 1e-6 m, same slick
        assert np.abs(one['z'] - two['z']).max() < 2e-6 and np.array_equal(one['z'] == 0, two['z'] == 0)
    else:
        assert np.array_equal(one['z'], two['z']), np.abs(one['z'] - two['z']).max()
    # the sharded run's communication: ONE collective per step (the all-gathered step summary: kept count, new status
    # reasons, the movers' reductions) -- plus OpenOil's two global means and the per-level reader headers; every rank
    # holds only its own ID range of the schedule
    for q in two['_parts']:
        steps, ncoll = int(q['timing'][0]), int(q['timing'][1])
        per_step = 2 if scenario == 'openoil' else 1
        assert ncoll <= per_step * steps + 4, (steps, ncoll)
        assert int(q['n_sched_local']) == int(q['shard'][1] - q['shard'][0]) and int(q['n_total']) == len(one['ID'])
    if scenario == 'oceandrift':
        assert 'outside' in cats1 and (one['status'] != 0).sum() > 10
    if scenario == 'ensemble':
        # (ensemble members: the rank of an element among the present ones of ALL ranks selects the member --
        # odr_particles_set_rank_offset from the step's collective; stranded elements shift the ranks during the run)
        assert 'stranded' in cats1 and (one['status'] != 0).sum() > 10
    else:
        assert (one['z'] < -1).sum() > 100
