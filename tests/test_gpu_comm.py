"""GPU: the communication entry points of the C ABI (include/odrift.h "communication", csrc/odr_comm.hip) over a ONE-rank RCCL
communicator -- all a box with one GPU can run (RCCL wants one device per rank): init through the id exchange of
opendrift_amd/distributed.py, the scalar all-reduce, the step's all-gather in two halves with the row taken from the status
scan ON THE DEVICE, the byte broadcast, and a reader level through odr_block_broadcast == the same level through
odr_block_upload bit for bit.  Each scenario runs in a process of its own (one communicator per process) WITHOUT torch."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import json, os, sys
sys.path.insert(0, %(root)r)
import numpy as np
from opendrift_amd import distributed as D, synthetic as synth
from opendrift_amd.device import Context
out = {}
ctx = Context(0, seed=0)
D.init_rccl(ctx, world1=True)
info = D.comm_info()
out['info'] = info
out['allreduce'] = [D.allreduce_scalars([3.0, -2.5, 7.0], op).tolist() for op in ('sum', 'min', 'max')]
rows = D.allgather_vector(np.arange(25.0))
out['allgather'] = rows.tolist()
out['object'] = D.broadcast_object({'a': [1, 2, 3], 'b': 'level', 'x': np.arange(5.0)})['x'].tolist()
D.barrier()
# a reader level by odr_block_broadcast against the same level by odr_block_upload: same samples bit for bit
U, V, W, KZ, DEP, LAND = ('x_sea_water_velocity', 'y_sea_water_velocity', 'upward_sea_water_velocity',
                          'ocean_vertical_diffusivity', 'sea_floor_depth_below_sea_level', 'land_binary_mask')
g = synth.grid3d(nx=96, ny=80, nz=8, nt=3, seed=2)
names = [U, V, W, KZ, DEP, LAND]
res = []
rng = np.random.default_rng(0)
n = 20000
lon = rng.uniform(g['x'][2], g['x'][-3], n); lat = rng.uniform(g['y'][2], g['y'][-3], n); z = -rng.uniform(0, 60, n)
for how in ('upload', 'broadcast'):
    sid = ctx.add_grid(g['x'], g['y'], z=g['z'])
    for k in range(3):
        arrays = {nm: g[nm][k] for nm in names}
        if how == 'upload':
            ctx.upload_block(sid, k, float(g['t'][k]), arrays)
        else:
            ctx.block_broadcast(sid, k, float(g['t'][k]), arrays, {nm: arrays[nm].shape for nm in names}, root=0)
            ctx.commit_block(sid, k)
    for nm in names:
        ctx.bind(nm, [sid], 0.0)
    ctx.bind('sea_surface_height', [], 0.0)
    P = ctx.particles(n)
    P.append(lon, lat, z=z)
    env = P.env_sample([U, V, W, DEP, LAND], float(g['t'][0]) + 700.0, download=True)
    # the step's collective with its row taken from the status scan on the device
    P.env_coast_advect([U, V, W, DEP, 'sea_surface_height', LAND], float(g['t'][0]) + 700.0, 'runge-kutta4', 600.0, coastline='previous',
                       store_previous=True, count=False, seafloor=True, age_dt=600.0)
    if how == 'broadcast':
        assert P.scan_status_begin()
        h = D.start_allgather_vector(np.full(25, -1.0), from_scan_ctx=ctx)
        kept, flags = P.scan_status_end()
        rows = D.finish_allgather_vector(h)
        out['scan_row'] = rows[0].tolist()
        out['scan_kept'] = int(kept)
    got = P.download()
    res.append((env, got))
    P.close()
(e0, g0), (e1, g1) = res
out['block_equal'] = bool(all(np.array_equal(e0[k].view(np.uint32), e1[k].view(np.uint32)) for k in e0) and
                          np.array_equal(g0['lon'], g1['lon']) and np.array_equal(g0['lat'], g1['lat']))
out['collectives'] = D.comm_info()['collectives']
D.shutdown()
out['after_shutdown'] = D.comm_info()['nranks_seen']
ctx.close()
out['torch_in_process'] = 'torch' in sys.modules
print('RESULT ' + json.dumps(out))
'''


def _run(code, env=None):
    e = dict(os.environ, **(env or {}))
    for k in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK'):
        e.pop(k, None)
    p = subprocess.run([sys.executable, '-c', code], capture_output=True, text=True, timeout=600, env=e, cwd=ROOT)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]
    line = [l for l in p.stdout.splitlines() if l.startswith('RESULT ')][-1]
    return json.loads(line[7:])


def test_one_rank_rccl_communicator_through_the_c_abi():
    r = _run(WORKER % dict(root=ROOT))
    assert r['info']['nranks_seen'] == 1 and r['info']['rank'] == 0 and r['info']['backend'].startswith('rccl')
    assert r['info']['unique_id_hash'] not in (None, '0' * 16) and r['info']['rccl_version'] > 0
    assert r['allreduce'] == [[3.0, -2.5, 7.0]] * 3                              # one rank: identity under sum, min and max
    assert r['allgather'] == [[float(k) for k in range(25)]]
    assert r['object'] == [0.0, 1.0, 2.0, 3.0, 4.0]
    assert r['block_equal']                                                      # odr_block_broadcast == odr_block_upload
    # the row of the step's collective: [0] = elements that stay and [1..8] = flags came from the scan on the device (the host
    # handed -1 over), the rest is the host's
    assert r['scan_row'][0] == r['scan_kept'] > 0 and r['scan_row'][1:9] == [0.0] * 8 and r['scan_row'][9:] == [-1.0] * 16
    assert r['collectives'] >= 10 and r['after_shutdown'] == 1
    assert r['torch_in_process'] is False


def test_bench_line_of_one_gpu_runs_without_torch_and_reports_the_comm_layer():
    code = ("import sys, runpy; sys.argv = ['bench.py', '--workload', 'c3', '--small', '--steps', '4', '--warmup', '2', '--no-cpu', "
            "'--no-extras']; runpy.run_path(%r, run_name='__main__')" % os.path.join(ROOT, 'bench.py'))
    e = dict(os.environ)
    for k in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK'):
        e.pop(k, None)
    p = subprocess.run([sys.executable, '-c', code], capture_output=True, text=True, timeout=900, env=e, cwd=ROOT)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]
    line = json.loads([l for l in p.stdout.splitlines() if l.startswith('{')][-1])
    assert line['torch_in_process'] is False
    assert line['comm']['nranks_seen'] == 1 and line['n_gpus'] == 1
