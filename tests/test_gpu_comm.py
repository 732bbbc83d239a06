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


READER_WORKER = r'''
import json, os, sys
sys.path.insert(0, %(root)r)
import numpy as np
from datetime import datetime, timedelta
from opendrift_amd import distributed as D, synthetic as synth, readers
from opendrift_amd.oceandrift import OceanDrift
from opendrift_amd.device import Context
U, V, W, KZ, DEP, LAND = ('x_sea_water_velocity', 'y_sea_water_velocity', 'upward_sea_water_velocity',
                          'ocean_vertical_diffusivity', 'sea_floor_depth_below_sea_level', 'land_binary_mask')
g = synth.grid3d(nx=96, ny=80, nz=8, nt=4, seed=5)
T0 = datetime(2020, 1, 1)
times = [T0 + timedelta(seconds=float(t)) for t in g['t']]
rng = np.random.default_rng(1)
n = 20000
lon = rng.uniform(g['x'][4], g['x'][-5], n); lat = rng.uniform(g['y'][4], g['y'][-5], n); z = -rng.uniform(0, 40, n)
steps = int((g['t'][-1] - g['t'][0]) / 600.0) - 1


def run(sharded_levels):
    o = OceanDrift(loglevel=50, seed=3)
    o.add_reader(readers.GridReader(g['x'], g['y'], times, {k: g[k] for k in (U, V, W, KZ, DEP, LAND)}, z=g['z']))
    o.set_config('drift:advection_scheme', 'runge-kutta4')
    o.set_config('drift:vertical_mixing', True)
    o.set_config('vertical_mixing:timestep', 60)
    o.set_config('drift:vertical_advection', True)
    o.set_config('general:coastline_action', 'previous')
    o.seed_elements(lon=lon, lat=lat, z=z, time=T0)
    if sharded_levels:
        # every reader level takes the path of a sharded run over the C-ABI collectives: header by odr_comm_broadcast_bytes, the
        # arrays by odr_block_broadcast (a one-rank communicator: this process is rank 0 of 1)
        init = readers.DeviceReaderBinding.__init__
        def patched(self, *a, **k):
            init(self, *a, **k)
            self.world = 2
        readers.DeviceReaderBinding.__init__ = patched
    try:
        o.run(time_step=600, steps=steps)
    finally:
        if sharded_levels:
            readers.DeviceReaderBinding.__init__ = init
    e = o.elements
    order = np.argsort(e.ID)
    return e.lon[order], e.lat[order], e.z[order], len(e.ID)

a = run(False)
D.init_rccl(Context(0, seed=0), world1=True)
k0 = D.comm_info()['collectives']
b = run(True)
out = dict(equal=bool(all(np.array_equal(x, y) for x, y in zip(a[:3], b[:3])) and a[3] == b[3]), n=a[3],
           collectives=D.comm_info()['collectives'] - k0, steps=steps, torch='torch' in sys.modules)
D.shutdown()
print('RESULT ' + json.dumps(out))
'''


def test_reader_levels_of_a_model_run_through_the_rccl_path_change_nothing():
    """DeviceReaderBinding in a sharded run over the C-ABI collectives (_read_level_rccl -> odr_block_broadcast, levels staged a
    period ahead from page-locked memory and committed when due) against the one-process path, on a one-rank communicator:
    OceanDrift.run() ends bit-identical; at least a header + length + level collective per reader level were made."""
    r = _run(READER_WORKER % dict(root=ROOT))
    assert r['equal'] and r['n'] > 1000, r
    assert r['collectives'] >= 9 and r['torch'] is False, r
