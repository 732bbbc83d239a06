"""CPU: `python bench.py --gpus N` started on its own (no RANK / WORLD_SIZE) launches its N ranks itself -- one process per
GPU through torch.distributed.run, rendezvous on 127.0.0.1 -- and rank 0 prints ONE JSON line with n_gpus == N.
--plumbing-only runs everything of the N-rank bench except the device path (gloo here, no GPU): the ID shards, the field
block broadcast from rank 0, the barrier-bracketed loop, the max / sum reductions."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, env_extra=None, timeout=600):
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT')}
    env.update(env_extra or {})
    p = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py')] + args, cwd=ROOT, env=env, capture_output=True,
                       text=True, timeout=timeout)
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith('{')]
    return p, lines


def test_gpus_2_spawns_two_ranks_and_prints_one_line():
    p, lines = _run(['--gpus', '2', '--small', '--plumbing-only', '--steps', '3', '--warmup', '1', '--particles', '1001'])
    assert p.returncode == 0, p.stderr[-2000:]
    assert len(lines) == 1, p.stdout
    d = json.loads(lines[0])
    assert d['n_gpus'] == 2 and d['steps'] == 3 and d['plumbing_only'] is True and d['value'] is None
    assert d['shards_ok'] is True                      # every rank got its ID range and the identical field block
    assert d['units_all_ranks'] == 2 * 1001 * 3        # units summed over the ranks
    assert d['config']['particles_total'] == 2002 and d['scaling'] == 'weak'
    # the communication layer reports how many ranks IT saw (the driver's scaling run reads `comm` the same way: there it says
    # rccl (libodrift_hip.so: odr_comm_*) and N; here, without a GPU, the rehearsal layer)
    assert d['comm']['nranks_seen'] == 2 and d['comm']['backend'] == 'torch.distributed/gloo'


def test_the_n_rank_loop_makes_one_collective_per_step_and_moves_the_reader_levels():
    """`--gpus N` times the SHARDED step (ShardedLoop in bench.py): exactly ONE collective per step (the all-gather of the
    step summaries, as OceanDrift.run() makes it), and a reader level from rank 0 every block_every steps, broadcast one
    period ahead; the level every rank holds at the end is rank 0's array of that level (gloo, two ranks)."""
    p, lines = _run(['--gpus', '2', '--small', '--plumbing-only', '--steps', '13', '--warmup', '1', '--particles', '500',
                     '--block-every', '4'])
    assert p.returncode == 0, p.stderr[-2000:]
    d = json.loads(lines[-1])
    sl = d['sharded_loop']
    assert sl['collectives'] == 13 and sl['collectives_per_step'] == 1.0
    assert sl['block_every'] == 4 and sl['reader_levels'] == 4          # steps 0, 4, 8, 12
    assert d['shards_ok'] is True                                       # kept counts summed over the ranks, the last level intact
    # default period of the N-rank loop: hourly fields at 10-minute steps
    p, lines = _run(['--gpus', '2', '--small', '--plumbing-only', '--steps', '7', '--warmup', '1', '--particles', '500'])
    d = json.loads(lines[-1])
    assert d['sharded_loop']['block_every'] == 6 and d['sharded_loop']['reader_levels'] == 2 and d['sharded_loop']['collectives'] == 7


def test_single_process_default_is_one_rank():
    p, lines = _run(['--small', '--plumbing-only', '--steps', '2', '--workload', 'c4'])
    assert p.returncode == 0, p.stderr[-2000:]
    d = json.loads(lines[-1])
    assert d['n_gpus'] == 1 and d['config']['workload'] == 'c4'
    assert d['comm']['nranks_seen'] == 1 and d['comm']['backend'] is None


def test_cpu_baseline_uses_every_host_core_with_shared_blocks(monkeypatch):
    """cpu_baseline.all_cores (the reference's quasi-parallel mode on ALL host cores, count stated): one
    oracle simulation per core, forked from a helper process that holds the field blocks once (copy-on-write) -- not from the
    bench process, which holds the GPU runtime and its threads."""
    import numpy as np
    sys.path.insert(0, ROOT)
    import bench
    monkeypatch.setenv('ODR_CPU_ALL_SECONDS', '1')
    f = bench.make_fields('c3', small=True)
    out = bench.cpu_baseline('c3', f, 500, np.random.default_rng(1), small=True)
    cores = len(os.sched_getaffinity(0))
    assert out['cores'] == 1 and out['kind'] == 'port' and out['value'] > 0
    a = out['all_cores']
    assert a['cores'] == a['host_cores'] == cores and a['value'] > out['value'] * 0.5, a
