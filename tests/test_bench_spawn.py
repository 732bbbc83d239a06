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


def test_single_process_default_is_one_rank():
    p, lines = _run(['--small', '--plumbing-only', '--steps', '2', '--workload', 'c4'])
    assert p.returncode == 0, p.stderr[-2000:]
    d = json.loads(lines[-1])
    assert d['n_gpus'] == 1 and d['config']['workload'] == 'c4'
