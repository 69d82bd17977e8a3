"""`python bench.py --gpus N` invoked plainly (no torchrun): the script launches one child
per GPU itself, relays rank 0's JSON line and fails when any rank fails.  Exercised here with
FPL_BENCH_SELFTEST (process group + collective over gloo, no GPU work)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(mode, gpus=2):
    env = dict(os.environ, FPL_BENCH_SELFTEST=mode)
    for k in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_PORT'):
        env.pop(k, None)
    return subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', str(gpus),
                           '--backend', 'gloo'], env=env, stdout=subprocess.PIPE,
                          stderr=subprocess.PIPE, text=True, timeout=300)


def test_plain_gpus_n_spawns_ranks_and_relays_rank0():
    r = _run('1', 3)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line['n_gpus'] == 3 and line['n_ranks_seen'] == 3


def test_a_failing_rank_fails_the_launcher():
    r = _run('fail1')
    assert r.returncode != 0
    assert 'ranks failed' in r.stderr
