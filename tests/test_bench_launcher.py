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


def test_a_rank_dying_inside_a_collective_does_not_hang_the_launcher():
    """rank 1 exits non-zero AFTER the process group is up, while ranks 0 and 2 wait for it in an
    all-reduce: the launcher notices, stops the survivors and fails - within seconds, not at the
    backend's timeout"""
    import time
    t0 = time.monotonic()
    r = _run('die1', 3)
    assert r.returncode != 0
    assert 'rank 1 exited with code 5' in r.stderr and 'the other ranks were stopped' in r.stderr
    assert not [ln for ln in r.stdout.splitlines() if ln.startswith('{')]
    assert time.monotonic() - t0 < 120


def test_the_launcher_gives_up_after_its_timeout():
    env_extra = dict(FPL_BENCH_TIMEOUT='3')
    os.environ.update(env_extra)
    try:
        r = _run('hang', 2)
    finally:
        os.environ.pop('FPL_BENCH_TIMEOUT')
    assert r.returncode != 0 and 'no result after 3 s' in r.stderr
