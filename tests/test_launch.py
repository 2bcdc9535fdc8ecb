"""The one-node launcher behind `python bench.py --gpus N` (vit_amd/launch.py): the reference's
`torch.distributed.launch --nproc_per_node=N` + env:// rendezvous (distributed_train.sh:7-9, utils/ops.py:124-133),
rehearsed on CPU with gloo."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, env=None, timeout=180):
    e = dict(os.environ)
    for k in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT'):
        e.pop(k, None)
    e.update(env or {})
    return subprocess.run([sys.executable, *args], cwd=ROOT, env=e, capture_output=True, text=True, timeout=timeout)


def test_spawn_two_ranks_one_json_line():
    r = _run(['-m', 'vit_amd.launch', '--selftest', '2'])
    assert r.returncode == 0, r.stderr
    assert 'noise from rank' not in r.stdout     # rank 1's stdout does not reach the parent's
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith('{')]      # (gloo prints a connection banner on stdout)
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out == {'sum': 3.0, 'world': 2}


def test_failing_rank_ends_the_job():
    r = _run(['-m', 'vit_amd.launch', '--selftest', '2'], env={'VITK_LAUNCH_FAIL_RANK': '1'})
    assert r.returncode == 7                     # rank 1's status; rank 0 (stuck in its barrier) was ended by PID
    assert not [ln for ln in r.stdout.splitlines() if ln.startswith('{')]


def test_bench_self_launch_plan():
    """bench.py decides BEFORE any GPU call: --gpus N with no WORLD_SIZE -> spawn N ranks; inside a launched job -> run
    as that rank; a mismatch is an error."""
    sys.path.insert(0, ROOT)
    import bench
    assert bench.launch_plan(1, {}) == 'run'
    assert bench.launch_plan(4, {}) == 'spawn'
    env = {'RANK': '1', 'LOCAL_RANK': '1', 'WORLD_SIZE': '4'}
    assert bench.launch_plan(4, env) == 'run'
    try:
        bench.launch_plan(2, env)
    except SystemExit as e:
        assert 'WORLD_SIZE' in str(e)
    else:
        raise AssertionError('mismatch accepted')
