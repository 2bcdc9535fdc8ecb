"""One process per GPU on one node: the launcher behind ``python bench.py --gpus N``.

The reference starts its data-parallel runs as ``python -m torch.distributed.launch --nproc_per_node=$1 main.py``
(distributed_train.sh:7-9) and reads the rendezvous from the environment (``init_method='env://'``,
utils/ops.py:124-133).  ``spawn_ranks`` is that launcher for callers that were started as ONE plain process:
it starts N children with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT set and waits for them.

Rules it keeps (the GPU pool's, and HIP's): the parent never touches the GPU (no HIP call, no
``torch.cuda.is_available()``), children are started as child processes (``subprocess``; nothing is exec'ed over a
process that holds a device), a failing rank ends the others by PID (never by pattern), and rank 0's stdout is the
parent's stdout, so the one JSON line a benchmark prints comes out unchanged.
"""
import os
import socket
import subprocess
import sys
import time

ENV_KEYS = ('RANK', 'LOCAL_RANK', 'WORLD_SIZE')


def launched_by_torchrun(env=None):
    """True when this process already is one rank of a launched job (torch.distributed.run or spawn_ranks)."""
    env = os.environ if env is None else env
    return all(k in env for k in ENV_KEYS)


def free_port(host='127.0.0.1'):
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind((host, 0))
        return s.getsockname()[1]


def rank_env(rank, world, port, base=None, master='127.0.0.1'):
    env = dict(os.environ if base is None else base)
    env.update({'RANK': str(rank), 'LOCAL_RANK': str(rank), 'WORLD_SIZE': str(world), 'LOCAL_WORLD_SIZE': str(world),
                'MASTER_ADDR': master, 'MASTER_PORT': str(port)})
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')      # dmabuf IPC: RCCL between processes needs it on this image
    return env


def spawn_ranks(nproc, argv, port=None, timeout=None, poll=0.2, grace=10.0):
    """Run ``argv`` (a full command line, e.g. [sys.executable, 'bench.py', ...]) as ``nproc`` ranks of one node.
    Returns the first non-zero exit status (0 when every rank succeeded).  When a rank fails or ``timeout``
    seconds pass, the remaining ranks get SIGTERM, then SIGKILL after ``grace`` seconds."""
    if nproc < 1:
        raise ValueError('spawn_ranks: nproc must be >= 1')
    port = port or free_port()
    procs = []
    for r in range(nproc):
        # ranks > 0 keep stderr (tracebacks) but must not write to the parent's stdout: one JSON line, from rank 0
        out = None if r == 0 else subprocess.DEVNULL
        procs.append(subprocess.Popen(list(argv), env=rank_env(r, nproc, port), stdout=out))
    t0 = time.monotonic()
    status = 0
    live = set(range(nproc))
    while live:
        for r in sorted(live):
            rc = procs[r].poll()
            if rc is not None:
                live.discard(r)
                if rc != 0 and status == 0:
                    status = rc
        if status != 0 or (timeout is not None and time.monotonic() - t0 > timeout):
            if status == 0:
                status = 124
            break
        if live:
            time.sleep(poll)
    if live:                                   # a rank failed (or the time ran out): stop the others, by PID
        for r in live:
            procs[r].terminate()
        t1 = time.monotonic()
        while live and time.monotonic() - t1 < grace:
            live = {r for r in live if procs[r].poll() is None}
            time.sleep(poll)
        for r in live:
            procs[r].kill()
        for r in live:
            procs[r].wait()
    return status


def _selftest():
    """A two-line rank program for tests/test_launch.py: gloo all-reduce of (rank + 1), rank 0 prints one JSON line."""
    import json
    import torch
    import torch.distributed as dist
    dist.init_process_group('gloo')
    t = torch.tensor([float(dist.get_rank() + 1)])
    dist.all_reduce(t)
    if os.environ.get('VITK_LAUNCH_FAIL_RANK') == os.environ['RANK']:
        raise SystemExit(7)
    dist.barrier()
    if dist.get_rank() == 0:
        print(json.dumps({'sum': t.item(), 'world': dist.get_world_size()}), flush=True)
    else:
        print('noise from rank', dist.get_rank(), flush=True)     # must not reach the parent's stdout
    dist.destroy_process_group()


if __name__ == '__main__':
    if '--selftest-rank' in sys.argv:
        _selftest()
    elif '--selftest' in sys.argv:
        n = int(sys.argv[sys.argv.index('--selftest') + 1])
        raise SystemExit(spawn_ranks(n, [sys.executable, '-m', 'vit_amd.launch', '--selftest-rank'], timeout=120))
