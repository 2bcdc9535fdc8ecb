"""The gradient reducer on the GPU over RCCL (backend "nccl"), world size 1 (the box has one GPU):
the flat-slab / hook / all-reduce path must give the same training trajectory as no reducer.
The world-2 behaviour is rehearsed on CPU in tests/test_ddp_gloo.py.

Each case runs in a PROCESS OF ITS OWN (this file started as a script with the case's name): a process group is made once
per process, as a training job makes it.  Inside the long-lived pytest process -- hundreds of kernels' worth of streams,
a process group destroyed by the case before -- the second case died once with SIGABRT in the head's weight-gradient GEMM
during the graph's warm-up step (one of three suite runs on the final tree, nothing on stderr, no GPU fault recorded;
not reproduced with the file alone or on two later suite runs), and an abort there ends the whole pytest session."""
import os
import subprocess
import sys

import pytest
import torch
import torch.distributed as dist

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _isolated(case):
    env = dict(os.environ, PYTHONPATH=os.pathsep.join([ROOT, os.path.join(ROOT, 'tests'), os.environ.get('PYTHONPATH', '')]))
    r = subprocess.run([sys.executable, '-u', os.path.abspath(__file__), case], env=env, cwd=ROOT, capture_output=True,
                       text=True, timeout=280)
    assert r.returncode == 0 and 'CASE OK' in r.stdout, f'rc {r.returncode}\n{r.stdout[-2000:]}\n{r.stderr[-4000:]}'


def test_reducer_nccl_world1_matches_plain():
    _isolated('case_reducer_nccl_world1_matches_plain')


def test_graph_captures_the_gradient_exchange():
    _isolated('case_graph_captures_the_gradient_exchange')


def case_reducer_nccl_world1_matches_plain():
    from vit_amd.ddp import GradReducer
    from vit_amd.train import build_model, train_step
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(29600 + os.getpid() % 300), RANK='0', WORLD_SIZE='1')
    dist.init_process_group('nccl', rank=0, world_size=1)
    try:
        torch.manual_seed(0)
        kw = dict(img_size=32, patch_size=4, num_classes=10, dtype=torch.bfloat16, depth=2)
        nets = []
        for _ in range(2):
            torch.manual_seed(0)
            nets.append(build_model('disentangled_lai_attention', 'cifar', **kw))
        nets[1].load_state_dict(nets[0].state_dict())
        x = torch.randn(16, 3, 32, 32, device='cuda')
        y = torch.randint(0, 10, (16,), device='cuda')
        opts = [torch.optim.AdamW(n.parameters(), lr=1e-3) for n in nets]
        red = GradReducer(nets[1], bucket_mb=1)
        losses = [[], []]
        for step in range(3):
            losses[0].append(float(train_step(nets[0], opts[0], x, y, 0)))
            losses[1].append(float(train_step(nets[1], opts[1], x, y, 0, reducer=red)))
        assert len(red.buckets) > 1
        # atomics make the two runs differ in the last bits only
        for a, b in zip(*losses):
            assert abs(a - b) < 2e-2 * max(1.0, abs(a)), (losses)
        unused = [n for n, p in nets[1].named_parameters() if p.grad is None]
        assert sorted(unused) == sorted(n for n, p in nets[0].named_parameters() if p.grad is None)
        for (n, p), (_, q) in zip(nets[0].named_parameters(), nets[1].named_parameters()):
            assert torch.allclose(p.float(), q.float(), atol=5e-3, rtol=5e-2), n
    finally:
        dist.destroy_process_group()


def case_graph_captures_the_gradient_exchange():
    """The whole step INCLUDING the reducer's RCCL all-reduces as one HIP graph (always_reduce: the collectives are
    issued at world size 1 too, on RCCL's stream, through async handles -- the launch structure of an N-GPU step):
    replays follow the eager trajectory."""
    from vit_amd.ddp import GradReducer
    from vit_amd.graph import GraphedTrainStep
    from vit_amd.optim import AdamW
    from vit_amd.train import build_model, train_step
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(29900 + os.getpid() % 90), RANK='0', WORLD_SIZE='1')
    dist.init_process_group('nccl', rank=0, world_size=1)
    try:
        kw = dict(img_size=32, patch_size=4, num_classes=10, dtype=torch.bfloat16, depth=2)
        torch.manual_seed(0)
        ref = build_model('disentangled_lai_attention', 'cifar', **kw)
        net = build_model('disentangled_lai_attention', 'cifar', **kw)
        net.load_state_dict(ref.state_dict())
        x = torch.randn(16, 3, 32, 32, device='cuda')
        y = torch.randint(0, 10, (16,), device='cuda')
        o_ref = AdamW(ref.parameters(), lr=1e-3)
        o_net = AdamW(net.parameters(), lr=1e-3, capturable=True)
        red = GradReducer(net, bucket_mb=1, always_reduce=True)
        g = GraphedTrainStep(net, o_net, x, y, epoch=0, warmup=2, reducer=red)
        assert len(red.buckets) > 1
        eager = [float(train_step(ref, o_ref, x, y, 0)) for _ in range(6)]
        graphed = [float(g.step(x, y)) for _ in range(4)]          # steps 3..6 (two warm-up steps ran eagerly)
        for a, b in zip(eager[2:], graphed):
            assert abs(a - b) < 2e-2 * max(1.0, abs(a)), (eager, graphed)
        assert graphed[-1] < eager[0]
    finally:
        dist.destroy_process_group()


if __name__ == '__main__':
    globals()[sys.argv[1]]()
    print('CASE OK', flush=True)
