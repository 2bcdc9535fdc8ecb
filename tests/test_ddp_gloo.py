"""world_size-2 rehearsal of the gradient reducer on CPU (gloo): averaged gradients equal the
single-process gradient of the concatenated batch, unused parameters keep grad None, and the
.grad tensors are views of the flat buckets."""
import os
import sys

import torch
import torch.distributed as dist
import torch.multiprocessing as mp
import torch.nn as nn

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


class Net(nn.Module):
    def __init__(self):
        super().__init__()
        self.a = nn.Linear(16, 32)
        self.unused = nn.Linear(8, 8)
        self.b = nn.Linear(32, 4)

    def forward(self, x):
        return self.b(torch.tanh(self.a(x)))


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from vit_amd.ddp import GradReducer
    torch.manual_seed(100 + rank)            # different init per rank: the broadcast must fix it
    net = Net()
    red = GradReducer(net, bucket_mb=0.001)  # tiny buckets -> several collectives
    torch.manual_seed(7)
    xs = torch.randn(world * 6, 16)
    ys = torch.randn(world * 6, 4)
    out = {}
    for step in range(3):
        red.prepare()
        x, y = xs[rank * 6:(rank + 1) * 6], ys[rank * 6:(rank + 1) * 6]
        ((net(x) - y) ** 2).mean().backward()
        red.finish()
        out[step] = {n: (None if p.grad is None else p.grad.clone().numpy()) for n, p in net.named_parameters()}
    views = all(p.grad is None or p.grad.untyped_storage().data_ptr() in
                {b.untyped_storage().data_ptr() for b in red.buckets} for p in net.parameters())
    # numpy arrays travel as plain pickles (tensors would be passed as file descriptors, which die
    # with a worker that exits before the parent has received them)
    q.put((rank, {n: p.detach().clone().numpy() for n, p in net.named_parameters()}, out, views, len(red.buckets)))
    dist.barrier()
    dist.destroy_process_group()


def test_reducer_world2_gloo():
    world = 2
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = 29500 + os.getpid() % 2000
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (_, w0, g0, v0, nb0), (_, w1, g1, v1, nb1) = res
    tt = lambda d: {k: (None if v is None else torch.from_numpy(v)) for k, v in d.items()}
    w0, w1 = tt(w0), tt(w1)
    g0, g1 = {s: tt(v) for s, v in g0.items()}, {s: tt(v) for s, v in g1.items()}
    assert v0 and v1 and nb0 == nb1 and nb0 > 1
    for n in w0:
        assert torch.equal(w0[n], w1[n]), f'{n}: ranks hold different weights after broadcast'
    # single-process reference on the concatenated batch with rank 0's (broadcast) weights
    net = Net()
    net.load_state_dict(w0)
    torch.manual_seed(7)
    xs = torch.randn(world * 6, 16)
    ys = torch.randn(world * 6, 4)
    ((net(xs) - ys) ** 2).mean().backward()
    for step in range(3):
        for n, p in net.named_parameters():
            if n.startswith('unused'):
                assert g0[step][n] is None and g1[step][n] is None
                continue
            assert torch.allclose(g0[step][n], p.grad, atol=1e-6), (step, n)
            assert torch.equal(g0[step][n], g1[step][n]), (step, n)
