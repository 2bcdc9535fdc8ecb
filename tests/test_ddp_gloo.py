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


SLOT_TAKEN = []


class SlotLinearFn(torch.autograd.Function):
    """y = x W^T whose backward writes dW the way the fused HIP blocks do (functional.grad_slot): straight into
    the zeroed view of the reducer's slab when one is on offer, returning a fresh alias of that view."""

    @staticmethod
    def forward(ctx, x, w):
        ctx.save_for_backward(x, w)
        return x @ w.t()

    @staticmethod
    def backward(ctx, dy):
        from vit_amd.functional import grad_slot
        x, w = ctx.saved_tensors
        slot = grad_slot(w)
        SLOT_TAKEN.append(slot is not None)
        if slot is None:
            return dy @ w, dy.t() @ x
        assert float(slot.abs().max()) == 0.0          # prepare() zeroed it
        slot.add_(dy.t() @ x)                          # "accumulate", as the split-K GEMM does
        return dy @ w, slot.detach()


class Net(nn.Module):
    def __init__(self):
        super().__init__()
        self.a = nn.Linear(16, 32)
        self.unused = nn.Linear(8, 8)
        self.mid = nn.Parameter(torch.randn(32, 32) * 0.2)     # takes the direct-write slot
        self.b = nn.Linear(32, 4)

    def forward(self, x):
        return self.b(SlotLinearFn.apply(torch.tanh(self.a(x)), self.mid))


def _worker(rank, world, port, q, comm_dtype=torch.float32):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from vit_amd.ddp import GradReducer
    torch.manual_seed(100 + rank)            # different init per rank: the broadcast must fix it
    net = Net()
    versions = [p._version for p in net.parameters()]
    red = GradReducer(net, bucket_mb=0.001, comm_dtype=comm_dtype)  # tiny buckets -> several collectives
    # the broadcast wrote through the parameters themselves: their version counters moved (functional.shadow keys
    # its cached compute-dtype copies on them)
    assert all(p._version > v for p, v in zip(net.parameters(), versions))
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
    # step 1 discovers the used parameters (no slot yet); from step 2 on the slab view is handed out and taken
    assert SLOT_TAKEN == [False, True, True], SLOT_TAKEN
    assert net.mid.grad.data_ptr() == red._view_of[net.mid].data_ptr()
    q.put((rank, {n: p.detach().clone().numpy() for n, p in net.named_parameters()}, out, views, len(red.buckets)))
    dist.barrier()
    dist.destroy_process_group()


def _run_world2(comm_dtype, port_off=0):
    world = 2
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = 29500 + (os.getpid() + port_off) % 2000
    procs = [ctx.Process(target=_worker, args=(r, world, port, q, comm_dtype)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    return res


def _reference(w0, world=2):
    net = Net()
    net.load_state_dict(w0)
    torch.manual_seed(7)
    xs = torch.randn(world * 6, 16)
    ys = torch.randn(world * 6, 4)
    ((net(xs) - ys) ** 2).mean().backward()
    return net


def test_reducer_world2_gloo():
    world = 2
    res = _run_world2(torch.float32)
    (_, w0, g0, v0, nb0), (_, w1, g1, v1, nb1) = res
    tt = lambda d: {k: (None if v is None else torch.from_numpy(v)) for k, v in d.items()}
    w0, w1 = tt(w0), tt(w1)
    g0, g1 = {s: tt(v) for s, v in g0.items()}, {s: tt(v) for s, v in g1.items()}
    assert v0 and v1 and nb0 == nb1 and nb0 > 1
    for n in w0:
        assert torch.equal(w0[n], w1[n]), f'{n}: ranks hold different weights after broadcast'
    # single-process reference on the concatenated batch with rank 0's (broadcast) weights
    net = _reference(w0)
    for step in range(3):
        for n, p in net.named_parameters():
            if n.startswith('unused'):
                assert g0[step][n] is None and g1[step][n] is None
                continue
            assert torch.allclose(g0[step][n], p.grad, atol=1e-6), (step, n)
            assert torch.equal(g0[step][n], g1[step][n]), (step, n)


def test_reducer_world2_bf16_buckets():
    """comm_dtype=bf16 halves the bytes on the wire; the averaged gradient then carries one bf16 rounding per rank."""
    res = _run_world2(torch.bfloat16, port_off=37)
    (_, w0, g0, _, _), (_, w1, g1, _, _) = res
    tt = lambda d: {k: (None if v is None else torch.from_numpy(v)) for k, v in d.items()}
    w0 = tt(w0)
    g0, g1 = {s: tt(v) for s, v in g0.items()}, {s: tt(v) for s, v in g1.items()}
    net = _reference(w0)
    for step in range(3):
        for n, p in net.named_parameters():
            if n.startswith('unused'):
                continue
            assert torch.equal(g0[step][n], g1[step][n]), (step, n)
            err = (g0[step][n] - p.grad).abs().max() / p.grad.abs().max()
            assert err < 2 ** -7, (step, n, float(err))       # bf16: 8 significant bits per addend
