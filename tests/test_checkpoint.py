"""Checkpoint format and rank-0-only saving (vit_amd/checkpoint.py; reference train.py:581-593, :619-620).  CPU."""
import argparse
import os

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from vit_amd import checkpoint as C
from vit_amd.optim import AdamW


def _tiny():
    torch.manual_seed(0)
    return torch.nn.Sequential(torch.nn.Linear(8, 16), torch.nn.LayerNorm(16), torch.nn.Linear(16, 4))


def test_format_and_round_trip(tmp_path):
    net = _tiny()
    opt = AdamW(net.parameters(), lr=1e-4)
    sch = torch.optim.lr_scheduler.StepLR(opt, step_size=3, gamma=0.5)
    args = argparse.Namespace(lr=1e-4, attention='disentangled_lai_attention', locality_aware_init=[10, 20, 30, 40])
    path = tmp_path / 'checkpoint.pth'
    C.save_checkpoint(path, net, opt, sch, 7, args)
    assert not os.path.exists(str(path) + '.tmp')
    raw = torch.load(path, weights_only=False)            # a file this test wrote
    assert sorted(raw) == ['args', 'epoch', 'model', 'optimizer', 'scheduler']          # train.py:587-593
    ref_opt = torch.optim.AdamW(_tiny().parameters(), lr=1e-4)
    assert set(raw['optimizer']) == set(ref_opt.state_dict())
    assert set(ref_opt.state_dict()['param_groups'][0]) >= {'lr', 'betas', 'eps', 'weight_decay', 'params'}
    assert {'lr', 'betas', 'eps', 'weight_decay', 'params'} <= set(raw['optimizer']['param_groups'][0])
    ref_opt.load_state_dict(raw['optimizer'])             # torch's optimizer accepts our file

    net2 = _tiny()
    with torch.no_grad():
        for p in net2.parameters():
            p.add_(1.0)
    opt2 = AdamW(net2.parameters(), lr=5e-2)
    sch2 = torch.optim.lr_scheduler.StepLR(opt2, step_size=3, gamma=0.5)
    epoch, a = C.load_checkpoint(path, net2, opt2, sch2)
    assert epoch == 7 and a.attention == 'disentangled_lai_attention' and a.locality_aware_init == [10, 20, 30, 40]
    for p, q in zip(net.parameters(), net2.parameters()):
        assert torch.equal(p, q)
    assert opt2.param_groups[0]['lr'] == opt.param_groups[0]['lr']
    assert sch2.state_dict() == sch.state_dict()

    # the final-model form (train.py:610-615) has no scheduler
    final = tmp_path / 'final.pth'
    C.save_checkpoint(final, net, opt, None, 9, args)
    assert 'scheduler' not in torch.load(final, weights_only=False)
    assert C.load_checkpoint(final, net2, opt2)[0] == 9
    try:
        C.load_checkpoint(final, net2, opt2, sch2)
        raise AssertionError('expected KeyError')
    except KeyError:
        pass


def test_loads_a_torch_adamw_checkpoint(tmp_path):
    """A file written around torch.optim.AdamW (what the reference writes) resumes into vit_amd.optim.AdamW."""
    net = _tiny()
    opt = torch.optim.AdamW(net.parameters(), lr=1e-4)
    net(torch.randn(5, 8)).sum().backward()
    opt.step()
    path = tmp_path / 'ref.pth'
    torch.save({'model': net.state_dict(), 'optimizer': opt.state_dict(), 'epoch': 3, 'args': argparse.Namespace(x=1)},
               path)
    net2 = _tiny()
    opt2 = AdamW(net2.parameters(), lr=1e-2)
    assert C.load_checkpoint(path, net2, opt2)[0] == 3
    for p, q in zip(net.parameters(), net2.parameters()):
        st, st2 = opt.state[p], opt2.state[q]
        assert torch.equal(st['exp_avg'], st2['exp_avg']) and torch.equal(st['exp_avg_sq'], st2['exp_avg_sq'])
        assert float(st2['step']) == 1.0 and 'native' in st2
    assert opt2.param_groups[0]['lr'] == 1e-4


def _rank_worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        net = _tiny()
        opt = AdamW(net.parameters(), lr=1e-4)
        C.save_checkpoint(os.path.join(out_dir, f'rank{rank}.pth'), net, opt, None, 0)
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_only_rank0_writes(tmp_path):
    import socket
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    mp.spawn(_rank_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    assert sorted(os.listdir(tmp_path)) == ['rank0.pth']


def test_loads_a_ddp_wrapped_reference_checkpoint(tmp_path):
    """The reference saves the state_dict of the DDP- / DataParallel-wrapped model in its multi-GPU modes
    (train.py:588 after :635 / :646): every key carries a 'module.' prefix."""
    net = _tiny()
    path = tmp_path / 'ddp.pth'
    torch.save({'model': {'module.' + k: v for k, v in net.state_dict().items()}, 'epoch': 3}, path)
    net2 = _tiny()
    with torch.no_grad():
        for p in net2.parameters():
            p.mul_(0.0)
    assert C.load_checkpoint(path, net2)[0] == 3
    for p, q in zip(net.parameters(), net2.parameters()):
        assert torch.equal(p, q)
