"""vit_amd.optim.AdamW (vitk_adamw through the C-ABI) against torch.optim.AdamW -- the optimizer the
reference builds (train.py:275) -- and the bf16 weight copies it maintains for the GEMMs."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def relerr(a, b):
    a, b = a.float(), b.float()
    return ((a - b).norm() / b.norm().clamp_min(1e-30)).item()


@pytest.mark.parametrize('shape', [(256, 512), (1000, 1024), (70, 36), (1, 1, 256), (64, 3, 4, 4), (3072, 1024)])
def test_adamw_matches_torch(shape):
    from vit_amd import optim
    torch.manual_seed(0)
    p0 = torch.randn(*shape, device='cuda')
    b0 = torch.randn(shape[0], device='cuda')
    pa, ba = torch.nn.Parameter(p0.clone()), torch.nn.Parameter(b0.clone())
    pb, bb = torch.nn.Parameter(p0.clone()), torch.nn.Parameter(b0.clone())
    oa = optim.AdamW([pa, ba], lr=1e-2, weight_decay=0.05)
    ob = torch.optim.AdamW([pb, bb], lr=1e-2, weight_decay=0.05)
    sched_a = torch.optim.lr_scheduler.StepLR(oa, step_size=2, gamma=0.5)
    sched_b = torch.optim.lr_scheduler.StepLR(ob, step_size=2, gamma=0.5)
    for it in range(5):
        g = torch.randn(*shape, device='cuda') * (1.0 + it)
        gb = torch.randn(shape[0], device='cuda')
        pa.grad, pb.grad, ba.grad, bb.grad = g.clone(), g.clone(), gb.clone(), gb.clone()
        oa.step(); ob.step()
        sched_a.step(); sched_b.step()
        assert relerr(pa, pb) < 2e-6, it
        assert relerr(ba, bb) < 2e-6, it
    assert relerr(oa.state[pa]['exp_avg'], ob.state[pb]['exp_avg']) < 2e-6
    assert relerr(oa.state[pa]['exp_avg_sq'], ob.state[pb]['exp_avg_sq']) < 2e-6
    assert int(oa.state[pa]['step']) == 5


def test_adamw_refreshes_weight_copies():
    """The cached bf16 copies (plain and transposed, single and fused-QKV) follow the in-place update
    bit for bit: equal to a fresh cast of the updated parameter."""
    from vit_amd import optim
    from vit_amd import functional as VF
    torch.manual_seed(1)
    D = 256
    wq, wk, wv = (torch.nn.Parameter(torch.randn(D, D, device='cuda') * 0.05) for _ in range(3))
    wo = torch.nn.Parameter(torch.randn(D, 4 * D, device='cuda') * 0.05)
    opt = optim.AdamW([wq, wk, wv, wo], lr=1e-2)
    bf = torch.bfloat16
    for it in range(3):
        n = VF.shadow((wq, wk, wv), bf)
        t = VF.shadow((wq, wk, wv), bf, transposed=True)
        on = VF.shadow(wo, bf)
        ot = VF.shadow(wo, bf, transposed=True)
        if it:   # second step on: the very buffers the optimizer wrote into are handed out again
            assert n.data_ptr() == ptrs[0] and t.data_ptr() == ptrs[1] and on.data_ptr() == ptrs[2] and ot.data_ptr() == ptrs[3]
        ptrs = (n.data_ptr(), t.data_ptr(), on.data_ptr(), ot.data_ptr())
        for p in (wq, wk, wv, wo):
            p.grad = torch.randn_like(p)
        opt.step()
        cat = torch.cat([wq, wk, wv], 0).detach()
        assert torch.equal(VF.shadow((wq, wk, wv), bf), cat.to(bf))
        assert torch.equal(VF.shadow((wq, wk, wv), bf, transposed=True), cat.t().to(bf).contiguous())
        assert torch.equal(VF.shadow(wo, bf), wo.detach().to(bf))
        assert torch.equal(VF.shadow(wo, bf, transposed=True), wo.detach().t().to(bf).contiguous())


def test_adamw_rejects_cpu_and_unsupported():
    from vit_amd import optim
    with pytest.raises(ValueError):
        optim.AdamW([torch.nn.Parameter(torch.zeros(4, 4, device='cuda'))], amsgrad=True)
    p = torch.nn.Parameter(torch.zeros(8))
    p.grad = torch.ones(8)
    with pytest.raises(RuntimeError):
        optim.AdamW([p]).step()
