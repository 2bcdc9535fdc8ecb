"""Resume from a checkpoint in the reference's format (vit_amd/checkpoint.py) on the GPU: a model that
already holds cached bf16 weight copies and optimizer launch plans loads another run's state and continues exactly
like that run."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _net(seed):
    from vit_amd.train import build_model
    torch.manual_seed(seed)
    return build_model('disentangled_lai_attention', 'S', img_size=32, patch_size=4, num_classes=10,
                       mask_epochs=[10, 20, 30, 40], depth=2, dim=128, mlp=256, heads=4)


@pytest.mark.parametrize('torch_writer', [False, True], ids=['vit_amd_adamw', 'torch_adamw'])
def test_resume_continues_the_run(tmp_path, torch_writer):
    from vit_amd import checkpoint as C, optim
    from vit_amd.train import train_step
    torch.manual_seed(1)
    imgs = [torch.randn(8, 3, 32, 32, device='cuda') for _ in range(3)]
    tgts = [torch.randint(0, 10, (8,), device='cuda') for _ in range(3)]

    a = _net(0)
    opt_a = (torch.optim.AdamW if torch_writer else optim.AdamW)(a.parameters(), lr=1e-3)
    sch_a = torch.optim.lr_scheduler.StepLR(opt_a, step_size=1, gamma=0.5)
    for x, y in zip(imgs[:2], tgts[:2]):
        train_step(a, opt_a, x, y, 30)
    sch_a.step()
    path = tmp_path / 'checkpoint.pth'
    C.save_checkpoint(path, a, opt_a, sch_a, 4)

    b = _net(5)                                   # another init, one step of its own: copies and plans are cached
    opt_b = optim.AdamW(b.parameters(), lr=3e-2)
    sch_b = torch.optim.lr_scheduler.StepLR(opt_b, step_size=1, gamma=0.5)
    train_step(b, opt_b, imgs[2], tgts[2], 30)
    epoch, _ = C.load_checkpoint(path, b, opt_b, sch_b)
    assert epoch == 4 and opt_b.param_groups[0]['lr'] == opt_a.param_groups[0]['lr'] == 5e-4
    for p, q in zip(a.parameters(), b.parameters()):
        assert torch.equal(p, q)

    la = train_step(a, opt_a, imgs[2], tgts[2], 30).item()
    lb = train_step(b, opt_b, imgs[2], tgts[2], 30).item()
    assert abs(la - lb) < 2e-3 * abs(la), (la, lb)            # same weights, same data: only atomics order differs
    for (n, p), q in zip(a.named_parameters(), b.parameters()):
        err = (p - q).abs().max().item() / p.abs().max().clamp_min(1e-6).item()
        assert err < 1e-2, (n, err)
    st = opt_b.state[next(iter(b.transformer.encoder.layers[0].dense_out.parameters()))]
    assert float(st['step']) == 3.0
