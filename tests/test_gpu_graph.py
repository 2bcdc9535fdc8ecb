"""The captured train step (vit_amd.graph) against the eager one: same losses, same weights."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _model():
    from vit_amd.train import build_model
    torch.manual_seed(0)
    return build_model('disentangled_lai_attention', 'S', img_size=32, patch_size=4, num_classes=10,
                       mask_epochs=[10, 20, 30, 40], depth=2, dim=128, mlp=256, heads=4)


def test_graphed_step_matches_eager():
    from vit_amd import optim
    from vit_amd.graph import GraphedTrainStep
    from vit_amd.train import train_step
    torch.manual_seed(1)
    imgs = [torch.randn(8, 3, 32, 32, device='cuda') for _ in range(4)]
    tgts = [torch.randint(0, 10, (8,), device='cuda') for _ in range(4)]
    warm = 2

    net_e = _model()
    opt_e = optim.AdamW(net_e.parameters(), lr=1e-3)
    losses_e = []
    for _ in range(warm):                                # the graph's warm-up steps, eagerly
        train_step(net_e, opt_e, imgs[0], tgts[0], 30)
    for x, y in zip(imgs, tgts):
        losses_e.append(train_step(net_e, opt_e, x, y, 30).item())

    net_g = _model()
    opt_g = optim.AdamW(net_g.parameters(), lr=1e-3, capturable=True)
    gs = GraphedTrainStep(net_g, opt_g, imgs[0], tgts[0], epoch=30, warmup=warm)
    losses_g = [gs.step(x, y).item() for x, y in zip(imgs, tgts)]

    for a, b in zip(losses_e, losses_g):
        assert abs(a - b) < 2e-2 * abs(a), (losses_e, losses_g)        # bf16 step, atomics reorder sums
    for (n, p), q in zip(net_e.named_parameters(), net_g.parameters()):
        err = (p - q).abs().max().item() / p.abs().max().clamp_min(1e-6).item()
        assert err < 2e-2, (n, err)
