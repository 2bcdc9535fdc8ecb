"""The captured train step (vit_amd.graph) against the eager one: same losses, same weights."""
import copy

import pytest
import torch

pytestmark = pytest.mark.gpu


def _model(variant='disentangled_lai_attention'):
    from vit_amd.train import build_model
    torch.manual_seed(0)
    net = build_model(variant, 'S', img_size=32, patch_size=4, num_classes=10,
                      mask_epochs=[10, 20, 30, 40], depth=2, dim=128, mlp=256, heads=4)
    if variant == 'disentangled_performer':      # the random features are drawn at first use: fix them per model
        with torch.no_grad():                    # pos_embed is trained by this variant only; zeros would make the
            net.pos_embed.normal_(0, 0.5, generator=torch.Generator(device='cuda').manual_seed(7))   # max-norm check moot
        for i, lay in enumerate(net.transformer.encoder.layers):
            g = torch.Generator().manual_seed(100 + i)
            lay.dis_attn.w = (torch.randn(16, 32, generator=g) * 0.5).cuda()
    return net


@pytest.mark.parametrize('variant,epoch', [('disentangled_lai_attention', 30), ('disentangled_performer', None)])
def test_graphed_step_matches_eager(variant, epoch):
    from vit_amd import optim
    from vit_amd.graph import GraphedTrainStep
    from vit_amd.train import train_step
    torch.manual_seed(1)
    imgs = [torch.randn(8, 3, 32, 32, device='cuda') for _ in range(4)]
    tgts = [torch.randint(0, 10, (8,), device='cuda') for _ in range(4)]
    warm = 2

    net_e = _model(variant)
    opt_e = optim.AdamW(net_e.parameters(), lr=1e-3)
    losses_e = []
    for _ in range(warm):                                # the graph's warm-up steps, eagerly
        train_step(net_e, opt_e, imgs[0], tgts[0], epoch)
    for x, y in zip(imgs, tgts):
        losses_e.append(train_step(net_e, opt_e, x, y, epoch).item())

    net_g = _model(variant)
    opt_g = optim.AdamW(net_g.parameters(), lr=1e-3, capturable=True)
    gs = GraphedTrainStep(net_g, opt_g, imgs[0], tgts[0], epoch=epoch, warmup=warm)
    losses_g = [gs.step(x, y).item() for x, y in zip(imgs, tgts)]

    for a, b in zip(losses_e, losses_g):
        assert abs(a - b) < 2e-2 * abs(a), (losses_e, losses_g)        # bf16 step, atomics reorder sums
    for (n, p), q in zip(net_e.named_parameters(), net_g.parameters()):
        err = (p - q).abs().max().item() / p.abs().max().clamp_min(1e-6).item()
        assert err < 2e-2, (n, err)
    if variant == 'disentangled_performer':      # the parameter only this variant trains: both runs moved it the same way
        p0 = _model(variant).pos_embed
        ue, ug = (net_e.pos_embed - p0).flatten().float(), (net_g.pos_embed - p0).flatten().float()
        assert ue.abs().max().item() > 0
        assert torch.nn.functional.cosine_similarity(ue, ug, dim=0).item() > 0.9


def test_replays_without_host_sync_match_eager():
    """The host runs many replays ahead of the GPU (nothing synchronises between steps): every replay must still
    see ITS OWN bias corrections / learning rate -- they travel as kernel arguments, not through a reused host buffer."""
    from vit_amd import optim
    from vit_amd.graph import GraphedTrainStep
    from vit_amd.train import train_step
    torch.manual_seed(2)
    x = torch.randn(8, 3, 32, 32, device='cuda')
    y = torch.randint(0, 10, (8,), device='cuda')
    steps, warm = 12, 2
    net_e = _model()
    opt_e = optim.AdamW(net_e.parameters(), lr=3e-3)
    sch_e = torch.optim.lr_scheduler.StepLR(opt_e, step_size=3, gamma=0.5)
    for _ in range(warm):
        train_step(net_e, opt_e, x, y, 30)
    for _ in range(steps):
        train_step(net_e, opt_e, x, y, 30)
        sch_e.step()
    net_g = _model()
    opt_g = optim.AdamW(net_g.parameters(), lr=3e-3, capturable=True)
    sch_g = torch.optim.lr_scheduler.StepLR(opt_g, step_size=3, gamma=0.5)
    gs = GraphedTrainStep(net_g, opt_g, x, y, epoch=30, warmup=warm)
    busy = torch.randn(4096, 4096, device='cuda')
    for _ in range(40):                      # keep the GPU behind the host while the replays are queued
        busy = busy @ busy * 1e-4
    for _ in range(steps):
        gs.step(x, y)                        # no .item(), no synchronize
        sch_g.step()
    torch.cuda.synchronize()
    for (n, p), q in zip(net_e.named_parameters(), net_g.parameters()):
        err = (p - q).abs().max().item() / p.abs().max().clamp_min(1e-6).item()
        assert err < 2e-2, (n, err)


def test_capturable_optimizer_state_resumes():
    """state_dict() of a capturable optimizer after k replays carries step = k; an eager vit_amd AdamW and a
    torch.optim.AdamW that load it take the same next step."""
    from vit_amd import optim
    from vit_amd.graph import GraphedTrainStep
    torch.manual_seed(3)
    x = torch.randn(8, 3, 32, 32, device='cuda')
    y = torch.randint(0, 10, (8,), device='cuda')
    net = _model()
    opt = optim.AdamW(net.parameters(), lr=1e-3, capturable=True)
    warm, k = 2, 3
    gs = GraphedTrainStep(net, opt, x, y, epoch=30, warmup=warm)
    for _ in range(k):
        gs.step(x, y)
    torch.cuda.synchronize()
    sd = opt.state_dict()
    steps = {int(st['step']) for st in sd['state'].values()}
    assert steps == {warm + k}, steps
    used = [p for p in net.parameters() if p in opt.state and opt.state[p]]
    grads = [torch.randn_like(p) for p in used]
    outs = []
    for make in (lambda ps: optim.AdamW(ps, lr=1e-3), lambda ps: torch.optim.AdamW(ps, lr=1e-3)):
        clones = [torch.nn.Parameter(p.detach().clone()) for p in net.parameters()]
        o = make(clones)
        o.load_state_dict(copy.deepcopy(sd))      # load_state_dict adopts the tensors it is given: the moments must not be shared
        by_src = dict(zip(net.parameters(), clones))
        for p, g in zip(used, grads):
            by_src[p].grad = g.clone()
        o.step()
        outs.append([by_src[p].detach().clone() for p in used])
    for a, b in zip(*outs):
        assert ((a - b).abs().max() / b.abs().max().clamp_min(1e-12)).item() < 5e-6
    # and the bias corrections really are those of step warm + k + 1, not of step 1
    fresh = torch.optim.AdamW([torch.nn.Parameter(used[0].detach().clone())], lr=1e-3)
    fresh.param_groups[0]['params'][0].grad = grads[0].clone()
    fresh.step()
    assert (fresh.param_groups[0]['params'][0] - outs[1][0]).abs().max().item() > 0
