"""Poison every torch.empty() of the package with NaN and run an eager step: any kernel that reads memory it
(or a kernel before it) did not write shows up as NaN in the loss or the gradients."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
_empty = torch.empty
def poisoned(*a, **k):
    t = _empty(*a, **k)
    if t.is_floating_point() and t.numel():
        t.fill_(float('nan'))
    elif t.dtype == torch.uint8 and t.numel():
        t.fill_(0xFF)
    return t
torch.empty = poisoned
from vit_amd.train import build_model, train_step
from vit_amd import optim
wl = sys.argv[1] if len(sys.argv) > 1 else 'disentangled_lai_attention'
preset = sys.argv[2] if len(sys.argv) > 2 else 'B'
torch.manual_seed(0)
kw = dict(depth=2) if preset != 'cifar' else {}
img = 224 if preset != 'cifar' else 32
net = build_model(wl, preset, img_size=img, patch_size=16 if img == 224 else 4, num_classes=1000 if img == 224 else 10,
                  mask_epochs=[10, 20, 30, 40] if 'lai' in wl else None, **kw)
opt = optim.AdamW(net.parameters(), lr=1e-4)
x = torch.randn(16, 3, img, img, device='cuda'); y = torch.randint(0, 10, (16,), device='cuda')
ep = 30 if 'lai' in wl else None
def hook(name):
    def f(mod, inp, out):
        o = out[0] if isinstance(out, (tuple, list)) else (out['hidden_states'] if isinstance(out, dict) else out)
        if torch.is_tensor(o) and not torch.isfinite(o.float()).all():
            print('  non-finite forward output:', name, type(mod).__name__)
    return f
for n, m in net.named_modules():
    m.register_forward_hook(hook(n))
for it in range(2):
    loss = train_step(net, opt, x, y, ep)
    torch.cuda.synchronize()
    bad = [n for n, p in net.named_parameters() if p.grad is not None and not torch.isfinite(p.grad).all()]
    print(wl, preset, 'step', it, 'loss', loss.item(), 'non-finite grads:', bad[:6])
