"""Debug: config-1 bf16 vs fp32 parameter gradients, per parameter max / L2 error."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..', 'tests'))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..'))
import torch
import golden_inputs as gi
from vit_amd import model, precision as prec
from vit_amd.train import label_smoothing_ce
import numpy as np
g = np.load(os.path.join(os.path.dirname(__file__), '..', 'tests', 'golden', 'config1_step.npz'))
res = {}
for dtype in (torch.float32, torch.bfloat16, torch.bfloat16):
    tr = model.Transformer(d_model=256, dropout=0.0, nhead=8, dim_feedforward=1024, num_encoder_layers=6, normalize_before=True)
    net = model.VisionTransformer(img_size=32, patch_size=4, in_chans=3, num_classes=10, transformer=tr)
    net.load_state_dict(gi.fill_state_dict(net.state_dict(), seed=51))
    prec.set_compute_dtype(net.cuda(), dtype)
    img = gi.randn(51, 'img', 8, 3, 32, 32).cuda()
    tgt = torch.from_numpy(g['target']).cuda()
    loss = label_smoothing_ce(net(img), tgt, 0.1)
    loss.backward()
    cur = {n: p.grad.double().cpu() for n, p in net.named_parameters() if p.grad is not None}
    if dtype == torch.float32:
        ref = cur
        continue
    print('---- bf16 run')
    for n in cur:
        a, b = cur[n].reshape(-1), ref[n].reshape(-1)
        mx = ((a - b).abs().max() / b.abs().max()).item()
        l2 = ((a - b).norm() / b.norm()).item()
        if mx > 0.05:
            i = (a - b).abs().argmax().item()
            print(f'{n:55s} max {mx:.3f} l2 {l2:.3f}  at {i} got {a[i]:.5f} want {b[i]:.5f} bmax {b.abs().max():.5f}')
