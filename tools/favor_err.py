import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
import torch
from oracle import ref_cpu as R
from vit_amd import functional as VF

def relerr(a, b):
    a, b = a.double().cpu().reshape(-1), b.double().cpu().reshape(-1)
    return ((a - b).abs().max() / b.abs().max()).item(), ((a - b).norm() / b.norm()).item()

for kernel, m, L in [('softmax', 64, 145), ('relu', 64, 145), ('softmax', 16, 145)]:
    B, H, dh = 2, 4, 64
    D = H * dh
    torch.manual_seed(5)
    qkv = torch.randn(B * L, 3 * D) * 0.6
    w = R.gaussian_orthogonal_random_matrix(m, dh, generator=torch.Generator().manual_seed(1))
    gout = torch.randn(B * L, D)
    ratio = m ** -0.25
    def oracle(x):
        qr = x.clone().requires_grad_(True)
        q, k, v = (qr[:, i * D:(i + 1) * D].view(B, L, H, dh) for i in range(3))
        if kernel == 'softmax':
            kp, qp = R.favor_softmax_features(k, w, False, ratio), R.favor_softmax_features(q, w, True, ratio)
        else:
            kp, qp = R.favor_relu_features(k, w, ratio), R.favor_relu_features(q, w, ratio)
        ref = R.favor_linear_attention(qp, kp, v).reshape(B * L, D)
        (ref * gout).sum().backward()
        return ref.detach(), qr.grad
    ref, gref = oracle(qkv)
    # oracle on bf16-rounded inputs (w too): the error floor of ANY bf16-input implementation
    w_keep = w
    w = w.bfloat16().float()
    ref_r, gref_r = oracle(qkv.bfloat16().float())
    w = w_keep
    print(kernel, m, 'input-rounding floor  y', relerr(ref_r, ref), 'dx', relerr(gref_r, gref))
    for dt in (torch.float32, torch.bfloat16):
        xg = qkv.clone().cuda().to(dt).requires_grad_(True)
        out = VF.favor_attention(xg, w.cuda(), B, L, H, dh, kernel, ratio)
        (out.float() * gout.cuda()).sum().backward()
        print(kernel, m, dt, 'y', relerr(out.detach().float(), ref), 'dx', relerr(xg.grad.float(), gref),
              ' vs rounded-input oracle: y', relerr(out.detach().float(), ref_r), 'dx', relerr(xg.grad.float(), gref_r))
