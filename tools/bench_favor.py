"""Time the FAVOR+ kernels alone at the config-5 shape (batch 64, 577 tokens, 12 heads x 64, m = 64)."""
import os, sys, argparse
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vit_amd import ops
ap = argparse.ArgumentParser()
ap.add_argument('--batch', type=int, default=64); ap.add_argument('--tokens', type=int, default=577)
ap.add_argument('--heads', type=int, default=12); ap.add_argument('--dh', type=int, default=64)
ap.add_argument('--m', type=int, default=64); ap.add_argument('--iters', type=int, default=10)
a = ap.parse_args()
B, L, H, dh, m = a.batch, a.tokens, a.heads, a.dh, a.m
D = H * dh
qkv = (torch.randn(B * L, 3 * D, device='cuda') * 0.5).bfloat16()
q, k, v = (qkv[:, i * D:(i + 1) * D] for i in range(3))
w = torch.randn(m, dh, device='cuda').bfloat16()
ws = ops.favor_workspace(q, B, H, dh, m)
ratio = m ** -0.25
out = ops.favor_fwd(q, k, v, B, H, L, dh, 3 * D, w, ops.FAVOR_SOFTMAX, ratio, 1e-4, ws)
dout = torch.randn_like(out)
dqkv = torch.empty_like(qkv)
dq, dk, dv = (dqkv[:, i * D:(i + 1) * D] for i in range(3))
def timed(fn):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(a.iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / a.iters * 1e3
tf = timed(lambda: ops.favor_fwd(q, k, v, B, H, L, dh, 3 * D, w, ops.FAVOR_SOFTMAX, ratio, 1e-4, ws))
tb = timed(lambda: ops.favor_bwd(q, k, v, B, H, L, dh, 3 * D, w, ops.FAVOR_SOFTMAX, ratio, 1e-4, out, dout, dq, dk, dv, 3 * D, ws))
print(f'B {B} L {L} H {H} dh {dh} m {m}: fwd {tf:.1f} us  bwd {tb:.1f} us')
