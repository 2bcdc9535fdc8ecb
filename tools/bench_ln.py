"""LayerNorm forward / backward and column-sum micro-benchmark at the headline shape."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vit_amd import ops

def bench(fn, iters=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3

M, D = 128 * 197, 1024
x = torch.randn(M, D, device='cuda').bfloat16()
dy = torch.randn(M, D, device='cuda').bfloat16()
add = torch.randn(M, D, device='cuda').bfloat16()
g = torch.randn(D, device='cuda'); b = torch.randn(D, device='cuda')
y, mean, rstd = ops.layernorm_fwd(x, g, b, 1e-5)
dg = torch.zeros(D, device='cuda'); db = torch.zeros(D, device='cuda')
mb = M * D * 2 / 1e6
t = bench(lambda: ops.layernorm_fwd(x, g, b, 1e-5)); print(f'ln fwd      {t:7.1f} us  {2*mb/t*1e-3*1e3/1e3:6.2f} TB/s'.replace('TB/s', 'GB/ms'), flush=True)
t = bench(lambda: ops.layernorm_bwd(dy, x, mean, rstd, g, dg, db)); print(f'ln bwd      {t:7.1f} us  ({3*mb/t:.2f} TB/s algorithmic)')
t = bench(lambda: ops.layernorm_bwd(dy, x, mean, rstd, g, dg, db, add=add)); print(f'ln bwd+add  {t:7.1f} us  ({4*mb/t:.2f} TB/s algorithmic)')
big = torch.randn(M, 4096, device='cuda').bfloat16()
t = bench(lambda: ops.colsum(big)); print(f'colsum 4096 {t:7.1f} us  ({4*mb/t:.2f} TB/s)')
t = bench(lambda: ops.colsum(x)); print(f'colsum 1024 {t:7.1f} us  ({mb/t:.2f} TB/s)')
t = bench(lambda: ops.layernorm_bwd(dy, x, mean, rstd, g, None, None, add=add)); print(f'ln bwd+add without the dgamma/dbeta flush {t:7.1f} us')
dxs = torch.zeros(D, device='cuda')
t = bench(lambda: ops.layernorm_bwd(dy, x, mean, rstd, g, dg, db, add=add, dxsum=dxs)); print(f'ln bwd+add+dxsum {t:7.1f} us')
