"""Ad-hoc: split-K sweep of the dW (TN) GEMM shape to separate main-loop rate from the atomic epilogue."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vit_amd import ops

def bench(fn, iters=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters

K = 128 * 197
for (m, n) in [(4096, 1024), (1024, 1024), (1024, 3072)]:
    a = torch.randn(K, m, device='cuda').bfloat16()
    b = torch.randn(K, n, device='cuda').bfloat16()
    out = torch.empty(m, n, device='cuda')
    for sk in (1, 2, 4, 8, 16, 32):
        ms = bench(lambda: ops.gemm(a, b, trans_a=True, trans_b=False, out=out, splitk=sk))
        print(f'TN M={m} N={n} K={K} splitk={sk:2d}: {ms:.3f} ms {2.0*m*n*K/ms/1e9:7.1f} TF/s', flush=True)
