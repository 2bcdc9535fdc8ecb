"""Micro-benchmark of vitk_gemm on the headline shapes (B=128 x 197 tokens, D=1024, F=4096)."""
import sys, os, time
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

M = 128 * 197
for (name, m, n, k, ta, tb, od) in [
    ('fwd qkv   NT', M, 3072, 1024, False, True, torch.bfloat16),
    ('fwd proj  NT', M, 1024, 1024, False, True, torch.bfloat16),
    ('fwd fc1   NT', M, 4096, 1024, False, True, torch.bfloat16),
    ('fwd fc2   NT', M, 1024, 4096, False, True, torch.bfloat16),
    ('dx  fc1   NN', M, 1024, 4096, False, False, torch.bfloat16),
    ('dx  fc2   NN', M, 4096, 1024, False, False, torch.bfloat16),
    ('dw  fc1   TN', 4096, 1024, M, True, False, torch.float32),
    ('dw  proj  TN', 1024, 1024, M, True, False, torch.float32),
    ('dw  fc2   TN', 1024, 4096, M, True, False, torch.float32),
    ('sq 4096   NT', 4096, 4096, 4096, False, True, torch.bfloat16),
    ('sq 8192   NT', 8192, 8192, 8192, False, True, torch.bfloat16),
]:
    a = torch.randn((k, m) if ta else (m, k), device='cuda').bfloat16()
    b = torch.randn((n, k) if tb else (k, n), device='cuda').bfloat16()
    out = torch.empty(m, n, device='cuda', dtype=od)
    ms = bench(lambda: ops.gemm(a, b, trans_a=ta, trans_b=tb, out=out))
    tf = 2.0 * m * n * k / ms / 1e9
    # torch (hipBLASLt) reference point, same math
    A = a.t() if ta else a
    Bm = b.t() if tb else b
    ms_t = bench(lambda: torch.matmul(A, Bm))
    print(f'{name}: M={m} N={n} K={k}  vitk {ms:.3f} ms {tf:7.1f} TF/s | torch {ms_t:.3f} ms {2.0*m*n*k/ms_t/1e9:7.1f} TF/s', flush=True)
