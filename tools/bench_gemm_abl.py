"""GEMM ablation timings (VITK_GEMM_DEBUG: 1 = no epilogue, 2 = no K-loop global loads)."""
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
    return s.elapsed_time(e) / iters

M = 128 * 197
for (name, m, n, k, ta, tb, od) in [
    ('qkv NT', M, 3072, 1024, False, True, torch.bfloat16),
    ('proj NT', M, 1024, 1024, False, True, torch.bfloat16),
    ('fc2 NT', M, 1024, 4096, False, True, torch.bfloat16),
    ('dx fc2 NN', M, 4096, 1024, False, False, torch.bfloat16),
    ('one round 8192x8192 K=1024', 8192, 8192, 1024, False, True, torch.bfloat16),
    ('one wave 4096x4096 K=1024', 4096, 4096, 1024, False, True, torch.bfloat16),
    ('one wave 4096x4096 K=4096', 4096, 4096, 4096, False, True, torch.bfloat16),
]:
    a = torch.randn((k, m) if ta else (m, k), device='cuda').bfloat16()
    b = torch.randn((n, k) if tb else (k, n), device='cuda').bfloat16()
    out = torch.empty(m, n, device='cuda', dtype=od)
    ms = bench(lambda: ops.gemm(a, b, trans_a=ta, trans_b=tb, out=out))
    print(f'dbg={os.environ.get("VITK_GEMM_DEBUG","0")} {name}: {ms*1000:.1f} us {2.0*m*n*k/ms/1e9:7.1f} TF/s', flush=True)
