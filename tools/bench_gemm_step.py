"""Every large GEMM of one DisLAI-B/16 encoder layer (B = 128, 197 tokens, D = 1024, F = 4096; the token
matrix of the qkv GEMMs carries the 448 relative-table rows), timed in isolation on random data, with the
per-step projection (24 layers).  The per-shape table the GEMM work is steered by."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vit_amd import ops


def bench(fn, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True)
    e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters


M1 = 128 * 197
ME = M1 + 448
D, F = 1024, 4096
BF, F32 = torch.bfloat16, torch.float32
# name, M, N, K, trans_a, trans_b, out dtype, epilogue kwargs builder
SHAPES = [
    ('fwd qkv  ', ME, 3 * D, D, False, True, BF, 'bias'),
    ('fwd proj ', M1, D, D, False, True, BF, 'bias+res'),
    ('fwd fc1  ', M1, F, D, False, True, BF, 'bias+gelu'),
    ('fwd fc2  ', M1, D, F, False, True, BF, 'bias+res'),
    ('dx  fc2  ', M1, F, D, False, True, BF, 'dgelu'),
    ('dx  fc1  ', M1, D, F, False, True, BF, 'res'),
    ('dx  proj ', M1, D, D, False, True, BF, ''),
    ('dx  qkv  ', ME, D, 3 * D, False, True, BF, 'res'),
    ('dw  fc2  ', D, F, M1, True, False, F32, 'acc'),
    ('dw  fc1  ', F, D, M1, True, False, F32, 'acc'),
    ('dw  proj ', D, D, M1, True, False, F32, 'acc'),
    ('dw  qkv  ', 3 * D, D, ME, True, False, F32, 'acc'),
]
only = sys.argv[1:] if len(sys.argv) > 1 else None
tot_ms = tot_fl = 0.0
for (name, m, n, k, ta, tb, od, epi) in SHAPES:
    if only and not any(o in name for o in only):
        continue
    a = torch.randn((k, m) if ta else (m, k), device='cuda').bfloat16()
    b = torch.randn((n, k) if tb else (k, n), device='cuda').bfloat16()
    out = torch.zeros(m, n, device='cuda', dtype=od)
    kw = {}
    if 'bias' in epi:
        kw['bias'] = torch.randn(n, device='cuda')
    if 'res' in epi:
        kw['residual'] = torch.randn(m, n, device='cuda').bfloat16()
    if 'gelu' in epi and 'dgelu' not in epi:
        kw['act'] = ops.ACT_GELU
        kw['aux_out'] = torch.empty(m, n, device='cuda', dtype=BF)
    if 'dgelu' in epi:
        kw['act'] = ops.ACT_DGELU
        kw['aux_in'] = torch.randn(m, n, device='cuda').bfloat16()
    if 'acc' in epi:
        kw['accumulate'] = True
    ms = bench(lambda: ops.gemm(a, b, trans_a=ta, trans_b=tb, out=out, **kw))
    fl = 2.0 * m * n * k
    tot_ms += ms
    tot_fl += fl
    tiles = -(-m // 256) * -(-n // 256)
    print(f'{name} M={m:5d} N={n:4d} K={k:5d} [{epi:9s}] tiles {tiles:4d} ({tiles / 256:5.2f} rounds)  {ms * 1e3:7.1f} us '
          f'{fl / ms / 1e9:7.1f} TF/s', flush=True)
print(f'layer total {tot_ms:.3f} ms = {tot_fl / tot_ms / 1e9:.1f} TF/s  ->  x24 = {24 * tot_ms:.1f} ms per step', flush=True)
