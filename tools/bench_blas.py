"""How fast are the library GEMMs (torch.mm -> hipBLASLt / rocBLAS) on the layer's shapes?  A yardstick for gemm256."""
import torch, sys
def timed(fn, iters=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
M1 = 128 * 197; ME = M1 + 448; D, F = 1024, 4096
shapes = [('fwd_qkv', ME, 3 * D, D, 'nt'), ('fwd_proj', M1, D, D, 'nt'), ('fwd_fc1', M1, F, D, 'nt'), ('fwd_fc2', M1, D, F, 'nt'),
          ('dx_fc2', M1, F, D, 'nn'), ('dx_fc1', M1, D, F, 'nn'), ('dx_proj', M1, D, D, 'nn'), ('dx_qkv', ME, D, 3 * D, 'nn'),
          ('dw_fc2', D, F, M1, 'tn'), ('dw_fc1', F, D, M1, 'tn'), ('dw_proj', D, D, M1, 'tn'), ('dw_qkv', 3 * D, D, ME, 'tn')]
tot = 0
for name, m, n, k, lay in shapes:
    if lay == 'nt':
        a = torch.randn(m, k, device='cuda').bfloat16(); b = torch.randn(n, k, device='cuda').bfloat16()
        f = lambda: torch.mm(a, b.t())
    elif lay == 'nn':
        a = torch.randn(m, k, device='cuda').bfloat16(); b = torch.randn(k, n, device='cuda').bfloat16()
        f = lambda: torch.mm(a, b)
    else:
        a = torch.randn(k, m, device='cuda').bfloat16(); b = torch.randn(k, n, device='cuda').bfloat16()
        f = lambda: torch.mm(a.t(), b)
    t = timed(f); tot += t
    print(f'{name:9s} {lay} {m:6d} {n:5d} {k:6d}  {t:7.1f} us  {2.0 * m * n * k / t / 1e6:7.1f} TF/s', flush=True)
print(f'layer {tot / 1e3:.3f} ms')
