"""Time vitk_adamw (update + both bf16 weight copies) on the headline model's matrix shapes."""
import torch
from vit_amd import ops

shapes = [(4096, 1024), (1024, 4096), (3072, 1024), (1024, 1024)]
for rows, cols in shapes:
    p = torch.randn(rows, cols, device='cuda')
    g = torch.randn_like(p)
    m = torch.zeros_like(p)
    v = torch.zeros_like(p)
    sh = torch.empty(rows, cols, device='cuda', dtype=torch.bfloat16)
    sht = torch.empty(cols, rows, device='cuda', dtype=torch.bfloat16)
    for _ in range(5):
        ops.adamw_step(p, g, m, v, 1e-4, 0.9, 0.999, 1e-8, 1e-2, 1, shadow=(sh, 0), shadow_t=(sht, 0))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 200
    e0.record()
    for i in range(n):
        ops.adamw_step(p, g, m, v, 1e-4, 0.9, 0.999, 1e-8, 1e-2, 2 + i, shadow=(sh, 0), shadow_t=(sht, 0))
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / n
    gb = rows * cols * 32 / 1e9
    print(f'{rows}x{cols}: {us:7.1f} us  {gb / (us * 1e-6) / 1e3:5.2f} TB/s', flush=True)
