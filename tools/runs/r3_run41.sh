#!/bin/bash
export TMPDIR=/tmp
timeout -k 10 120 python - <<'PY'
import torch
def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1)/n*1e-3
for mb in (52, 210, 1024, 4096):
    n=mb*1024*1024//2
    x=torch.randn(n, device='cuda').bfloat16(); y=torch.empty_like(x)
    s=t(lambda: y.copy_(x))
    print(f'copy {mb} MB: {2*n*2/s/1e12:.2f} TB/s ({s*1e6:.1f} us)')
    s=t(lambda: x.mul_(1.0001))
    print(f'inplace scale {mb} MB: {2*n*2/s/1e12:.2f} TB/s')
    s=t(lambda: x.sum())
    print(f'read-only sum {mb} MB: {n*2/s/1e12:.2f} TB/s')
PY
