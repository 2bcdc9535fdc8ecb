"""Per-item timeline of the persistent 256-tile GEMM (build gemm256.hip with -DVITK_GEMM_STAMPS):
main-loop and epilogue cycles of every item a workgroup walks, and the shader clock under load."""
import sys, os
os.environ.setdefault('VITK_GEMM_DEBUG', '64')
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vit_amd import _lib
_lib.LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), '_stamps', 'libvitk.so')   # see header
from vit_amd import ops
M = 128 * 197
MP = M + 448
for (name, m, n, k, ta, tb) in [('fc1 NT', M, 4096, 1024, False, True), ('proj NT', M, 1024, 1024, False, True),
                                ('fc2 NT', M, 1024, 4096, False, True), ('dx qkv NT', MP, 1024, 3072, False, True),
                                ('dw fc1 TN', 4096, 1024, M, True, False), ('dw proj TN', 1024, 1024, M, True, False),
                                ('dw qkv TN', 3072, 1024, MP, True, False), ('sq4096 NT', 4096, 4096, 4096, False, True)]:
    a = (torch.randn(k, m, device='cuda') if ta else torch.randn(m, k, device='cuda')).bfloat16()
    b = (torch.randn(n, k, device='cuda') if tb else torch.randn(k, n, device='cuda')).bfloat16()
    out = torch.empty(m, n, device='cuda', dtype=torch.float32 if ta else torch.bfloat16)
    st = torch.zeros(m, n, device='cuda', dtype=torch.bfloat16)
    for _ in range(3):
        ops.gemm(a, b, trans_a=ta, trans_b=tb, out=out, aux_out=st)
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    st.zero_(); torch.cuda.synchronize()
    e0.record(); ops.gemm(a, b, trans_a=ta, trans_b=tb, out=out, aux_out=st); e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3
    s = st.view(-1).view(torch.int64)[:256 * 64].view(256, 2, 32).cpu()
    t0 = s[:, :, 0].min().item()
    tend = s[:, :, 1:].max().item()
    print(f'{name}: event {us:.1f} us (cold, one call)', flush=True)
    for blk in (0, 8, 96, 248):
        row = s[blk, 0]
        items = []
        for r in range(10):
            if row[3 + 3 * r].item() == 0: break
            items.append((row[1 + 3 * r].item() - t0, row[2 + 3 * r].item() - row[1 + 3 * r].item(), row[3 + 3 * r].item() - row[2 + 3 * r].item()))
        print(f'  blk {blk:3d} start {row[0].item() - t0:6d}: ' + ' | '.join(f'@{a_:7d} main {b_:6d} epi {c_:6d}' for a_, b_, c_ in items))
    main = []; epi = []
    for r in range(10):
        ok = s[:, 0, 3 + 3 * r] != 0
        if ok.any():
            main.append((s[:, 0, 2 + 3 * r] - s[:, 0, 1 + 3 * r])[ok].float().mean().item())
            epi.append((s[:, 0, 3 + 3 * r] - s[:, 0, 2 + 3 * r])[ok].float().mean().item())
    print('  mean main per round', [int(x) for x in main], ' mean epilogue per round', [int(x) for x in epi])
