"""In-kernel stamps of the 256-tile GEMM (VITK_GEMM_DEBUG=32): cycles per phase section."""
import sys, os
os.environ['VITK_GEMM_DEBUG'] = '32'
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vit_amd import _lib
_lib.LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), '_stamps', 'libvitk.so')   # gemm256.hip built with -DVITK_GEMM_STAMPS
from vit_amd import ops
for (name, m, n, k, tb) in [('sq4096 NT', 4096, 4096, 4096, True), ('qkv NT', 128 * 197, 3072, 1024, True), ('sq4096 NN', 4096, 4096, 4096, False)]:
    a = torch.randn(m, k, device='cuda').bfloat16()
    b = (torch.randn(n, k, device='cuda') if tb else torch.randn(k, n, device='cuda')).bfloat16()
    out = torch.empty(m, n, device='cuda', dtype=torch.bfloat16)
    st = torch.zeros(m, n, device='cuda', dtype=torch.bfloat16)
    for _ in range(5):
        ops.gemm(a, b, trans_b=tb, out=out, aux_out=st)
    torch.cuda.synchronize()
    s = st.view(-1).view(torch.int64)[:256 * 64].view(256, 2, 32).cpu()
    for blk in (0, 7, 100):
        for grp in (0, 1):
            t = s[blk, grp, :21].tolist()
            d = [t[i + 1] - t[i] for i in range(20)]
            print(name, f'blk {blk} grp {grp} t0={t[0] - s[blk,0,0].item():6d}:',
                  ' | '.join(' '.join(f'{x:4d}' for x in d[5 * ph:5 * ph + 5]) for ph in range(4)), ' tile', t[20] - t[0], flush=True)
    print(name, 'epilogue cycles grp0/grp1', (s[:, 0, 22] - s[:, 0, 21]).float().mean().item(), (s[:, 1, 22] - s[:, 1, 21]).float().mean().item())
    print('   columns per phase: mem(reads+glds+vmcnt) lgkmwait barrier1 mfma barrier2')
    d = (s[:, :, 1:21] - s[:, :, 0:20]).float().mean(0)
    for grp in (0, 1):
        print(name, f'MEAN grp {grp}:', ' | '.join(' '.join(f'{x:5.0f}' for x in d[grp, 5 * ph:5 * ph + 5].tolist()) for ph in range(4)),
              ' tile', (s[:, grp, 20] - s[:, grp, 0]).float().mean().item())
