"""Same-process, interleaved A/B of two builds of libvitk.so on the encoder layer's GEMM shapes
(cdna guide rule 24: perf deltas come from interleaved rounds in ONE process on one device).
usage: python tools/ab_gemm.py libA.so libB.so [shape-name-filter ...]"""
import sys, os, ctypes, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vit_amd import _lib, ops


def load(path):
    L = ctypes.CDLL(os.path.abspath(path))
    for name, sig in _lib._SIGNATURES.items():
        if not hasattr(L, name):        # an older build: only the GEMM entry is exercised here
            continue
        fn = getattr(L, name)
        fn.argtypes, fn.restype = sig
    return L


libs = [load(sys.argv[1]), load(sys.argv[2])]
only = sys.argv[3:]
M1 = 128 * 197
ME = M1 + 448
D, F = 1024, 4096
BF, F32 = torch.bfloat16, torch.float32
SHAPES = [
    ('fwd_qkv', ME, 3 * D, D, False, True, BF, 'bias'),
    ('fwd_proj', M1, D, D, False, True, BF, 'bias+res'),
    ('fwd_fc1', M1, F, D, False, True, BF, 'bias+gelu'),
    ('fwd_fc2', M1, D, F, False, True, BF, 'bias+res'),
    ('dx_fc2', M1, F, D, False, True, BF, 'dgelu'),
    ('dx_fc1', M1, D, F, False, True, BF, 'res'),
    ('dx_proj', M1, D, D, False, True, BF, ''),
    ('dx_qkv', ME, D, 3 * D, False, True, BF, 'res'),
    ('dw_fc2', D, F, M1, True, False, F32, 'acc'),
    ('dw_fc1', F, D, M1, True, False, F32, 'acc'),
    ('dw_proj', D, D, M1, True, False, F32, 'acc'),
    ('dw_qkv', 3 * D, D, ME, True, False, F32, 'acc'),
    ('sq4096', 4096, 4096, 4096, False, True, BF, ''),
]
tot = [0.0, 0.0]
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
    fn = lambda: ops.gemm(a, b, trans_a=ta, trans_b=tb, out=out, **kw)
    outs = []
    ts = [[], []]
    for rnd in range(8):
        for w in (0, 1):
            _lib._lib = libs[w]
            for _ in range(2):
                fn()
            torch.cuda.synchronize()
            s = torch.cuda.Event(enable_timing=True)
            e = torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(10):
                fn()
            e.record()
            torch.cuda.synchronize()
            ts[w].append(s.elapsed_time(e) / 10)
            if rnd == 0 and 'acc' not in epi:
                outs.append(out.float().clone())
    med = [statistics.median(t) for t in ts]
    same = '' if len(outs) < 2 else f' maxdiff {float((outs[0] - outs[1]).abs().max()):.3g}'
    if name != 'sq4096':
        tot[0] += med[0]
        tot[1] += med[1]
    print(f'{name:9s} A {med[0] * 1e3:7.1f} us (min {min(ts[0]) * 1e3:7.1f})  B {med[1] * 1e3:7.1f} us (min {min(ts[1]) * 1e3:7.1f})  '
          f'B/A {med[1] / med[0]:.3f}{same}', flush=True)
print(f'layer: A {tot[0]:.3f} ms  B {tot[1]:.3f} ms  B/A {tot[1] / tot[0]:.3f}')
