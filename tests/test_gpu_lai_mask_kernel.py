"""The LAI mask AS THE KERNELS EVALUATE IT, bit for bit against the reference's get_attn_mask
(timm_classes/timm_locality_aware_transformer.py:11-21, dup disentangled_transformer.py:1000-1010).

The kernels never build a mask tensor: they evaluate |i - j| <= w per score tile (csrc/attention_band.hip,
csrc/attention.hip).  To see that predicate from outside, the forward runs with V = one-hot rows: head h gets
V[j][d] = (j == 64 h + d), so out[i][h][d] = P_h[i][64 h + d] -- the probability itself, one term, no sum.  Inside the band
a probability is a positive float (never rounds to 0 in bf16 or fp32 at these score magnitudes), outside the kernels
produce exactly 0 (the reference's -1e9 fill underflows to 0 in fp32 as well).  The recovered zero pattern,
packed to bits, must equal tests/golden/lai_masks.npz (packbits of the reference's own mask, written by
tools/gen_goldens.py) for every N x w the survey names, on the whole-head bf16 kernels and the generic fp32 ones, with
and without the disentangled bias."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = np.load(os.path.join(ROOT, 'tests', 'golden', 'lai_masks.npz'))


@pytest.mark.parametrize('dtype', [torch.bfloat16, torch.float32], ids=['bf16', 'f32'])
@pytest.mark.parametrize('bias', [False, True], ids=['plain', 'dis'])
@pytest.mark.parametrize('w', [8, 12, 18, 20])
@pytest.mark.parametrize('N', [65, 197, 577])
def test_kernel_mask_bits_equal_reference(N, w, bias, dtype):
    from vit_amd import ops
    dh = 64
    H = (N + dh - 1) // dh
    D = H * dh
    g = torch.Generator(device='cuda').manual_seed(100 * N + w)
    # every head gets the SAME q and k (and table) rows, so all heads compute one P and the pieces read from different heads
    # belong to one softmax (the row sums below then check that nothing leaked outside the band)
    rep = lambda rows: (torch.randn(rows, dh, device='cuda', generator=g) * 0.3).to(dtype).repeat(1, H)
    qkv = torch.empty(N, 3 * D, device='cuda', dtype=dtype)
    qkv[:, :D] = rep(N)
    qkv[:, D:2 * D] = rep(N)
    v = torch.zeros(N, H, dh, device='cuda', dtype=dtype)
    j = torch.arange(N, device='cuda')
    v[j, j // dh, j % dh] = 1.0
    qkv[:, 2 * D:] = v.view(N, D)
    pk = pq = None
    if bias:
        pk = rep(2 * N - 1).contiguous()
        pq = rep(2 * N - 1).contiguous()
    q, k, vv = qkv[:, :D], qkv[:, D:2 * D], qkv[:, 2 * D:]
    scale = (dh * (3 if bias else 1)) ** -0.5
    out, _ = ops.attn_fwd(q, k, vv, 1, H, N, dh, qkv.stride(0), pk, pq, D, scale, w)
    P = out.float().view(N, D)[:, :N]                      # P[i][j] of head j // 64
    pattern = (P != 0).cpu().numpy().astype(np.uint8)
    assert (P >= 0).all()
    bits = np.packbits(pattern.reshape(-1))
    want = GOLD[f'bits_{N}_{w}']
    if not np.array_equal(bits, want):
        ref = np.unpackbits(want)[:N * N].reshape(N, N)
        bad = np.argwhere(ref != pattern)
        raise AssertionError(f'kernel mask differs from get_attn_mask({N}, {w}) at {len(bad)} positions, first {bad[:5].tolist()}')
    # rows sum to 1 over the band: nothing leaked outside it (bf16: up to 2 w + 1 = 41 outputs, each rounded to 8 bits)
    assert torch.allclose(P.sum(1), torch.ones(N, device='cuda'), atol=5e-2 if dtype == torch.bfloat16 else 1e-5)
