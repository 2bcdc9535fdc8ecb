"""GPU parity of the fused attention kernels against the closed form in fp64."""
import pytest
import torch

from attn_ref import attention_ref

pytestmark = pytest.mark.gpu


def relerr(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return ((a - b).abs().max() / (b.abs().max() + 1e-30)).item()


def make(B, N, H, dh, dtype, bias, seed=0):
    g = torch.Generator(device='cuda').manual_seed(seed)
    D = H * dh
    qkv = (torch.randn(B * N, 3 * D, device='cuda', generator=g) * 0.8).to(dtype)
    pk = pq = None
    if bias:
        pk = (torch.randn(2 * N - 1, D, device='cuda', generator=g) * 0.8).to(dtype)
        pq = (torch.randn(2 * N - 1, D, device='cuda', generator=g) * 0.8).to(dtype)
    return qkv, pk, pq


def views(qkv, B, N, H, dh):
    D = H * dh
    return tuple(qkv[:, t * D:(t + 1) * D] for t in range(3))


CASES = [(2, 65, 4, 32), (1, 197, 3, 64), (1, 197, 2, 80), (1, 130, 2, 128), (2, 16, 2, 64), (1, 577, 2, 64),
         (2, 130, 3, 72), (1, 208, 2, 80)]      # heads of 72 / 80: whole-head kernels on 80-wide row images


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16], ids=['f32', 'bf16'])
@pytest.mark.parametrize('bias', [False, True], ids=['plain', 'dis'])
@pytest.mark.parametrize('w', [-1, 8, 20, 40])
@pytest.mark.parametrize('B,N,H,dh', CASES)
def test_attn_fwd(dtype, bias, w, B, N, H, dh):
    from vit_amd import ops
    if N == 577 and (dtype == torch.float32 and bias):
        pytest.skip('covered by the bf16 / plain cases; keeps the suite short')
    qkv, pk, pq = make(B, N, H, dh, dtype, bias, seed=N + dh)
    q, k, v = views(qkv, B, N, H, dh)
    scale = (dh * (3 if bias else 1)) ** -0.5
    out, lse = ops.attn_fwd(q, k, v, B, H, N, dh, qkv.stride(0), pk, pq, H * dh, scale, w)
    f = lambda t: t.double().reshape(B, N, H, dh)
    ref, lse_ref, _ = attention_ref(f(q), f(k), f(v),
                                    None if pk is None else pk.double().view(-1, H, dh),
                                    None if pq is None else pq.double().view(-1, H, dh), scale, w)
    tol = 1e-5 if dtype == torch.float32 else 1.5e-2
    assert relerr(out.view(B, N, H, dh), ref) < tol
    assert relerr(lse, lse_ref) < (1e-5 if dtype == torch.float32 else 2e-3)


BWD_CASES = [(2, 65, 4, 32), (1, 197, 3, 64), (1, 197, 2, 80), (2, 16, 2, 64), (1, 300, 2, 64), (3, 130, 2, 128),
             (2, 130, 3, 72), (1, 208, 2, 80)]


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16], ids=['f32', 'bf16'])
@pytest.mark.parametrize('bias', [False, True], ids=['plain', 'dis'])
@pytest.mark.parametrize('w', [-1, 8, 20, 40])
@pytest.mark.parametrize('B,N,H,dh', BWD_CASES)
def test_attn_bwd(dtype, bias, w, B, N, H, dh):
    from vit_amd import ops
    D = H * dh
    qkv, pk, pq = make(B, N, H, dh, dtype, bias, seed=3 * N + dh)
    q, k, v = views(qkv, B, N, H, dh)
    scale = (dh * (3 if bias else 1)) ** -0.5
    out, lse = ops.attn_fwd(q, k, v, B, H, N, dh, qkv.stride(0), pk, pq, D, scale, w)
    g = torch.Generator(device='cuda').manual_seed(7)
    dout = torch.randn(B * N, D, device='cuda', generator=g).to(dtype)
    dqkv = torch.full_like(qkv, float('nan'))
    dq, dk, dv = views(dqkv, B, N, H, dh)
    dpk = dpq = None
    if bias:
        dpk = torch.zeros(2 * N - 1, D, device='cuda')
        dpq = torch.zeros(2 * N - 1, D, device='cuda')
    dbias = torch.ones(3 * D, device='cuda')                  # accumulated into: starts at 1
    ops.attn_bwd(q, k, v, B, H, N, dh, qkv.stride(0), pk, pq, D, scale, w, out, dout, lse,
                 dq, dk, dv, dqkv.stride(0), dpk, dpq, dbias=dbias)
    # fp64 autograd of the closed form on the same (rounded) inputs
    f = lambda t: t.double().reshape(B, N, H, dh).clone().requires_grad_(True)
    qr, kr, vr = f(q), f(k), f(v)
    pkr = pqr = None
    if bias:
        pkr = pk.double().view(-1, H, dh).clone().requires_grad_(True)
        pqr = pq.double().view(-1, H, dh).clone().requires_grad_(True)
    ref, _, _ = attention_ref(qr, kr, vr, pkr, pqr, scale, w)
    (ref * dout.double().view(B, N, H, dh)).sum().backward()
    tol = 1e-5 if dtype == torch.float32 else 2e-2
    assert relerr(dq.reshape(B, N, H, dh), qr.grad) < tol
    assert relerr(dk.reshape(B, N, H, dh), kr.grad) < tol
    assert relerr(dv.reshape(B, N, H, dh), vr.grad) < tol
    if bias:
        assert relerr(dpk.view(-1, H, dh), pkr.grad) < tol
        assert relerr(dpq.view(-1, H, dh), pqr.grad) < tol
    # the fused q|k|v bias gradient: column sums of dq | dk | dv over every token
    db_ref = torch.cat([t.grad.sum((0, 1)).reshape(-1) for t in (qr, kr, vr)]) + 1.0
    assert relerr(dbias, db_ref) < tol


# The banded disentangled kernels of attention_lai.hip (bf16, head width 64, one chunk, 0 <= w <= 20): every LAI stage of
# the reference (8 / 12 / 18 / 20), the 3- and 5-diagonal instances on both sides of w = 16, one-tile and 13-tile heads,
# ragged last tiles, and a batch larger than the workgroups per head (20 > 256 / 16: a workgroup walks two batch elements
# and keeps its table gradients across them).
LAI_CASES = [(3, 65, 2, 64, 8), (3, 65, 2, 64, 20), (2, 208, 2, 64, 12), (2, 208, 2, 64, 18), (2, 100, 4, 64, 20),
             (20, 33, 16, 64, 8), (20, 33, 16, 64, 18), (1, 197, 3, 64, 12), (1, 197, 3, 64, 18), (2, 197, 2, 64, 1),
             (2, 197, 2, 64, 5), (2, 197, 2, 64, 16), (2, 197, 2, 64, 17), (2, 16, 2, 64, 15), (4, 21, 1, 64, 20),
             (128, 197, 16, 64, 8), (128, 197, 16, 64, 20)]        # the last two: BASELINE.json's full shape (8 heads per workgroup)


@pytest.mark.parametrize('B,N,H,dh,w', LAI_CASES)
def test_lai_dis_kernels(B, N, H, dh, w):
    from vit_amd import ops
    dtype = torch.bfloat16
    D = H * dh
    qkv, pk, pq = make(B, N, H, dh, dtype, True, seed=5 * N + w)
    q, k, v = views(qkv, B, N, H, dh)
    scale = (dh * 3) ** -0.5
    out, lse = ops.attn_fwd(q, k, v, B, H, N, dh, qkv.stride(0), pk, pq, D, scale, w)
    f = lambda t: t.double().reshape(B, N, H, dh).clone().requires_grad_(True)
    qr, kr, vr = f(q), f(k), f(v)
    pkr = pk.double().view(-1, H, dh).clone().requires_grad_(True)
    pqr = pq.double().view(-1, H, dh).clone().requires_grad_(True)
    ref, lse_ref, _ = attention_ref(qr, kr, vr, pkr, pqr, scale, w)
    assert relerr(out.view(B, N, H, dh), ref.detach()) < 1.5e-2
    assert relerr(lse, lse_ref.detach()) < 2e-3
    g = torch.Generator(device='cuda').manual_seed(11)
    dout = torch.randn(B * N, D, device='cuda', generator=g).to(dtype)
    dqkv = torch.full_like(qkv, float('nan'))
    dq, dk, dv = views(dqkv, B, N, H, dh)
    dpk = torch.zeros(2 * N - 1, D, device='cuda')
    dpq = torch.zeros(2 * N - 1, D, device='cuda')
    dbias = torch.ones(3 * D, device='cuda')
    ops.attn_bwd(q, k, v, B, H, N, dh, qkv.stride(0), pk, pq, D, scale, w, out, dout, lse,
                 dq, dk, dv, dqkv.stride(0), dpk, dpq, dbias=dbias)
    (ref * dout.double().view(B, N, H, dh)).sum().backward()
    tol = 2e-2
    assert relerr(dq.reshape(B, N, H, dh), qr.grad) < tol
    assert relerr(dk.reshape(B, N, H, dh), kr.grad) < tol
    assert relerr(dv.reshape(B, N, H, dh), vr.grad) < tol
    assert relerr(dpk.view(-1, H, dh), pkr.grad) < tol
    assert relerr(dpq.view(-1, H, dh), pqr.grad) < tol
    db_ref = torch.cat([t.grad.sum((0, 1)).reshape(-1) for t in (qr, kr, vr)]) + 1.0
    assert relerr(dbias, db_ref) < tol


@pytest.mark.parametrize('B,N,H,dh,w', LAI_CASES)
def test_lai_plain_kernels(B, N, H, dh, w):
    """The same kernels without relative tables (AttentionLAI of the timm LAI ViT, reference
    timm_classes/timm_locality_aware_transformer.py:25-79): forward, lse and dq / dk / dv / bias-gradient sums."""
    from vit_amd import ops
    dtype = torch.bfloat16
    D = H * dh
    qkv, _, _ = make(B, N, H, dh, dtype, False, seed=7 * N + w)
    q, k, v = views(qkv, B, N, H, dh)
    scale = dh ** -0.5
    out, lse = ops.attn_fwd(q, k, v, B, H, N, dh, qkv.stride(0), None, None, D, scale, w)
    f = lambda t: t.double().reshape(B, N, H, dh).clone().requires_grad_(True)
    qr, kr, vr = f(q), f(k), f(v)
    ref, lse_ref, _ = attention_ref(qr, kr, vr, None, None, scale, w)
    assert relerr(out.view(B, N, H, dh), ref.detach()) < 1.5e-2
    assert relerr(lse, lse_ref.detach()) < 2e-3
    g = torch.Generator(device='cuda').manual_seed(13)
    dout = torch.randn(B * N, D, device='cuda', generator=g).to(dtype)
    dqkv = torch.full_like(qkv, float('nan'))
    dq, dk, dv = views(dqkv, B, N, H, dh)
    dbias = torch.ones(3 * D, device='cuda')
    ops.attn_bwd(q, k, v, B, H, N, dh, qkv.stride(0), None, None, D, scale, w, out, dout, lse,
                 dq, dk, dv, dqkv.stride(0), None, None, dbias=dbias)
    (ref * dout.double().view(B, N, H, dh)).sum().backward()
    tol = 2e-2
    assert relerr(dq.reshape(B, N, H, dh), qr.grad) < tol
    assert relerr(dk.reshape(B, N, H, dh), kr.grad) < tol
    assert relerr(dv.reshape(B, N, H, dh), vr.grad) < tol
    db_ref = torch.cat([t.grad.sum((0, 1)).reshape(-1) for t in (qr, kr, vr)]) + 1.0
    assert relerr(dbias, db_ref) < tol


@pytest.mark.parametrize('bias', [True, False], ids=['dis', 'plain'])
@pytest.mark.parametrize('B,N,H,dh', [(2, 197, 2, 64), (1, 201, 3, 64), (3, 130, 2, 64), (20, 150, 16, 64), (128, 197, 16, 64)])
def test_full_forward(B, N, H, dh, bias):
    """Unmasked forward on the per-head kernel (whole score rows in registers; with tables the position terms summed into
    one bf16 array per head): output and lse against fp64 (the last case is BASELINE.json's full shape)."""
    from vit_amd import ops
    dtype = torch.bfloat16
    D = H * dh
    qkv, pk, pq = make(B, N, H, dh, dtype, bias, seed=3 * N + 1)
    q, k, v = views(qkv, B, N, H, dh)
    scale = (dh * (3 if bias else 1)) ** -0.5
    out, lse = ops.attn_fwd(q, k, v, B, H, N, dh, qkv.stride(0), pk, pq, D, scale, -1)
    f = lambda t: t.double().reshape(B, N, H, dh)
    ref, lse_ref, _ = attention_ref(f(q), f(k), f(v), pk.double().view(-1, H, dh) if bias else None,
                                    pq.double().view(-1, H, dh) if bias else None, scale, -1)
    assert relerr(out.view(B, N, H, dh), ref) < 1.5e-2
    assert relerr(lse, lse_ref) < 2e-3
    out2, lse2 = ops.attn_fwd(q, k, v, B, H, N, dh, qkv.stride(0), pk, pq, D, scale, N - 1)     # a band that masks nothing
    assert torch.equal(out, out2) and torch.equal(lse, lse2)


@pytest.mark.parametrize('bias', [True, False], ids=['dis', 'plain'])
@pytest.mark.parametrize('B,N,H,dh', [(2, 197, 2, 64), (1, 200, 3, 64), (1, 201, 3, 64), (3, 130, 2, 64), (2, 144, 2, 64), (20, 150, 16, 64),
                                      (128, 197, 16, 64)])        # (201 tokens: forward on the new kernel, backward on attention_band.hip)
def test_full_backward(B, N, H, dh, bias):
    """... and its backward (Bias / P / dS in one LDS array; table gradients from the dS tiles it leaves): dq, dk, dv, both
    table gradients and the q/k/v bias sums against fp64 autograd."""
    from vit_amd import ops
    dtype = torch.bfloat16
    D = H * dh
    qkv, pk, pq = make(B, N, H, dh, dtype, bias, seed=3 * N + 2)
    q, k, v = views(qkv, B, N, H, dh)
    scale = (dh * (3 if bias else 1)) ** -0.5
    out, lse = ops.attn_fwd(q, k, v, B, H, N, dh, qkv.stride(0), pk, pq, D, scale, -1)
    f = lambda t: t.double().reshape(B, N, H, dh).clone().requires_grad_(True)
    qr, kr, vr = f(q), f(k), f(v)
    pkr = pk.double().view(-1, H, dh).clone().requires_grad_(True) if bias else None
    pqr = pq.double().view(-1, H, dh).clone().requires_grad_(True) if bias else None
    ref, _, _ = attention_ref(qr, kr, vr, pkr, pqr, scale, -1)
    g = torch.Generator(device='cuda').manual_seed(17)
    dout = torch.randn(B * N, D, device='cuda', generator=g).to(dtype)
    dqkv = torch.full_like(qkv, float('nan'))
    dq, dk, dv = views(dqkv, B, N, H, dh)
    dpk = torch.zeros(2 * N - 1, D, device='cuda') if bias else None
    dpq = torch.zeros(2 * N - 1, D, device='cuda') if bias else None
    dbias = torch.ones(3 * D, device='cuda')
    ops.attn_bwd(q, k, v, B, H, N, dh, qkv.stride(0), pk, pq, D, scale, -1, out, dout, lse,
                 dq, dk, dv, dqkv.stride(0), dpk, dpq, dbias=dbias)
    (ref * dout.double().view(B, N, H, dh)).sum().backward()
    tol = 2e-2
    assert relerr(dq.reshape(B, N, H, dh), qr.grad) < tol
    assert relerr(dk.reshape(B, N, H, dh), kr.grad) < tol
    assert relerr(dv.reshape(B, N, H, dh), vr.grad) < tol
    if bias:
        assert relerr(dpk.view(-1, H, dh), pkr.grad) < tol
        assert relerr(dpq.view(-1, H, dh), pqr.grad) < tol
    db_ref = torch.cat([t.grad.sum((0, 1)).reshape(-1) for t in (qr, kr, vr)]) + 1.0
    assert relerr(dbias, db_ref) < tol
