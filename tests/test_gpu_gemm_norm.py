"""GPU parity of the GEMM / LayerNorm / reduction kernels against plain fp32 torch math."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def relerr(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return ((a - b).abs().max() / (b.abs().max() + 1e-30)).item()


def _ops():
    from vit_amd import ops
    return ops


SHAPES = [(128, 128, 64), (256, 384, 128), (197, 200, 72), (393, 320, 96), (25, 10, 256), (130, 1000, 40),
          (1000, 136, 520)]


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16], ids=['f32', 'bf16'])
@pytest.mark.parametrize('ta,tb', [(False, True), (False, False), (True, False), (True, True)])
@pytest.mark.parametrize('M,N,K', SHAPES)
def test_gemm_layouts(dtype, ta, tb, M, N, K):
    ops = _ops()
    torch.manual_seed(M * 7 + N * 3 + K)
    a = torch.randn((K, M) if ta else (M, K), device='cuda').to(dtype)
    b = torch.randn((N, K) if tb else (K, N), device='cuda').to(dtype)
    ref = (a.float().t() if ta else a.float()).double() @ (b.float().t() if tb else b.float()).double()
    out = ops.gemm(a, b, trans_a=ta, trans_b=tb, out_dtype=torch.float32)
    tol = 2e-6 if dtype == torch.float32 else 1e-5   # inputs are exactly representable; fp32 accumulate
    assert relerr(out, ref) < tol
    if dtype == torch.bfloat16:
        outb = ops.gemm(a, b, trans_a=ta, trans_b=tb)
        assert outb.dtype == torch.bfloat16 and relerr(outb, ref) < 1e-2


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16], ids=['f32', 'bf16'])
def test_gemm_epilogues(dtype):
    ops = _ops()
    torch.manual_seed(3)
    M, N, K = 300, 264, 136
    a = torch.randn(M, K, device='cuda').to(dtype)
    w = (torch.randn(N, K, device='cuda') / K ** 0.5).to(dtype)
    bias = torch.randn(N, device='cuda')
    res = torch.randn(M, N, device='cuda').to(dtype)
    lin = a.float() @ w.float().t() + bias
    tol = 1e-5 if dtype == torch.float32 else 1.5e-2
    # bias + residual
    assert relerr(ops.gemm(a, w, bias=bias, residual=res), lin + res.float()) < tol
    # gelu with pre-activation side output
    pre = torch.empty(M, N, device='cuda', dtype=dtype)
    y = ops.gemm(a, w, bias=bias, act=ops.ACT_GELU, aux_out=pre)
    assert relerr(pre, lin) < tol
    assert relerr(y, torch.nn.functional.gelu(lin)) < tol
    # relu
    assert relerr(ops.gemm(a, w, bias=bias, act=ops.ACT_RELU), torch.relu(lin)) < tol
    # dgelu / drelu: out = (a @ w^T) * act'(aux)
    h = torch.randn(M, N, device='cuda').to(dtype)
    hf = h.float().requires_grad_(True)
    torch.nn.functional.gelu(hf).sum().backward()
    raw = a.float() @ w.float().t()
    assert relerr(ops.gemm(a, w, act=ops.ACT_DGELU, aux_in=h), raw * hf.grad) < tol
    assert relerr(ops.gemm(a, w, act=ops.ACT_DRELU, aux_in=h), raw * (h.float() > 0)) < tol
    # fp32 accumulate + split-K
    acc0 = torch.randn(M, N, device='cuda')
    acc = acc0.clone()
    ops.gemm(a, w, out=acc, accumulate=True, splitk=3)
    assert relerr(acc, acc0 + raw) < 1e-5
    fresh = ops.gemm(a, w, out_dtype=torch.float32, splitk=2)
    assert relerr(fresh, raw) < 1e-5
    # periodic residual + grouped output rows (patch rows into a token matrix)
    per, groups = 60, 5
    pos = torch.randn(per + 1, N, device='cuda').to(dtype)
    tok = torch.zeros(groups * (per + 1), N, device='cuda', dtype=dtype)
    ops.gemm(a, w, bias=bias, residual=pos, res_period=per, res_off=1, out=tok, grp=(per, per + 1, 1))
    want = torch.zeros(groups, per + 1, N, device='cuda')
    want[:, 1:] = lin.view(groups, per, N) + pos.float()[1:]
    assert relerr(tok.float().view(groups, per + 1, N), want) < tol
    assert tok.view(groups, per + 1, N)[:, 0].abs().max().item() == 0.0


def test_gemm_large_bf16_and_strided():
    ops = _ops()
    torch.manual_seed(5)
    M, N, K = 1576, 3072, 1024
    a = torch.randn(M, K, device='cuda').bfloat16()
    w = (torch.randn(N, K, device='cuda') / 32).bfloat16()
    ref = a.float() @ w.float().t()
    assert relerr(ops.gemm(a, w), ref) < 1e-2
    # strided A (a column slice of a wider buffer) and dW-shaped TN product with auto split-K
    wide = torch.randn(M, 3 * K, device='cuda').bfloat16()
    sl = wide[:, K:2 * K]
    assert relerr(ops.gemm(sl, w), sl.float() @ w.float().t()) < 1e-2
    dy = torch.randn(M, 512, device='cuda').bfloat16()
    dw = ops.gemm(dy, sl, trans_a=True, trans_b=False, out_dtype=torch.float32)
    assert relerr(dw, dy.float().t() @ sl.float()) < 1e-4


def test_gemm_rejects_bad_args():
    ops = _ops()
    from vit_amd._lib import VitkError
    a = torch.randn(8, 16, device='cuda')
    with pytest.raises(ValueError):
        ops.gemm(a, torch.randn(8, 24, device='cuda'))
    with pytest.raises(VitkError):
        ops.gemm(a, torch.randn(8, 16, device='cuda'), act=ops.ACT_DGELU)
    with pytest.raises(VitkError):
        ops.gemm(torch.randn(8, 16), torch.randn(8, 16))


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16], ids=['f32', 'bf16'])
@pytest.mark.parametrize('rows,D', [(37, 128), (394, 768), (1000, 1024), (130, 1280), (64, 320)])
def test_layernorm(dtype, rows, D):
    ops = _ops()
    torch.manual_seed(rows + D)
    x = (torch.randn(rows, D, device='cuda') * 2 + 0.5).to(dtype)
    g = 1 + 0.1 * torch.randn(D, device='cuda')
    b = 0.1 * torch.randn(D, device='cuda')
    dy = torch.randn(rows, D, device='cuda').to(dtype)
    add = torch.randn(rows, D, device='cuda').to(dtype)
    xf = x.float().requires_grad_(True)
    gf, bf = g.clone().requires_grad_(True), b.clone().requires_grad_(True)
    ref = torch.nn.functional.layer_norm(xf, (D,), gf, bf, 1e-5)
    (ref * dy.float()).sum().backward()
    y, mean, rstd = ops.layernorm_fwd(x, g, b, 1e-5)
    tol = 1e-5 if dtype == torch.float32 else 1.2e-2
    assert relerr(y, ref) < tol
    dg = torch.zeros(D, device='cuda')
    db = torch.zeros(D, device='cuda')
    dx = ops.layernorm_bwd(dy, x, mean, rstd, g, dg, db, add=add)
    assert relerr(dx, xf.grad + add.float()) < tol
    assert relerr(dg, gf.grad) < 1e-5 and relerr(db, bf.grad) < 1e-5
    # the same with the fused column sums of dx (bias gradient of the Linear in front of a post-LN)
    dg2, db2, dxs = torch.zeros(D, device='cuda'), torch.zeros(D, device='cuda'), torch.zeros(D, device='cuda')
    dx2 = ops.layernorm_bwd(dy, x, mean, rstd, g, dg2, db2, add=add, dxsum=dxs)
    assert torch.equal(dx2, dx)
    assert relerr(dg2, gf.grad) < 1e-5 and relerr(db2, bf.grad) < 1e-5
    assert relerr(dxs, (xf.grad + add.float()).sum(0)) < (1e-5 if dtype == torch.float32 else 2e-3)


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16], ids=['f32', 'bf16'])
@pytest.mark.parametrize('M,N', [(1000, 768), (197, 10), (4096, 3072), (33, 1000)])
def test_colsum_cast(dtype, M, N):
    ops = _ops()
    x = torch.randn(M, N, device='cuda').to(dtype)
    assert relerr(ops.colsum(x), x.float().sum(0)) < 1e-5
    if dtype == torch.float32:
        flat = x.reshape(-1)
        dst = torch.empty_like(flat, dtype=torch.bfloat16)
        ops.cast(flat, dst)
        assert torch.equal(dst, flat.bfloat16())
        back = torch.empty_like(flat)
        ops.cast(dst, back)
        assert torch.equal(back, dst.float())


def test_patchify_roundtrip():
    ops = _ops()
    B, C, H, P = 3, 3, 32, 4
    img = torch.randn(B, C, H, H, device='cuda')
    pat = ops.patchify(img, P, torch.float32)
    ref = torch.nn.functional.unfold(img, P, stride=P).transpose(1, 2).reshape(B * (H // P) ** 2, C * P * P)
    assert torch.equal(pat, ref)
    assert torch.equal(ops.unpatchify(pat, B, C, H, H, P), img)
    assert torch.equal(ops.patchify(img, P, torch.bfloat16), ref.bfloat16())


BIG_SHAPES = [(520, 1000, 192), (1032, 536, 320), (2048, 1024, 128), (776, 776, 448)]


@pytest.mark.parametrize('ta,tb', [(False, True), (False, False), (True, False), (True, True)])
@pytest.mark.parametrize('M,N,K', BIG_SHAPES)
def test_gemm256_layouts(ta, tb, M, N, K):
    """Shapes routed to the 256x256 direct-to-LDS kernel (bf16, K % 64 == 0, M*N >= 512^2)."""
    ops = _ops()
    torch.manual_seed(M + 3 * N + K)
    a = torch.randn((K, M) if ta else (M, K), device='cuda').bfloat16()
    b = torch.randn((N, K) if tb else (K, N), device='cuda').bfloat16()
    ref = (a.float().t() if ta else a.float()).double() @ (b.float().t() if tb else b.float()).double()
    out = ops.gemm(a, b, trans_a=ta, trans_b=tb, out_dtype=torch.float32)
    assert relerr(out, ref) < 1e-5
    assert relerr(ops.gemm(a, b, trans_a=ta, trans_b=tb), ref) < 1e-2
    sk = ops.gemm(a, b, trans_a=ta, trans_b=tb, out_dtype=torch.float32, splitk=2)
    assert relerr(sk, ref) < 1e-5


def test_gemm256_epilogues():
    ops = _ops()
    torch.manual_seed(11)
    M, N, K = 1000, 776, 256
    a = torch.randn(M, K, device='cuda').bfloat16()
    w = (torch.randn(N, K, device='cuda') / K ** 0.5).bfloat16()
    bias = torch.randn(N, device='cuda')
    res = torch.randn(M, N, device='cuda').bfloat16()
    lin = a.float() @ w.float().t() + bias
    tol = 1.5e-2
    assert relerr(ops.gemm(a, w, bias=bias, residual=res), lin + res.float()) < tol
    pre = torch.empty(M, N, device='cuda', dtype=torch.bfloat16)
    y = ops.gemm(a, w, bias=bias, act=ops.ACT_GELU, aux_out=pre)
    assert relerr(pre, lin) < tol and relerr(y, torch.nn.functional.gelu(lin)) < tol
    assert relerr(ops.gemm(a, w, bias=bias, act=ops.ACT_RELU), torch.relu(lin)) < tol
    h = torch.randn(M, N, device='cuda').bfloat16()
    hf = h.float().requires_grad_(True)
    torch.nn.functional.gelu(hf).sum().backward()
    raw = a.float() @ w.float().t()
    assert relerr(ops.gemm(a, w, act=ops.ACT_DGELU, aux_in=h), raw * hf.grad) < tol
    assert relerr(ops.gemm(a, w, act=ops.ACT_DRELU, aux_in=h), raw * (h.float() > 0)) < tol
    acc0 = torch.randn(M, N, device='cuda')
    acc = acc0.clone()
    ops.gemm(a, w, out=acc, accumulate=True, splitk=2)
    assert relerr(acc, acc0 + raw) < 1e-5
    # strided operands (column slices of wider buffers), as the attention backward feeds them
    wide = torch.randn(M, 3 * K, device='cuda').bfloat16()
    sl = wide[:, K:2 * K]
    assert relerr(ops.gemm(sl, w, out_dtype=torch.float32), sl.float() @ w.float().t()) < 1e-5
    # grouped rows + periodic residual (patch-embed epilogue)
    per, groups = 200, 5
    pos = torch.randn(per + 1, N, device='cuda').bfloat16()
    tok = torch.zeros(groups * (per + 1), N, device='cuda', dtype=torch.bfloat16)
    ops.gemm(a, w, bias=bias, residual=pos, res_period=per, res_off=1, out=tok, grp=(per, per + 1, 1))
    want = torch.zeros(groups, per + 1, N, device='cuda')
    want[:, 1:] = lin.view(groups, per, N) + pos.float()[1:]
    assert relerr(tok.float().view(groups, per + 1, N), want) < tol


@pytest.mark.parametrize('ta,tb', [(False, True), (False, False), (True, False)])
@pytest.mark.parametrize('M,N,K', [(4360, 4096, 192), (25216, 1024, 128), (25664, 3072, 128), (25664 + 16, 1024, 192)])
def test_gemm256_persistent_items(ta, tb, M, N, K):
    """More tiles than CUs: a workgroup of the persistent kernel walks several work items and the
    K-tile pipeline crosses item boundaries (ragged last row of tiles included)."""
    ops = _ops()
    torch.manual_seed(M + N + K)
    a = torch.randn((K, M) if ta else (M, K), device='cuda').bfloat16()
    b = torch.randn((N, K) if tb else (K, N), device='cuda').bfloat16()
    A = a.float().t() if ta else a.float()
    Bm = b.float().t() if tb else b.float()
    ref = A @ Bm                                  # fp32 products of bf16 values, fp32 accumulation
    out = ops.gemm(a, b, trans_a=ta, trans_b=tb, out_dtype=torch.float32)
    assert relerr(out, ref) < 1e-5
    if not ta:
        bias = torch.randn(N, device='cuda')
        res = torch.randn(M, N, device='cuda').bfloat16()
        y = ops.gemm(a, b, trans_b=tb, bias=bias, residual=res)
        assert relerr(y, ref + bias + res.float()) < 1.5e-2
        pre = torch.empty(M, N, device='cuda', dtype=torch.bfloat16)
        y = ops.gemm(a, b, trans_b=tb, bias=bias, act=ops.ACT_GELU, aux_out=pre)
        assert relerr(pre, ref + bias) < 1.5e-2
        assert relerr(y, torch.nn.functional.gelu(ref + bias)) < 1.5e-2
        # every element, not only the max-norm: a misplaced tile would pass a max-norm check on noise-like data
        assert (y.float() - torch.nn.functional.gelu(ref + bias)).abs().max().item() < 0.25
        # side-operand epilogue (the hand-counted loads): relu gate on a saved activation + plain store
        gate = torch.randn(M, N, device='cuda').bfloat16()
        y = ops.gemm(a, b, trans_b=tb, act=ops.ACT_DRELU, aux_in=gate)
        want = torch.where(gate.float() > 0, ref, torch.zeros_like(ref))
        assert (y.float() - want).abs().max().item() < 0.25 and relerr(y, want) < 1.5e-2


@pytest.mark.parametrize('splitk', [2, 4, 16])
def test_dw_gemm_explicit_splitk(splitk):
    """The weight-gradient GEMM with the K splits chosen by the caller (VITK_DW_SPLITK / bench.py --max-overlap) instead
    of the library: same sum whatever the split, accumulated into a zeroed fp32 buffer as the fused blocks do."""
    ops = _ops()
    torch.manual_seed(splitk)
    rows = 25216                                   # the headline's token rows (128 x 197): proj dW, 4 x 4 tiles
    dy = (torch.randn(rows, 1024, device='cuda') * 0.1).bfloat16()
    x = torch.randn(rows, 1024, device='cuda').bfloat16()
    ref = dy.float().t() @ x.float()
    out = torch.zeros(1024, 1024, device='cuda')
    ops.gemm(dy, x, trans_a=True, trans_b=False, out=out, accumulate=True, splitk=splitk)
    assert relerr(out, ref) < 1e-5
    auto = torch.zeros(1024, 1024, device='cuda')
    ops.gemm(dy, x, trans_a=True, trans_b=False, out=auto, accumulate=True)
    assert relerr(out, auto) < 1e-5
