"""torch.ops.vit_amd.* on the GPU: same numbers as the autograd.Function layer, opcheck, and a traced (aot_eager) step."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _inputs(dtype, bias):
    B, N, H, D = 2, 65, 4, 128
    g = torch.Generator(device='cuda').manual_seed(3)
    x = torch.randn(B * N, D, device='cuda', generator=g).to(dtype)
    w = (torch.randn(3 * D, D, device='cuda', generator=g) / D ** 0.5).to(dtype)
    b = torch.randn(3 * D, device='cuda', generator=g)
    tab = (torch.randn(2 * N - 1, 2 * D, device='cuda', generator=g) * 0.5).to(dtype) if bias else None
    return B, N, H, D, x, w, b, tab


def _step(x, w, b, tab, gamma, beta, B, N, H, band):
    y, _, _ = torch.ops.vit_amd.layer_norm(x, gamma, beta, 1e-6)
    qkv = torch.ops.vit_amd.linear(y, w, b, 'none')
    D = x.shape[1]
    out, _ = torch.ops.vit_amd.attention(qkv, tab, B, N, H, (D // H * (3 if tab is not None else 1)) ** -0.5, band)
    return out


@pytest.mark.parametrize('bias', [False, True], ids=['plain', 'dis'])
@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16], ids=['f32', 'bf16'])
def test_ops_match_functional_layer(dtype, bias):
    import vit_amd.torch_ops  # noqa: F401
    from vit_amd import functional as VF
    B, N, H, D, x, w, b, tab = _inputs(dtype, bias)
    gamma = torch.rand(D, device='cuda') + 0.5
    beta = torch.randn(D, device='cuda') * 0.1
    leaves = [t.clone().requires_grad_(True) for t in (x, w, b, gamma, beta)] + ([tab.clone().requires_grad_(True)] if bias else [])
    xa, wa, ba, ga, be = leaves[:5]
    ta = leaves[5] if bias else None
    out = _step(xa, wa, ba, ta, ga, be, B, N, H, 8)
    gout = torch.randn_like(out)
    (out.float() * gout.float()).sum().backward()
    # the same chain through vit_amd.functional
    ref_leaves = [t.clone().requires_grad_(True) for t in (x, w, b, gamma, beta)] + ([tab.clone().requires_grad_(True)] if bias else [])
    xr, wr, br, gr, ber = ref_leaves[:5]
    tr = ref_leaves[5] if bias else None
    y = VF.layer_norm(xr, gr, ber, 1e-6)
    qkv = VF.linear(y, wr, br)
    ref = VF.attention(qkv, tr, B, N, H, D // H, (D // H * (3 if bias else 1)) ** -0.5, 8)
    (ref.float() * gout.float()).sum().backward()
    tol = 1e-5 if dtype == torch.float32 else 2e-2
    def rel(a, b_):
        return ((a.double() - b_.double()).abs().max() / (b_.double().abs().max() + 1e-30)).item()
    assert rel(out, ref) < tol
    for a, r in zip(leaves, ref_leaves):
        assert rel(a.grad, r.grad) < tol * 2


def test_opcheck_and_traced_step():
    import vit_amd.torch_ops  # noqa: F401
    B, N, H, D, x, w, b, tab = _inputs(torch.float32, True)
    gamma = torch.ones(D, device='cuda')
    beta = torch.zeros(D, device='cuda')
    qkv = torch.ops.vit_amd.linear(x, w, b, 'none')
    torch.library.opcheck(torch.ops.vit_amd.linear.default, (x.requires_grad_(True), w.requires_grad_(True), b.requires_grad_(True), 'none'))
    torch.library.opcheck(torch.ops.vit_amd.attention.default, (qkv.detach().requires_grad_(True), tab.requires_grad_(True), B, N, H, 0.1, 8))
    torch.library.opcheck(torch.ops.vit_amd.layer_norm.default, (x, gamma.requires_grad_(True), beta.requires_grad_(True), 1e-6))
    logits = torch.randn(8, 10, device='cuda', requires_grad=True)
    target = torch.randint(0, 10, (8,), device='cuda')
    torch.library.opcheck(torch.ops.vit_amd.label_smoothing_ce.default, (logits, target, 0.1))
    # the whole chain under torch.compile's tracer (aot_eager: dynamo + AOT autograd, no code generation)
    fn = torch.compile(lambda *a: _step(*a, B, N, H, 8), backend='aot_eager', fullgraph=True)
    args = [t.detach().clone().requires_grad_(True) for t in (x, w, b, tab, gamma, beta)]
    eager = [t.detach().clone().requires_grad_(True) for t in (x, w, b, tab, gamma, beta)]
    o1 = fn(*args)
    o2 = _step(*eager, B, N, H, 8)
    assert torch.allclose(o1, o2, atol=1e-6)
    o1.sum().backward()
    o2.sum().backward()
    for a, e in zip(args, eager):
        assert torch.allclose(a.grad, e.grad, atol=1e-5, rtol=1e-4)
