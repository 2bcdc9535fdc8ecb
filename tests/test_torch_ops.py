"""torch.library registration (vit_amd/torch_ops.py): dispatcher-visible ops with fake functions -- CPU part."""
import torch


def test_ops_are_registered_with_schemas_and_fake_functions():
    import vit_amd.torch_ops  # noqa: F401
    from torch._subclasses.fake_tensor import FakeTensorMode
    names = ['linear', 'linear_backward', 'layer_norm', 'layer_norm_backward', 'attention', 'attention_backward',
             'label_smoothing_ce', 'label_smoothing_ce_backward', 'allreduce_bucket']
    for n in names:
        assert hasattr(torch.ops.vit_amd, n), n
    assert 'Tensor? tables' in str(torch.ops.vit_amd.attention.default._schema)
    assert 'Tensor(a0!) flat' in str(torch.ops.vit_amd.allreduce_bucket.default._schema)
    B, N, H, D = 2, 17, 4, 64
    with FakeTensorMode():
        x = torch.empty(B * N, D, device='cuda', dtype=torch.bfloat16)
        w = torch.empty(3 * D, D, device='cuda', dtype=torch.bfloat16)
        b = torch.empty(3 * D, device='cuda')
        y, mean, rstd = torch.ops.vit_amd.layer_norm(x, torch.empty(D, device='cuda'), torch.empty(D, device='cuda'), 1e-6)
        qkv = torch.ops.vit_amd.linear(y, w, b, 'none')
        tab = torch.empty(2 * N - 1, 2 * D, device='cuda', dtype=torch.bfloat16)
        out, lse = torch.ops.vit_amd.attention(qkv, tab, B, N, H, 0.125, 8)
        dqkv, dtab = torch.ops.vit_amd.attention_backward(out, qkv, tab, out, lse, B, N, H, 0.125, 8)
        loss, rowlse = torch.ops.vit_amd.label_smoothing_ce(torch.empty(B, 10, device='cuda'), torch.empty(B, device='cuda', dtype=torch.int64), 0.1)
    assert qkv.shape == (B * N, 3 * D) and qkv.dtype == torch.bfloat16
    assert out.shape == (B * N, D) and lse.shape == (B, H, N) and lse.dtype == torch.float32
    assert dqkv.shape == qkv.shape and dtab.shape == tab.shape and mean.shape == (B * N,)
    assert loss.shape == () and rowlse.shape == (B,)


def test_allreduce_bucket_single_process_scales_in_place():
    import vit_amd.torch_ops  # noqa: F401
    flat = torch.arange(8, dtype=torch.float32)
    torch.ops.vit_amd.allreduce_bucket(flat, 0.5)
    assert torch.equal(flat, torch.arange(8, dtype=torch.float32) * 0.5)
