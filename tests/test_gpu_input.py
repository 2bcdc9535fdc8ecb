"""uint8 images through the patch embedding (vitk_patchify_u8): ToTensor + Normalize folded into the
gather must equal torchvision's arithmetic (x / 255, then (x - mean) / std; train.py:340-341,362-363) bit for bit in
fp32, and a model fed uint8 pixels must equal the same model fed the normalised floats."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _normalise_like_torchvision(u8_nchw, mean, std):
    x = u8_nchw.to(torch.float32).div(255)                       # ToTensor
    m = torch.tensor(mean, dtype=torch.float32).view(1, -1, 1, 1)
    s = torch.tensor(std, dtype=torch.float32).view(1, -1, 1, 1)
    return x.sub(m).div(s)                                       # Normalize


@pytest.mark.parametrize('layout', ['nchw', 'nhwc'])
@pytest.mark.parametrize('B,C,H,W,P', [(3, 3, 32, 32, 4), (2, 3, 224, 224, 16), (5, 1, 28, 28, 7), (1, 4, 16, 48, 8)])
def test_patchify_u8_bit_exact(B, C, H, W, P, layout):
    from vit_amd import ops
    from vit_amd.data import IMAGENET_NORM
    mean, std = (IMAGENET_NORM[0] + (0.5,))[:C], (IMAGENET_NORM[1] + (0.25,))[:C]
    u8 = torch.randint(0, 256, (B, C, H, W), dtype=torch.uint8, generator=torch.Generator().manual_seed(B * H + P))
    ref_img = _normalise_like_torchvision(u8, mean, std)
    ref = ref_img.view(B, C, H // P, P, W // P, P).permute(0, 2, 4, 1, 3, 5).reshape(B * (H // P) * (W // P), C * P * P)
    src = (u8.permute(0, 2, 3, 1).contiguous() if layout == 'nhwc' else u8).cuda()
    out = ops.patchify_u8(src, P, torch.float32, mean, std, channels_last=layout == 'nhwc')
    assert torch.equal(out.cpu(), ref)
    # and the existing float gather on the normalised image gives the same patches
    assert torch.equal(ops.patchify(ref_img.cuda(), P, torch.float32).cpu(), ref)
    ob = ops.patchify_u8(src, P, torch.bfloat16, mean, std, channels_last=layout == 'nhwc')
    assert torch.equal(ob.cpu(), ref.to(torch.bfloat16))


@pytest.mark.parametrize('variant', ['disentangled_lai_attention', 'transformer_timm_lai'])
def test_model_on_uint8_equals_model_on_normalised_floats(variant):
    from vit_amd.data import CIFAR10_NORM, set_input_norm
    from vit_amd.train import build_model, label_smoothing_ce
    torch.manual_seed(0)
    net = build_model(variant, 'cifar', img_size=32, patch_size=4, num_classes=10, mask_epochs=[10, 20, 30, 40],
                      dtype=torch.float32, depth=2)
    u8 = torch.randint(0, 256, (4, 3, 32, 32), dtype=torch.uint8, generator=torch.Generator().manual_seed(3))
    tgt = torch.tensor([1, 2, 3, 4], device='cuda')
    xf = _normalise_like_torchvision(u8, *CIFAR10_NORM).cuda()
    la = label_smoothing_ce(net(xf, 0), tgt)
    la.backward()
    ga = {n: p.grad.clone() for n, p in net.named_parameters() if p.grad is not None}
    net.zero_grad(set_to_none=True)
    with pytest.raises(ValueError):
        net(u8.cuda(), 0)                                  # no normalisation configured
    set_input_norm(net, *CIFAR10_NORM, layout='nhwc')
    lb = label_smoothing_ce(net(u8.permute(0, 2, 3, 1).contiguous().cuda(), 0), tgt)
    lb.backward()
    assert abs(la.item() - lb.item()) <= 1e-6 * abs(la.item())
    for n, p in net.named_parameters():
        if p.grad is not None:
            err = (p.grad - ga[n]).abs().max().item() / (ga[n].abs().max().item() + 1e-30)
            assert err < 1e-5, (n, err)                    # split-K atomics reorder sums; inputs are identical


def test_patchify_u8_rejects():
    from vit_amd import ops
    from vit_amd._lib import VitkError
    from vit_amd.data import set_input_norm
    u8 = torch.zeros(1, 3, 32, 32, dtype=torch.uint8, device='cuda')
    with pytest.raises(ValueError):
        ops.patchify_u8(u8, 4, torch.float32, (0.5,), (0.5,))              # channel count
    with pytest.raises(ValueError):
        ops.patchify_u8(u8.float(), 4, torch.float32, (0.5,) * 3, (0.5,) * 3)
    with pytest.raises(VitkError):
        ops.patchify_u8(u8, 5, torch.float32, (0.5,) * 3, (0.5,) * 3)      # 32 not divisible by 5
    with pytest.raises(VitkError):
        ops.patchify_u8(u8, 4, torch.float32, (0.5,) * 3, (0.5, 0.0, 0.5))  # zero std
    with pytest.raises(VitkError):
        ops.patchify_u8(u8.cpu(), 4, torch.float32, (0.5,) * 3, (0.5,) * 3)
    with pytest.raises(ValueError):
        set_input_norm(torch.nn.Identity(), (0.5,), (0.5,), layout='chw')
