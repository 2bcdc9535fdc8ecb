"""The validation pass of the reference (train.py:556-579: model.eval(), torch.no_grad(), model(data[, epoch])) on
modules built WITH dropout, as train.py builds them (--dropout 0.1): in eval mode dropout is the identity, so the
kernels' p = 0 arithmetic is exact there; in training mode with p > 0 the dropout sites are live (tests/test_gpu_dropout.py
checks them against the oracle): the output must differ from the p = 0 one and change from call to call."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _build(variant, dropout):
    from vit_amd import model as M
    from vit_amd.disentangled_transformer.disentangled_transformer import DisentangledTransformer
    from vit_amd.precision import set_compute_dtype
    torch.manual_seed(0)
    if variant == 'lai':
        from vit_amd.timm_classes.timm_locality_aware_transformer import VisionTransformerLAI
        net = VisionTransformerLAI(img_size=32, patch_size=4, num_classes=10, embed_dim=128, depth=2, num_heads=4,
                                   mlp_ratio=2.0, qkv_bias=True, drop_rate=dropout, attn_drop_rate=dropout,
                                   mask_epochs=[10, 20, 30, 40]).cuda()
        return set_compute_dtype(net, torch.float32)
    if variant == 'regular':
        tr = M.Transformer(d_model=128, dropout=dropout, nhead=4, dim_feedforward=256, num_encoder_layers=2,
                           normalize_before=True)
    else:
        tr = DisentangledTransformer(d_model=128, dropout=dropout, nhead=4, dim_feedforward=256, num_encoder_layers=2,
                                     normalize_before=True, pos_att_type='c2p|p2c', max_position_embeddings=128,
                                     relative_attention=True, lai=variant == 'dislai',
                                     mask_epochs=[10, 20, 30, 40] if variant == 'dislai' else None,
                                     performers=variant == 'disperf', norf=16)
    net = M.VisionTransformer(img_size=32, patch_size=4, in_chans=3, num_classes=10, transformer=tr).cuda()
    if variant == 'disperf':
        for i, lay in enumerate(tr.encoder.layers):
            lay.dis_attn.w = (torch.randn(16, 32, generator=torch.Generator().manual_seed(i)) * 0.5).cuda()
        with torch.no_grad():
            net.pos_embed.normal_(0, 0.5, generator=torch.Generator(device='cuda').manual_seed(7))
    return set_compute_dtype(net, torch.float32)


@pytest.mark.parametrize('variant', ['regular', 'dis', 'dislai', 'disperf', 'lai'])
def test_eval_no_grad_with_dropout_modules(variant):
    img = torch.randn(4, 3, 32, 32, device='cuda', generator=torch.Generator(device='cuda').manual_seed(1))
    args = (img, 30) if variant in ('dislai', 'lai') else (img,)
    ref = _build(variant, 0.0)
    want = ref(*args)                                   # training mode, p = 0, autograd on
    net = _build(variant, 0.1)
    net.load_state_dict(ref.state_dict())
    with torch.no_grad():
        t1, t2 = net(*args), net(*args)                 # training mode with p > 0: dropout is applied, not ignored
    assert not torch.equal(t1, want.detach()) and not torch.equal(t1, t2)
    assert torch.isfinite(t1).all()
    net.eval()
    with torch.no_grad():
        got = net(*args)
    assert not got.requires_grad and got.shape == (4, 10)
    assert torch.equal(got, want.detach())              # every forward kernel reduces in a fixed order
    with torch.no_grad():                               # batch of one: the reference squeezes it (model.py:62)
        one = net(*((img[:1],) + args[1:]))
    assert one.shape == ((1, 10) if variant == 'lai' else (10,))       # the timm shell keeps the batch dim
    one = one.reshape(10)
    if variant != 'disperf':      # FAVOR+'s key stabiliser is the max over the WHOLE batch (dis :736): its outputs depend
        assert (one - want[0].detach()).abs().max().item() <= 1e-5 * want.abs().max().item()     # on the batch, upstream too
