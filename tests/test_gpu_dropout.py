"""Dropout p > 0 (the reference trains with --dropout 0.1, train.py:79): the kernels' counter-based masks.

Parity at p > 0 cannot be bit-for-bit with torch's Philox masks, so it is defined with the mask the kernels applied
(ops.dropout_mask regenerates it from (p, seed, offset)) handed to the CPU oracle, which restates every nn.Dropout site
of the reference as x * keep / (1 - p) in the reference's order.  Bars as elsewhere: fp32 1e-5 (whole models too),
bf16 2e-2 on forward values."""
import pytest
import torch

import golden_inputs as gi

pytestmark = pytest.mark.gpu


def relerr(a, b):
    a, b = a.detach().float().cpu(), b.detach().float().cpu()
    return ((a - b).abs().max() / b.abs().max().clamp_min(1e-30)).item()


class Feeder:
    """DROPOUT provider for the oracle: pops the streams the GPU run recorded, in order."""

    def __init__(self, streams):
        self.streams = list(streams)
        self.i = 0

    def __call__(self, p, shape):
        from vit_amd import ops
        assert self.i < len(self.streams), 'the oracle reached more dropout sites than the kernels drew'
        st = self.streams[self.i]
        self.i += 1
        assert abs(st[0] - p) < 1e-12, (st, p)
        n = 1
        for d in shape:
            n *= d
        return ops.dropout_mask(n, st).view(*shape).cpu().float()


def test_elementwise_dropout_statistics_and_backward():
    from vit_amd import ops, functional as VF, dropout as VD
    VD.manual_seed(123)
    x = torch.randn(4096, 1024, device='cuda')
    for p in (0.1, 0.5):
        xg = x.clone().requires_grad_(True)
        with VD.record() as streams:
            y = VF.dropout(xg, p)
        keep = ops.dropout_mask(x.numel(), streams[0]).view_as(x).bool()
        frac = keep.float().mean().item()
        assert abs(frac - (1 - p)) < 3e-3, frac                       # 4M draws: sigma ~ 2e-4
        assert torch.equal(y.detach(), torch.where(keep, x / (1 - p), torch.zeros_like(x)).to(y.dtype)) or \
            relerr(y, torch.where(keep, x / (1 - p), torch.zeros_like(x))) < 1e-6
        # rows and columns are not correlated with the index pattern
        assert (keep.float().mean(0) - (1 - p)).abs().max().item() < 0.05
        assert (keep.float().mean(1) - (1 - p)).abs().max().item() < 0.08
        g = torch.randn_like(x)
        y.backward(g)
        assert relerr(xg.grad, torch.where(keep, g / (1 - p), torch.zeros_like(g))) < 1e-6
    # two sites never share a stream; the same seed reproduces the run
    VD.manual_seed(7)
    a = VF.dropout(x, 0.3)
    b = VF.dropout(x, 0.3)
    VD.manual_seed(7)
    a2 = VF.dropout(x, 0.3)
    assert torch.equal(a, a2) and not torch.equal(a, b)
    # bf16 with the residual folded in
    xb, rb = x.bfloat16(), torch.randn_like(x).bfloat16()
    with VD.record() as streams:
        yb = VF.dropout(xb, 0.25, res=rb)
    keep = ops.dropout_mask(x.numel(), streams[0]).view_as(x).bool()
    want = torch.where(keep, xb.float() / 0.75, torch.zeros_like(x)) + rb.float()
    assert relerr(yb, want) < 1e-2


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16], ids=['f32', 'bf16'])
@pytest.mark.parametrize('B,H,N,dh,w,bias', [(2, 4, 65, 32, 8, True), (2, 4, 197, 64, 20, True), (3, 2, 100, 64, 8, True), (20, 16, 33, 64, 12, True),
                                                  (1, 2, 197, 64, -1, False), (2, 3, 197, 64, -1, True), (3, 2, 150, 64, -1, True), (9, 16, 197, 64, -1, True),
                                            (1, 2, 300, 80, 12, True)])
def test_attention_prob_dropout_vs_closed_form(B, H, N, dh, w, bias, dtype):
    """Forward and every gradient of the fused attention with dropout on the probabilities, against fp64 math with the
    exported mask: whole-head kernels (N <= 208, dh <= 64, bf16) and the generic ones (fp32, dh 80, N 300)."""
    from vit_amd import ops, functional as VF, dropout as VD
    torch.manual_seed(N + dh)
    D = H * dh
    p = 0.2
    qkv = (torch.randn(B * N, 3 * D, device='cuda') * 0.5).to(dtype).requires_grad_(True)
    tables = (torch.randn(2 * N - 1, 2 * D, device='cuda') * 0.5).to(dtype).requires_grad_(True) if bias else None
    scale = 1.0 / (dh * (3 if bias else 1)) ** 0.5
    VD.manual_seed(99)
    with VD.record() as streams:
        out = VF.attention(qkv, tables, B, N, H, dh, scale, None if w < 0 else w, p_drop=p)
    g = torch.randn_like(out)
    out.backward(g)
    keep = ops.dropout_mask(B * H * N * N, streams[0]).view(B, H, N, N).double()

    q64 = qkv.detach().double().requires_grad_(True)
    t64 = tables.detach().double().requires_grad_(True) if bias else None
    q, k, v = (q64[:, i * D:(i + 1) * D].view(B, N, H, dh).transpose(1, 2) for i in range(3))
    s = q @ k.transpose(-1, -2)
    if bias:
        idx = (torch.arange(N, device='cuda')[:, None] - torch.arange(N, device='cuda')[None, :]) + (N - 1)
        pq = t64[:, :D].view(-1, H, dh).transpose(0, 1)         # [H, R, dh]
        pk = t64[:, D:].view(-1, H, dh).transpose(0, 1)
        c2p = torch.gather(q @ pk.transpose(-1, -2), -1, idx.expand(B, H, N, N))
        p2c = torch.gather(k @ pq.transpose(-1, -2), -1, idx.t().expand(B, H, N, N)).transpose(-1, -2)
        s = s + c2p + p2c
    s = s * scale
    if w >= 0:
        i = torch.arange(N, device='cuda')
        s = s.masked_fill((i[:, None] - i[None, :]).abs() > w, -1e9)
    pr = torch.softmax(s, -1) * keep / (1 - p)
    ref = (pr @ v).transpose(1, 2).reshape(B * N, D)
    ref.backward(g.double())
    tol = 1e-5 if dtype == torch.float32 else 2e-2
    assert relerr(out, ref) < tol
    assert relerr(qkv.grad, q64.grad) < (1e-5 if dtype == torch.float32 else 3e-2)
    if bias:
        assert relerr(tables.grad, t64.grad) < (1e-5 if dtype == torch.float32 else 3e-2)


def _dis_pair(dtype, p):
    from oracle import ref_cpu as R
    from vit_amd import model as M
    from vit_amd.disentangled_transformer.disentangled_transformer import DisentangledTransformer
    from vit_amd.precision import set_compute_dtype
    kw = dict(d_model=128, dropout=p, nhead=4, dim_feedforward=256, num_encoder_layers=2, normalize_before=True,
              pos_att_type='c2p|p2c', max_position_embeddings=128, relative_attention=True, lai=True,
              mask_epochs=[10, 20, 30, 40])
    ref = R.RefVisionTransformer(img_size=32, patch_size=4, in_chans=3, num_classes=10,
                                 transformer=R.RefDisentangledTransformer(**kw))
    sd = gi.fill_state_dict(ref.state_dict(), seed=11)
    ref.load_state_dict(sd)
    net = M.VisionTransformer(img_size=32, patch_size=4, in_chans=3, num_classes=10, transformer=DisentangledTransformer(**kw))
    net.load_state_dict(sd)
    return ref, set_compute_dtype(net.cuda(), dtype)


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16], ids=['f32', 'bf16'])
def test_dislai_model_with_dropout_vs_oracle(dtype):
    """2-layer DisLAI ViT in training mode at the reference's default --dropout 0.1: logits, loss and every parameter
    gradient against the oracle fed the kernels' masks (table rows, probabilities, attention output, block output)."""
    from oracle import ref_cpu as R
    from vit_amd import dropout as VD
    from vit_amd.train import label_smoothing_ce
    p = 0.1
    ref, net = _dis_pair(dtype, p)
    net.train()
    img = gi.randn(3, 'img', 4, 3, 32, 32)
    tgt = torch.tensor([1, 3, 5, 7])
    VD.manual_seed(5)
    with VD.record() as streams:
        out = net(img.cuda(), 30)
    assert len(streams) == 2 * 4, streams                 # 4 sites per layer
    loss = label_smoothing_ce(out, tgt.cuda())
    loss.backward()
    feeder = Feeder(streams)
    R.DROPOUT = feeder
    try:
        out_r = ref(img, 30)
        loss_r = R.label_smoothing_ce(out_r, tgt)
        loss_r.backward()
    finally:
        R.DROPOUT = None
    assert feeder.i == len(streams)
    f32 = dtype == torch.float32
    assert relerr(out, out_r) < (1e-5 if f32 else 3e-2)
    assert abs(loss.item() - loss_r.item()) < (1e-5 if f32 else 2e-2) * abs(loss_r.item())
    rp = dict(ref.named_parameters())
    for n, q in net.named_parameters():
        if rp[n].grad is None:
            assert q.grad is None, n
            continue
        assert relerr(q.grad, rp[n].grad) < (1e-5 if f32 else 8e-2), n
    # eval mode: no site draws a stream and the fused path runs (bit-identical to a p = 0 model)
    net.eval()
    with VD.record() as streams, torch.no_grad():
        net(img.cuda(), 30)
    assert streams == []


def test_p0_training_is_bit_identical_to_eval_path():
    """dropout = 0 in training mode must take the fused blocks exactly as before (no stream drawn, same bits)."""
    from vit_amd import dropout as VD
    _, net = _dis_pair(torch.bfloat16, 0.0)
    img = gi.randn(3, 'img', 4, 3, 32, 32).cuda()
    net.train()
    with VD.record() as streams:
        a = net(img, 30)
    net.eval()
    b = net(img, 30)
    assert streams == [] and torch.equal(a, b)


@pytest.mark.parametrize('variant', ['transformer', 'transformer_timm_lai', 'performer_timm'])
def test_other_stacks_with_dropout_vs_oracle(variant):
    """The regular pre-LN encoder (MHA dropout, dropout1, dropout, dropout2), the timm LAI ViT (attn_drop) and the
    FAVOR+ ViT (attn_drop on the attention output) at p = 0.1, fp32, against the oracle with the kernels' masks."""
    from oracle import ref_cpu as R
    from vit_amd import dropout as VD
    from vit_amd.train import build_model, label_smoothing_ce
    p = 0.1
    me = [10, 20, 30, 40]
    torch.manual_seed(0)
    net = build_model(variant, 'S', img_size=32, patch_size=4, num_classes=10, mask_epochs=me, depth=2, dim=128, mlp=256,
                      heads=4, dtype=torch.float32, dropout=p, num_orf=16)
    if variant == 'transformer':
        tr = R.RefTransformer(d_model=128, dropout=p, nhead=4, dim_feedforward=256, num_encoder_layers=2, normalize_before=True)
        ref = R.RefVisionTransformer(img_size=32, patch_size=4, in_chans=3, num_classes=10, transformer=tr)
        epoch = None
    else:
        ref = R.RefVisionTransformerLAI(img_size=32, patch_size=4, num_classes=10, embed_dim=128, depth=2, num_heads=4,
                                        mlp_ratio=2.0, mask_epochs=me if variant == 'transformer_timm_lai' else None,
                                        attn_drop_rate=p if variant == 'transformer_timm_lai' else 0.)
        epoch = 30
        if variant == 'performer_timm':
            class NoEpoch(torch.nn.Module):
                def __init__(self, mod):
                    super().__init__()
                    self.mod = mod

                def forward(self, x, epoch=None):
                    return self.mod(x)
            for blk, gblk in zip(ref.blocks, net.blocks):
                pa = R.RefPerformerAttention(128, num_heads=4, qkv_bias=True, n_orf=16, attn_drop=p)
                g = torch.Generator().manual_seed(3)
                pa.w = torch.randn(16, 32, generator=g) * 0.5
                gblk.attn.w = pa.w.cuda()
                blk.attn = NoEpoch(pa)
    sd = {k.replace('.attn.mod.', '.attn.'): v for k, v in ref.state_dict().items()}
    sd = gi.fill_state_dict(sd, seed=21)
    net.load_state_dict(sd)
    ref.load_state_dict({k: sd[k.replace('.attn.mod.', '.attn.')] for k in ref.state_dict()})
    net.train()
    img = gi.randn(9, 'img', 4, 3, 32, 32)
    tgt = torch.tensor([0, 2, 4, 6])
    VD.manual_seed(17)
    with VD.record() as streams:
        out = net(img.cuda(), epoch) if epoch is not None else net(img.cuda())
    assert len(streams) == {'transformer': 8, 'transformer_timm_lai': 2, 'performer_timm': 2}[variant], streams
    loss = label_smoothing_ce(out, tgt.cuda())
    loss.backward()
    feeder = Feeder(streams)
    R.DROPOUT = feeder
    try:
        out_r = ref(img, epoch) if variant != 'transformer' else ref(img)
        R.label_smoothing_ce(out_r, tgt).backward()
    finally:
        R.DROPOUT = None
    assert feeder.i == len(streams)
    assert relerr(out, out_r) < 1e-5
    rp = dict(ref.named_parameters())
    for n, q in net.named_parameters():
        r = rp.get(n) if n in rp else rp.get(n.replace('.attn.', '.attn.mod.'))
        if r is None or r.grad is None:
            continue
        assert relerr(q.grad, r.grad) < 5e-4, n


def test_graph_capture_refuses_active_dropout():
    from vit_amd import optim
    from vit_amd.graph import GraphedTrainStep
    _, net = _dis_pair(torch.bfloat16, 0.1)
    net.train()
    opt = optim.AdamW(net.parameters(), lr=1e-3, capturable=True)
    x = torch.randn(4, 3, 32, 32, device='cuda')
    y = torch.randint(0, 10, (4,), device='cuda')
    with pytest.raises(ValueError, match='dropout is active'):
        GraphedTrainStep(net, opt, x, y, epoch=30)


def test_drop_path_kernel_per_sample():
    """vitk_drop_path: one decision per sample (element s of the stream), all rows of the sample scaled by 1 / (1 - p);
    residual folded in; the backward is the same scaling of dy."""
    from vit_amd import ops, functional as VF, dropout as VD
    B, N, C, p = 24, 65, 128, 0.4
    VD.manual_seed(31)
    for dtype, tol in ((torch.float32, 1e-6), (torch.bfloat16, 1e-2)):
        x = torch.randn(B * N, C, device='cuda').to(dtype).requires_grad_(True)
        res = torch.randn(B * N, C, device='cuda').to(dtype).requires_grad_(True)
        with VD.record() as streams:
            y = VF.drop_path(x, p, N, res=res)
        keep = ops.dropout_mask(B, streams[0]).float()
        assert 0 < keep.sum().item() < B                                  # the seed gives both kinds of sample
        mul = (keep / (1 - p)).repeat_interleave(N)[:, None]
        assert relerr(y, x.detach().float() * mul + res.detach().float()) < tol
        g = torch.randn_like(y)
        y.backward(g)
        assert relerr(x.grad, g.float() * mul) < tol and torch.equal(res.grad, g)
    with pytest.raises(RuntimeError, match='whole number of samples'):
        ops.dropout(torch.zeros(10, 8, device='cuda'), (0.5, 1, 0), rows_per_sample=3)


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16], ids=['f32', 'bf16'])
def test_timm_lai_with_drop_path_vs_oracle(dtype):
    """VisionTransformerLAI(drop_path_rate > 0) in training mode (reference :94,97-98,124: per-block rates from
    linspace(0, rate, depth), a per-sample decision on each residual branch), together with attention dropout: logits and
    every parameter gradient against the oracle fed the kernels' decisions."""
    from oracle import ref_cpu as R
    from vit_amd import dropout as VD
    from vit_amd.precision import set_compute_dtype
    from vit_amd.timm_classes.timm_locality_aware_transformer import VisionTransformerLAI
    from vit_amd.train import label_smoothing_ce
    me = [10, 20, 30, 40]
    kw = dict(img_size=32, patch_size=4, num_classes=10, embed_dim=128, depth=3, num_heads=4, mlp_ratio=2.0,
              mask_epochs=me, attn_drop_rate=0.1, drop_path_rate=0.5)
    ref = R.RefVisionTransformerLAI(**kw)
    sd = gi.fill_state_dict(ref.state_dict(), seed=23)
    ref.load_state_dict(sd)
    net = VisionTransformerLAI(**kw)
    net.load_state_dict(sd)
    net = set_compute_dtype(net.cuda(), dtype).train()
    img = gi.randn(13, 'img', 8, 3, 32, 32)
    tgt = torch.arange(8)
    VD.manual_seed(4)
    with VD.record() as streams:
        out = net(img.cuda(), 15)
    # block 0 has rate 0 (no decision drawn): 3 attention-dropout sites + 2 blocks x 2 branches
    assert [round(s[0], 3) for s in streams] == [0.1, 0.1, 0.25, 0.25, 0.1, 0.5, 0.5], streams
    label_smoothing_ce(out, tgt.cuda()).backward()
    feeder = Feeder(streams)
    R.DROPOUT = feeder
    try:
        out_r = ref(img, 15)
        R.label_smoothing_ce(out_r, tgt).backward()
    finally:
        R.DROPOUT = None
    assert feeder.i == len(streams)
    f32 = dtype == torch.float32
    assert relerr(out, out_r) < (1e-5 if f32 else 2e-2)
    rp = dict(ref.named_parameters())
    for n, q in net.named_parameters():
        assert relerr(q.grad, rp[n].grad) < (5e-5 if f32 else 8e-2), n
    net.eval()                                                            # eval: the fused one-node blocks, nothing drawn
    with VD.record() as streams, torch.no_grad():
        net(img.cuda(), 15)
    assert streams == []
