"""GPU parity of the FAVOR+ kernels / Performer modules against the reference goldens
(tests/golden/performer.npz: PerformerAttention and MultiheadAttentionPerformer outputs and
gradients produced by the reference itself, with its random matrix ``w`` stored) and the oracle."""
import os

import numpy as np
import pytest
import torch

import golden_inputs as gi
from oracle import ref_cpu as R

pytestmark = pytest.mark.gpu

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')
# fp32: 1e-5 max-norm, the bar of every other fp32 kernel (measured 2-3e-6 on outputs and gradients).
# bf16: FAVOR+ is discontinuous in its inputs (ReLU gates, the argmax that receives the max
# gradient), so merely ROUNDING q/k/v/w to bf16 and running the fp32 oracle already moves single
# gradient elements by 1-10 % (tests/test_favor_floor.py asserts that floor on the oracle alone;
# tools/favor_err.py prints it next to the kernels' errors).  The bf16 bar is therefore
# relative L2 <= 3e-2 (the floor is 1-2e-2) with a loose 0.2 max-norm guard; whole modules (the
# qkv activations are rounded to bf16 as well, and m = 16 features flip more weight per gate) get
# 5e-2 (measured 3.1-3.9e-2 on the two worst gradients, <= 2e-2 on every output).
TOL = {torch.float32: 1e-5, torch.bfloat16: 3e-2}
# fp32 modules (projection GEMMs around the FAVOR+ kernels): the same 1e-5.  Round 2 allowed 1.5e-5 because one gradient
# element of the m = 64 softmax golden moved between 0.8e-5 and 1.0001e-5 from run to run with the order of the split-K
# atomics of the fp32 weight-gradient GEMMs; the fp32 path no longer splits K (csrc/gemm.hip: one K-ordered chain per
# element, bit-reproducible).  bf16 modules: 5e-2 relative L2 (the fp32 oracle on bf16-rounded inputs alone is already at
# 1-2e-2, tests/test_favor_floor.py).  The per-tensor errors measured against these bars are written by the tests
# themselves when VITK_REPORT_ERRORS names a file; profiles/r03_favor_parity_errors.jsonl is that file from one MI355X run.
MODULE_SCALE = {torch.float32: 1.0, torch.bfloat16: 5.0 / 3.0}
DTYPES = [torch.float32, torch.bfloat16]
IDS = ['f32', 'bf16']


def relerr(a, b):
    a = torch.as_tensor(np.asarray(a)).double().reshape(-1)
    b = torch.as_tensor(np.asarray(b)).double().reshape(-1)
    return ((a - b).abs().max() / (b.abs().max() + 1e-30)).item()


def l2err(a, b):
    a = torch.as_tensor(np.asarray(a)).double().reshape(-1)
    b = torch.as_tensor(np.asarray(b)).double().reshape(-1)
    return ((a - b).norm() / (b.norm() + 1e-30)).item()


def ok(a, b, dtype, scale=1.0):
    if dtype == torch.float32:
        return relerr(a, b) < TOL[dtype] * scale
    return l2err(a, b) < TOL[dtype] * scale and relerr(a, b) < 0.2


def report(test, dtype, errs):
    """errs: {tensor: (max-norm error, relative L2 error)} -> one JSON line in $VITK_REPORT_ERRORS (the margin to the bars)."""
    path = os.environ.get('VITK_REPORT_ERRORS')
    if path:
        import json
        with open(path, 'a') as f:
            f.write(json.dumps({'test': test, 'dtype': str(dtype).replace('torch.', ''),
                                'bar': {'max_norm': TOL[dtype] * MODULE_SCALE[dtype] if dtype == torch.float32 else 0.2,
                                        'rel_l2': None if dtype == torch.float32 else TOL[dtype] * MODULE_SCALE[dtype]},
                                'errors': {k: {'max_norm': v[0], 'rel_l2': v[1]} for k, v in errs.items()}}) + '\n')


def run_with_grads(fn, x, params, gout_seed):
    x = x.clone().cuda().requires_grad_(True)
    y = fn(x)
    g = gi.randn(gout_seed, 'gout', *y.shape).cuda()
    (y.float() * g).sum().backward()
    res = {'y': gi.sub(y.detach().float().cpu(), 1).numpy(), 'dx': gi.sub(x.grad.cpu(), 1).numpy()}
    for n, p in params:
        if p.grad is not None:
            res['d_' + n] = gi.sub(p.grad.float().cpu(), 1).numpy()
            p.grad = None
    return res


@pytest.mark.parametrize('dtype', DTYPES, ids=IDS)
@pytest.mark.parametrize('kernel', ['softmax', 'relu'])
@pytest.mark.parametrize('m', [16, 64])
def test_performer_attention_vs_reference_golden(m, kernel, dtype):
    from vit_amd.precision import set_compute_dtype
    from vit_amd.timm_classes.timm_vision_performer import Attention, PerformerAttention
    g = np.load(os.path.join(G, 'performer.npz'))
    B, N, D, H = 2, 145, 256, 4
    x = gi.randn(41, 'x', B, N, D)
    mod = PerformerAttention(Attention(D, num_heads=H, qkv_bias=True), D, num_heads=H, qkv_bias=True, n_orf=m,
                             kernel=kernel)
    mod.load_state_dict(gi.fill_state_dict(mod.state_dict(), seed=42))
    pre = f'timm_{kernel}_m{m}_'
    set_compute_dtype(mod.cuda(), dtype)
    mod.w = torch.from_numpy(g[pre + 'w']).cuda()
    r = run_with_grads(lambda t: mod(t), x, list(mod.named_parameters()), 44)
    report(f'performer_attention[{kernel},m{m}]', dtype, {k: (relerr(v, g[pre + k]), l2err(v, g[pre + k])) for k, v in r.items()})
    bad = [(k, relerr(v, g[pre + k]), l2err(v, g[pre + k])) for k, v in r.items() if not ok(v, g[pre + k], dtype, MODULE_SCALE[dtype])]
    assert not bad, bad


@pytest.mark.parametrize('dtype', DTYPES, ids=IDS)
@pytest.mark.parametrize('kernel', ['softmax', 'relu'])
def test_mha_performer_vs_reference_golden(kernel, dtype):
    from vit_amd.model import MultiheadAttentionPerformer
    from vit_amd.precision import set_compute_dtype
    g = np.load(os.path.join(G, 'performer.npz'))
    L, B, E = 65, 2, 128
    x = gi.randn(45, 'xm', L, B, E)
    mod = MultiheadAttentionPerformer(E, 4, dropout=0.0, num_orf=32, kernel=kernel)
    mod.load_state_dict(gi.fill_state_dict(mod.state_dict(), seed=46))
    pre = f'mha_{kernel}_'
    set_compute_dtype(mod.cuda(), dtype)
    mod.w = torch.from_numpy(g[pre + 'w']).cuda()
    r = run_with_grads(lambda t: mod(t, t, t)[0], x, list(mod.named_parameters()), 48)
    report(f'mha_performer[{kernel}]', dtype, {k: (relerr(v, g[pre + k]), l2err(v, g[pre + k])) for k, v in r.items()})
    bad = [(k, relerr(v, g[pre + k]), l2err(v, g[pre + k])) for k, v in r.items() if not ok(v, g[pre + k], dtype, MODULE_SCALE[dtype])]
    assert not bad, bad


def _favor_vs_oracle(kernel, B, L, H, dh, m, dtype):
    from vit_amd import functional as VF
    D = H * dh
    torch.manual_seed(5)
    qkv = (torch.randn(B * L, 3 * D) * 0.5)
    w = R.gaussian_orthogonal_random_matrix(m, dh, generator=torch.Generator().manual_seed(1))
    gout = torch.randn(B * L, D)
    ratio = m ** -0.25
    if dtype == torch.bfloat16:      # the oracle sees what the kernel sees: a rounded input that crosses a ReLU gate or
        qkv, w = qkv.bfloat16().float(), w.bfloat16().float()      # moves the arg-max is not the kernel's error
    # oracle
    qr = qkv.clone().requires_grad_(True)
    q, k, v = (qr[:, i * D:(i + 1) * D].view(B, L, H, dh) for i in range(3))
    if kernel == 'softmax':
        kp, qp = R.favor_softmax_features(k, w, False, ratio), R.favor_softmax_features(q, w, True, ratio)
    else:
        kp, qp = R.favor_relu_features(k, w, ratio), R.favor_relu_features(q, w, ratio)
    ref = R.favor_linear_attention(qp, kp, v).reshape(B * L, D)
    (ref * gout).sum().backward()
    xg = qkv.clone().cuda().to(dtype).requires_grad_(True)
    out = VF.favor_attention(xg, w.cuda(), B, L, H, dh, kernel, ratio)
    (out.float() * gout.cuda()).sum().backward()
    assert ok(out.detach().float().cpu(), ref.detach(), dtype, 2), (relerr(out.detach().float().cpu(), ref.detach()))
    assert ok(xg.grad.float().cpu(), qr.grad, dtype, 2), (relerr(xg.grad.float().cpu(), qr.grad), l2err(xg.grad.float().cpu(), qr.grad))


@pytest.mark.parametrize('dtype', DTYPES, ids=IDS)
@pytest.mark.parametrize('kernel,m,L', [('softmax', 256, 577), ('softmax', 64, 577), ('relu', 48, 100)])
def test_favor_large_vs_oracle(kernel, m, L, dtype):
    """Config-5 shape (577 tokens, dh 64), m = 256 / 64, and a ragged m: kernel vs the CPU oracle."""
    _favor_vs_oracle(kernel, 2, L, 3, 64, m, dtype)


@pytest.mark.parametrize('dtype', DTYPES, ids=IDS)
@pytest.mark.parametrize('kernel', ['softmax', 'relu'])
@pytest.mark.parametrize('B,L,H,dh,m', [(1, 5, 3, 24, 10), (3, 33, 1, 8, 16), (1, 130, 5, 40, 17), (2, 64, 2, 64, 64),
                                        (1, 1, 1, 16, 1), (5, 129, 3, 32, 33), (2, 70, 7, 56, 7)])
def test_favor_edge_shapes_vs_oracle(kernel, B, L, H, dh, m, dtype):
    """The LDS-staged kernels (csrc/favor_fast.hip: m <= 64) at their edges: fewer tokens than a wave's block, a single
    token, (b, h) counts that are no multiple of the XCD count, ragged feature counts on both pitches (m = 1, 7, 10 on
    16; 17, 33 on 64), heads of 8 ... 64 with a half-filled last K block (24, 40, 56)."""
    _favor_vs_oracle(kernel, B, L, H, dh, m, dtype)


DISPERF_SHAPES = (('a', 2, 65, 128, 4, 16), ('b', 2, 145, 256, 4, 64), ('c', 3, 50, 96, 2, 40))


def _dis_performer_oracle_bf16_inputs(g, name, B, N, D, H, m, kernel):
    """The CPU oracle in fp32 arithmetic on what the bf16 path actually sees: x, pos_embed, the weight
    matrices, w and the five projections' outputs rounded to bf16 (straight-through for the gradient)."""
    bf = lambda t: t.to(torch.bfloat16).float()
    mod = R.RefDisentangledSelfAttentionPerformer(D, H, norf=m, kernel=kernel)
    sd = gi.fill_state_dict(mod.state_dict(), seed=62)
    mod.load_state_dict({k: bf(v) if v.dim() == 2 else v for k, v in sd.items()})
    mod.w = bf(torch.from_numpy(g[f'attn_{name}_{kernel}_w']))
    for lin in (mod.query_proj, mod.key_proj, mod.value_proj, mod.pos_key_proj, mod.pos_query_proj):
        lin.register_forward_hook(lambda m_, i, o: o + (bf(o.detach()) - o.detach()))
    x = bf(gi.randn(61, 'x_' + name, B, N, D)).requires_grad_(True)
    pos = bf(gi.randn(61, 'pos_' + name, N, 1, D, scale=0.5)).requires_grad_(True)
    y = mod(x, x, x, pos_embed=pos)['hidden_states']
    (y * gi.randn(64, 'gout', *y.shape)).sum().backward()
    res = {'y': gi.sub(y.detach(), 1).numpy(), 'dx': gi.sub(x.grad, 1).numpy(), 'd_pos': gi.sub(pos.grad, 1).numpy()}
    for n, p in mod.named_parameters():
        res['d_' + n] = gi.sub(p.grad, 1).numpy()
    return res


@pytest.mark.parametrize('dtype', DTYPES, ids=IDS)
@pytest.mark.parametrize('kernel', ['softmax', 'relu'])
@pytest.mark.parametrize('shape', DISPERF_SHAPES, ids=[s[0] for s in DISPERF_SHAPES])
def test_dis_performer_attention_vs_reference_golden(shape, kernel, dtype):
    """DisentangledSelfAttentionPerformer against the reference's own outputs and
    gradients (tests/golden/dis_performer.npz), d pos_embed included."""
    from vit_amd.precision import set_compute_dtype
    from vit_amd.disentangled_transformer.disentangled_transformer import DisentangledSelfAttentionPerformer
    g = np.load(os.path.join(G, 'dis_performer.npz'))
    name, B, N, D, H, m = shape
    x = gi.randn(61, 'x_' + name, B, N, D)
    pos = gi.randn(61, 'pos_' + name, N, 1, D, scale=0.5).cuda().requires_grad_(True)
    mod = DisentangledSelfAttentionPerformer(D, H, dropout=0.0, relative_attention=True, max_position_embeddings=D,
                                             norf=m, kernel=kernel)
    mod.load_state_dict(gi.fill_state_dict(mod.state_dict(), seed=62))
    pre = f'attn_{name}_{kernel}_'
    set_compute_dtype(mod.cuda(), dtype)
    mod.w = torch.from_numpy(g[pre + 'w']).cuda()
    r = run_with_grads(lambda t: mod(t, t, t, pos_embed=pos)['hidden_states'], x, list(mod.named_parameters()), 64)
    r['d_pos'] = gi.sub(pos.grad.float().cpu(), 1).numpy()
    assert set(r) == set(k[len(pre):] for k in g.files if k.startswith(pre)) - {'w'}
    if dtype == torch.float32:
        bad = [(k, relerr(v, g[pre + k])) for k, v in r.items() if not ok(v, g[pre + k], dtype)]
        assert not bad, bad
        return
    # bf16.  With Pq added the output is no longer invariant to the scale of q', so the gradient through the
    # row max (routed to the argmax feature) is large, and ROUNDING the inputs to bf16 flips argmaxes: the fp32
    # oracle fed bf16-rounded inputs / projections is itself up to 0.13 (relative L2) away from the fp32 golden
    # on the query-side gradients (case b, softmax; 0.02-0.06 elsewhere).  So the kernel is held to the usual
    # bf16 bar against THAT oracle (same rounded inputs: what is left is the kernel's own error), and to the
    # rounding floor (0.2) against the reference's fp32 numbers.
    ref = _dis_performer_oracle_bf16_inputs(g, name, B, N, D, H, m, kernel)
    bad = [(k, relerr(v, ref[k]), l2err(v, ref[k])) for k, v in r.items() if not ok(v, ref[k], dtype, MODULE_SCALE[dtype])]
    assert not bad, bad
    far = [(k, l2err(v, g[pre + k])) for k, v in r.items() if l2err(v, g[pre + k]) > 0.2]
    assert not far, far


@pytest.mark.parametrize('dtype', DTYPES, ids=IDS)
@pytest.mark.parametrize('kernel', ['softmax', 'relu'])
def test_dis_performer_layer_vs_reference_golden(kernel, dtype):
    from vit_amd.precision import set_compute_dtype
    from vit_amd.disentangled_transformer.disentangled_transformer import DisentangledEncoderLayer
    g = np.load(os.path.join(G, 'dis_performer.npz'))
    B, N, D, H, m = 2, 65, 128, 4, 32
    x = gi.randn(65, 'x_layer', B, N, D)
    pos = gi.randn(65, 'pos_layer', N, 1, D, scale=0.5).cuda().requires_grad_(True)
    lay = DisentangledEncoderLayer(d_model=D, nhead=H, dropout=0.0, relative_attention=True, max_position_embeddings=D,
                                   dim_feedforward=256, performers=True, norf=m, kernel=kernel)
    lay.load_state_dict(gi.fill_state_dict(lay.state_dict(), seed=66))
    pre = f'layer_{kernel}_'
    set_compute_dtype(lay.cuda(), dtype)
    lay.dis_attn.w = torch.from_numpy(g[pre + 'w']).cuda()
    r = run_with_grads(lambda t: lay(t, pos_embed=pos), x, list(lay.named_parameters()), 68)
    r['d_pos'] = gi.sub(pos.grad.float().cpu(), 1).numpy()
    assert set(r) == set(k[len(pre):] for k in g.files if k.startswith(pre)) - {'w'}
    if dtype == torch.float32:
        bad = [(k, relerr(v, g[pre + k])) for k, v in r.items() if not ok(v, g[pre + k], dtype)]
    else:       # bf16: the output to the module bar, gradients to the input-rounding floor (see the attention test)
        bad = [(k, l2err(v, g[pre + k])) for k, v in r.items()
               if l2err(v, g[pre + k]) > (TOL[dtype] * MODULE_SCALE[dtype] if k == 'y' else 0.2)]
    assert not bad, bad
    with pytest.raises(TypeError):          # the reference's forward has no ``epoch`` (dis :218-220, :496)
        lay(x.cuda(), pos_embed=pos, epoch=0)


@pytest.mark.parametrize('dtype', DTYPES, ids=IDS)
def test_dis_performer_model_vs_reference_golden(dtype):
    """2-layer --attention disentangled_performer model through VisionTransformer: logits, d img, d pos_embed (a
    parameter that only this variant trains), a few weights, and the set of parameters without gradient."""
    from vit_amd import model as M
    from vit_amd.precision import set_compute_dtype
    from vit_amd.disentangled_transformer.disentangled_transformer import DisentangledTransformer
    g = np.load(os.path.join(G, 'dis_performer.npz'))
    tr = DisentangledTransformer(d_model=128, dropout=0.0, nhead=4, dim_feedforward=256, num_encoder_layers=2,
                                 normalize_before=True, pos_att_type='c2p|p2c', max_position_embeddings=128,
                                 relative_attention=True, performers=True, norf=16, kernel='softmax')
    net = M.VisionTransformer(img_size=32, patch_size=4, in_chans=3, num_classes=10, transformer=tr)
    net.load_state_dict(gi.fill_state_dict(net.state_dict(), seed=71))
    set_compute_dtype(net.cuda(), dtype)
    for i, lay in enumerate(tr.encoder.layers):
        lay.dis_attn.w = torch.from_numpy(g[f'model_w{i}']).cuda()
    img = gi.randn(71, 'img', 2, 3, 32, 32)
    r = run_with_grads(lambda t: net(t), img, list(net.named_parameters()), 73)
    nograd = sorted(n for n, p in net.named_parameters() if ('d_' + n) not in r)
    assert nograd == sorted(g['model_nograd'].tolist())
    keys = [k[len('model_'):] for k in g.files if k.startswith('model_') and k[6:7] in 'yd']
    if dtype == torch.float32:
        bad = [(k, relerr(r[k], g['model_' + k])) for k in keys if not ok(r[k], g['model_' + k], dtype, 2.0)]
    else:       # bf16: logits to twice the module bar (two layers), gradients to the input-rounding floor
        bad = [(k, l2err(r[k], g['model_' + k])) for k in keys
               if l2err(r[k], g['model_' + k]) > (2 * TOL[dtype] * MODULE_SCALE[dtype] if k == 'y' else 0.2)]
    assert not bad, bad


@pytest.mark.parametrize('dtype', DTYPES, ids=IDS)
@pytest.mark.parametrize('kernel,m,B,L,H,dh', [('softmax', 16, 37, 197, 3, 64), ('softmax', 256, 9, 100, 2, 64),
                                                 ('relu', 48, 21, 65, 4, 32), ('softmax', 64, 130, 50, 2, 48)])
def test_favor_pos_attention_large_batch_vs_oracle(kernel, m, B, L, H, dh, dtype):
    """Batches beyond the goldens' (B <= 3): the backward's reduction over b (4 waves, several b per pass when
    mp = 16) and every feature pitch (16 / 64 / 256), kernel vs the CPU oracle's formulas on the same inputs."""
    from vit_amd import functional as VF
    D = H * dh
    gen = torch.Generator().manual_seed(B * L + m)
    rnd = lambda *s: torch.randn(*s, generator=gen)
    bf = (lambda t: t.to(torch.bfloat16).float()) if dtype == torch.bfloat16 else (lambda t: t)
    qkv, posqk, gout = bf(rnd(B * L, 3 * D) * 0.5), bf(rnd(L, 2 * D) * 0.5), rnd(B * L, D)
    w = bf(R.gaussian_orthogonal_random_matrix(m, dh, generator=torch.Generator().manual_seed(2)))
    ratio = m ** -0.25
    feats = (lambda d, isq: R.favor_softmax_features(d, w, isq, ratio)) if kernel == 'softmax' else \
        (lambda d, isq: R.favor_relu_features(d, w, ratio))
    qr, pr = qkv.clone().requires_grad_(True), posqk.clone().requires_grad_(True)
    q, k, v = (qr[:, i * D:(i + 1) * D].view(B, L, H, dh) for i in range(3))
    Pq, Pk = feats(pr[:, :D].view(1, L, H, dh), True), feats(pr[:, D:].view(1, L, H, dh), False)
    ref = R.favor_linear_attention(feats(q, True) + Pq, feats(k, False) + Pk, v).reshape(B * L, D)
    (ref * gout).sum().backward()
    xg = qkv.clone().cuda().to(dtype).requires_grad_(True)
    pg = posqk.clone().cuda().to(dtype).requires_grad_(True)
    out = VF.favor_pos_attention(xg, pg, w.cuda(), B, L, H, dh, kernel, ratio)
    (out.float() * gout.cuda()).sum().backward()
    for name, a, b_ in (('out', out.detach(), ref.detach()), ('dqkv', xg.grad, qr.grad), ('dpos', pg.grad, pr.grad)):
        a = a.float().cpu()
        assert ok(a, b_, dtype, 2), (name, relerr(a, b_), l2err(a, b_))


def test_favor_pos_rejects():
    from vit_amd import ops
    x = torch.randn(10, 64, device='cuda')
    w = torch.randn(16, 32, device='cuda')
    with pytest.raises(ValueError):
        ops.favor_pos_fwd(x, 2, 32, w.to(torch.bfloat16), 0, True, 0.5, 1e-6)       # dtype mismatch
    feat, st = ops.favor_pos_fwd(x, 2, 32, w, 0, True, 0.5, 1e-6)
    assert feat.shape == (10, 2, ops.favor_feature_pitch(16))
    with pytest.raises(ValueError):
        ops.favor_pos_bwd(x, 2, 32, w, 0, True, 0.5, 1e-6, torch.zeros(3, 10, 2, 8, device='cuda'), st)
    with pytest.raises(RuntimeError):
        ops.favor_pos_fwd(torch.randn(10, 200, device='cuda'), 2, 100, torch.randn(16, 100, device='cuda'),
                          0, True, 0.5, 1e-6)                                       # head dim > 64


def test_performer_models_run():
    """Whole models with FAVOR+ blocks (train.py --attention performer_timm / performer) step."""
    from vit_amd.train import build_model, label_smoothing_ce
    for variant in ('performer_timm', 'performer', 'disentangled_performer'):
        net = build_model(variant, 'cifar', img_size=32, patch_size=4, num_classes=10, dtype=torch.bfloat16,
                          depth=2, num_orf=16)
        img = torch.randn(4, 3, 32, 32, device='cuda')
        out = net(img, 0) if variant == 'performer_timm' else net(img)
        loss = label_smoothing_ce(out, torch.tensor([1, 2, 3, 4], device='cuda'))
        loss.backward()
        assert torch.isfinite(loss).item()
        assert all(torch.isfinite(p.grad).all().item() for p in net.parameters() if p.grad is not None)


@pytest.mark.gpu
@pytest.mark.parametrize('kernel', ['softmax', 'relu'])
def test_performer_block_one_node_matches_op_by_op(kernel):
    """A timm block whose attention was replaced by PerformerAttention (train.py:147-165) runs as ONE autograd node
    (functional.PreLNBlockFn with FAVOR+ inside: residuals in the GEMM epilogues); the op-by-op form of the same block
    (layer_norm / linear / favor_attention / mlp nodes, torch adds) is the check: same kernels for everything but the
    residual adds, so the two agree to a few bf16 roundings."""
    from vit_amd.timm_classes import timm_locality_aware_transformer as TL
    from vit_amd.train import build_model
    torch.manual_seed(0)
    net = build_model('performer_timm', preset='S', img_size=64, patch_size=16, num_classes=10, depth=2, dim=128,
                      mlp=256, heads=4, kernel=kernel, num_orf=32)
    img = torch.randn(4, 3, 64, 64, device='cuda')

    def run(fuse):
        TL.FUSE_SWAPPED = fuse
        try:
            net.zero_grad(set_to_none=True)
            out = net(img, 0)
            out.float().square().sum().backward()
            return out.detach().float().clone(), {n: p.grad.float().clone() for n, p in net.named_parameters() if p.grad is not None}
        finally:
            TL.FUSE_SWAPPED = True

    out1, g1 = run(True)
    out0, g0 = run(False)
    assert g1.keys() == g0.keys()
    rel = lambda a, b: float((a - b).norm() / b.norm().clamp_min(1e-12))
    assert rel(out1, out0) < 2e-2
    for n in g0:
        assert rel(g1[n], g0[n]) < 3e-2, n
