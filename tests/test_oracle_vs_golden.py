"""Pin the CPU oracle (oracle/ref_cpu.py) against outputs of the reference itself
(tests/golden/*.npz, produced by tools/gen_goldens.py).  CPU only."""
import hashlib
import os
import sys

import numpy as np
import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))

import golden_inputs as gi  # noqa: E402
from oracle import ref_cpu as R  # noqa: E402

G = os.path.join(HERE, 'golden')


def load(name):
    return np.load(os.path.join(G, name))


def close(a, b, rtol=2e-5, atol=2e-6, what=''):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    assert a.shape == b.shape, (what, a.shape, b.shape)
    err = np.abs(a - b).max()
    ref = np.abs(b).max()
    assert err <= atol + rtol * ref, f'{what}: max err {err:.3e} vs max ref {ref:.3e}'


def run_with_grads(fn, x, params, gout_seed, step):
    x = x.clone().requires_grad_(True)
    y = fn(x)
    g = gi.randn(gout_seed, 'gout', *y.shape)
    (y * g).sum().backward()
    res = {'y': gi.sub(y, step).numpy(), 'dx': gi.sub(x.grad, step).numpy()}
    for n, p in params:
        if p.grad is not None:
            res['d_' + n] = gi.sub(p.grad, step).numpy()
            p.grad = None
    return res


def test_masks_bit_exact():
    g = load('lai_masks.npz')
    for N in (65, 197, 577):
        for w in (8, 12, 18, 20):
            m = R.lai_band_mask(N, w)
            assert m.shape == (1, 1, N, N) and m.dtype == torch.float32
            bits = np.packbits(m.numpy().reshape(N, N).astype(np.uint8), axis=None)
            assert np.array_equal(bits, g[f'bits_{N}_{w}'])
            sha = np.frombuffer(hashlib.sha256(m.numpy().tobytes()).digest(), dtype=np.uint8)
            assert np.array_equal(sha, g[f'sha_{N}_{w}'])


def test_stage_schedule():
    me = gi.LAI_MASK_EPOCHS
    assert [R.lai_stage_width(e, me) for e in (0, 9, 10, 19, 20, 29, 30, 39, 40, 99)] == \
        [8, 8, 12, 12, 18, 18, 20, 20, None, None]
    assert R.lai_stage_width(0, None) is None


@pytest.mark.parametrize('shape', gi.ATTN_SHAPES, ids=lambda s: s[0])
def test_lai_attention(shape):
    g = load('lai_attention.npz')
    name, B, N, D, H, step = shape
    mod = R.RefAttentionLAI(D, num_heads=H, qkv_bias=True, mask_epochs=gi.LAI_MASK_EPOCHS)
    mod.load_state_dict(gi.fill_state_dict(mod.state_dict(), seed=11))
    x = gi.randn(11, 'x_' + name, B, N, D)
    for ep in gi.LAI_STAGE_EPOCHS:
        r = run_with_grads(lambda t: mod(t, ep), x, list(mod.named_parameters()), 12, step)
        for k, v in r.items():
            close(v, g[f'{name}_ep{ep}_{k}'], what=f'{name} ep{ep} {k}')


@pytest.mark.parametrize('shape', gi.ATTN_SHAPES, ids=lambda s: s[0])
@pytest.mark.parametrize('kind', ['dis', 'dislai'])
def test_dis_attention(shape, kind):
    g = load('dis_attention.npz')
    name, B, N, D, H, step = shape
    S = D
    rel0 = gi.randn(21, 'rel_' + name, 2 * S, D)
    x = gi.randn(21, 'x_' + name, B, N, D)
    mod = R.RefDisentangledSelfAttention(
        D, H, dropout=0.0, max_position_embeddings=S,
        mask_epochs=gi.LAI_MASK_EPOCHS if kind == 'dislai' else None, lai=(kind == 'dislai'))
    mod.load_state_dict(gi.fill_state_dict(mod.state_dict(), seed=22))
    for ep in ([None] if kind == 'dis' else gi.LAI_STAGE_EPOCHS):
        rel = rel0.clone().requires_grad_(True)
        logits = {}

        def fn(t):
            o = mod(t, t, t, rel_embeddings=rel, epoch=0 if ep is None else ep)
            logits['l'] = o['attention_logits']
            return o['hidden_states']
        r = run_with_grads(fn, x, list(mod.named_parameters()), 23, step)
        r['d_rel'] = gi.sub(rel.grad, step).numpy()
        r['logits'] = gi.sub(logits['l'], step).numpy()
        for k, v in r.items():
            close(v, g[f'{name}_{kind}_ep{ep}_{k}'], what=f'{name} {kind} ep{ep} {k}',
                  rtol=1e-4 if k == 'logits' else 2e-5)  # logits hold -1e9 fills


@pytest.mark.parametrize('kind', ['dis', 'dislai'])
def test_dis_model(kind):
    g = load('dis_model.npz')
    kw = dict(lai=True, mask_epochs=gi.LAI_MASK_EPOCHS) if kind == 'dislai' else {}
    tr = R.RefDisentangledTransformer(d_model=128, dropout=0.0, nhead=4, dim_feedforward=256,
                                      num_encoder_layers=2, normalize_before=True,
                                      pos_att_type='c2p|p2c', max_position_embeddings=128,
                                      relative_attention=True, **kw)
    net = R.RefVisionTransformer(img_size=32, patch_size=4, in_chans=3, num_classes=10, transformer=tr)
    net.load_state_dict(gi.fill_state_dict(net.state_dict(), seed=31))
    img = gi.randn(31, 'img', 2, 3, 32, 32)
    for ep in ([None] if kind == 'dis' else [0, 30, 40]):
        fn = (lambda t: net(t)) if ep is None else (lambda t: net(t, ep))
        r = run_with_grads(fn, img, list(net.named_parameters()), 32, 1)
        pre = f'{kind}_ep{ep}_'
        for k in g.files:
            if k.startswith(pre) and not k.endswith('nograd'):
                close(r[k[len(pre):]], g[k], what=k, rtol=5e-5)
        nograd = sorted(n for n, p in net.named_parameters() if ('d_' + n) not in r)
        assert nograd == sorted(g[pre + 'nograd'].tolist())


def test_performer_timm():
    g = load('performer.npz')
    B, N, D, H = 2, 145, 256, 4
    x = gi.randn(41, 'x', B, N, D)
    for m in (16, 64):
        for kernel in ('softmax', 'relu'):
            mod = R.RefPerformerAttention(D, num_heads=H, qkv_bias=True, n_orf=m, kernel=kernel)
            mod.load_state_dict(gi.fill_state_dict(mod.state_dict(), seed=42))
            pre = f'timm_{kernel}_m{m}_'
            mod.w = torch.from_numpy(g[pre + 'w'])
            r = run_with_grads(lambda t: mod(t), x, list(mod.named_parameters()), 44, 1)
            for k, v in r.items():
                close(v, g[pre + k], what=pre + k, rtol=5e-5)


def test_performer_mha():
    g = load('performer.npz')
    L, B, E = 65, 2, 128
    x = gi.randn(45, 'xm', L, B, E)
    for kernel in ('softmax', 'relu'):
        mod = R.RefMultiheadAttentionPerformer(E, 4, dropout=0.0, num_orf=32, kernel=kernel)
        mod.load_state_dict(gi.fill_state_dict(mod.state_dict(), seed=46))
        pre = f'mha_{kernel}_'
        mod.w = torch.from_numpy(g[pre + 'w'])
        r = run_with_grads(lambda t: mod(t, t, t)[0], x, list(mod.named_parameters()), 48, 1)
        for k, v in r.items():
            close(v, g[pre + k], what=pre + k, rtol=5e-5)


DISPERF_SHAPES = (('a', 2, 65, 128, 4, 16), ('b', 2, 145, 256, 4, 64), ('c', 3, 50, 96, 2, 40))


def test_dis_performer_attention():
    """RefDisentangledSelfAttentionPerformer vs the reference's DisentangledSelfAttentionPerformer
    (dis :431), outputs and every gradient incl. d pos_embed."""
    g = load('dis_performer.npz')
    for name, B, N, D, H, m in DISPERF_SHAPES:
        x = gi.randn(61, 'x_' + name, B, N, D)
        pos0 = gi.randn(61, 'pos_' + name, N, 1, D, scale=0.5)
        for kernel in ('softmax', 'relu'):
            mod = R.RefDisentangledSelfAttentionPerformer(D, H, norf=m, kernel=kernel)
            mod.load_state_dict(gi.fill_state_dict(mod.state_dict(), seed=62))
            pre = f'attn_{name}_{kernel}_'
            mod.w = torch.from_numpy(g[pre + 'w'])
            pos = pos0.clone().requires_grad_(True)
            r = run_with_grads(lambda t: mod(t, t, t, pos_embed=pos)['hidden_states'], x,
                               list(mod.named_parameters()), 64, 1)
            r['d_pos'] = gi.sub(pos.grad, 1).numpy()
            assert set(r) == set(k[len(pre):] for k in g.files if k.startswith(pre)) - {'w'}
            for k, v in r.items():
                close(v, g[pre + k], what=pre + k, rtol=5e-5)


def test_dis_performer_layer():
    g = load('dis_performer.npz')
    B, N, D, H, m = 2, 65, 128, 4, 32
    x = gi.randn(65, 'x_layer', B, N, D)
    pos0 = gi.randn(65, 'pos_layer', N, 1, D, scale=0.5)
    for kernel in ('softmax', 'relu'):
        lay = R.RefDisentangledPerformerLayer(D, H, 256, norf=m, kernel=kernel)
        sd = gi.fill_state_dict({k: torch.empty(v) for k, v in DISPERF_LAYER_KEYS(D, 256).items()}, seed=66)
        lay.load_state_dict({k: v for k, v in sd.items() if k in lay.state_dict()})
        pre = f'layer_{kernel}_'
        lay.dis_attn.w = torch.from_numpy(g[pre + 'w'])
        pos = pos0.clone().requires_grad_(True)
        r = run_with_grads(lambda t: lay(t, pos), x, list(lay.named_parameters()), 68, 1)
        r['d_pos'] = gi.sub(pos.grad, 1).numpy()
        assert set(r) == set(k[len(pre):] for k in g.files if k.startswith(pre)) - {'w'}
        for k, v in r.items():
            close(v, g[pre + k], what=pre + k, rtol=5e-5)


def test_dis_performer_model():
    """RefVisionTransformer around RefDisentangledPerformerTransformer vs the reference's 2-layer
    --attention disentangled_performer model: logits, d img, d pos_embed, a few weights."""
    g = load('dis_performer.npz')
    tr = R.RefDisentangledPerformerTransformer(d_model=128, nhead=4, num_encoder_layers=2, dim_feedforward=256,
                                               normalize_before=True, norf=16, kernel='softmax')
    net = R.RefVisionTransformer(img_size=32, patch_size=4, in_chans=3, num_classes=10, transformer=tr)
    # values depend on key names and shapes only; the reference's layers also carry three unused projections
    net.load_state_dict(gi.fill_state_dict(net.state_dict(), seed=71))
    for i, lay in enumerate(tr.encoder.layers):
        lay.dis_attn.w = torch.from_numpy(g[f'model_w{i}'])
    img = gi.randn(71, 'img', 2, 3, 32, 32)
    r = run_with_grads(lambda t: net(t), img, list(net.named_parameters()), 73, 1)
    keys = [k[len('model_'):] for k in g.files if k.startswith('model_') and k[6:7] in 'yd']
    assert len(keys) == 9
    for k in keys:
        close(r[k], g['model_' + k], what='model_' + k, rtol=1e-4)
    unused = {n for n, p in net.named_parameters() if ('d_' + n) not in r}
    assert unused <= set(g['model_nograd'].tolist())


def DISPERF_LAYER_KEYS(D, F):
    """state_dict shapes of the reference's DisentangledEncoderLayer(performers=True) (values in the
    fixtures depend on key names only: golden_inputs.fill_state_dict)."""
    shapes = {}
    for lin, (o, i) in {'dis_attn.query_proj': (D, D), 'dis_attn.key_proj': (D, D), 'dis_attn.value_proj': (D, D),
                        'dis_attn.pos_key_proj': (D, D), 'dis_attn.pos_query_proj': (D, D), 'query_proj': (D, D),
                        'key_proj': (D, D), 'value_proj': (D, D), 'dense_AttnOut': (D, D),
                        'dense_intermediate': (F, D), 'dense_out': (D, F)}.items():
        shapes[lin + '.weight'] = (o, i)
        shapes[lin + '.bias'] = (o,)
    for ln in ('layer_norm_AttnOut', 'layer_norm_out'):
        shapes[ln + '.weight'] = (D,)
        shapes[ln + '.bias'] = (D,)
    return shapes


def test_orthogonal_features_shape():
    w = R.gaussian_orthogonal_random_matrix(80, 32, generator=torch.Generator().manual_seed(0))
    assert w.shape == (80, 32)
    q = w[:32] / w[:32].norm(dim=1, keepdim=True)
    close((q @ q.t()).numpy(), np.eye(32), atol=1e-5)


def test_config1_step():
    g = load('config1_step.npz')
    tr = R.RefTransformer(d_model=256, dropout=0.0, nhead=8, dim_feedforward=1024,
                          num_encoder_layers=6, normalize_before=True)
    net = R.RefVisionTransformer(img_size=32, patch_size=4, in_chans=3, num_classes=10, transformer=tr)
    net.load_state_dict(gi.fill_state_dict(net.state_dict(), seed=51))
    img = gi.randn(51, 'img', 8, 3, 32, 32)
    tgt = torch.from_numpy(g['target'])
    opt = torch.optim.AdamW(net.parameters(), lr=1e-4)
    close(net(img).detach().numpy(), g['logits'], what='logits', rtol=5e-5)
    loss = R.train_step(net, opt, img, tgt)
    close(loss.numpy(), g['loss'], what='loss')
    close(net.head.weight.detach().numpy(), g['head.weight_after'], what='head after step')
    close(torch.nn.functional.cross_entropy(net(img), tgt).detach().numpy(), g['loss_after'], what='loss after')
