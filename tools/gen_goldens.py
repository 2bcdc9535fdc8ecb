#!/usr/bin/env python3
"""Generate ``tests/golden/*.npz`` by running the REFERENCE (``/root/reference``) on CPU.

Runs only in the build container (the reference never travels to the GPU box).  The
reference is imported unmodified; three generator-side shims make it importable/runnable
on this CPU-only torch 2.10 image:
  * ``torch._six`` (removed in torch 2.x; model.py:9) -> a module exposing container_abcs;
  * ``Tensor.cuda`` -> identity and ``Tensor.get_device`` -> ``.device`` (get_attn_mask
    hard-codes .cuda(), timm_lai:12 / dis:1001; ``mask.to(attn.get_device())`` gets -1 on CPU);
  * a stub ``timm`` package exposing only the four names the reference imports
    (Attention, Mlp, DropPath, VisionTransformer) -- timm==0.4.5 is not installed.  The
    arithmetic under test (AttentionLAI.forward, PerformerAttention.*) lives entirely in
    the reference files.
Inputs and weights come from ``tests/golden_inputs.py`` (seeded); only outputs are stored.
"""
import collections.abc
import hashlib
import os
import sys
import types

import numpy as np
import torch
import torch.nn as nn

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, 'tests'))
sys.dont_write_bytecode = True
import golden_inputs as gi  # noqa: E402

OUT = os.path.join(REPO, 'tests', 'golden')


def install_shims():
    six = types.ModuleType('torch._six')
    six.container_abcs = collections.abc
    sys.modules['torch._six'] = six
    torch.Tensor.cuda = lambda self, *a, **k: self
    torch.Tensor.get_device = lambda self: self.device

    timm = types.ModuleType('timm')
    models = types.ModuleType('timm.models')
    layers = types.ModuleType('timm.models.layers')
    drop = types.ModuleType('timm.models.layers.drop')
    vt = types.ModuleType('timm.models.vision_transformer')

    class DropPath(nn.Identity):
        def __init__(self, p=0.):
            super().__init__()

    class Attention(nn.Module):
        def __init__(self, dim, num_heads=8, qkv_bias=False, qk_scale=None, attn_drop=0., proj_drop=0.):
            super().__init__()
            self.num_heads = num_heads
            self.scale = qk_scale or (dim // num_heads) ** -0.5
            self.qkv = nn.Linear(dim, dim * 3, bias=qkv_bias)
            self.attn_drop = nn.Dropout(attn_drop)
            self.proj = nn.Linear(dim, dim)
            self.proj_drop = nn.Dropout(proj_drop)

    class Mlp(nn.Module):
        def __init__(self, in_features, hidden_features=None, out_features=None, act_layer=nn.GELU, drop=0.):
            super().__init__()
            self.fc1 = nn.Linear(in_features, hidden_features)
            self.act = act_layer()
            self.fc2 = nn.Linear(hidden_features, out_features or in_features)

        def forward(self, x):
            return self.fc2(self.act(self.fc1(x)))

    class VisionTransformer(nn.Module):  # shell only; never instantiated by the generator
        def __init__(self, *a, **k):
            super().__init__()

    drop.DropPath = DropPath
    vt.Attention, vt.Mlp, vt.VisionTransformer = Attention, Mlp, VisionTransformer
    timm.models, models.layers, models.vision_transformer, layers.drop = models, layers, vt, drop
    for name, mod in [('timm', timm), ('timm.models', models), ('timm.models.layers', layers),
                      ('timm.models.layers.drop', drop), ('timm.models.vision_transformer', vt)]:
        sys.modules[name] = mod
    sys.path.insert(0, '/root/reference')
    return Attention


def np_(t):
    return t.detach().cpu().numpy()


def run_with_grads(fn, x, params, gout_seed, step):
    """y = fn(x); loss = sum(y * g); returns y, dx, {name: dparam} (all subsampled by step)."""
    x = x.clone().requires_grad_(True)
    y = fn(x)
    g = gi.randn(gout_seed, 'gout', *y.shape)
    (y * g).sum().backward()
    res = {'y': np_(gi.sub(y, step)), 'dx': np_(gi.sub(x.grad, step))}
    for n, p in params:
        if p.grad is not None:
            res['d_' + n] = np_(gi.sub(p.grad, step))
            p.grad = None
    return res


def gen_masks():
    from timm_classes.timm_locality_aware_transformer import get_attn_mask
    from disentangled_transformer.disentangled_transformer import get_attn_mask as get_attn_mask_dis
    out = {}
    for N in (65, 197, 577):
        for w in (8, 12, 18, 20):
            m = get_attn_mask(N, w)
            assert torch.equal(m, get_attn_mask_dis(N, w))
            assert m.shape == (1, 1, N, N) and m.dtype == torch.float32
            bits = np_(m).reshape(N, N).astype(np.uint8)
            out[f'bits_{N}_{w}'] = np.packbits(bits, axis=None)
            out[f'sha_{N}_{w}'] = np.frombuffer(
                hashlib.sha256(np_(m).tobytes()).digest(), dtype=np.uint8)
    np.savez_compressed(os.path.join(OUT, 'lai_masks.npz'), **out)


def gen_lai_attention():
    from timm_classes.timm_locality_aware_transformer import AttentionLAI
    out = {}
    for name, B, N, D, H, step in gi.ATTN_SHAPES:
        mod = AttentionLAI(D, num_heads=H, qkv_bias=True, mask_epochs=gi.LAI_MASK_EPOCHS)
        mod.load_state_dict(gi.fill_state_dict(mod.state_dict(), seed=11))
        x = gi.randn(11, 'x_' + name, B, N, D)
        for ep in gi.LAI_STAGE_EPOCHS:
            r = run_with_grads(lambda t: mod(t, ep), x, list(mod.named_parameters()), 12, step)
            for k, v in r.items():
                out[f'{name}_ep{ep}_{k}'] = v
    np.savez_compressed(os.path.join(OUT, 'lai_attention.npz'), **out)


def gen_dis_attention():
    from disentangled_transformer.disentangled_transformer import (
        DisentangledSelfAttention, DisentangledLAISelfAttention)
    from utils.ops import build_relative_position
    out = {}
    for name, B, N, D, H, step in gi.ATTN_SHAPES:
        S = D  # train.py:211 max_position_embeddings = embed_dim
        rel0 = gi.randn(21, 'rel_' + name, 2 * S, D)
        x = gi.randn(21, 'x_' + name, B, N, D)
        relpos = build_relative_position(N, N)
        for kind in ('dis', 'dislai'):
            if kind == 'dis':
                mod = DisentangledSelfAttention(D, H, dropout=0.0, max_position_embeddings=S)
                epochs = [None]
            else:
                mod = DisentangledLAISelfAttention(D, H, dropout=0.0, max_position_embeddings=S,
                                                   mask_epochs=gi.LAI_MASK_EPOCHS)
                epochs = gi.LAI_STAGE_EPOCHS
            mod.load_state_dict(gi.fill_state_dict(mod.state_dict(), seed=22))
            for ep in epochs:
                rel = rel0.clone().requires_grad_(True)
                kw = {} if ep is None else {'epoch': ep}
                logits = {}

                def fn(t):
                    o = mod(t, t, t, relative_pos=relpos, rel_embeddings=rel, **kw)
                    logits['l'] = o['attention_logits']
                    return o['hidden_states']
                r = run_with_grads(fn, x, list(mod.named_parameters()), 23, step)
                r['d_rel'] = np_(gi.sub(rel.grad, step))
                r['logits'] = np_(gi.sub(logits['l'], step))
                for k, v in r.items():
                    out[f'{name}_{kind}_ep{ep}_{k}'] = v
    np.savez_compressed(os.path.join(OUT, 'dis_attention.npz'), **out)


def gen_dis_model():
    """2-layer Dis / DisLAI encoder inside model.VisionTransformer (N=65, D=128), fwd + grads,
    and one DisentangledEncoderLayer on its own."""
    import model as M
    from disentangled_transformer.disentangled_transformer import DisentangledTransformer
    out = {}
    for kind in ('dis', 'dislai'):
        kw = dict(lai=True, mask_epochs=gi.LAI_MASK_EPOCHS) if kind == 'dislai' else {}
        tr = DisentangledTransformer(d_model=128, dropout=0.0, nhead=4, dim_feedforward=256,
                                     num_encoder_layers=2, normalize_before=True,
                                     pos_att_type='c2p|p2c', max_position_embeddings=128,
                                     relative_attention=True, **kw)
        net = M.VisionTransformer(img_size=32, patch_size=4, in_chans=3, num_classes=10, transformer=tr)
        net.load_state_dict(gi.fill_state_dict(net.state_dict(), seed=31))
        img = gi.randn(31, 'img', 2, 3, 32, 32)
        for ep in ([None] if kind == 'dis' else [0, 30, 40]):
            fn = (lambda t: net(t)) if ep is None else (lambda t: net(t, ep))
            r = run_with_grads(fn, img, list(net.named_parameters()), 32, 1)
            # keep the fixture small: logits, d(img) and a few representative grads
            keep = ['y', 'dx', 'd_transformer.encoder.rel_embeddings.weight',
                    'd_transformer.encoder.layers.0.dis_attn.query_proj.weight',
                    'd_transformer.encoder.layers.1.dis_attn.key_proj.bias',
                    'd_transformer.encoder.layers.0.dense_intermediate.weight',
                    'd_transformer.encoder.LayerNorm.weight', 'd_patch_embed.proj.weight',
                    'd_cls_token', 'd_head.weight']
            for k in keep:
                out[f'{kind}_ep{ep}_{k}'] = r[k]
            out[f'{kind}_ep{ep}_nograd'] = np.array(
                [n for n, p in net.named_parameters() if ('d_' + n) not in r], dtype='U')
    np.savez_compressed(os.path.join(OUT, 'dis_model.npz'), **out)


def gen_performer(Attention):
    from timm_classes.timm_vision_performer import PerformerAttention
    import model as M
    out = {}
    B, N, D, H = 2, 145, 256, 4   # dh 64
    x = gi.randn(41, 'x', B, N, D)
    for m in (16, 64):
        for kernel in ('softmax', 'relu'):
            base = Attention(D, num_heads=H, qkv_bias=True)
            mod = PerformerAttention(base, D, num_heads=H, qkv_bias=True, n_orf=m, kernel=kernel)
            mod.load_state_dict(gi.fill_state_dict(mod.state_dict(), seed=42))
            torch.manual_seed(43)
            mod.w = mod.gaussian_orthogonal_random_matrix(m, D // H)
            r = run_with_grads(lambda t: mod(t), x, list(mod.named_parameters()), 44, 1)
            r['w'] = np_(mod.w)
            for k, v in r.items():
                out[f'timm_{kernel}_m{m}_{k}'] = v
    # model.py variant: [L,B,E] layout, softmax ratio m^-1/2
    L, Bm, E = 65, 2, 128
    xm = gi.randn(45, 'xm', L, Bm, E)
    for kernel in ('softmax', 'relu'):
        mod = M.MultiheadAttentionPerformer(E, 4, dropout=0.0, num_orf=32, kernel=kernel)
        mod.load_state_dict(gi.fill_state_dict(mod.state_dict(), seed=46))
        torch.manual_seed(47)
        mod.w = mod.gaussian_orthogonal_random_matrix(32, E // 4)
        r = run_with_grads(lambda t: mod(t, t, t)[0], xm, list(mod.named_parameters()), 48, 1)
        r['w'] = np_(mod.w)
        for k, v in r.items():
            out[f'mha_{kernel}_{k}'] = v
    np.savez_compressed(os.path.join(OUT, 'performer.npz'), **out)


def gen_dis_performer():
    """DisentangledSelfAttentionPerformer (dis :431) alone, one DisentangledEncoderLayer(performers=True),
    and a 2-layer model through model.VisionTransformer (train.py:215-234's wiring); ``w`` drawn by the
    reference's own generator under a fixed seed and stored."""
    from disentangled_transformer.disentangled_transformer import (
        DisentangledSelfAttentionPerformer, DisentangledEncoderLayer, DisentangledTransformer)
    import model as M
    out = {}
    for name, B, N, D, H, m in (('a', 2, 65, 128, 4, 16), ('b', 2, 145, 256, 4, 64), ('c', 3, 50, 96, 2, 40)):
        x = gi.randn(61, 'x_' + name, B, N, D)
        pos0 = gi.randn(61, 'pos_' + name, N, 1, D, scale=0.5)
        for kernel in ('softmax', 'relu'):
            mod = DisentangledSelfAttentionPerformer(D, H, dropout=0.0, relative_attention=True,
                                                     max_position_embeddings=D, norf=m, kernel=kernel)
            mod.load_state_dict(gi.fill_state_dict(mod.state_dict(), seed=62))
            torch.manual_seed(63)
            mod.w = mod.gaussian_orthogonal_random_matrix(m, D // H)
            pos = pos0.clone().requires_grad_(True)
            r = run_with_grads(lambda t: mod(t, t, t, pos_embed=pos)['hidden_states'], x,
                               list(mod.named_parameters()), 64, 1)
            r['d_pos'] = np_(gi.sub(pos.grad, 1))
            r['w'] = np_(mod.w)
            for k, v in r.items():
                out[f'attn_{name}_{kernel}_{k}'] = v
    # one encoder layer
    B, N, D, H, m = 2, 65, 128, 4, 32
    x = gi.randn(65, 'x_layer', B, N, D)
    pos0 = gi.randn(65, 'pos_layer', N, 1, D, scale=0.5)
    for kernel in ('softmax', 'relu'):
        lay = DisentangledEncoderLayer(d_model=D, nhead=H, dropout=0.0, relative_attention=True,
                                       max_position_embeddings=D, dim_feedforward=256, performers=True,
                                       norf=m, kernel=kernel)
        lay.load_state_dict(gi.fill_state_dict(lay.state_dict(), seed=66))
        torch.manual_seed(67)
        lay.dis_attn.w = lay.dis_attn.gaussian_orthogonal_random_matrix(m, D // H)
        pos = pos0.clone().requires_grad_(True)
        r = run_with_grads(lambda t: lay(t, pos_embed=pos), x, list(lay.named_parameters()), 68, 1)
        r['d_pos'] = np_(gi.sub(pos.grad, 1))
        r['w'] = np_(lay.dis_attn.w)
        for k, v in r.items():
            out[f'layer_{kernel}_{k}'] = v
    # 2-layer model (N = 65, D = 128)
    tr = DisentangledTransformer(d_model=128, dropout=0.0, nhead=4, dim_feedforward=256, num_encoder_layers=2,
                                 normalize_before=True, pos_att_type='c2p|p2c', max_position_embeddings=128,
                                 relative_attention=True, performers=True, norf=16, kernel='softmax')
    net = M.VisionTransformer(img_size=32, patch_size=4, in_chans=3, num_classes=10, transformer=tr)
    net.load_state_dict(gi.fill_state_dict(net.state_dict(), seed=71))
    torch.manual_seed(72)
    for i, lay in enumerate(tr.encoder.layers):
        lay.dis_attn.w = lay.dis_attn.gaussian_orthogonal_random_matrix(16, 32)
        out[f'model_w{i}'] = np_(lay.dis_attn.w)
    img = gi.randn(71, 'img', 2, 3, 32, 32)
    r = run_with_grads(lambda t: net(t), img, list(net.named_parameters()), 73, 1)
    keep = ['y', 'dx', 'd_pos_embed', 'd_transformer.encoder.layers.0.dis_attn.query_proj.weight',
            'd_transformer.encoder.layers.1.dis_attn.pos_key_proj.weight',
            'd_transformer.encoder.layers.0.dis_attn.pos_query_proj.bias',
            'd_transformer.encoder.layers.1.dense_out.weight', 'd_patch_embed.proj.weight', 'd_cls_token']
    for k in keep:
        out['model_' + k] = r[k]
    out['model_nograd'] = np.array([n for n, p in net.named_parameters() if ('d_' + n) not in r], dtype='U')
    np.savez_compressed(os.path.join(OUT, 'dis_performer.npz'), **out)


def gen_config1():
    """BASELINE config 1: regular ViT, CIFAR 32x32, patch 4, 6 layers, dim 256, 8 heads,
    one fwd/bwd/AdamW step (train.py:530-554 restated: timm LabelSmoothingCrossEntropy(0.1),
    AdamW lr 1e-4).  Dropout 0 (parity is defined at p=0)."""
    import model as M
    tr = M.Transformer(d_model=256, dropout=0.0, nhead=8, dim_feedforward=1024,
                       num_encoder_layers=6, normalize_before=True)
    net = M.VisionTransformer(img_size=32, patch_size=4, in_chans=3, num_classes=10, transformer=tr)
    net.load_state_dict(gi.fill_state_dict(net.state_dict(), seed=51))
    img = gi.randn(51, 'img', 8, 3, 32, 32)
    tgt = torch.randint(0, 10, (8,), generator=torch.Generator().manual_seed(52))
    opt = torch.optim.AdamW(net.parameters(), lr=1e-4)
    logits = net(img)
    logp = torch.log_softmax(logits, -1)
    loss = (0.9 * (-logp.gather(-1, tgt[:, None]).squeeze(1)) + 0.1 * (-logp.mean(-1))).mean()
    opt.zero_grad()
    loss.backward()
    out = {'logits': np_(logits), 'loss': np_(loss), 'target': np_(tgt)}
    for n in ('head.weight', 'cls_token', 'patch_embed.proj.weight',
              'transformer.encoder.layers.0.self_attn.in_proj_weight',
              'transformer.encoder.layers.5.linear1.weight', 'transformer.encoder.norm.weight'):
        out['d_' + n] = np_(gi.sub(dict(net.named_parameters())[n].grad, 5))
    opt.step()
    out['head.weight_after'] = np_(net.head.weight)
    out['loss_after'] = np_(torch.nn.functional.cross_entropy(net(img), tgt))
    np.savez_compressed(os.path.join(OUT, 'config1_step.npz'), **out)


def main():
    os.makedirs(OUT, exist_ok=True)
    Attention = install_shims()
    torch.set_num_threads(8)
    gens = {'masks': gen_masks, 'lai_attention': gen_lai_attention, 'dis_attention': gen_dis_attention,
            'dis_model': gen_dis_model, 'performer': lambda: gen_performer(Attention),
            'dis_performer': gen_dis_performer, 'config1': gen_config1}
    for name in (sys.argv[1:] or list(gens)):      # optional: only the named fixtures
        gens[name]()
    for f in sorted(os.listdir(OUT)):
        print(f, os.path.getsize(os.path.join(OUT, f)))


if __name__ == '__main__':
    main()
