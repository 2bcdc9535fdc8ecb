"""Training-step harness for the hot path: the per-batch sequence of the reference's
``training()`` loop (train.py:530-554) and the model factory subset of ``build_model``
(train.py:96-277) that north_star names, on synthetic data.

Not restated: datasets, wandb, checkpoints, the eval loop.
"""
import torch

from . import functional as VF
from . import model as M
from .disentangled_transformer.disentangled_transformer import DisentangledTransformer
from .precision import set_compute_dtype
from .timm_classes.timm_locality_aware_transformer import VisionTransformerLAI
from .timm_classes.timm_vision_performer import PerformerAttention

# Presets as the reference names them (README.md:50-54): S/16 = 12 L x 768 (== timm
# vit_base_patch16_224, train.py:121), B/16 = 24 L x 1024 x 16 h, L/16 = 32 L x 1280 x 16 h.
PRESETS = {
    'S': dict(depth=12, dim=768, mlp=3072, heads=12),
    'B': dict(depth=24, dim=1024, mlp=4096, heads=16),
    'L': dict(depth=32, dim=1280, mlp=5120, heads=16),
    'cifar': dict(depth=6, dim=256, mlp=1024, heads=8),
}


def label_smoothing_ce(logits, target, smoothing=0.1):
    """timm 0.4.5 LabelSmoothingCrossEntropy (train.py:273): (1-s) * NLL + s * mean(-log p), on the HIP
    kernels of csrc/loss.hip (forward and backward); logits [B, C] on the GPU."""
    return VF.label_smoothing_ce(logits if logits.dim() == 2 else logits.reshape(-1, logits.shape[-1]), target.reshape(-1),
                                 smoothing)


def build_model(attention, preset='B', img_size=224, patch_size=16, in_chans=3, num_classes=1000,
                mask_epochs=None, dtype=torch.bfloat16, device='cuda', dropout=0.0, **over):
    """Model factory for the --attention variants on the hot path (train.py:97-253):
    'transformer' (regular, pre-LN, ReLU), 'transformer_timm_lai', 'disentangled_attention',
    'disentangled_lai_attention', 'disentangled_performer', 'performer_timm', 'performer'.  ``dropout`` goes where the
    reference's --dropout goes (train.py:79; dropout= of the model.py / disentangled stacks, attn_drop_rate of the timm
    ones, attn_drop of PerformerAttention); the default here is 0 (benchmarks and reference parity are defined at
    p = 0), the reference's is 0.1."""
    p = dict(PRESETS[preset], **{k: v for k, v in over.items() if k in ('depth', 'dim', 'mlp', 'heads')})
    if attention == 'transformer':
        tr = M.Transformer(d_model=p['dim'], dropout=dropout, nhead=p['heads'], dim_feedforward=p['mlp'],
                           num_encoder_layers=p['depth'], normalize_before=True)
        net = M.VisionTransformer(img_size=img_size, patch_size=patch_size, in_chans=in_chans,
                                  num_classes=num_classes, transformer=tr)
    elif attention == 'transformer_timm_lai':
        net = VisionTransformerLAI(img_size=img_size, patch_size=patch_size, in_chans=in_chans,
                                   num_classes=num_classes, embed_dim=p['dim'], depth=p['depth'],
                                   num_heads=p['heads'], mlp_ratio=p['mlp'] / p['dim'], qkv_bias=True,
                                   attn_drop_rate=dropout, mask_epochs=mask_epochs)
    elif attention in ('disentangled_attention', 'disentangled_lai_attention'):
        lai = attention == 'disentangled_lai_attention'
        tr = DisentangledTransformer(d_model=p['dim'], dropout=dropout, nhead=p['heads'], dim_feedforward=p['mlp'],
                                     num_encoder_layers=p['depth'], normalize_before=True, pos_att_type='c2p|p2c',
                                     max_position_embeddings=p['dim'], relative_attention=True, lai=lai,
                                     mask_epochs=mask_epochs if lai else None)
        net = M.VisionTransformer(img_size=img_size, patch_size=patch_size, in_chans=in_chans,
                                  num_classes=num_classes, transformer=tr)
    elif attention == 'disentangled_performer':
        # train.py:215-234: FAVOR+ with position features inside the DeBERTa-style post-LN layers
        tr = DisentangledTransformer(d_model=p['dim'], dropout=dropout, nhead=p['heads'], dim_feedforward=p['mlp'],
                                     num_encoder_layers=p['depth'], normalize_before=True, pos_att_type='c2p|p2c',
                                     max_position_embeddings=p['dim'], relative_attention=True, performers=True,
                                     norf=over.get('num_orf', 16), kernel=over.get('kernel', 'softmax'))
        net = M.VisionTransformer(img_size=img_size, patch_size=patch_size, in_chans=in_chans,
                                  num_classes=num_classes, transformer=tr)
    elif attention == 'performer_timm':
        # train.py:147-165: a timm ViT whose every blk.attn is replaced by PerformerAttention
        net = VisionTransformerLAI(img_size=img_size, patch_size=patch_size, in_chans=in_chans,
                                   num_classes=num_classes, embed_dim=p['dim'], depth=p['depth'],
                                   num_heads=p['heads'], mlp_ratio=p['mlp'] / p['dim'], qkv_bias=True,
                                   attn_drop_rate=dropout, mask_epochs=None)
        for blk in net.blocks:
            blk.attn = PerformerAttention(blk.attn, p['dim'], num_heads=p['heads'], qkv_bias=True, attn_drop=dropout,
                                          n_orf=over.get('num_orf', 64), kernel=over.get('kernel', 'softmax'))
    elif attention == 'performer':
        tr = M.Performer(d_model=p['dim'], dropout=dropout, nhead=p['heads'], dim_feedforward=p['mlp'],
                         num_encoder_layers=p['depth'], normalize_before=True,
                         num_orf=over.get('num_orf', 64), kernel=over.get('kernel', 'softmax'))
        net = M.VisionTransformer(img_size=img_size, patch_size=patch_size, in_chans=in_chans,
                                  num_classes=num_classes, transformer=tr)
    else:
        raise ValueError(f'unknown attention variant {attention!r}')
    net = net.to(device)
    return set_compute_dtype(net, dtype)


def takes_epoch(net):
    tr = getattr(net, 'transformer', None)
    return isinstance(net, VisionTransformerLAI) or \
        (isinstance(tr, DisentangledTransformer) and not tr.uses_pos_embed)    # the Performer variant takes none


def train_step(net, optimizer, img, target, epoch=None, smoothing=0.1, reducer=None):
    """One iteration of train.py:530-554: forward, label-smoothing CE, zero_grad, backward
    (gradient all-reduce overlapped when ``reducer`` is given), optimizer step.  The reference's
    clip_grad_norm_ sits before backward on zeroed grads (train.py:544-546) and is a no-op."""
    out = net(img, epoch) if epoch is not None else net(img)
    loss = label_smoothing_ce(out, target, smoothing)
    if reducer is not None:
        reducer.prepare()          # zeroes the gradient slabs the .grad views live in
    else:
        optimizer.zero_grad(set_to_none=True)
    loss.backward()
    if reducer is not None:
        reducer.finish()
    optimizer.step()
    return loss.detach()
