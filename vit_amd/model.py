"""MI355X drop-ins for the reference's ``model.py``: the ViT shell (VisionTransformer :14,
PatchEmbed :79) and the stock encoder (Transformer :103, TransformerEncoder :137,
TransformerEncoderLayer :158) on ``nn.MultiheadAttention`` parameters.  Same constructor /
forward signatures and state_dict keys; the arithmetic runs on the HIP library.

Token matrices travel as 2-D ``[B*N, D]`` tensors in the compute dtype between fused blocks;
the reference's ``[N, B, D]`` layout appears only at the module boundaries that expose it.
"""
import copy
import math
from functools import partial

import torch
import torch.nn as nn

from . import functional as VF
from .precision import ComputeDtypeMixin
from .timm_classes.timm_locality_aware_transformer import PatchEmbed
from .timm_classes.timm_vision_performer import gaussian_orthogonal_random_matrix, orthogonal_matrix_chunk


class VisionTransformer(ComputeDtypeMixin, nn.Module):
    def __init__(self, img_size=224, patch_size=16, in_chans=3, num_classes=1000, drop_rate=0., attn_drop_rate=0.,
                 drop_path_rate=0., hybrid_backbone=None, norm_layer=None, num_queries=100, transformer=None,
                 shuffle=False):
        super().__init__()
        if shuffle:
            raise NotImplementedError('shuffle transformers are not part of the reference hot path')
        self.num_classes = num_classes
        self.num_features = self.embed_dim = transformer.d_model
        norm_layer = norm_layer or partial(nn.LayerNorm, eps=1e-6)
        self.patch_embed = PatchEmbed(img_size=img_size, patch_size=patch_size, in_chans=in_chans,
                                      embed_dim=self.embed_dim)
        num_patches = self.patch_embed.num_patches
        self.cls_token = nn.Parameter(torch.zeros(1, 1, self.embed_dim))
        self.pos_embed = nn.Parameter(torch.zeros(1, num_patches + 1, self.embed_dim))   # never added (model.py:185,198)
        self.pos_drop = nn.Dropout(p=drop_rate)
        self.num_queries = num_queries
        self.transformer = transformer
        self.head = nn.Linear(self.embed_dim, num_classes)
        self.norm = norm_layer(self.embed_dim)
        self.act = nn.GELU()
        self.out = nn.Linear(num_classes, num_classes)    # unused, as in the reference (:46)

    def forward(self, img, epoch=None):
        norm = getattr(self, 'input_norm', None)
        self.patch_embed.check(img.permute(0, 3, 1, 2) if (img.dtype == torch.uint8 and norm and norm[2] == 'nhwc') else img)
        B = img.shape[0]
        dt = self.compute_dtype
        N = self.patch_embed.num_patches + 1
        D = self.embed_dim
        tok = VF.patch_embed(img, self.patch_embed.proj.weight, self.patch_embed.proj.bias, self.cls_token, None, dt,
                             norm=norm)
        tr = self.transformer
        if hasattr(tr, 'encode2d'):
            if getattr(tr, 'uses_pos_embed', False):      # DisentangledSelfAttentionPerformer reads it (dis :614)
                x2 = tr.encode2d(tok, B, N, epoch, pos_embed=self.pos_embed)
            else:
                x2 = tr.encode2d(tok, B, N, epoch)
            cls = x2.view(B, N, D)[:, 0]          # both trailing LayerNorms are row-wise: class rows only
            if tr.final_norm is not None:
                cls = VF.layer_norm(cls, tr.final_norm.weight, tr.final_norm.bias, tr.final_norm.eps)
        else:                                      # a foreign transformer module: reference calling convention
            x3 = tok.view(B, N, D).float()
            out = tr(x3, self.pos_embed) if epoch is None else tr(x3, self.pos_embed, epoch=epoch)
            cls = out.permute(1, 0, 2)[:, 0].to(dt)
        cls = VF.layer_norm(cls, self.norm.weight, self.norm.bias, self.norm.eps)
        logits = VF.linear(cls, self.head.weight, self.head.bias, out_dtype=torch.float32)
        return logits.squeeze(0) if B == 1 else logits


def _get_clones(module, N):
    return nn.ModuleList([copy.deepcopy(module) for _ in range(N)])


def _get_activation_fn(activation):
    if activation in ("relu", "gelu"):
        return activation
    if activation == "glu":
        raise NotImplementedError('glu feed-forward is not used by the reference hot path')
    raise RuntimeError(F"activation should be relu/gelu, not {activation}.")


class TransformerEncoderLayer(ComputeDtypeMixin, nn.Module):
    """Stock pre-/post-LN block; parameters live in an ``nn.MultiheadAttention`` (in_proj_weight,
    in_proj_bias, out_proj.*) exactly as in the reference so checkpoints interchange."""

    def __init__(self, d_model, nhead, dim_feedforward=2048, dropout=0.1, activation="relu", normalize_before=False):
        super().__init__()
        self.self_attn = nn.MultiheadAttention(d_model, nhead, dropout=dropout)
        self.linear1 = nn.Linear(d_model, dim_feedforward)
        self.dropout = nn.Dropout(dropout)
        self.linear2 = nn.Linear(dim_feedforward, d_model)
        self.norm1 = nn.LayerNorm(d_model)
        self.norm2 = nn.LayerNorm(d_model)
        self.dropout1 = nn.Dropout(dropout)
        self.dropout2 = nn.Dropout(dropout)
        self.activation = _get_activation_fn(activation)
        self.normalize_before = normalize_before

    def forward2d(self, x2, B, N, epoch=None):
        a = self.self_attn
        if hasattr(a, 'forward2d'):      # a FAVOR+ attention swapped in (model.Performer)
            return self._forward2d_generic(x2, B, N)
        H = a.num_heads
        D = x2.shape[1]
        scale = float(D // H) ** -0.5
        if self.training and (a.dropout > 0 or self.dropout.p > 0 or self.dropout1.p > 0 or self.dropout2.p > 0):
            return self._forward2d_dropout(x2, B, N, H, scale)
        if self.norm1.eps != self.norm2.eps:
            raise NotImplementedError('TransformerEncoderLayer: norm1 / norm2 must share eps')
        if self.normalize_before:
            cfg = (B, N, H, self.norm1.eps, scale, None, self.activation)
            return VF.PreLNBlockFn.apply(x2, self.norm1.weight, self.norm1.bias, a.in_proj_weight, a.in_proj_bias,
                                         a.out_proj.weight, a.out_proj.bias, self.norm2.weight, self.norm2.bias,
                                         self.linear1.weight, self.linear1.bias, self.linear2.weight,
                                         self.linear2.bias, cfg)
        w, b = a.in_proj_weight, a.in_proj_bias
        cfg = (B, N, H, self.norm1.eps, scale, None, self.activation)
        return VF.DisLayerFn.apply(x2, None, w[:D], b[:D], w[D:2 * D], b[D:2 * D], w[2 * D:], b[2 * D:],
                                   a.out_proj.weight, a.out_proj.bias, self.norm1.weight, self.norm1.bias,
                                   self.linear1.weight, self.linear1.bias, self.linear2.weight, self.linear2.bias,
                                   self.norm2.weight, self.norm2.bias, cfg)

    def _forward2d_dropout(self, x2, B, N, H, scale):
        """Training with dropout (the reference default 0.1, model.py:160-203): the block as separate autograd nodes.
        Sites in the reference's order: attention probabilities (nn.MultiheadAttention's dropout), dropout1 on the
        projected attention output, self.dropout after the activation, dropout2 on the MLP output."""
        a = self.self_attn
        ln = VF.layer_norm
        D = x2.shape[1]

        def attend(h, res):
            qkv = VF.linear(h, a.in_proj_weight, a.in_proj_bias)
            ctx = VF.attention(qkv, None, B, N, H, D // H, scale, None, p_drop=a.dropout)
            y = VF.linear(ctx, a.out_proj.weight, a.out_proj.bias)
            return VF.dropout(y, self.dropout1.p, res=res)

        def ffn(h, res):
            y = VF.mlp(h, self.linear1.weight, self.linear1.bias, self.linear2.weight, self.linear2.bias,
                       self.activation, p_act=self.dropout.p, p_out=self.dropout2.p)
            return y + res

        if self.normalize_before:
            x2 = attend(ln(x2, self.norm1.weight, self.norm1.bias, self.norm1.eps), x2)
            return ffn(ln(x2, self.norm2.weight, self.norm2.bias, self.norm2.eps), x2)
        x2 = ln(attend(x2, x2), self.norm1.weight, self.norm1.bias, self.norm1.eps)
        return ln(ffn(x2, x2), self.norm2.weight, self.norm2.bias, self.norm2.eps)

    def _forward2d_generic(self, x2, B, N):
        """Block around an attention module that brings its own forward2d (FAVOR+)."""
        a = self.self_attn
        ln = VF.layer_norm
        p1, pa, p2 = (self.dropout1.p, self.dropout.p, self.dropout2.p) if self.training else (0., 0., 0.)
        mlp = lambda h: VF.mlp(h, self.linear1.weight, self.linear1.bias, self.linear2.weight, self.linear2.bias,   # noqa: E731
                               self.activation, p_act=pa, p_out=p2)
        if self.normalize_before:
            x2 = VF.dropout(a.forward2d(ln(x2, self.norm1.weight, self.norm1.bias, self.norm1.eps), B, N), p1, res=x2)
            h = ln(x2, self.norm2.weight, self.norm2.bias, self.norm2.eps)
            return x2 + mlp(h)
        x2 = ln(VF.dropout(a.forward2d(x2, B, N), p1, res=x2), self.norm1.weight, self.norm1.bias, self.norm1.eps)
        return ln(x2 + mlp(x2), self.norm2.weight, self.norm2.bias, self.norm2.eps)

    def forward(self, src, pos=None):
        """[N, B, D] -> [N, B, D] (the reference layout)."""
        N, B, D = src.shape
        x2 = src.permute(1, 0, 2).reshape(B * N, D).to(self.compute_dtype)
        y = self.forward2d(x2, B, N)
        return y.view(B, N, D).permute(1, 0, 2).to(src.dtype)


class TransformerEncoder(ComputeDtypeMixin, nn.Module):
    def __init__(self, encoder_layer, num_layers, norm=None):
        super().__init__()
        self.layers = _get_clones(encoder_layer, num_layers)
        self.num_layers = num_layers
        self.norm = norm

    def encode2d(self, x2, B, N, epoch=None):
        for layer in self.layers:
            x2 = layer.forward2d(x2, B, N)
        return x2

    def forward(self, src, pos=None):
        N, B, D = src.shape
        x2 = src.permute(1, 0, 2).reshape(B * N, D).to(self.compute_dtype)
        x2 = self.encode2d(x2, B, N)
        if self.norm is not None:
            x2 = VF.layer_norm(x2, self.norm.weight, self.norm.bias, self.norm.eps)
        return x2.view(B, N, D).permute(1, 0, 2).to(src.dtype)


class Transformer(ComputeDtypeMixin, nn.Module):
    def __init__(self, d_model=512, nhead=8, num_encoder_layers=6, dim_feedforward=2048, dropout=0.1,
                 activation="relu", normalize_before=False, return_intermediate_dec=False):
        super().__init__()
        encoder_layer = TransformerEncoderLayer(d_model, nhead, dim_feedforward, dropout, activation,
                                                normalize_before)
        encoder_norm = nn.LayerNorm(d_model) if normalize_before else None
        self.encoder = TransformerEncoder(encoder_layer, num_encoder_layers, encoder_norm)
        self._reset_parameters()
        self.d_model = d_model
        self.nhead = nhead

    def _reset_parameters(self):
        for p in self.parameters():
            if p.dim() > 1:
                nn.init.xavier_uniform_(p)

    def encode2d(self, x2, B, N, epoch=None):
        return self.encoder.encode2d(x2, B, N)

    @property
    def final_norm(self):
        return self.encoder.norm

    def forward(self, src, pos_embed):
        """[B, N, D] -> [N, B, D]; pos_embed is accepted and never added (model.py:185,198)."""
        return self.encoder(src.permute(1, 0, 2), pos=None)


class MultiheadAttentionPerformer(ComputeDtypeMixin, nn.MultiheadAttention):
    """FAVOR+ attention on ``nn.MultiheadAttention`` parameters (model.py:248-388): [L, B, E] in,
    (out [L, B, E], None) out.  Differences from PerformerAttention kept as upstream: the softmax
    feature map scales by m^-1/2 (:323), the relu one by m^-1/4 (:347)."""

    def __init__(self, d_model, nhead, dropout=0.1, num_orf=32, kernel='softmax'):
        super().__init__(d_model, nhead, dropout=dropout)
        self.kernel = kernel
        self.m = int(num_orf)
        self.w = None

    gaussian_orthogonal_random_matrix = staticmethod(gaussian_orthogonal_random_matrix)
    orthogonal_matrix_chunk = staticmethod(orthogonal_matrix_chunk)

    def ratio(self):
        return 1.0 / math.sqrt(self.m) if self.kernel == 'softmax' else 1.0 / math.sqrt(math.sqrt(self.m))

    def forward2d(self, x2, B, N):
        """[B*N, E] batch-major rows in compute dtype -> [B*N, E]."""
        if not self._qkv_same_embed_dim:
            raise NotImplementedError('separate q/k/v embedding dims are not used by the reference models')
        if self.w is None:
            self.w = gaussian_orthogonal_random_matrix(self.m, self.head_dim, device=x2.device)
        qkv = VF.linear(x2, self.in_proj_weight, self.in_proj_bias)
        y = VF.favor_attention(qkv, self.w.to(x2.device), B, N, self.num_heads, self.head_dim, self.kernel, self.ratio())
        if self.training and self.dropout > 0:
            # the reference drops the NUMERATOR before dividing by the normaliser (model.py:305); the normaliser is one
            # scalar per (token, head) and is not dropped, so that equals dropping the quotient element-wise
            y = VF.dropout(y, self.dropout)
        return VF.linear(y, self.out_proj.weight, self.out_proj.bias)

    def forward(self, query, key, value, key_padding_mask=None, need_weights=True, attn_mask=None):
        if not (query is key and key is value):
            raise NotImplementedError('MultiheadAttentionPerformer: self-attention (q = k = v) only on the hot path')
        L, B, E = query.shape
        x2 = query.permute(1, 0, 2).reshape(B * L, E).to(self.compute_dtype)
        y = self.forward2d(x2, B, L)
        return y.view(B, L, E).permute(1, 0, 2).to(query.dtype), None


class Performer(Transformer):
    """model.py:391-408: the stock encoder with every self_attn replaced by FAVOR+."""

    def __init__(self, d_model=512, nhead=8, num_encoder_layers=6, dim_feedforward=2048, dropout=0.1,
                 activation="relu", normalize_before=False, return_intermediate_dec=False, num_orf=64, kernel='relu'):
        super().__init__(d_model=d_model, nhead=nhead, num_encoder_layers=num_encoder_layers,
                         dim_feedforward=dim_feedforward, dropout=dropout, activation=activation,
                         normalize_before=normalize_before, return_intermediate_dec=return_intermediate_dec)
        for layer in self.encoder.layers:
            layer.self_attn = MultiheadAttentionPerformer(d_model, nhead, dropout=dropout, num_orf=num_orf,
                                                          kernel=kernel)


def build_transformer(args):
    return Transformer(d_model=args.hidden_dim, dropout=args.dropout, nhead=args.nheads,
                       dim_feedforward=args.dim_feedforward, num_encoder_layers=args.enc_layers,
                       normalize_before=args.pre_norm, return_intermediate_dec=True)
