"""CPU oracle: a plain-PyTorch restatement of the reference's ViT encoder hot path.

TEST INFRASTRUCTURE ONLY.  Nothing in ``vit_amd/`` may import this file; only
``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg do,
and there only as the checker / the timed CPU baseline -- never as the product path.

Every class below restates (in closed form, not line by line) one piece of the
reference at ``/root/reference`` and cites the file:line it follows.  The state_dict
keys are the reference's, so one set of weights loads into the reference module,
this oracle and the MI355X module alike.

Pinning: ``tests/golden/*.npz`` hold outputs of the reference itself, produced in the
build container by ``tools/gen_goldens.py`` (imports the reference with three shims,
see that file); ``tests/test_oracle_vs_golden.py`` checks this file against them.
The timm-side shells (``Mlp``, ``Block`` pre-LN wiring, ``VisionTransformer`` shell)
come from timm==0.4.5, which is not installed here: their semantics are restated
from the published timm 0.4.5 source and are "parity unpinned" beyond the
attention module boundary.
"""
import math
from functools import partial

import torch
import torch.nn as nn
import torch.nn.functional as F


# --------------------------------------------------------------------------- M1 / M2
def lai_band_mask(N, w):
    """float32 [1,1,N,N], 1 iff |i-j| <= w.  Closed form of ``get_attn_mask``
    (timm_classes/timm_locality_aware_transformer.py:11-21, dup
    disentangled_transformer.py:1000-1010)."""
    idx = torch.arange(N)
    return ((idx[:, None] - idx[None, :]).abs() <= w).to(torch.float32)[None, None]


def lai_stage_width(epoch, mask_epochs):
    """Band half-width for ``epoch`` or None (= no mask).  Stage schedule of
    AttentionLAI.forward (timm_locality_aware_transformer.py:52-72) and
    DisentangledLAISelfAttention.forward (disentangled_transformer.py:893-911)."""
    if mask_epochs is None:
        return None
    if epoch >= mask_epochs[-1]:
        return None
    if epoch < mask_epochs[-4]:
        return 8
    if epoch < mask_epochs[-3]:
        return 12
    if epoch < mask_epochs[-2]:
        return 18
    return 20


MASK_FILL = -1e9  # timm_locality_aware_transformer.py:70, disentangled_transformer.py:911


# --------------------------------------------------------------------------- dropout
# The reference's nn.Dropout sites draw from torch's generator; the oracle restates them as x * keep / (1 - p) with the
# keep mask SUPPLIED: tests install DROPOUT = callable(p, shape) -> 0/1 tensor and hand it the very masks the HIP
# kernels applied (vit_amd.ops.dropout_mask), in the order the reference reaches its sites.  None (default): p = 0.
# Masks are indexed over batch-major tensors ([B, N, D] / [B, H, N, N] / [rows, D]), the layout of the kernels.
DROPOUT = None


def _drop(x, p):
    if DROPOUT is None or p is None or p <= 0:
        return x
    keep = DROPOUT(float(p), tuple(x.shape))
    return x * keep.to(dtype=x.dtype, device=x.device) / (1.0 - p)


def _drop_path(x, p):
    """timm 0.4.5 layers/drop.py ``drop_path`` (the DropPath of timm_locality_aware_transformer.py:94; timm is a
    third-party dependency that is not under /root/reference: published algorithm restated, parity unpinned):
    x / (1 - p) * floor(1 - p + rand(B, 1, ..., 1)) -- one keep decision per sample, supplied like the dropout masks
    (DROPOUT(p, (B,)))."""
    if DROPOUT is None or p is None or p <= 0:
        return x
    keep = DROPOUT(float(p), (x.shape[0],)).to(dtype=x.dtype, device=x.device)
    return x / (1.0 - p) * keep.view(-1, *([1] * (x.dim() - 1)))


# --------------------------------------------------------------------------- A1
class RefAttentionLAI(nn.Module):
    """timm_classes/timm_locality_aware_transformer.py:25-79."""

    def __init__(self, dim, num_heads=8, qkv_bias=False, qk_scale=None, attn_drop=0.,
                 proj_drop=0, mask_epochs=None):
        super().__init__()
        self.num_heads = num_heads
        self.scale = qk_scale or (dim // num_heads) ** -0.5
        self.qkv = nn.Linear(dim, dim * 3, bias=qkv_bias)
        self.proj = nn.Linear(dim, dim)
        self.mask_epochs = mask_epochs
        self.p_attn, self.p_proj = attn_drop, proj_drop

    def forward(self, x, epoch):
        B, N, C = x.shape
        H = self.num_heads
        qkv = self.qkv(x).view(B, N, 3, H, C // H)
        q, k, v = (qkv[:, :, t].transpose(1, 2) for t in range(3))  # [B,H,N,dh]
        s = torch.matmul(q, k.transpose(-1, -2)) * self.scale
        w = lai_stage_width(epoch, self.mask_epochs)
        if w is not None:
            s = s.masked_fill(lai_band_mask(N, w).to(s.device) == 0, MASK_FILL)
        p = _drop(s.softmax(dim=-1), self.p_attn)                    # attn_drop (:74)
        y = torch.matmul(p, v).transpose(1, 2).reshape(B, N, C)
        return _drop(self.proj(y), self.p_proj)                       # proj_drop (:78)


class RefMlp(nn.Module):
    """timm 0.4.5 ``Mlp``: fc1 -> GELU(erf) -> drop -> fc2 -> drop (parity unpinned)."""

    def __init__(self, in_features, hidden_features, drop=0.):
        super().__init__()
        self.fc1 = nn.Linear(in_features, hidden_features)
        self.fc2 = nn.Linear(hidden_features, in_features)
        self.p = drop

    def forward(self, x):
        return _drop(self.fc2(_drop(F.gelu(self.fc1(x)), self.p)), self.p)


# --------------------------------------------------------------------------- A2
class RefBlockLAI(nn.Module):
    """timm_classes/timm_locality_aware_transformer.py:81-99 (pre-LN residual block)."""

    def __init__(self, dim, num_heads, mlp_ratio=4., qkv_bias=False, qk_scale=None,
                 norm_layer=nn.LayerNorm, mask_epochs=None, drop=0., attn_drop=0., drop_path=0.):
        super().__init__()
        self.p_path = drop_path
        self.norm1 = norm_layer(dim)
        self.attn = RefAttentionLAI(dim, num_heads, qkv_bias, qk_scale, attn_drop=attn_drop, proj_drop=drop,
                                    mask_epochs=mask_epochs)
        self.norm2 = norm_layer(dim)
        self.mlp = RefMlp(dim, int(dim * mlp_ratio), drop=drop)

    def forward(self, x, epoch):
        x = x + _drop_path(self.attn(self.norm1(x), epoch), self.p_path)       # :97
        return x + _drop_path(self.mlp(self.norm2(x)), self.p_path)            # :98


class RefPatchEmbed(nn.Module):
    """model.py:79-101 / timm 0.4.5 PatchEmbed: Conv2d(k=s=P) -> flatten -> transpose."""

    def __init__(self, img_size=224, patch_size=16, in_chans=3, embed_dim=768):
        super().__init__()
        self.img_size = (img_size, img_size)
        self.patch_size = (patch_size, patch_size)
        self.num_patches = (img_size // patch_size) ** 2
        self.proj = nn.Conv2d(in_chans, embed_dim, kernel_size=patch_size, stride=patch_size)

    def forward(self, x):
        B, C, H, W = x.shape
        assert H == self.img_size[0] and W == self.img_size[1], \
            f"Input image size ({H}*{W}) doesn't match model ({self.img_size[0]}*{self.img_size[1]})."
        return self.proj(x).flatten(2).transpose(1, 2)


# --------------------------------------------------------------------------- A3
class RefVisionTransformerLAI(nn.Module):
    """timm_classes/timm_locality_aware_transformer.py:102-151 on the timm 0.4.5
    VisionTransformer shell (cls_token, pos_embed, blocks, norm eps 1e-6, head)."""

    def __init__(self, img_size=224, patch_size=16, in_chans=3, num_classes=1000, embed_dim=768,
                 depth=12, num_heads=12, mlp_ratio=4., qkv_bias=True, qk_scale=None, mask_epochs=None,
                 drop_rate=0., attn_drop_rate=0., drop_path_rate=0.):
        super().__init__()
        self.p_pos = drop_rate
        dpr = [v.item() for v in torch.linspace(0, drop_path_rate, depth)]       # stochastic depth decay rule (:124)
        norm_layer = partial(nn.LayerNorm, eps=1e-6)
        self.patch_embed = RefPatchEmbed(img_size, patch_size, in_chans, embed_dim)
        self.cls_token = nn.Parameter(torch.zeros(1, 1, embed_dim))
        self.pos_embed = nn.Parameter(torch.zeros(1, self.patch_embed.num_patches + 1, embed_dim))
        self.blocks = nn.ModuleList([
            RefBlockLAI(embed_dim, num_heads, mlp_ratio, qkv_bias, qk_scale, norm_layer, mask_epochs,
                        drop=drop_rate, attn_drop=attn_drop_rate, drop_path=dpr[i])
            for i in range(depth)])
        self.norm = norm_layer(embed_dim)
        self.head = nn.Linear(embed_dim, num_classes)
        nn.init.trunc_normal_(self.pos_embed, std=.02)
        nn.init.trunc_normal_(self.cls_token, std=.02)
        for m in self.modules():
            if isinstance(m, nn.Linear):
                nn.init.trunc_normal_(m.weight, std=.02)
                if m.bias is not None:
                    nn.init.zeros_(m.bias)

    def forward(self, x, epoch):
        B = x.shape[0]
        x = self.patch_embed(x)
        x = _drop(torch.cat((self.cls_token.expand(B, -1, -1), x), dim=1) + self.pos_embed, self.p_pos)   # pos_drop (:139)
        for blk in self.blocks:
            x = blk(x, epoch)
        return self.head(self.norm(x)[:, 0])


# --------------------------------------------------------------------------- D1 / D2
def dis_delta_index(N, S):
    """delta(i,j) = clamp(i - j + S, 0, 2S-1); utils/ops.py:61-70 +
    disentangled_transformer.py:392,405 (c2p and p2c resolve to the same index)."""
    idx = torch.arange(N)
    return (idx[:, None] - idx[None, :] + S).clamp(0, 2 * S - 1)


class RefDisentangledSelfAttention(nn.Module):
    """DisentangledSelfAttention (disentangled_transformer.py:248-426) and, with
    ``mask_epochs`` given, DisentangledLAISelfAttention (:788-998).  share_att_key is
    hard-wired True in the reference (:270), so the relative table is projected with
    the *content* query/key projections; pos_key_proj / pos_query_proj exist only as
    (unused) state_dict entries."""

    def __init__(self, d_model=512, nhead=8, dropout=0.1, relative_attention=True,
                 pos_att_type='c2p|p2c', position_buckets=-1, max_relative_positions=-1,
                 max_position_embeddings=512, mask_epochs=None, lai=False):
        super().__init__()
        self.num_attention_heads = nhead
        self.pos_att_type = [t.strip() for t in pos_att_type.lower().split('|')]
        self.relative_attention = relative_attention
        self.query_proj = nn.Linear(d_model, d_model)
        self.key_proj = nn.Linear(d_model, d_model)
        self.value_proj = nn.Linear(d_model, d_model)
        self.pos_ebd_size = max_relative_positions if max_relative_positions >= 1 else max_position_embeddings
        if relative_attention:
            if 'c2p' in self.pos_att_type or 'p2p' in self.pos_att_type:
                self.pos_key_proj = nn.Linear(d_model, d_model)
            if 'p2c' in self.pos_att_type or 'p2p' in self.pos_att_type:
                self.pos_query_proj = nn.Linear(d_model, d_model)
        self.mask_epochs = mask_epochs
        self.lai = lai
        self.p = dropout          # pos_dropout and self.dropout share it (:286,293); active only with DROPOUT installed

    def forward(self, q, k, value, relative_pos=None, rel_embeddings=None, query=None,
                pos_embed=None, epoch=0):
        B, N, D = q.shape
        H = self.num_attention_heads
        dh = D // H

        def heads(t):
            return t.view(t.shape[0], t.shape[1], H, dh).transpose(1, 2)

        Q, K, V = heads(self.query_proj(q)), heads(self.key_proj(k)), heads(self.value_proj(value))
        n_terms = 1 + ('c2p' in self.pos_att_type) + ('p2c' in self.pos_att_type)
        scale = 1.0 / math.sqrt(dh * n_terms)
        s = torch.matmul(Q, K.transpose(-1, -2))
        if self.relative_attention:
            S = self.pos_ebd_size
            lo, hi = max(S - (N - 1), 0), min(S + (N - 1), 2 * S - 1)
            # pos_dropout (:332,883) is drawn over the whole [2S, D] table upstream; only these rows are ever read, so
            # the mask is drawn over them (same distribution, and the layout the kernels index)
            band = _drop(rel_embeddings[lo:hi + 1], self.p)        # only reachable rows
            idx = (dis_delta_index(N, S) - lo).to(q.device)        # [N,N] into the band
            if 'c2p' in self.pos_att_type:
                Pk = self.key_proj(band).view(-1, H, dh).transpose(0, 1)       # [H,R,dh]
                c2p = torch.matmul(Q, Pk.transpose(-1, -2))                      # [B,H,N,R]
                s = s + torch.gather(c2p, -1, idx.expand(B, H, N, N))
            if 'p2c' in self.pos_att_type:
                Pq = self.query_proj(band).view(-1, H, dh).transpose(0, 1)
                p2c = torch.matmul(K, Pq.transpose(-1, -2))                      # [B,H,N(j),R]
                # score[i,j] += K_j . Pq[delta(i,j)]
                s = s + torch.gather(p2c, -1, idx.t().expand(B, H, N, N)).transpose(-1, -2)
        s = s * scale
        w = lai_stage_width(epoch, self.mask_epochs) if self.lai else None
        if w is not None:
            s = s.masked_fill(lai_band_mask(N, w).to(s.device) == 0, MASK_FILL)
        p = torch.softmax(s, -1)
        ctx = torch.matmul(_drop(p, self.p), V).transpose(1, 2).reshape(B, N, D)      # self.dropout (:343,915)
        return {'hidden_states': ctx, 'attention_probs': p, 'attention_logits': s}


# --------------------------------------------------------------------------- D3
class RefDisentangledEncoderLayer(nn.Module):
    """disentangled_transformer.py:130-246: post-LN BERT block, GELU MLP, eps 1e-5."""

    def __init__(self, d_model=512, nhead=8, dropout=0.1, relative_attention=True,
                 pos_att_type='c2p|p2c', position_buckets=-1, max_relative_positions=-1,
                 max_position_embeddings=512, dim_feedforward=2048, lai=False, mask_epochs=None):
        super().__init__()
        self.dis_attn = RefDisentangledSelfAttention(
            d_model, nhead, dropout, relative_attention, pos_att_type, position_buckets,
            max_relative_positions, max_position_embeddings, mask_epochs=mask_epochs, lai=lai)
        self.embed_size = d_model
        # allocated but never used by the reference (:168-170, :207-210)
        self.query_proj = nn.Linear(d_model, d_model)
        self.key_proj = nn.Linear(d_model, d_model)
        self.value_proj = nn.Linear(d_model, d_model)
        self.dense_AttnOut = nn.Linear(d_model, d_model)
        self.layer_norm_AttnOut = nn.LayerNorm(d_model)
        self.dense_intermediate = nn.Linear(d_model, dim_feedforward)
        self.dense_out = nn.Linear(dim_feedforward, d_model)
        self.layer_norm_out = nn.LayerNorm(d_model)
        self.p = dropout

    def forward(self, src, relative_pos=None, rel_embeddings=None, query=None, pos_embed=None, epoch=None):
        ctx = self.dis_attn(src, src, src, rel_embeddings=rel_embeddings,
                            epoch=0 if epoch is None else epoch)['hidden_states']
        a = self.layer_norm_AttnOut(_drop(self.dense_AttnOut(ctx), self.p) + src)                    # dropout_AttnOut (:189)
        return self.layer_norm_out(_drop(self.dense_out(F.gelu(self.dense_intermediate(a))), self.p) + a)   # dropout_out (:195)


class RefDisentangledTransformer(nn.Module):
    """DisentangledTransformer + DisentangledEncoder (disentangled_transformer.py:15-128).
    The encoder forces max_position_embeddings = d_model (:87) for the table size, while
    the attention modules take the constructor's value for att_span (train.py passes
    embed_dim for both, so they agree there)."""

    def __init__(self, d_model=512, nhead=8, num_encoder_layers=6, dim_feedforward=2048, dropout=0.1,
                 activation="relu", normalize_before=False, return_intermediate_dec=False,
                 pos_att_type='c2p|p2c', relative_attention=False, position_buckets=-1,
                 max_relative_positions=-1, max_position_embeddings=512, performers=False,
                 norf=32, kernel='softmax', lai=False, mask_epochs=None):
        super().__init__()
        assert not performers, "DisentangledSelfAttentionPerformer is out of scope (SURVEY 8f)"
        enc = nn.Module()
        enc.layers = nn.ModuleList([
            RefDisentangledEncoderLayer(d_model, nhead, dropout, relative_attention, pos_att_type,
                                        position_buckets, max_relative_positions,
                                        max_position_embeddings, dim_feedforward, lai, mask_epochs)
            for _ in range(num_encoder_layers)])
        # DisentangledEncoder is built with relative_attention=True regardless (:52, :77)
        mrp = d_model
        enc.rel_embeddings = nn.Embedding(mrp * 2, d_model)
        enc.LayerNorm = nn.LayerNorm(d_model)
        if normalize_before:
            enc.norm = nn.LayerNorm(d_model)
        self.has_norm = normalize_before
        self.encoder = enc
        self.d_model = d_model
        self.nhead = nhead
        # clones of one layer share the same init (:81, _get_clones deep-copies)
        for l in enc.layers[1:]:
            l.load_state_dict(enc.layers[0].state_dict())
        for p in self.parameters():
            if p.dim() > 1:
                nn.init.xavier_uniform_(p)

    def forward(self, src, pos_embed, epoch=None):
        enc = self.encoder
        rel = enc.LayerNorm(enc.rel_embeddings.weight)
        out = src
        for layer in enc.layers:
            out = layer(out, rel_embeddings=rel, epoch=epoch)
        if self.has_norm:
            out = enc.norm(out)
        return out.permute(1, 0, 2)


# --------------------------------------------------------------------------- R1
class RefTransformerEncoderLayer(nn.Module):
    """model.py:158-210 (stock nn.MultiheadAttention, ReLU MLP by default)."""

    def __init__(self, d_model, nhead, dim_feedforward=2048, dropout=0.1, activation="relu",
                 normalize_before=False):
        super().__init__()
        self.self_attn = nn.MultiheadAttention(d_model, nhead, dropout=dropout)
        self.linear1 = nn.Linear(d_model, dim_feedforward)
        self.linear2 = nn.Linear(dim_feedforward, d_model)
        self.norm1 = nn.LayerNorm(d_model)
        self.norm2 = nn.LayerNorm(d_model)
        self.act = {'relu': F.relu, 'gelu': F.gelu}[activation]
        self.normalize_before = normalize_before
        self.p = dropout

    @staticmethod
    def _dropnb(x, p):   # [N,B,D] tensor, mask indexed batch-major
        return _drop(x.transpose(0, 1), p).transpose(0, 1)

    def _attn(self, x):  # [N,B,D]
        N, B, D = x.shape
        H = self.self_attn.num_heads
        dh = D // H
        qkv = F.linear(x, self.self_attn.in_proj_weight, self.self_attn.in_proj_bias)
        q, k, v = (t.reshape(N, B * H, dh).transpose(0, 1) for t in qkv.chunk(3, dim=-1))
        p = _drop(torch.softmax(torch.bmm(q, k.transpose(1, 2)) / math.sqrt(dh), -1), self.p)   # MHA's dropout on the weights
        y = torch.bmm(p, v).transpose(0, 1).reshape(N, B, D)
        return self._dropnb(self.self_attn.out_proj(y), self.p)                                  # dropout1 (:187,200)

    def _ffn(self, x):
        h = self._dropnb(self.act(self.linear1(x)), self.p)                                      # self.dropout (:189,202)
        return self._dropnb(self.linear2(h), self.p)                                             # dropout2 (:190,203)

    def forward(self, src, pos=None):
        if self.normalize_before:
            src = src + self._attn(self.norm1(src))
            return src + self._ffn(self.norm2(src))
        src = self.norm1(src + self._attn(src))
        return self.norm2(src + self._ffn(src))


class RefTransformer(nn.Module):
    """model.py:103-156."""

    def __init__(self, d_model=512, nhead=8, num_encoder_layers=6, dim_feedforward=2048, dropout=0.1,
                 activation="relu", normalize_before=False, return_intermediate_dec=False):
        super().__init__()
        enc = nn.Module()
        enc.layers = nn.ModuleList([
            RefTransformerEncoderLayer(d_model, nhead, dim_feedforward, dropout, activation, normalize_before)
            for _ in range(num_encoder_layers)])
        for l in enc.layers[1:]:
            l.load_state_dict(enc.layers[0].state_dict())
        if normalize_before:
            enc.norm = nn.LayerNorm(d_model)
        self.has_norm = normalize_before
        self.encoder = enc
        self.d_model = d_model
        self.nhead = nhead
        for p in self.parameters():
            if p.dim() > 1:
                nn.init.xavier_uniform_(p)

    def forward(self, src, pos_embed):
        out = src.permute(1, 0, 2)
        for layer in self.encoder.layers:
            out = layer(out)
        if self.has_norm:
            out = self.encoder.norm(out)
        return out


# --------------------------------------------------------------------------- D4
class RefVisionTransformer(nn.Module):
    """model.py:14-68: patch-embed, cat cls, transformer(img, pos_embed[, epoch]),
    LayerNorm(1e-6) on the permuted-back tensor, token 0, head.  pos_embed is handed to
    the transformer, which never adds it (model.py:185,198; dis:121)."""

    def __init__(self, img_size=224, patch_size=16, in_chans=3, num_classes=1000, transformer=None):
        super().__init__()
        D = transformer.d_model
        self.patch_embed = RefPatchEmbed(img_size, patch_size, in_chans, D)
        self.cls_token = nn.Parameter(torch.zeros(1, 1, D))
        self.pos_embed = nn.Parameter(torch.zeros(1, self.patch_embed.num_patches + 1, D))
        self.transformer = transformer
        self.head = nn.Linear(D, num_classes)
        self.norm = nn.LayerNorm(D, eps=1e-6)
        self.out = nn.Linear(num_classes, num_classes)  # unused (model.py:46)

    def forward(self, img, epoch=None):
        B = img.shape[0]
        x = torch.cat((self.cls_token.expand(B, -1, -1), self.patch_embed(img)), dim=1)
        out = self.transformer(x, self.pos_embed) if epoch is None else \
            self.transformer(x, self.pos_embed, epoch=epoch)
        return self.head(self.norm(out.permute(1, 0, 2))[:, 0])


# --------------------------------------------------------------------------- P1 / P2 / P3
def favor_softmax_features(data, w, is_query, ratio, eps=1e-6):
    """softmax_kernel_transformation (timm_vision_performer.py:69-91, model.py:320-342).
    data [..., H, dh] (any leading layout), w [m, dh].  Quirks kept: ``diag`` is the plain
    sum of the features (the square is overwritten, :76-78), data is not pre-scaled, the
    key stabiliser is the max over the WHOLE tensor (:88)."""
    dh = data.shape[-1]
    dd = torch.matmul(data, w.t())
    diag = (data.sum(-1, keepdim=True) / 2.0) * (dh ** -0.5)
    stab = dd.max(dim=-1, keepdim=True)[0] if is_query else dd.max()
    return ratio * (torch.exp(dd - diag - stab) + eps)


def favor_relu_features(data, w, ratio, eps=1e-3):
    """relu_kernel_transformation (timm_vision_performer.py:93-101, model.py:344-352)."""
    return F.relu(ratio * torch.matmul(data, w.t())) + eps


def favor_linear_attention(qp, kp, v):
    """[B,L,H,m],[B,L,H,m],[B,L,H,dh] -> [B,L,H,dh]  (timm_vision_performer.py:50-58)."""
    kv = torch.einsum('blhm,blhd->bhmd', kp, v)
    num = torch.einsum('blhm,bhmd->blhd', qp, kv)
    den = torch.einsum('blhm,bhm->blh', qp, kp.sum(1))
    return num / den.unsqueeze(-1)


def gaussian_orthogonal_random_matrix(m, dh, generator=None):
    """timm_vision_performer.py:103-137, scaling=0: stacked Q^T blocks of QR(randn(dh,dh)),
    rows scaled by the norms of fresh randn(m,dh) rows."""
    blocks, rows = [], 0
    while rows < m:
        q, _ = torch.linalg.qr(torch.randn(dh, dh, generator=generator))
        blocks.append(q.t()[:min(dh, m - rows)])
        rows += dh
    mult = torch.randn(m, dh, generator=generator).norm(dim=1)
    return mult[:, None] * torch.cat(blocks)


class RefPerformerAttention(nn.Module):
    """PerformerAttention (timm_classes/timm_vision_performer.py:10-65); ``w`` is an explicit
    attribute ([m, dh])."""

    def __init__(self, dim, num_heads=8, qkv_bias=False, n_orf=64, kernel='softmax', attn_drop=0., proj_drop=0.):
        super().__init__()
        self.num_heads, self.m, self.kernel = num_heads, int(n_orf), kernel
        self.qkv = nn.Linear(dim, dim * 3, bias=qkv_bias)
        self.proj = nn.Linear(dim, dim)
        self.w = None
        self.p_attn, self.p_proj = attn_drop, proj_drop

    def forward(self, x):
        B, N, C = x.shape
        H = self.num_heads
        if self.w is None:
            self.w = gaussian_orthogonal_random_matrix(self.m, C // H).to(x.device)
        qkv = self.qkv(x).view(B, N, 3, H, C // H)
        q, k, v = qkv[:, :, 0], qkv[:, :, 1], qkv[:, :, 2]      # [B,L,H,dh]
        ratio = self.m ** -0.25
        if self.kernel == 'softmax':
            kp = favor_softmax_features(k, self.w, False, ratio)
            qp = favor_softmax_features(q, self.w, True, ratio)
        else:
            kp = favor_relu_features(k, self.w, ratio)
            qp = favor_relu_features(q, self.w, ratio)
        x = _drop(favor_linear_attention(qp, kp, v).flatten(-2), self.p_attn)      # attn_drop on the OUTPUT (:61)
        return _drop(self.proj(x), self.p_proj)                                    # proj_drop (:64)


class RefDisentangledSelfAttentionPerformer(nn.Module):
    """DisentangledSelfAttentionPerformer (disentangled_transformer/disentangled_transformer.py:431-785),
    live code only (:586-602 + :612-644; the CC/CP/PC block :535-584 is a string literal).
    new_Q = q' + Pq, new_K = k' + Pk; Pq / Pk = query / key feature maps of
    pos_query_proj / pos_key_proj(pos_embed) with batch 1 (so the key-side "whole tensor" max is over
    [1,L,H,m]), broadcast over the batch.  The class's feature maps (:717-749) are the same formulas as
    timm_vision_performer's, ratio m^-1/4 for both kernels.  ``w`` is an explicit attribute ([m, dh])."""

    def __init__(self, d_model=512, nhead=8, norf=32, kernel='softmax'):
        super().__init__()
        self.num_attention_heads, self.m, self.kernel = nhead, int(norf), kernel
        self.query_proj = nn.Linear(d_model, d_model)
        self.key_proj = nn.Linear(d_model, d_model)
        self.value_proj = nn.Linear(d_model, d_model)
        self.pos_key_proj = nn.Linear(d_model, d_model)
        self.pos_query_proj = nn.Linear(d_model, d_model)
        self.w = None

    def features(self, data, is_query):
        ratio = self.m ** -0.25
        if self.kernel == 'softmax':
            return favor_softmax_features(data, self.w, is_query, ratio)
        return favor_relu_features(data, self.w, ratio)

    def forward(self, q, k, value, relative_pos=None, rel_embeddings=None, query=None, pos_embed=None):
        B, L, D = q.shape
        H = self.num_attention_heads
        if self.w is None:
            self.w = gaussian_orthogonal_random_matrix(self.m, D // H).to(q.device)

        def heads(t):
            return t.reshape(t.shape[0], t.shape[1], H, D // H)
        Q, K, V = heads(self.query_proj(q)), heads(self.key_proj(k)), heads(self.value_proj(value))
        kp, qp = self.features(K, False), self.features(Q, True)
        pos = pos_embed.permute(1, 0, 2)                                   # [L,1,D] -> [1,L,D]  (:614)
        Pk = self.features(heads(self.pos_key_proj(pos)), False)           # [1,L,H,m]
        Pq = self.features(heads(self.pos_query_proj(pos)), True)
        ctx = favor_linear_attention(qp + Pq, kp + Pk, V).flatten(-2)
        return {'hidden_states': ctx, 'attention_probs': None, 'attention_logits': None}


class RefDisentangledPerformerLayer(nn.Module):
    """DisentangledEncoderLayer (:130-246) with performers=True: the same post-LN block around the
    attention above; pos_embed is handed through (:213-215)."""

    def __init__(self, d_model, nhead, dim_feedforward, norf=32, kernel='softmax'):
        super().__init__()
        self.dis_attn = RefDisentangledSelfAttentionPerformer(d_model, nhead, norf, kernel)
        self.dense_AttnOut = nn.Linear(d_model, d_model)
        self.layer_norm_AttnOut = nn.LayerNorm(d_model)
        self.dense_intermediate = nn.Linear(d_model, dim_feedforward)
        self.dense_out = nn.Linear(dim_feedforward, d_model)
        self.layer_norm_out = nn.LayerNorm(d_model)

    def forward(self, src, pos_embed):
        ctx = self.dis_attn(src, src, src, pos_embed=pos_embed)['hidden_states']
        a = self.layer_norm_AttnOut(self.dense_AttnOut(ctx) + src)
        return self.layer_norm_out(self.dense_out(F.gelu(self.dense_intermediate(a))) + a)


class RefDisentangledPerformerTransformer(nn.Module):
    """DisentangledTransformer(performers=True) (dis:15-128 around the layer above): pos_embed [1,L,D] is permuted to
    [L,1,D] (:69) and handed to every layer; final LayerNorm when normalize_before; output [L,B,D] (:73).  The
    relative table and its LayerNorm exist as parameters (:97-99) and are never read by this variant."""

    def __init__(self, d_model=512, nhead=8, num_encoder_layers=6, dim_feedforward=2048, normalize_before=False,
                 norf=32, kernel='softmax'):
        super().__init__()
        enc = nn.Module()
        enc.layers = nn.ModuleList([RefDisentangledPerformerLayer(d_model, nhead, dim_feedforward, norf, kernel)
                                    for _ in range(num_encoder_layers)])
        enc.rel_embeddings = nn.Embedding(d_model * 2, d_model)
        enc.LayerNorm = nn.LayerNorm(d_model)
        if normalize_before:
            enc.norm = nn.LayerNorm(d_model)
        self.has_norm = normalize_before
        self.encoder = enc
        self.d_model = d_model
        self.nhead = nhead
        for p in self.parameters():
            if p.dim() > 1:
                nn.init.xavier_uniform_(p)

    def forward(self, src, pos_embed):
        pos = pos_embed.permute(1, 0, 2)
        out = src
        for layer in self.encoder.layers:
            out = layer(out, pos)
        if self.has_norm:
            out = self.encoder.norm(out)
        return out.permute(1, 0, 2)


class RefMultiheadAttentionPerformer(nn.MultiheadAttention):
    """MultiheadAttentionPerformer (model.py:248-388): [L,B,E] layout, packed in_proj,
    softmax-kernel ratio m^-1/2 (:323) but relu-kernel ratio m^-1/4 (:347)."""

    def __init__(self, d_model, nhead, dropout=0.1, num_orf=32, kernel='softmax'):
        super().__init__(d_model, nhead, dropout=dropout)
        self.kernel, self.m, self.w = kernel, int(num_orf), None

    def forward(self, query, key, value, key_padding_mask=None, need_weights=True, attn_mask=None):
        L, B, E = query.shape
        H, dh = self.num_heads, self.head_dim
        if self.w is None:
            self.w = gaussian_orthogonal_random_matrix(self.m, dh).to(query.device)
        Wq, Wk, Wv = self.in_proj_weight.chunk(3)
        bq, bk, bv = self.in_proj_bias.chunk(3)
        q = F.linear(query, Wq, bq).view(L, B, H, dh).transpose(0, 1)
        k = F.linear(key, Wk, bk).view(L, B, H, dh).transpose(0, 1)
        v = F.linear(value, Wv, bv).view(L, B, H, dh).transpose(0, 1)
        if self.kernel == 'softmax':
            r = self.m ** -0.5
            kp, qp = favor_softmax_features(k, self.w, False, r), favor_softmax_features(q, self.w, True, r)
        else:
            r = self.m ** -0.25
            kp, qp = favor_relu_features(k, self.w, r), favor_relu_features(q, self.w, r)
        out = favor_linear_attention(qp, kp, v).transpose(0, 1).reshape(L, B, E)
        return self.out_proj(out), None


# --------------------------------------------------------------------------- T1
def label_smoothing_ce(logits, target, smoothing=0.1):
    """timm 0.4.5 LabelSmoothingCrossEntropy used at train.py:273:
    (1-s)*NLL + s*mean(-log p)."""
    logp = F.log_softmax(logits.float(), dim=-1)
    nll = -logp.gather(-1, target[:, None]).squeeze(1)
    return ((1.0 - smoothing) * nll + smoothing * (-logp.mean(-1))).mean()


def train_step(model, optimizer, img, target, epoch=None, smoothing=0.1):
    """One iteration of train.py:530-554 (clip_grad_norm_ at :545 runs before backward on
    zeroed/None grads and is a no-op, so it is not restated)."""
    out = model(img) if epoch is None else model(img, epoch)
    loss = label_smoothing_ce(out, target, smoothing)
    optimizer.zero_grad()
    loss.backward()
    optimizer.step()
    return loss.detach()
