"""MI355X drop-ins for the reference's DeBERTa-style disentangled encoder.

Class names, constructor / forward signatures and state_dict keys follow
``disentangled_transformer/disentangled_transformer.py`` in the reference
(DisentangledTransformer :15, DisentangledEncoder :75, DisentangledEncoderLayer :130,
DisentangledSelfAttention :248, DisentangledSelfAttentionPerformer :431, DisentangledLAISelfAttention :788).  Parameters the reference
allocates but never uses (pos_key_proj / pos_query_proj, the layer-level query/key/value_proj)
are kept so checkpoints load; they receive no gradient here either.

What runs: with delta = i - j,
    score[b,h,i,j] = (Q_i.K_j + Q_i.Pk[delta] + K_j.Pq[delta]) / sqrt(3 dh)
where Pk / Pq are the LayerNormed relative table projected with the CONTENT key / query
projections (share_att_key = True, reference :270,373-377).  Only the 2N-1 reachable rows of the
table are projected; the kernels index them by delta, so neither relative_pos nor any
[B*H, N, 2S] tensor exists.  The LAI variant adds the band predicate |i-j| <= w in-kernel.
"""
import copy
import math

import torch
import torch.nn as nn

from .. import functional as VF
from ..precision import ComputeDtypeMixin
from ..timm_classes.timm_locality_aware_transformer import get_attn_mask, lai_stage_width  # noqa: F401
from .. import dropout as VD
from ..timm_classes.timm_vision_performer import gaussian_orthogonal_random_matrix, orthogonal_matrix_chunk

_band_idx_cache = {}


def band_rows(rel_embeddings, N, span):
    """Rows of the (LayerNormed) relative table reachable at sequence length N, ordered by
    delta = -(N-1) .. N-1: row r <-> clamp(delta + span, 0, 2 span - 1)
    (reference :392,405 with build_relative_position, utils/ops.py:61-70)."""
    key = (N, span, rel_embeddings.device)
    idx = _band_idx_cache.get(key)
    if idx is None:
        idx = (torch.arange(-(N - 1), N, device=rel_embeddings.device) + span).clamp_(0, 2 * span - 1)
        _band_idx_cache[key] = idx
    return rel_embeddings.index_select(0, idx)


class DisentangledSelfAttention(ComputeDtypeMixin, nn.Module):
    _lai = False

    def __init__(self, d_model=512, nhead=8, dropout=0.1, relative_attention=True, pos_att_type='c2p|p2c',
                 position_buckets=-1, max_relative_positions=-1, max_position_embeddings=512):
        super().__init__()
        self.num_attention_heads = nhead
        self.attention_head_size = int(d_model / nhead)
        self.all_head_size = self.num_attention_heads * self.attention_head_size
        self.pos_att_type = [x.strip() for x in pos_att_type.lower().split('|')]
        self.relative_attention = relative_attention
        self.share_att_key = True
        self.query_proj = nn.Linear(d_model, self.all_head_size, bias=True)
        self.key_proj = nn.Linear(d_model, self.all_head_size, bias=True)
        self.value_proj = nn.Linear(d_model, self.all_head_size, bias=True)
        if self.relative_attention:
            if position_buckets > 0:
                raise NotImplementedError('log-bucketed relative positions are dead code in the reference '
                                          '(position_buckets = -1 at every call site) and are not built')
            if sorted(t for t in self.pos_att_type if t) != ['c2p', 'p2c']:
                raise NotImplementedError("only pos_att_type 'c2p|p2c' (the reference's CLI default) is built")
            self.position_buckets = position_buckets
            self.max_relative_positions = max_relative_positions
            if self.max_relative_positions < 1:
                self.max_relative_positions = max_position_embeddings
            self.pos_ebd_size = self.max_relative_positions
            self.pos_dropout = nn.Dropout(dropout)
            self.pos_key_proj = nn.Linear(d_model, self.all_head_size, bias=True)   # unused, as in the reference
            self.pos_query_proj = nn.Linear(d_model, self.all_head_size)            # unused, as in the reference
        self.dropout = nn.Dropout(dropout)
        self.mask_epochs = None

    def scale(self):
        terms = 1 + (2 if self.relative_attention else 0)
        return 1.0 / math.sqrt(self.attention_head_size * terms)

    def band_for(self, rel_embeddings, N, dtype):
        if not self.relative_attention:
            return None
        if rel_embeddings is None:
            raise ValueError('relative attention needs rel_embeddings')
        return band_rows(rel_embeddings, N, self.pos_ebd_size).to(dtype)

    def _width(self, epoch):
        return lai_stage_width(epoch, self.mask_epochs) if self._lai else None

    def forward(self, q, k, value, relative_pos=None, rel_embeddings=None, query=None, pos_embed=None, epoch=0):
        """Returns {'hidden_states': [B,N,D], 'attention_probs': None, 'attention_logits': None}.
        The probabilities / logits the reference also returns are discarded by its only caller
        (reference :222) and are never materialised here.  ``relative_pos`` must be the default
        i - j grid (it is regenerated from indices in-kernel)."""
        # dropout (training): pos_dropout on the relative rows (reference :332,883), self.dropout on the probabilities
        # (:343,915) -- in that order, one stream each (vit_amd.dropout)
        p_pos = self.pos_dropout.p if (self.training and self.relative_attention) else 0.
        p_attn = self.dropout.p if self.training else 0.
        if relative_pos is not None and relative_pos.dim() not in (2, 3, 4):
            raise ValueError(f'Relative postion ids must be of dim 2 or 3 or 4. {relative_pos.dim()}')
        if query is not None:
            q = query
        B, N, D = q.shape
        if k.shape != q.shape or value.shape != q.shape:
            raise NotImplementedError('cross attention (different q / k lengths) is not on the reference hot path')
        dt = self.compute_dtype
        H = self.num_attention_heads
        band = self.band_for(rel_embeddings, N, dt)
        if band is not None and p_pos > 0:
            band = VF.dropout(band, p_pos)
        w = self._width(epoch)
        qp, kp, vp = self.query_proj, self.key_proj, self.value_proj
        if q is k and k is value:
            x2 = q.reshape(B * N, D).to(dt)
            ctx = VF.DisAttentionFn.apply(x2, band, qp.weight, qp.bias, kp.weight, kp.bias, vp.weight, vp.bias,
                                          (B, N, H, self.scale(), w, VD.next_stream(p_attn)))
        else:
            r2 = lambda t: t.reshape(B * N, D).to(dt)
            qkv = torch.cat([VF.linear(r2(q), qp.weight, qp.bias), VF.linear(r2(k), kp.weight, kp.bias),
                             VF.linear(r2(value), vp.weight, vp.bias)], dim=1)
            tables = None
            if band is not None:
                tables = torch.cat([VF.linear(band, qp.weight, qp.bias), VF.linear(band, kp.weight, kp.bias)], dim=1)
            ctx = VF.attention(qkv, tables, B, N, H, D // H, self.scale(), w, p_drop=p_attn)
        return {'hidden_states': ctx.view(B, N, D).to(q.dtype), 'attention_probs': None, 'attention_logits': None}


class DisentangledLAISelfAttention(DisentangledSelfAttention):
    _lai = True

    def __init__(self, d_model=512, nhead=8, dropout=0.1, relative_attention=True, pos_att_type='c2p|p2c',
                 position_buckets=-1, max_relative_positions=-1, max_position_embeddings=512, mask_epochs=None):
        super().__init__(d_model, nhead, dropout, relative_attention, pos_att_type, position_buckets,
                         max_relative_positions, max_position_embeddings)
        self.mask_epochs = mask_epochs


class DisentangledSelfAttentionPerformer(ComputeDtypeMixin, nn.Module):
    """FAVOR+ with position features (reference :431-785).  What the reference's live
    code computes (:586-602): new_Q = q' + Pq, new_K = k' + Pk, linear attention over them, where
    q' / k' are the random-feature maps of the content projections and Pq / Pk those of
    pos_query_proj / pos_key_proj(pos_embed) (:612-644) -- one [N, H, m] table each, shared by the
    batch.  The feature maps are the class's own (:717-749), quirks kept: projection unscaled, diag
    from the plain sum, key stabiliser = max over the whole tensor, ratio m^-1/4 for both kernels.
    ``w`` ([m, dh], drawn on first use, not in the state_dict) as upstream."""

    def __init__(self, d_model=512, nhead=8, dropout=0.1, relative_attention=False, pos_att_type='c2p|p2c',
                 position_buckets=-1, max_relative_positions=-1, max_position_embeddings=512, norf=32,
                 kernel='softmax'):
        super().__init__()
        self.m = int(norf)
        self.w = None
        self.kernel = kernel
        self.num_attention_heads = nhead
        self.attention_head_size = self.head_dim = int(d_model / nhead)
        self.all_head_size = self.num_attention_heads * self.attention_head_size
        self.pos_att_type = [x.strip() for x in pos_att_type.lower().split('|')]
        self.relative_attention = relative_attention
        self.share_att_key = True
        if kernel not in ('softmax', 'relu'):
            raise ValueError(f'unknown FAVOR+ kernel {kernel!r}')
        if not relative_attention or 'c2p' not in self.pos_att_type or 'p2c' not in self.pos_att_type:
            # the reference leaves Pq / Pk undefined in these cases (:621-644) and fails at :587
            raise NotImplementedError("DisentangledSelfAttentionPerformer needs relative_attention=True and "
                                      "pos_att_type with both 'c2p' and 'p2c' (the reference's only working form)")
        self.query_proj = nn.Linear(d_model, self.all_head_size, bias=True)
        self.key_proj = nn.Linear(d_model, self.all_head_size, bias=True)
        self.value_proj = nn.Linear(d_model, self.all_head_size, bias=True)
        self.position_buckets = position_buckets
        self.max_relative_positions = max_relative_positions
        if self.max_relative_positions < 1:
            self.max_relative_positions = max_position_embeddings
        self.pos_ebd_size = self.max_relative_positions
        if self.position_buckets > 0:
            self.pos_ebd_size = self.position_buckets
        self.pos_dropout = nn.Dropout(dropout)
        self.pos_key_proj = nn.Linear(d_model, self.all_head_size, bias=True)
        self.pos_query_proj = nn.Linear(d_model, self.all_head_size)
        self.dropout = nn.Dropout(dropout)
        self.mask_epochs = None

    gaussian_orthogonal_random_matrix = staticmethod(gaussian_orthogonal_random_matrix)
    orthogonal_matrix_chunk = staticmethod(orthogonal_matrix_chunk)

    def ratio(self):
        return 1.0 / math.sqrt(math.sqrt(self.m))           # reference :720,744

    def attend2d(self, x2, pos2, B, N):
        """x2 [B*N, D], pos2 [N, D] (pos_embed rows), compute dtype -> context [B*N, D]."""
        if pos2 is None:
            raise ValueError('DisentangledSelfAttentionPerformer needs pos_embed (reference :614)')
        if pos2.shape != (N, x2.shape[1]):
            raise ValueError(f'pos_embed has {tuple(pos2.shape)} rows x cols, the sequence needs {(N, x2.shape[1])}')
        if self.w is None:
            self.w = gaussian_orthogonal_random_matrix(self.m, self.head_dim, device=x2.device)
        qkv = VF.fused_linear(x2, self.query_proj, self.key_proj, self.value_proj)
        posqk = VF.fused_linear(pos2, self.pos_query_proj, self.pos_key_proj)
        ctx = VF.favor_pos_attention(qkv, posqk, self.w.to(x2.device), B, N, self.num_attention_heads,
                                     self.attention_head_size, self.kernel, self.ratio())
        if self.training and self.dropout.p > 0:       # context_layer = self.dropout(context_layer) (reference :604)
            ctx = VF.dropout(ctx, self.dropout.p)
        return ctx

    def forward(self, q, k, value, relative_pos=None, rel_embeddings=None, query=None, pos_embed=None):
        """pos_embed: [N, 1, D] as the encoder hands it down (reference :69,614).  relative_pos /
        rel_embeddings are accepted and unused, as upstream (:614 overwrites them)."""
        if query is not None:
            q = query
        if not (q is k and k is value):
            raise NotImplementedError('cross attention is not on the reference hot path (q = k = value = src, :210)')
        B, N, D = q.shape
        dt = self.compute_dtype
        pos2 = None if pos_embed is None else pos_embed.reshape(-1, D).to(dt)
        ctx = self.attend2d(q.reshape(B * N, D).to(dt), pos2, B, N)
        return {'hidden_states': ctx.view(B, N, D).to(q.dtype), 'attention_probs': None, 'attention_logits': None}


class DisentangledEncoderLayer(ComputeDtypeMixin, nn.Module):
    def __init__(self, d_model=512, nhead=8, dropout=0.1, relative_attention=True, pos_att_type='c2p|p2c',
                 position_buckets=-1, max_relative_positions=-1, max_position_embeddings=512, dim_feedforward=2048,
                 performers=False, lai=False, mask_epochs=None, norf=32, kernel='softmax'):
        super().__init__()
        kw = dict(d_model=d_model, nhead=nhead, dropout=dropout, relative_attention=relative_attention,
                  pos_att_type=pos_att_type, position_buckets=position_buckets,
                  max_relative_positions=max_relative_positions, max_position_embeddings=max_position_embeddings)
        self.performers = bool(performers)
        if performers:
            self.dis_attn = DisentangledSelfAttentionPerformer(kernel=kernel, norf=norf, **kw)
        elif lai:
            self.dis_attn = DisentangledLAISelfAttention(mask_epochs=mask_epochs, **kw)
        else:
            self.dis_attn = DisentangledSelfAttention(**kw)
        self.embed_size = d_model
        self.num_attention_heads = nhead
        self.attention_head_size = int(d_model / nhead)
        self.all_head_size = self.num_attention_heads * self.attention_head_size
        # allocated and never used by the reference (:168-170, :207-210); kept for state_dict parity
        self.query_proj = nn.Linear(d_model, self.all_head_size, bias=True)
        self.key_proj = nn.Linear(d_model, self.all_head_size, bias=True)
        self.value_proj = nn.Linear(d_model, self.all_head_size, bias=True)
        self.dense_AttnOut = nn.Linear(d_model, d_model)
        self.layer_norm_AttnOut = nn.LayerNorm(d_model)
        self.dropout_AttnOut = nn.Dropout(dropout)
        self.dense_intermediate = nn.Linear(d_model, dim_feedforward)
        self.act_intermediate = nn.GELU()
        self.dense_out = nn.Linear(dim_feedforward, d_model)
        self.layer_norm_out = nn.LayerNorm(d_model)
        self.dropout_out = nn.Dropout(dropout)

    def forward2d_performer(self, x2, pos2, B, N):
        """The layer with FAVOR+ attention (reference :200-246 with dis_attn = the Performer variant): post-LN,
        residuals in the GEMM epilogues."""
        a = self.dis_attn
        ctx = a.attend2d(x2, pos2, B, N)
        return self._post_attention(ctx, x2)

    def _drop_ps(self):
        """(dropout_AttnOut.p, dropout_out.p) in training mode, else zeros."""
        return (self.dropout_AttnOut.p, self.dropout_out.p) if self.training else (0., 0.)

    def _post_attention(self, ctx, x2):
        """forward_self_output / forward_intermediate / forward_output (reference :183-198) as separate autograd nodes:
        LN(dropout(dense_AttnOut(ctx)) + x), then LN(dropout(dense_out(GELU(dense_intermediate(a)))) + a)."""
        p1, p2 = self._drop_ps()
        n1, n2 = self.layer_norm_AttnOut, self.layer_norm_out
        if p1 > 0:
            s1 = VF.dropout(VF.linear(ctx, self.dense_AttnOut.weight, self.dense_AttnOut.bias), p1, res=x2)
        else:
            s1 = VF.linear_res(ctx, self.dense_AttnOut.weight, self.dense_AttnOut.bias, x2)
        h = VF.layer_norm(s1, n1.weight, n1.bias, n1.eps)
        y = VF.mlp(h, self.dense_intermediate.weight, self.dense_intermediate.bias, self.dense_out.weight,
                   self.dense_out.bias, 'gelu', add_input=True, p_out=p2)
        return VF.layer_norm(y, n2.weight, n2.bias, n2.eps)

    def dropout_active(self):
        a = self.dis_attn
        ps = [a.dropout.p, self.dropout_AttnOut.p, self.dropout_out.p]
        if getattr(a, 'relative_attention', False) and hasattr(a, 'pos_dropout'):
            ps.append(a.pos_dropout.p)
        return self.training and any(p > 0 for p in ps)

    def forward2d(self, x2, band, B, N, epoch, x_ext=False):
        """x2 [B*N, D] and band [2N-1, D] (or None) in compute dtype -> [B*N, D]; one autograd node.
        ``x_ext``: x2 is the output of the previous layer's forward2d with the same band (its buffer
        already has the rows the relative table rides in: functional.DisLayerFn)."""
        a = self.dis_attn
        if self.layer_norm_AttnOut.eps != self.layer_norm_out.eps:
            raise NotImplementedError('DisentangledEncoderLayer: the two LayerNorms must share eps')
        drops = None
        if self.dropout_active():
            # training with dropout (the reference default, train.py:79): one stream per dropout site of the reference, drawn
            # in its order -- table rows, probabilities, attention output, block output -- and handed to the fused node
            p_pos = a.pos_dropout.p if (a.relative_attention and band is not None) else 0.
            p1, p2 = self._drop_ps()
            drops = (VD.next_stream(p_pos), VD.next_stream(a.dropout.p), VD.next_stream(p1), VD.next_stream(p2))
        cfg = (B, N, a.num_attention_heads, self.layer_norm_out.eps, a.scale(), a._width(0 if epoch is None else epoch),
               'gelu', bool(x_ext and band is not None), drops)
        q, k, v = a.query_proj, a.key_proj, a.value_proj
        return VF.DisLayerFn.apply(x2, band, q.weight, q.bias, k.weight, k.bias, v.weight, v.bias,
                                   self.dense_AttnOut.weight, self.dense_AttnOut.bias,
                                   self.layer_norm_AttnOut.weight, self.layer_norm_AttnOut.bias,
                                   self.dense_intermediate.weight, self.dense_intermediate.bias,
                                   self.dense_out.weight, self.dense_out.bias,
                                   self.layer_norm_out.weight, self.layer_norm_out.bias, cfg)

    def forward(self, src, relative_pos=None, rel_embeddings=None, query=None, pos_embed=None, epoch=None):
        if query is not None:
            raise NotImplementedError('query != src is not on the reference hot path')
        B, N, D = src.shape
        dt = self.compute_dtype
        if self.performers:
            if epoch is not None:       # the reference passes it on (:218-220) to a forward() that has no such argument
                raise TypeError("DisentangledSelfAttentionPerformer.forward() got an unexpected keyword argument 'epoch'")
            pos2 = None if pos_embed is None else pos_embed.reshape(-1, D).to(dt)
            y = self.forward2d_performer(src.reshape(B * N, D).to(dt), pos2, B, N)
            return y.view(B, N, D).to(src.dtype)
        band = self.dis_attn.band_for(rel_embeddings, N, dt)
        y = self.forward2d(src.reshape(B * N, D).to(dt), band, B, N, epoch)
        return y.view(B, N, D).to(src.dtype)


def _get_clones(module, N):
    return nn.ModuleList([copy.deepcopy(module) for _ in range(N)])


class DisentangledEncoder(ComputeDtypeMixin, nn.Module):
    def __init__(self, encoder_layer, num_layers, norm=None, relative_attention=True, max_relative_positions=-1,
                 position_buckets=-1, performers=False):
        super().__init__()
        self.layers = _get_clones(encoder_layer, num_layers)
        self.num_layers = num_layers
        self.norm = norm
        self.performers = bool(performers)
        d_model = encoder_layer.embed_size
        max_position_embeddings = d_model   # as the reference (:87)
        self.relative_attention = relative_attention
        if self.relative_attention:
            self.max_relative_positions = max_relative_positions
            if self.max_relative_positions < 1:
                self.max_relative_positions = max_position_embeddings
            self.position_buckets = position_buckets
            pos_ebd_size = self.max_relative_positions * 2
            if self.position_buckets > 0:
                pos_ebd_size = self.position_buckets * 2
            self.rel_embeddings = nn.Embedding(pos_ebd_size, d_model)
            self.LayerNorm = nn.LayerNorm(d_model)

    def get_rel_embedding(self):
        """LayerNorm of the whole relative table, once per forward, shared by every layer
        (reference :102-106); runs on the fp32 LayerNorm kernel."""
        if not self.relative_attention:
            return None
        return VF.layer_norm(self.rel_embeddings.weight, self.LayerNorm.weight, self.LayerNorm.bias, self.LayerNorm.eps)

    def encode2d(self, x2, B, N, epoch, pos_embed=None):
        """Layers only (no final norm) on a 2-D token matrix in compute dtype.  ``pos_embed`` ([1, N, D] or
        [N, 1, D]): read by the Performer variant only (reference :124,614)."""
        dt = self.compute_dtype
        if self.performers:
            if epoch is not None:
                raise TypeError("DisentangledSelfAttentionPerformer.forward() got an unexpected keyword argument 'epoch'")
            # the relative table and its LayerNorm (:117-118) are computed and discarded upstream: not run here
            pos2 = None if pos_embed is None else pos_embed.reshape(-1, x2.shape[1]).to(dt)
            for layer in self.layers:
                x2 = layer.forward2d_performer(x2, pos2, B, N)
            return x2
        rel = self.get_rel_embedding()
        band = None
        for i, layer in enumerate(self.layers):
            if band is None:
                band = layer.dis_attn.band_for(rel, N, dt)   # identical for every (cloned) layer
            x2 = layer.forward2d(x2, band, B, N, epoch, x_ext=i > 0)
        return x2

    def forward(self, src, pos=None, epoch=None):
        B, N, D = src.shape
        dt = self.compute_dtype
        x2 = self.encode2d(src.reshape(B * N, D).to(dt), B, N, epoch, pos_embed=pos)
        if self.norm is not None:
            x2 = VF.layer_norm(x2, self.norm.weight, self.norm.bias, self.norm.eps)
        return x2.view(B, N, D).to(src.dtype)


class DisentangledTransformer(ComputeDtypeMixin, nn.Module):
    def __init__(self, d_model=512, nhead=8, num_encoder_layers=6, dim_feedforward=2048, dropout=0.1,
                 activation="relu", normalize_before=False, return_intermediate_dec=False, pos_att_type='c2p|p2c',
                 relative_attention=False, position_buckets=-1, max_relative_positions=-1,
                 max_position_embeddings=512, performers=False, norf=32, kernel='softmax', lai=False,
                 mask_epochs=None):
        super().__init__()
        encoder_layer = DisentangledEncoderLayer(
            d_model=d_model, nhead=nhead, dropout=dropout, dim_feedforward=dim_feedforward,
            relative_attention=relative_attention, pos_att_type=pos_att_type, position_buckets=position_buckets,
            max_relative_positions=max_relative_positions, max_position_embeddings=max_position_embeddings,
            performers=performers, lai=lai, mask_epochs=mask_epochs, norf=norf, kernel=kernel)
        encoder_norm = nn.LayerNorm(d_model) if normalize_before else None
        self.encoder = DisentangledEncoder(encoder_layer, num_encoder_layers, encoder_norm, performers=performers)
        self._reset_parameters()
        self.d_model = d_model
        self.nhead = nhead

    def _reset_parameters(self):
        for p in self.parameters():
            if p.dim() > 1:
                nn.init.xavier_uniform_(p)

    # protocol used by vit_amd.model.VisionTransformer's fused path
    def encode2d(self, x2, B, N, epoch, pos_embed=None):
        return self.encoder.encode2d(x2, B, N, epoch, pos_embed=pos_embed)

    @property
    def uses_pos_embed(self):
        return self.encoder.performers

    @property
    def final_norm(self):
        return self.encoder.norm

    def forward(self, src, pos_embed, epoch=None):
        """[B,N,D] -> [N,B,D]; ``pos_embed`` ([1,N,D]) is, as in the reference (:121 commented out), never
        added to the tokens; only the Performer variant reads it (:614)."""
        pos = pos_embed.permute(1, 0, 2) if (self.encoder.performers and pos_embed is not None) else None
        memory = self.encoder(src, pos=pos, epoch=epoch)
        return memory.permute(1, 0, 2)
