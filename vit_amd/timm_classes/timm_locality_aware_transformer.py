"""MI355X drop-ins for the reference's LAI (locality-aware-init) ViT modules.

Same class names, constructor / forward signatures and state_dict keys as
``timm_classes/timm_locality_aware_transformer.py`` in the reference (AttentionLAI :25,
BlockLAI :81, VisionTransformerLAI :102, get_attn_mask :11), so ``load_state_dict`` from a timm
0.4.5 ViT keeps working (train.py:194).  timm itself is not a dependency: the few timm 0.4.5
pieces the reference inherits (PatchEmbed, Mlp, the VisionTransformer shell) are restated here.

The compute path is the HIP library (vit_amd.functional); there is no torch fallback.  The LAI
mask is never built as a tensor on the hot path: the kernels evaluate |i - j| <= w on indices.
``get_attn_mask`` exists for parity tests and plotting only.
"""
from functools import partial

import torch
import torch.nn as nn

from .. import functional as VF
from ..precision import ComputeDtypeMixin


def get_attn_mask(N, w, device=None):
    """float32 [1,1,N,N], 1 iff |i-j| <= w (closed form of the reference's row-by-row fill)."""
    idx = torch.arange(N, device=device)
    return ((idx[:, None] - idx[None, :]).abs() <= w).to(torch.float32)[None, None]


def lai_stage_width(epoch, mask_epochs):
    """Band half width for ``epoch`` (None = unmasked): the schedule of AttentionLAI.forward
    (reference :52-72): < me[-4] -> 8, < me[-3] -> 12, < me[-2] -> 18, < me[-1] -> 20."""
    if mask_epochs is None or epoch >= mask_epochs[-1]:
        return None
    if epoch < mask_epochs[-4]:
        return 8
    if epoch < mask_epochs[-3]:
        return 12
    if epoch < mask_epochs[-2]:
        return 18
    return 20


class Mlp(nn.Module):
    """timm 0.4.5 Mlp: fc1 -> act -> drop -> fc2 -> drop (state_dict keys fc1.*, fc2.*)."""

    def __init__(self, in_features, hidden_features=None, out_features=None, act_layer=nn.GELU, drop=0.):
        super().__init__()
        out_features = out_features or in_features
        hidden_features = hidden_features or in_features
        self.fc1 = nn.Linear(in_features, hidden_features)
        self.act = act_layer()
        self.fc2 = nn.Linear(hidden_features, out_features)
        self.drop = nn.Dropout(drop)


class PatchEmbed(nn.Module):
    """timm 0.4.5 PatchEmbed (state_dict key proj.*): Conv2d(k = s = P), run as a GEMM."""

    def __init__(self, img_size=224, patch_size=16, in_chans=3, embed_dim=768):
        super().__init__()
        img_size = (img_size, img_size) if isinstance(img_size, int) else tuple(img_size)
        patch_size = (patch_size, patch_size) if isinstance(patch_size, int) else tuple(patch_size)
        self.img_size = img_size
        self.patch_size = patch_size
        self.num_patches = (img_size[1] // patch_size[1]) * (img_size[0] // patch_size[0])
        self.proj = nn.Conv2d(in_chans, embed_dim, kernel_size=patch_size, stride=patch_size)

    def check(self, x):
        B, C, H, W = x.shape
        assert H == self.img_size[0] and W == self.img_size[1], \
            f"Input image size ({H}*{W}) doesn't match model ({self.img_size[0]}*{self.img_size[1]})."


class AttentionLAI(ComputeDtypeMixin, nn.Module):
    def __init__(self, dim, num_heads=8, qkv_bias=False, qk_scale=None, attn_drop=0., proj_drop=0, mask_epochs=None):
        super().__init__()
        self.num_heads = num_heads
        head_dim = dim // num_heads
        self.scale = qk_scale or head_dim ** -0.5
        self.qkv = nn.Linear(dim, dim * 3, bias=qkv_bias)
        self.attn_drop = nn.Dropout(attn_drop)
        self.proj = nn.Linear(dim, dim)
        self.proj_drop = nn.Dropout(proj_drop)
        self.mask_epochs = mask_epochs

    def forward2d(self, x2, B, N, epoch, res=None):
        """[B*N, C] in compute dtype -> proj_drop(proj(softmax-attention)) [+ res]; attn_drop acts on the probabilities
        inside the attention kernel (reference :74), proj_drop on the projected output (:78)."""
        C = x2.shape[1]
        p_attn = self.attn_drop.p if self.training else 0.
        p_proj = self.proj_drop.p if self.training else 0.
        qkv = VF.linear(x2, self.qkv.weight, self.qkv.bias)
        ctx = VF.attention(qkv, None, B, N, self.num_heads, C // self.num_heads, float(self.scale),
                           lai_stage_width(epoch, self.mask_epochs), p_drop=p_attn)
        if p_proj > 0:
            return VF.dropout(VF.linear(ctx, self.proj.weight, self.proj.bias), p_proj, res=res)
        if res is not None:
            return VF.linear_res(ctx, self.proj.weight, self.proj.bias, res)
        return VF.linear(ctx, self.proj.weight, self.proj.bias)

    def forward(self, x, epoch):
        B, N, C = x.shape
        y = self.forward2d(x.reshape(B * N, C).to(self.compute_dtype), B, N, epoch)
        return y.view(B, N, C).to(x.dtype)


FUSE_SWAPPED = True      # tests switch it off to compare the one-node block with the op-by-op form


class BlockLAI(ComputeDtypeMixin, nn.Module):
    def __init__(self, dim, num_heads, mlp_ratio=4., qkv_bias=False, qk_scale=None, drop=0., attn_drop=0.,
                 drop_path=0., act_layer=nn.GELU, norm_layer=nn.LayerNorm, mask_epochs=None):
        super().__init__()
        if act_layer is not nn.GELU:
            raise NotImplementedError('BlockLAI: only the GELU Mlp of the reference is implemented')
        self.norm1 = norm_layer(dim)
        self.attn = AttentionLAI(dim, num_heads=num_heads, qkv_bias=qkv_bias, qk_scale=qk_scale,
                                 attn_drop=attn_drop, proj_drop=drop, mask_epochs=mask_epochs)
        # DropPath (reference :94) has no parameters or buffers: the rate is kept, the per-sample scaling runs in
        # forward2d (VF.drop_path) when the module trains with drop_path > 0
        self.drop_path_rate = float(drop_path)
        self.drop_path = nn.Identity()
        self.norm2 = norm_layer(dim)
        self.mlp = Mlp(in_features=dim, hidden_features=int(dim * mlp_ratio), act_layer=act_layer, drop=drop)

    def forward2d(self, x2, B, N, epoch):
        """x2: [B*N, C] in compute dtype -> same.  The whole block is one autograd node."""
        a = self.attn
        p_mlp = self.mlp.drop.p if self.training else 0.
        p_path = self.drop_path_rate if self.training else 0.
        if p_path > 0:
            # stochastic depth (reference :94,97-98): both residual branches of a SAMPLE are dropped / rescaled as a
            # whole, decisions drawn after the branch's own dropout sites as in the reference; separate autograd nodes
            h = VF.layer_norm(x2, self.norm1.weight, self.norm1.bias, self.norm1.eps)
            br = a.forward2d(h, B, N, epoch) if isinstance(a, AttentionLAI) else a.forward2d(h, B, N)
            x2 = VF.drop_path(br, p_path, N, res=x2)
            h = VF.layer_norm(x2, self.norm2.weight, self.norm2.bias, self.norm2.eps)
            y = VF.mlp(h, self.mlp.fc1.weight, self.mlp.fc1.bias, self.mlp.fc2.weight, self.mlp.fc2.bias, 'gelu',
                       p_act=p_mlp, p_out=p_mlp)
            return VF.drop_path(y, p_path, N, res=x2)
        if not isinstance(a, AttentionLAI):      # an attention swapped in (FAVOR+, train.py:147-165)
            favor = a.favor_cfg(x2) if hasattr(a, 'favor_cfg') and FUSE_SWAPPED else None
            drops = self.training and (p_mlp > 0 or a.attn_drop.p > 0 or a.proj_drop.p > 0)
            if favor is not None and not drops and self.norm1.eps == self.norm2.eps:
                # the whole block as ONE node, as for AttentionLAI below (residuals in the proj / fc2 epilogues, weight
                # gradients on the second stream): 33.0 -> see DESIGN.md for the 384-px Performer step
                cfg = (B, N, a.num_heads, self.norm1.eps, 0.0, -1, 'gelu', favor)
                return VF.PreLNBlockFn.apply(x2, self.norm1.weight, self.norm1.bias, a.qkv.weight, a.qkv.bias,
                                             a.proj.weight, a.proj.bias, self.norm2.weight, self.norm2.bias,
                                             self.mlp.fc1.weight, self.mlp.fc1.bias, self.mlp.fc2.weight,
                                             self.mlp.fc2.bias, cfg)
            x2 = x2 + a.forward2d(VF.layer_norm(x2, self.norm1.weight, self.norm1.bias, self.norm1.eps), B, N)
            h = VF.layer_norm(x2, self.norm2.weight, self.norm2.bias, self.norm2.eps)
            return x2 + VF.mlp(h, self.mlp.fc1.weight, self.mlp.fc1.bias, self.mlp.fc2.weight, self.mlp.fc2.bias,
                               'gelu', p_act=p_mlp, p_out=p_mlp)
        if self.training and (a.attn_drop.p > 0 or a.proj_drop.p > 0 or p_mlp > 0):
            # training with dropout (train.py:179,191 pass attn_drop_rate = --dropout): the block as separate autograd
            # nodes, dropout sites in the reference's order (probabilities, proj output, Mlp activation, Mlp output)
            x2 = a.forward2d(VF.layer_norm(x2, self.norm1.weight, self.norm1.bias, self.norm1.eps), B, N, epoch, res=x2)
            h = VF.layer_norm(x2, self.norm2.weight, self.norm2.bias, self.norm2.eps)
            y = VF.mlp(h, self.mlp.fc1.weight, self.mlp.fc1.bias, self.mlp.fc2.weight, self.mlp.fc2.bias, 'gelu',
                       p_act=p_mlp, p_out=p_mlp)
            return x2 + y
        if self.norm1.eps != self.norm2.eps:
            raise NotImplementedError('BlockLAI: norm1 / norm2 must share eps')
        cfg = (B, N, a.num_heads, self.norm1.eps, float(a.scale), lai_stage_width(epoch, a.mask_epochs), 'gelu')
        return VF.PreLNBlockFn.apply(x2, self.norm1.weight, self.norm1.bias, a.qkv.weight, a.qkv.bias,
                                     a.proj.weight, a.proj.bias, self.norm2.weight, self.norm2.bias,
                                     self.mlp.fc1.weight, self.mlp.fc1.bias, self.mlp.fc2.weight, self.mlp.fc2.bias,
                                     cfg)

    def forward(self, x, epoch):
        B, N, C = x.shape
        y = self.forward2d(x.reshape(B * N, C).to(self.compute_dtype), B, N, epoch)
        return y.view(B, N, C).to(x.dtype)


class VisionTransformerLAI(ComputeDtypeMixin, nn.Module):
    """timm 0.4.5 VisionTransformer shell (patch_embed, cls_token, pos_embed, blocks, norm, head)
    with BlockLAI blocks; forward(x, epoch) as in the reference (:131-151)."""

    def __init__(self, img_size=224, patch_size=16, in_chans=3, num_classes=1000, embed_dim=768, depth=12,
                 num_heads=12, mlp_ratio=4., qkv_bias=True, qk_scale=None, representation_size=None,
                 drop_rate=0., attn_drop_rate=0., drop_path_rate=0., hybrid_backbone=None, norm_layer=None,
                 mask_epochs=None):
        super().__init__()
        if hybrid_backbone is not None or representation_size is not None:
            raise NotImplementedError('hybrid backbones / pre_logits heads are outside the reference hot path')
        self.num_classes = num_classes
        self.num_features = self.embed_dim = embed_dim
        norm_layer = norm_layer or partial(nn.LayerNorm, eps=1e-6)
        self.patch_embed = PatchEmbed(img_size=img_size, patch_size=patch_size, in_chans=in_chans,
                                      embed_dim=embed_dim)
        num_patches = self.patch_embed.num_patches
        self.cls_token = nn.Parameter(torch.zeros(1, 1, embed_dim))
        self.pos_embed = nn.Parameter(torch.zeros(1, num_patches + 1, embed_dim))
        self.pos_drop = nn.Dropout(p=drop_rate)
        dpr = [x.item() for x in torch.linspace(0, drop_path_rate, depth)]
        self.blocks = nn.ModuleList([
            BlockLAI(dim=embed_dim, num_heads=num_heads, mlp_ratio=mlp_ratio, qkv_bias=qkv_bias, qk_scale=qk_scale,
                     drop=drop_rate, attn_drop=attn_drop_rate, drop_path=dpr[i], norm_layer=norm_layer,
                     mask_epochs=mask_epochs)
            for i in range(depth)])
        self.norm = norm_layer(embed_dim)
        self.pre_logits = nn.Identity()
        self.head = nn.Linear(self.num_features, num_classes) if num_classes > 0 else nn.Identity()
        nn.init.trunc_normal_(self.pos_embed, std=.02)
        nn.init.trunc_normal_(self.cls_token, std=.02)
        self.apply(self._init_weights)

    @staticmethod
    def _init_weights(m):
        if isinstance(m, nn.Linear):
            nn.init.trunc_normal_(m.weight, std=.02)
            if m.bias is not None:
                nn.init.constant_(m.bias, 0)
        elif isinstance(m, nn.LayerNorm):
            nn.init.constant_(m.bias, 0)
            nn.init.constant_(m.weight, 1.0)

    def forward_features(self, x, epoch):
        norm = getattr(self, 'input_norm', None)
        self.patch_embed.check(x.permute(0, 3, 1, 2) if (x.dtype == torch.uint8 and norm and norm[2] == 'nhwc') else x)
        B = x.shape[0]
        dt = self.compute_dtype
        N = self.patch_embed.num_patches + 1
        D = self.embed_dim
        tok = VF.patch_embed(x, self.patch_embed.proj.weight, self.patch_embed.proj.bias, self.cls_token,
                             self.pos_embed, dt, norm=norm)
        if self.training and self.pos_drop.p > 0:      # x = self.pos_drop(x) (reference :139)
            tok = VF.dropout(tok, self.pos_drop.p)
        for blk in self.blocks:
            tok = blk.forward2d(tok, B, N, epoch)
        cls = tok.view(B, N, D)[:, 0]           # LayerNorm is row-wise: normalise the class rows only
        return VF.layer_norm(cls, self.norm.weight, self.norm.bias, self.norm.eps)

    def forward(self, x, epoch):
        f = self.forward_features(x, epoch)
        return VF.linear(f, self.head.weight, self.head.bias, out_dtype=torch.float32)
