"""MI355X drop-in for the reference's FAVOR+ attention (timm_classes/timm_vision_performer.py:
PerformerAttention :10).  Same constructor (it wraps / deep-copies a timm-style ``Attention``),
forward signature and state_dict keys (qkv.*, proj.*); the random projection ``w`` [m, dh] is a
plain attribute created lazily on the first forward, exactly as upstream (:29-31), so it is absent
from the state_dict.  Set ``module.w`` explicitly for reproducible tests / data-parallel runs.

VisionPerformerTIMM / test_vip (:140-209) need ``timm.create_model(pretrained=True)`` (a network
fetch) and are not restated; ``vit_amd.train.build_model('performer_timm')`` builds the same
architecture with random weights.
"""
import copy
import math

import torch
import torch.nn as nn

from .. import functional as VF
from ..precision import ComputeDtypeMixin


def orthogonal_matrix_chunk(cols, qr_uniform_q=False, device=None):
    """QR of a Gaussian block, transposed (reference :130-137)."""
    unstructured_block = torch.randn((cols, cols), device=device)
    q, r = torch.linalg.qr(unstructured_block, mode='reduced')
    if qr_uniform_q:
        d = torch.diag(r, 0)
        q *= d.sign()
    return q.t()


def gaussian_orthogonal_random_matrix(nb_rows, nb_columns, scaling=0, qr_uniform_q=False, device=None):
    """Stacked orthogonal blocks with rows rescaled to Gaussian norms (reference :103-128).
    Host-side setup (run once per module), so plain torch."""
    nb_full_blocks = int(nb_rows / nb_columns)
    block_list = [orthogonal_matrix_chunk(nb_columns, qr_uniform_q, device) for _ in range(nb_full_blocks)]
    remaining_rows = nb_rows - nb_full_blocks * nb_columns
    if remaining_rows > 0:
        block_list.append(orthogonal_matrix_chunk(nb_columns, qr_uniform_q, device)[:remaining_rows])
    final_matrix = torch.cat(block_list)
    if scaling == 0:
        multiplier = torch.randn((nb_rows, nb_columns), device=device).norm(dim=1)
    elif scaling == 1:
        multiplier = math.sqrt(float(nb_columns)) * torch.ones((nb_rows,), device=device)
    else:
        raise ValueError(f'Invalid scaling {scaling}')
    return torch.diag(multiplier) @ final_matrix


class Attention(nn.Module):
    """timm 0.4.5 ``Attention`` parameter container (qkv, proj): what PerformerAttention copies."""

    def __init__(self, dim, num_heads=8, qkv_bias=False, qk_scale=None, attn_drop=0., proj_drop=0.):
        super().__init__()
        self.num_heads = num_heads
        self.scale = qk_scale or (dim // num_heads) ** -0.5
        self.qkv = nn.Linear(dim, dim * 3, bias=qkv_bias)
        self.attn_drop = nn.Dropout(attn_drop)
        self.proj = nn.Linear(dim, dim)
        self.proj_drop = nn.Dropout(proj_drop)


class PerformerAttention(ComputeDtypeMixin, Attention):
    def __init__(self, attention, dim, num_heads=8, qkv_bias=False, attn_drop=0., proj_drop=0., n_orf=64,
                 kernel='softmax'):
        super().__init__(dim, num_heads=num_heads, qkv_bias=qkv_bias, attn_drop=attn_drop, proj_drop=proj_drop)
        self.m = int(n_orf)
        self.w = None
        self.num_heads = copy.deepcopy(attention.num_heads)
        self.kernel = kernel
        self.head_dim = dim // num_heads
        self.qkv = copy.deepcopy(attention.qkv)
        self.attn_drop = copy.deepcopy(attention.attn_drop)
        self.proj = copy.deepcopy(attention.proj)
        self.proj_drop = copy.deepcopy(attention.proj_drop)

    gaussian_orthogonal_random_matrix = staticmethod(gaussian_orthogonal_random_matrix)
    orthogonal_matrix_chunk = staticmethod(orthogonal_matrix_chunk)

    def ratio(self):
        return 1.0 / math.sqrt(math.sqrt(self.m))          # both feature maps use m^-1/4 here (:72,96)

    def features(self, device):
        if self.w is None:
            self.w = gaussian_orthogonal_random_matrix(self.m, self.head_dim, device=device)
        return self.w

    def favor_cfg(self, x2):
        """(features in compute dtype, kind, ratio, eps) for functional.PreLNBlockFn: a block that holds this attention
        runs as one autograd node."""
        kind, eps = VF._favor_kind(self.kernel)
        return self.features(x2.device).to(device=x2.device, dtype=x2.dtype).contiguous(), kind, float(self.ratio()), eps

    def forward2d(self, x2, B, N):
        """[B*N, C] in compute dtype -> [B*N, C]."""
        self.features(x2.device)
        C = x2.shape[1]
        qkv = VF.linear(x2, self.qkv.weight, self.qkv.bias)
        y = VF.favor_attention(qkv, self.w.to(x2.device), B, N, self.num_heads, C // self.num_heads, self.kernel,
                               self.ratio())
        if self.training and self.attn_drop.p > 0:     # attn_drop acts on the attention OUTPUT here (reference :61)
            y = VF.dropout(y, self.attn_drop.p)
        y = VF.linear(y, self.proj.weight, self.proj.bias)
        if self.training and self.proj_drop.p > 0:     # reference :64
            y = VF.dropout(y, self.proj_drop.p)
        return y

    def forward(self, x):
        B, N, C = x.shape
        y = self.forward2d(x.reshape(B * N, C).to(self.compute_dtype), B, N)
        return y.view(B, N, C).to(x.dtype)
