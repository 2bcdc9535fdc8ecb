"""``torch.library`` registration of the C-ABI kernels: the ops of the hot path as dispatcher-visible custom ops
(``torch.ops.vit_amd.*``), each with a fake (shape) function and an autograd formula made of other vit_amd ops -- so
they trace under ``torch.compile`` / ``make_fx`` / export, show up in the profiler by name, and can be called by code
that never imports ``vit_amd.functional``.

The nn.Module drop-ins keep calling ``vit_amd.functional`` (hand-scheduled fused blocks over ``vit_amd.ops``: parameter
shadows, extended rows, the zero pool, the side stream); these ops are the same kernels at single-op granularity,
without any of that state: tensors in, tensors out, compute dtype = the dtype of the inputs.

    linear(x[M,K], w[N,K], bias fp32[N]?, act 'none'|'gelu'|'relu') -> y[M,N]
    layer_norm(x[M,D], gamma fp32[D], beta fp32[D], eps) -> (y, mean fp32[M], rstd fp32[M])
    attention(qkv[B*N,3D], tables[2N-1,2D]?, B, N, H, scale, band) -> (out[B*N,D], lse fp32[B,H,N])
    label_smoothing_ce(logits fp32[B,C], target int64[B], smoothing) -> loss fp32[]
    allreduce_bucket(flat (mutated), scale) -> ()      the gradient exchange of one flat bucket (RCCL sum, then * scale)

Reference surfaces: nn.Linear / nn.LayerNorm calls of the encoder layers (model.py:183-204), AttentionLAI.forward
(timm_classes/timm_locality_aware_transformer.py:45-79) and DisentangledSelfAttention.forward
(disentangled_transformer/disentangled_transformer.py:301-355), LabelSmoothingCrossEntropy (train.py:273),
DistributedDataParallel's bucket all-reduce (train.py:635)."""
from typing import Optional, Tuple

import torch
from torch import Tensor

from . import ops

_ACT = {'none': ops.ACT_NONE, 'gelu': ops.ACT_GELU, 'relu': ops.ACT_RELU}


# ------------------------------------------------------------------------------------------------ linear
@torch.library.custom_op('vit_amd::linear', mutates_args=(), device_types='cuda')
def linear(x: Tensor, w: Tensor, bias: Optional[Tensor], act: str) -> Tensor:
    return ops.gemm(x, w, bias=bias, act=_ACT[act])


@linear.register_fake
def _(x, w, bias, act):
    return x.new_empty(x.shape[0], w.shape[0])


@torch.library.custom_op('vit_amd::linear_backward', mutates_args=(), device_types='cuda')
def linear_backward(dy: Tensor, x: Tensor, w: Tensor, has_bias: bool) -> Tuple[Tensor, Tensor, Tensor]:
    dy = dy.contiguous()
    dx = ops.gemm(dy, w, trans_b=False)
    dw = ops.gemm(dy, x, trans_a=True, trans_b=False, out_dtype=torch.float32)        # split-K, fp32 accumulation
    db = ops.colsum(dy) if has_bias else torch.zeros(0, device=dy.device, dtype=torch.float32)
    return dx, dw.to(w.dtype), db


@linear_backward.register_fake
def _(dy, x, w, has_bias):
    return torch.empty_like(x), torch.empty_like(w), dy.new_empty(w.shape[0] if has_bias else 0, dtype=torch.float32)


def _linear_setup(ctx, inputs, output):
    x, w, bias, act = inputs
    if act != 'none':
        raise NotImplementedError('vit_amd::linear: autograd is defined for act = "none" (the fused blocks of '
                                  'vit_amd.functional carry the activation backward in the dX GEMM epilogue)')
    ctx.save_for_backward(x, w)
    ctx.has_bias = bias is not None


def _linear_backward(ctx, dy):
    x, w = ctx.saved_tensors
    dx, dw, db = torch.ops.vit_amd.linear_backward(dy, x, w, ctx.has_bias)
    return dx, dw, (db if ctx.has_bias else None), None


linear.register_autograd(_linear_backward, setup_context=_linear_setup)


# -------------------------------------------------------------------------------------------- layer norm
@torch.library.custom_op('vit_amd::layer_norm', mutates_args=(), device_types='cuda')
def layer_norm(x: Tensor, gamma: Tensor, beta: Tensor, eps: float) -> Tuple[Tensor, Tensor, Tensor]:
    return ops.layernorm_fwd(x, gamma, beta, eps)


@layer_norm.register_fake
def _(x, gamma, beta, eps):
    return torch.empty_like(x), x.new_empty(x.shape[0], dtype=torch.float32), x.new_empty(x.shape[0], dtype=torch.float32)


@torch.library.custom_op('vit_amd::layer_norm_backward', mutates_args=(), device_types='cuda')
def layer_norm_backward(dy: Tensor, x: Tensor, mean: Tensor, rstd: Tensor, gamma: Tensor) -> Tuple[Tensor, Tensor, Tensor]:
    D = x.shape[1]
    dg = torch.zeros(D, device=x.device, dtype=torch.float32)
    db = torch.zeros(D, device=x.device, dtype=torch.float32)
    dx = ops.layernorm_bwd(dy.contiguous(), x, mean, rstd, gamma, dg, db)
    return dx, dg, db


@layer_norm_backward.register_fake
def _(dy, x, mean, rstd, gamma):
    return torch.empty_like(x), torch.empty_like(gamma), torch.empty_like(gamma)


def _ln_setup(ctx, inputs, output):
    x, gamma, beta, eps = inputs
    y, mean, rstd = output
    ctx.save_for_backward(x, mean, rstd, gamma)


def _ln_backward(ctx, dy, dmean, drstd):
    x, mean, rstd, gamma = ctx.saved_tensors
    dx, dg, db = torch.ops.vit_amd.layer_norm_backward(dy, x, mean, rstd, gamma)
    return dx, dg, db, None


layer_norm.register_autograd(_ln_backward, setup_context=_ln_setup)


# --------------------------------------------------------------------------------------------- attention
def _split(qkv, tables, H):
    D = qkv.shape[1] // 3
    q, k, v = qkv[:, :D], qkv[:, D:2 * D], qkv[:, 2 * D:]
    pq = pk = None
    ld_p = 0
    if tables is not None:
        pq, pk = tables[:, :D], tables[:, D:]
        ld_p = tables.stride(0)
    return D, q, k, v, pq, pk, ld_p


@torch.library.custom_op('vit_amd::attention', mutates_args=(), device_types='cuda')
def attention(qkv: Tensor, tables: Optional[Tensor], B: int, N: int, H: int, scale: float, band: int) -> Tuple[Tensor, Tensor]:
    D, q, k, v, pq, pk, ld_p = _split(qkv, tables, H)
    return ops.attn_fwd(q, k, v, B, H, N, D // H, qkv.stride(0), pk, pq, ld_p, scale, band)


@attention.register_fake
def _(qkv, tables, B, N, H, scale, band):
    return qkv.new_empty(qkv.shape[0], qkv.shape[1] // 3), qkv.new_empty(B, H, N, dtype=torch.float32)


@torch.library.custom_op('vit_amd::attention_backward', mutates_args=(), device_types='cuda')
def attention_backward(dout: Tensor, qkv: Tensor, tables: Optional[Tensor], out: Tensor, lse: Tensor, B: int, N: int,
                       H: int, scale: float, band: int) -> Tuple[Tensor, Tensor]:
    D, q, k, v, pq, pk, ld_p = _split(qkv, tables, H)
    dqkv = torch.empty_like(qkv)
    dq, dk, dv = dqkv[:, :D], dqkv[:, D:2 * D], dqkv[:, 2 * D:]
    dtab = torch.zeros(2 * N - 1 if tables is not None else 0, 2 * D, device=qkv.device, dtype=torch.float32)
    dpq = dpk = None
    if tables is not None:
        dpq, dpk = dtab[:, :D], dtab[:, D:]
    ops.attn_bwd(q, k, v, B, H, N, D // H, qkv.stride(0), pk, pq, ld_p, scale, band, out, dout.contiguous(), lse,
                 dq, dk, dv, 3 * D, dpk, dpq, ld_dp=2 * D)
    return dqkv, dtab.to(qkv.dtype)


@attention_backward.register_fake
def _(dout, qkv, tables, out, lse, B, N, H, scale, band):
    D = qkv.shape[1] // 3
    return torch.empty_like(qkv), qkv.new_empty(2 * N - 1 if tables is not None else 0, 2 * D)


def _attn_setup(ctx, inputs, output):
    qkv, tables, B, N, H, scale, band = inputs
    out, lse = output
    ctx.save_for_backward(qkv, tables, out, lse)
    ctx.cfg = (B, N, H, scale, band)


def _attn_backward(ctx, dout, dlse):
    qkv, tables, out, lse = ctx.saved_tensors
    dqkv, dtab = torch.ops.vit_amd.attention_backward(dout, qkv, tables, out, lse, *ctx.cfg)
    return dqkv, (dtab if tables is not None else None), None, None, None, None, None


attention.register_autograd(_attn_backward, setup_context=_attn_setup)


# ---------------------------------------------------------------------------------- label-smoothing CE
@torch.library.custom_op('vit_amd::label_smoothing_ce', mutates_args=(), device_types='cuda')
def label_smoothing_ce(logits: Tensor, target: Tensor, smoothing: float) -> Tuple[Tensor, Tensor]:
    loss, lse = ops.ce_smooth_fwd(logits, target, smoothing)    # (mean loss, per-row log-sum-exp), views of one buffer
    return loss.clone(), lse.clone()                             # custom-op outputs may not alias each other


@label_smoothing_ce.register_fake
def _(logits, target, smoothing):
    return logits.new_empty(()), logits.new_empty(logits.shape[0])


@torch.library.custom_op('vit_amd::label_smoothing_ce_backward', mutates_args=(), device_types='cuda')
def label_smoothing_ce_backward(gout: Tensor, logits: Tensor, target: Tensor, lse: Tensor, smoothing: float) -> Tensor:
    return ops.ce_smooth_bwd(logits, target, smoothing, lse, gout.float().contiguous())


@label_smoothing_ce_backward.register_fake
def _(gout, logits, target, lse, smoothing):
    return torch.empty_like(logits)


def _ce_setup(ctx, inputs, output):
    logits, target, smoothing = inputs
    ctx.save_for_backward(logits, target, output[1])
    ctx.smoothing = smoothing


def _ce_backward(ctx, gloss, glse):
    logits, target, lse = ctx.saved_tensors
    return torch.ops.vit_amd.label_smoothing_ce_backward(gloss, logits, target, lse, ctx.smoothing), None, None


label_smoothing_ce.register_autograd(_ce_backward, setup_context=_ce_setup)


# ----------------------------------------------------------------------------------- gradient exchange
@torch.library.custom_op('vit_amd::allreduce_bucket', mutates_args=('flat',), device_types=('cuda', 'cpu'))
def allreduce_bucket(flat: Tensor, scale: float) -> None:
    """Sum ``flat`` over the default process group in place (backend "nccl" is RCCL over xGMI on ROCm; gloo on CPU),
    then multiply by ``scale`` (1 / world size for the gradient mean): what vit_amd.ddp.GradReducer does per bucket."""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(flat)
    if scale != 1.0:
        flat.mul_(scale)


@allreduce_bucket.register_fake
def _(flat, scale):
    return None
