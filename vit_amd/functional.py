"""Autograd layer over the C-ABI kernels: one ``torch.autograd.Function`` per fused unit.

Each Function runs a hand-scheduled forward and backward made of ``ops.*`` launches (HIP kernels
through libvitk.so); torch supplies only tensors, the stream and the autograd graph edges.  There
is no fallback path: every op here needs GPU tensors.

Numerics policy: parameters stay fp32 (the reference is fp32-only).  ``dtype`` selects the compute
type: fp32 runs the exact f32 MFMA path (parity to 1e-5), bf16 keeps a bf16 shadow of each weight
(refreshed when the parameter's version counter moves) and bf16 activations with fp32
accumulation, statistics, softmax, bias, LayerNorm affine and all parameter gradients.
"""
import os
import weakref

import torch

from . import ops
from . import dropout as VD
from .ops import ACT_NONE, ACT_GELU, ACT_RELU, ACT_DGELU, ACT_DRELU, NO_DROP

_shadow = {}          # key -> (versions, copy, weakrefs, transposed)
_shadow_index = {}    # id(param) -> keys of the cached copies it is part of
_shadow_gen = [0]     # bumped whenever a copy is created or dropped (optim.AdamW caches its launch plans on it)


def shadow_generation():
    return _shadow_gen[0]


def _rows_of(p):
    return p.shape[0]


def shadow(params, dtype, transposed=False):
    """Compute-dtype copy of one parameter, or of the row-concatenation of several (fused QKV).
    ``transposed``: the copy is [in, sum(out)] (each parameter viewed as [out, in] = [shape[0], -1]) --
    the layout in which the dX GEMMs read both operands K-contiguous; None when the compute dtype
    is the parameters' own (the fp32 parity path multiplies by the parameter itself).
    Copies are cached on the parameters' versions; optim.AdamW rewrites them in its update pass
    (shadow_targets below), so in steady state no cast or transpose is ever launched."""
    if isinstance(params, torch.Tensor):
        params = (params,)
    if all(p.dtype == dtype for p in params):
        if transposed:
            return None
        if len(params) == 1:
            return params[0].detach()
    key = tuple(id(p) for p in params) + (dtype, transposed)
    ver = tuple((p._version, p.data_ptr()) for p in params)
    hit = _shadow.get(key)
    if hit is not None and hit[0] == ver and all(r() is p for r, p in zip(hit[2], params)):
        return hit[1]
    with torch.no_grad():
        if transposed:
            src = torch.cat([p.detach().reshape(p.shape[0], -1) for p in params], 0)
            dst = src.t().to(dtype).contiguous()
        else:
            src = params[0].detach() if len(params) == 1 else torch.cat([p.detach() for p in params], 0)
            if dtype == torch.bfloat16 and src.dtype == torch.float32:
                src = src.contiguous()
                dst = torch.empty(src.shape, device=src.device, dtype=dtype)
                ops.cast(src.view(-1), dst.view(-1))
            else:
                dst = src.to(dtype)
    if len(_shadow) > 4096:
        for k in [k for k, v in _shadow.items() if any(r() is None for r in v[2])]:
            del _shadow[k]
        _shadow_gen[0] += 1
        for pid in list(_shadow_index):
            _shadow_index[pid] = [k for k in _shadow_index[pid] if k in _shadow]
            if not _shadow_index[pid]:
                del _shadow_index[pid]
    _shadow[key] = (ver, dst, tuple(weakref.ref(p) for p in params), transposed)
    _shadow_gen[0] += 1
    for p in params:
        keys = _shadow_index.setdefault(id(p), [])
        if key not in keys:
            keys.append(key)
    return dst


def shadow_targets(p):
    """The live cached copies that hold parameter ``p``: a list of (key, copy, offset, transposed), offset =
    the first row (column when transposed) of p in the copy.  For an optimizer that updates p in
    place through the C-ABI (no version bump): it must rewrite each of them or drop it
    (shadow_drop)."""
    out = []
    for key in _shadow_index.get(id(p), ()):
        hit = _shadow.get(key)
        if hit is None:
            continue
        ver, dst, refs, transposed = hit
        ps = [r() for r in refs]
        if any(q is None for q in ps) or ver != tuple((q._version, q.data_ptr()) for q in ps):
            continue                      # stale anyway: rebuilt at its next use
        off = 0
        for q in ps:
            if q is p:
                out.append((key, dst, off, transposed))
                break
            off += q.shape[0]
    return out


def shadow_drop(key):
    if _shadow.pop(key, None) is not None:
        _shadow_gen[0] += 1


def _f32(t):
    return None if t is None else t.detach()


def _act_codes(act):
    """(forward epilogue, backward epilogue, backward aux is the pre-activation?)"""
    if act == 'gelu':
        return ACT_GELU, ACT_DGELU
    if act == 'relu':
        return ACT_RELU, ACT_DRELU
    raise ValueError(f'unsupported activation {act!r}')


class _ZeroPool:
    """One zero-filled fp32 buffer per fused backward, handed out in 64-byte aligned slices: the
    ten-odd small accumulators of a block (LayerNorm / bias gradients, relative-table gradient)
    cost one fill kernel instead of one each."""

    def __init__(self, device, *sizes):
        self.buf = torch.zeros(sum((n + 15) // 16 * 16 for n in sizes), device=device, dtype=torch.float32)
        self.off = 0

    def take(self, *shape):
        n = 1
        for d in shape:
            n *= d
        v = self.buf[self.off:self.off + n].view(*shape)
        self.off += (n + 15) // 16 * 16
        return v


def grad_slot(param):
    """The zeroed view of the data-parallel reducer's gradient slab that the weight gradient of ``param`` may be
    accumulated into, when the reducer offered one for this step (ddp.GradReducer.prepare; taken at most
    once) -- else None (the caller uses a zeroed slice of its pool).  Marks the parameter as able to take one."""
    param._vitk_direct = True
    slot = getattr(param, '_vitk_grad_slot', None)
    if slot is not None and slot.dtype == torch.float32 and slot.shape == param.shape:
        param._vitk_grad_slot = None
        return slot
    return None


# ---- the weight-gradient GEMMs of a fused block run on a second stream ------------------------------------------
# A layer's dX GEMMs are a dependent chain; its dW GEMMs hang off that chain and nothing in the layer reads them.
# Every GEMM here is a persistent launch of one workgroup per CU whose last round of tiles leaves CUs idle (396
# tiles of 256 x 256 on 256 CUs: the second round is 55 % full).  With the dW GEMMs on a stream of their own the
# hardware starts their workgroups on whatever CUs the dX launch has let go of, and the other way round.
# Lifetimes: the operands of a pending dW GEMM are kept referenced until side_join(), which every fused backward
# calls before it returns (the main stream then waits for the side stream), so neither the caching allocator nor
# the per-model scratch buffers (_ext_workspace) can hand their memory to a later main-stream kernel too early.
SIDE_DW = os.environ.get('VITK_SIDE_DW', '1') != '0'
_side_streams = {}
_side_pending = {}


def streams_overlap(a, b, device, cycles=1_500_000):
    """True when kernels queued on streams ``a`` and ``b`` run at the same time.  HIP maps its streams onto a handful of
    hardware queues (GPU_MAX_HW_QUEUES, 4 here) and two streams that share one run in turns, whatever the events between
    them say.  Probe: one single-thread spin kernel (torch.cuda._sleep) on each, timed together."""
    def run(both):
        torch.cuda.synchronize(device)
        e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
        e0.record(a)
        b.wait_event(e0)
        with torch.cuda.stream(a):
            torch.cuda._sleep(cycles)
        e1.record(a)
        if both:
            with torch.cuda.stream(b):
                torch.cuda._sleep(cycles)
        e2.record(b)
        a.wait_event(e2)
        e3 = torch.cuda.Event(enable_timing=True)
        e3.record(a)
        e3.synchronize()
        return e0.elapsed_time(e3)
    run(True)                                  # warm-up (first launch of the spin kernel)
    one, two = run(False), run(True)
    return two < 1.5 * one


def _side_stream(device):
    """The second stream of a device, on a hardware queue of its own.  Which queue a new stream gets depends on how many
    streams the process made before (a process that initialised RCCL first got a side stream on the MAIN stream's
    queue: every dW GEMM then ran in turn with the main stream's kernels, 83 ms per headline step instead of 65 -- seen in
    a rocprofv3 trace, profiles/README.md r03).  So: take streams from torch's pool until one really overlaps the current
    stream (probe above); normal priority first -- the high-priority class, which HIP keeps on queues of its own, is
    left to RCCL's stream (ddp.init_process_group asks for it), so that the three never share a queue."""
    st = _side_streams.get(device)
    if st is None:
        cur = torch.cuda.current_stream(device)
        cands, st = [], None
        for attempt in range(6):
            c = torch.cuda.Stream(device=device, priority=0 if attempt < 5 else -1)
            cands.append(c)                    # keep the rejected ones alive until the choice is made: a freed stream's queue is re-used
            if torch.cuda.is_current_stream_capturing() or streams_overlap(cur, c, device):
                st = c
                break
        if st is None:
            import warnings
            warnings.warn('vit_amd: no stream found that overlaps the current one; the dW GEMMs will run in turn with the '
                          'main stream (set VITK_DW_SPLITK= to give them the whole chip)')
            st = cands[0]
        _side_streams[device] = st
    return st


def forget_side_streams():
    """Drop the chosen side streams (after everything queued on them has finished): the next fused backward probes again."""
    for dev, st in list(_side_streams.items()):
        st.synchronize()
    _side_streams.clear()
    _side_pending.clear()


def side_join(device):
    """Main stream waits for the dW GEMMs queued on the side stream since the last join."""
    keep = _side_pending.get(device)
    if keep:
        ev = torch.cuda.Event()
        ev.record(_side_stream(device))
        torch.cuda.current_stream(device).wait_event(ev)
        keep.clear()


# "tiles:splitk,..." overrides the library's split-K choice (256 / tiles: every CU busy) for the weight-gradient GEMMs with
# that many 256 x 256 output tiles.  With FEWER splits a dW GEMM holds fewer CUs for longer and runs beside the main
# stream instead of in turns with it, and it issues fewer split-K atomics: headline step, same box, '16:2,64:2' (proj:
# 2 splits instead of 16; fc1 / fc2: 2 instead of 4) 1984-1990 img/s against 1930-1938, '16:4,64:3' 1976-1982, '64:3' 1959-1969;
# more splits, or any change to the qkv one (48 tiles, 5 splits), lose 1-4 %.  The default since round 3 (round 2 kept the
# library's own splits so that the in-step launch times of the main stream's GEMMs described those kernels alone: 175 us
# per NT launch against 203 beside the dW GEMMs; bench.py now measures kernel quality in a pass of its own,
# roofline.isolated, and the step runs the faster schedule).  VITK_DW_SPLITK='' / `bench.py --no-max-overlap`: the old one.
# Swept again at the end of round 3, with the new attention kernels in the step (one box, `tools/runs/r3_run34.sh` /
# `r3_run35.sh`, img/s): '16:2,64:2' 2092-2099 | proj 16:3 2121, **16:4 2128-2134**, 16:6 2127, 16:8 2123, 16:1 1835 |
# fc 64:3 2119, 64:4 2059 | qkv 48:2 1977, 48:3 / 48:4 2088 (the library's 5: best).  The proj dW GEMM with four splits (64
# CUs for half as long) is +1.8 %: the default.  768-wide models (tools/runs/r3_run37.sh): fc1 / fc2 dW (36 tiles) with 4 splits
# instead of the library's 7: lai_s16 6315-6323 -> 6450, performer_s16_384 2287 -> 2292; proj (9 tiles) and qkv (27): nothing
# that helps both.
_DW_SPLITK = os.environ.get('VITK_DW_SPLITK', '16:4,64:2,36:4')


def _dw_splitk(dy, x):
    if not _DW_SPLITK:
        return 0
    tiles = ((dy.shape[1] + 255) // 256) * ((x.shape[1] + 255) // 256)
    for item in _DW_SPLITK.split(','):
        t, k = item.split(':')
        if int(t) == tiles:
            return int(k)
    return 0


def _dw(dy, x, out=None, side=False):
    """dW[out, in] = dy^T x in fp32 (split-K over the token dimension).  ``out``: a ZEROED fp32 buffer to
    add into (a slice of the block's one zero-filled pool: the split-K atomics then need no memset of
    their own -- four 6 us fills per block become part of one).  ``side``: queue it on the side stream (the
    caller must side_join() before anything reads ``out`` or its operands' memory is released)."""
    if out is not None and dy.dtype == torch.bfloat16:
        if side and SIDE_DW and not torch.cuda.is_current_stream_capturing():
            dev = dy.device
            st = _side_stream(dev)
            ev = torch.cuda.Event()
            ev.record()                       # dy, x and the zero fill of ``out`` are queued before this point
            st.wait_event(ev)
            with torch.cuda.stream(st):
                ops.gemm(dy, x, trans_a=True, trans_b=False, out=out, accumulate=True, splitk=_dw_splitk(dy, x))
            _side_pending.setdefault(dev, []).append((dy, x, out))
            return out
        return ops.gemm(dy, x, trans_a=True, trans_b=False, out=out, accumulate=True, splitk=_dw_splitk(dy, x))
    return ops.gemm(dy, x, trans_a=True, trans_b=False, out_dtype=torch.float32)


def _colsum(x, out=None, side=False):
    """Column sums (a bias gradient) of x into ``out`` (zeroed fp32; allocated when None).  ``side``: on the side stream,
    under _dw's rules -- nothing on the main stream waits for a bias gradient before the block's side_join(), and beside
    a persistent GEMM on every CU the 30 us pass took 180 us of the main stream's time."""
    if side and out is not None and SIDE_DW and SIDE_COLSUM and x.is_cuda and not torch.cuda.is_current_stream_capturing():
        dev = x.device
        st = _side_stream(dev)
        ev = torch.cuda.Event()
        ev.record()
        st.wait_event(ev)
        with torch.cuda.stream(st):
            ops.colsum(x, out=out)
        _side_pending.setdefault(dev, []).append((x, out))
        return out
    return ops.colsum(x, out=out)


# The bias-gradient column sums of a fused block run on the side stream too: beside a persistent GEMM on every CU the 30 us
# pass took 180 us of the main stream's time (1878 -> 1917 img/s).  On since round 3 (see _DW_SPLITK above for why round 2
# left it off); VITK_SIDE_COLSUM=0 / `bench.py --no-max-overlap` puts them back on the main stream.
SIDE_COLSUM = os.environ.get('VITK_SIDE_COLSUM', '1') != '0'


def _dx(dy, params, **kw):
    """dX = dy W for W = the row-concatenation of ``params`` ([out, in] each).  bf16: against the
    transposed copy (both operands K-contiguous: ~10 % faster than reading W K-major); fp32: W itself."""
    wt = shadow(params, dy.dtype, transposed=True)
    if wt is not None:
        return ops.gemm(dy, wt, **kw)
    return ops.gemm(dy, shadow(params, dy.dtype), trans_b=False, **kw)


def _dx_nn(dy, w, **kw):
    return ops.gemm(dy, w, trans_b=False, **kw)


# --------------------------------------------------------------------------- LayerNorm
class LayerNormFn(torch.autograd.Function):
    """y = LayerNorm(x) over the last dim of a 2-D row view (rows may be strided)."""

    @staticmethod
    def forward(ctx, x, gamma, beta, eps):
        y, mean, rstd = ops.layernorm_fwd(x, _f32(gamma), _f32(beta), eps)
        ctx.save_for_backward(x, mean, rstd, gamma)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, mean, rstd, gamma = ctx.saved_tensors
        D = x.shape[1]
        dg = torch.zeros(D, device=x.device, dtype=torch.float32)
        db = torch.zeros(D, device=x.device, dtype=torch.float32)
        dx = ops.layernorm_bwd(dy if dy.stride(1) == 1 else dy.contiguous(), x, mean, rstd, _f32(gamma), dg, db)
        return dx, dg, db, None


def layer_norm(x2d, gamma, beta, eps):
    return LayerNormFn.apply(x2d, gamma, beta, eps)


# --------------------------------------------------------------------------- Linear
class LinearFn(torch.autograd.Function):
    """y = x W^T + b on 2-D rows."""

    @staticmethod
    def forward(ctx, x, weight, bias, out_dtype):
        w = shadow(weight, x.dtype)
        ctx.save_for_backward(x, weight)
        ctx.has_bias = bias is not None
        return ops.gemm(x, w, bias=_f32(bias), out_dtype=out_dtype)

    @staticmethod
    def backward(ctx, dy):
        x, weight = ctx.saved_tensors
        dy = dy.to(x.dtype).contiguous()
        dx = _dx(dy, weight) if ctx.needs_input_grad[0] else None
        dw = _dw(dy, x)
        db = ops.colsum(dy) if ctx.has_bias else None
        return dx, dw, db, None


def linear(x2d, weight, bias=None, out_dtype=None):
    return LinearFn.apply(x2d, weight, bias, out_dtype)


class FusedLinearFn(torch.autograd.Function):
    """y = x [W_1; ..; W_k]^T + [b_1; ..; b_k]: several nn.Linear over the same rows as one GEMM
    (one cached copy of the stacked weights: shadow)."""

    @staticmethod
    def forward(ctx, x, *wb):
        ws, bs = wb[0::2], wb[1::2]
        ctx.save_for_backward(x, *ws)
        return ops.gemm(x, shadow(tuple(ws), x.dtype), bias=torch.cat([b.detach() for b in bs]))

    @staticmethod
    def backward(ctx, dy):
        x, *ws = ctx.saved_tensors
        dy = dy.to(x.dtype).contiguous()
        dx = _dx(dy, tuple(ws)) if ctx.needs_input_grad[0] else None
        dw, db = _dw(dy, x), ops.colsum(dy)
        grads, off = [], 0
        for w in ws:
            n = w.shape[0]
            grads += [dw[off:off + n], db[off:off + n]]
            off += n
        return (dx, *grads)


def fused_linear(x2d, *linears):
    """``linears``: nn.Linear modules (all with bias) applied to the same rows; outputs side by side."""
    args = []
    for lin in linears:
        args += [lin.weight, lin.bias]
    return FusedLinearFn.apply(x2d, *args)


class LinearResFn(torch.autograd.Function):
    """y = x W^T + b + res on 2-D rows (the residual rides in the GEMM epilogue)."""

    @staticmethod
    def forward(ctx, x, weight, bias, res):
        ctx.save_for_backward(x, weight)
        return ops.gemm(x, shadow(weight, x.dtype), bias=_f32(bias), residual=res)

    @staticmethod
    def backward(ctx, dy):
        x, weight = ctx.saved_tensors
        dy = dy.to(x.dtype).contiguous()
        return _dx(dy, weight), _dw(dy, x), ops.colsum(dy), dy


def linear_res(x2d, weight, bias, res):
    return LinearResFn.apply(x2d, weight, bias, res)


# --------------------------------------------------------------------------- attention core
def _attn_fwd(qkv, B, N, H, dh, tables, scale, w, drop=NO_DROP):
    """qkv [B*N, 3D] (q | k | v).  tables: None or [2N-1, 2D] (Pq | Pk).  drop: (p, seed, offset) of the dropout on
    the probabilities."""
    D = H * dh
    ld = qkv.stride(0)
    q, k, v = qkv[:, :D], qkv[:, D:2 * D], qkv[:, 2 * D:]
    pq = pk = None
    ld_p = 0
    if tables is not None:
        pq, pk = tables[:, :D], tables[:, D:]
        ld_p = tables.stride(0)
    return ops.attn_fwd(q, k, v, B, H, N, dh, ld, pk, pq, ld_p, scale, -1 if w is None else w, drop=drop)


def _attn_bwd(qkv, B, N, H, dh, tables, scale, w, out, dout, lse, dtab=None, dqkv=None, drop=NO_DROP, dbias=None):
    """Returns dqkv [B*N, 3D] and (for tables) fp32 dtables [2N-1, 2D] (dPq | dPk); ``dtab``: a
    zeroed buffer to accumulate into."""
    D = H * dh
    ld = qkv.stride(0)
    q, k, v = qkv[:, :D], qkv[:, D:2 * D], qkv[:, 2 * D:]
    if dqkv is None:
        dqkv = torch.empty(B * N, 3 * D, device=qkv.device, dtype=qkv.dtype)
    dq, dk, dv = dqkv[:, :D], dqkv[:, D:2 * D], dqkv[:, 2 * D:]
    pq = pk = dpq = dpk = None
    ld_p = 0
    if tables is None:
        dtab = None
    else:
        pq, pk = tables[:, :D], tables[:, D:]
        ld_p = tables.stride(0)
        if dtab is None:
            dtab = torch.zeros(2 * N - 1, 2 * D, device=qkv.device, dtype=torch.float32)
        dpq, dpk = dtab[:, :D], dtab[:, D:]
    ops.attn_bwd(q, k, v, B, H, N, dh, ld, pk, pq, ld_p, scale, -1 if w is None else w, out, dout, lse,
                 dq, dk, dv, 3 * D, dpk, dpq, ld_dp=2 * D, drop=drop, dbias=dbias)
    return dqkv, dtab


class AttentionFn(torch.autograd.Function):
    """Fused attention on a packed qkv buffer: softmax(scale * (QK^T [+ c2p + p2c]) | LAI band) V."""

    @staticmethod
    def forward(ctx, qkv, tables, B, N, H, dh, scale, w, drop=NO_DROP):
        out, lse = _attn_fwd(qkv, B, N, H, dh, tables, scale, w, drop)
        ctx.save_for_backward(qkv, tables, out, lse)
        ctx.cfg = (B, N, H, dh, scale, w, drop)
        return out

    @staticmethod
    def backward(ctx, dout):
        qkv, tables, out, lse = ctx.saved_tensors
        B, N, H, dh, scale, w, drop = ctx.cfg
        dqkv, dtab = _attn_bwd(qkv, B, N, H, dh, tables, scale, w, out, dout.contiguous(), lse, drop=drop)
        if dtab is not None:
            dtab = dtab.to(tables.dtype)
        return dqkv, dtab, None, None, None, None, None, None, None


def attention(qkv, tables, B, N, H, dh, scale, w, p_drop=0.0):
    """p_drop: dropout on the probabilities (a fresh stream is drawn per call when > 0)."""
    return AttentionFn.apply(qkv, tables, B, N, H, dh, scale, w, VD.next_stream(p_drop))


# --------------------------------------------------------------------------- dropout
class DropoutFn(torch.autograd.Function):
    """y = x * keep / (1 - p) [+ res]: nn.Dropout (training) with the residual add that follows it folded in.  The mask
    is a function of (seed, offset, element index): the backward regenerates it, nothing is saved."""

    @staticmethod
    def forward(ctx, x, drop, res, rows_per_sample=0):
        ctx.drop = drop
        ctx.has_res = res is not None
        ctx.rps = rows_per_sample
        return ops.dropout(x, drop, res=res, rows_per_sample=rows_per_sample)

    @staticmethod
    def backward(ctx, dy):
        dy = dy.contiguous()
        return ops.dropout(dy, ctx.drop, rows_per_sample=ctx.rps), None, (dy if ctx.has_res else None), None


def dropout(x2d, p, res=None):
    """Training-mode dropout of a 2-D row view (+ res); p == 0: identity (+ res).  Draws one stream (vit_amd.dropout)."""
    drop = VD.next_stream(p)
    if drop[0] == 0.0:
        return x2d if res is None else x2d + res
    return DropoutFn.apply(x2d.contiguous(), drop, res)


def drop_path(x2d, p, rows_per_sample, res=None):
    """DropPath / stochastic depth of a [B * rows_per_sample, C] row view (+ res): each SAMPLE's branch is kept with
    probability 1 - p and scaled by 1 / (1 - p) (timm 0.4.5 layers/drop.py; timm_locality_aware_transformer.py:94,97-98).
    Draws one stream; sample s is its element s."""
    drop = VD.next_stream(p)
    if drop[0] == 0.0:
        return x2d if res is None else x2d + res
    return DropoutFn.apply(x2d.contiguous(), drop, res, int(rows_per_sample))


# --------------------------------------------------------------------------- fused encoder blocks
def _mlp_fwd(x, w1, b1, w2, b2, act, residual, drop_out=NO_DROP):
    """y = dropout(act(x W1^T + b1) W2^T + b2) + residual.  Returns y, hmid, pre (pre is None for relu)."""
    fwd_code, _ = _act_codes(act)
    pre = torch.empty(x.shape[0], w1.shape[0], device=x.device, dtype=x.dtype) if act == 'gelu' else None
    hmid = ops.gemm(x, w1, bias=b1, act=fwd_code, aux_out=pre)
    if drop_out[0] > 0.0:
        y = ops.gemm(hmid, w2, bias=b2)
        y = ops.dropout(y, drop_out, res=residual, out=y)
    else:
        y = ops.gemm(hmid, w2, bias=b2, residual=residual)
    return y, hmid, pre


def _mlp_bwd(dy, x, w1, w2, act, hmid, pre, residual_grad, db1=None, db2=None, db2_done=False, dw1=None, dw2=None):
    """w1 / w2: the fc1 / fc2 weight PARAMETERS (not their copies).
    Returns dx (+ residual_grad), dW1, db1, dW2, db2 (db1 / db2: zeroed accumulators, optional;
    db2_done: db2 already holds the column sums of dy -- the LayerNorm backward that made dy summed them)."""
    _, bwd_code = _act_codes(act)
    if not db2_done:
        db2 = _colsum(dy, out=db2, side=True)
    dw2 = _dw(dy, hmid, out=dw2, side=True)
    dpre = _dx(dy, w2, act=bwd_code, aux_in=pre if act == 'gelu' else hmid)
    db1 = _colsum(dpre, out=db1, side=True)
    dw1 = _dw(dpre, x, out=dw1, side=True)
    dx = _dx(dpre, w1, residual=residual_grad)
    return dx, dw1, db1, dw2, db2


class PreLNBlockFn(torch.autograd.Function):
    """x + proj(attn(LN1(x))) then + mlp(LN2(.)) -- BlockLAI.forward
    (timm_classes/timm_locality_aware_transformer.py:96-99) and, with act='relu' and no band,
    TransformerEncoderLayer.forward_pre (model.py:194-204).  cfg = (B, N, H, eps, scale, w, act[, favor]): with
    favor = (features [m, dh] in compute dtype, kind, ratio, feature eps) the attention is FAVOR+ (a timm block whose
    attn was replaced by PerformerAttention, train.py:147-165) -- the same node, the saved log-sum-exp replaced by the
    FAVOR+ workspace."""

    @staticmethod
    def forward(ctx, x, g1, be1, wqkv, bqkv, wproj, bproj, g2, be2, w1, b1, w2, b2, cfg):
        B, N, H, eps, scale, w, act = cfg[:7]
        favor = cfg[7] if len(cfg) > 7 else None
        D = x.shape[1]
        dh = D // H
        dt = x.dtype
        h1, mean1, rstd1 = ops.layernorm_fwd(x, _f32(g1), _f32(be1), eps)
        qkv = ops.gemm(h1, shadow(wqkv, dt), bias=_f32(bqkv))
        if favor is None:
            attn, lse = _attn_fwd(qkv, B, N, H, dh, None, scale, w)
        else:
            fw, kind, ratio, feps = favor
            lse = ops.favor_workspace(qkv, B, H, dh, fw.shape[0])          # the op's saved state
            attn = ops.favor_fwd(qkv[:, :D], qkv[:, D:2 * D], qkv[:, 2 * D:], B, H, N, dh, qkv.stride(0), fw, kind, ratio,
                                 feps, lse)
        x1 = ops.gemm(attn, shadow(wproj, dt), bias=_f32(bproj), residual=x)
        h2, mean2, rstd2 = ops.layernorm_fwd(x1, _f32(g2), _f32(be2), eps)
        y, hmid, pre = _mlp_fwd(h2, shadow(w1, dt), _f32(b1), shadow(w2, dt), _f32(b2), act, x1)
        ctx.save_for_backward(x, g1, wqkv, wproj, g2, w1, w2, mean1, rstd1, h1, qkv, attn, lse, x1, mean2, rstd2,
                              h2, hmid, pre)
        ctx.cfg = cfg
        ctx.has_qkv_bias = bqkv is not None
        return y

    @staticmethod
    def backward(ctx, dy):
        (x, g1, wqkv, wproj, g2, w1, w2, mean1, rstd1, h1, qkv, attn, lse, x1, mean2, rstd2, h2, hmid,
         pre) = ctx.saved_tensors
        B, N, H, eps, scale, w, act = ctx.cfg[:7]
        favor = ctx.cfg[7] if len(ctx.cfg) > 7 else None
        D = x.shape[1]
        dh = D // H
        dt = x.dtype
        dy = dy.contiguous()
        F = w1.shape[0]
        direct = dt == torch.bfloat16      # the fp32 path's dW GEMMs allocate their own outputs
        s_w1, s_w2, s_wproj, s_wqkv = (grad_slot(q) if direct else None for q in (w1, w2, wproj, wqkv))
        zp = _ZeroPool(x.device, D, D, D, D, D, D, F, D, 3 * D,
                       0 if s_w1 is not None else F * D, 0 if s_w2 is not None else D * F,
                       0 if s_wproj is not None else D * D, 0 if s_wqkv is not None else 3 * D * D)
        dh2, dw1, db1, dw2, db2 = _mlp_bwd(dy, h2, w1, w2, act, hmid, pre, None,
                                           db1=zp.take(F), db2=zp.take(D),
                                           dw1=s_w1 if s_w1 is not None else zp.take(F, D),
                                           dw2=s_w2 if s_w2 is not None else zp.take(D, F))
        dg2, dbe2, dbproj = zp.take(D), zp.take(D), zp.take(D)
        dx1 = ops.layernorm_bwd(dh2, x1, mean2, rstd2, _f32(g2), dg2, dbe2, add=dy, dxsum=dbproj)
        dwproj = _dw(dx1, attn, out=s_wproj if s_wproj is not None else zp.take(D, D), side=True)
        dattn = _dx(dx1, wproj)
        dbqkv = zp.take(3 * D) if ctx.has_qkv_bias else None        # summed by the attention backward itself
        if favor is None:
            dqkv, _ = _attn_bwd(qkv, B, N, H, dh, None, scale, w, attn, dattn, lse, dbias=dbqkv)
        else:
            fw, kind, ratio, feps = favor
            dqkv = torch.empty_like(qkv)
            ops.favor_bwd(qkv[:, :D], qkv[:, D:2 * D], qkv[:, 2 * D:], B, H, N, dh, qkv.stride(0), fw, kind, ratio, feps,
                          attn, dattn, dqkv[:, :D], dqkv[:, D:2 * D], dqkv[:, 2 * D:], 3 * D, lse)
            if dbqkv is not None:
                _colsum(dqkv, out=dbqkv, side=True)
        dwqkv = _dw(dqkv, h1, out=s_wqkv if s_wqkv is not None else zp.take(3 * D, D), side=True)
        # a fresh alias of a slab view: autograd adopts it as .grad without a copy
        dw1, dw2, dwproj, dwqkv = (t.detach() if t is s else t
                                   for t, s in ((dw1, s_w1), (dw2, s_w2), (dwproj, s_wproj), (dwqkv, s_wqkv)))
        dh1 = _dx(dqkv, wqkv)
        dg1, dbe1 = zp.take(D), zp.take(D)
        dx = ops.layernorm_bwd(dh1, x, mean1, rstd1, _f32(g1), dg1, dbe1, add=dx1)
        side_join(x.device)
        return dx, dg1, dbe1, dwqkv, dbqkv, dwproj, dbproj, dg2, dbe2, dw1, db1, dw2, db2, None


_ext_ws = {}


def _ext_workspace(tag, rows, cols, dtype, device, layout):
    """Backward scratch of the extended-row layout ([tokens + table rows, cols]), one per shape for the
    whole model: layers run one after the other on the stream, nothing here is saved or returned,
    and the rows past the tokens that must read as zero are zeroed ONCE (kernels only write the
    token rows; the table-gradient rows are rewritten whole every layer).  ``layout`` = (token rows,
    table rows): two layouts with the same padded row count must not share a buffer -- the stale table
    gradients of one would lie in the must-be-zero rows of the other."""
    key = (tag, rows, cols, dtype, device, layout, torch.cuda.current_stream(device).cuda_stream)
    buf = _ext_ws.get(key)
    if buf is None:
        if len(_ext_ws) > 16:
            _ext_ws.clear()
        buf = _ext_ws[key] = torch.zeros(rows, cols, device=device, dtype=dtype)
    return buf


def _tail_rows(M1, R):
    """Rows appended to a [M1, D] token matrix for R relative-table rows: R plus zero rows up to
    the next multiple of 64 of the total (the K dimension of the dW GEMM)."""
    return R + (-(M1 + R)) % 64


class DisLayerFn(torch.autograd.Function):
    """DisentangledEncoderLayer.forward (disentangled_transformer.py:228-246): post-LN block whose
    attention adds the c2p / p2c terms from relative rows projected with the content q/k
    projections (share_att_key, :270,373-377).  ``band`` = the 2N-1 reachable rows of the
    LayerNormed relative table (row r <-> i - j = r - (N-1)), already in compute dtype.

    The table rows ride along as EXTRA ROWS of the token matrix: x_ext = [x ; band ; 0] goes through
    the qkv GEMM once (the tail of its output is (Pq | Pk | unused)), and in backward
    dqkv_ext = [dqkv ; (dPq | dPk | 0)] gives the bias / weight gradients of the shared projections
    and dband from the same three GEMMs as the tokens -- no 393-row GEMMs, extra column sums or
    accumulate passes.  cfg[7] (``x_ext``) says that x is the head of a buffer that already has
    the tail rows allocated (the previous layer's output), so nothing is copied.

    cfg[8] (optional): the layer's four dropout streams (p, seed, offset) in the reference's order -- relative rows
    (pos_dropout, :332/:883), probabilities (:343/:915, inside the attention kernels), attention output
    (dropout_AttnOut, :189), block output (dropout_out, :195).  The reference trains with dropout 0.1 by default
    (train.py:79): the node keeps its layout, its side-stream weight-gradient GEMMs and the gradient slots with it; the
    dropped tensors are regenerated in backward from the three numbers, nothing is stored."""

    @staticmethod
    def forward(ctx, x, band, wq, bq, wk, bk, wv, bv, wo, bo, g1, be1, wi, bi, wout, bout, g2, be2, cfg):
        B, N, H, eps, scale, w, act = cfg[:7]
        x_is_ext = len(cfg) > 7 and cfg[7]
        d_pos, d_attn, d_o1, d_o2 = cfg[8] if len(cfg) > 8 and cfg[8] is not None else (NO_DROP,) * 4
        M1, D = x.shape
        dh = D // H
        dt = x.dtype
        wqkv = shadow((wq, wk, wv), dt)
        bqkv = torch.cat([bq.detach(), bk.detach(), bv.detach()])
        tables = None
        if band is not None:
            R = band.shape[0]
            ME = M1 + _tail_rows(M1, R)
            if x_is_ext and (x.storage_offset() != 0 or not x.is_contiguous()
                             or x.untyped_storage().nbytes() < ME * D * x.element_size()):
                raise ValueError('DisLayerFn: x_ext set but x is not the head of a [%d, %d] buffer' % (ME, D))
            if x_is_ext:
                x_ext = torch.empty(0, device=x.device, dtype=dt).set_(x.untyped_storage(), 0, (ME, D), (D, 1))
            else:
                x_ext = torch.empty(ME, D, device=x.device, dtype=dt)
                x_ext[:M1].copy_(x)
            if d_pos[0] > 0.0:
                ops.dropout(band, d_pos, out=x_ext[M1:M1 + R])
            else:
                x_ext[M1:M1 + R].copy_(band)
            x_ext[M1 + R:].zero_()
            x = x_ext[:M1]
            qkv_ext = ops.gemm(x_ext, wqkv, bias=bqkv)
            qkv = qkv_ext[:M1]
            tables = qkv_ext[M1:M1 + R, :2 * D]                   # (Pq | Pk), row stride 3 D
            y_ext = torch.empty(ME, D, device=x.device, dtype=dt)
        else:
            x_ext, qkv_ext = x, ops.gemm(x, wqkv, bias=bqkv)
            qkv = qkv_ext
            y_ext = torch.empty(M1, D, device=x.device, dtype=dt)
        attn, lse = _attn_fwd(qkv, B, N, H, dh, tables, scale, w, d_attn)
        if d_o1[0] > 0.0:
            s1 = ops.gemm(attn, shadow(wo, dt), bias=_f32(bo))
            s1 = ops.dropout(s1, d_o1, res=x, out=s1)
        else:
            s1 = ops.gemm(attn, shadow(wo, dt), bias=_f32(bo), residual=x)
        a, mean1, rstd1 = ops.layernorm_fwd(s1, _f32(g1), _f32(be1), eps)
        s2, hmid, pre = _mlp_fwd(a, shadow(wi, dt), _f32(bi), shadow(wout, dt), _f32(bout), act, a, d_o2)
        y, mean2, rstd2 = ops.layernorm_fwd(s2, _f32(g2), _f32(be2), eps, out=y_ext[:M1])
        ctx.save_for_backward(x_ext, band, wq, wk, wv, wo, g1, wi, wout, g2, qkv_ext, attn, lse, s1, mean1, rstd1,
                              a, hmid, pre, s2, mean2, rstd2)
        ctx.cfg = cfg
        return y

    @staticmethod
    def backward(ctx, dy):
        (x_ext, band, wq, wk, wv, wo, g1, wi, wout, g2, qkv_ext, attn, lse, s1, mean1, rstd1, a, hmid, pre, s2,
         mean2, rstd2) = ctx.saved_tensors
        B, N, H, eps, scale, w, act = ctx.cfg[:7]
        d_pos, d_attn, d_o1, d_o2 = ctx.cfg[8] if len(ctx.cfg) > 8 and ctx.cfg[8] is not None else (NO_DROP,) * 4
        M1 = B * N
        ME, D = x_ext.shape
        dh = D // H
        dt = x_ext.dtype
        dy = dy.contiguous()
        F = wi.shape[0]
        R = band.shape[0] if band is not None else 0
        direct = dt == torch.bfloat16      # the fp32 path's dW GEMMs allocate their own outputs
        s_wi, s_wout, s_wo = (grad_slot(q) if direct else None for q in (wi, wout, wo))
        zp = _ZeroPool(x_ext.device, D, D, D, D, D, D, D, F, D, 3 * D, R * 2 * D, 3 * D * D,
                       0 if s_wi is not None else F * D, 0 if s_wout is not None else D * F, 0 if s_wo is not None else D * D)
        dg2, dbe2, dbout = zp.take(D), zp.take(D), zp.take(D)
        if d_o2[0] > 0.0:
            # ds2 goes on to the residual as it is; dense_out sees it through the output dropout's mask
            ds2 = ops.layernorm_bwd(dy, s2, mean2, rstd2, _f32(g2), dg2, dbe2)
            ds2_lin = ops.dropout(ds2, d_o2)
        else:
            ds2 = ds2_lin = ops.layernorm_bwd(dy, s2, mean2, rstd2, _f32(g2), dg2, dbe2, dxsum=dbout)
        da, dwi, dbi, dwout, dbout = _mlp_bwd(ds2_lin, a, wi, wout, act, hmid, pre, ds2,
                                              db1=zp.take(F), db2=dbout, db2_done=d_o2[0] == 0.0,
                                              dw1=s_wi if s_wi is not None else zp.take(F, D),
                                              dw2=s_wout if s_wout is not None else zp.take(D, F))
        if dwi is s_wi:
            dwi = dwi.detach()          # a fresh alias of the slab view: autograd adopts it as .grad without a copy
        if dwout is s_wout:
            dwout = dwout.detach()
        dg1, dbe1, dbo = zp.take(D), zp.take(D), zp.take(D)
        ds1_ext = _ext_workspace('ds1', ME, D, dt, x_ext.device, (M1, R)) if band is not None else \
            torch.empty(ME, D, device=x_ext.device, dtype=dt)
        if d_o1[0] > 0.0:
            ds1 = ops.layernorm_bwd(da, s1, mean1, rstd1, _f32(g1), dg1, dbe1, out=ds1_ext[:M1])
            ds1_lin = ops.dropout(ds1, d_o1)
            _colsum(ds1_lin, out=dbo, side=True)
        else:
            ds1 = ds1_lin = ops.layernorm_bwd(da, s1, mean1, rstd1, _f32(g1), dg1, dbe1, out=ds1_ext[:M1], dxsum=dbo)
        dwo = _dw(ds1_lin, attn, out=s_wo if s_wo is not None else zp.take(D, D), side=True)
        if dwo is s_wo:
            dwo = dwo.detach()
        dattn = _dx(ds1_lin, wo)
        tables = dtab = None
        if band is not None:
            dqkv_ext = _ext_workspace('dqkv', ME, 3 * D, dt, x_ext.device, (M1, R))
            tables = qkv_ext[M1:M1 + R, :2 * D]
            dtab = zp.take(R, 2 * D)
        else:
            dqkv_ext = torch.empty(ME, 3 * D, device=x_ext.device, dtype=dt)
        dbqkv = zp.take(3 * D)             # token rows: summed by the attention backward; table rows: below
        _attn_bwd(qkv_ext[:M1], B, N, H, dh, tables, scale, w, attn, dattn, lse, dtab=dtab, dqkv=dqkv_ext[:M1], drop=d_attn,
                  dbias=dbqkv)
        if band is not None:
            dqkv_ext[M1:M1 + R, :2 * D].copy_(dtab)               # (dPq | dPk) in compute dtype
            ops.colsum(dtab, out=dbqkv[:2 * D])                   # the table rows went through the same projections
        dwqkv = _dw(dqkv_ext, x_ext, out=zp.take(3 * D, D), side=True)
        dx_ext = _dx(dqkv_ext, (wq, wk, wv), residual=ds1_ext)
        side_join(x_ext.device)
        dx = dx_ext[:M1]
        dband = dx_ext[M1:M1 + R] if band is not None else None
        if dband is not None and d_pos[0] > 0.0:
            dband = ops.dropout(dband, d_pos)         # the rows went in through pos_dropout
        dwq, dwk, dwv = dwqkv[:D], dwqkv[D:2 * D], dwqkv[2 * D:]
        dbq, dbk, dbv = dbqkv[:D], dbqkv[D:2 * D], dbqkv[2 * D:]
        return (dx, dband, dwq, dbq, dwk, dbk, dwv, dbv, dwo, dbo, dg1, dbe1, dwi, dbi, dwout, dbout, dg2, dbe2,
                None)


class DisAttentionFn(torch.autograd.Function):
    """DisentangledSelfAttention.forward's 'hidden_states' (disentangled_transformer.py:301-355,
    LAI twin :851-927) for q = k = value = x: three projections, c2p / p2c bias, softmax, PV."""

    @staticmethod
    def forward(ctx, x, band, wq, bq, wk, bk, wv, bv, cfg):
        B, N, H, scale, w = cfg[:5]
        drop = cfg[5] if len(cfg) > 5 else NO_DROP
        D = x.shape[1]
        dt = x.dtype
        wqkv = shadow((wq, wk, wv), dt)
        bqkv = torch.cat([bq.detach(), bk.detach(), bv.detach()])
        qkv = ops.gemm(x, wqkv, bias=bqkv)
        tables = ops.gemm(band, wqkv[:2 * D], bias=bqkv[:2 * D]) if band is not None else None
        attn, lse = _attn_fwd(qkv, B, N, H, D // H, tables, scale, w, drop)
        ctx.save_for_backward(x, band, wq, wk, wv, qkv, tables, attn, lse)
        ctx.cfg = cfg
        return attn

    @staticmethod
    def backward(ctx, dattn):
        x, band, wq, wk, wv, qkv, tables, attn, lse = ctx.saved_tensors
        B, N, H, scale, w = ctx.cfg[:5]
        drop = ctx.cfg[5] if len(ctx.cfg) > 5 else NO_DROP
        D = x.shape[1]
        dt = x.dtype
        dqkv, dtab = _attn_bwd(qkv, B, N, H, D // H, tables, scale, w, attn, dattn.contiguous(), lse, drop=drop)
        dbqkv = ops.colsum(dqkv)
        dwqkv = _dw(dqkv, x)
        dx = _dx(dqkv, (wq, wk, wv))
        dband = None
        if band is not None:
            dtab_t = dtab.to(dt)
            ops.colsum(dtab_t, out=dbqkv[:2 * D])
            ops.gemm(dtab_t, band, trans_a=True, trans_b=False, out=dwqkv[:2 * D], accumulate=True)
            dband = _dx(dtab_t, (wq, wk))
        return (dx, dband, dwqkv[:D], dbqkv[:D], dwqkv[D:2 * D], dbqkv[D:2 * D], dwqkv[2 * D:], dbqkv[2 * D:], None)


# --------------------------------------------------------------------------- patch embedding
class PatchEmbedFn(torch.autograd.Function):
    """tokens[b] = [cls (+ pos[0]); Conv2d(k=s=P)(img[b]) flattened (+ pos[1:])]  as ONE GEMM over
    gathered patches with the bias / position add / row placement in its epilogue
    (model.py:52-55,92-99; timm_locality_aware_transformer.py:133-138)."""

    @staticmethod
    def forward(ctx, img, weight, bias, cls_token, pos_embed, dtype, norm=None):
        D = weight.shape[0]
        P = weight.shape[2]
        if img.dtype == torch.uint8:
            # raw pixels: ToTensor + Normalize ride in the gather (train.py:340-341,362-363); norm = (mean, std, layout)
            if norm is None:
                raise ValueError('uint8 images need the input normalisation (vit_amd.data.set_input_norm)')
            nhwc = len(norm) > 2 and norm[2] == 'nhwc'
            B, C, Hh, Ww = (img.shape[0], img.shape[3], img.shape[1], img.shape[2]) if nhwc else img.shape
            patches = ops.patchify_u8(img, P, dtype, norm[0], norm[1], channels_last=nhwc)
        else:
            B, C, Hh, Ww = img.shape
            patches = ops.patchify(img, P, dtype)
        npatch = (Hh // P) * (Ww // P)
        N = npatch + 1
        w2d = shadow(weight, dtype).view(D, C * P * P)
        pos_t = None
        if pos_embed is not None:
            pos_t = shadow(pos_embed, dtype).view(N, D)
        tok = torch.empty(B * N, D, device=img.device, dtype=dtype)
        ops.gemm(patches, w2d, bias=_f32(bias), residual=pos_t, res_period=npatch if pos_t is not None else 0,
                 res_off=1, out=tok, grp=(npatch, N, 1))
        ops.cls_rows(tok.view(B, N, D), cls_token.detach().view(D),
                     None if pos_embed is None else pos_embed.detach().view(N, D))
        ctx.save_for_backward(patches, weight)
        ctx.dims = (B, C, Hh, Ww, P, N, D)
        ctx.has_pos = pos_embed is not None
        ctx.has_bias = bias is not None
        return tok

    @staticmethod
    def backward(ctx, dtok):
        patches, weight = ctx.saved_tensors
        B, C, Hh, Ww, P, N, D = ctx.dims
        dtok3 = dtok.contiguous().view(B, N, D)
        dpatch_rows = dtok3[:, 1:].reshape(B * (N - 1), D)
        dw = _dw(dpatch_rows, patches).view(weight.shape)
        db = ops.colsum(dpatch_rows) if ctx.has_bias else None
        dcls = torch.zeros(D, device=dtok.device, dtype=torch.float32)
        dpos = torch.zeros(N, D, device=dtok.device, dtype=torch.float32) if ctx.has_pos else None
        ops.token_grads(dtok3, dpos, dcls)
        dimg = None
        if ctx.needs_input_grad[0]:
            wt = shadow(weight, dtok.dtype).view(D, C * P * P)
            dimg = ops.unpatchify(_dx_nn(dpatch_rows, wt), B, C, Hh, Ww, P)
        return (dimg, dw, db, dcls.view(1, 1, D), None if dpos is None else dpos.view(1, N, D), None, None)


def patch_embed(img, weight, bias, cls_token, pos_embed, dtype, norm=None):
    """img: float [B,C,H,W] (already normalised, as the reference's loader delivers it), or uint8 [B,C,H,W] /
    [B,H,W,C] with ``norm`` = (mean, std[, 'nchw' | 'nhwc'])."""
    return PatchEmbedFn.apply(img, weight, bias, cls_token, pos_embed, dtype, norm)


# --------------------------------------------------------------------------- MLP
class MlpFn(torch.autograd.Function):
    """act(x W1^T + b1) W2^T + b2 with the activation fused into the GEMM epilogues."""

    @staticmethod
    def forward(ctx, x, w1, b1, w2, b2, act, add_input=False, drop1=NO_DROP, drop2=NO_DROP):
        dt = x.dtype
        ctx.act = act
        ctx.add_input = add_input
        ctx.drops = (drop1, drop2)
        if drop1[0] == 0.0 and drop2[0] == 0.0:
            y, hmid, pre = _mlp_fwd(x, shadow(w1, dt), _f32(b1), shadow(w2, dt), _f32(b2), act, x if add_input else None)
            ctx.save_for_backward(x, w1, w2, hmid, pre)
            return y
        # drop1 after the activation (timm Mlp's first drop, model.py:202 self.dropout), drop2 on the output (timm Mlp's
        # second drop / dropout2 / dropout_out), the residual added after it
        fwd_code, _ = _act_codes(act)
        pre = torch.empty(x.shape[0], w1.shape[0], device=x.device, dtype=dt) if act == 'gelu' else None
        hmid = ops.gemm(x, shadow(w1, dt), bias=_f32(b1), act=fwd_code, aux_out=pre)
        hd = ops.dropout(hmid, drop1) if drop1[0] > 0.0 else hmid
        res = x if add_input else None
        if drop2[0] > 0.0:
            y = ops.dropout(ops.gemm(hd, shadow(w2, dt), bias=_f32(b2)), drop2, res=res)
        else:
            y = ops.gemm(hd, shadow(w2, dt), bias=_f32(b2), residual=res)
        ctx.save_for_backward(x, w1, w2, hmid, pre)      # the dropped activation is regenerated in backward
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w1, w2, hmid, pre = ctx.saved_tensors
        dy = dy.contiguous()
        drop1, drop2 = ctx.drops
        if drop1[0] == 0.0 and drop2[0] == 0.0:
            dx, dw1, db1, dw2, db2 = _mlp_bwd(dy, x, w1, w2, ctx.act, hmid, pre, dy if ctx.add_input else None)
            return dx, dw1, db1, dw2, db2, None, None, None, None
        _, bwd_code = _act_codes(ctx.act)
        dyl = ops.dropout(dy, drop2) if drop2[0] > 0.0 else dy
        hd = ops.dropout(hmid, drop1) if drop1[0] > 0.0 else hmid
        db2 = ops.colsum(dyl)
        dw2 = _dw(dyl, hd)
        # the activation's derivative and the first dropout are both element-wise factors: they commute
        dpre = _dx(dyl, w2, act=bwd_code, aux_in=pre if ctx.act == 'gelu' else hmid)
        if drop1[0] > 0.0:
            dpre = ops.dropout(dpre, drop1)
        db1 = ops.colsum(dpre)
        dw1 = _dw(dpre, x)
        dx = _dx(dpre, w1, residual=dy if ctx.add_input else None)
        return dx, dw1, db1, dw2, db2, None, None, None, None


def mlp(x2d, w1, b1, w2, b2, act='gelu', add_input=False, p_act=0.0, p_out=0.0):
    """add_input: y = mlp(x) + x (the post-LN layers' residual, fused into the fc2 epilogue).  p_act / p_out: dropout
    after the activation / on the output (before the residual); each draws a stream when > 0."""
    return MlpFn.apply(x2d, w1, b1, w2, b2, act, add_input, VD.next_stream(p_act), VD.next_stream(p_out))


# --------------------------------------------------------------------------- FAVOR+ (Performer)
class FavorAttentionFn(torch.autograd.Function):
    """FAVOR+ linear attention on a packed qkv buffer: feature maps (softmax / relu kernel, with
    the reference's stabilisers) + KV contraction + normaliser, forward and backward in HIP."""

    @staticmethod
    def forward(ctx, qkv, w, B, L, H, dh, kind, ratio, eps):
        D = H * dh
        q, k, v = qkv[:, :D], qkv[:, D:2 * D], qkv[:, 2 * D:]
        ws = ops.favor_workspace(qkv, B, H, dh, w.shape[0])
        out = ops.favor_fwd(q, k, v, B, H, L, dh, qkv.stride(0), w, kind, ratio, eps, ws)
        ctx.save_for_backward(qkv, w, out, ws)
        ctx.cfg = (B, L, H, dh, kind, ratio, eps)
        return out

    @staticmethod
    def backward(ctx, dout):
        qkv, w, out, ws = ctx.saved_tensors
        B, L, H, dh, kind, ratio, eps = ctx.cfg
        D = H * dh
        q, k, v = qkv[:, :D], qkv[:, D:2 * D], qkv[:, 2 * D:]
        dqkv = torch.empty_like(qkv)
        ops.favor_bwd(q, k, v, B, H, L, dh, qkv.stride(0), w, kind, ratio, eps, out, dout.contiguous(),
                      dqkv[:, :D], dqkv[:, D:2 * D], dqkv[:, 2 * D:], 3 * D, ws)
        return dqkv, None, None, None, None, None, None, None, None


class FavorPosAttentionFn(torch.autograd.Function):
    """FAVOR+ with position features (DisentangledSelfAttentionPerformer, reference dis :586-602):
    new_Q = q' + Pq, new_K = k' + Pk with Pq / Pk the query- / key-form feature maps of ``posqk``
    = [pos_query_proj(pos_embed) | pos_key_proj(pos_embed)]  ([L, 2D])."""

    @staticmethod
    def forward(ctx, qkv, posqk, w, B, L, H, dh, kind, ratio, eps):
        D = H * dh
        q, k, v = qkv[:, :D], qkv[:, D:2 * D], qkv[:, 2 * D:]
        pq, sq = ops.favor_pos_fwd(posqk[:, :D], H, dh, w, kind, True, ratio, eps)
        pk, sk = ops.favor_pos_fwd(posqk[:, D:], H, dh, w, kind, False, ratio, eps)
        ws = ops.favor_workspace(qkv, B, H, dh, w.shape[0])
        out = ops.favor_pos_attn_fwd(q, k, v, B, H, L, dh, qkv.stride(0), w, kind, ratio, eps, pq, pk, ws)
        ctx.save_for_backward(qkv, posqk, w, out, ws, pq, pk, sq, sk)
        ctx.cfg = (B, L, H, dh, kind, ratio, eps)
        return out

    @staticmethod
    def backward(ctx, dout):
        qkv, posqk, w, out, ws, pq, pk, sq, sk = ctx.saved_tensors
        B, L, H, dh, kind, ratio, eps = ctx.cfg
        D = H * dh
        q, k, v = qkv[:, :D], qkv[:, D:2 * D], qkv[:, 2 * D:]
        dqkv = torch.empty_like(qkv)
        dnq, dnk = ops.favor_pos_attn_bwd(q, k, v, B, H, L, dh, qkv.stride(0), w, kind, ratio, eps, pq, pk, out,
                                          dout.contiguous(), dqkv[:, :D], dqkv[:, D:2 * D], dqkv[:, 2 * D:], 3 * D, ws)
        dpos = torch.cat([ops.favor_pos_bwd(posqk[:, :D], H, dh, w, kind, True, ratio, eps, dnq, sq),
                          ops.favor_pos_bwd(posqk[:, D:], H, dh, w, kind, False, ratio, eps, dnk, sk)], dim=1)
        return dqkv, dpos, None, None, None, None, None, None, None, None


def _favor_kind(kernel):
    if kernel == 'softmax':
        return ops.FAVOR_SOFTMAX, 1e-6
    if kernel == 'relu':
        return ops.FAVOR_RELU, 1e-3
    raise ValueError(f'unknown FAVOR+ kernel {kernel!r}')


def favor_pos_attention(qkv, posqk, w, B, L, H, dh, kernel, ratio):
    """qkv [B*L, 3D], posqk [L, 2D] (position query | key projections), w [m, dh]."""
    kind, eps = _favor_kind(kernel)
    return FavorPosAttentionFn.apply(qkv, posqk, w.to(qkv.dtype).contiguous(), B, L, H, dh, kind, float(ratio), eps)


def favor_attention(qkv, w, B, L, H, dh, kernel, ratio):
    """kernel: 'softmax' (eps 1e-6) or 'relu' (eps 1e-3), as the reference's two feature maps."""
    kind, eps = _favor_kind(kernel)
    return FavorAttentionFn.apply(qkv, w.to(qkv.dtype).contiguous(), B, L, H, dh, kind, float(ratio), eps)


# --------------------------------------------------------------------------- loss
class LabelSmoothingCEFn(torch.autograd.Function):
    """timm 0.4.5 LabelSmoothingCrossEntropy (train.py:273,541), mean over the batch: two small HIP kernels forward,
    one backward (the softmax is recomputed from the saved log-sum-exp)."""

    @staticmethod
    def forward(ctx, logits, target, smoothing):
        loss, lse = ops.ce_smooth_fwd(logits, target, smoothing)
        ctx.save_for_backward(logits, target, lse)
        ctx.smoothing = smoothing
        return loss

    @staticmethod
    def backward(ctx, g):
        logits, target, lse = ctx.saved_tensors
        return ops.ce_smooth_bwd(logits, target, ctx.smoothing, lse, g.float().contiguous()), None, None


def label_smoothing_ce(logits, target, smoothing=0.1):
    if logits.dtype != torch.float32 or logits.stride(-1) != 1:
        logits = logits.float().contiguous()
    if target.dtype != torch.int64:
        target = target.long()
    return LabelSmoothingCEFn.apply(logits, target.contiguous(), float(smoothing))
