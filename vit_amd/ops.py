"""Thin Python wrappers over the C-ABI (raw pointers + the current HIP stream).

No autograd here and no fallbacks: these take CUDA(HIP) tensors, check layouts on the host
(a faulting kernel can take the whole box down) and launch.  Autograd lives in functional.py.
"""
import torch

from . import _lib
from ._lib import F32, BF16, ACT_NONE, ACT_GELU, ACT_RELU, ACT_DGELU, ACT_DRELU, check, lib  # noqa: F401


# bench.py sets this to a list to collect (kernel, algorithmic flops, start event, end event) per
# launch: HIP events recorded on the launch stream (torch's current stream) around the kernel.
KERNEL_TIMER = None      # bench.py: a list that receives (name, work, event0, event1, unit) per timed launch


def _t0():
    if KERNEL_TIMER is None:
        return None
    e = torch.cuda.Event(enable_timing=True)
    e.record()
    return e


def _t1(e0, name, work, unit):
    """work: algorithmic FLOPs (unit 'flop') or algorithmic HBM bytes (unit 'byte') of the launch(es) since _t0()."""
    if e0 is not None and KERNEL_TIMER is not None:
        e1 = torch.cuda.Event(enable_timing=True)
        e1.record()
        KERNEL_TIMER.append((name, float(work), e0, e1, unit))


def set_reserved_cus(n):
    """CUs the persistent GEMM launches leave free for other kernels (RCCL's reductions while gradient buckets are in
    flight): vitk_set_reserved_cus.  Process-wide; returns the previous value."""
    old = lib().vitk_get_reserved_cus()
    check(lib().vitk_set_reserved_cus(int(n)), 'vitk_set_reserved_cus')
    return old


def _dt(t):
    if t.dtype == torch.float32:
        return F32
    if t.dtype == torch.bfloat16:
        return BF16
    raise TypeError(f'vit_amd supports float32 / bfloat16 tensors, got {t.dtype}')


_raw_stream = getattr(torch._C, '_cuda_getCurrentRawStream', None)


def _stream():
    """hipStream_t of torch's current stream (the raw getter is ~20x cheaper than building a Stream object:
    it is called once per kernel launch)."""
    if _raw_stream is not None:
        return _raw_stream(torch.cuda.current_device())
    return torch.cuda.current_stream().cuda_stream


def _ptr(t):
    return 0 if t is None else t.data_ptr()


def _req_cuda(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise _lib.VitkError('vit_amd ops need tensors on the GPU (there is no CPU path)')


def _rows2d(t, what):
    """(ptr, rows, cols, ld) of a 2-D row-major view whose last dim is contiguous."""
    if t.dim() != 2 or t.stride(1) != 1 and t.shape[1] != 1:
        raise ValueError(f'{what}: need a 2-D tensor with contiguous rows, got {tuple(t.shape)} strides {t.stride()}')
    ld = t.stride(0) if t.shape[0] > 1 else max(t.shape[1], t.stride(0))
    return t.data_ptr(), t.shape[0], t.shape[1], ld


def gemm(a, b, *, trans_a=False, trans_b=True, out=None, out_dtype=None, bias=None, act=ACT_NONE,
         aux_in=None, aux_out=None, residual=None, res_period=0, res_off=0, out_rows=None, grp=None,
         accumulate=False, splitk=0, alpha=1.0):
    """out = epilogue(alpha * op(a) @ op(b) + bias) [+ residual].

    op(a) is [M,K]: ``a`` stored [M,K] (trans_a False) or [K,M] (True).
    op(b) is [K,N]: ``b`` stored [N,K] (trans_b True, the nn.Linear weight layout) or [K,N].
    grp = (grp_rows, grp_stride, grp_off) remaps output row r to
    (r // grp_rows) * grp_stride + grp_off + r % grp_rows (patch rows into a token matrix).
    """
    _req_cuda(a, b, out, bias, aux_in, aux_out, residual)
    pa, ra, ca, lda = _rows2d(a, 'gemm a')
    pb, rb, cb, ldb = _rows2d(b, 'gemm b')
    M, K = (ca, ra) if trans_a else (ra, ca)
    N, Kb = (rb, cb) if trans_b else (cb, rb)
    if K != Kb:
        raise ValueError(f'gemm: inner dims differ ({K} vs {Kb})')
    if a.dtype != b.dtype:
        raise TypeError('gemm: a and b must share a dtype')
    if out is None:
        odt = out_dtype or a.dtype
        rows = out_rows if out_rows is not None else M
        out = torch.empty(rows, N, device=a.device, dtype=odt)
    pc, rc, cc, ldc = _rows2d(out, 'gemm out')
    if cc != N:
        raise ValueError('gemm: out has the wrong width')
    grp_rows, grp_stride, grp_off = grp or (0, 0, 0)
    if grp_rows:
        last = ((M - 1) // grp_rows) * grp_stride + grp_off + (M - 1) % grp_rows
        if last >= rc:
            raise ValueError('gemm: grouped output rows exceed out')
    elif rc < M:
        raise ValueError('gemm: out has too few rows')
    for t, nm in ((aux_in, 'aux_in'), (aux_out, 'aux_out'), (residual, 'residual')):
        if t is not None and t.dtype != a.dtype:
            raise TypeError(f'gemm: {nm} must have the input dtype')
    ld_aux = 0
    for t in (aux_in, aux_out):
        if t is not None:
            _, r_, c_, ld_aux = _rows2d(t, 'gemm aux')
            if c_ != N or r_ < rc:
                raise ValueError('gemm: aux shape mismatch')
    ldr = 0
    if residual is not None:
        _, rr, cr, ldr = _rows2d(residual, 'gemm residual')
        need = (res_off + min(res_period, M)) if res_period else rc
        if cr != N or rr < need:
            raise ValueError('gemm: residual shape mismatch')
    if bias is not None and (bias.dtype != torch.float32 or bias.numel() != N or not bias.is_contiguous()):
        raise ValueError('gemm: bias must be contiguous fp32 [N]')
    e0 = _t0()
    st = lib().vitk_gemm(_dt(a), _dt(out), int(trans_a), int(trans_b), M, N, K, float(alpha),
                         pa, lda, pb, ldb, pc, ldc, _ptr(bias), act, _ptr(aux_in), _ptr(aux_out), ld_aux,
                         _ptr(residual), ldr, res_period, res_off, grp_rows, grp_stride, grp_off,
                         int(accumulate), splitk, _stream())
    check(st, 'vitk_gemm')
    if e0 is not None:
        layout = ('T' if trans_a else 'N') + ('T' if trans_b else 'N')
        _t1(e0, 'gemm_' + layout + ('_bf16' if a.dtype == torch.bfloat16 else '_f32'), 2.0 * M * N * K, 'flop')
    return out


def layernorm_fwd(x, gamma, beta, eps, want_stats=True, out=None):
    _req_cuda(x, gamma, beta, out)
    px, rows, D, ldx = _rows2d(x, 'layernorm x')
    y = out if out is not None else torch.empty(rows, D, device=x.device, dtype=x.dtype)
    if y.shape != (rows, D) or y.dtype != x.dtype or not y.is_contiguous():
        raise ValueError('layernorm_fwd: out must be a contiguous [rows, D] tensor of the input dtype')
    mean = torch.empty(rows, device=x.device, dtype=torch.float32) if want_stats else None
    rstd = torch.empty(rows, device=x.device, dtype=torch.float32) if want_stats else None
    st = lib().vitk_layernorm_fwd(_dt(x), rows, D, px, ldx, gamma.data_ptr(), beta.data_ptr(), float(eps),
                                  y.data_ptr(), D, _ptr(mean), _ptr(rstd), _stream())
    check(st, 'vitk_layernorm_fwd')
    return y, mean, rstd


def layernorm_bwd(dy, x, mean, rstd, gamma, dgamma, dbeta, add=None, out=None, dxsum=None):
    """Returns dx (+ add).  dgamma / dbeta (fp32 [D]) are accumulated into, as is dxsum (fp32 [D],
    column sums of the returned dx) when given."""
    _req_cuda(dy, x, mean, rstd, gamma, dgamma, dbeta, add, out, dxsum)
    pdy, rows, D, lddy = _rows2d(dy, 'layernorm dy')
    px, _, _, ldx = _rows2d(x, 'layernorm x')
    dx = out if out is not None else torch.empty(rows, D, device=x.device, dtype=x.dtype)
    if dx.shape != (rows, D) or dx.dtype != x.dtype or not dx.is_contiguous():
        raise ValueError('layernorm_bwd: out must be a contiguous [rows, D] tensor of the input dtype')
    padd, ldadd = 0, 0
    if add is not None:
        padd, ra, ca, ldadd = _rows2d(add, 'layernorm add')
        if (ra, ca) != (rows, D) or add.dtype != x.dtype:
            raise ValueError('layernorm_bwd: add shape/dtype mismatch')
    if x.shape != dy.shape or x.dtype != dy.dtype:
        raise ValueError('layernorm_bwd: x / dy mismatch')
    st = lib().vitk_layernorm_bwd(_dt(x), rows, D, pdy, lddy, px, ldx, mean.data_ptr(), rstd.data_ptr(),
                                  gamma.data_ptr(), padd, ldadd, dx.data_ptr(), D, _ptr(dgamma), _ptr(dbeta),
                                  _ptr(dxsum), _stream())
    check(st, 'vitk_layernorm_bwd')
    return dx


def adamw_step(p, g, m, v, lr, beta1, beta2, eps, weight_decay, step, shadow=None, shadow_t=None):
    """In-place torch.optim.AdamW step of one fp32 parameter (>= 2-D, viewed as [shape[0], -1]); also
    rewrites its bf16 copies: ``shadow`` = (tensor, row offset) in the [out, in] layout, ``shadow_t`` =
    (tensor, column offset) in the transposed layout (both optional)."""
    _req_cuda(p, g, m, v)
    rows = p.shape[0]
    cols = p.numel() // rows if rows else 0
    for t in (p, g, m, v):
        if t.dtype != torch.float32 or not t.is_contiguous() or t.shape != p.shape:
            raise ValueError('adamw_step: p, g, m, v must be contiguous fp32 tensors of one shape')
    ps = pt = 0
    lds = ldt = 0
    if shadow is not None:
        buf, off = shadow
        _req_cuda(buf)
        if buf.dtype != torch.bfloat16 or buf.dim() != 2 or buf.stride(1) != 1 or buf.shape[1] != cols \
                or off < 0 or off + rows > buf.shape[0]:
            raise ValueError('adamw_step: shadow must be a bf16 [>= off + rows, cols] matrix')
        lds = buf.stride(0)
        ps = buf.data_ptr() + off * lds * 2
    if shadow_t is not None:
        buf, off = shadow_t
        _req_cuda(buf)
        if buf.dtype != torch.bfloat16 or buf.dim() != 2 or buf.stride(1) != 1 or buf.shape[0] != cols \
                or off < 0 or off + rows > buf.shape[1]:
            raise ValueError('adamw_step: shadow_t must be a bf16 [cols, >= off + rows] matrix')
        ldt = buf.stride(0)
        pt = buf.data_ptr() + off * 2
    st = lib().vitk_adamw(rows, cols, p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(), float(lr), float(beta1),
                          float(beta2), float(eps), float(weight_decay), int(step), ps, lds, pt, ldt, _stream())
    check(st, 'vitk_adamw')


def colsum(x, out=None):
    """out[n] += sum_m x[m][n] (fp32).  A fresh zeroed ``out`` is made when none is given."""
    _req_cuda(x, out)
    px, M, N, ldx = _rows2d(x, 'colsum x')
    if out is None:
        out = torch.zeros(N, device=x.device, dtype=torch.float32)
    st = lib().vitk_colsum(_dt(x), M, N, px, ldx, out.data_ptr(), _stream())
    check(st, 'vitk_colsum')
    return out


def cast(src, dst):
    """fp32 -> bf16 or bf16 -> fp32 over flat contiguous buffers."""
    _req_cuda(src, dst)
    if not (src.is_contiguous() and dst.is_contiguous()) or src.numel() != dst.numel():
        raise ValueError('cast: need contiguous buffers of equal length')
    if src.dtype == torch.float32 and dst.dtype == torch.bfloat16:
        d = 0
    elif src.dtype == torch.bfloat16 and dst.dtype == torch.float32:
        d = 1
    else:
        raise TypeError('cast: f32<->bf16 only')
    check(lib().vitk_cast(d, src.numel(), src.data_ptr(), dst.data_ptr(), _stream()), 'vitk_cast')
    return dst


def patchify(img, P, out_dtype):
    _req_cuda(img)
    B, C, H, W = img.shape
    img = img.contiguous()
    out = torch.empty(B * (H // P) * (W // P), C * P * P, device=img.device, dtype=out_dtype)
    check(lib().vitk_patchify(_dt(img), _dt(out), B, C, H, W, P, img.data_ptr(), out.data_ptr(), _stream()),
          'vitk_patchify')
    return out


def patchify_u8(img, P, out_dtype, mean, std, channels_last=False):
    """Patches [B*nph*npw, C*P*P] from uint8 images ([B,C,H,W], or [B,H,W,C] when channels_last) with
    x/255 and (x - mean[c]) / std[c] applied on the way (the loader's ToTensor + Normalize)."""
    import ctypes
    _req_cuda(img)
    if img.dtype != torch.uint8 or img.dim() != 4:
        raise ValueError('patchify_u8: uint8 [B,C,H,W] / [B,H,W,C] images expected')
    if channels_last:
        B, H, W, C = img.shape
    else:
        B, C, H, W = img.shape
    if len(mean) != C or len(std) != C:
        raise ValueError(f'patchify_u8: {C} channels need {C} mean / std values')
    img = img.contiguous()
    out = torch.empty(B * (H // P) * (W // P), C * P * P, device=img.device, dtype=out_dtype)
    arr = ctypes.c_float * C
    m, s = arr(*[float(v) for v in mean]), arr(*[float(v) for v in std])
    check(lib().vitk_patchify_u8(_dt(out), B, C, H, W, P, int(bool(channels_last)), img.data_ptr(),
                                 ctypes.cast(m, ctypes.c_void_p), ctypes.cast(s, ctypes.c_void_p), out.data_ptr(),
                                 _stream()), 'vitk_patchify_u8')
    return out


def unpatchify(dpatch, B, C, H, W, P):
    _req_cuda(dpatch)
    dpatch = dpatch.contiguous()
    dimg = torch.empty(B, C, H, W, device=dpatch.device, dtype=torch.float32)
    check(lib().vitk_unpatchify(_dt(dpatch), B, C, H, W, P, dpatch.data_ptr(), dimg.data_ptr(), _stream()),
          'vitk_unpatchify')
    return dimg


def cls_rows(x, cls, pos):
    """x [B,N,D] (contiguous): x[:,0] = cls + pos[0]."""
    B, N, D = x.shape
    check(lib().vitk_cls_rows(_dt(x), B, N, D, cls.data_ptr(), _ptr(pos), x.data_ptr(), _stream()), 'vitk_cls_rows')


def token_grads(dx, dpos, dcls):
    B, N, D = dx.shape
    check(lib().vitk_token_grads(_dt(dx), B, N, D, dx.data_ptr(), _ptr(dpos), _ptr(dcls), _stream()),
          'vitk_token_grads')


NO_DROP = (0.0, 0, 0)     # (p, seed, offset) of a dropout site that is off


def attn_fwd(qkv_q, qkv_k, qkv_v, B, H, N, dh, ld, pk, pq, ld_p, scale, band_w, drop=NO_DROP):
    """q/k/v: views into [B*N, ld]-strided buffers (element (b,n,h,d) at (b*N+n)*ld + h*dh + d).
    pk/pq: band tables [2N-1, ld_p] (row r <-> i-j = r-(N-1)) or None.  drop = (p, seed, offset): dropout on the
    probabilities (vit_amd.dropout).  Returns (out [B*N, H*dh], lse [B,H,N])."""
    _req_cuda(qkv_q, qkv_k, qkv_v, pk, pq)
    out = torch.empty(B * N, H * dh, device=qkv_q.device, dtype=qkv_q.dtype)
    lse = torch.empty(B, H, N, device=qkv_q.device, dtype=torch.float32)
    e0 = _t0()
    st = lib().vitk_attn_fwd(_dt(qkv_q), B, H, N, dh, qkv_q.data_ptr(), qkv_k.data_ptr(), qkv_v.data_ptr(), ld,
                             _ptr(pk), _ptr(pq), ld_p, float(scale), int(band_w),
                             out.data_ptr(), H * dh, lse.data_ptr(), float(drop[0]), int(drop[1]), int(drop[2]), _stream())
    check(st, 'vitk_attn_fwd')
    _t1(e0, 'attn_fwd', 4.0 * B * N * H * dh * out.element_size() + 4.0 * lse.numel(), 'byte')   # q, k, v in, out + lse out
    return out, lse


def attn_bwd(q, k, v, B, H, N, dh, ld, pk, pq, ld_p, scale, band_w, out, dout, lse, dq, dk, dv, ld_d, dpk, dpq,
             workspace=None, ld_dp=None, drop=NO_DROP, dbias=None):
    """dq/dk/dv: views into a [B*N, ld_d] buffer (same addressing as q/k/v).  dpk/dpq: fp32
    [2N-1, H*dh] (row stride ld_dp, default H*dh), ACCUMULATED into (caller zeroes).  ``workspace``: optional uint8 scratch of at
    least attn_bwd_workspace_bytes(...) (allocated when not given).  ``dbias``: optional fp32 [3*H*dh], += column sums
    of dq | dk | dv (the q/k/v bias gradient)."""
    _req_cuda(q, k, v, pk, pq, out, dout, lse, dq, dk, dv, dpk, dpq, workspace, dbias)
    if dbias is not None and (dbias.dtype != torch.float32 or dbias.numel() != 3 * H * dh or not dbias.is_contiguous()):
        raise ValueError('attn_bwd: dbias must be contiguous fp32 [3*H*dh]')
    need = lib().vitk_attn_bwd_workspace_bytes(_dt(q), B, H, N, dh)
    if workspace is None:
        workspace = torch.empty(need, device=q.device, dtype=torch.uint8)
    elif workspace.numel() * workspace.element_size() < need:
        raise ValueError('attn_bwd: workspace too small')
    if not (out.is_contiguous() and dout.is_contiguous()) or out.shape != (B * N, H * dh) or dout.shape != out.shape:
        raise ValueError('attn_bwd: out / dout must be contiguous [B*N, H*dh]')
    e0 = _t0()
    st = lib().vitk_attn_bwd(_dt(q), B, H, N, dh, q.data_ptr(), k.data_ptr(), v.data_ptr(), ld,
                             _ptr(pk), _ptr(pq), ld_p, float(scale), int(band_w),
                             out.data_ptr(), dout.data_ptr(), H * dh, lse.data_ptr(),
                             dq.data_ptr(), dk.data_ptr(), dv.data_ptr(), ld_d,
                             _ptr(dpk), _ptr(dpq), H * dh if ld_dp is None else ld_dp, workspace.data_ptr(),
                             float(drop[0]), int(drop[1]), int(drop[2]), _ptr(dbias), _stream())
    check(st, 'vitk_attn_bwd')
    _t1(e0, 'attn_bwd', 8.0 * B * N * H * dh * out.element_size() + 4.0 * lse.numel(), 'byte')   # q, k, v, out, dout in, dq, dk, dv out


def dropout(x, drop, res=None, out=None, rows_per_sample=0):
    """y = x * keep / (1 - p) [+ res] on 2-D rows; drop = (p, seed, offset).  Element (r, c) is element r * cols + c of
    the stream, so the backward is the same call on dy.  rows_per_sample > 0: DropPath -- one decision per sample
    (rows [s * rows_per_sample, (s + 1) * rows_per_sample) share element s of the stream)."""
    _req_cuda(x, res, out)
    px, rows, cols, ldx = _rows2d(x, 'dropout x')
    if out is None:
        out = torch.empty(rows, cols, device=x.device, dtype=x.dtype)
    py, ry, cy, ldy = _rows2d(out, 'dropout out')
    if (ry, cy) != (rows, cols) or out.dtype != x.dtype:
        raise ValueError('dropout: out must match x')
    pr = ldr = 0
    if res is not None:
        pr, rr, cr, ldr = _rows2d(res, 'dropout res')
        if (rr, cr) != (rows, cols) or res.dtype != x.dtype:
            raise ValueError('dropout: res must match x')
    if rows_per_sample > 0:
        st = lib().vitk_drop_path(_dt(x), rows, cols, int(rows_per_sample), px, ldx, pr, ldr, py, ldy, float(drop[0]),
                                  int(drop[1]), int(drop[2]), _stream())
        check(st, 'vitk_drop_path')
        return out
    st = lib().vitk_dropout(_dt(x), rows, cols, px, ldx, pr, ldr, py, ldy, float(drop[0]), int(drop[1]), int(drop[2]),
                            _stream())
    check(st, 'vitk_dropout')
    return out


def dropout_mask(n, drop, device='cuda'):
    """keep[i] (uint8) of the first n elements of the stream drop = (p, seed, offset): what the kernels apply (test hook)."""
    out = torch.empty(n, device=device, dtype=torch.uint8)
    check(lib().vitk_dropout_mask(n, float(drop[0]), int(drop[1]), int(drop[2]), out.data_ptr(), _stream()),
          'vitk_dropout_mask')
    return out


FAVOR_SOFTMAX, FAVOR_RELU = 0, 1


def favor_workspace(q, B, H, dh, m):
    n = lib().vitk_favor_workspace_bytes(_dt(q), B, H, dh, m)
    return torch.empty(n, device=q.device, dtype=torch.uint8)


def favor_fwd(q, k, v, B, H, L, dh, ld, w, kind, ratio, eps, workspace):
    """FAVOR+ attention forward on views of a packed qkv buffer (element (b,l,h,d) at (b*L+l)*ld + h*dh + d).
    w: [m, dh] in the compute dtype.  ``workspace`` (favor_workspace) receives KV / ksum / the key
    max and must be handed unchanged to favor_bwd.  Returns out [B*L, H*dh]."""
    _req_cuda(q, k, v, w, workspace)
    if w.dtype != q.dtype or not w.is_contiguous() or w.shape[1] != dh:
        raise ValueError('favor: w must be contiguous [m, dh] in the compute dtype')
    out = torch.empty(B * L, H * dh, device=q.device, dtype=q.dtype)
    e0 = _t0()
    st = lib().vitk_favor_fwd(_dt(q), kind, B, H, L, dh, w.shape[0], q.data_ptr(), k.data_ptr(), v.data_ptr(), ld,
                              w.data_ptr(), float(ratio), float(eps), out.data_ptr(), H * dh,
                              workspace.data_ptr(), _stream())
    check(st, 'vitk_favor_fwd')
    _t1(e0, 'favor_fwd', 4.0 * B * L * H * dh * q.element_size(), 'byte')     # q, k, v read once, out written once
    return out


def favor_bwd(q, k, v, B, H, L, dh, ld, w, kind, ratio, eps, out, dout, dq, dk, dv, ld_d, workspace):
    _req_cuda(q, k, v, w, out, dout, dq, dk, dv, workspace)
    if not (out.is_contiguous() and dout.is_contiguous()) or dout.shape != out.shape:
        raise ValueError('favor_bwd: out / dout must be contiguous [B*L, H*dh]')
    e0 = _t0()
    st = lib().vitk_favor_bwd(_dt(q), kind, B, H, L, dh, w.shape[0], q.data_ptr(), k.data_ptr(), v.data_ptr(), ld,
                              w.data_ptr(), float(ratio), float(eps), out.data_ptr(), dout.data_ptr(), H * dh,
                              dq.data_ptr(), dk.data_ptr(), dv.data_ptr(), ld_d, workspace.data_ptr(), _stream())
    check(st, 'vitk_favor_bwd')
    _t1(e0, 'favor_bwd', 7.0 * B * L * H * dh * q.element_size(), 'byte')     # q, k, v, dout read, dq, dk, dv written


def favor_feature_pitch(m):
    return lib().vitk_favor_feature_pitch(int(m))


def favor_pos_fwd(x, H, dh, w, kind, is_query, ratio, eps):
    """Position-feature table of x [L, >= H*dh] (pos_query_proj / pos_key_proj of pos_embed):
    (feat fp32 [L, H, mp], state) -- state keeps the key max / argmax for favor_pos_bwd."""
    _req_cuda(x, w)
    if w.dtype != x.dtype or not w.is_contiguous() or w.shape[1] != dh or x.dim() != 2 or x.stride(1) != 1:
        raise ValueError('favor_pos: x must be 2-D rows, w contiguous [m, dh], both in the compute dtype')
    L, m = x.shape[0], w.shape[0]
    feat = torch.empty(L, H, favor_feature_pitch(m), device=x.device, dtype=torch.float32)
    state = torch.empty(lib().vitk_favor_pos_state_bytes(H, L), device=x.device, dtype=torch.uint8)
    st = lib().vitk_favor_pos_fwd(_dt(x), kind, int(is_query), H, L, dh, m, x.data_ptr(), x.stride(0), w.data_ptr(),
                                  float(ratio), float(eps), feat.data_ptr(), state.data_ptr(), _stream())
    check(st, 'vitk_favor_pos_fwd')
    return feat, state


def favor_pos_bwd(x, H, dh, w, kind, is_query, ratio, eps, dfeat, state):
    """dx [L, H*dh] from dfeat fp32 [nsum, L, H, mp] (summed over nsum in the kernel)."""
    _req_cuda(x, w, dfeat, state)
    L, m = x.shape[0], w.shape[0]
    if dfeat.dtype != torch.float32 or not dfeat.is_contiguous() or \
            tuple(dfeat.shape[1:]) != (L, H, favor_feature_pitch(m)):
        raise ValueError('favor_pos_bwd: dfeat must be contiguous fp32 [nsum, L, H, mp]')
    dx = torch.empty(L, H * dh, device=x.device, dtype=x.dtype)
    st = lib().vitk_favor_pos_bwd(_dt(x), kind, int(is_query), dfeat.shape[0], H, L, dh, m, x.data_ptr(), x.stride(0),
                                  w.data_ptr(), float(ratio), float(eps), dfeat.data_ptr(), dx.data_ptr(), H * dh,
                                  state.data_ptr(), _stream())
    check(st, 'vitk_favor_pos_bwd')
    return dx


def _check_pos_tables(L, H, m, *ts):
    for t in ts:
        if t.dtype != torch.float32 or not t.is_contiguous() or tuple(t.shape[-3:]) != (L, H, favor_feature_pitch(m)):
            raise ValueError('favor_pos_attn: position tables / gradients must be contiguous fp32 [.., L, H, mp]')


def favor_pos_attn_fwd(q, k, v, B, H, L, dh, ld, w, kind, ratio, eps, pq, pk, workspace):
    """favor_fwd with position features: q' + pq, k' + pk (fp32 [L, H, mp] from favor_pos_fwd)."""
    _req_cuda(q, k, v, w, pq, pk, workspace)
    if w.dtype != q.dtype or not w.is_contiguous() or w.shape[1] != dh:
        raise ValueError('favor: w must be contiguous [m, dh] in the compute dtype')
    _check_pos_tables(L, H, w.shape[0], pq, pk)
    out = torch.empty(B * L, H * dh, device=q.device, dtype=q.dtype)
    e0 = _t0()
    st = lib().vitk_favor_pos_attn_fwd(_dt(q), kind, B, H, L, dh, w.shape[0], q.data_ptr(), k.data_ptr(), v.data_ptr(),
                                       ld, w.data_ptr(), float(ratio), float(eps), pq.data_ptr(), pk.data_ptr(),
                                       out.data_ptr(), H * dh, workspace.data_ptr(), _stream())
    check(st, 'vitk_favor_pos_attn_fwd')
    _t1(e0, 'favor_fwd', 4.0 * B * L * H * dh * q.element_size(), 'byte')
    return out


def favor_pos_attn_bwd(q, k, v, B, H, L, dh, ld, w, kind, ratio, eps, pq, pk, out, dout, dq, dk, dv, ld_d, workspace):
    """Returns (dnq, dnk): fp32 [B, L, H, mp] gradients of the summed features, per batch element."""
    _req_cuda(q, k, v, w, pq, pk, out, dout, dq, dk, dv, workspace)
    if not (out.is_contiguous() and dout.is_contiguous()) or dout.shape != out.shape:
        raise ValueError('favor_bwd: out / dout must be contiguous [B*L, H*dh]')
    _check_pos_tables(L, H, w.shape[0], pq, pk)
    dnq = torch.empty(B, L, H, pq.shape[-1], device=q.device, dtype=torch.float32)
    dnk = torch.empty_like(dnq)
    e0 = _t0()
    st = lib().vitk_favor_pos_attn_bwd(_dt(q), kind, B, H, L, dh, w.shape[0], q.data_ptr(), k.data_ptr(), v.data_ptr(),
                                       ld, w.data_ptr(), float(ratio), float(eps), pq.data_ptr(), pk.data_ptr(),
                                       out.data_ptr(), dout.data_ptr(), H * dh, dq.data_ptr(), dk.data_ptr(),
                                       dv.data_ptr(), ld_d, dnq.data_ptr(), dnk.data_ptr(), workspace.data_ptr(),
                                       _stream())
    check(st, 'vitk_favor_pos_attn_bwd')
    # q, k, v, dout read, dq, dk, dv written, plus the two fp32 feature-gradient tables written
    _t1(e0, 'favor_bwd', 7.0 * B * L * H * dh * q.element_size() + 2.0 * dnq.numel() * 4, 'byte')
    return dnq, dnk


def ce_smooth_fwd(logits, target, smoothing):
    """Mean label-smoothing cross entropy of fp32 logits [B, C] vs int64 target [B]: (loss 0-dim, lse [B])."""
    _req_cuda(logits, target)
    if logits.dim() != 2 or logits.dtype != torch.float32 or logits.stride(1) != 1 or logits.numel() == 0:
        raise ValueError('ce_smooth: logits must be non-empty fp32 [B, C] rows')
    if target.dtype != torch.int64 or target.shape != (logits.shape[0],) or not target.is_contiguous():
        raise ValueError('ce_smooth: target must be contiguous int64 [B]')
    B, C = logits.shape
    buf = torch.empty(2 * B + 1, device=logits.device, dtype=torch.float32)
    st = lib().vitk_ce_smooth_fwd(B, C, logits.data_ptr(), logits.stride(0), target.data_ptr(), float(smoothing),
                                  buf.data_ptr(), buf[B:].data_ptr(), buf[2 * B:].data_ptr(), _stream())
    check(st, 'vitk_ce_smooth_fwd')
    return buf[2 * B], buf[B:2 * B]


def ce_smooth_bwd(logits, target, smoothing, lse, gout):
    _req_cuda(logits, target, lse, gout)
    if gout.dtype != torch.float32 or gout.numel() != 1:
        raise ValueError('ce_smooth_bwd: the upstream gradient must be one fp32 value')
    B, C = logits.shape
    d = torch.empty(B, C, device=logits.device, dtype=torch.float32)
    st = lib().vitk_ce_smooth_bwd(B, C, logits.data_ptr(), logits.stride(0), target.data_ptr(), float(smoothing),
                                  lse.data_ptr(), gout.data_ptr(), d.data_ptr(), C, _stream())
    check(st, 'vitk_ce_smooth_bwd')
    return d
