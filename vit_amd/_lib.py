"""ctypes binding of libvitk.so -- the C-ABI boundary declared in include/vitk.h.

There is NO fallback: if the library is missing or a call fails, the op raises.
"""
import ctypes
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('VITK_LIB') or os.path.join(HERE, 'lib', 'libvitk.so')   # VITK_LIB: an instrumented build (tools/)

ABI_VERSION = 3        # include/vitk.h VITK_ABI_VERSION: the argument lists below are those of exactly this version
F32, BF16 = 0, 1
ACT_NONE, ACT_GELU, ACT_RELU, ACT_DGELU, ACT_DRELU = 0, 1, 2, 3, 4

_lib = None

c_int, c_long, c_float, c_double, c_vp = ctypes.c_int, ctypes.c_long, ctypes.c_float, ctypes.c_double, ctypes.c_void_p
c_u64 = ctypes.c_ulonglong

_SIGNATURES = {
    'vitk_version': ([], c_int),
    'vitk_last_error': ([], ctypes.c_char_p),
    'vitk_set_reserved_cus': ([c_int], c_int),
    'vitk_get_reserved_cus': ([], c_int),
    'vitk_gemm': ([c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_float,
                   c_vp, c_long, c_vp, c_long, c_vp, c_long,
                   c_vp, c_int, c_vp, c_vp, c_long,
                   c_vp, c_long, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_vp], c_int),
    'vitk_layernorm_fwd': ([c_int, c_int, c_int, c_vp, c_long, c_vp, c_vp, c_float, c_vp, c_long,
                            c_vp, c_vp, c_vp], c_int),
    'vitk_layernorm_bwd': ([c_int, c_int, c_int, c_vp, c_long, c_vp, c_long, c_vp, c_vp, c_vp,
                            c_vp, c_long, c_vp, c_long, c_vp, c_vp, c_vp, c_vp], c_int),
    'vitk_adamw_dyn': ([c_int, c_int, c_vp, c_vp, c_vp, c_vp, c_double, c_double, c_double, c_vp,
                        c_vp, c_long, c_vp, c_long, c_vp], c_int),
    'vitk_adamw_multi': ([c_int, c_vp, c_long, c_double, c_double, c_double, c_vp, c_vp], c_int),
    'vitk_adamw_desc_bytes': ([], c_int),
    'vitk_upload_args': ([c_vp, c_vp, ctypes.c_size_t, c_vp], c_int),
    'vitk_set_f32': ([c_vp, c_int, c_float, c_float, c_float, c_float, c_vp], c_int),
    'vitk_colsum': ([c_int, c_int, c_int, c_vp, c_long, c_vp, c_vp], c_int),
    'vitk_cast': ([c_int, c_long, c_vp, c_vp, c_vp], c_int),
    'vitk_adamw': ([c_int, c_int, c_vp, c_vp, c_vp, c_vp, c_double, c_double, c_double, c_double, c_double, c_int,
                    c_vp, c_long, c_vp, c_long, c_vp], c_int),
    'vitk_patchify': ([c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_vp, c_vp, c_vp], c_int),
    'vitk_patchify_u8': ([c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_vp, c_vp, c_vp, c_vp, c_vp], c_int),
    'vitk_unpatchify': ([c_int, c_int, c_int, c_int, c_int, c_int, c_vp, c_vp, c_vp], c_int),
    'vitk_cls_rows': ([c_int, c_int, c_int, c_int, c_vp, c_vp, c_vp, c_vp], c_int),
    'vitk_token_grads': ([c_int, c_int, c_int, c_int, c_vp, c_vp, c_vp, c_vp], c_int),
    'vitk_attn_fwd': ([c_int, c_int, c_int, c_int, c_int,
                       c_vp, c_vp, c_vp, c_long, c_vp, c_vp, c_long,
                       c_float, c_int, c_vp, c_long, c_vp, c_float, c_u64, c_u64, c_vp], c_int),
    'vitk_attn_bwd': ([c_int, c_int, c_int, c_int, c_int,
                       c_vp, c_vp, c_vp, c_long, c_vp, c_vp, c_long,
                       c_float, c_int, c_vp, c_vp, c_long, c_vp,
                       c_vp, c_vp, c_vp, c_long, c_vp, c_vp, c_long, c_vp, c_float, c_u64, c_u64, c_vp, c_vp], c_int),
    'vitk_dropout': ([c_int, c_int, c_int, c_vp, c_long, c_vp, c_long, c_vp, c_long, c_float, c_u64, c_u64, c_vp], c_int),
    'vitk_drop_path': ([c_int, c_int, c_int, c_int, c_vp, c_long, c_vp, c_long, c_vp, c_long, c_float, c_u64, c_u64, c_vp],
                       c_int),
    'vitk_dropout_mask': ([c_long, c_float, c_u64, c_u64, c_vp, c_vp], c_int),
    'vitk_attn_bwd_workspace_bytes': ([c_int, c_int, c_int, c_int, c_int], c_long),
    'vitk_favor_workspace_bytes': ([c_int, c_int, c_int, c_int, c_int], c_long),
    'vitk_favor_fwd': ([c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_vp, c_vp, c_vp, c_long, c_vp,
                        c_float, c_float, c_vp, c_long, c_vp, c_vp], c_int),
    'vitk_favor_bwd': ([c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_vp, c_vp, c_vp, c_long, c_vp,
                        c_float, c_float, c_vp, c_vp, c_long, c_vp, c_vp, c_vp, c_long, c_vp, c_vp], c_int),
    'vitk_ce_smooth_fwd': ([c_int, c_int, c_vp, c_long, c_vp, c_float, c_vp, c_vp, c_vp, c_vp], c_int),
    'vitk_ce_smooth_fwd_static': ([c_int, c_int, c_vp, c_long, c_vp, c_float, c_vp, c_vp, c_vp], c_int),
    'vitk_ce_static_max_rows': ([], c_int),
    'vitk_ce_smooth_bwd': ([c_int, c_int, c_vp, c_long, c_vp, c_float, c_vp, c_vp, c_vp, c_long, c_vp], c_int),
    'vitk_favor_feature_pitch': ([c_int], c_int),
    'vitk_favor_pos_state_bytes': ([c_int, c_int], c_long),
    'vitk_favor_pos_fwd': ([c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_vp, c_long, c_vp, c_float, c_float,
                            c_vp, c_vp, c_vp], c_int),
    'vitk_favor_pos_bwd': ([c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_vp, c_long, c_vp, c_float,
                            c_float, c_vp, c_vp, c_long, c_vp, c_vp], c_int),
    'vitk_favor_pos_attn_fwd': ([c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_vp, c_vp, c_vp, c_long, c_vp,
                                 c_float, c_float, c_vp, c_vp, c_vp, c_long, c_vp, c_vp], c_int),
    'vitk_favor_pos_attn_bwd': ([c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_vp, c_vp, c_vp, c_long, c_vp,
                                 c_float, c_float, c_vp, c_vp, c_vp, c_vp, c_long, c_vp, c_vp, c_vp, c_long, c_vp,
                                 c_vp, c_vp, c_vp], c_int),
}


class VitkError(RuntimeError):
    pass


def lib():
    """Load libvitk.so (once).  Raises if it has not been built: there is no other path."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise VitkError(f'{LIB_PATH} is missing: run `python -m vit_amd._build` (hipcc, gfx950) first')
        L = ctypes.CDLL(LIB_PATH)
        for name, sig in _SIGNATURES.items():
            if sig is None:
                continue
            if not hasattr(L, name):
                continue
            fn = getattr(L, name)
            fn.argtypes, fn.restype = sig
        have = L.vitk_version() if hasattr(L, 'vitk_version') else None
        if have != ABI_VERSION:
            raise VitkError(f'{LIB_PATH} has ABI version {have}, this binding is written for {ABI_VERSION}: rebuild it '
                            '(`python -m vit_amd._build`); a stale library would misalign arguments silently')
        _lib = L
    return _lib


def check(status, what):
    if status != 0:
        msg = lib().vitk_last_error()
        raise VitkError(f'{what} failed (status {status}): {msg.decode() if msg else "?"}')
