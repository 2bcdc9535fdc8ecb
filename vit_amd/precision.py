"""Compute-dtype selection for the MI355X modules.

Parameters always stay fp32 (the reference is fp32-only).  ``compute_dtype`` picks what the
kernels run in: ``torch.float32`` (exact f32 MFMA, parity to 1e-5) or ``torch.bfloat16`` (bf16
MFMA with fp32 accumulation; the speed path).  Set it per module tree with ``set_compute_dtype``.
"""
import torch


class ComputeDtypeMixin:
    compute_dtype = torch.float32


def set_compute_dtype(module, dtype):
    """Set the kernel compute dtype on every vit_amd module under ``module``; returns ``module``."""
    if dtype not in (torch.float32, torch.bfloat16):
        raise TypeError('compute dtype must be torch.float32 or torch.bfloat16')
    for m in module.modules():
        if isinstance(m, ComputeDtypeMixin):
            m.compute_dtype = dtype
    return module
