"""The allocation-free entry point (vitk_ce_smooth_fwd_static) against vitk_ce_smooth_fwd, both through the C-ABI:
same kernels, the per-row partial losses in the library's static device array instead of a caller buffer.
Two calls in a row, the second with a smaller input (the array keeps the first call's contents)."""
import ctypes

import pytest
import torch

pytestmark = pytest.mark.gpu


def _both(L, logits, target, smoothing):
    B, C = logits.shape
    s = torch.cuda.current_stream().cuda_stream
    ref = torch.full((2 * B + 1,), float('nan'), device='cuda')
    st = L.vitk_ce_smooth_fwd(B, C, logits.data_ptr(), logits.stride(0), target.data_ptr(), smoothing,
                              ref.data_ptr(), ref[B:].data_ptr(), ref[2 * B:].data_ptr(), s)
    assert st == 0, L.vitk_last_error()
    out = torch.full((B + 1,), float('nan'), device='cuda')
    st = L.vitk_ce_smooth_fwd_static(B, C, logits.data_ptr(), logits.stride(0), target.data_ptr(), smoothing,
                                     out.data_ptr(), out[B:].data_ptr(), s)
    assert st == 0, L.vitk_last_error()
    torch.cuda.synchronize()
    return ref[B:2 * B], ref[2 * B], out[:B], out[B]


def test_static_scratch_entry_matches_workspace_entry():
    from vit_amd._lib import lib
    L = lib()
    torch.manual_seed(0)
    for B, C in ((512, 1000), (37, 10)):          # the second call is smaller: rows 37.. of the array hold stale losses
        logits = torch.randn(B, C, device='cuda') * 3
        target = torch.randint(0, C, (B,), device='cuda')
        lse_ref, loss_ref, lse, loss = _both(L, logits, target, 0.1)
        assert torch.equal(lse, lse_ref)                         # same kernels, same order of operations: bit-identical
        assert torch.equal(loss, loss_ref)
        want = torch.nn.functional.cross_entropy(logits.double(), target, label_smoothing=0.1)
        assert abs(loss.item() - want.item()) < 1e-5 * abs(want.item())
    # sizes beyond the static array are refused with a message, nothing is launched
    big = L.vitk_ce_static_max_rows() + 1
    x = torch.zeros(1, 4, device='cuda')
    t = torch.zeros(1, dtype=torch.int64, device='cuda')
    o = torch.zeros(2, device='cuda')
    st = L.vitk_ce_smooth_fwd_static(big, 4, x.data_ptr(), 4, t.data_ptr(), 0.1, o.data_ptr(), o[1:].data_ptr(),
                                     torch.cuda.current_stream().cuda_stream)
    assert st != 0 and b'exceeds the static scratch' in L.vitk_last_error()
