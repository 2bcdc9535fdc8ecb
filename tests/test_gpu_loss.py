"""Label-smoothing cross entropy (csrc/loss.hip) through the C-ABI against the oracle's restatement of
timm 0.4.5 LabelSmoothingCrossEntropy (oracle/ref_cpu.py::label_smoothing_ce; pinned by the 'loss' entry of
tests/golden/config1_step.npz, which the reference produced).  fp32: 1e-5 relative."""
import os

import numpy as np
import pytest
import torch

import golden_inputs as gi
from oracle import ref_cpu as R

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


def rel(a, b):
    a, b = a.double().reshape(-1), b.double().reshape(-1)
    return ((a - b).abs().max() / (b.abs().max() + 1e-30)).item()


@pytest.mark.parametrize('B,C', [(8, 10), (128, 1000), (5, 1000), (3, 7), (1, 64), (130, 65)])
@pytest.mark.parametrize('smoothing', [0.1, 0.0, 0.35])
def test_ce_vs_oracle(B, C, smoothing):
    from vit_amd import functional as VF
    logits = gi.randn(91, f'logits_{B}_{C}', B, C, scale=3.0)
    tgt = torch.randint(0, C, (B,), generator=torch.Generator().manual_seed(B * 1000 + C))
    xr = logits.clone().requires_grad_(True)
    ref = R.label_smoothing_ce(xr, tgt, smoothing)
    (ref * 1.7).backward()
    xg = logits.clone().cuda().requires_grad_(True)
    out = VF.label_smoothing_ce(xg, tgt.cuda(), smoothing)
    (out * 1.7).backward()
    assert out.shape == ref.shape
    assert rel(out.detach().cpu(), ref.detach()) < 1e-5
    assert rel(xg.grad.cpu(), xr.grad) < 1e-5


def test_ce_reference_golden_loss():
    """The loss the reference computed for the config-1 step (train.py:541) from the logits it produced."""
    from vit_amd.train import label_smoothing_ce
    g = np.load(os.path.join(G, 'config1_step.npz'))
    logits = torch.from_numpy(g['logits']).cuda()
    tgt = torch.from_numpy(g['target']).cuda()
    loss = label_smoothing_ce(logits, tgt, 0.1)
    assert abs(loss.item() - float(g['loss'])) < 1e-5 * abs(float(g['loss']))


def test_ce_strided_rows_dtypes_and_rejects():
    from vit_amd import functional as VF, ops
    from vit_amd._lib import VitkError
    big = torch.randn(6, 40, device='cuda')
    view = big[:, :17]                                   # row pitch 40
    tgt = torch.randint(0, 17, (6,), device='cuda')
    a = VF.label_smoothing_ce(view, tgt, 0.1)
    b = R.label_smoothing_ce(view.cpu(), tgt.cpu(), 0.1)
    assert rel(a.cpu(), b) < 1e-5
    c = VF.label_smoothing_ce(view.to(torch.bfloat16), tgt.int(), 0.1)          # cast on the host side
    assert rel(c.cpu(), R.label_smoothing_ce(view.to(torch.bfloat16).float().cpu(), tgt.cpu(), 0.1)) < 1e-5
    bad = VF.label_smoothing_ce(view, torch.full((6,), 17, device='cuda'), 0.1)   # class out of range: NaN, no fault
    assert torch.isnan(bad).item()
    with pytest.raises(ValueError):
        ops.ce_smooth_fwd(view, tgt[:5], 0.1)
    with pytest.raises(ValueError):
        ops.ce_smooth_fwd(torch.empty(0, 5, device='cuda'), torch.empty(0, dtype=torch.long, device='cuda'), 0.1)
    with pytest.raises(VitkError):
        ops.ce_smooth_fwd(view, tgt, 1.5)
    with pytest.raises(VitkError):
        ops.ce_smooth_fwd(torch.randn(2, 3), torch.zeros(2, dtype=torch.long), 0.1)   # CPU tensors: no fallback
