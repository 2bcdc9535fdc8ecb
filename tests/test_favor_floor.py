"""The error floor under the bf16 FAVOR+ tolerance (tests/test_gpu_performer.py: relative L2 <= 3e-2, max-norm guard 0.2).

FAVOR+ (timm_classes/timm_vision_performer.py:69-91 of the reference) exponentiates projections and, in its ReLU
variant, gates them: rounding q, k, v and the random matrix w to bf16 and running the fp32 oracle on the rounded values
already moves outputs and gradients by about a percent -- before any kernel has done arithmetic.  No bf16-input
implementation can sit under that, so the usual 2e-2 max-norm bar is not meaningful here; this test pins the floor
(CPU only, the oracle against itself) so that the relaxed bar rests on an assertion, not on a tool's printout."""
import pytest
import torch

from oracle import ref_cpu as R


def errs(a, b):
    a, b = a.double().reshape(-1), b.double().reshape(-1)
    return ((a - b).abs().max() / b.abs().max()).item(), ((a - b).norm() / b.norm()).item()


@pytest.mark.parametrize('kernel,m', [('softmax', 64), ('relu', 64), ('softmax', 16)])
def test_bf16_input_rounding_floor_of_the_oracle(kernel, m):
    B, H, dh, L = 2, 4, 64, 145
    D = H * dh
    torch.manual_seed(5)
    qkv = torch.randn(B * L, 3 * D) * 0.6
    w0 = R.gaussian_orthogonal_random_matrix(m, dh, generator=torch.Generator().manual_seed(1))
    gout = torch.randn(B * L, D)
    ratio = m ** -0.25

    def oracle(x, w):
        qr = x.clone().requires_grad_(True)
        q, k, v = (qr[:, i * D:(i + 1) * D].view(B, L, H, dh) for i in range(3))
        if kernel == 'softmax':
            kp, qp = R.favor_softmax_features(k, w, False, ratio), R.favor_softmax_features(q, w, True, ratio)
        else:
            kp, qp = R.favor_relu_features(k, w, ratio), R.favor_relu_features(q, w, ratio)
        y = R.favor_linear_attention(qp, kp, v).reshape(B * L, D)
        (y * gout).sum().backward()
        return y.detach(), qr.grad

    y, dx = oracle(qkv, w0)
    y_r, dx_r = oracle(qkv.bfloat16().float(), w0.bfloat16().float())
    (y_max, y_l2), (dx_max, dx_l2) = errs(y_r, y), errs(dx_r, dx)
    # measured (seeds 5-7): softmax y max 1.4-2.4e-2 / L2 7-9e-3, dx max 1.0-2.0e-2 / L2 1.1-1.3e-2;
    #                       relu    y max 2.5e-3    / L2 2e-3,   dx max 1.0-1.2e-1 / L2 1.5-2.2e-2
    assert dx_l2 > 8e-3, dx_l2                  # rounding alone costs about a percent of the gradient in L2 ...
    assert dx_l2 < 3e-2 and y_l2 < 3e-2         # ... and stays under the 3e-2 L2 bar the bf16 kernels are held to
    if kernel == 'relu':
        assert dx_max > 5e-2, dx_max            # single gated elements move by ~10 %: why the max-norm guard is 0.2
    else:
        assert max(y_max, dx_max) > 1e-2        # a 2e-2 max-norm bar would leave the kernels no room at all
