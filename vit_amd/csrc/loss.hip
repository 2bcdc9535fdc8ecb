// Label-smoothing cross entropy over fp32 logits, forward and backward.
//
// Reference: criterion = LabelSmoothingCrossEntropy(smoothing=args.smoothing) (train.py:273; timm 0.4.5
// timm/loss/cross_entropy.py), applied at train.py:541 and reduced to a mean:
//     logp = log_softmax(x);  loss_row = (1 - s) * (-logp[target]) + s * (-mean_c logp);  loss = mean_rows
// HBM-bound and tiny (B x C floats read twice): one wave per row, 4 rows per workgroup; the mean over rows is a
// second one-wave kernel so that the summation order is fixed (no atomics).
#include "common.h"

namespace vitk {

__global__ __launch_bounds__(256) void ce_fwd_kernel(int B, int C, const float* __restrict__ logits, long ld,
                                                     const long long* __restrict__ target, float smoothing,
                                                     float* __restrict__ loss_rows, float* __restrict__ lse_out) {
    const int lane = threadIdx.x & 63, row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= B) return;
    const float* x = logits + (long)row * ld;
    float mx = -INFINITY, sx = 0.f;
    for (int c = lane; c < C; c += 64) {
        const float v = x[c];
        mx = fmaxf(mx, v);
        sx += v;
    }
    mx = wave_max(mx);
    sx = wave_sum(sx);
    float se = 0.f;
    for (int c = lane; c < C; c += 64) se += expf(x[c] - mx);
    se = wave_sum(se);
    const float lse = mx + logf(se);
    if (lane == 0) {
        const long long t = target[row];
        const float xt = (t >= 0 && t < C) ? x[t] : NAN;        // an out-of-range class poisons the loss (torch raises)
        loss_rows[row] = (1.f - smoothing) * (lse - xt) + smoothing * (lse - sx / (float)C);
        lse_out[row] = lse;
    }
}

__global__ __launch_bounds__(64) void ce_mean_kernel(int B, const float* __restrict__ loss_rows, float* __restrict__ out) {
    float s = 0.f;
    for (int i = threadIdx.x; i < B; i += 64) s += loss_rows[i];
    s = wave_sum(s);
    if (threadIdx.x == 0) out[0] = s / (float)B;
}

// dlogits[r][c] = (softmax[r][c] - (1 - s) [c == target_r] - s / C) * gout / B
__global__ __launch_bounds__(256) void ce_bwd_kernel(int B, int C, const float* __restrict__ logits, long ld,
                                                     const long long* __restrict__ target, float smoothing,
                                                     const float* __restrict__ lse, const float* __restrict__ gout,
                                                     float* __restrict__ dlogits, long ld_d) {
    const int lane = threadIdx.x & 63, row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= B) return;
    const float* x = logits + (long)row * ld;
    float* d = dlogits + (long)row * ld_d;
    const float scale = gout[0] / (float)B, l = lse[row], off = smoothing / (float)C;
    const long long t = target[row];
    for (int c = lane; c < C; c += 64) {
        const float p = expf(x[c] - l);
        d[c] = (p - (c == t ? 1.f - smoothing : 0.f) - off) * scale;
    }
}

}  // namespace vitk

using namespace vitk;

// loss[0] = mean over rows of the label-smoothing cross entropy of logits [B, C] (fp32, row pitch ld) against
// target [B] (int64).  loss_rows [B] and lse [B] (fp32) are outputs as well: lse is what the backward needs.
extern "C" int vitk_ce_smooth_fwd(int B, int C, const float* logits, long ld, const long long* target, float smoothing,
                                  float* loss_rows, float* lse, float* loss, hipStream_t stream) {
    VITK_CHECK_ARG(B > 0 && C > 0, "vitk_ce_smooth_fwd: empty problem");
    VITK_CHECK_ARG(logits && target && loss_rows && lse && loss && ld >= C, "vitk_ce_smooth_fwd: bad pointer / pitch");
    VITK_CHECK_ARG(smoothing >= 0.f && smoothing < 1.f, "vitk_ce_smooth_fwd: smoothing must be in [0, 1)");
    hipLaunchKernelGGL(ce_fwd_kernel, dim3((B + 3) / 4), dim3(256), 0, stream, B, C, logits, ld, target, smoothing,
                       loss_rows, lse);
    hipLaunchKernelGGL(ce_mean_kernel, dim3(1), dim3(64), 0, stream, B, loss_rows, loss);
    VITK_CHECK_LAUNCH("vitk_ce_smooth_fwd");
    return VITK_OK;
}

// dlogits [B, C] (fp32, row pitch ld_d) = d (gout * loss) / d logits; gout: one fp32 value in DEVICE memory (the
// upstream gradient of the scalar loss), lse from the forward.
extern "C" int vitk_ce_smooth_bwd(int B, int C, const float* logits, long ld, const long long* target, float smoothing,
                                  const float* lse, const float* gout, float* dlogits, long ld_d, hipStream_t stream) {
    VITK_CHECK_ARG(B > 0 && C > 0, "vitk_ce_smooth_bwd: empty problem");
    VITK_CHECK_ARG(logits && target && lse && gout && dlogits && ld >= C && ld_d >= C,
                   "vitk_ce_smooth_bwd: bad pointer / pitch");
    hipLaunchKernelGGL(ce_bwd_kernel, dim3((B + 3) / 4), dim3(256), 0, stream, B, C, logits, ld, target, smoothing, lse,
                       gout, dlogits, ld_d);
    VITK_CHECK_LAUNCH("vitk_ce_smooth_bwd");
    return VITK_OK;
}
