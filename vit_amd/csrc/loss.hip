// Label-smoothing cross entropy over fp32 logits, forward and backward.
//
// Reference: criterion = LabelSmoothingCrossEntropy(smoothing=args.smoothing) (train.py:273; timm 0.4.5
// timm/loss/cross_entropy.py), applied at train.py:541 and reduced to a mean:
//     logp = log_softmax(x);  loss_row = (1 - s) * (-logp[target]) + s * (-mean_c logp);  loss = mean_rows
// HBM-bound and tiny (B x C floats read twice): one wave per row, 4 rows per workgroup; the mean over rows is a
// second one-wave kernel so that the summation order is fixed (no atomics).
#include <mutex>

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

// ------------------------------------------------------------------------------------------------------------
// The forward for callers that cannot allocate (code under HIP graph capture, tight serving loops, harnesses that
// forbid hipMalloc): same kernels as vitk_ce_smooth_fwd, but the per-row losses -- partial results that only the
// one-wave mean kernel reads -- live in a file-scope device array instead of a caller / library buffer.  The caller
// passes inputs and outputs only.
//   * limit: B <= VITK_CE_STATIC_MAX_ROWS (65536 rows, 256 KiB of device memory per GPU the library runs on,
//     held until the process ends);
//   * every call on one GPU shares that GPU's array: calls that overlap in time on one GPU (two streams) race.
//     Nothing in it has to start at zero: a call writes rows [0, B) before it reads them, rows past B are
//     never read, so a smaller call after a larger one is unaffected by what the larger one left behind.
// The array's NAME is a host-side stand-in, not a device address: its address on the current device is looked up
// once per device with hipGetSymbolAddress and only that pointer is ever handed to a kernel.
constexpr int VITK_CE_STATIC_MAX_ROWS = 65536;
static __device__ __align__(256) float scratch[VITK_CE_STATIC_MAX_ROWS];

constexpr int VITK_MAX_DEVICES = 64;
static float* g_scratch_ptr[VITK_MAX_DEVICES];     // per device: each GPU holds its own copy of the array
static std::mutex g_scratch_mutex;

// Address of this device's copy of `scratch` (looked up on the first call on that device, with it current).
static int ce_static_scratch(float** out) {
    int dev = -1;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= VITK_MAX_DEVICES) {
        vitk_set_error("vitk_ce_smooth_fwd_static: no current device (or more than %d devices)", VITK_MAX_DEVICES);
        return VITK_ERR_ARG;
    }
    std::lock_guard<std::mutex> lock(g_scratch_mutex);
    if (g_scratch_ptr[dev] == nullptr) {
        float* ptr = nullptr;
        const hipError_t e = hipGetSymbolAddress(reinterpret_cast<void**>(&ptr), HIP_SYMBOL(scratch));
        if (e != hipSuccess || ptr == nullptr) {
            vitk_set_error("vitk_ce_smooth_fwd_static: hipGetSymbolAddress failed on device %d: %s", dev, hipGetErrorString(e));
            return VITK_ERR_LAUNCH;
        }
        g_scratch_ptr[dev] = ptr;
    }
    *out = g_scratch_ptr[dev];
    return VITK_OK;
}

extern "C" int vitk_ce_smooth_fwd_static(int B, int C, const float* logits, long ld, const long long* target,
                                         float smoothing, float* lse, float* loss, hipStream_t stream) {
    VITK_CHECK_ARG(B > 0 && C > 0, "vitk_ce_smooth_fwd_static: empty problem");
    VITK_CHECK_ARG(B <= VITK_CE_STATIC_MAX_ROWS,
                   "vitk_ce_smooth_fwd_static: B = %d rows exceeds the static scratch (%d rows); use vitk_ce_smooth_fwd", B,
                   VITK_CE_STATIC_MAX_ROWS);
    VITK_CHECK_ARG(logits && target && lse && loss && ld >= C, "vitk_ce_smooth_fwd_static: bad pointer / pitch");
    VITK_CHECK_ARG(smoothing >= 0.f && smoothing < 1.f, "vitk_ce_smooth_fwd_static: smoothing must be in [0, 1)");
    float* rows = nullptr;
    const int st = ce_static_scratch(&rows);
    if (st != VITK_OK) return st;
    hipLaunchKernelGGL(ce_fwd_kernel, dim3((B + 3) / 4), dim3(256), 0, stream, B, C, logits, ld, target, smoothing, rows, lse);
    hipLaunchKernelGGL(ce_mean_kernel, dim3(1), dim3(64), 0, stream, B, rows, loss);
    VITK_CHECK_LAUNCH("vitk_ce_smooth_fwd_static");
    return VITK_OK;
}

extern "C" int vitk_ce_static_max_rows(void) { return VITK_CE_STATIC_MAX_ROWS; }
