// AdamW step of one [rows, cols] fp32 parameter, fused with the refresh of its compute-dtype copies.
//
// Reference: torch.optim.AdamW(model.parameters(), lr=1e-4) built at train.py:275 and stepped at
// train.py:549 (torch defaults: betas 0.9 / 0.999, eps 1e-8, weight_decay 1e-2, no amsgrad):
//     p *= 1 - lr * wd;  m = b1 m + (1 - b1) g;  v = b2 v + (1 - b2) g^2
//     p -= (lr / (1 - b1^t)) * m / (sqrt(v) / sqrt(1 - b2^t) + eps)
// HBM-bound: 16 B read + 12 B written per element, plus 2 + 2 B for the two bf16 copies the GEMMs
// read: `shadow` (same [out, in] layout: forward and dW) and `shadow_t` (transposed, [in, out]: the
// dX GEMMs then read both operands K-contiguous).  A 64 x 64 tile per workgroup; the transposed copy
// goes through LDS so that both copies are written in 8/16-byte pieces.
#include "common.h"

namespace vitk {

constexpr int OPT_TILE = 64;
constexpr int OPT_LD = OPT_TILE + 8;   // LDS row stride (bf16): 144 B keeps 16-byte alignment, spreads banks

struct AdamWParams {
    float* p;
    const float* g;
    float *m, *v;
    __bf16* shadow;
    long ld_shadow;
    __bf16* shadow_t;
    long ld_shadow_t;
    int rows, cols;
    float decay;        // 1 - lr * wd
    float b1, b2;
    float omb1, omb2;   // 1 - b1, 1 - b2, rounded from double as torch does
    float step_size;    // lr / (1 - b1^t)
    float inv_bc2_sqrt; // 1 / sqrt(1 - b2^t)
    float eps;
    const float* dyn;   // optional device scalars {decay, step_size, inv_bc2_sqrt}: a captured HIP graph replays with new ones
};

// One 64 x 64 tile of one parameter (tile origin r0, c0); ``tile`` = the workgroup's LDS staging of the transposed copy.
__device__ __forceinline__ void adamw_tile(const AdamWParams& a, int r0, int c0, __bf16* tile) {
    const int t = threadIdx.x;
    const int cg = t & 15, rr = t >> 4;
    const int c = c0 + 4 * cg;
    const float decay = a.dyn ? a.dyn[0] : a.decay;
    const float step_size = a.dyn ? a.dyn[1] : a.step_size;
    const float inv_bc2_sqrt = a.dyn ? a.dyn[2] : a.inv_bc2_sqrt;
#pragma unroll
    for (int pass = 0; pass < 4; ++pass) {
        const int rl = rr + 16 * pass;
        const int r = r0 + rl;
        f32x4 pn = {0.f, 0.f, 0.f, 0.f};
        if (r < a.rows && c < a.cols) {
            const long off = (long)r * a.cols + c;
            const bool vec = (a.cols & 3) == 0;      // else (no copies, host-checked): element by element, ragged row end
            f32x4 g = {0.f, 0.f, 0.f, 0.f}, p = g, m = g, v = g;
            if (vec) {
                g = *reinterpret_cast<const f32x4*>(a.g + off);
                p = *reinterpret_cast<const f32x4*>(a.p + off);
                m = *reinterpret_cast<const f32x4*>(a.m + off);
                v = *reinterpret_cast<const f32x4*>(a.v + off);
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (c + e < a.cols) {
                        g[e] = a.g[off + e];
                        p[e] = a.p[off + e];
                        m[e] = a.m[off + e];
                        v[e] = a.v[off + e];
                    }
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                p[e] *= decay;
                m[e] += a.omb1 * (g[e] - m[e]);                       // torch: exp_avg.lerp_(grad, 1 - beta1)
                v[e] = a.b2 * v[e] + a.omb2 * g[e] * g[e];
                const float denom = sqrtf(v[e]) * inv_bc2_sqrt + a.eps;
                p[e] -= step_size * (m[e] / denom);
            }
            if (vec) {
                *reinterpret_cast<f32x4*>(a.p + off) = p;
                *reinterpret_cast<f32x4*>(a.m + off) = m;
                *reinterpret_cast<f32x4*>(a.v + off) = v;
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (c + e < a.cols) {
                        a.p[off + e] = p[e];
                        a.m[off + e] = m[e];
                        a.v[off + e] = v[e];
                    }
            }
            pn = p;
            if (a.shadow)
                *reinterpret_cast<bf16x4*>(a.shadow + (long)r * a.ld_shadow + c) =
                    bf16x4{(__bf16)p[0], (__bf16)p[1], (__bf16)p[2], (__bf16)p[3]};
        }
        if (a.shadow_t) {
#pragma unroll
            for (int e = 0; e < 4; ++e) tile[(4 * cg + e) * OPT_LD + rl] = (__bf16)pn[e];
        }
    }
    if (!a.shadow_t) return;     // (uniform over the workgroup)
    __syncthreads();
    // transposed copy: row (c0 + cl) of shadow_t gets rows r0 .. r0+63 of this tile
    const int rg = t & 7;
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
        const int cl = (t >> 3) + 32 * pass;
        const int cc = c0 + cl, r = r0 + 8 * rg;
        if (cc >= a.cols || r >= a.rows) continue;
        __bf16* dst = a.shadow_t + (long)cc * a.ld_shadow_t + r;
        const __bf16* src = tile + cl * OPT_LD + 8 * rg;
        if (r + 8 <= a.rows && (a.ld_shadow_t & 7) == 0) {
            *reinterpret_cast<bf16x8*>(dst) = *reinterpret_cast<const bf16x8*>(src);
        } else {
            for (int e = 0; e < 8 && r + e < a.rows; ++e) dst[e] = src[e];
        }
    }
}

__global__ __launch_bounds__(256) void adamw_kernel(const AdamWParams a) {
    __shared__ __attribute__((aligned(16))) __bf16 tile[OPT_TILE * OPT_LD];   // [col][row]
    adamw_tile(a, blockIdx.y * OPT_TILE, blockIdx.x * OPT_TILE, tile);
}

// Multi-tensor form: ONE launch updates every parameter of a group.  ``desc`` (device memory, written by the
// host once per set of tensors) lists the parameters; tile ids run through them in order (desc[i].tile0 = the
// first id of parameter i, desc[n].tile0 = the total).  Persistent workgroups walk the ids blockIdx.x,
// + gridDim.x, ...: 148 launches of the per-parameter form (2.6 ms, 4.1 TB/s for the 306 M parameters of the
// headline model: the small ones cannot fill the chip and every launch has its ramp) become one.
struct AdamWDesc {       // 72 bytes; the host writes it as 9 int64 words (little endian: low int first)
    float* p;
    const float* g;
    float* m;
    float* v;
    __bf16* shadow;
    __bf16* shadow_t;
    int ld_shadow, ld_shadow_t;
    int rows, cols;
    int tile0, tiles_x;
};
static_assert(sizeof(AdamWDesc) == 72, "AdamWDesc layout");

__global__ __launch_bounds__(256) void adamw_multi_kernel(const AdamWDesc* __restrict__ desc, int n, AdamWParams base) {
    __shared__ __attribute__((aligned(16))) __bf16 tile[OPT_TILE * OPT_LD];
    const int total = desc[n].tile0;
    for (int id = blockIdx.x; id < total; id += gridDim.x) {
        int lo = 0, hi = n;                       // last i with desc[i].tile0 <= id
        while (hi - lo > 1) {
            const int mid = (lo + hi) >> 1;
            if (desc[mid].tile0 <= id) lo = mid;
            else hi = mid;
        }
        const AdamWDesc d = desc[lo];
        AdamWParams a = base;
        a.p = d.p; a.g = d.g; a.m = d.m; a.v = d.v;
        a.shadow = d.shadow; a.ld_shadow = d.ld_shadow;
        a.shadow_t = d.shadow_t; a.ld_shadow_t = d.ld_shadow_t;
        a.rows = d.rows; a.cols = d.cols;
        const int local = id - d.tile0;
        const int ty = local / d.tiles_x, tx = local - ty * d.tiles_x;
        adamw_tile(a, ty * OPT_TILE, tx * OPT_TILE, tile);
        __syncthreads();                          // the LDS tile is reused by the next id
    }
}

__global__ void set_f32_kernel(float* dst, int n, float v0, float v1, float v2, float v3) {
    const float v[4] = {v0, v1, v2, v3};
    if ((int)threadIdx.x < n) dst[threadIdx.x] = v[threadIdx.x];
}

}  // namespace vitk

using namespace vitk;

static int adamw_impl(int rows, int cols, float* p, const float* g, float* m, float* v, double lr, double beta1,
                      double beta2, double eps, double weight_decay, int step, void* shadow, long ld_shadow,
                      void* shadow_t, long ld_shadow_t, const float* dyn, hipStream_t stream) {
    VITK_CHECK_ARG(rows >= 0 && cols >= 0, "vitk_adamw: negative shape");
    VITK_CHECK_ARG(cols % 4 == 0 || (!shadow && !shadow_t), "vitk_adamw: cols=%d must be a multiple of 4 when copies are written", cols);
    VITK_CHECK_ARG(step >= 1, "vitk_adamw: step counts from 1");
    if (rows == 0 || cols == 0) return VITK_OK;
    VITK_CHECK_ARG(p && g && m && v, "vitk_adamw: null tensor");
    auto al = [](const void* q, int a) { return (reinterpret_cast<uintptr_t>(q) & (a - 1)) == 0; };
    VITK_CHECK_ARG(cols % 4 != 0 || (al(p, 16) && al(g, 16) && al(m, 16) && al(v, 16)),
                   "vitk_adamw: tensors must be 16-byte aligned");
    VITK_CHECK_ARG(!shadow || (ld_shadow >= cols && ld_shadow % 4 == 0 && al(shadow, 8)),
                   "vitk_adamw: shadow needs ld >= cols, ld %% 4 == 0 and 8-byte alignment");
    VITK_CHECK_ARG(!shadow_t || (ld_shadow_t >= rows && al(shadow_t, 16)),
                   "vitk_adamw: shadow_t needs ld >= rows and 16-byte alignment");
    AdamWParams a;
    a.p = p; a.g = g; a.m = m; a.v = v;
    a.shadow = (__bf16*)shadow; a.ld_shadow = ld_shadow;
    a.shadow_t = (__bf16*)shadow_t; a.ld_shadow_t = ld_shadow_t;
    a.rows = rows; a.cols = cols;
    // hyper-parameters arrive as doubles (Python floats) and are rounded once, as torch rounds them
    a.decay = (float)(1.0 - lr * weight_decay);
    a.b1 = (float)beta1; a.b2 = (float)beta2;
    a.omb1 = (float)(1.0 - beta1);
    a.omb2 = (float)(1.0 - beta2);
    const double bc1 = 1.0 - pow(beta1, (double)step);
    const double bc2 = 1.0 - pow(beta2, (double)step);
    a.step_size = (float)(lr / bc1);
    a.inv_bc2_sqrt = (float)(1.0 / sqrt(bc2));
    a.eps = (float)eps;
    a.dyn = dyn;
    const dim3 grid((cols + OPT_TILE - 1) / OPT_TILE, (rows + OPT_TILE - 1) / OPT_TILE);
    VITK_CHECK_ARG(grid.y <= 65535, "vitk_adamw: too many rows");
    hipLaunchKernelGGL(adamw_kernel, grid, dim3(256), 0, stream, a);
    VITK_CHECK_LAUNCH("vitk_adamw");
    return VITK_OK;
}

extern "C" int vitk_adamw(int rows, int cols, float* p, const float* g, float* m, float* v, double lr, double beta1,
                          double beta2, double eps, double weight_decay, int step, void* shadow, long ld_shadow,
                          void* shadow_t, long ld_shadow_t, hipStream_t stream) {
    return adamw_impl(rows, cols, p, g, m, v, lr, beta1, beta2, eps, weight_decay, step, shadow, ld_shadow, shadow_t,
                      ld_shadow_t, nullptr, stream);
}

// The same step with the three step-dependent scalars read from DEVICE memory at run time:
// dyn = {1 - lr*wd, lr / (1 - beta1^t), 1 / sqrt(1 - beta2^t)} (fp32).  For a training step captured in a HIP
// graph: the host rewrites dyn before every replay, the captured launch stays valid.
extern "C" int vitk_adamw_dyn(int rows, int cols, float* p, const float* g, float* m, float* v, double beta1,
                              double beta2, double eps, const float* dyn, void* shadow, long ld_shadow,
                              void* shadow_t, long ld_shadow_t, hipStream_t stream) {
    VITK_CHECK_ARG(dyn != nullptr, "vitk_adamw_dyn: null dyn");
    return adamw_impl(rows, cols, p, g, m, v, 0.0, beta1, beta2, eps, 0.0, 1, shadow, ld_shadow, shadow_t, ld_shadow_t,
                      dyn, stream);
}

// n <= 4 floats written to device memory from KERNEL ARGUMENTS (by value): the stream-ordered way to hand a captured
// graph its per-replay scalars -- no host buffer has to stay untouched until the copy runs.
extern "C" int vitk_set_f32(float* dst, int n, float v0, float v1, float v2, float v3, hipStream_t stream) {
    VITK_CHECK_ARG(dst && n >= 1 && n <= 4, "vitk_set_f32: n must be 1..4");
    hipLaunchKernelGGL(set_f32_kernel, dim3(1), dim3(64), 0, stream, dst, n, v0, v1, v2, v3);
    VITK_CHECK_LAUNCH("vitk_set_f32");
    return VITK_OK;
}

// Multi-tensor AdamW: every parameter listed in ``desc`` in ONE launch.  desc = DEVICE memory, n + 1 records of
// 9 int64 words {p, g, m, v, shadow, shadow_t, ld_shadow | ld_shadow_t << 32, rows | cols << 32,
// tile0 | tiles_x << 32}: tiles are 64 x 64, tiles_x = ceil(cols / 64), tile0 = the running sum of
// tiles_x * ceil(rows / 64) over the records before; record n carries only tile0 = total_tiles.  The three
// step-dependent scalars come from device memory (dyn = {1 - lr wd, lr / (1 - b1^t), 1 / sqrt(1 - b2^t)}), as in
// vitk_adamw_dyn.  Same per-element arithmetic as vitk_adamw.
extern "C" int vitk_adamw_multi(int n, const void* desc, long total_tiles, double beta1, double beta2, double eps,
                                const float* dyn, hipStream_t stream) {
    VITK_CHECK_ARG(n >= 0 && total_tiles >= 0, "vitk_adamw_multi: negative count");
    if (n == 0 || total_tiles == 0) return VITK_OK;
    VITK_CHECK_ARG(desc && dyn, "vitk_adamw_multi: null table");
    VITK_CHECK_ARG(total_tiles < (1L << 31), "vitk_adamw_multi: too many tiles");
    AdamWParams a = {};
    a.b1 = (float)beta1; a.b2 = (float)beta2;
    a.omb1 = (float)(1.0 - beta1);
    a.omb2 = (float)(1.0 - beta2);
    a.eps = (float)eps;
    a.dyn = dyn;
    long grid = total_tiles < 4096 ? total_tiles : 4096;
    hipLaunchKernelGGL(adamw_multi_kernel, dim3((unsigned)grid), dim3(256), 0, stream,
                       reinterpret_cast<const AdamWDesc*>(desc), n, a);
    VITK_CHECK_LAUNCH("vitk_adamw_multi");
    return VITK_OK;
}

// Size in bytes of one descriptor record (the host checks its layout against it).
extern "C" int vitk_adamw_desc_bytes(void) { return (int)sizeof(AdamWDesc); }
