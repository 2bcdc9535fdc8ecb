// HBM-bound row kernels: LayerNorm forward/backward, column sums (bias gradients), casts,
// token assembly (cls + patches + pos_embed) and patch gathering.
//
// Reference ops replaced: nn.LayerNorm (timm_locality_aware_transformer.py:87,92,123;
// disentangled_transformer.py:99,173,180; model.py:22,44) and their autograd backward;
// the bias-gradient reductions of every nn.Linear; `torch.cat((cls_tokens, x)) + pos_embed`
// (timm_locality_aware_transformer.py:136-138, model.py:54-55); the im2col that a
// Conv2d(k = s = P) amounts to (model.py:92-99).
//
// Each row is owned by one 64-lane wave, 16 bytes (bf16) / 32 bytes (f32) per lane per step,
// row kept in registers between the statistics pass and the normalise pass (D <= 2048).
#include "common.h"

namespace vitk {

constexpr int LN_MAXV = 4;  // 8-element vectors per lane: D <= 64 * 8 * 4 = 2048

template <typename T> __device__ __forceinline__ void load8(float (&v)[8], const T* p);
template <> __device__ __forceinline__ void load8<float>(float (&v)[8], const float* p) {
    const f32x4 a = *reinterpret_cast<const f32x4*>(p);
    const f32x4 b = *reinterpret_cast<const f32x4*>(p + 4);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        v[e] = a[e];
        v[4 + e] = b[e];
    }
}
template <> __device__ __forceinline__ void load8<__bf16>(float (&v)[8], const __bf16* p) {
    const bf16x8 a = *reinterpret_cast<const bf16x8*>(p);
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = (float)a[e];
}
// 8 elements as loaded (packed): lets a kernel keep the NEXT row's loads in flight in few registers
template <typename T> struct Raw8;
template <> struct Raw8<float> {
    f32x4 a, b;
    __device__ __forceinline__ void load(const float* p) {
        a = *reinterpret_cast<const f32x4*>(p);
        b = *reinterpret_cast<const f32x4*>(p + 4);
    }
    __device__ __forceinline__ void unpack(float (&v)[8]) const {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            v[e] = a[e];
            v[4 + e] = b[e];
        }
    }
};
template <> struct Raw8<__bf16> {
    bf16x8 a;
    __device__ __forceinline__ void load(const __bf16* p) { a = *reinterpret_cast<const bf16x8*>(p); }
    __device__ __forceinline__ void unpack(float (&v)[8]) const {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = (float)a[e];
    }
};
template <typename T> __device__ __forceinline__ void store8(T* p, const float (&v)[8]);
template <> __device__ __forceinline__ void store8<float>(float* p, const float (&v)[8]) {
    *reinterpret_cast<f32x4*>(p) = f32x4{v[0], v[1], v[2], v[3]};
    *reinterpret_cast<f32x4*>(p + 4) = f32x4{v[4], v[5], v[6], v[7]};
}
template <> __device__ __forceinline__ void store8<__bf16>(__bf16* p, const float (&v)[8]) {
    bf16x8 o;
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = (__bf16)v[e];
    *reinterpret_cast<bf16x8*>(p) = o;
}

// ------------------------------------------------------------------ LayerNorm forward
template <typename T, int NV>
__global__ __launch_bounds__(256) void layernorm_fwd_kernel(const T* __restrict__ x, long ldx,
                                                            const float* __restrict__ gamma,
                                                            const float* __restrict__ beta, float eps,
                                                            T* __restrict__ y, long ldy,
                                                            float* __restrict__ mean_out,
                                                            float* __restrict__ rstd_out, int rows, int D) {
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int nchunk = D >> 3;
    const long stride = (long)gridDim.x * 4;
    long row = (long)blockIdx.x * 4 + wave;
    if (row >= rows) return;
    // gamma / beta once per wave, and the next row's loads in flight while this one is reduced and stored
    float g[NV][8], b[NV][8], v[NV][8];
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = lane + 64 * i;
        if (c < nchunk) {
            load8<float>(g[i], gamma + c * 8);
            load8<float>(b[i], beta + c * 8);
            load8<T>(v[i], x + row * ldx + c * 8);
        }
    }
    for (; row < rows; row += stride) {
        float nx[NV][8];
        const long nrow = row + stride;
        if (nrow < rows) {
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                const int c = lane + 64 * i;
                if (c < nchunk) load8<T>(nx[i], x + nrow * ldx + c * 8);
            }
        }
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int c = lane + 64 * i;
            if (c < nchunk) {
#pragma unroll
                for (int e = 0; e < 8; ++e) s += v[i][e];
            }
        }
        const float mean = wave_sum(s) / (float)D;
        float q = 0.f;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int c = lane + 64 * i;
            if (c < nchunk) {
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float d = v[i][e] - mean;
                    q += d * d;
                }
            }
        }
        const float rstd = rsqrtf(wave_sum(q) / (float)D + eps);
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int c = lane + 64 * i;
            if (c < nchunk) {
                float o[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) o[e] = (v[i][e] - mean) * rstd * g[i][e] + b[i][e];
                store8<T>(y + row * ldy + c * 8, o);
            }
        }
        if (lane == 0) {
            if (mean_out) mean_out[row] = mean;
            if (rstd_out) rstd_out[row] = rstd;
        }
        if (nrow < rows) {
#pragma unroll
            for (int i = 0; i < NV; ++i)
#pragma unroll
                for (int e = 0; e < 8; ++e) v[i][e] = nx[i][e];
        }
    }
}

// ------------------------------------------------------------------ LayerNorm backward
// dx = rstd * (g*dy - mean(g*dy) - xhat * mean(g*dy*xhat)) [+ add];  dgamma += sum dy*xhat;
// dbeta += sum dy;  optionally dxsum += column sums of dx (the bias gradient of the Linear that
// produced the LayerNorm's input in a post-LN block -- saves a separate pass over dx).  Each wave walks rows with a grid stride and keeps its dgamma/dbeta partial
// sums in registers; waves combine through LDS and issue one fp32 atomic per column per block.
// 8 waves per block, one block per CU: the cost of the final cross-block reduction grows with the
// number of BLOCKS, not waves (16 waves per block spill: 72 us against 52).
constexpr int LN_BWD_WAVES = 8;
constexpr int LN_BWD_THREADS = 64 * LN_BWD_WAVES;
template <typename T, int NV>
__global__ __launch_bounds__(LN_BWD_THREADS) void layernorm_bwd_kernel(const T* __restrict__ dy, long lddy,
                                                            const T* __restrict__ x, long ldx,
                                                            const float* __restrict__ mean,
                                                            const float* __restrict__ rstd,
                                                            const float* __restrict__ gamma,
                                                            const T* __restrict__ add, long ldadd,
                                                            T* __restrict__ dx, long lddx,
                                                            float* __restrict__ dgamma,
                                                            float* __restrict__ dbeta,
                                                            float* __restrict__ dxsum, int rows, int D) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    float* red = reinterpret_cast<float*>(smem_raw);  // [LN_BWD_WAVES][D]
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int nchunk = D >> 3;
    float dg[NV][8], db[NV][8], g[NV][8], ds[NV][8];
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = lane + 64 * i;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            dg[i][e] = 0.f;
            db[i][e] = 0.f;
            ds[i][e] = 0.f;
            g[i][e] = 0.f;
        }
        if (c < nchunk) load8<float>(g[i], gamma + c * 8);
    }
    // The row loop is latency-bound (two waves per SIMD): the next row's dy / x / add loads are issued
    // before this row's reductions, kept packed (12 registers per chunk for bf16).
    const long stride = (long)gridDim.x * LN_BWD_WAVES;
    long row = (long)blockIdx.x * LN_BWD_WAVES + wave;
    Raw8<T> ndy[NV], nx[NV], nadd[NV];
    float nmu = 0.f, nrs = 0.f;
    auto fetch = [&](long r) {
        nmu = mean[r];
        nrs = rstd[r];
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int c = lane + 64 * i;
            if (c < nchunk) {
                ndy[i].load(dy + r * lddy + c * 8);
                nx[i].load(x + r * ldx + c * 8);
                if (add) nadd[i].load(add + r * ldadd + c * 8);
            }
        }
    };
    if (row < rows) fetch(row);
    for (; row < rows; row += stride) {
        const float mu = nmu, rs = nrs;
        float vdy[NV][8], xh[NV][8], va[NV][8];
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            ndy[i].unpack(vdy[i]);
            nx[i].unpack(xh[i]);
            if (add) nadd[i].unpack(va[i]);
        }
        if (row + stride < rows) fetch(row + stride);
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int c = lane + 64 * i;
            if (c < nchunk) {
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    xh[i][e] = (xh[i][e] - mu) * rs;
                    const float gd = g[i][e] * vdy[i][e];
                    s1 += gd;
                    s2 += gd * xh[i][e];
                    dg[i][e] += vdy[i][e] * xh[i][e];
                    db[i][e] += vdy[i][e];
                }
            }
        }
        s1 = wave_sum(s1) / (float)D;
        s2 = wave_sum(s2) / (float)D;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int c = lane + 64 * i;
            if (c < nchunk) {
                float o[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) o[e] = rs * (g[i][e] * vdy[i][e] - s1 - xh[i][e] * s2);
                if (add) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) o[e] += va[i][e];
                }
                if (dxsum) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) ds[i][e] += o[e];
                }
                store8<T>(dx + row * lddx + c * 8, o);
            }
        }
    }
    // Block reduction without LDS atomics (lane-strided LDS atomics are 8-way bank conflicts: ~10 us
    // per array): every wave parks its partial row in LDS, the block sums the 8 rows per column and
    // issues one global atomic per column.  Every block finishes at about the same time: each
    // block starts at a rotated column so that concurrent blocks add to different addresses.
    const int rot = (blockIdx.x * 256) % D;
    auto flush = [&](const float (&part)[NV][8], float* __restrict__ out) {
        __syncthreads();                       // the previous array's column sums are done with `red`
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int c = lane + 64 * i;
            if (c < nchunk) {
                float* dst = red + wave * D + c * 8;
                *reinterpret_cast<f32x4*>(dst) = f32x4{part[i][0], part[i][1], part[i][2], part[i][3]};
                *reinterpret_cast<f32x4*>(dst + 4) = f32x4{part[i][4], part[i][5], part[i][6], part[i][7]};
            }
        }
        __syncthreads();
        for (int i = threadIdx.x; i < D; i += LN_BWD_THREADS) {
            int c = i + rot;
            if (c >= D) c -= D;
            float sum = 0.f;
#pragma unroll
            for (int w = 0; w < LN_BWD_WAVES; ++w) sum += red[w * D + c];
            atomicAdd(&out[c], sum);
        }
    };
    if (dgamma) flush(dg, dgamma);
    if (dbeta) flush(db, dbeta);
    if (dxsum) flush(ds, dxsum);
}

// ------------------------------------------------------------------ column sums
// out[n] (+)= sum_m X[m][n].  256 threads = 16 row lanes x 16 column groups of 8.
template <typename T>
__global__ __launch_bounds__(256) void colsum_kernel(const T* __restrict__ X, long ldx, int M, int N,
                                                     float* __restrict__ out) {
    __shared__ float red[16][129];
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
    const int n0 = blockIdx.x * 128 + tx * 8;
    float s[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) s[e] = 0.f;
    if (n0 < N) {
        const bool full = n0 + 8 <= N;
        for (long m = (long)blockIdx.y * 16 + ty; m < M; m += (long)gridDim.y * 16) {
            if (full) {
                float v[8];
                load8<T>(v, X + m * ldx + n0);
#pragma unroll
                for (int e = 0; e < 8; ++e) s[e] += v[e];
            } else {
                for (int e = 0; e < 8 && n0 + e < N; ++e) s[e] += to_f32<T>(X[m * ldx + n0 + e]);
            }
        }
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) red[ty][tx * 8 + e] = s[e];
    __syncthreads();
    if (threadIdx.x < 128) {
        float t = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) t += red[r][threadIdx.x];
        const int n = blockIdx.x * 128 + threadIdx.x;
        if (n < N) atomicAdd(&out[n], t);
    }
}

// unaligned / odd-width variant: one thread per column, rows strided over blockIdx.y
template <typename T>
__global__ void colsum_slow_kernel(const T* __restrict__ X, long ldx, int M, int N, float* __restrict__ out) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N) return;
    float s = 0.f;
    for (long m = blockIdx.y; m < M; m += gridDim.y) s += to_f32<T>(X[m * ldx + n]);
    atomicAdd(&out[n], s);
}

// ------------------------------------------------------------------ casts / elementwise
__global__ void cast_f32_bf16_kernel(const float* __restrict__ src, __bf16* __restrict__ dst, long n) {
    const long stride = (long)gridDim.x * blockDim.x * 8;
    for (long i = ((long)blockIdx.x * blockDim.x + threadIdx.x) * 8; i < n; i += stride) {
        if (i + 8 <= n) {
            float v[8];
            load8<float>(v, src + i);
            store8<__bf16>(dst + i, v);
        } else {
            for (long j = i; j < n; ++j) dst[j] = (__bf16)src[j];
        }
    }
}
__global__ void cast_bf16_f32_kernel(const __bf16* __restrict__ src, float* __restrict__ dst, long n) {
    const long stride = (long)gridDim.x * blockDim.x * 8;
    for (long i = ((long)blockIdx.x * blockDim.x + threadIdx.x) * 8; i < n; i += stride) {
        if (i + 8 <= n) {
            float v[8];
            load8<__bf16>(v, src + i);
            store8<float>(dst + i, v);
        } else {
            for (long j = i; j < n; ++j) dst[j] = (float)src[j];
        }
    }
}

// ------------------------------------------------------------------ patches
// out[(b*nph + ph)*npw + pw][(c*P + py)*P + px] = img[b][c][ph*P+py][pw*P+px]
// (the row order of Conv2d(k=s=P).flatten(2).transpose(1,2), the column order of
// conv.weight.view(D, C*P*P)).  img is fp32 or bf16 (TI); out is T.
template <typename TI, typename T>
__global__ void patchify_kernel(const TI* __restrict__ img, T* __restrict__ out, int B, int C, int Hh, int Ww,
                                int P) {
    const int nph = Hh / P, npw = Ww / P;
    const long total = (long)B * C * Hh * Ww;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        // i indexes the OUTPUT so stores coalesce; px runs fastest in both images
        const int px = i % P;
        long t = i / P;
        const int py = t % P;
        t /= P;
        const int c = t % C;
        t /= C;
        const int pw = t % npw;
        t /= npw;
        const int ph = t % nph;
        const int b = t / nph;
        out[i] = from_f32<T>(to_f32<TI>(img[(((long)b * C + c) * Hh + ph * P + py) * Ww + pw * P + px]));
    }
}
// The same gather from raw uint8 pixels, with ToTensor + Normalize folded in (train.py:340-341,362-363:
// x / 255, then (x - mean[c]) / std[c], in that order and with true divisions, so fp32 output equals torchvision's
// bit for bit).  img is [B][C][H][W] or, channels_last, [B][H][W][C] as decoders deliver it.
struct PixelNorm {
    float mean[4], std[4];
};
template <typename T>
__global__ void patchify_u8_kernel(const unsigned char* __restrict__ img, T* __restrict__ out, int B, int C, int Hh, int Ww,
                                   int P, int channels_last, const PixelNorm nrm) {
    const int nph = Hh / P, npw = Ww / P;
    const long total = (long)B * C * Hh * Ww;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int px = i % P;
        long t = i / P;
        const int py = t % P;
        t /= P;
        const int c = t % C;
        t /= C;
        const int pw = t % npw;
        t /= npw;
        const int ph = t % nph;
        const int b = t / nph;
        const int y = ph * P + py, x = pw * P + px;
        const long src = channels_last ? (((long)b * Hh + y) * Ww + x) * C + c : (((long)b * C + c) * Hh + y) * Ww + x;
        const float v = (float)img[src] / 255.0f;
        out[i] = from_f32<T>((v - nrm.mean[c]) / nrm.std[c]);
    }
}
// scatter-back for d(img): the exact inverse permutation
template <typename T>
__global__ void unpatchify_kernel(const T* __restrict__ dpatch, float* __restrict__ dimg, int B, int C, int Hh,
                                  int Ww, int P) {
    const int nph = Hh / P, npw = Ww / P;
    const long total = (long)B * C * Hh * Ww;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int px = i % P;
        long t = i / P;
        const int py = t % P;
        t /= P;
        const int c = t % C;
        t /= C;
        const int pw = t % npw;
        t /= npw;
        const int ph = t % nph;
        const int b = t / nph;
        dimg[(((long)b * C + c) * Hh + ph * P + py) * Ww + pw * P + px] = to_f32<T>(dpatch[i]);
    }
}

// x[b][0] = cls + pos[0] (pos may be null).  Patch rows are written by the GEMM epilogue.
template <typename T>
__global__ void cls_rows_kernel(const float* __restrict__ cls, const float* __restrict__ pos, T* __restrict__ x,
                                int B, int N, int D) {
    const long total = (long)B * D;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int d = i % D;
        const long b = i / D;
        x[b * N * D + d] = from_f32<T>(cls[d] + (pos ? pos[d] : 0.f));
    }
}

// d(pos)[n][d] (+)= sum_b dx[b][n][d]; d(cls)[d] (+)= sum_b dx[b][0][d]
template <typename T>
__global__ void token_grads_kernel(const T* __restrict__ dx, float* __restrict__ dpos, float* __restrict__ dcls,
                                   int B, int N, int D) {
    const long total = (long)N * D;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        float s = 0.f;
        for (int b = 0; b < B; ++b) s += to_f32<T>(dx[(long)b * N * D + i]);
        if (dpos) dpos[i] += s;
        if (dcls && i < D) dcls[i] += s;
    }
}

}  // namespace vitk

using namespace vitk;

static bool al16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

template <typename T>
static int ln_fwd_dispatch(int rows, int D, const T* x, long ldx, const float* gamma, const float* beta, float eps,
                           T* y, long ldy, float* mean, float* rstd, hipStream_t st) {
    const int nv = (D / 8 + 63) / 64;
    const int grid = rows < 4 * 2048 ? (rows + 3) / 4 : 2048;
#define L(NV) hipLaunchKernelGGL((layernorm_fwd_kernel<T, NV>), dim3(grid), dim3(256), 0, st, x, ldx, gamma, beta, eps, y, ldy, mean, rstd, rows, D)
    if (nv <= 1) L(1); else if (nv <= 2) L(2); else if (nv <= 3) L(3); else L(4);
#undef L
    return 0;
}

extern "C" int vitk_layernorm_fwd(int dtype, int rows, int D, const void* x, long ldx, const float* gamma,
                                  const float* beta, float eps, void* y, long ldy, float* mean, float* rstd,
                                  hipStream_t stream) {
    VITK_CHECK_ARG(dtype == VITK_F32 || dtype == VITK_BF16, "vitk_layernorm_fwd: bad dtype");
    VITK_CHECK_ARG(D > 0 && D % 8 == 0 && D <= 64 * 8 * LN_MAXV, "vitk_layernorm_fwd: D=%d must be a multiple of 8, <= 2048", D);
    VITK_CHECK_ARG(ldx % 8 == 0 && ldy % 8 == 0 && al16(x) && al16(y) && al16(gamma) && al16(beta),
                   "vitk_layernorm_fwd: rows must be 16-byte aligned");
    if (rows == 0) return VITK_OK;
    if (dtype == VITK_BF16)
        ln_fwd_dispatch<__bf16>(rows, D, (const __bf16*)x, ldx, gamma, beta, eps, (__bf16*)y, ldy, mean, rstd, stream);
    else
        ln_fwd_dispatch<float>(rows, D, (const float*)x, ldx, gamma, beta, eps, (float*)y, ldy, mean, rstd, stream);
    VITK_CHECK_LAUNCH("vitk_layernorm_fwd");
    return VITK_OK;
}

template <typename T>
static void ln_bwd_dispatch(int rows, int D, const T* dy, long lddy, const T* x, long ldx, const float* mean,
                            const float* rstd, const float* gamma, const T* add, long ldadd, T* dx, long lddx,
                            float* dgamma, float* dbeta, float* dxsum, hipStream_t st) {
    const int nv = (D / 8 + 63) / 64;
    int grid = (rows + LN_BWD_WAVES - 1) / LN_BWD_WAVES;
    // one block per CU: the rows are dealt out evenly over the blocks, so a CU holding two blocks
    // would finish last, and the final cross-block atomics grow with the block count
    // (MI355X, rows 25216 x 1024 bf16: 256 blocks 52 us, 384 -> 80, 512 -> 76, 1024 -> 124)
    static int cus = 0;
    if (cus == 0) {
        int dev = 0, n = 0;
        if (hipGetDevice(&dev) != hipSuccess ||
            hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0)
            n = 256;
        cus = n;
    }
    if (grid > cus) grid = cus;
    const size_t lds = (size_t)LN_BWD_WAVES * D * sizeof(float);
#define L(NV) hipLaunchKernelGGL((layernorm_bwd_kernel<T, NV>), dim3(grid), dim3(LN_BWD_THREADS), lds, st, dy, lddy, x, ldx, mean, rstd, gamma, add, ldadd, dx, lddx, dgamma, dbeta, dxsum, rows, D)
    if (nv <= 1) L(1); else if (nv <= 2) L(2); else if (nv <= 3) L(3); else L(4);
#undef L
}

// dgamma / dbeta / dxsum are ACCUMULATED into (fp32, caller zeroes them when a fresh sum is wanted).
extern "C" int vitk_layernorm_bwd(int dtype, int rows, int D, const void* dy, long lddy, const void* x, long ldx,
                                  const float* mean, const float* rstd, const float* gamma, const void* add,
                                  long ldadd, void* dx, long lddx, float* dgamma, float* dbeta, float* dxsum,
                                  hipStream_t stream) {
    VITK_CHECK_ARG(dtype == VITK_F32 || dtype == VITK_BF16, "vitk_layernorm_bwd: bad dtype");
    VITK_CHECK_ARG(D > 0 && D % 8 == 0 && D <= 64 * 8 * LN_MAXV, "vitk_layernorm_bwd: D=%d must be a multiple of 8, <= 2048", D);
    VITK_CHECK_ARG(lddy % 8 == 0 && ldx % 8 == 0 && lddx % 8 == 0 && (add == nullptr || ldadd % 8 == 0) &&
                       al16(dy) && al16(x) && al16(dx) && al16(gamma) && (add == nullptr || al16(add)),
                   "vitk_layernorm_bwd: rows must be 16-byte aligned");
    VITK_CHECK_ARG(mean && rstd, "vitk_layernorm_bwd: null statistics");
    if (rows == 0) return VITK_OK;
    if (dtype == VITK_BF16)
        ln_bwd_dispatch<__bf16>(rows, D, (const __bf16*)dy, lddy, (const __bf16*)x, ldx, mean, rstd, gamma,
                                (const __bf16*)add, ldadd, (__bf16*)dx, lddx, dgamma, dbeta, dxsum, stream);
    else
        ln_bwd_dispatch<float>(rows, D, (const float*)dy, lddy, (const float*)x, ldx, mean, rstd, gamma,
                               (const float*)add, ldadd, (float*)dx, lddx, dgamma, dbeta, dxsum, stream);
    VITK_CHECK_LAUNCH("vitk_layernorm_bwd");
    return VITK_OK;
}

// out[n] += sum_m X[m][n]   (fp32 out, accumulated; caller zeroes for a fresh sum)
extern "C" int vitk_colsum(int dtype, int M, int N, const void* X, long ldx, float* out, hipStream_t stream) {
    VITK_CHECK_ARG(dtype == VITK_F32 || dtype == VITK_BF16, "vitk_colsum: bad dtype");
    VITK_CHECK_ARG(X && out && ldx >= N, "vitk_colsum: bad arguments");
    if (M == 0 || N == 0) return VITK_OK;
    const bool fast = al16(X) && (ldx % 8 == 0);
    if (fast) {
        int gy = (M + 255) / 256;
        if (gy > 128) gy = 128;
        if (gy < 1) gy = 1;
        dim3 grid((N + 127) / 128, gy);
        if (dtype == VITK_BF16)
            hipLaunchKernelGGL(colsum_kernel<__bf16>, grid, dim3(256), 0, stream, (const __bf16*)X, ldx, M, N, out);
        else
            hipLaunchKernelGGL(colsum_kernel<float>, grid, dim3(256), 0, stream, (const float*)X, ldx, M, N, out);
    } else {
        int gy = (M + 63) / 64;
        if (gy > 256) gy = 256;
        dim3 grid((N + 63) / 64, gy);
        if (dtype == VITK_BF16)
            hipLaunchKernelGGL(colsum_slow_kernel<__bf16>, grid, dim3(64), 0, stream, (const __bf16*)X, ldx, M, N, out);
        else
            hipLaunchKernelGGL(colsum_slow_kernel<float>, grid, dim3(64), 0, stream, (const float*)X, ldx, M, N, out);
    }
    VITK_CHECK_LAUNCH("vitk_colsum");
    return VITK_OK;
}

// direction 0: f32 -> bf16, 1: bf16 -> f32
extern "C" int vitk_cast(int direction, long n, const void* src, void* dst, hipStream_t stream) {
    VITK_CHECK_ARG(src && dst && n >= 0, "vitk_cast: bad arguments");
    VITK_CHECK_ARG(al16(src) && al16(dst), "vitk_cast: buffers must be 16-byte aligned");
    if (n == 0) return VITK_OK;
    long blocks = (n / 8 + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    if (blocks < 1) blocks = 1;
    if (direction == 0)
        hipLaunchKernelGGL(cast_f32_bf16_kernel, dim3((int)blocks), dim3(256), 0, stream, (const float*)src, (__bf16*)dst, n);
    else
        hipLaunchKernelGGL(cast_bf16_f32_kernel, dim3((int)blocks), dim3(256), 0, stream, (const __bf16*)src, (float*)dst, n);
    VITK_CHECK_LAUNCH("vitk_cast");
    return VITK_OK;
}

extern "C" int vitk_patchify(int img_dtype, int out_dtype, int B, int C, int H, int W, int P, const void* img,
                             void* out, hipStream_t stream) {
    VITK_CHECK_ARG(P > 0 && H % P == 0 && W % P == 0, "vitk_patchify: image %dx%d not divisible by patch %d", H, W, P);
    VITK_CHECK_ARG(img && out, "vitk_patchify: null pointer");
    const long total = (long)B * C * H * W;
    if (total == 0) return VITK_OK;
    long blocks = (total + 255) / 256;
    if (blocks > 8192) blocks = 8192;
#define L(TI, TO) hipLaunchKernelGGL((patchify_kernel<TI, TO>), dim3((int)blocks), dim3(256), 0, stream, (const TI*)img, (TO*)out, B, C, H, W, P)
    if (img_dtype == VITK_F32 && out_dtype == VITK_F32) L(float, float);
    else if (img_dtype == VITK_F32 && out_dtype == VITK_BF16) L(float, __bf16);
    else if (img_dtype == VITK_BF16 && out_dtype == VITK_BF16) L(__bf16, __bf16);
    else { vitk_set_error("vitk_patchify: unsupported dtype pair"); return VITK_ERR_ARG; }
#undef L
    VITK_CHECK_LAUNCH("vitk_patchify");
    return VITK_OK;
}

// Patches straight from uint8 images: ToTensor + Normalize(mean, std) + the patch gather in one pass (1 byte read per
// pixel instead of 4: the host sends uint8).  mean / std: C values each in HOST memory, C <= 4.
extern "C" int vitk_patchify_u8(int out_dtype, int B, int C, int H, int W, int P, int channels_last,
                                const unsigned char* img, const float* mean, const float* stdv, void* out,
                                hipStream_t stream) {
    VITK_CHECK_ARG(P > 0 && H % P == 0 && W % P == 0, "vitk_patchify_u8: image %dx%d not divisible by patch %d", H, W, P);
    VITK_CHECK_ARG(C >= 1 && C <= 4, "vitk_patchify_u8: 1..4 channels supported, got %d", C);
    VITK_CHECK_ARG(img && out && mean && stdv, "vitk_patchify_u8: null pointer");
    PixelNorm nrm = {};
    for (int c = 0; c < C; ++c) {
        VITK_CHECK_ARG(stdv[c] != 0.f, "vitk_patchify_u8: std[%d] is zero", c);
        nrm.mean[c] = mean[c];
        nrm.std[c] = stdv[c];
    }
    const long total = (long)B * C * H * W;
    if (total == 0) return VITK_OK;
    long blocks = (total + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    if (out_dtype == VITK_F32)
        hipLaunchKernelGGL(patchify_u8_kernel<float>, dim3((int)blocks), dim3(256), 0, stream, img, (float*)out, B, C, H, W, P, channels_last, nrm);
    else if (out_dtype == VITK_BF16)
        hipLaunchKernelGGL(patchify_u8_kernel<__bf16>, dim3((int)blocks), dim3(256), 0, stream, img, (__bf16*)out, B, C, H, W, P, channels_last, nrm);
    else { vitk_set_error("vitk_patchify_u8: unsupported output dtype"); return VITK_ERR_ARG; }
    VITK_CHECK_LAUNCH("vitk_patchify_u8");
    return VITK_OK;
}

extern "C" int vitk_unpatchify(int dtype, int B, int C, int H, int W, int P, const void* dpatch, float* dimg,
                               hipStream_t stream) {
    VITK_CHECK_ARG(P > 0 && H % P == 0 && W % P == 0 && dpatch && dimg, "vitk_unpatchify: bad arguments");
    const long total = (long)B * C * H * W;
    if (total == 0) return VITK_OK;
    long blocks = (total + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    if (dtype == VITK_BF16)
        hipLaunchKernelGGL(unpatchify_kernel<__bf16>, dim3((int)blocks), dim3(256), 0, stream, (const __bf16*)dpatch, dimg, B, C, H, W, P);
    else
        hipLaunchKernelGGL(unpatchify_kernel<float>, dim3((int)blocks), dim3(256), 0, stream, (const float*)dpatch, dimg, B, C, H, W, P);
    VITK_CHECK_LAUNCH("vitk_unpatchify");
    return VITK_OK;
}

extern "C" int vitk_cls_rows(int dtype, int B, int N, int D, const float* cls, const float* pos, void* x,
                             hipStream_t stream) {
    VITK_CHECK_ARG(cls && x, "vitk_cls_rows: null pointer");
    long blocks = ((long)B * D + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    if (blocks < 1) return VITK_OK;
    if (dtype == VITK_BF16)
        hipLaunchKernelGGL(cls_rows_kernel<__bf16>, dim3((int)blocks), dim3(256), 0, stream, cls, pos, (__bf16*)x, B, N, D);
    else
        hipLaunchKernelGGL(cls_rows_kernel<float>, dim3((int)blocks), dim3(256), 0, stream, cls, pos, (float*)x, B, N, D);
    VITK_CHECK_LAUNCH("vitk_cls_rows");
    return VITK_OK;
}

// dpos [N][D] and dcls [D] are ACCUMULATED into (either may be null).
extern "C" int vitk_token_grads(int dtype, int B, int N, int D, const void* dx, float* dpos, float* dcls,
                                hipStream_t stream) {
    VITK_CHECK_ARG(dx, "vitk_token_grads: null pointer");
    long blocks = ((long)N * D + 255) / 256;
    if (blocks < 1) return VITK_OK;
    if (dtype == VITK_BF16)
        hipLaunchKernelGGL(token_grads_kernel<__bf16>, dim3((int)blocks), dim3(256), 0, stream, (const __bf16*)dx, dpos, dcls, B, N, D);
    else
        hipLaunchKernelGGL(token_grads_kernel<float>, dim3((int)blocks), dim3(256), 0, stream, (const float*)dx, dpos, dcls, B, N, D);
    VITK_CHECK_LAUNCH("vitk_token_grads");
    return VITK_OK;
}

// ------------------------------------------------------------------------------------------------ dropout
// y = x * keep / (1 - p) [+ res] over a [rows, cols] row view (nn.Dropout at disentangled_transformer.py:189,195,883,
// timm_locality_aware_transformer.py:78, timm Mlp's drop, timm_vision_performer.py:61, model.py:201-203): element
// (r, c) is element r * cols + c of the stream (seed, offset).  The backward is the same kernel on dy (no residual).
// grp > 0: DropPath (stochastic depth, timm_locality_aware_transformer.py:94,97-98): ONE keep decision per sample, rows
// [s * grp, (s + 1) * grp) share element s of the stream.
namespace vitk {
template <typename T>
__global__ __launch_bounds__(256) void dropout_kernel(const T* __restrict__ x, long ldx, const T* __restrict__ res, long ldr,
                                                      T* __restrict__ y, long ldy, int rows, int cols, int grp, DropParams d) {
    constexpr int V = 16 / sizeof(T);
    const int cv = cols / V;          // host guarantees cols % V == 0
    const long total = (long)rows * cv;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int r = (int)(i / cv), c = (int)(i - (long)r * cv) * V;
        const uint64_t base = (uint64_t)r * (uint64_t)cols + (uint64_t)c;
        const float mul_s = grp > 0 ? drop_mul(d, (uint64_t)(r / grp)) : 0.f;
        if (sizeof(T) == 2) {
            bf16x8 v = *reinterpret_cast<const bf16x8*>(x + r * ldx + c);
            bf16x8 a = {};
            if (res) a = *reinterpret_cast<const bf16x8*>(res + r * ldr + c);
            bf16x8 o;
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = (__bf16)((float)v[e] * (grp > 0 ? mul_s : drop_mul(d, base + e)) + (res ? (float)a[e] : 0.f));
            *reinterpret_cast<bf16x8*>(y + r * ldy + c) = o;
        } else {
            f32x4 v = *reinterpret_cast<const f32x4*>(x + r * ldx + c);
            f32x4 a = {0.f, 0.f, 0.f, 0.f};
            if (res) a = *reinterpret_cast<const f32x4*>(res + r * ldr + c);
            f32x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = v[e] * (grp > 0 ? mul_s : drop_mul(d, base + e)) + a[e];
            *reinterpret_cast<f32x4*>(y + r * ldy + c) = o;
        }
    }
}

__global__ __launch_bounds__(256) void dropout_mask_kernel(unsigned char* __restrict__ out, long n, DropParams d) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) out[i] = drop_keep(d, (uint64_t)i) ? 1 : 0;
}
}  // namespace vitk

static int dropout_launch(const char* who, int dtype, int rows, int cols, int grp, const void* x, long ldx, const void* res,
                          long ldr, void* y, long ldy, float p, unsigned long long seed, unsigned long long offset,
                          hipStream_t stream) {
    VITK_CHECK_ARG(dtype == VITK_F32 || dtype == VITK_BF16, "%s: bad dtype", who);
    VITK_CHECK_ARG(rows >= 0 && cols >= 0, "%s: negative shape", who);
    VITK_CHECK_ARG(p >= 0.f && p < 1.f, "%s: p must be in [0, 1)", who);
    if (rows == 0 || cols == 0) return VITK_OK;
    const int esz = dtype == VITK_BF16 ? 2 : 4, V = 16 / esz;
    auto ok = [&](const void* q, long ld) {
        return q != nullptr && (reinterpret_cast<uintptr_t>(q) & 15) == 0 && ld >= cols && (ld * esz) % 16 == 0;
    };
    VITK_CHECK_ARG(cols % V == 0 && ok(x, ldx) && ok(y, ldy) && (res == nullptr || ok(res, ldr)),
                   "%s: rows must be 16-byte aligned, cols a multiple of %d", who, V);
    const DropParams d = vitk_drop_params(p, seed, offset);
    const long total = (long)rows * (cols / V);
    const unsigned grid = (unsigned)(total / 256 + 1 < 8192 ? total / 256 + 1 : 8192);
    if (dtype == VITK_BF16)
        hipLaunchKernelGGL(dropout_kernel<__bf16>, dim3(grid), dim3(256), 0, stream, (const __bf16*)x, ldx, (const __bf16*)res, ldr,
                           (__bf16*)y, ldy, rows, cols, grp, d);
    else
        hipLaunchKernelGGL(dropout_kernel<float>, dim3(grid), dim3(256), 0, stream, (const float*)x, ldx, (const float*)res, ldr,
                           (float*)y, ldy, rows, cols, grp, d);
    VITK_CHECK_LAUNCH(who);
    return VITK_OK;
}

extern "C" int vitk_dropout(int dtype, int rows, int cols, const void* x, long ldx, const void* res, long ldr, void* y,
                            long ldy, float p, unsigned long long seed, unsigned long long offset, hipStream_t stream) {
    return dropout_launch("vitk_dropout", dtype, rows, cols, 0, x, ldx, res, ldr, y, ldy, p, seed, offset, stream);
}

// DropPath: y = x * keep[r / rows_per_sample] / (1 - p) [+ res]; sample s is element s of the stream (vitk_dropout_mask
// over the batch gives the decisions).  The backward is the same call on dy.
extern "C" int vitk_drop_path(int dtype, int rows, int cols, int rows_per_sample, const void* x, long ldx, const void* res,
                              long ldr, void* y, long ldy, float p, unsigned long long seed, unsigned long long offset,
                              hipStream_t stream) {
    VITK_CHECK_ARG(rows_per_sample > 0 && rows % rows_per_sample == 0,
                   "vitk_drop_path: rows (%d) must be a whole number of samples of %d rows", rows, rows_per_sample);
    return dropout_launch("vitk_drop_path", dtype, rows, cols, rows_per_sample, x, ldx, res, ldr, y, ldy, p, seed, offset, stream);
}

// out[i] = 1 if element i of the stream (seed, offset) is kept at probability p, else 0: the mask any kernel of this
// library applies to element i.  A test hook (the oracle is fed this mask), not part of the training path.
extern "C" int vitk_dropout_mask(long n, float p, unsigned long long seed, unsigned long long offset, unsigned char* out,
                                 hipStream_t stream) {
    VITK_CHECK_ARG(n >= 0 && p >= 0.f && p < 1.f, "vitk_dropout_mask: bad argument");
    if (n == 0) return VITK_OK;
    VITK_CHECK_ARG(out, "vitk_dropout_mask: null output");
    const unsigned grid = (unsigned)(n / 256 + 1 < 8192 ? n / 256 + 1 : 8192);
    hipLaunchKernelGGL(dropout_mask_kernel, dim3(grid), dim3(256), 0, stream, out, n, vitk_drop_params(p, seed, offset));
    VITK_CHECK_LAUNCH("vitk_dropout_mask");
    return VITK_OK;
}
