// Shared by the MFMA GEMM kernels: parameter block and the fused epilogue.
#pragma once
#include "common.h"

namespace vitk {

enum { ACT_NONE = 0, ACT_GELU = 1, ACT_RELU = 2, ACT_DGELU = 3, ACT_DRELU = 4 };

struct GemmParams {
    const void* A;
    const void* B;
    void* C;
    int M, N, K;
    long lda, ldb, ldc;
    const float* bias;     // [N] fp32 or null
    int act;
    const void* aux_in;    // act 3/4: pre-activation (dgelu) or activation output (drelu), dtype T
    void* aux_out;         // act 1: pre-activation copy, dtype T
    long ld_aux;
    const void* residual;  // dtype T, added after the activation
    long ldr;
    int res_period;        // 0: residual row = output row; >0: row = res_off + (r % res_period)
    int res_off;
    int grp_rows;          // 0: identity; >0: out row = (r / grp_rows) * grp_stride + grp_off + r % grp_rows
    int grp_stride, grp_off;
    int accumulate;        // C += (fp32 C only)
    int splitk;
    int a_vec_ok, b_vec_ok;  // 16-byte vector loads legal (base + ld aligned)
    int c_vec_ok;            // vector epilogue legal (C/aux/residual bases + lds aligned, N % 4 == 0)
    int c_vec8_ok;           // 8-column (16-byte) epilogue accesses legal
    float alpha;
    int tiles_m, tiles_n;
    int debug;
    int stagger;             // 256-tile kernel: start delay (shader clock ticks) of the workgroups that can afford one, 0 = none
};


// Epilogue for one lane-owned group of 4 consecutive output columns: v[e] = acc of C[m][n4 + e].
// Order: alpha, bias (first K split only), activation (saving / using aux), residual, then
// store / accumulate / atomic add (split-K).
template <typename T, typename TC>
__device__ __forceinline__ void gemm_epilogue4(const GemmParams& p, int split, int m, int n4, f32x4 acc) {
    if (m >= p.M || n4 >= p.N) return;
    TC* __restrict__ C = reinterpret_cast<TC*>(p.C);
    const T* __restrict__ aux_in = reinterpret_cast<const T*>(p.aux_in);
    T* __restrict__ aux_out = reinterpret_cast<T*>(p.aux_out);
    const T* __restrict__ res = reinterpret_cast<const T*>(p.residual);
    const bool atomic = p.splitk > 1;
    const bool add_bias = p.bias != nullptr && split == 0;
    long orow = m;
    if (p.grp_rows > 0) orow = (long)(m / p.grp_rows) * p.grp_stride + p.grp_off + m % p.grp_rows;
    const long rrow = p.res_period > 0 ? (long)(p.res_off + m % p.res_period) : orow;
    float v[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = acc[e] * p.alpha;
    const bool full = p.c_vec_ok && (n4 + 3 < p.N);
    if (full) {
        // vector path: every side operand moves as one 8- / 16-byte access
        if (add_bias) {
            const f32x4 b = *reinterpret_cast<const f32x4*>(p.bias + n4);
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] += b[e];
        }
        if (p.act == ACT_GELU) {
            if (aux_out) {
                if (sizeof(T) == 2)
                    *reinterpret_cast<bf16x4*>(aux_out + orow * p.ld_aux + n4) =
                        bf16x4{(__bf16)v[0], (__bf16)v[1], (__bf16)v[2], (__bf16)v[3]};
                else
                    *reinterpret_cast<f32x4*>(aux_out + orow * p.ld_aux + n4) = f32x4{v[0], v[1], v[2], v[3]};
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = gelu_erf(v[e]);
        } else if (p.act == ACT_RELU) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
        } else if (p.act == ACT_DGELU || p.act == ACT_DRELU) {
            float h[4];
            if (sizeof(T) == 2) {
                const bf16x4 a = *reinterpret_cast<const bf16x4*>(aux_in + orow * p.ld_aux + n4);
#pragma unroll
                for (int e = 0; e < 4; ++e) h[e] = (float)a[e];
            } else {
                const f32x4 a = *reinterpret_cast<const f32x4*>(aux_in + orow * p.ld_aux + n4);
#pragma unroll
                for (int e = 0; e < 4; ++e) h[e] = a[e];
            }
#pragma unroll
            for (int e = 0; e < 4; ++e)
                v[e] *= (p.act == ACT_DGELU) ? gelu_erf_grad(h[e]) : (h[e] > 0.f ? 1.f : 0.f);
        }
        if (res) {
            if (sizeof(T) == 2) {
                const bf16x4 a = *reinterpret_cast<const bf16x4*>(res + rrow * p.ldr + n4);
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] += (float)a[e];
            } else {
                const f32x4 a = *reinterpret_cast<const f32x4*>(res + rrow * p.ldr + n4);
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] += a[e];
            }
        }
        TC* dst = C + orow * p.ldc + n4;
        if (atomic) {
#pragma unroll
            for (int e = 0; e < 4; ++e) atomicAdd(reinterpret_cast<float*>(dst) + e, v[e]);
        } else if (sizeof(TC) == 4) {
            f32x4 o = {v[0], v[1], v[2], v[3]};
            if (p.accumulate) o += *reinterpret_cast<const f32x4*>(dst);
            *reinterpret_cast<f32x4*>(dst) = o;
        } else {
            *reinterpret_cast<bf16x4*>(dst) = bf16x4{(__bf16)v[0], (__bf16)v[1], (__bf16)v[2], (__bf16)v[3]};
        }
        return;
    }
    // scalar tail path
    if (add_bias) {
#pragma unroll
        for (int e = 0; e < 4; ++e)
            if (n4 + e < p.N) v[e] += p.bias[n4 + e];
    }
    if (p.act == ACT_GELU) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            if (aux_out && n4 + e < p.N) aux_out[orow * p.ld_aux + n4 + e] = from_f32<T>(v[e]);
            v[e] = gelu_erf(v[e]);
        }
    } else if (p.act == ACT_RELU) {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
    } else if (p.act == ACT_DGELU || p.act == ACT_DRELU) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            if (n4 + e < p.N) {
                const float h = to_f32<T>(aux_in[orow * p.ld_aux + n4 + e]);
                v[e] *= (p.act == ACT_DGELU) ? gelu_erf_grad(h) : (h > 0.f ? 1.f : 0.f);
            }
        }
    }
    if (res) {
#pragma unroll
        for (int e = 0; e < 4; ++e)
            if (n4 + e < p.N) v[e] += to_f32<T>(res[rrow * p.ldr + n4 + e]);
    }
    TC* dst = C + orow * p.ldc + n4;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        if (n4 + e < p.N) {
            if (atomic) {
                atomicAdd(reinterpret_cast<float*>(dst) + e, v[e]);
            } else {
                float o = v[e];
                if (p.accumulate) o += to_f32<TC>(dst[e]);
                dst[e] = from_f32<TC>(o);
            }
        }
    }
}

// Same for 8 consecutive columns (16-byte bf16 accesses): used by the 256-tile kernel, whose
// epilogue re-reads the accumulators row-major from LDS so that 8 lanes cover a 128-byte line.
// gemm_epilogue8_fast: the caller has checked c_vec8_ok, n8 + 7 < N, splitk == 1, bf16 operands.
template <typename T, typename TC>
__device__ __forceinline__ void gemm_epilogue8_fast(const GemmParams& p, int m, int n8, const float (&acc)[8],
                                                    const float (&bias8)[8]);

template <typename T, typename TC>
__device__ __forceinline__ void gemm_epilogue8(const GemmParams& p, int split, int m, int n8, const float (&acc)[8],
                                               const float (&bias8)[8]) {
    if (m >= p.M || n8 >= p.N) return;
    const bool fast = p.c_vec8_ok && n8 + 7 < p.N && p.splitk == 1 && sizeof(T) == 2;
    if (!fast) {
        gemm_epilogue4<T, TC>(p, split, m, n8, f32x4{acc[0], acc[1], acc[2], acc[3]});
        gemm_epilogue4<T, TC>(p, split, m, n8 + 4, f32x4{acc[4], acc[5], acc[6], acc[7]});
        return;
    }
    gemm_epilogue8_fast<T, TC>(p, m, n8, acc, bias8);
}

template <typename T, typename TC>
__device__ __forceinline__ void gemm_epilogue8_fast(const GemmParams& p, int m, int n8, const float (&acc)[8],
                                                    const float (&bias8)[8]) {
    if (m >= p.M) return;
    TC* __restrict__ C = reinterpret_cast<TC*>(p.C);
    const __bf16* __restrict__ aux_in = reinterpret_cast<const __bf16*>(p.aux_in);
    __bf16* __restrict__ aux_out = reinterpret_cast<__bf16*>(p.aux_out);
    const __bf16* __restrict__ res = reinterpret_cast<const __bf16*>(p.residual);
    long orow = m;
    if (p.grp_rows > 0) orow = (long)(m / p.grp_rows) * p.grp_stride + p.grp_off + m % p.grp_rows;
    const long rrow = p.res_period > 0 ? (long)(p.res_off + m % p.res_period) : orow;
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = acc[e] * p.alpha + bias8[e];
    if (p.act == ACT_GELU) {
        if (aux_out) {
            bf16x8 o;
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = (__bf16)v[e];
            *reinterpret_cast<bf16x8*>(aux_out + orow * p.ld_aux + n8) = o;
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = gelu_erf(v[e]);
    } else if (p.act == ACT_RELU) {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = fmaxf(v[e], 0.f);
    } else if (p.act == ACT_DGELU || p.act == ACT_DRELU) {
        const bf16x8 a = *reinterpret_cast<const bf16x8*>(aux_in + orow * p.ld_aux + n8);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float h = (float)a[e];
            v[e] *= (p.act == ACT_DGELU) ? gelu_erf_grad(h) : (h > 0.f ? 1.f : 0.f);
        }
    }
    if (res) {
        const bf16x8 a = *reinterpret_cast<const bf16x8*>(res + rrow * p.ldr + n8);
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] += (float)a[e];
    }
    TC* dst = C + orow * p.ldc + n8;
    if (sizeof(TC) == 4) {
        f32x4 o0 = {v[0], v[1], v[2], v[3]}, o1 = {v[4], v[5], v[6], v[7]};
        if (p.accumulate) {
            o0 += *reinterpret_cast<const f32x4*>(dst);
            o1 += *reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>(dst) + 4);
        }
        *reinterpret_cast<f32x4*>(dst) = o0;
        *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(dst) + 4) = o1;
    } else {
        bf16x8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = (__bf16)v[e];
        *reinterpret_cast<bf16x8*>(dst) = o;
    }
}

// XCD-aware (bijective) remap of a linear id in [0, n): workgroup i runs on XCD i & 7 (grids are
// multiples of 8), so XCD x gets the x-th contiguous eighth of the ids.
__device__ __forceinline__ int gemm_xcd_remap(int bid, int n) {
    const int q = n >> 3, r = n & 7, x = bid & 7;
    return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (bid >> 3);
}

// Tile order in groups of GROUP_M row panels: consecutive ids share an A row panel / B column panel.
__device__ __forceinline__ void gemm_grouped_tile(int bid, int tiles_m, int tiles_n, int& tile_m, int& tile_n) {
    constexpr int GROUP_M = 8;
    const int per_group = GROUP_M * tiles_n;
    const int grp = bid / per_group;
    const int first_m = grp * GROUP_M;
    const int gsize = min(tiles_m - first_m, GROUP_M);
    const int in_grp = bid - grp * per_group;
    tile_m = first_m + in_grp % gsize;
    tile_n = in_grp / gsize;
}

// Both: workgroups that share a panel run on one XCD (private L2) back to back.
__device__ __forceinline__ void gemm_tile_of(int bid, int tiles_m, int tiles_n, int& tile_m, int& tile_n) {
    gemm_grouped_tile(gemm_xcd_remap(bid, tiles_m * tiles_n), tiles_m, tiles_n, tile_m, tile_n);
}

}  // namespace vitk
