// Building blocks shared by the fused attention forward and backward kernels.
//
// Score model (one closed form for all reference variants):
//   raw[i][j] = Q_i.K_j + Q_i.Pk[i-j] + K_j.Pq[i-j]          (bias terms only when tables given)
//   s[i][j]   = scale * raw[i][j],  masked out (probability exactly 0) when |i-j| > w  (w >= 0)
// - plain / LAI attention: AttentionLAI.forward (timm_classes/timm_locality_aware_transformer.py:45-79),
//   mask = get_attn_mask (:11-21), fill -1e9 then softmax == probability 0 (the diagonal is
//   always kept, so no row is empty).
// - disentangled attention: DisentangledSelfAttention.forward + disentangled_attention_bias
//   (disentangled_transformer/disentangled_transformer.py:301-426), LAI twin :851-998.  c2p and
//   p2c both index the projected relative table at delta = i - j (+ span); the host hands the
//   kernel the 2N-1 reachable rows ("band"), row r <-> i - j = r - (N-1).
//
// All tiles are 16x16 MFMA tiles (common.h fragment convention).  A 16(query) x 16(key) score
// tile only touches relative rows delta0 .. delta0+30, delta0 = i0 - j0 - 15, so the two bias
// products are two 16-row MFMA tiles each (X^T[t][ii] = Pk[delta0+t].Q_ii, Y^T[t][jj] =
// Pq[delta0+t].K_jj); they go through a per-wave LDS scratch, written row-wise and read back
// along the diagonals t = ii - jj + 15.  Nothing of size N x N ever leaves the CU.
#pragma once
#include <type_traits>
#include "common.h"

namespace vitk {

constexpr int SCR_LD = 36;                 // floats per scratch row (32 used + pad, keeps 16-B alignment)
constexpr int SCR_FLOATS = 16 * SCR_LD;    // one 16 x 32 skew buffer

// Zero-padded bf16 copies of a 16 x 16 dS tile for the skewed (Toeplitz) fragments of the
// relative-table gradients.  Sx[t][i] = dS[i][i + 15 - t] and Sy[t][j] = dS[j + t - 15][j] run
// along diagonals; with the tile embedded in zeros every out-of-range element reads 0 and the
// 8 elements of a fragment sit at a constant stride, so a fragment is 8 plain 16-bit LDS reads
// at immediate offsets -- no clamp / compare / select per element (those ~10 VALU per element
// were 60 % of the table-gradient kernel's time).
//   X copy: 16 rows x 48 columns, dS[i][j] at X[i * 48 + 16 + j]     (columns -16 .. 31)
//   Y copy: 47 rows x 17 columns, dS[i][j] at Y[(15 + i) * 17 + j]   (rows -15 .. 31)
constexpr int DSX_LD = 48;
constexpr int DSX_ELEMS = 16 * DSX_LD;     // 768
constexpr int DSY_LD = 17;
constexpr int DSY_ELEMS = 47 * DSY_LD;     // 799
constexpr int DSXY_ELEMS = 1568;           // X copy + Y copy of one tile, rounded up to 4 bytes

// DW: width of an LDS row image in elements.  32 * DKB by default; the whole-head kernels stage a head of 80 as
// DW = 80 (DKB = 3): the last K block is half there, row_frag zeroes the other half in registers, and the LDS the
// 16 missing columns would take is what lets the 197-token images of such a head fit (attention_band.hip).
template <typename T, int DKB, int DW = 32 * DKB> struct ACfg {
    static_assert(DW <= 32 * DKB && DW > 32 * (DKB - 1) && DW % 16 == 0, "row image width vs K blocks");
    static constexpr int EPC = 16 / sizeof(T);
    static constexpr int DP = DW;                       // head dim padded to the staged width
    static constexpr int ASTR = DP + 16 / sizeof(T);    // LDS row stride (elements): +16 B de-conflicts banks
    static constexpr int CPR = DP / EPC;                // 16-byte chunks per row
};

// order LDS traffic of ONE wave (write -> read by other lanes of the same wave)
__device__ __forceinline__ void wave_lds_sync() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }

// Stage `nrows` rows x DP columns into LDS.  Row r of the tile is global row (grow0 + r) of a
// matrix with leading dimension ld; rows outside [lo, hi) and columns >= dh are zero-filled.
template <typename T, int DKB, int DW = 32 * DKB>
__device__ __forceinline__ void stage_rows(T* lds, const T* __restrict__ g, long ld, int grow0, int nrows,
                                           int lo, int hi, int dh, int tid, int nthreads) {
    using C = ACfg<T, DKB, DW>;
    const int total = nrows * C::CPR;
    for (int c = tid; c < total; c += nthreads) {
        const int r = c / C::CPR, ch = c - r * C::CPR;
        const int d = ch * C::EPC;
        const int gr = grow0 + r;
        u32x4 v = {0, 0, 0, 0};
        if (gr >= lo && gr < hi && d < dh) v = *reinterpret_cast<const u32x4*>(g + (long)gr * ld + d);
        *reinterpret_cast<u32x4*>(lds + r * C::ASTR + d) = v;
    }
}

// Row fragment (K-contiguous): tile rows [row_base, +16), head-dim block kb.
template <typename T, int DKB, int DW = 32 * DKB>
__device__ __forceinline__ void row_frag(Frag<T>& f, const T* lds, int row_base, int kb, int lane) {
    frag_load(f, lds + (row_base + (lane & 15)) * ACfg<T, DKB, DW>::ASTR + 32 * kb + 8 * (lane >> 4));
    if (DW < 32 * DKB && 32 * kb + 8 * (lane >> 4) >= DW) frag_zero(f);   // past the image: pad / the next row, not data
}

// Row fragment straight from global memory (the block's own 16 rows, read once).
template <typename T>
__device__ __forceinline__ void row_frag_global(Frag<T>& f, const T* __restrict__ g, long ld, int row, int nrows,
                                                int dh, int kb, int lane) {
    const int d = 32 * kb + 8 * (lane >> 4);
    if (row < nrows && d < dh) frag_load(f, g + (long)row * ld + d);
    else frag_zero(f);
}

// Column fragment (K-major read of a [row][col] image): element e < 4 <- img[r0 + e][col0 + (l&15)],
// e >= 4 <- img[r1 + e - 4][col0 + (l&15)].  r0 / r1 may depend on the lane's 16-lane group only.
// bf16: two ds_read_b64_tr_b16 (EXEC must be all ones, addresses 8-byte aligned: col0 % 4 == 0).
__device__ __forceinline__ void col_frag(Frag<__bf16>& f, const __bf16* img, int ld, int r0, int r1, int col0,
                                         int lane) {
    const int i = lane & 15, q = i >> 2, p = i & 3;
    const __bf16* a0 = img + (r0 + q) * ld + col0 + 4 * p;
    const __bf16* a1 = img + (r1 + q) * ld + col0 + 4 * p;
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(uintptr_t)a0);
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(uintptr_t)a1);
    const s16x8 r = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    f.v = __builtin_bit_cast(bf16x8, r);
}
__device__ __forceinline__ void col_frag(Frag<float>& f, const float* img, int ld, int r0, int r1, int col0,
                                         int lane) {
    const int i = lane & 15;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        f.v[e] = img[(r0 + e) * ld + col0 + i];
        f.v[4 + e] = img[(r1 + e) * ld + col0 + i];
    }
}

// Two accumulator tiles (rows 4g+r of K-block halves 0 and 1) -> the B/A fragment of the next
// product, k order: e < 4 -> row 4g+e of the first tile, e >= 4 -> row 4g+e-4 of the second
// (pair with col_frag(r0 = base + 4g, r1 = base + 16 + 4g)).
__device__ __forceinline__ void acc_to_frag(Frag<__bf16>& f, const f32x4& a, const f32x4& b) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        f.v[e] = (__bf16)a[e];
        f.v[4 + e] = (__bf16)b[e];
    }
}
__device__ __forceinline__ void acc_to_frag(Frag<float>& f, const f32x4& a, const f32x4& b) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        f.v[e] = a[e];
        f.v[4 + e] = b[e];
    }
}

// Sum over the 16 lanes of a DPP row (the lanes that share lane >> 4), in every lane of the row: two quad
// permutes and two row rotations, all on the VALU (a __shfl_xor would go through the LDS pipe).
__device__ inline float row16_sum(float t) {
    auto dpp = [](float x, auto ctrl) {
        return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), decltype(ctrl)::value, 0xf, 0xf, true));
    };
    t += dpp(t, std::integral_constant<int, 0xB1>{});    // quad_perm [1,0,3,2]
    t += dpp(t, std::integral_constant<int, 0x4E>{});    // quad_perm [2,3,0,1]
    t += dpp(t, std::integral_constant<int, 0x124>{});   // row_ror:4
    t += dpp(t, std::integral_constant<int, 0x128>{});   // row_ror:8
    return t;
}

// Parameters of the whole-head kernels (attention_band.hip, attention_lai.hip)
struct BandParams {
    const void *q, *k, *v;
    long ld;
    const void *pk, *pq;
    long ld_p;
    void* out;              // forward: written; backward: read
    long ldo;
    float* lse;             // forward: written; backward: read
    const void* dout;
    float* delta;           // backward: delta_i = dO_i . O_i, WRITTEN here (read by attn_bwd_rel_kernel)
    void *dq, *dk, *dv;
    long ld_d;
    int B, H, N, dh;
    float scale;
    int w;
    __bf16* ds_ws;          // backward: dS tiles, written by the dK / dV kernel, read by the dQ kernel and the table-gradient kernel
    DropParams drop;        // dropout on the probabilities, element ((b*H + h)*N + i)*N + j of the stream (attention.hip)
    float *dpk, *dpq;       // backward, with tables: fp32 table gradients [2N-1][ld_dp], accumulated into (attention_lai.hip;
    long ld_dp;             // the whole-head kernels of attention_band.hip leave them to attn_bwd_relds_kernel)
    float* dbias;           // backward, optional: [3 * H * dh] fp32 (q | k | v), += the column sums of dQ / dK / dV over all
                            // tokens -- the bias gradient of the fused qkv projection, without a pass over dqkv
};

// Raw (unscaled) score tile for 16 queries x 16 keys.
//   qf / kf : row fragments of the 16 queries / 16 keys (DKB head-dim blocks each)
//   TRANS   : true  -> result[r] = raw[query l&15][key 4g+r]   (lane owns a query)
//             false -> result[r] = raw[query 4g+r][key l&15]   (lane owns a key)
//   pk / pq : staged band tiles; `rel` = band row of delta0 = i0 - j0 - 15 for this tile pair
//   xs / ys : this wave's skew scratch (SCR_FLOATS each)
template <typename T, int DKB, bool BIAS, bool TRANS, int DW = 32 * DKB>
__device__ __forceinline__ f32x4 score_tile(const Frag<T> (&qf)[DKB], const Frag<T> (&kf)[DKB], const T* pk,
                                            const T* pq, int rel, float* xs, float* ys, int lane) {
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kb = 0; kb < DKB; ++kb) s = TRANS ? mma(kf[kb], qf[kb], s) : mma(qf[kb], kf[kb], s);
    if (BIAS) {
        const int g = lane >> 4, li = lane & 15;
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            f32x4 x = {0.f, 0.f, 0.f, 0.f}, y = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int kb = 0; kb < DKB; ++kb) {
                Frag<T> a;
                row_frag<T, DKB, DW>(a, pk, rel + 16 * u, kb, lane);
                x = mma(a, qf[kb], x);   // X^T[t = 16u + 4g + r][ii = l&15]
                row_frag<T, DKB, DW>(a, pq, rel + 16 * u, kb, lane);
                y = mma(a, kf[kb], y);   // Y^T[t][jj = l&15]
            }
            *reinterpret_cast<f32x4*>(xs + li * SCR_LD + 16 * u + 4 * g) = x;
            *reinterpret_cast<f32x4*>(ys + li * SCR_LD + 16 * u + 4 * g) = y;
        }
        wave_lds_sync();
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int ii = TRANS ? li : 4 * g + r;
            const int jj = TRANS ? 4 * g + r : li;
            const int t = ii - jj + 15;
            s[r] += xs[ii * SCR_LD + t] + ys[jj * SCR_LD + t];
        }
        wave_lds_sync();  // reads done before the next tile overwrites the scratch
    }
    return s;
}

}  // namespace vitk
