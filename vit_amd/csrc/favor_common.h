// FAVOR+ kernels: parameter block and device helpers shared by favor.hip (every shape) and favor_fast.hip
// (up to 64 random features: per-(b, h) operands staged in LDS).  See favor.hip for the algorithm.
#pragma once
#include "attn_common.h"

namespace vitk {

enum { FAVOR_SOFTMAX = 0, FAVOR_RELU = 1 };

struct FavorParams {
    const void *q, *k, *v;
    long ld;
    const void* w;          // [m][dh], T
    void* kvt;              // [B][H][dh][mp]  T   (KV transposed: rows d)
    void* kv;               // [B][H][mp][dhp] T   (KV: rows m)
    float* ksum;            // [B][H][mp]
    unsigned long long* kmax;   // packed (ordered float bits << 32 | flat index); kmax_slots cells, the maximum is over all
    int kmax_slots;             // 1 (favor.hip) or KMAX_SLOTS (favor_fast.hip: one cell took ~10 ns per visiting wave)
    void* out;
    long ldo;
    const void* dout;
    float* dkvt;            // [B][H][dh][mp]  fp32 (backward: d KV transposed)
    float* dksum;           // [B][H][mp]
    void *dkvt_t, *dkv_t;   // T copies of dKV in both layouts (written by the q-side backward)
    float *kv32, *dkv32;    // fp32 [B][H][mp][dhp] copies: the two cancelling products of the backward use them
    float* esum;            // partial sums of the key-side exp gradients: [nesum] (one per (b, h); per (l, h) for the
                            // position tables), added up by favor_fix -- one shared address would serialise ~25k atomics
    int nesum;
    void *dq, *dk, *dv;
    long ld_d;
    // position features (DisentangledSelfAttentionPerformer): q' += pq[l][h][.], k' += pk[l][h][.]  (fp32 [L][H][mp],
    // the same for every b); the backward leaves d(q' + pq) / d(k' + pk) per b in dnq / dnk (fp32 [B][L][H][mp])
    const float *pq, *pk;
    float *dnq, *dnk;
    int B, H, L, dh, m, mp, dhp;
    int groups;             // favor_fast.hip: 64-token groups per workgroup of the kernels that walk tokens
    int kind;
    float ratio, eps, diag_c;
};

__device__ __forceinline__ unsigned int ordered_bits(float f) {
    const unsigned int u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float unordered_bits(unsigned int k) {
    const unsigned int u = (k & 0x80000000u) ? (k & 0x7fffffffu) : ~k;
    return __uint_as_float(u);
}

template <typename T> __device__ __forceinline__ f32x4 load4(const T* p);
template <> __device__ __forceinline__ f32x4 load4<float>(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
template <> __device__ __forceinline__ f32x4 load4<__bf16>(const __bf16* p) {
    const bf16x4 a = *reinterpret_cast<const bf16x4*>(p);
    return f32x4{(float)a[0], (float)a[1], (float)a[2], (float)a[3]};
}
template <typename T> __device__ __forceinline__ void store4(T* p, const f32x4& v);
template <> __device__ __forceinline__ void store4<float>(float* p, const f32x4& v) { *reinterpret_cast<f32x4*>(p) = v; }
template <> __device__ __forceinline__ void store4<__bf16>(__bf16* p, const f32x4& v) {
    *reinterpret_cast<bf16x4*>(p) = bf16x4{(__bf16)v[0], (__bf16)v[1], (__bf16)v[2], (__bf16)v[3]};
}

// Fragment whose 8 k-slots are two runs of 4 consecutive elements (p0[0..3], p1[0..3]): the k order
// of acc_to_frag (tile a rows 4g..4g+3, tile b rows 4g..4g+3).
template <typename T> __device__ __forceinline__ void frag_load_split(Frag<T>& f, const T* p0, const T* p1, bool ok1) {
    const f32x4 a = load4<T>(p0);
    f32x4 b = {0.f, 0.f, 0.f, 0.f};
    if (ok1) b = load4<T>(p1);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        frag_set_f(f, e, a[e]);
        frag_set_f(f, 4 + e, b[e]);
    }
}

// bf16 hi / lo split of fp32 values: x ~ hi + lo to ~16 mantissa bits.  The backward's two
// cancelling products (KV . dnum + dden ksum, dKV . v + dksum) run as hi.hi + hi.lo + lo.hi so the
// bf16 path keeps fp32-like accuracy exactly where rounding would be amplified.
__device__ __forceinline__ void split_set(Frag<__bf16>& hi, Frag<__bf16>& lo, int e, float x) {
    const __bf16 h = (__bf16)x;
    hi.v[e] = h;
    lo.v[e] = (__bf16)(x - (float)h);
}
__device__ __forceinline__ void acc_to_frag_hilo(Frag<__bf16>& hi, Frag<__bf16>& lo, const f32x4& a, const f32x4& b) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        split_set(hi, lo, e, a[e]);
        split_set(hi, lo, 4 + e, b[e]);
    }
}

// feature value and its derivative factor d val / d dd (without the max-routing term)
__device__ __forceinline__ float feat(int kind, float dd, float diag, float stab, float ratio, float eps) {
    if (kind == FAVOR_SOFTMAX) return ratio * (expf(dd - diag - stab) + eps);
    return fmaxf(ratio * dd, 0.f) + eps;
}
__device__ __forceinline__ float feat_grad(int kind, float dd, float diag, float stab, float ratio) {
    if (kind == FAVOR_SOFTMAX) return ratio * expf(dd - diag - stab);
    return dd > 0.f ? ratio : 0.f;
}

// dd^T tiles for the 16 tokens of a fragment set: acc[mt][r] = dd[token li][feature 16mt + 4g + r]
template <typename T, int DKB, int MT>
__device__ __forceinline__ void ddT_tiles(f32x4 (&acc)[MT], const Frag<T> (&xf)[DKB], const T* __restrict__ w, int m,
                                          int dh, int lane) {
    const int g = lane >> 4, li = lane & 15;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        f32x4 s = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kb = 0; kb < DKB; ++kb) {
            Frag<T> wf;
            row_frag_global<T>(wf, w, dh, 16 * mt + li, m, dh, kb, lane);
            s = mma(wf, xf[kb], s);
        }
        acc[mt] = s;
    }
}

// per-token sum of the data row (lane = token): sum of the lane's fragment elements over g
template <typename T, int DKB> __device__ __forceinline__ float row_sum(const Frag<T> (&xf)[DKB]) {
    float s = 0.f;
#pragma unroll
    for (int kb = 0; kb < DKB; ++kb)
#pragma unroll
        for (int e = 0; e < 8; ++e) s += frag_get_f(xf[kb], e);
    s += __shfl_xor(s, 16, 64);
    s += __shfl_xor(s, 32, 64);
    return s;
}

// the same with the kernel kind known at compile time (favor_fast.hip): straight-line code, no per-element branch
// bf16 data: the hardware exp2 on x * log2(e) (2 instructions; the argument's rounding costs |x| * 6e-8 relative,
// far below bf16's 4e-3) -- expf proper is 15 instructions and was 40 % of the VALU work of the backward.  fp32 data
// keeps expf: its parity bar is 1e-5.
template <int KIND, typename T> __device__ __forceinline__ float feat_exp(float dd, float diag, float stab) {
    if constexpr (KIND != FAVOR_SOFTMAX) return 0.f;
    else if constexpr (sizeof(T) == 2) return __builtin_amdgcn_exp2f((dd - diag - stab) * 1.44269504088896341f);
    else return expf(dd - diag - stab);
}
template <int KIND> __device__ __forceinline__ float feat_t(float dd, float ex, float ratio, float eps) {
    if constexpr (KIND == FAVOR_SOFTMAX) return ratio * (ex + eps);
    else return fmaxf(ratio * dd, 0.f) + eps;
}
template <int KIND> __device__ __forceinline__ float feat_grad_t(float dd, float ex, float ratio) {
    if constexpr (KIND == FAVOR_SOFTMAX) return ratio * ex;
    else return dd > 0.f ? ratio : 0.f;
}

// The key maximum is kept in up to KMAX_SLOTS cells (the first 256 bytes of the workspace), each the maximum of the
// workgroups that hash to it: every reader takes the maximum over the cells.  All lanes of the wave get the result.
constexpr int KMAX_SLOTS = 32;
__device__ __forceinline__ unsigned long long kmax_all(const unsigned long long* kmax, int nslots, int lane) {
    unsigned long long v = lane < nslots ? kmax[lane] : 0ull;
    if (nslots > 1) {
#pragma unroll
        for (int o = 16; o > 0; o >>= 1) {
            const unsigned long long other = __shfl_xor(v, o, 64);
            v = other > v ? other : v;
        }
    }
    return __shfl(v, 0, 64);
}

// the global-max element of the key features receives minus the whole tensor's exp-gradient sum
template <typename T>
__global__ void favor_fix_kernel(const FavorParams p) {
    const unsigned long long key = kmax_all(p.kmax, p.kmax_slots, threadIdx.x);      // launched with one wave
    const unsigned long long idx = 0xffffffffull - (key & 0xffffffffull);
    const int mi = idx % p.m;
    const unsigned long long t = idx / p.m;
    const int h = t % p.H;
    const long bl = t / p.H;
    float s = 0.f;
    for (int i = threadIdx.x; i < p.nesum; i += 64) s += p.esum[i];      // launched with one wave
    s = wave_sum(s);
    const T* W = reinterpret_cast<const T*>(p.w);
    T* DK = reinterpret_cast<T*>(p.dk) + bl * p.ld_d + h * p.dh;
    for (int d = threadIdx.x; d < p.dh; d += blockDim.x)
        DK[d] = from_f32<T>(to_f32<T>(DK[d]) - s * to_f32<T>(W[(long)mi * p.dh + d]));
}

// favor_fast.hip: launches the forward / backward for up to 64 features (mp == 16 or 64); false when the shape is not its
bool favor_fast_launch(const FavorParams& p, int dtype, bool bwd, hipStream_t st);

}  // namespace vitk
