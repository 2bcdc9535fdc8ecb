// Shared device helpers for the vitk kernels (gfx950 / CDNA4 only).
//
// Every MFMA kernel in this library is templated on the element type T in {float, __bf16} and
// shares one fragment convention, so the fp32 parity path and the bf16 speed path run the SAME
// code with a different instruction underneath:
//
//   Frag<T>  = 8 K-consecutive elements of ONE row, held by lane l for
//              row (l & 15) and k = 8*(l >> 4) + e, e = 0..7, of a 32-deep K block.
//   mma<T>   = D[4*(l>>4)+j][l&15] += sum_k A[.][k] * B[k][.]   (16x16 tile, K = 32)
//              bf16: one v_mfma_f32_16x16x32_bf16
//              f32 : eight v_mfma_f32_16x16x4_f32, step e consuming element e of both
//                    fragments (lane group g then covers k = 8g+e: a permutation of the K
//                    order inside the block, identical for A and B, hence the same sum).
//              The f32 form is an exact k-ordered fmaf chain (MI355X guide, "FP32-input MFMA").
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace vitk {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(8))) short s16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
typedef __attribute__((ext_vector_type(2))) uint32_t u32x2;

enum { VITK_F32 = 0, VITK_BF16 = 1 };

template <typename T> struct Frag;
template <> struct Frag<__bf16> {
    bf16x8 v;
};
template <> struct Frag<float> {
    float v[8];
};

__device__ __forceinline__ void frag_zero(Frag<__bf16>& f) {
#pragma unroll
    for (int e = 0; e < 8; ++e) f.v[e] = (__bf16)0.0f;
}
__device__ __forceinline__ void frag_zero(Frag<float>& f) {
#pragma unroll
    for (int e = 0; e < 8; ++e) f.v[e] = 0.0f;
}

__device__ __forceinline__ void frag_set_f(Frag<__bf16>& f, int e, float v) { f.v[e] = (__bf16)v; }
__device__ __forceinline__ void frag_set_f(Frag<float>& f, int e, float v) { f.v[e] = v; }
__device__ __forceinline__ float frag_get_f(const Frag<__bf16>& f, int e) { return (float)f.v[e]; }
__device__ __forceinline__ float frag_get_f(const Frag<float>& f, int e) { return f.v[e]; }

// D(16x16) += A(16x32) * B(32x16); a = A rows on lanes, b = B columns on lanes.
__device__ __forceinline__ f32x4 mma(const Frag<__bf16>& a, const Frag<__bf16>& b, f32x4 acc) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.v, b.v, acc, 0, 0, 0);
}
__device__ __forceinline__ f32x4 mma(const Frag<float>& a, const Frag<float>& b, f32x4 acc) {
#pragma unroll
    for (int e = 0; e < 8; ++e) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.v[e], b.v[e], acc, 0, 0, 0);
    return acc;
}

// 8 K-contiguous elements from a (16-byte aligned) pointer, LDS or global.
__device__ __forceinline__ void frag_load(Frag<__bf16>& f, const __bf16* p) {
    f.v = *reinterpret_cast<const bf16x8*>(p);
}
__device__ __forceinline__ void frag_load(Frag<float>& f, const float* p) {
    f32x4 a = *reinterpret_cast<const f32x4*>(p);
    f32x4 b = *reinterpret_cast<const f32x4*>(p + 4);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        f.v[e] = a[e];
        f.v[4 + e] = b[e];
    }
}

// Fragment from a K-MAJOR LDS image img[k][col] (row stride `ld` elements): lane l gets
// img[8*(l>>4)+e][col0 + (l&15)], e = 0..7.  bf16 uses the hardware transpose read
// (ds_read_b64_tr_b16: lane 4q+p of a 16-lane group supplies the address of row q, columns
// 4p..4p+3; lane i receives column i of the 4 rows) and needs EXEC all ones, 8-byte aligned
// addresses.  `base` points at img[k0][col0]; k0, col0*sizeof(T) multiples of 8 bytes.
__device__ __forceinline__ void frag_load_kmajor(Frag<__bf16>& f, const __bf16* base, int ld, int lane) {
    const int g = lane >> 4, i = lane & 15;
    const int q = i >> 2, p = i & 3;
    const __bf16* a0 = base + (8 * g + q) * ld + 4 * p;
    const __bf16* a1 = a0 + 4 * ld;
    s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
        (__attribute__((address_space(3))) s16x4*)(uintptr_t)a0);
    s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
        (__attribute__((address_space(3))) s16x4*)(uintptr_t)a1);
    s16x8 r = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    f.v = __builtin_bit_cast(bf16x8, r);
}
__device__ __forceinline__ void frag_load_kmajor(Frag<float>& f, const float* base, int ld, int lane) {
    const int g = lane >> 4, i = lane & 15;
#pragma unroll
    for (int e = 0; e < 8; ++e) f.v[e] = base[(8 * g + e) * ld + i];
}

template <typename T> __device__ __forceinline__ float to_f32(T x) { return (float)x; }
template <typename T> __device__ __forceinline__ T from_f32(float x) { return (T)x; }

__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }
__device__ __forceinline__ float gelu_erf_grad(float x) {
    const float cdf = 0.5f * (1.0f + erff(x * 0.70710678118654752440f));
    const float pdf = 0.39894228040143267794f * __expf(-0.5f * x * x);
    return cdf + x * pdf;
}

// erf-GELU for the bf16 epilogues (~16 VALU per element instead of ~35-50 of erff / expf: the
// fused fc1 / dfc2 epilogues are VALU-bound).  erf by Abramowitz-Stegun 7.1.26 (|error| < 1.5e-7,
// three orders below bf16 rounding); exp(-x^2/2) is shared by erf(x / sqrt 2) and the density.
// The fp32 parity path keeps erff / expf (gelu_erf, gelu_erf_grad above).
__device__ __forceinline__ void gelu_fast_parts(float x, float& erfv, float& e) {
    const float z = fabsf(x) * 0.70710678118654752440f;
    const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, z, 1.0f));
    e = __builtin_amdgcn_exp2f(x * x * -0.72134752044448170368f);   // exp(-x^2 / 2)
    float pl = fmaf(1.061405429f, t, -1.453152027f);
    pl = fmaf(pl, t, 1.421413741f);
    pl = fmaf(pl, t, -0.284496736f);
    pl = fmaf(pl, t, 0.254829592f);
    erfv = copysignf(fmaf(-pl * t, e, 1.0f), x);
}
__device__ __forceinline__ float gelu_fast(float x) {
    float erfv, e;
    gelu_fast_parts(x, erfv, e);
    const float hx = 0.5f * x;
    return fmaf(hx, erfv, hx);
}
__device__ __forceinline__ float gelu_grad_fast(float x) {
    float erfv, e;
    gelu_fast_parts(x, erfv, e);
    return fmaf(0.5f, erfv, 0.5f) + x * (0.39894228040143267794f * e);
}

// erf-GELU without transcendentals, two elements per instruction (v_pk_fma_f32): round 3's form of the bf16 epilogues.
//   gelu(x)  = x / 2 + E(x),   E(x) = x (Phi(x) - 1/2)   even, = x^2 P(z) on |x| <= 4,  z = x^2 / 8 - 1 in [-1, 1]
//   gelu'(x) = 1 / 2 + x R(z)                                                                (R fitted on its own)
// P, R: degree-8 least-squares fits on Chebyshev nodes (tools/fit_gelu.py), evaluated by Horner in z; |x| > 4 continues the
// value at 4 with slope 1 / 0 (gelu) or holds it (gelu').  Max |error| in fp32 Horner: 9.5e-5 (gelu, in the tails; 3e-5
// inside), 6e-4 (gelu', tails; 1.1e-4 inside) -- an order below the 2^-9 rounding of the bf16 value it is stored as, and
// 10.5 VALU per element where the exp2 / rcp form above costs 12 + two quarter-rate ops: the fc1 forward and fc2 dX GEMMs
// spent a fifth of their time in this arithmetic with the matrix cores idle.
typedef __attribute__((ext_vector_type(2))) float f32x2;
__device__ __forceinline__ f32x2 gelu_clamp2(f32x2 x) {
    return f32x2{__builtin_amdgcn_fmed3f(x[0], -4.f, 4.f), __builtin_amdgcn_fmed3f(x[1], -4.f, 4.f)};
}
__device__ __forceinline__ f32x2 gelu_poly2(f32x2 x) {
    const f32x2 xc = gelu_clamp2(x);
    const f32x2 z = xc * (xc * 0.125f) - 1.0f;
    f32x2 p = {1.605170928e-03f, 1.605170928e-03f};
    p = p * z + -3.991263335e-03f;
    p = p * z + 5.828486749e-03f;
    p = p * z + -1.154668287e-02f;
    p = p * z + 2.266773498e-02f;
    p = p * z + -3.713346277e-02f;
    p = p * z + 5.591522291e-02f;
    p = p * z + -8.430131655e-02f;
    p = p * z + 1.759501642e-01f;
    const f32x2 s = {x[0] + (__builtin_fabsf(x[0]) - __builtin_fabsf(xc[0])), x[1] + (__builtin_fabsf(x[1]) - __builtin_fabsf(xc[1]))};
    return s * 0.5f + (xc * xc) * p;
}
__device__ __forceinline__ f32x2 gelu_grad_poly2(f32x2 x) {
    const f32x2 xc = gelu_clamp2(x);
    const f32x2 z = xc * (xc * 0.125f) - 1.0f;
    f32x2 r = {1.995524858e-02f, 1.995524858e-02f};
    r = r * z + -4.264639257e-02f;
    r = r * z + 4.136232702e-02f;
    r = r * z + -6.191498239e-02f;
    r = r * z + 1.028313072e-01f;
    r = r * z + -1.185188937e-01f;
    r = r * z + 1.140870964e-01f;
    r = r * z + -1.132650675e-01f;
    r = r * z + 1.832627819e-01f;
    return xc * r + 0.5f;
}

// ---------------------------------------------------------------------------------------- dropout
// Counter-based keep mask: element `idx` of dropout stream (seed, offset) is kept iff a 32-bit hash of
// (seed, offset, idx) is >= thresh = round(p * 2^32).  Stateless, so a backward kernel regenerates the mask
// of its forward from the same three numbers and nothing is stored (the reference's nn.Dropout draws from
// torch's Philox stream, which a fused kernel cannot reproduce bit for bit: parity at p > 0 is checked with
// this mask exported -- vitk_dropout_mask -- and fed to the oracle).  Two rounds of the lowbias32 mixer.
struct DropParams {
    uint32_t seed_lo, seed_hi, off_lo, off_hi;
    uint32_t thresh;     // 0: dropout off
    float scale;         // 1 / (1 - p)
};
__device__ __forceinline__ uint32_t drop_mix(uint32_t x) {
    x ^= x >> 16;
    x *= 0x7feb352du;
    x ^= x >> 15;
    x *= 0x846ca68bu;
    x ^= x >> 16;
    return x;
}
__device__ __forceinline__ bool drop_keep(const DropParams& d, uint64_t idx) {
    uint32_t x = drop_mix((uint32_t)idx * 0x9E3779B1u + d.seed_lo);
    x = drop_mix(x + (uint32_t)(idx >> 32) * 0x85EBCA77u + d.seed_hi + d.off_lo * 0xC2B2AE3Du + d.off_hi * 0x27D4EB2Fu);
    return x >= d.thresh;
}
// multiplier of a value under dropout: 0 or 1 / (1 - p)
__device__ __forceinline__ float drop_mul(const DropParams& d, uint64_t idx) { return drop_keep(d, idx) ? d.scale : 0.f; }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

}  // namespace vitk

// ---- host side -------------------------------------------------------------------------
#define VITK_OK 0
#define VITK_ERR_ARG 1      // bad argument (shape / alignment / null pointer)
#define VITK_ERR_LAUNCH 2   // HIP launch error
#define VITK_ERR_WORKSPACE 3

extern "C" void vitk_set_error(const char* fmt, ...);

#define VITK_CHECK_ARG(cond, ...)        \
    do {                                 \
        if (!(cond)) {                   \
            vitk_set_error(__VA_ARGS__); \
            return VITK_ERR_ARG;         \
        }                                \
    } while (0)

// host side of DropParams: p in [0, 1)
static inline vitk::DropParams vitk_drop_params(float p, unsigned long long seed, unsigned long long offset) {
    vitk::DropParams d;
    d.seed_lo = (uint32_t)seed; d.seed_hi = (uint32_t)(seed >> 32);
    d.off_lo = (uint32_t)offset; d.off_hi = (uint32_t)(offset >> 32);
    double t = (double)p * 4294967296.0;
    d.thresh = p <= 0.f ? 0u : (t >= 4294967295.0 ? 4294967295u : (uint32_t)(t + 0.5));
    d.scale = p <= 0.f ? 1.f : 1.f / (1.f - p);
    return d;
}

// Zero-fill as a kernel of the library's own (runtime.hip) instead of hipMemsetAsync: an ordinary kernel node
// when the stream is being captured into a HIP graph.  rows x row_bytes with a row pitch (pitch == row_bytes: flat).
extern "C" int vitk_zero_async(void* ptr, size_t pitch, size_t row_bytes, size_t rows, hipStream_t stream);

// Opt a kernel in to `bytes` of dynamic LDS (> 64 KiB needs hipFuncAttributeMaxDynamicSharedMemorySize).  The attribute
// is PER DEVICE: it is tracked per (kernel, current device) under a mutex (forward and autograd threads both launch),
// and a failure is kept for the VITK_CHECK_LAUNCH that follows the launch (runtime.hip).
extern "C" int vitk_lds_optin(const void* kernel, size_t bytes);
extern "C" int vitk_optin_failed(void);      // 1 (and cleared) when an opt-in failed on this thread since the last check

#define VITK_CHECK_LAUNCH(what)                                              \
    do {                                                                     \
        if (vitk_optin_failed()) return VITK_ERR_LAUNCH;                     \
        hipError_t e__ = hipGetLastError();                                  \
        if (e__ != hipSuccess) {                                             \
            vitk_set_error("%s: HIP error %s", what, hipGetErrorString(e__)); \
            return VITK_ERR_LAUNCH;                                          \
        }                                                                    \
    } while (0)
