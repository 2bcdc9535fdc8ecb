// Fused attention forward: softmax(scale * (QK^T + c2p + p2c), LAI band) V, flash style.
//
// Replaces AttentionLAI.forward's q@k^T / masked_fill / softmax / @v
// (timm_classes/timm_locality_aware_transformer.py:51-76) and
// Disentangled[LAI]SelfAttention.forward's bmm / disentangled_attention_bias / softmax / bmm
// (disentangled_transformer/disentangled_transformer.py:324-347, 357-426, 875-919, 929-998).
// The N x N scores, the [B*H, N, 2S] c2p / p2c products and the mask tensor of the reference
// never exist in HBM; the LAI mask is the predicate |i-j| <= w evaluated on indices.
//
// Work split: one 256-thread workgroup = 64 queries of one (batch, head); each of its 4 waves
// owns 16 queries (a lane owns ONE query: scores are computed transposed, S^T[key][query], so
// the running max / sum of the online softmax live on the lane and the probability tile is
// already the B operand of O^T += V^T P^T).  Keys advance in tiles of KT (64 bf16 / 32 f32)
// staged in LDS together with the KT+63 relative rows the tile pair can reach.
#include "attn_common.h"

namespace vitk {

struct AttnFwdParams {
    const void *q, *k, *v;   // element (b, n, h, d) at ((b*N + n) * ld + h*dh + d)
    long ld;
    const void *pk, *pq;     // band tables [2N-1][ld_p] (+ h*dh + d), row r <-> i - j = r - (N-1)
    long ld_p;
    void* out;               // [B*N][ldo] (+ h*dh + d)
    long ldo;
    float* lse;              // [B][H][N], natural log of the softmax denominator (incl. max)
    int B, H, N, dh;
    float scale;
    int w;                   // LAI half width, < 0: no mask
    DropParams drop;         // dropout on the probabilities (disentangled_transformer.py:915, timm_locality_aware_transformer.py:74):
                             // element (b, h, i, j) is element ((b*H + h)*N + i)*N + j of the stream; the normaliser keeps every term
};

template <typename T, int DKB, int KT, bool BIAS>
__global__ __launch_bounds__(256) void attn_fwd_kernel(const AttnFwdParams p) {
    using C = ACfg<T, DKB>;
    constexpr int ASTR = C::ASTR;
    constexpr int NS = KT / 16;        // key sub-tiles per tile
    constexpr int BAND = KT + 64;      // staged relative rows (KT + 63 used)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    T* Ks = reinterpret_cast<T*>(smem);
    T* Vs = Ks + KT * ASTR;
    T* Pks = Vs + KT * ASTR;
    T* Pqs = Pks + (BIAS ? BAND * ASTR : 0);
    float* scr = reinterpret_cast<float*>(Pqs + (BIAS ? BAND * ASTR : 0));

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g = lane >> 4, li = lane & 15;
    const int N = p.N, dh = p.dh;
    const int i0 = blockIdx.x * 64, h = blockIdx.y, b = blockIdx.z;
    const int iw = i0 + 16 * wave;
    float* xs = scr + wave * 2 * SCR_FLOATS;
    float* ys = xs + SCR_FLOATS;

    const T* Q = reinterpret_cast<const T*>(p.q) + (long)b * N * p.ld + h * dh;
    const T* K = reinterpret_cast<const T*>(p.k) + (long)b * N * p.ld + h * dh;
    const T* V = reinterpret_cast<const T*>(p.v) + (long)b * N * p.ld + h * dh;
    const T* PK = BIAS ? reinterpret_cast<const T*>(p.pk) + h * dh : nullptr;
    const T* PQ = BIAS ? reinterpret_cast<const T*>(p.pq) + h * dh : nullptr;

    Frag<T> qf[DKB];
#pragma unroll
    for (int kb = 0; kb < DKB; ++kb) row_frag_global<T>(qf[kb], Q, p.ld, iw + li, N, dh, kb, lane);

    constexpr float LOG2E = 1.4426950408889634f;
    const float sc2 = p.scale * LOG2E;
    float m2 = -INFINITY, lsum = 0.f;
    f32x4 oacc[2 * DKB];
#pragma unroll
    for (int dt = 0; dt < 2 * DKB; ++dt) oacc[dt] = f32x4{0.f, 0.f, 0.f, 0.f};

    // key range the 64 queries of this block can see
    int jlo = 0, jhi = N;
    if (p.w >= 0) {
        jlo = max(0, i0 - p.w);
        jhi = min(N, i0 + 63 + p.w + 1);
        jlo = (jlo / KT) * KT;
    }
    const int qi = iw + li;  // this lane's query
    const bool drop_on = p.drop.thresh != 0;
    const uint64_t drow = ((uint64_t)(b * p.H + h) * (uint64_t)N + (uint64_t)min(qi, N - 1)) * (uint64_t)N;

    for (int j0 = jlo; j0 < jhi; j0 += KT) {
        __syncthreads();
        stage_rows<T, DKB>(Ks, K, p.ld, j0, KT, 0, N, dh, tid, 256);
        stage_rows<T, DKB>(Vs, V, p.ld, j0, KT, 0, N, dh, tid, 256);
        if (BIAS) {
            const int r0 = i0 - j0 - (KT - 1) + (N - 1);  // table row of band row 0
            stage_rows<T, DKB>(Pks, PK, p.ld_p, r0, BAND, 0, 2 * N - 1, dh, tid, 256);
            stage_rows<T, DKB>(Pqs, PQ, p.ld_p, r0, BAND, 0, 2 * N - 1, dh, tid, 256);
        }
        __syncthreads();

        f32x4 s[NS];
#pragma unroll
        for (int sidx = 0; sidx < NS; ++sidx) {
            const int j0s = j0 + 16 * sidx;
            // wave-uniform skip of sub-tiles with no visible key
            const bool dead = (j0s >= N) || (p.w >= 0 && (j0s > iw + 15 + p.w || j0s + 15 < iw - p.w));
            if (dead) {
                s[sidx] = f32x4{-INFINITY, -INFINITY, -INFINITY, -INFINITY};
                continue;
            }
            Frag<T> kf[DKB];
#pragma unroll
            for (int kb = 0; kb < DKB; ++kb) row_frag<T, DKB>(kf[kb], Ks, 16 * sidx, kb, lane);
            const int rel = 16 * wave - 16 * sidx + KT - 16;
            f32x4 raw = score_tile<T, DKB, BIAS, true>(qf, kf, Pks, Pqs, rel, xs, ys, lane);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int j = j0s + 4 * g + r;
                const bool ok = j < N && (p.w < 0 || abs(qi - j) <= p.w);
                raw[r] = ok ? raw[r] * sc2 : -INFINITY;
            }
            s[sidx] = raw;
        }
        // ---- online softmax (log2 domain); a lane's 4*NS scores all belong to query qi ----
        float mt = -INFINITY;
#pragma unroll
        for (int sidx = 0; sidx < NS; ++sidx)
#pragma unroll
            for (int r = 0; r < 4; ++r) mt = fmaxf(mt, s[sidx][r]);
        mt = fmaxf(mt, __shfl_xor(mt, 16, 64));
        mt = fmaxf(mt, __shfl_xor(mt, 32, 64));
        const float mn = fmaxf(m2, mt);
        const float mref = (mn == -INFINITY) ? 0.f : mn;
        const float alpha = exp2f(m2 - mref);   // m2 = -inf -> 0
        float psum = 0.f;
#pragma unroll
        for (int sidx = 0; sidx < NS; ++sidx)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float pv = exp2f(s[sidx][r] - mref);
                psum += pv;
                s[sidx][r] = drop_on ? pv * drop_mul(p.drop, drow + (uint64_t)min(j0 + 16 * sidx + 4 * g + r, N - 1)) : pv;
            }
        lsum = lsum * alpha + psum;
        m2 = mn;
#pragma unroll
        for (int dt = 0; dt < 2 * DKB; ++dt) oacc[dt] *= alpha;
        // ---- O^T[d][query] += V^T[d][key] P^T[key][query], 32 keys per MFMA K block ----
#pragma unroll
        for (int kk = 0; kk < NS / 2; ++kk) {
            Frag<T> pf;
            acc_to_frag(pf, s[2 * kk], s[2 * kk + 1]);
#pragma unroll
            for (int dt = 0; dt < 2 * DKB; ++dt) {
                Frag<T> vf;
                col_frag(vf, Vs, ASTR, 32 * kk + 4 * g, 32 * kk + 16 + 4 * g, 16 * dt, lane);
                oacc[dt] = mma(vf, pf, oacc[dt]);
            }
        }
    }

    lsum += __shfl_xor(lsum, 16, 64);
    lsum += __shfl_xor(lsum, 32, 64);
    if (qi < N) {
        const float inv = 1.f / lsum;
        T* O = reinterpret_cast<T*>(p.out) + ((long)b * N + qi) * p.ldo + h * dh;
#pragma unroll
        for (int dt = 0; dt < 2 * DKB; ++dt) {
            const int d = 16 * dt + 4 * g;
            if (d < dh) {
                if (sizeof(T) == 2) {
                    bf16x4 o = {(__bf16)(oacc[dt][0] * inv), (__bf16)(oacc[dt][1] * inv),
                                (__bf16)(oacc[dt][2] * inv), (__bf16)(oacc[dt][3] * inv)};
                    *reinterpret_cast<bf16x4*>(O + d) = o;
                } else {
                    *reinterpret_cast<f32x4*>(O + d) = oacc[dt] * inv;
                }
            }
        }
        if (g == 0) p.lse[((long)b * p.H + h) * N + qi] = (m2 + log2f(lsum)) * 0.6931471805599453f;
    }
}

template <typename T, int DKB, int KT, bool BIAS> static size_t attn_fwd_lds() {
    using C = ACfg<T, DKB>;
    return (size_t)(2 * KT + (BIAS ? 2 * (KT + 64) : 0)) * C::ASTR * sizeof(T) + 4 * 2 * SCR_FLOATS * sizeof(float);
}

template <typename T, int DKB, int KT, bool BIAS>
static void attn_fwd_launch(const AttnFwdParams& p, hipStream_t st) {
    const size_t lds = attn_fwd_lds<T, DKB, KT, BIAS>();
    (void)vitk_lds_optin((const void*)attn_fwd_kernel<T, DKB, KT, BIAS>, lds);
    dim3 grid((p.N + 63) / 64, p.H, p.B);
    hipLaunchKernelGGL((attn_fwd_kernel<T, DKB, KT, BIAS>), grid, dim3(256), lds, st, p);
}

template <typename T, int KT> static void attn_fwd_dispatch(const AttnFwdParams& p, hipStream_t st) {
    const int dkb = (p.dh + 31) / 32;
    const bool bias = p.pk != nullptr;
#define VITK_CASE(D)                                         \
    case D:                                                  \
        if (bias) attn_fwd_launch<T, D, KT, true>(p, st);    \
        else attn_fwd_launch<T, D, KT, false>(p, st);        \
        break;
    switch (dkb) {
        VITK_CASE(1)
        VITK_CASE(2)
        VITK_CASE(3)
        VITK_CASE(4)
    }
#undef VITK_CASE
}

}  // namespace vitk

namespace vitk {
bool band_eligible(int dtype, int N, int dh, bool bias, int band_w);
bool band_launch(const BandParams& p, bool backward, hipStream_t st);
}  // namespace vitk

using namespace vitk;

static bool a16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

extern "C" int vitk_attn_fwd(int dtype, int B, int H, int N, int dh, const void* q, const void* k, const void* v,
                             long ld, const void* pk, const void* pq, long ld_p, float scale, int band_w,
                             void* out, long ldo, float* lse, float drop_p, unsigned long long drop_seed,
                             unsigned long long drop_offset, hipStream_t stream) {
    VITK_CHECK_ARG(drop_p >= 0.f && drop_p < 1.f, "vitk_attn_fwd: dropout p must be in [0, 1)");
    VITK_CHECK_ARG(dtype == VITK_F32 || dtype == VITK_BF16, "vitk_attn_fwd: bad dtype");
    VITK_CHECK_ARG(B > 0 && H > 0 && N > 0, "vitk_attn_fwd: empty problem");
    const int esz = dtype == VITK_BF16 ? 2 : 4;
    VITK_CHECK_ARG(dh > 0 && dh <= 128 && (dh * esz) % 16 == 0, "vitk_attn_fwd: head dim %d unsupported", dh);
    VITK_CHECK_ARG(q && k && v && out && lse, "vitk_attn_fwd: null pointer");
    VITK_CHECK_ARG((pk == nullptr) == (pq == nullptr), "vitk_attn_fwd: pk and pq go together");
    VITK_CHECK_ARG(a16(q) && a16(k) && a16(v) && a16(out) && (ld * esz) % 16 == 0 && (ldo * esz) % 16 == 0 &&
                       ld >= (long)H * dh && ldo >= (long)H * dh,
                   "vitk_attn_fwd: q/k/v/out rows must be 16-byte aligned and at least H*dh wide");
    VITK_CHECK_ARG(!pk || (a16(pk) && a16(pq) && (ld_p * esz) % 16 == 0 && ld_p >= (long)H * dh),
                   "vitk_attn_fwd: band tables must be 16-byte aligned rows of at least H*dh");
    VITK_CHECK_ARG(H <= 65535 && B <= 65535, "vitk_attn_fwd: grid too large");
    if (band_eligible(dtype, N, dh, pk != nullptr, band_w)) {   // whole-head kernel (attention_band.hip)
        BandParams bp = {};
        bp.q = q; bp.k = k; bp.v = v; bp.ld = ld;
        bp.pk = pk; bp.pq = pq; bp.ld_p = ld_p;
        bp.out = out; bp.ldo = ldo; bp.lse = lse;
        bp.B = B; bp.H = H; bp.N = N; bp.dh = dh;
        bp.scale = scale; bp.w = band_w;
        bp.drop = vitk_drop_params(drop_p, drop_seed, drop_offset);
        band_launch(bp, false, stream);
        VITK_CHECK_LAUNCH("vitk_attn_fwd");
        return VITK_OK;
    }
    AttnFwdParams p;
    p.q = q; p.k = k; p.v = v; p.ld = ld;
    p.pk = pk; p.pq = pq; p.ld_p = ld_p;
    p.out = out; p.ldo = ldo; p.lse = lse;
    p.B = B; p.H = H; p.N = N; p.dh = dh;
    p.scale = scale; p.w = band_w;
    p.drop = vitk_drop_params(drop_p, drop_seed, drop_offset);
    if (dtype == VITK_BF16) attn_fwd_dispatch<__bf16, 64>(p, stream);
    else attn_fwd_dispatch<float, 32>(p, stream);
    VITK_CHECK_LAUNCH("vitk_attn_fwd");
    return VITK_OK;
}
