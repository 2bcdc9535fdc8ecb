// FAVOR+ for up to 64 random features (feature pitch mp = 16 or 64: one or four 16-wide tiles), the shapes of the
// reference's configurations (nb_features = 16 by default, train.py:75; 64: timm_classes/timm_vision_performer.py:26-101, model.py:257-352).  Same algorithm,
// workspace and results as the kernels of favor.hip (which keep m > 64); what differs is where the
// operands live:
//   - everything that belongs to a (b, h) rather than to a token -- w row-major and transposed, KV / dKV as bf16
//     hi / lo images, KV^T / dKV^T, ksum / dksum -- is staged in LDS once per workgroup (the plain kernels fetch ~45
//     fragments per 16 tokens from L2, in dependent stages, and split KV into hi / lo again every time);
//   - a workgroup walks several 16-token blocks per wave and fetches the next block's rows while it works on the
//     current one;
//   - the kernel kind and the presence of position tables are template parameters: the loop bodies are straight-line
//     code (with both at run time every exp sat in a basic block of its own, each with its own waits).
// Measured at B = 64, L = 577, H = 12, dh = 64, m = 64, bf16 (us, plain -> here): see DESIGN.md "FAVOR+".
#include "favor_common.h"

namespace vitk {

template <typename T, int DKB, int MT> struct FastCfg {
    using C = ACfg<T, DKB>;
    static constexpr int MP = 16 * MT;
    static constexpr int DHP = 32 * DKB;
    static constexpr int EPC = 16 / sizeof(T);
    static constexpr int WLD = DHP + EPC;                     // row-major images [MP][WLD]
    static constexpr int WTLD = MP + 8;                       // transposed images [DHP][WTLD]
    static constexpr int DLD = DHP + 8;                       // bf16 hi / lo images of KV / dKV [MP][DLD]
    static constexpr int RLD = MP + 4;                        // fp32 reduction image [DHP][RLD]
    static constexpr size_t hilo_bytes = sizeof(T) == 2 ? (size_t)2 * MP * DLD * 2 : (size_t)MP * WLD * 4;
    static constexpr size_t out_bytes = (size_t)(MP * WLD + DHP * WTLD) * sizeof(T) + (size_t)MP * 4;
    static constexpr size_t kv_bytes(int mtc) { return (size_t)4 * 32 * C::ASTR * sizeof(T) + (size_t)(DHP * (16 * mtc + 4) + 16 * mtc) * 4; }
    static constexpr size_t bwd_q_bytes = (size_t)(MP * WLD + DHP * WTLD + 4 * 16 * C::ASTR) * sizeof(T) + hilo_bytes +
                                          (size_t)(DHP * RLD + 2 * MP + 4 * 64) * 4;
    static constexpr size_t bwd_kv_bytes = (size_t)(MP * WLD + 2 * DHP * WTLD) * sizeof(T) + hilo_bytes + (size_t)MP * 4;
};

// w [m][dh] -> row-major image wr [MP][WLD] (zero-padded) and, when wT is given, its transpose [DHP][WTLD]:
// 16-byte chunks in, rows and columns out
template <typename T, int DKB, int MT>
__device__ __forceinline__ void stage_w(T* wr, T* wT, const T* __restrict__ W, int m, int dh, int tid) {
    using F = FastCfg<T, DKB, MT>;
    constexpr int CPR = F::DHP / F::EPC;
    for (int c = tid; c < F::MP * CPR; c += blockDim.x) {
        const int mm = c / CPR, d0 = (c - mm * CPR) * F::EPC;
        u32x4 raw = {0, 0, 0, 0};
        if (mm < m && d0 < dh) raw = *reinterpret_cast<const u32x4*>(W + (long)mm * dh + d0);
        *reinterpret_cast<u32x4*>(wr + mm * F::WLD + d0) = raw;
        if (wT) {
            const T* e = reinterpret_cast<const T*>(&raw);
#pragma unroll
            for (int j = 0; j < F::EPC; ++j) wT[(d0 + j) * F::WTLD + mm] = e[j];
        }
    }
}

// X^T [dh][mp] (T, global) -> [DHP][WTLD] image, rows d >= dh zero
template <typename T, int DKB, int MT>
__device__ __forceinline__ void stage_t(T* img, const T* __restrict__ XT, int dh, int mp, int tid) {
    using F = FastCfg<T, DKB, MT>;
    constexpr int CPR = F::MP / F::EPC;
    for (int c = tid; c < F::DHP * CPR; c += blockDim.x) {
        const int d = c / CPR, m0 = (c - d * CPR) * F::EPC;
        u32x4 raw = {0, 0, 0, 0};
        if (d < dh) raw = *reinterpret_cast<const u32x4*>(XT + (long)d * mp + m0);
        *reinterpret_cast<u32x4*>(img + d * F::WTLD + m0) = raw;
    }
}

// fp32 [mp][dhp] (global) -> bf16 hi / lo images [MP][DLD] x 2 (x ~ hi + lo), or one fp32 image [MP][WLD]
template <typename T, int DKB, int MT>
__device__ __forceinline__ void stage_hilo(char* img, const float* __restrict__ X32, int dhp, int tid, T* imgT = nullptr) {
    using F = FastCfg<T, DKB, MT>;
    for (int c = tid; c < F::MP * (F::DHP / 4); c += blockDim.x) {
        const int mm = c / (F::DHP / 4), d0 = (c - mm * (F::DHP / 4)) * 4;
        const f32x4 v = *reinterpret_cast<const f32x4*>(X32 + (long)mm * dhp + d0);
        if (imgT) {                                               // and X^T in T: [DHP][WTLD]
#pragma unroll
            for (int j = 0; j < 4; ++j) imgT[(d0 + j) * F::WTLD + mm] = from_f32<T>(v[j]);
        }
        if constexpr (sizeof(T) == 2) {
            __bf16* hi = reinterpret_cast<__bf16*>(img) + mm * F::DLD + d0;
            bf16x4 vh, vl;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                vh[j] = (__bf16)v[j];
                vl[j] = (__bf16)(v[j] - (float)vh[j]);
            }
            *reinterpret_cast<bf16x4*>(hi) = vh;
            *reinterpret_cast<bf16x4*>(hi + F::MP * F::DLD) = vl;
        } else {
            *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(img) + mm * F::WLD + d0) = v;
        }
    }
}

template <typename T, int DKB> __device__ __forceinline__ void frag_copy(Frag<T> (&d)[DKB], const Frag<T> (&s)[DKB]) {
#pragma unroll
    for (int kb = 0; kb < DKB; ++kb) d[kb] = s[kb];
}

// Sum of the four waves' accumulator tiles acc[dt][mt][r] = X[16 dt + 4 g + r][16 mt + li] (+ a per-column scalar sv[mt])
// into red[row * RLD + col] and sums[col] (sums = red + 16 ND * RLD): waves 0 and 1 store into two copies (redB lies
// over LDS the loop no longer needs), waves 2 and 3 add to them, everybody adds copy B to copy A.  Plain LDS
// operations in a fixed order: bit-reproducible, two hand-offs instead of four, nothing to zero beforehand.
template <int ND, int MTN, int RLD>
__device__ __forceinline__ void wave_quad_sum(float* red, float* redB, const f32x4 (&acc)[ND][MTN], const float (&sv)[MTN],
                                              int wave, int lane, int tid) {
    const int g = lane >> 4, li = lane & 15;
    constexpr int NR = 16 * ND * RLD;
    __syncthreads();                        // every wave has left the loop
#pragma unroll
    for (int turn = 0; turn < 2; ++turn) {
        if ((wave >> 1) == turn) {
            float* r = (wave & 1) ? redB : red;
#pragma unroll
            for (int mt = 0; mt < MTN; ++mt) {
                float s = sv[mt];
                s += __shfl_xor(s, 16, 64);
                s += __shfl_xor(s, 32, 64);
                if (g == 0) r[NR + 16 * mt + li] = turn ? r[NR + 16 * mt + li] + s : s;
#pragma unroll
                for (int dt = 0; dt < ND; ++dt)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        float* c = r + (16 * dt + 4 * g + q) * RLD + 16 * mt + li;
                        *c = turn ? *c + acc[dt][mt][q] : acc[dt][mt][q];
                    }
            }
        }
        __syncthreads();
    }
    for (int i = tid; i < NR + 16 * MTN; i += 256) red[i] += redB[i];
    __syncthreads();
}

// ------------------------------------------------------------------ global key max (+ argmax)
// grid (H, B): w fragments stay in registers, the next 16 K rows are in flight during the products.
template <typename T, int DKB, int MT>
__global__ __launch_bounds__(256) void favor_kmax_fast_kernel(const FavorParams p) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int g = lane >> 4, li = lane & 15;
    const int h = blockIdx.x, b = blockIdx.y;
    const int L = p.L, dh = p.dh, m = p.m;
    const T* K = reinterpret_cast<const T*>(p.k) + (long)b * L * p.ld + h * dh;
    const T* W = reinterpret_cast<const T*>(p.w);
    Frag<T> wf[MT][DKB];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int kb = 0; kb < DKB; ++kb) row_frag_global<T>(wf[mt][kb], W, dh, 16 * mt + li, m, dh, kb, lane);
    unsigned int bestb = 0u, besti = 0u;                      // ordered bits of the lane's largest dd and its flat index
    // PF blocks of 16 rows are requested together and then worked through: one block ahead, a wave has 1 KB in flight
    // and the kernel runs at the latency of its loads (12 waves per CU: ~1.5 TB/s)
    constexpr int PF = 5;
    const int lfirst = 16 * (wave + 4 * blockIdx.z), lstep = 64 * gridDim.z;   // grid z: workgroups that share a (b, h)
    for (int lb = lfirst; lb < L; lb += PF * lstep) {
        Frag<T> xf[PF][DKB];
#pragma unroll
        for (int j = 0; j < PF; ++j)
#pragma unroll
            for (int kb = 0; kb < DKB; ++kb) row_frag_global<T>(xf[j][kb], K, p.ld, lb + j * lstep + li, L, dh, kb, lane);
#pragma unroll
        for (int j = 0; j < PF; ++j) {
            const int l = lb + j * lstep + li;
            const unsigned int ibase = (unsigned int)((((unsigned long long)b * L + l) * p.H + h) * m);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                f32x4 s = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int kb = 0; kb < DKB; ++kb) s = mma(wf[mt][kb], xf[j][kb], s);  // dd^T[feature 16mt+4g+r][token li]
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int mi = 16 * mt + 4 * g + r;
                    const unsigned int ob = (l < L && mi < m) ? ordered_bits(s[r]) : 0u;
                    const bool up = ob > bestb;               // flat indices grow along the lane's walk: ties keep the first
                    bestb = up ? ob : bestb;
                    besti = up ? ibase + mi : besti;
                }
            }
        }
    }
    unsigned long long best = bestb ? (((unsigned long long)bestb << 32) | (0xffffffffull - besti)) : 0ull;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const unsigned long long other = __shfl_xor(best, o, 64);
        best = other > best ? other : best;
    }
    // one visit per workgroup, to the cell its (b, h) hashes to (3072 waves visiting ONE cell were most of this kernel's time)
    __shared__ unsigned long long wbest[4];
    if (lane == 0) wbest[wave] = best;
    __syncthreads();
    if (threadIdx.x == 0) {
#pragma unroll
        for (int wv = 1; wv < 4; ++wv) best = wbest[wv] > best ? wbest[wv] : best;
        unsigned long long* cell = p.kmax + ((b * p.H + h + blockIdx.z) & (KMAX_SLOTS - 1));
        if (best > __atomic_load_n(cell, __ATOMIC_RELAXED)) atomicMax(cell, best);
    }
}

// ------------------------------------------------------------------ KV, ksum
// grid 8 * ceil(B H / 8) * (MT / MTC), one-dimensional; as favor_kv_kernel, with the next 32 K / V rows in flight.  The
// MT / MTC workgroups of a (b, h) all read its K and V rows: they get consecutive slots of ONE XCD (workgroup id mod 8
// is the XCD), so the second reads hit that XCD's L2 -- as a third grid dimension they ran a round apart, on any XCD,
// and the kernel moved 252 MB per launch for 113 MB of rows (PMC), at 6 TB/s.  A workgroup owns MTC of the MT
// feature tiles (its own columns of KV: no exchange, the same sums in the same order whatever MTC is): with all four
// the accumulators take half the register file, two workgroups fit a CU and 768 of them run in a round and a half;
// with two, 1536 half-sized workgroups run three to a CU in two full rounds.
template <typename T, int DKB, int MT, int MTC, int KIND, bool POS>
__global__ __launch_bounds__(256, sizeof(T) == 2 ? (MTC <= 2 ? 3 : 2) : 1) void favor_kv_fast_kernel(const FavorParams p) {
    using F = FastCfg<T, DKB, MT>;
    using C = ACfg<T, DKB>;
    constexpr int ASTR = C::ASTR, ND = 2 * DKB, MPC = 16 * MTC, RLD = MPC + 4, DHP = F::DHP, EPC = F::EPC;
    constexpr int NV = 32 * C::CPR / 64;                      // 16-byte chunks of the 32-row V image per lane
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g = lane >> 4, li = lane & 15;
    constexpr int NZ = MT / MTC;
    const int slot = blockIdx.x >> 3, bhi = (slot / NZ) * 8 + (blockIdx.x & 7);
    if (bhi >= p.B * p.H) return;
    const int h = bhi % p.H, b = bhi / p.H, mc0 = (slot % NZ) * MPC;
    const int L = p.L, dh = p.dh, m = p.m;
    const long bh = (long)b * p.H + h;
    T* Vw = reinterpret_cast<T*>(smem) + wave * 32 * ASTR;
    float* red = reinterpret_cast<float*>(reinterpret_cast<T*>(smem) + 4 * 32 * ASTR);   // [DHP][RLD] + ksum[MPC]
    float* ksr = red + DHP * RLD;
    static_assert((size_t)4 * 32 * ASTR * sizeof(T) >= (size_t)(DHP * RLD + MPC) * 4, "second copy of the sums over the V images");

    const T* K = reinterpret_cast<const T*>(p.k) + (long)b * L * p.ld + h * dh;
    const T* V = reinterpret_cast<const T*>(p.v) + (long)b * L * p.ld + h * dh;
    const T* W = reinterpret_cast<const T*>(p.w);
    float stab = 0.f;
    if constexpr (KIND == FAVOR_SOFTMAX) stab = unordered_bits((unsigned int)(kmax_all(p.kmax, KMAX_SLOTS, lane) >> 32));

    Frag<T> ones;
#pragma unroll
    for (int e = 0; e < 8; ++e) frag_set_f(ones, e, 1.f);
    Frag<T> wf[MTC][DKB];
#pragma unroll
    for (int mt = 0; mt < MTC; ++mt)
#pragma unroll
        for (int kb = 0; kb < DKB; ++kb) row_frag_global<T>(wf[mt][kb], W, dh, mc0 + 16 * mt + li, m, dh, kb, lane);
    f32x4 acc[ND][MTC];
    float ks[MTC];
#pragma unroll
    for (int mt = 0; mt < MTC; ++mt) {
        ks[mt] = 0.f;
#pragma unroll
        for (int dt = 0; dt < ND; ++dt) acc[dt][mt] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    Frag<T> xf[2][DKB];
    u32x4 vraw[NV];
    auto fetch = [&](int l0, Frag<T> (&x)[2][DKB], u32x4 (&vr)[NV]) {
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int kb = 0; kb < DKB; ++kb) row_frag_global<T>(x[u][kb], K, p.ld, l0 + 16 * u + li, L, dh, kb, lane);
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            const int c = lane + 64 * j, r = c / C::CPR, d = (c - r * C::CPR) * EPC;
            vr[j] = u32x4{0, 0, 0, 0};
            if (l0 + r < L && d < dh) vr[j] = *reinterpret_cast<const u32x4*>(V + (long)(l0 + r) * p.ld + d);
        }
    };
    fetch(32 * wave, xf, vraw);
    for (int l0 = 32 * wave; l0 < L; l0 += 128) {
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            const int c = lane + 64 * j, r = c / C::CPR, d = (c - r * C::CPR) * EPC;
            *reinterpret_cast<u32x4*>(Vw + r * ASTR + d) = vraw[j];
        }
        Frag<T> xn[2][DKB];
        u32x4 vrawn[NV];
        fetch(l0 + 128 < L ? l0 + 128 : L, xn, vrawn);        // past the end: zeros, no loads
        f32x4 kp[2][MTC];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            f32x4 dg = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int kb = 0; kb < DKB; ++kb) dg = mma(xf[u][kb], ones, dg);      // row sums of K: dg[r] = sum_d K[4g+r][d]
#pragma unroll
            for (int mt = 0; mt < MTC; ++mt) {
                f32x4 s = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int kb = 0; kb < DKB; ++kb) s = mma(xf[u][kb], wf[mt][kb], s);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int lt = l0 + 16 * u + 4 * g + r;
                    const bool ok = (lt < L) && (mc0 + 16 * mt + li < m);
                    float v = feat_t<KIND>(s[r], feat_exp<KIND, T>(s[r], dg[r] * p.diag_c, stab), p.ratio, p.eps);
                    if constexpr (POS) v += p.pk[((long)min(lt, L - 1) * p.H + h) * p.mp + mc0 + 16 * mt + li];
                    v = ok ? v : 0.f;
                    kp[u][mt][r] = v;
                    ks[mt] += v;
                }
            }
        }
        wave_lds_sync();
#pragma unroll
        for (int mt = 0; mt < MTC; ++mt) {
            Frag<T> bf;
            acc_to_frag(bf, kp[0][mt], kp[1][mt]);
#pragma unroll
            for (int dt = 0; dt < ND; ++dt) {
                Frag<T> a;
                col_frag(a, Vw, ASTR, 4 * g, 16 + 4 * g, 16 * dt, lane);
                acc[dt][mt] = mma(a, bf, acc[dt][mt]);
            }
        }
        wave_lds_sync();
#pragma unroll
        for (int u = 0; u < 2; ++u) frag_copy<T, DKB>(xf[u], xn[u]);
#pragma unroll
        for (int j = 0; j < NV; ++j) vraw[j] = vrawn[j];
    }
    // the four waves' sums, in a fixed order (the forward is bit-reproducible)
    wave_quad_sum<ND, MTC, RLD>(red, reinterpret_cast<float*>(smem), acc, ks, wave, lane, tid);
    T* KVT = reinterpret_cast<T*>(p.kvt) + bh * dh * p.mp;
    T* KV = reinterpret_cast<T*>(p.kv) + bh * p.mp * p.dhp;
    // each layout written along ITS rows, four elements per lane (one loop over (d, m) for all three left two of them
    // as 2- and 4-byte stores a row apart per lane: 16 of this kernel's 54 us)
    for (int i = tid; i < dh * (MPC / 4); i += 256) {
        const int d = i / (MPC / 4), m4 = (i - d * (MPC / 4)) * 4;
        store4<T>(KVT + (long)d * p.mp + mc0 + m4, *reinterpret_cast<const f32x4*>(red + d * RLD + m4));
    }
    for (int i = tid; i < MPC * (DHP / 4); i += 256) {
        const int mm = i / (DHP / 4), d4 = (i - mm * (DHP / 4)) * 4;
        f32x4 v;
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = d4 + j < dh ? red[(d4 + j) * RLD + mm] : 0.f;
        store4<T>(KV + (long)(mc0 + mm) * p.dhp + d4, v);
        *reinterpret_cast<f32x4*>(p.kv32 + (bh * p.mp + mc0 + mm) * p.dhp + d4) = v;
    }
    for (int i = tid; i < MPC; i += 256) p.ksum[bh * p.mp + mc0 + i] = ksr[i];
}

// ------------------------------------------------------------------ out = (q' KV) / (q' . ksum)
// grid (ceil(groups of 64 tokens / p.groups), H, B); a wave owns 16 tokens of a group, lane = token.
template <typename T, int DKB, int MT, int KIND, bool POS>
__global__ __launch_bounds__(256) void favor_out_fast_kernel(const FavorParams p) {
    using F = FastCfg<T, DKB, MT>;
    constexpr int ND = 2 * DKB, MP = F::MP, DHP = F::DHP, WLD = F::WLD, WTLD = F::WTLD;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g = lane >> 4, li = lane & 15;
    const int h = blockIdx.y, b = blockIdx.z;
    const int L = p.L, dh = p.dh, m = p.m;
    const long bh = (long)b * p.H + h;
    T* wr = reinterpret_cast<T*>(smem);
    T* kvT = wr + MP * WLD;
    float* ks = reinterpret_cast<float*>(kvT + DHP * WTLD);
    stage_w<T, DKB, MT>(wr, nullptr, reinterpret_cast<const T*>(p.w), m, dh, tid);
    stage_t<T, DKB, MT>(kvT, reinterpret_cast<const T*>(p.kvt) + bh * dh * p.mp, dh, p.mp, tid);
    for (int i = tid; i < MP; i += 256) ks[i] = p.ksum[bh * p.mp + i];
    __syncthreads();

    const T* Q = reinterpret_cast<const T*>(p.q) + (long)b * L * p.ld + h * dh;
    const int g0 = blockIdx.x * p.groups, g1 = min(g0 + p.groups, (L + 63) / 64);
    Frag<T> xf[DKB];
#pragma unroll
    for (int kb = 0; kb < DKB; ++kb) row_frag_global<T>(xf[kb], Q, p.ld, g0 * 64 + 16 * wave + li, L, dh, kb, lane);
    for (int grp = g0; grp < g1; ++grp) {
        const int l0 = grp * 64 + 16 * wave, l = l0 + li;
        Frag<T> xn[DKB];
#pragma unroll
        for (int kb = 0; kb < DKB; ++kb) row_frag_global<T>(xn[kb], Q, p.ld, grp + 1 < g1 ? l + 64 : L, L, dh, kb, lane);
        if (l0 < L) {
            f32x4 dd[MT];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                f32x4 s = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int kb = 0; kb < DKB; ++kb) {
                    Frag<T> wf;
                    frag_load(wf, wr + (16 * mt + li) * WLD + 32 * kb + 8 * g);
                    s = mma(wf, xf[kb], s);
                }
                dd[mt] = s;
            }
            const float diag = row_sum<T, DKB>(xf) * p.diag_c;
            float mx = -INFINITY;
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (16 * mt + 4 * g + r < m) mx = fmaxf(mx, dd[mt][r]);
            mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
            mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
            float den = 0.f;
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                const f32x4 ksv = *reinterpret_cast<const f32x4*>(ks + 16 * mt + 4 * g);
                f32x4 pqv = {0.f, 0.f, 0.f, 0.f};
                if constexpr (POS) pqv = *reinterpret_cast<const f32x4*>(p.pq + ((long)min(l, L - 1) * p.H + h) * p.mp + 16 * mt + 4 * g);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const bool ok = 16 * mt + 4 * g + r < m;
                    const float f = feat_t<KIND>(dd[mt][r], feat_exp<KIND, T>(dd[mt][r], diag, mx), p.ratio, p.eps) + pqv[r];
                    const float v = ok ? f : 0.f;
                    dd[mt][r] = v;
                    den += v * ksv[r];
                }
            }
            den += __shfl_xor(den, 16, 64);
            den += __shfl_xor(den, 32, 64);
            const float inv = 1.f / den;
            T* O = reinterpret_cast<T*>(p.out) + ((long)b * L + l) * p.ldo + h * dh;
            Frag<T> bq[(MT + 1) / 2];
#pragma unroll
            for (int mb = 0; mb < (MT + 1) / 2; ++mb)
                acc_to_frag(bq[mb], dd[2 * mb], 2 * mb + 1 < MT ? dd[2 * mb + 1] : f32x4{0.f, 0.f, 0.f, 0.f});
#pragma unroll
            for (int dt = 0; dt < ND; ++dt) {
                f32x4 num = {0.f, 0.f, 0.f, 0.f};
                const T* krow = kvT + (16 * dt + li) * WTLD;                      // A row: KV^T[d][.]
#pragma unroll
                for (int mb = 0; mb < (MT + 1) / 2; ++mb) {
                    Frag<T> a;
                    frag_load_split<T>(a, krow + 32 * mb + 4 * g, krow + 32 * mb + 16 + 4 * g, 2 * mb + 1 < MT);
                    num = mma(a, bq[mb], num);
                }
                const int dd0 = 16 * dt + 4 * g;
                if (l < L && dd0 < dh) store4<T>(O + dd0, num * inv);
            }
        }
        frag_copy<T, DKB>(xf, xn);
    }
}

// ------------------------------------------------------------------ backward, query side
// One workgroup per (b, h); waves split the tokens in blocks of 16 (see favor_bwd_q_kernel for the algebra).
template <typename T, int DKB, int MT, int KIND, bool POS>
__global__ __launch_bounds__(256, sizeof(T) == 2 ? 2 : 1) void favor_bwd_q_fast_kernel(const FavorParams p) {
    using F = FastCfg<T, DKB, MT>;
    using C = ACfg<T, DKB>;
    constexpr int ASTR = C::ASTR, ND = 2 * DKB;
    constexpr int MP = F::MP, DHP = F::DHP, WLD = F::WLD, WTLD = F::WTLD, DLD = F::DLD, RLD = F::RLD, EPC = F::EPC;
    constexpr int NCH = 16 * C::CPR / 64;                        // 16-byte chunks of the dO image per lane
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g = lane >> 4, li = lane & 15;
    const int h = blockIdx.x, b = blockIdx.y;
    const int L = p.L, dh = p.dh, m = p.m;
    T* wr = reinterpret_cast<T*>(smem);                          // [MP][WLD]: w, rows m
    T* wT = wr + MP * WLD;                                       // [DHP][WTLD]: w^T, rows d
    T* dOw = wT + DHP * WTLD + wave * 16 * ASTR;                 // wave-private dO image (16 rows)
    char* kvimg = reinterpret_cast<char*>(wT + DHP * WTLD + 4 * 16 * ASTR);
    float* red = reinterpret_cast<float*>(kvimg + F::hilo_bytes);    // [DHP][RLD]
    float* dksr = red + DHP * RLD;                               // [MP]
    float* ks = dksr + MP;                                       // [MP]
    float* rowv = ks + MP + wave * 64;                           // per wave: inv_den[16], dden[16], mx[16], diag[16]
    const long bh = (long)b * p.H + h;

    stage_w<T, DKB, MT>(wr, wT, reinterpret_cast<const T*>(p.w), m, dh, tid);
    stage_hilo<T, DKB, MT>(kvimg, p.kv32 + bh * p.mp * p.dhp, p.dhp, tid);
    static_assert((size_t)(MP * WLD + DHP * WTLD) * sizeof(T) >= (size_t)(DHP * RLD + MP) * 4, "second copy of the sums over the w images");
    for (int i = tid; i < MP; i += 256) ks[i] = p.ksum[bh * p.mp + i];
    __syncthreads();

    const T* Q = reinterpret_cast<const T*>(p.q) + (long)b * L * p.ld + h * dh;
    const T* OUT = reinterpret_cast<const T*>(p.out) + (long)b * L * p.ldo + h * dh;
    const T* DO = reinterpret_cast<const T*>(p.dout) + (long)b * L * p.ldo + h * dh;

    f32x4 dkvt[ND][MT];
    float dks[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        dks[mt] = 0.f;
#pragma unroll
        for (int dt = 0; dt < ND; ++dt) dkvt[dt][mt] = f32x4{0.f, 0.f, 0.f, 0.f};
    }

    Frag<T> xf[DKB], of[DKB];
    u32x4 doraw[NCH];
    auto fetch = [&](int l0, Frag<T> (&x)[DKB], Frag<T> (&o)[DKB], u32x4 (&dr)[NCH]) {
#pragma unroll
        for (int kb = 0; kb < DKB; ++kb) {
            row_frag_global<T>(x[kb], Q, p.ld, l0 + li, L, dh, kb, lane);
            row_frag_global<T>(o[kb], OUT, p.ldo, l0 + li, L, dh, kb, lane);
        }
#pragma unroll
        for (int j = 0; j < NCH; ++j) {
            const int c = lane + 64 * j, r = c / C::CPR, d = (c - r * C::CPR) * EPC;
            dr[j] = u32x4{0, 0, 0, 0};
            if (l0 + r < L && d < dh) dr[j] = *reinterpret_cast<const u32x4*>(DO + (long)(l0 + r) * p.ldo + d);
        }
    };
    fetch(16 * wave, xf, of, doraw);

    for (int l0 = 16 * wave; l0 < L; l0 += 64) {
        const int l = l0 + li;
#pragma unroll
        for (int j = 0; j < NCH; ++j) {
            const int c = lane + 64 * j, r = c / C::CPR, d = (c - r * C::CPR) * EPC;
            *reinterpret_cast<u32x4*>(dOw + r * ASTR + d) = doraw[j];
        }
        wave_lds_sync();
        Frag<T> xn[DKB], on[DKB];
        u32x4 dorawn[NCH];
        fetch(l0 + 64 < L ? l0 + 64 : L, xn, on, dorawn);        // past the end: zeros, no loads
        // ---- phase A: lane = token -------------------------------------------------------------
        f32x4 dd[MT];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            f32x4 s = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int kb = 0; kb < DKB; ++kb) {
                Frag<T> wf;
                frag_load(wf, wr + (16 * mt + li) * WLD + 32 * kb + 8 * g);
                s = mma(wf, xf[kb], s);
            }
            dd[mt] = s;
        }
        const float diag = row_sum<T, DKB>(xf) * p.diag_c;
        float mx = -INFINITY;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (16 * mt + 4 * g + r < m) mx = fmaxf(mx, dd[mt][r]);
        mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        float den = 0.f;
        f32x4 ex[MT];                                            // exp(dd - diag - mx): feature and its derivative
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const f32x4 ksv = *reinterpret_cast<const f32x4*>(ks + 16 * mt + 4 * g);
            f32x4 pqv = {0.f, 0.f, 0.f, 0.f};
            if constexpr (POS) pqv = *reinterpret_cast<const f32x4*>(p.pq + ((long)min(l, L - 1) * p.H + h) * p.mp + 16 * mt + 4 * g);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const bool ok = 16 * mt + 4 * g + r < m;
                ex[mt][r] = feat_exp<KIND, T>(dd[mt][r], diag, mx);
                const float f = feat_t<KIND>(dd[mt][r], ex[mt][r], p.ratio, p.eps) + pqv[r];
                den += ok ? f * ksv[r] : 0.f;
            }
        }
        den += __shfl_xor(den, 16, 64);
        den += __shfl_xor(den, 32, 64);
        const float inv = (l < L) ? 1.f / den : 0.f;
        // dden = -(dout . out) / den: the lane's 8-element blocks of both rows, then over the 4 lane groups
        float dden = 0.f;
#pragma unroll
        for (int kb = 0; kb < DKB; ++kb) {
            Frag<T> gf;
            frag_load(gf, dOw + li * ASTR + 32 * kb + 8 * g);
#pragma unroll
            for (int e = 0; e < 8; ++e) dden -= frag_get_f(gf, e) * frag_get_f(of[kb], e);
        }
        dden += __shfl_xor(dden, 16, 64);
        dden += __shfl_xor(dden, 32, 64);
        dden *= inv;
        // dnum^T tiles (acc layout: rows d = 16dt+4g+r, column = token)
        f32x4 dnum[ND];
#pragma unroll
        for (int dt = 0; dt < ND; ++dt) dnum[dt] = load4<T>(dOw + li * ASTR + 16 * dt + 4 * g) * inv;
        // dq'^T[m][l] = sum_d KV[m][d] dnum^T[d][l] + dden ksum[m];  e = dq' * feat'
        float esum = 0.f;
        f32x4 e[MT];
        [[maybe_unused]] Frag<__bf16> bhi[DKB], blo[DKB];
        [[maybe_unused]] Frag<T> bn[DKB];
#pragma unroll
        for (int kb = 0; kb < DKB; ++kb) {
            if constexpr (sizeof(T) == 2) acc_to_frag_hilo(bhi[kb], blo[kb], dnum[2 * kb], dnum[2 * kb + 1]);
            else acc_to_frag(bn[kb], dnum[2 * kb], dnum[2 * kb + 1]);
        }
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            f32x4 s = {0.f, 0.f, 0.f, 0.f};
            const int mrow = 16 * mt + li;
#pragma unroll
            for (int kb = 0; kb < DKB; ++kb) {
                if constexpr (sizeof(T) == 2) {
                    const __bf16* hi = reinterpret_cast<const __bf16*>(kvimg) + mrow * DLD + 32 * kb + 4 * g;
                    Frag<__bf16> ah, alo;
                    frag_load_split<__bf16>(ah, hi, hi + 16, true);
                    frag_load_split<__bf16>(alo, hi + MP * DLD, hi + MP * DLD + 16, true);
                    s = mma(ah, bhi[kb], s);
                    s = mma(ah, blo[kb], s);
                    s = mma(alo, bhi[kb], s);
                } else {
                    const float* kr = reinterpret_cast<const float*>(kvimg) + mrow * WLD + 32 * kb + 4 * g;
                    Frag<T> a;
                    frag_load_split<T>(a, kr, kr + 16, true);
                    s = mma(a, bn[kb], s);
                }
            }
            const f32x4 ksv = *reinterpret_cast<const f32x4*>(ks + 16 * mt + 4 * g);
            f32x4 dnew;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const bool ok = 16 * mt + 4 * g + r < m;
                const float dqp = s[r] + dden * ksv[r];
                dnew[r] = ok ? dqp : 0.f;
                e[mt][r] = ok ? dqp * feat_grad_t<KIND>(dd[mt][r], ex[mt][r], p.ratio) : 0.f;
                esum += e[mt][r];
            }
            if constexpr (POS) {
                if (l < L) *reinterpret_cast<f32x4*>(p.dnq + (((long)b * L + l) * p.H + h) * p.mp + 16 * mt + 4 * g) = dnew;
            }
        }
        esum += __shfl_xor(esum, 16, 64);
        esum += __shfl_xor(esum, 32, 64);
        if constexpr (KIND == FAVOR_SOFTMAX) {
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const bool top = 16 * mt + 4 * g + r < m && dd[mt][r] == mx;
                    e[mt][r] -= top ? esum : 0.f;
                }
        }
        const float ddiag = KIND == FAVOR_SOFTMAX ? -esum * p.diag_c : 0.f;
        // dQ^T[d][l] = sum_m w^T[d][m] ddd^T[m][l] + ddiag
        T* DQ = reinterpret_cast<T*>(p.dq) + ((long)b * L + l) * p.ld_d + h * dh;
        Frag<T> be[(MT + 1) / 2];
#pragma unroll
        for (int mb = 0; mb < (MT + 1) / 2; ++mb)
            acc_to_frag(be[mb], e[2 * mb], 2 * mb + 1 < MT ? e[2 * mb + 1] : f32x4{0.f, 0.f, 0.f, 0.f});
#pragma unroll
        for (int dt = 0; dt < ND; ++dt) {
            f32x4 s = {ddiag, ddiag, ddiag, ddiag};
            const T* wrow = wT + (16 * dt + li) * WTLD;
#pragma unroll
            for (int mb = 0; mb < (MT + 1) / 2; ++mb) {
                Frag<T> a;
                frag_load_split<T>(a, wrow + 32 * mb + 4 * g, wrow + 32 * mb + 16 + 4 * g, 2 * mb + 1 < MT);
                s = mma(a, be[mb], s);
            }
            const int d0 = 16 * dt + 4 * g;
            if (l < L && d0 < dh) store4<T>(DQ + d0, s);
        }
        // hand the per-token scalars to phase B (rows in registers need them by row index)
        if (g == 0) {
            rowv[li] = inv;
            rowv[16 + li] = dden;
            rowv[32 + li] = mx;
            rowv[48 + li] = diag;
        }
        wave_lds_sync();
        // ---- phase B: tokens in accumulator rows: dKV^T[d][m] += dO^T[d][l] (q'[l][m] / den[l]) ----
        Frag<T> ad[ND];
#pragma unroll
        for (int dt = 0; dt < ND; ++dt) col_frag(ad[dt], dOw, ASTR, 4 * g, 4 * g, 16 * dt, lane);
        const f32x4 r_inv = *reinterpret_cast<const f32x4*>(rowv + 4 * g);
        const f32x4 r_dden = *reinterpret_cast<const f32x4*>(rowv + 16 + 4 * g);
        const f32x4 r_mx = *reinterpret_cast<const f32x4*>(rowv + 32 + 4 * g);
        const f32x4 r_diag = *reinterpret_cast<const f32x4*>(rowv + 48 + 4 * g);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            f32x4 s = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int kb = 0; kb < DKB; ++kb) {
                Frag<T> wf;
                frag_load(wf, wr + (16 * mt + li) * WLD + 32 * kb + 8 * g);
                s = mma(xf[kb], wf, s);                 // dd[token 4g+r][feature 16mt+li]
            }
            f32x4 qn;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int lr = 4 * g + r;
                const bool ok = (l0 + lr < L) && (16 * mt + li < m);
                float v = feat_t<KIND>(s[r], feat_exp<KIND, T>(s[r], r_diag[r], r_mx[r]), p.ratio, p.eps);
                if constexpr (POS) v += p.pq[((long)min(l0 + lr, L - 1) * p.H + h) * p.mp + 16 * mt + li];
                v = ok ? v : 0.f;
                dks[mt] += v * r_dden[r];
                qn[r] = v * r_inv[r];
            }
            Frag<T> bq;
            acc_to_frag(bq, qn, f32x4{0.f, 0.f, 0.f, 0.f});     // 16 tokens: second half of K is zero
#pragma unroll
            for (int dt = 0; dt < ND; ++dt) dkvt[dt][mt] = mma(ad[dt], bq, dkvt[dt][mt]);
        }
        wave_lds_sync();
        frag_copy<T, DKB>(xf, xn);
        frag_copy<T, DKB>(of, on);
#pragma unroll
        for (int j = 0; j < NCH; ++j) doraw[j] = dorawn[j];
    }
    // reduce over the 4 waves and write dKV in both layouts (+ fp32 transposed copy) and dksum
    // the four waves' sums (plain LDS adds: LDS float atomics from four waves onto the same 4 K addresses took ~40 us)
    wave_quad_sum<ND, MT, RLD>(red, reinterpret_cast<float*>(smem), dkvt, dks, wave, lane, tid);
    // dKV and dksum leave as fp32 sums only: the key-side kernel makes its bf16 images (both layouts) from them
    float* DKV32 = p.dkv32 + bh * p.mp * p.dhp;
    for (int i = tid; i < MP * (DHP / 4); i += 256) {
        const int mm = i / (DHP / 4), d4 = (i - mm * (DHP / 4)) * 4;
        f32x4 v;
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = d4 + j < dh ? red[(d4 + j) * RLD + mm] : 0.f;
        *reinterpret_cast<f32x4*>(DKV32 + (long)mm * p.dhp + d4) = v;
    }
    for (int i = tid; i < MP; i += 256) p.dksum[bh * p.mp + i] = dksr[i];
}

// ------------------------------------------------------------------ backward, key / value side
// grid (ceil(groups of 16 * NW tokens / p.groups), H, B), NW waves; lane = token (see favor_bwd_kv_kernel for the
// algebra).  bf16: eight waves share one set of staged images, so two workgroups put four waves on a SIMD (with four
// waves per workgroup the 46 KB of images allowed three).
template <typename T> struct BkvWaves { static constexpr int NW = sizeof(T) == 2 ? 8 : 4; };

template <typename T, int DKB, int MT, int KIND, bool POS>
__global__ __launch_bounds__(BkvWaves<T>::NW * 64) void favor_bwd_kv_fast_kernel(const FavorParams p) {
    using F = FastCfg<T, DKB, MT>;
    constexpr int NW = BkvWaves<T>::NW, GT = 16 * NW;     // waves per workgroup, tokens per group
    constexpr int ND = 2 * DKB;
    constexpr int MP = F::MP, DHP = F::DHP, WLD = F::WLD, WTLD = F::WTLD, DLD = F::DLD;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g = lane >> 4, li = lane & 15;
    const int h = blockIdx.y, b = blockIdx.z;
    const int L = p.L, dh = p.dh, m = p.m;
    T* wr = reinterpret_cast<T*>(smem);                       // w, rows m
    T* wT = wr + MP * WLD;                                    // w^T, rows d
    T* dkvT = wT + DHP * WTLD;                                // dKV^T, rows d
    char* dimg = reinterpret_cast<char*>(dkvT + DHP * WTLD);  // dKV, rows m
    float* dks = reinterpret_cast<float*>(dimg + F::hilo_bytes);
    const long bh = (long)b * p.H + h;
    stage_w<T, DKB, MT>(wr, wT, reinterpret_cast<const T*>(p.w), m, dh, tid);
    stage_hilo<T, DKB, MT>(dimg, p.dkv32 + bh * p.mp * p.dhp, p.dhp, tid, dkvT);   // both from the fp32 sums (rows d >= dh: 0)
    for (int i = tid; i < MP; i += 64 * NW) dks[i] = p.dksum[bh * p.mp + i];
    __syncthreads();

    const T* K = reinterpret_cast<const T*>(p.k) + (long)b * L * p.ld + h * dh;
    const T* V = reinterpret_cast<const T*>(p.v) + (long)b * L * p.ld + h * dh;
    float stab = 0.f;
    if constexpr (KIND == FAVOR_SOFTMAX) stab = unordered_bits((unsigned int)(kmax_all(p.kmax, KMAX_SLOTS, lane) >> 32));
    const int g0 = blockIdx.x * p.groups, g1 = min(g0 + p.groups, (L + GT - 1) / GT);
    Frag<T> xf[DKB], vf[DKB];
#pragma unroll
    for (int kb = 0; kb < DKB; ++kb) {
        row_frag_global<T>(xf[kb], K, p.ld, g0 * GT + 16 * wave + li, L, dh, kb, lane);
        row_frag_global<T>(vf[kb], V, p.ld, g0 * GT + 16 * wave + li, L, dh, kb, lane);
    }
    float esum_w = 0.f;
    for (int grp = g0; grp < g1; ++grp) {
        const int l0 = grp * GT + 16 * wave, l = l0 + li;
        Frag<T> xn[DKB], vn[DKB];
#pragma unroll
        for (int kb = 0; kb < DKB; ++kb) {
            const int ln = grp + 1 < g1 ? l + GT : L;         // past the end: zero fragments, no load
            row_frag_global<T>(xn[kb], K, p.ld, ln, L, dh, kb, lane);
            row_frag_global<T>(vn[kb], V, p.ld, ln, L, dh, kb, lane);
        }
        if (l0 < L) {
            const float diag = row_sum<T, DKB>(xf) * p.diag_c;
            f32x4 kp[MT], e[MT];
            float esum = 0.f;
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                f32x4 dd = {0.f, 0.f, 0.f, 0.f}, s = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int kb = 0; kb < DKB; ++kb) {
                    Frag<T> wf;
                    frag_load(wf, wr + (16 * mt + li) * WLD + 32 * kb + 8 * g);
                    dd = mma(wf, xf[kb], dd);                                        // dd^T[m][l]
                    if constexpr (sizeof(T) == 2) {
                        const __bf16* hi = reinterpret_cast<const __bf16*>(dimg) + (16 * mt + li) * DLD + 32 * kb + 8 * g;
                        Frag<__bf16> ah, alo;
                        frag_load(ah, hi);
                        frag_load(alo, hi + MP * DLD);
                        s = mma(ah, vf[kb], s);
                        s = mma(alo, vf[kb], s);
                    } else {
                        Frag<T> a;
                        frag_load(a, reinterpret_cast<const float*>(dimg) + (16 * mt + li) * WLD + 32 * kb + 8 * g);
                        s = mma(a, vf[kb], s);                                       // dk'^T[m][l]
                    }
                }
                const f32x4 dksv = *reinterpret_cast<const f32x4*>(dks + 16 * mt + 4 * g);
                f32x4 pkv = {0.f, 0.f, 0.f, 0.f}, dnew;
                if constexpr (POS) pkv = *reinterpret_cast<const f32x4*>(p.pk + ((long)min(l, L - 1) * p.H + h) * p.mp + 16 * mt + 4 * g);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const bool ok = (16 * mt + 4 * g + r < m) && (l < L);
                    const float ex = feat_exp<KIND, T>(dd[r], diag, stab);
                    const float f = feat_t<KIND>(dd[r], ex, p.ratio, p.eps) + pkv[r];
                    kp[mt][r] = ok ? f : 0.f;
                    dnew[r] = ok ? s[r] + dksv[r] : 0.f;
                    e[mt][r] = ok ? dnew[r] * feat_grad_t<KIND>(dd[r], ex, p.ratio) : 0.f;    // select: ex of a pad lane may be inf
                    esum += e[mt][r];
                }
                if constexpr (POS) {
                    if (l < L) *reinterpret_cast<f32x4*>(p.dnk + (((long)b * L + l) * p.H + h) * p.mp + 16 * mt + 4 * g) = dnew;
                }
            }
            esum_w += esum;
            float esum_l = esum;
            esum_l += __shfl_xor(esum_l, 16, 64);
            esum_l += __shfl_xor(esum_l, 32, 64);
            const float ddiag = KIND == FAVOR_SOFTMAX ? -esum_l * p.diag_c : 0.f;
            T* DK = reinterpret_cast<T*>(p.dk) + ((long)b * L + l) * p.ld_d + h * dh;
            T* DV = reinterpret_cast<T*>(p.dv) + ((long)b * L + l) * p.ld_d + h * dh;
            Frag<T> be[(MT + 1) / 2], bk[(MT + 1) / 2];
            const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int mb = 0; mb < (MT + 1) / 2; ++mb) {
                const bool has2 = 2 * mb + 1 < MT;
                acc_to_frag(be[mb], e[2 * mb], has2 ? e[2 * mb + 1] : zero);
                acc_to_frag(bk[mb], kp[2 * mb], has2 ? kp[2 * mb + 1] : zero);
            }
#pragma unroll
            for (int dt = 0; dt < ND; ++dt) {
                f32x4 sk = {ddiag, ddiag, ddiag, ddiag}, sv = {0.f, 0.f, 0.f, 0.f};
                const T* wrow = wT + (16 * dt + li) * WTLD;
                const T* drow = dkvT + (16 * dt + li) * WTLD;
#pragma unroll
                for (int mb = 0; mb < (MT + 1) / 2; ++mb) {
                    const bool has2 = 2 * mb + 1 < MT;
                    Frag<T> a, a2;
                    frag_load_split<T>(a, wrow + 32 * mb + 4 * g, wrow + 32 * mb + 16 + 4 * g, has2);
                    frag_load_split<T>(a2, drow + 32 * mb + 4 * g, drow + 32 * mb + 16 + 4 * g, has2);
                    sk = mma(a, be[mb], sk);
                    sv = mma(a2, bk[mb], sv);
                }
                const int d0 = 16 * dt + 4 * g;
                if (l < L && d0 < dh) {
                    store4<T>(DK + d0, sk);
                    store4<T>(DV + d0, sv);
                }
            }
        }
        frag_copy<T, DKB>(xf, xn);
        frag_copy<T, DKB>(vf, vn);
    }
    if constexpr (KIND == FAVOR_SOFTMAX) {
        const float tot = wave_sum(esum_w);
        if (lane == 0) atomicAdd(p.esum + bh, tot);
    }
}

namespace {

template <typename K> void set_lds(K kernel, size_t bytes) { (void)vitk_lds_optin((const void*)kernel, bytes); }

// 64-token groups per workgroup for the kernels that stage per-(b, h) state and then walk tokens: as many as still
// leaves `want` workgroups in the launch
int pick_groups(const FavorParams& p, long want) {
    const long ng = (p.L + 63) / 64;
    return (int)std::max(1L, std::min(ng, (long)p.B * p.H * ng / want));
}

template <typename T, int DKB, int MT, int KIND, bool POS> void launch_fast(const FavorParams& p0, bool bwd, hipStream_t st) {
    constexpr int MTC = MT >= 4 ? 2 : 1;   // feature tiles per workgroup of the KV kernel
    using F = FastCfg<T, DKB, MT>;
    FavorParams p = p0;
    p.kmax_slots = KMAX_SLOTS;
    const int ng = (p.L + 63) / 64;
    // per (kernel, device), thread-safe (vitk_lds_optin): the opt-in is a per-device attribute
    if (!bwd) {
        set_lds(favor_kv_fast_kernel<T, DKB, MT, MTC, KIND, POS>, F::kv_bytes(MTC));
        set_lds(favor_out_fast_kernel<T, DKB, MT, KIND, POS>, F::out_bytes);
    } else {
        set_lds(favor_bwd_q_fast_kernel<T, DKB, MT, KIND, POS>, F::bwd_q_bytes);
        set_lds(favor_bwd_kv_fast_kernel<T, DKB, MT, KIND, POS>, F::bwd_kv_bytes);
    }
    if (!bwd) {
        if (KIND == FAVOR_SOFTMAX) {
            (void)vitk_zero_async(p.kmax, 8 * KMAX_SLOTS, 8 * KMAX_SLOTS, 1, st);
            hipLaunchKernelGGL((favor_kmax_fast_kernel<T, DKB, MT>), dim3(p.H, p.B, 1), dim3(256), 0, st, p);
        }
        hipLaunchKernelGGL((favor_kv_fast_kernel<T, DKB, MT, MTC, KIND, POS>), dim3(8 * ((p.B * p.H + 7) / 8) * (MT / MTC)), dim3(256),
                           F::kv_bytes(MTC), st, p);
        p.groups = pick_groups(p, 3072);
        hipLaunchKernelGGL((favor_out_fast_kernel<T, DKB, MT, KIND, POS>), dim3((ng + p.groups - 1) / p.groups, p.H, p.B),
                           dim3(256), F::out_bytes, st, p);
    } else {
        if (KIND == FAVOR_SOFTMAX) (void)vitk_zero_async(p.esum, 4 * (size_t)p.nesum, 4 * (size_t)p.nesum, 1, st);
        // one workgroup per (b, h).  Two (fp32 sums added into zeroed memory: still order-independent) were measured: 112.6
        // us against 105.9 -- the fixed cost per workgroup outweighs the even rounds
        hipLaunchKernelGGL((favor_bwd_q_fast_kernel<T, DKB, MT, KIND, POS>), dim3(p.H, p.B), dim3(256), F::bwd_q_bytes, st, p);
        constexpr int NWK = BkvWaves<T>::NW;
        const int ngk = (p.L + 16 * NWK - 1) / (16 * NWK);
        p.groups = (int)std::max(1L, std::min((long)ngk, (long)p.B * p.H * ngk / (NWK == 8 ? 512 : 768)));
        hipLaunchKernelGGL((favor_bwd_kv_fast_kernel<T, DKB, MT, KIND, POS>), dim3((ngk + p.groups - 1) / p.groups, p.H, p.B),
                           dim3(64 * NWK), F::bwd_kv_bytes, st, p);
        if (KIND == FAVOR_SOFTMAX) hipLaunchKernelGGL(favor_fix_kernel<T>, dim3(1), dim3(64), 0, st, p);
    }
}

template <typename T, int DKB, int MT> void fast_kind(const FavorParams& p, bool bwd, hipStream_t st) {
    const bool pos = p.pq != nullptr;
    if (p.kind == FAVOR_SOFTMAX) {
        if (pos) launch_fast<T, DKB, MT, FAVOR_SOFTMAX, true>(p, bwd, st);
        else launch_fast<T, DKB, MT, FAVOR_SOFTMAX, false>(p, bwd, st);
    } else {
        if (pos) launch_fast<T, DKB, MT, FAVOR_RELU, true>(p, bwd, st);
        else launch_fast<T, DKB, MT, FAVOR_RELU, false>(p, bwd, st);
    }
}

template <typename T, int DKB> void fast_mt(const FavorParams& p, bool bwd, hipStream_t st) {
    if (p.mp == 16) fast_kind<T, DKB, 1>(p, bwd, st);
    else fast_kind<T, DKB, 4>(p, bwd, st);
}

}  // namespace

bool favor_fast_launch(const FavorParams& p, int dtype, bool bwd, hipStream_t st) {
    if (p.mp != 16 && p.mp != 64) return false;
    const int dkb = (p.dh + 31) / 32;
    if (dtype == VITK_BF16) {
        if (dkb == 1) fast_mt<__bf16, 1>(p, bwd, st);
        else fast_mt<__bf16, 2>(p, bwd, st);
    } else {
        if (dkb == 1) fast_mt<float, 1>(p, bwd, st);
        else fast_mt<float, 2>(p, bwd, st);
    }
    return true;
}

}  // namespace vitk
