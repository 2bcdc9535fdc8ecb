// Whole-head attention kernels for short sequences (bf16, head dim <= 80): the shapes of every ViT-224/16
// configuration (197 tokens: one 13-tile chunk), of the CIFAR plumbing config (65) and, chunk by chunk, of 384 px
// (577 tokens: three chunks).
//
// A workgroup owns one (batch, head, chunk of 13 tiles): the rows it needs are staged in LDS as whole chunk images, and
// between the few workgroup barriers the waves run independently -- no per-tile barriers, no cross-workgroup
// reductions, no atomics, no fp32 workspace:
//   forward : 16 waves, wave < 13 owns query tile 13 chunk + wave; lane = query (scores transposed), online softmax
//             in registers, O^T += V^T P^T with P taken straight from the score accumulators; the K / V images of the
//             key chunks its band reaches are staged one after the other.
//   backward: pass 1 is KEY-stationary (8 waves, a wave owns up to two key tiles; dK^T, dV^T stay in registers while
//             it sweeps the live query tiles; Q and dO images in LDS, its own K / V rows from global memory) and leaves
//             every 16 x 16 dS tile (bf16) in a workspace; pass 2 is QUERY-stationary (dQ^T in registers, K image in
//             LDS) and only reads those tiles back -- no second recompute of P -- as does the table-gradient kernel
//             (attn_bwd_relds_kernel, attention_bwd.hip).  A head of ONE chunk (every 224-px configuration) runs both
//             passes in one kernel, K staged over Q in between (two row images, not three: what lets 8 waves'
//             scratch fit in the 160 KiB); longer heads run them as two kernels over (head, chunk) grids.
//             No dQ atomics, no fp32 dQ accumulator.
// The LAI band (|i-j| <= w) limits a tile row / column to 2*cmax+1 live tile diagonals,
// cmax = (w+15)/16 (NT-1 when unmasked).  With the disentangled bias the live diagonals are walked
// in GROUPS of up to 5 (c_lo .. c_lo+4): the 96 relative-table rows a group can touch are staged,
// every wave sweeps the tile pairs of that group for the (at most two) tiles it owns -- whose
// softmax state / gradient accumulators stay in registers across groups -- and the next group's
// rows replace them (two workgroup barriers per group).  Every LAI stage of the reference
// (w = 8..20) is a single group; the unmasked disentangled stage of 197 tokens is five.
#include "attn_common.h"

namespace vitk {

constexpr int BAND_DS_ELEMS = 800;    // per wave: one zero-padded bf16 copy of a dS tile (X or Y layout, attn_common.h)
constexpr int BAND_WAVES = 8;         // backward, key-stationary kernel
constexpr int BAND_THREADS = 64 * BAND_WAVES;
// Half width of a diagonal group with the bias: 2 * 2 + 1 = 5 diagonals, 96 table rows -- or 3 diagonals / 64 rows
// for a head wider than 64, whose row images leave less of the LDS.
__host__ __device__ constexpr int band_gc(int DW) { return DW > 64 ? 1 : 2; }
constexpr int BAND_TPW = 2;           // tiles a wave of the key-stationary kernel can own: 13 tiles over 8 waves
// A CHUNK is 13 tiles = 208 rows: what one row image holds (plus one tile behind it, which the paired reads of the
// last tile touch).  Heads of up to 208 tokens -- every ViT-224/16 configuration -- are one chunk; longer ones (577
// tokens at 384 px) are walked chunk by chunk: a workgroup owns one chunk of queries (forward, dQ kernel) or keys
// (dK / dV kernel) and restages the other side's chunk images as it goes, its state staying in registers.
constexpr int BAND_CH = 13;
constexpr int BAND_NRC = 16 * BAND_CH + 16;
// The forward and the dQ kernel hold less per wave: 16 waves, every wave owns ONE query tile (the critical path is
// one tile's sweep, not two) and four waves per SIMD hide each other's LDS / MFMA / exp latencies.
constexpr int BAND_FWD_WAVES = 16;
constexpr int BAND_FWD_THREADS = 64 * BAND_FWD_WAVES;
static_assert(BAND_FWD_WAVES >= BAND_CH && BAND_WAVES * BAND_TPW >= BAND_CH, "a chunk's tiles all have an owner");


template <int DW, bool BIAS> struct BandCfg {
    static constexpr int DKB = (DW + 31) / 32;     // K blocks of a score product
    static constexpr int ND = DW / 16;             // 16-wide tiles of the head dim in the outputs
    using C = ACfg<__bf16, DKB, DW>;
    static constexpr int ASTR = C::ASTR;
    static constexpr int GC = band_gc(DW);
    static constexpr int WSCR_BYTES = (BIAS ? 2 * SCR_FLOATS * 4 + BAND_DS_ELEMS * 2 : 0);   // per wave, dK / dV kernel
    static int chunk_rows(int NT) { return 16 * (NT < BAND_CH ? NT : BAND_CH) + 16; }
    static size_t fwd_bytes(int NT, int cmax) {
        cmax = cmax < GC ? cmax : GC;
        const size_t rows = 2 * (size_t)chunk_rows(NT) + (BIAS ? 2 * (size_t)(32 * cmax + 32) : 0);
        const int sw = NT < BAND_CH ? NT : BAND_CH;   // waves that own a tile (the rest only stage)
        return rows * ASTR * 2 + (size_t)sw * (BIAS ? 2 * SCR_FLOATS : 0) * 4;
    }
    static size_t bwd_bytes(int NT, int cmax) {      // the one-chunk kernel: NT <= 13
        cmax = cmax < GC ? cmax : GC;
        const size_t rows = 2 * (size_t)(16 * NT + 16) + (BIAS ? 2 * (size_t)(32 * cmax + 32) : 0);
        return rows * ASTR * 2 + (size_t)2 * (16 * NT + 16) * 4 + (size_t)BAND_WAVES * WSCR_BYTES + 3 * 16 * ND * 4;   // + column sums
    }
    static size_t bwd_kv_bytes(int NT, int cmax) {
        cmax = cmax < GC ? cmax : GC;
        const size_t rows = 2 * (size_t)chunk_rows(NT) + (BIAS ? 2 * (size_t)(32 * cmax + 32) : 0);
        return rows * ASTR * 2 + (size_t)2 * chunk_rows(NT) * 4 + (size_t)BAND_WAVES * WSCR_BYTES + 2 * 16 * ND * 4;   // + column sums
    }
    static size_t bwd_q_bytes(int NT, int cmax) {
        cmax = cmax < GC ? cmax : GC;
        const size_t rows = (size_t)chunk_rows(NT) + (BIAS ? (size_t)(32 * cmax + 32) : 0);
        const int sw = NT < BAND_CH ? NT : BAND_CH;
        return rows * ASTR * 2 + (size_t)sw * (BIAS ? BAND_DS_ELEMS * 2 : 0) + 16 * ND * 4;
    }
};

// ------------------------------------------------------------------------------------------ forward
// Diagonal groups: group gi covers tile diagonals c = qt - kt in [c_lo, c_hi]; without the bias there
// is one group holding every live diagonal.
struct BandGroups {
    int cmax, gc, ngroups;
    __device__ __forceinline__ BandGroups(int w, int NT, bool bias, int gc_max) {
        cmax = w >= 0 ? min((w + 15) / 16, NT - 1) : NT - 1;
        gc = bias ? min(cmax, gc_max) : cmax;
        ngroups = (2 * cmax + 1 + 2 * gc) / (2 * gc + 1);
    }
    __device__ __forceinline__ int c_lo(int gi) const { return -cmax + gi * (2 * gc + 1); }
    __device__ __forceinline__ int c_hi(int gi) const { return min(cmax, c_lo(gi) + 2 * gc); }
};

// relative-table rows of one diagonal group -> LDS (row 0 <-> i - j = 16 c_lo - 15); Pqs may be null
template <int DKB, int DW>
__device__ __forceinline__ void band_stage_tables(const BandParams& p, __bf16* Pks, __bf16* Pqs, int WR, int c_lo, int h,
                                                  int tid, int nthreads) {
    const __bf16* PK = reinterpret_cast<const __bf16*>(p.pk) + h * p.dh;
    const __bf16* PQ = reinterpret_cast<const __bf16*>(p.pq) + h * p.dh;
    const int r_lo = 16 * c_lo - 15 + (p.N - 1);
    stage_rows<__bf16, DKB, DW>(Pks, PK, p.ld_p, r_lo, WR, 0, 2 * p.N - 1, p.dh, tid, nthreads);
    if (Pqs) stage_rows<__bf16, DKB, DW>(Pqs, PQ, p.ld_p, r_lo, WR, 0, 2 * p.N - 1, p.dh, tid, nthreads);
}

// grid (H, B, query chunks).  Wave `wave` < 13 owns query tile 13 * chunk + wave; the key chunks its band reaches are
// staged (K and V images) one after the other, the online-softmax state stays in registers.
template <int DW, bool BIAS>
__global__ __launch_bounds__(BAND_FWD_THREADS) void attn_fwd_band_kernel(const BandParams p) {
    using T = __bf16;
    using BC = BandCfg<DW, BIAS>;
    using C = typename BC::C;
    constexpr int DKB = BC::DKB;
    constexpr int ASTR = C::ASTR;
    constexpr int ND = BC::ND;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g = lane >> 4, li = lane & 15;
    const int N = p.N, dh = p.dh;
    const int NT = (N + 15) / 16;
    const int NRC = 16 * min(NT, BAND_CH) + 16;        // image rows: the chunk and one tile behind it
    const BandGroups grp(p.w, NT, BIAS, BC::GC);
    const int WR = 32 * grp.gc + 32;
    const int h = blockIdx.x, b = blockIdx.y, qc = blockIdx.z;

    T* Ks = reinterpret_cast<T*>(smem);
    T* Vs = Ks + NRC * ASTR;
    T* Pks = Vs + NRC * ASTR;
    T* Pqs = Pks + (BIAS ? WR * ASTR : 0);
    float* scr = reinterpret_cast<float*>(Pqs + (BIAS ? WR * ASTR : 0));
    float* xs = scr + min(wave, BAND_CH - 1) * 2 * SCR_FLOATS;     // waves past the chunk own no tile and touch no scratch
    float* ys = xs + SCR_FLOATS;

    const T* Q = reinterpret_cast<const T*>(p.q) + (long)b * N * p.ld + h * dh;
    const T* K = reinterpret_cast<const T*>(p.k) + (long)b * N * p.ld + h * dh;
    const T* V = reinterpret_cast<const T*>(p.v) + (long)b * N * p.ld + h * dh;

    constexpr float LOG2E = 1.4426950408889634f;
    const float sc2 = p.scale * LOG2E;
    const bool drop_on = p.drop.thresh != 0;
    const uint64_t dhead = (uint64_t)(b * p.H + h) * (uint64_t)N;
    // the query tile this wave owns
    const int qt = qc * BAND_CH + wave;
    const bool own = wave < BAND_CH && qt < NT;
    const int qi0 = 16 * qt, qi = qi0 + li;
    Frag<T> qf[DKB];
    float m2 = -INFINITY, lsum = 0.f;
    f32x4 oacc[ND];
#pragma unroll
    for (int kb = 0; kb < DKB; ++kb) row_frag_global<T>(qf[kb], Q, p.ld, own ? qi : N, N, dh, kb, lane);
#pragma unroll
    for (int dt = 0; dt < ND; ++dt) oacc[dt] = f32x4{0.f, 0.f, 0.f, 0.f};

    // key chunks the band of this query chunk reaches
    const int kt_min = max(0, qc * BAND_CH - grp.cmax), kt_max = min(NT - 1, qc * BAND_CH + BAND_CH - 1 + grp.cmax);
    bool first = true;
    for (int kc = kt_min / BAND_CH; kc <= kt_max / BAND_CH; ++kc) {
        const int k0 = kc * BAND_CH;                       // first key tile of the chunk
        if (!first) __syncthreads();                       // every wave is done with the previous images
        stage_rows<T, DKB, DW>(Ks, K, p.ld, 16 * k0, NRC, 0, N, dh, tid, BAND_FWD_THREADS);
        stage_rows<T, DKB, DW>(Vs, V, p.ld, 16 * k0, NRC, 0, N, dh, tid, BAND_FWD_THREADS);
        for (int gi = 0; gi < grp.ngroups; ++gi) {
            const int c_lo = grp.c_lo(gi), c_hi = grp.c_hi(gi);
            if (BIAS) {
                if (!first) __syncthreads();                      // every wave is done with the previous group's rows
                band_stage_tables<DKB, DW>(p, Pks, Pqs, WR, c_lo, h, tid, BAND_FWD_THREADS);
            }
            if (BIAS || gi == 0) __syncthreads();
            first = false;
            if (!own) continue;
            const int klo = max(k0, max(0, qt - c_hi)), khi = min(min(NT - 1, k0 + BAND_CH - 1), qt - c_lo);
            for (int ka = klo; ka <= khi; ka += 2) {
                f32x4 sc[2];
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const int kt = ka + u;
                    if (kt > khi) {
                        sc[u] = f32x4{-INFINITY, -INFINITY, -INFINITY, -INFINITY};
                        continue;
                    }
                    const int kj0 = 16 * kt;
                    Frag<T> kf[DKB];
#pragma unroll
                    for (int kb = 0; kb < DKB; ++kb) row_frag<T, DKB, DW>(kf[kb], Ks, kj0 - 16 * k0, kb, lane);
                    f32x4 raw = score_tile<T, DKB, BIAS, true, DW>(qf, kf, Pks, Pqs, qi0 - kj0 - 16 * c_lo, xs, ys, lane);
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int j = kj0 + 4 * g + r;
                        const bool ok = j < N && (p.w < 0 || abs(qi - j) <= p.w);
                        raw[r] = ok ? raw[r] * sc2 : -INFINITY;
                    }
                    sc[u] = raw;
                }
                float mt = -INFINITY;
#pragma unroll
                for (int u = 0; u < 2; ++u)
#pragma unroll
                    for (int r = 0; r < 4; ++r) mt = fmaxf(mt, sc[u][r]);
                mt = fmaxf(mt, __shfl_xor(mt, 16, 64));
                mt = fmaxf(mt, __shfl_xor(mt, 32, 64));
                const float mn = fmaxf(m2, mt);
                const float mref = (mn == -INFINITY) ? 0.f : mn;
                const float alpha = exp2f(m2 - mref);
                float psum = 0.f;
#pragma unroll
                for (int u = 0; u < 2; ++u)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float pv = exp2f(sc[u][r] - mref);
                        psum += pv;
                        sc[u][r] = drop_on ? pv * drop_mul(p.drop, (dhead + (uint64_t)min(qi, N - 1)) * (uint64_t)N +
                                                                       (uint64_t)min(16 * (ka + u) + 4 * g + r, N - 1))
                                           : pv;
                    }
                lsum = lsum * alpha + psum;
                m2 = mn;
                Frag<T> pf;
                acc_to_frag(pf, sc[0], sc[1]);
                const int kr = 16 * (ka - k0);
#pragma unroll
                for (int dt = 0; dt < ND; ++dt) {
                    oacc[dt] *= alpha;
                    Frag<T> vf;
                    col_frag(vf, Vs, ASTR, kr + 4 * g, kr + 16 + 4 * g, 16 * dt, lane);
                    oacc[dt] = mma(vf, pf, oacc[dt]);
                }
            }
        }
    }

    if (own) {
        float ls = lsum;
        ls += __shfl_xor(ls, 16, 64);
        ls += __shfl_xor(ls, 32, 64);
        if (qi < N) {
            const float inv = 1.f / ls;
            T* O = reinterpret_cast<T*>(p.out) + ((long)b * N + qi) * p.ldo + h * dh;
#pragma unroll
            for (int dt = 0; dt < ND; ++dt) {
                const int d = 16 * dt + 4 * g;
                if (d < dh)
                    *reinterpret_cast<bf16x4*>(O + d) =
                        bf16x4{(__bf16)(oacc[dt][0] * inv), (__bf16)(oacc[dt][1] * inv),
                               (__bf16)(oacc[dt][2] * inv), (__bf16)(oacc[dt][3] * inv)};
            }
            if (g == 0) p.lse[((long)b * p.H + h) * N + qi] = (m2 + log2f(ls)) * 0.6931471805599453f;
        }
    }
}

// ----------------------------------------------------------------------------------------- backward
// acc[tile][dt][r] = X^T[d = 16 dt + 4 g + r][row li]: sum over the wave's tiles and its 16 row lanes, then one LDS
// atomic per (group, d) into dst[d]
template <int TPW, int ND>
__device__ __forceinline__ void band_colsum(const f32x4 (&acc)[TPW][ND], float* dst, int g, int li) {
#pragma unroll
    for (int dt = 0; dt < ND; ++dt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float t = 0.f;
#pragma unroll
            for (int s2 = 0; s2 < TPW; ++s2) t += acc[s2][dt][r];
            t = row16_sum(t);
            if (li == 0) atomicAdd(dst + 16 * dt + 4 * g + r, t);
        }
}

// Backward of a head of ONE chunk (N <= 208: every ViT-224/16 configuration) in one kernel: pass 1 is the key-stationary
// sweep of attn_bwd_band_kv_kernel below, pass 2 the query-stationary one of attn_bwd_band_q_kernel, K staged over Q
// in between -- the workgroup reads back the dS tiles it has just written (L2-hot), and there is no second launch.
// Measured at 197 tokens, w = 20: 254 us against 212 + 84 us for the two kernels (which is what longer heads get).
template <int DW, bool BIAS>
__global__ __launch_bounds__(BAND_THREADS) void attn_bwd_band_kernel(const BandParams p) {
    using T = __bf16;
    using BC = BandCfg<DW, BIAS>;
    using C = typename BC::C;
    constexpr int DKB = BC::DKB;
    constexpr int ASTR = C::ASTR;
    constexpr int ND = BC::ND;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g = lane >> 4, li = lane & 15;
    const int N = p.N, dh = p.dh;
    const int NT = (N + 15) / 16;
    const int NR = 16 * NT + 16;
    const BandGroups grp(p.w, NT, BIAS, BC::GC);
    const int WR = 32 * grp.gc + 32;
    const int h = blockIdx.x, b = blockIdx.y;

    T* Qs = reinterpret_cast<T*>(smem);   // pass 1: Q rows; pass 2: K rows
    T* Ks = Qs;
    T* dOs = Qs + NR * ASTR;
    T* Pks = dOs + NR * ASTR;
    T* Pqs = Pks + (BIAS ? WR * ASTR : 0);
    float* lse2s = reinterpret_cast<float*>(Pqs + (BIAS ? WR * ASTR : 0));
    float* dls = lse2s + NR;
    float* scr = dls + NR;
    float* xs = scr + wave * (BC::WSCR_BYTES / 4);
    float* ys = xs + (BIAS ? SCR_FLOATS : 0);
    T* dsc = reinterpret_cast<T*>(ys + (BIAS ? SCR_FLOATS : 0));   // zero-padded dS copy (attn_common.h), bf16 as the MFMA takes it
    constexpr int CA = 16 * ND;                                    // floats per matrix in colacc
    float* colacc = scr + BAND_WAVES * (BC::WSCR_BYTES / 4);       // [3][CA]: column sums of dQ / dK / dV of this head (p.dbias)
    const bool want_bias = p.dbias != nullptr;
    if (want_bias && tid < 3 * CA) colacc[tid] = 0.f;
    // sum over the wave's tiles and its 16 key / query lanes, then one LDS atomic per (group, d)
    auto colsum_acc = [&](const f32x4 (&acc)[BAND_TPW][ND], float* dst) {
#pragma unroll
        for (int dt = 0; dt < ND; ++dt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float t = 0.f;
#pragma unroll
                for (int s2 = 0; s2 < BAND_TPW; ++s2) t += acc[s2][dt][r];
                t = row16_sum(t);
                if (li == 0) atomicAdd(dst + 16 * dt + 4 * g + r, t);
            }
    };

    const T* Q = reinterpret_cast<const T*>(p.q) + (long)b * N * p.ld + h * dh;
    const T* K = reinterpret_cast<const T*>(p.k) + (long)b * N * p.ld + h * dh;
    const T* V = reinterpret_cast<const T*>(p.v) + (long)b * N * p.ld + h * dh;
    const T* dO = reinterpret_cast<const T*>(p.dout) + (long)b * N * p.ldo + h * dh;
    const T* O = reinterpret_cast<const T*>(p.out) + (long)b * N * p.ldo + h * dh;
    const float* lse = p.lse + ((long)b * p.H + h) * N;
    float* delta = p.delta + ((long)b * p.H + h) * N;
    __bf16* ds_head = p.ds_ws + ((long)b * p.H + h) * (2 * grp.cmax + 1) * NT * 256;   // this head's dS tiles

    constexpr float LOG2E = 1.4426950408889634f;
    const float sc2 = p.scale * LOG2E;
    const bool drop_on = p.drop.thresh != 0;
    const uint64_t dhead = (uint64_t)(b * p.H + h) * (uint64_t)N;

    stage_rows<T, DKB, DW>(Qs, Q, p.ld, 0, NR, 0, N, dh, tid, BAND_THREADS);
    // dO rows go to LDS and, on the way, delta_i = dO_i . O_i is formed (the CPR lanes of a row
    // each hold one 16-byte chunk of dO and of O): no separate delta kernel, no second read of dO.
    // a row's chunks sit in LG adjacent lanes (LG = CPR rounded up to a power of two; lanes past CPR idle)
    constexpr int LG = C::CPR <= 4 ? 4 : (C::CPR <= 8 ? 8 : 16);
    for (int c = tid; c < NR * LG; c += BAND_THREADS) {
        const int r = c / LG, ch = c - r * LG;
        const int d = ch * C::EPC;
        bf16x8 v = {}, o = {};
        if (r < N && d < dh) {
            v = *reinterpret_cast<const bf16x8*>(dO + (long)r * p.ldo + d);
            o = *reinterpret_cast<const bf16x8*>(O + (long)r * p.ldo + d);
        }
        if (ch < C::CPR) *reinterpret_cast<bf16x8*>(dOs + r * ASTR + d) = v;
        float sdot = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) sdot += (float)v[e] * (float)o[e];
#pragma unroll
        for (int m = 1; m < LG; m <<= 1) sdot += __shfl_xor(sdot, m, 64);
        if (ch == 0) {
            dls[r] = sdot;
            if (r < N) delta[r] = sdot;
        }
    }
    for (int i = tid; i < NR; i += BAND_THREADS) lse2s[i] = i < N ? lse[i] * LOG2E : 0.f;

    // ---- pass 1: key-stationary, dK^T / dV^T of the wave's key tiles in registers ----------------
    if (BIAS) {   // Y-layout copy: everything outside rows 15..30 stays zero for the whole pass
        for (int i = lane; i < BAND_DS_ELEMS / 2; i += 64) reinterpret_cast<uint32_t*>(dsc)[i] = 0u;
    }
    {
        Frag<T> kf[BAND_TPW][DKB], vf[BAND_TPW][DKB];
        f32x4 dkacc[BAND_TPW][ND], dvacc[BAND_TPW][ND];
#pragma unroll
        for (int s = 0; s < BAND_TPW; ++s) {
            const int kj = 16 * (wave + BAND_WAVES * s) + li;
#pragma unroll
            for (int kb = 0; kb < DKB; ++kb) {
                row_frag_global<T>(kf[s][kb], K, p.ld, kj, N, dh, kb, lane);
                row_frag_global<T>(vf[s][kb], V, p.ld, kj, N, dh, kb, lane);
            }
#pragma unroll
            for (int dt = 0; dt < ND; ++dt) {
                dkacc[s][dt] = f32x4{0.f, 0.f, 0.f, 0.f};
                dvacc[s][dt] = f32x4{0.f, 0.f, 0.f, 0.f};
            }
        }
        for (int gi = 0; gi < grp.ngroups; ++gi) {
            const int c_lo = grp.c_lo(gi), c_hi = grp.c_hi(gi);
            if (BIAS) {
                if (gi > 0) __syncthreads();
                band_stage_tables<DKB, DW>(p, Pks, Pqs, WR, c_lo, h, tid, BAND_THREADS);
            }
            if (BIAS || gi == 0) __syncthreads();
#pragma unroll
            for (int s = 0; s < BAND_TPW; ++s) {
                const int kt = wave + BAND_WAVES * s;
                if (kt >= NT) continue;
                const int kj0 = 16 * kt;
                const int qlo = max(0, kt + c_lo), qhi = min(NT - 1, kt + c_hi);
                for (int qa = qlo; qa <= qhi; qa += 2) {
                    f32x4 pt[2], dst[2];
#pragma unroll
                    for (int u = 0; u < 2; ++u) {
                        const int qt = qa + u;
                        f32x4 pv = {0.f, 0.f, 0.f, 0.f}, dsv = {0.f, 0.f, 0.f, 0.f};
                        if (qt <= qhi) {
                            const int qi0 = 16 * qt;
                            const int rel = qi0 - kj0 - 16 * c_lo;
                            Frag<T> qf[DKB], dof[DKB];
#pragma unroll
                            for (int kb = 0; kb < DKB; ++kb) {
                                row_frag<T, DKB, DW>(qf[kb], Qs, qi0, kb, lane);
                                row_frag<T, DKB, DW>(dof[kb], dOs, qi0, kb, lane);
                            }
                            const f32x4 raw = score_tile<T, DKB, BIAS, false, DW>(qf, kf[s], Pks, Pqs, rel, xs, ys, lane);
                            f32x4 dp = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                            for (int kb = 0; kb < DKB; ++kb) dp = mma(dof[kb], vf[s][kb], dp);
#pragma unroll
                            for (int r = 0; r < 4; ++r) {
                                const int qi = qi0 + 4 * g + r, kj = kj0 + li;
                                const bool ok = qi < N && kj < N && (p.w < 0 || abs(qi - kj) <= p.w);
                                const float pe = ok ? exp2f(raw[r] * sc2 - lse2s[qi]) : 0.f;
                                // dropout: dV sees the dropped probabilities, dP = (dO V^T) * keep / (1 - p); delta is unchanged
                                const float dm = (drop_on && ok) ? drop_mul(p.drop, (dhead + (uint64_t)qi) * (uint64_t)N + (uint64_t)kj) : 1.f;
                                pv[r] = pe * dm;
                                dsv[r] = pe * (dp[r] * dm - dls[qi]) * p.scale;
                            }
                            // the dS tile is kept for pass 2 and for the table-gradient kernel:
                            // entry (b, h, c + cmax, kt), lane l holds dS[4 (l >> 4) + r][l & 15]
                            *reinterpret_cast<bf16x4*>(ds_head + ((long)(qt - kt + grp.cmax) * NT + kt) * 256 + lane * 4) =
                                bf16x4{(__bf16)dsv[0], (__bf16)dsv[1], (__bf16)dsv[2], (__bf16)dsv[3]};
                            if (BIAS) {   // dK^T[d][jj] += sum_t Pq^T[d][t] Sy[t][jj]
#pragma unroll
                                for (int r = 0; r < 4; ++r) dsc[(15 + 4 * g + r) * DSY_LD + li] = (__bf16)dsv[r];
                                wave_lds_sync();
                                Frag<T> sy;   // Sy[t = 8g + e][jj = li] = dS[li + t - 15][li]
#pragma unroll
                                for (int e = 0; e < 8; ++e) sy.v[e] = dsc[(li + 8 * g + e) * DSY_LD + li];
#pragma unroll
                                for (int dt = 0; dt < ND; ++dt) {
                                    Frag<T> a;
                                    col_frag(a, Pqs, ASTR, rel + 8 * g, rel + 8 * g + 4, 16 * dt, lane);
                                    dkacc[s][dt] = mma(a, sy, dkacc[s][dt]);
                                }
                                wave_lds_sync();
                            }
                        }
                        pt[u] = pv;
                        dst[u] = dsv;
                    }
                    Frag<T> bp, bs;
                    acc_to_frag(bp, pt[0], pt[1]);
                    acc_to_frag(bs, dst[0], dst[1]);
#pragma unroll
                    for (int dt = 0; dt < ND; ++dt) {
                        Frag<T> a;
                        col_frag(a, dOs, ASTR, 16 * qa + 4 * g, 16 * qa + 16 + 4 * g, 16 * dt, lane);
                        dvacc[s][dt] = mma(a, bp, dvacc[s][dt]);
                        col_frag(a, Qs, ASTR, 16 * qa + 4 * g, 16 * qa + 16 + 4 * g, 16 * dt, lane);
                        dkacc[s][dt] = mma(a, bs, dkacc[s][dt]);
                    }
                }
            }
        }
        if (want_bias) {     // tiles / keys past N hold zeros
            colsum_acc(dkacc, colacc + CA);
            colsum_acc(dvacc, colacc + 2 * CA);
        }
#pragma unroll
        for (int s = 0; s < BAND_TPW; ++s) {
            const int kj = 16 * (wave + BAND_WAVES * s) + li;
            if (kj < N) {
                T* dK = reinterpret_cast<T*>(p.dk) + ((long)b * N + kj) * p.ld_d + h * dh;
                T* dV = reinterpret_cast<T*>(p.dv) + ((long)b * N + kj) * p.ld_d + h * dh;
#pragma unroll
                for (int dt = 0; dt < ND; ++dt) {
                    const int d = 16 * dt + 4 * g;
                    if (d < dh) {
                        const f32x4 a = dkacc[s][dt], c = dvacc[s][dt];
                        *reinterpret_cast<bf16x4*>(dK + d) = bf16x4{(__bf16)a[0], (__bf16)a[1], (__bf16)a[2], (__bf16)a[3]};
                        *reinterpret_cast<bf16x4*>(dV + d) = bf16x4{(__bf16)c[0], (__bf16)c[1], (__bf16)c[2], (__bf16)c[3]};
                    }
                }
            }
        }
    }

    // ---- K replaces Q in LDS -------------------------------------------------------------------
    __syncthreads();
    if (BIAS) {   // X-layout copy for pass 2: everything outside columns 16..31 stays zero
        for (int i = lane; i < BAND_DS_ELEMS / 2; i += 64) reinterpret_cast<uint32_t*>(dsc)[i] = 0u;
    }
    stage_rows<T, DKB, DW>(Ks, K, p.ld, 0, NR, 0, N, dh, tid, BAND_THREADS);

    // ---- pass 2: query-stationary, dQ^T of the wave's query tiles in registers (lane = query) ----
    // dS is NOT recomputed: pass 1 left every live tile in the workspace (same workgroup, visible
    // after the barriers above); a lane reads its 4 keys of the transposed tile as 4 bf16 loads.
    {
        f32x4 dqacc[BAND_TPW][ND];
#pragma unroll
        for (int s = 0; s < BAND_TPW; ++s)
#pragma unroll
            for (int dt = 0; dt < ND; ++dt) dqacc[s][dt] = f32x4{0.f, 0.f, 0.f, 0.f};
        const int tr_off = ((li >> 2) * 16 + 4 * g) * 4 + (li & 3);   // dS[query li][key 4g + r] at tr_off + 4 r
        for (int gi = 0; gi < grp.ngroups; ++gi) {
            const int c_lo = grp.c_lo(gi), c_hi = grp.c_hi(gi);
            if (BIAS && grp.ngroups > 1) {   // a single group's rows are still in place from pass 1
                if (gi > 0) __syncthreads();
                band_stage_tables<DKB, DW>(p, Pks, Pqs, WR, c_lo, h, tid, BAND_THREADS);
            }
            if ((BIAS && grp.ngroups > 1) || gi == 0) __syncthreads();
#pragma unroll
            for (int s = 0; s < BAND_TPW; ++s) {
                const int qt = wave + BAND_WAVES * s;
                if (qt >= NT) continue;
                const int qi0 = 16 * qt;
                const int klo = max(0, qt - c_hi), khi = min(NT - 1, qt - c_lo);
                // the tile values of the NEXT key-tile pair are asked for before the current pair is worked on (they
                // come back from L2: one round trip per pair otherwise).  Loads without a condition -- a tile past khi
                // reads tile khi and is zeroed -- so that the compiler can count them (vmcnt) instead of waiting for all.
                auto ds_pair = [&](int ka, __bf16 (&raw)[2][4]) {
#pragma unroll
                    for (int u = 0; u < 2; ++u) {
                        const int kt = min(ka + u, khi);
                        const __bf16* e = ds_head + ((long)(qt - kt + grp.cmax) * NT + kt) * 256 + tr_off;
#pragma unroll
                        for (int r = 0; r < 4; ++r) raw[u][r] = e[4 * r];
                    }
                };
                __bf16 cur[2][4], nxt[2][4];
                if (klo <= khi) ds_pair(klo, cur);
                for (int ka = klo; ka <= khi; ka += 2) {
                    ds_pair(min(ka + 2, khi), nxt);
                    f32x4 dst[2];
#pragma unroll
                    for (int u = 0; u < 2; ++u) {
                        const int kt = ka + u;
                        f32x4 dsv = {0.f, 0.f, 0.f, 0.f};
                        if (kt <= khi) {
#pragma unroll
                            for (int r = 0; r < 4; ++r) dsv[r] = (float)cur[u][r];
                            if (BIAS) {   // dQ^T[d][ii] += sum_t Pk^T[d][t] Sx[t][ii]
                                const int rel = qi0 - 16 * kt - 16 * c_lo;
                                *reinterpret_cast<bf16x4*>(dsc + li * DSX_LD + 16 + 4 * g) =
                                    bf16x4{(__bf16)dsv[0], (__bf16)dsv[1], (__bf16)dsv[2], (__bf16)dsv[3]};
                                wave_lds_sync();
                                Frag<T> sx;   // Sx[t = 8g + e][ii = li] = dS^T[li][li + 15 - t]
#pragma unroll
                                for (int ee = 0; ee < 8; ++ee) sx.v[ee] = dsc[li * (DSX_LD + 1) + 31 - 8 * g - ee];
#pragma unroll
                                for (int dt = 0; dt < ND; ++dt) {
                                    Frag<T> a;
                                    col_frag(a, Pks, ASTR, rel + 8 * g, rel + 8 * g + 4, 16 * dt, lane);
                                    dqacc[s][dt] = mma(a, sx, dqacc[s][dt]);
                                }
                                wave_lds_sync();
                            }
                        }
                        dst[u] = dsv;
                    }
                    // content: dQ^T[d][q] += K^T[d][key] dS^T[key][q], 32 keys per MFMA
                    Frag<T> bs;
                    acc_to_frag(bs, dst[0], dst[1]);
#pragma unroll
                    for (int dt = 0; dt < ND; ++dt) {
                        Frag<T> a;
                        col_frag(a, Ks, ASTR, 16 * ka + 4 * g, 16 * ka + 16 + 4 * g, 16 * dt, lane);
                        dqacc[s][dt] = mma(a, bs, dqacc[s][dt]);
                    }
#pragma unroll
                    for (int u = 0; u < 2; ++u)
#pragma unroll
                        for (int r = 0; r < 4; ++r) cur[u][r] = nxt[u][r];
                }
            }
        }
        if (want_bias) colsum_acc(dqacc, colacc);
#pragma unroll
        for (int s = 0; s < BAND_TPW; ++s) {
            const int qi = 16 * (wave + BAND_WAVES * s) + li;
            if (qi < N) {
                T* dQ = reinterpret_cast<T*>(p.dq) + ((long)b * N + qi) * p.ld_d + h * dh;
#pragma unroll
                for (int dt = 0; dt < ND; ++dt) {
                    const int d = 16 * dt + 4 * g;
                    if (d < dh) {
                        const f32x4 a = dqacc[s][dt];
                        *reinterpret_cast<bf16x4*>(dQ + d) = bf16x4{(__bf16)a[0], (__bf16)a[1], (__bf16)a[2], (__bf16)a[3]};
                    }
                }
            }
        }
    }
    if (want_bias) {     // one contiguous 256-byte atomic per matrix and head
        __syncthreads();
        if (tid < 3 * CA) {
            const int which = tid / CA, d = tid - which * CA;
            if (d < dh) atomicAdd(p.dbias + (long)which * p.H * dh + h * dh + d, colacc[tid]);
        }
    }
}


// dK / dV: grid (H, B, key chunks), key-stationary.  A wave owns up to two key tiles of the chunk (K / V fragments
// from global memory, dK^T / dV^T in registers) and sweeps the query tiles its band reaches, chunk image by chunk
// image (Q and dO rows; delta_i = dO_i . O_i is formed while dO is staged).  Every live 16 x 16 dS tile (bf16) is left
// in the workspace for the dQ kernel and the table-gradient kernel: no second recompute of P, no atomics.
template <int DW, bool BIAS>
__global__ __launch_bounds__(BAND_THREADS) void attn_bwd_band_kv_kernel(const BandParams p) {
    using T = __bf16;
    using BC = BandCfg<DW, BIAS>;
    using C = typename BC::C;
    constexpr int DKB = BC::DKB;
    constexpr int ASTR = C::ASTR;
    constexpr int ND = BC::ND;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g = lane >> 4, li = lane & 15;
    const int N = p.N, dh = p.dh;
    const int NT = (N + 15) / 16;
    const int NRC = 16 * min(NT, BAND_CH) + 16;
    const BandGroups grp(p.w, NT, BIAS, BC::GC);
    const int WR = 32 * grp.gc + 32;
    const int h = blockIdx.x, b = blockIdx.y, kc = blockIdx.z;
    const int kt0 = kc * BAND_CH, kt1 = min(NT, kt0 + BAND_CH);      // this workgroup's key tiles

    T* Qs = reinterpret_cast<T*>(smem);
    T* dOs = Qs + NRC * ASTR;
    T* Pks = dOs + NRC * ASTR;
    T* Pqs = Pks + (BIAS ? WR * ASTR : 0);
    float* lse2s = reinterpret_cast<float*>(Pqs + (BIAS ? WR * ASTR : 0));
    float* dls = lse2s + NRC;
    float* scr = dls + NRC;
    float* xs = scr + wave * (BC::WSCR_BYTES / 4);
    float* ys = xs + (BIAS ? SCR_FLOATS : 0);
    T* dsc = reinterpret_cast<T*>(ys + (BIAS ? SCR_FLOATS : 0));   // zero-padded dS copy (attn_common.h), bf16 as the MFMA takes it
    constexpr int CA = 16 * ND;                                    // floats per matrix in colacc
    float* colacc = scr + BAND_WAVES * (BC::WSCR_BYTES / 4);       // [2][CA]: column sums of dK / dV of this chunk (p.dbias)
    const bool want_bias = p.dbias != nullptr;
    if (want_bias && tid < 2 * CA) colacc[tid] = 0.f;

    const T* Q = reinterpret_cast<const T*>(p.q) + (long)b * N * p.ld + h * dh;
    const T* K = reinterpret_cast<const T*>(p.k) + (long)b * N * p.ld + h * dh;
    const T* V = reinterpret_cast<const T*>(p.v) + (long)b * N * p.ld + h * dh;
    const T* dO = reinterpret_cast<const T*>(p.dout) + (long)b * N * p.ldo + h * dh;
    const T* O = reinterpret_cast<const T*>(p.out) + (long)b * N * p.ldo + h * dh;
    const float* lse = p.lse + ((long)b * p.H + h) * N;
    float* delta = p.delta + ((long)b * p.H + h) * N;
    __bf16* ds_head = p.ds_ws + ((long)b * p.H + h) * (2 * grp.cmax + 1) * NT * 256;   // this head's dS tiles

    constexpr float LOG2E = 1.4426950408889634f;
    const float sc2 = p.scale * LOG2E;
    const bool drop_on = p.drop.thresh != 0;
    const uint64_t dhead = (uint64_t)(b * p.H + h) * (uint64_t)N;

    if (BIAS) {   // Y-layout copy: everything outside rows 15..30 stays zero for the whole kernel
        for (int i = lane; i < BAND_DS_ELEMS / 2; i += 64) reinterpret_cast<uint32_t*>(dsc)[i] = 0u;
    }
    Frag<T> kf[BAND_TPW][DKB], vf[BAND_TPW][DKB];
    f32x4 dkacc[BAND_TPW][ND], dvacc[BAND_TPW][ND];
#pragma unroll
    for (int s = 0; s < BAND_TPW; ++s) {
        const int kt = kt0 + wave + BAND_WAVES * s;
        const int kj = kt < kt1 ? 16 * kt + li : N;
#pragma unroll
        for (int kb = 0; kb < DKB; ++kb) {
            row_frag_global<T>(kf[s][kb], K, p.ld, kj, N, dh, kb, lane);
            row_frag_global<T>(vf[s][kb], V, p.ld, kj, N, dh, kb, lane);
        }
#pragma unroll
        for (int dt = 0; dt < ND; ++dt) {
            dkacc[s][dt] = f32x4{0.f, 0.f, 0.f, 0.f};
            dvacc[s][dt] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
    }

    // query chunks the band of this key chunk reaches
    const int qt_min = max(0, kt0 - grp.cmax), qt_max = min(NT - 1, kt1 - 1 + grp.cmax);
    bool first = true;
    for (int qc = qt_min / BAND_CH; qc <= qt_max / BAND_CH; ++qc) {
        const int q0 = qc * BAND_CH, rq0 = 16 * q0;        // first query tile / row of the chunk
        if (!first) __syncthreads();
        stage_rows<T, DKB, DW>(Qs, Q, p.ld, rq0, NRC, 0, N, dh, tid, BAND_THREADS);
        // dO rows go to LDS and, on the way, delta_i = dO_i . O_i is formed: a row's chunks sit in LG adjacent lanes
        // (LG = CPR rounded up to a power of two; lanes past CPR idle)
        constexpr int LG = C::CPR <= 4 ? 4 : (C::CPR <= 8 ? 8 : 16);
        for (int c = tid; c < NRC * LG; c += BAND_THREADS) {
            const int r = c / LG, ch = c - r * LG;
            const int d = ch * C::EPC;
            const int gr = rq0 + r;
            bf16x8 v = {}, o = {};
            if (gr < N && d < dh) {
                v = *reinterpret_cast<const bf16x8*>(dO + (long)gr * p.ldo + d);
                o = *reinterpret_cast<const bf16x8*>(O + (long)gr * p.ldo + d);
            }
            if (ch < C::CPR) *reinterpret_cast<bf16x8*>(dOs + r * ASTR + d) = v;
            float sdot = 0.f;
#pragma unroll
            for (int e = 0; e < 8; ++e) sdot += (float)v[e] * (float)o[e];
#pragma unroll
            for (int m = 1; m < LG; m <<= 1) sdot += __shfl_xor(sdot, m, 64);
            if (ch == 0) {
                dls[r] = sdot;
                if (gr < N && qc == kc && r < 16 * BAND_CH) delta[gr] = sdot;   // once per row: the diagonal chunk is always in the band
            }
        }
        for (int i = tid; i < NRC; i += BAND_THREADS) lse2s[i] = rq0 + i < N ? lse[rq0 + i] * LOG2E : 0.f;

        for (int gi = 0; gi < grp.ngroups; ++gi) {
            const int c_lo = grp.c_lo(gi), c_hi = grp.c_hi(gi);
            if (BIAS) {
                if (!first) __syncthreads();
                band_stage_tables<DKB, DW>(p, Pks, Pqs, WR, c_lo, h, tid, BAND_THREADS);
            }
            if (BIAS || gi == 0) __syncthreads();
            first = false;
#pragma unroll
            for (int s = 0; s < BAND_TPW; ++s) {
                const int kt = kt0 + wave + BAND_WAVES * s;
                if (kt >= kt1) continue;
                const int kj0 = 16 * kt;
                const int qlo = max(q0, max(0, kt + c_lo)), qhi = min(min(NT - 1, q0 + BAND_CH - 1), kt + c_hi);
                for (int qa = qlo; qa <= qhi; qa += 2) {
                    f32x4 pt[2], dst[2];
#pragma unroll
                    for (int u = 0; u < 2; ++u) {
                        const int qt = qa + u;
                        f32x4 pv = {0.f, 0.f, 0.f, 0.f}, dsv = {0.f, 0.f, 0.f, 0.f};
                        if (qt <= qhi) {
                            const int qi0 = 16 * qt, ql0 = qi0 - rq0;      // global / image row of the tile
                            const int rel = qi0 - kj0 - 16 * c_lo;
                            Frag<T> qf[DKB], dof[DKB];
#pragma unroll
                            for (int kb = 0; kb < DKB; ++kb) {
                                row_frag<T, DKB, DW>(qf[kb], Qs, ql0, kb, lane);
                                row_frag<T, DKB, DW>(dof[kb], dOs, ql0, kb, lane);
                            }
                            const f32x4 raw = score_tile<T, DKB, BIAS, false, DW>(qf, kf[s], Pks, Pqs, rel, xs, ys, lane);
                            f32x4 dp = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                            for (int kb = 0; kb < DKB; ++kb) dp = mma(dof[kb], vf[s][kb], dp);
#pragma unroll
                            for (int r = 0; r < 4; ++r) {
                                const int qi = qi0 + 4 * g + r, kj = kj0 + li;
                                const int ql = ql0 + 4 * g + r;
                                const bool ok = qi < N && kj < N && (p.w < 0 || abs(qi - kj) <= p.w);
                                const float pe = ok ? exp2f(raw[r] * sc2 - lse2s[ql]) : 0.f;
                                // dropout: dV sees the dropped probabilities, dP = (dO V^T) * keep / (1 - p); delta is unchanged
                                const float dm = (drop_on && ok) ? drop_mul(p.drop, (dhead + (uint64_t)qi) * (uint64_t)N + (uint64_t)kj) : 1.f;
                                pv[r] = pe * dm;
                                dsv[r] = pe * (dp[r] * dm - dls[ql]) * p.scale;
                            }
                            // the dS tile is kept for the dQ kernel and for the table-gradient kernel:
                            // entry (b, h, c + cmax, kt), lane l holds dS[4 (l >> 4) + r][l & 15]
                            *reinterpret_cast<bf16x4*>(ds_head + ((long)(qt - kt + grp.cmax) * NT + kt) * 256 + lane * 4) =
                                bf16x4{(__bf16)dsv[0], (__bf16)dsv[1], (__bf16)dsv[2], (__bf16)dsv[3]};
                            if (BIAS) {   // dK^T[d][jj] += sum_t Pq^T[d][t] Sy[t][jj]
#pragma unroll
                                for (int r = 0; r < 4; ++r) dsc[(15 + 4 * g + r) * DSY_LD + li] = (__bf16)dsv[r];
                                wave_lds_sync();
                                Frag<T> sy;   // Sy[t = 8g + e][jj = li] = dS[li + t - 15][li]
#pragma unroll
                                for (int e = 0; e < 8; ++e) sy.v[e] = dsc[(li + 8 * g + e) * DSY_LD + li];
#pragma unroll
                                for (int dt = 0; dt < ND; ++dt) {
                                    Frag<T> a;
                                    col_frag(a, Pqs, ASTR, rel + 8 * g, rel + 8 * g + 4, 16 * dt, lane);
                                    dkacc[s][dt] = mma(a, sy, dkacc[s][dt]);
                                }
                                wave_lds_sync();
                            }
                        }
                        pt[u] = pv;
                        dst[u] = dsv;
                    }
                    Frag<T> bp, bs;
                    acc_to_frag(bp, pt[0], pt[1]);
                    acc_to_frag(bs, dst[0], dst[1]);
                    const int qr = 16 * qa - rq0;
#pragma unroll
                    for (int dt = 0; dt < ND; ++dt) {
                        Frag<T> a;
                        col_frag(a, dOs, ASTR, qr + 4 * g, qr + 16 + 4 * g, 16 * dt, lane);
                        dvacc[s][dt] = mma(a, bp, dvacc[s][dt]);
                        col_frag(a, Qs, ASTR, qr + 4 * g, qr + 16 + 4 * g, 16 * dt, lane);
                        dkacc[s][dt] = mma(a, bs, dkacc[s][dt]);
                    }
                }
            }
        }
    }
    if (want_bias) {     // tiles / keys past N hold zeros
        band_colsum<BAND_TPW, ND>(dkacc, colacc, g, li);
        band_colsum<BAND_TPW, ND>(dvacc, colacc + CA, g, li);
    }
#pragma unroll
    for (int s = 0; s < BAND_TPW; ++s) {
        const int kt = kt0 + wave + BAND_WAVES * s;
        const int kj = 16 * kt + li;
        if (kt < kt1 && kj < N) {
            T* dK = reinterpret_cast<T*>(p.dk) + ((long)b * N + kj) * p.ld_d + h * dh;
            T* dV = reinterpret_cast<T*>(p.dv) + ((long)b * N + kj) * p.ld_d + h * dh;
#pragma unroll
            for (int dt = 0; dt < ND; ++dt) {
                const int d = 16 * dt + 4 * g;
                if (d < dh) {
                    const f32x4 a = dkacc[s][dt], c = dvacc[s][dt];
                    *reinterpret_cast<bf16x4*>(dK + d) = bf16x4{(__bf16)a[0], (__bf16)a[1], (__bf16)a[2], (__bf16)a[3]};
                    *reinterpret_cast<bf16x4*>(dV + d) = bf16x4{(__bf16)c[0], (__bf16)c[1], (__bf16)c[2], (__bf16)c[3]};
                }
            }
        }
    }
    if (want_bias) {     // one contiguous segment per matrix, head and chunk
        __syncthreads();
        if (tid < 2 * CA) {
            const int which = tid / CA, d = tid - which * CA;
            if (d < dh) atomicAdd(p.dbias + (long)(1 + which) * p.H * dh + h * dh + d, colacc[tid]);
        }
    }
}

// dQ: grid (H, B, query chunks), query-stationary, 16 waves (wave < 13 owns query tile 13 * chunk + wave, dQ^T in
// registers).  dS is NOT recomputed: the dK / dV kernel left every live tile in the workspace; a lane reads its 4 keys
// of the transposed tile as 4 bf16 loads.  Needs only the K image (and the Pk rows with the bias): half the LDS of the
// other kernel, so two workgroups -- 32 waves -- share a CU.
template <int DW, bool BIAS>
__global__ __launch_bounds__(BAND_FWD_THREADS) void attn_bwd_band_q_kernel(const BandParams p) {
    using T = __bf16;
    using BC = BandCfg<DW, BIAS>;
    using C = typename BC::C;
    constexpr int DKB = BC::DKB;
    constexpr int ASTR = C::ASTR;
    constexpr int ND = BC::ND;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g = lane >> 4, li = lane & 15;
    const int N = p.N, dh = p.dh;
    const int NT = (N + 15) / 16;
    const int NRC = 16 * min(NT, BAND_CH) + 16;
    const BandGroups grp(p.w, NT, BIAS, BC::GC);
    const int WR = 32 * grp.gc + 32;
    const int h = blockIdx.x, b = blockIdx.y, qc = blockIdx.z;

    T* Ks = reinterpret_cast<T*>(smem);
    T* Pks = Ks + NRC * ASTR;
    T* dsc = Pks + (BIAS ? WR * ASTR : 0) + min(wave, BAND_CH - 1) * (BIAS ? BAND_DS_ELEMS : 0);
    constexpr int CA = 16 * ND;
    float* colacc = reinterpret_cast<float*>(Pks + (BIAS ? WR * ASTR : 0) + min(NT, BAND_CH) * (BIAS ? BAND_DS_ELEMS : 0));
    const bool want_bias = p.dbias != nullptr;
    if (want_bias && tid < CA) colacc[tid] = 0.f;

    const T* K = reinterpret_cast<const T*>(p.k) + (long)b * N * p.ld + h * dh;
    const __bf16* ds_head = p.ds_ws + ((long)b * p.H + h) * (2 * grp.cmax + 1) * NT * 256;

    const int qt = qc * BAND_CH + wave;
    const bool own = wave < BAND_CH && qt < NT;
    const int qi0 = 16 * qt;
    if (BIAS && own) {   // X-layout copy: everything outside columns 16..31 stays zero
        for (int i = lane; i < BAND_DS_ELEMS / 2; i += 64) reinterpret_cast<uint32_t*>(dsc)[i] = 0u;
    }
    f32x4 dqacc[1][ND];
#pragma unroll
    for (int dt = 0; dt < ND; ++dt) dqacc[0][dt] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int tr_off = ((li >> 2) * 16 + 4 * g) * 4 + (li & 3);   // dS[query li][key 4g + r] at tr_off + 4 r

    const int kt_min = max(0, qc * BAND_CH - grp.cmax), kt_max = min(NT - 1, qc * BAND_CH + BAND_CH - 1 + grp.cmax);
    bool first = true;
    for (int kc = kt_min / BAND_CH; kc <= kt_max / BAND_CH; ++kc) {
        const int k0 = kc * BAND_CH;
        if (!first) __syncthreads();
        stage_rows<T, DKB, DW>(Ks, K, p.ld, 16 * k0, NRC, 0, N, dh, tid, BAND_FWD_THREADS);
        for (int gi = 0; gi < grp.ngroups; ++gi) {
            const int c_lo = grp.c_lo(gi), c_hi = grp.c_hi(gi);
            if (BIAS) {
                if (!first) __syncthreads();
                band_stage_tables<DKB, DW>(p, Pks, nullptr, WR, c_lo, h, tid, BAND_FWD_THREADS);
            }
            if (BIAS || gi == 0) __syncthreads();
            first = false;
            if (!own) continue;
            const int klo = max(k0, max(0, qt - c_hi)), khi = min(min(NT - 1, k0 + BAND_CH - 1), qt - c_lo);
            // The tiles come from HBM / a remote L2 (another kernel wrote them): up to NB of them are fetched at once,
            // so a wave has NB round trips in flight instead of one per tile pair.
            constexpr int NB = 6;
            for (int kb0 = klo; kb0 <= khi; kb0 += NB) {
                __bf16 raw[NB][4];
#pragma unroll
                for (int t = 0; t < NB; ++t) {
                    const int kt = kb0 + t;
                    const __bf16* e = ds_head + ((long)(qt - min(kt, khi) + grp.cmax) * NT + min(kt, khi)) * 256 + tr_off;
#pragma unroll
                    for (int r = 0; r < 4; ++r) raw[t][r] = e[4 * r];
                }
#pragma unroll
                for (int t2 = 0; t2 < NB; t2 += 2) {
                    const int ka = kb0 + t2;
                    if (ka > khi) break;
                    f32x4 dst[2];
#pragma unroll
                    for (int u = 0; u < 2; ++u) {
                        const int kt = ka + u;
                        f32x4 dsv = {0.f, 0.f, 0.f, 0.f};
                        if (kt <= khi) {
#pragma unroll
                            for (int r = 0; r < 4; ++r) dsv[r] = (float)raw[t2 + u][r];
                            if (BIAS) {   // dQ^T[d][ii] += sum_t Pk^T[d][t] Sx[t][ii]
                                const int rel = qi0 - 16 * kt - 16 * c_lo;
                                *reinterpret_cast<bf16x4*>(dsc + li * DSX_LD + 16 + 4 * g) =
                                    bf16x4{raw[t2 + u][0], raw[t2 + u][1], raw[t2 + u][2], raw[t2 + u][3]};
                                wave_lds_sync();
                                Frag<T> sx;   // Sx[t = 8g + e][ii = li] = dS^T[li][li + 15 - t]
#pragma unroll
                                for (int ee = 0; ee < 8; ++ee) sx.v[ee] = dsc[li * (DSX_LD + 1) + 31 - 8 * g - ee];
#pragma unroll
                                for (int dt = 0; dt < ND; ++dt) {
                                    Frag<T> a;
                                    col_frag(a, Pks, ASTR, rel + 8 * g, rel + 8 * g + 4, 16 * dt, lane);
                                    dqacc[0][dt] = mma(a, sx, dqacc[0][dt]);
                                }
                                wave_lds_sync();
                            }
                        }
                        dst[u] = dsv;
                    }
                    // content: dQ^T[d][q] += K^T[d][key] dS^T[key][q], 32 keys per MFMA
                    Frag<T> bs;
                    acc_to_frag(bs, dst[0], dst[1]);
                    const int kr = 16 * (ka - k0);
#pragma unroll
                    for (int dt = 0; dt < ND; ++dt) {
                        Frag<T> a;
                        col_frag(a, Ks, ASTR, kr + 4 * g, kr + 16 + 4 * g, 16 * dt, lane);
                        dqacc[0][dt] = mma(a, bs, dqacc[0][dt]);
                    }
                }
            }
        }
    }
    if (own) {
        if (want_bias) band_colsum<1, ND>(dqacc, colacc, g, li);
        const int qi = qi0 + li;
        if (qi < N) {
            T* dQ = reinterpret_cast<T*>(p.dq) + ((long)b * N + qi) * p.ld_d + h * dh;
#pragma unroll
            for (int dt = 0; dt < ND; ++dt) {
                const int d = 16 * dt + 4 * g;
                if (d < dh) {
                    const f32x4 a = dqacc[0][dt];
                    *reinterpret_cast<bf16x4*>(dQ + d) = bf16x4{(__bf16)a[0], (__bf16)a[1], (__bf16)a[2], (__bf16)a[3]};
                }
            }
        }
    }
    if (want_bias) {
        __syncthreads();
        if (tid < CA && tid < dh) atomicAdd(p.dbias + h * dh + tid, colacc[tid]);
    }
}

// ------------------------------------------------------------------------------------------- host
constexpr int BAND_MAX_TOKENS = 16 * BAND_CH * 3;      // three chunks: covers the 577 tokens of 384 px / 16

bool band_eligible(int dtype, int N, int dh, bool bias, int band_w) {
    if (dtype != VITK_BF16 || N > BAND_MAX_TOKENS || dh > 80 || dh % 8 != 0) return false;
    (void)bias;
    (void)band_w;   // any band, with or without the bias (diagonal groups)
    return true;
}
// every instantiation fits the 160 KiB of a CU at the largest chunk it is dispatched for (13 tiles, full diagonal groups)
static_assert(2 * 224 * 88 * 2 + 2 * 64 * 88 * 2 + 13 * 2 * SCR_FLOATS * 4 <= 160 * 1024, "forward, 80-wide head");
static_assert(2 * 224 * 88 * 2 + 2 * 64 * 88 * 2 + 2 * 224 * 4 + BAND_WAVES * (2 * SCR_FLOATS * 4 + BAND_DS_ELEMS * 2) + 2 * 80 * 4 <= 160 * 1024, "dK / dV, 80-wide head");
static_assert(2 * 224 * 72 * 2 + 2 * 96 * 72 * 2 + 2 * 224 * 4 + BAND_WAVES * (2 * SCR_FLOATS * 4 + BAND_DS_ELEMS * 2) + 2 * 64 * 4 <= 160 * 1024, "dK / dV, 64-wide head");
static_assert(224 * 88 * 2 + 64 * 88 * 2 + 13 * BAND_DS_ELEMS * 2 + 80 * 4 <= 80 * 1024, "dQ, 80-wide head: two workgroups per CU");
static_assert(224 * 72 * 2 + 96 * 72 * 2 + 13 * BAND_DS_ELEMS * 2 + 64 * 4 <= 80 * 1024, "dQ, 64-wide head: two workgroups per CU");

// per (kernel, device), grows monotonically, thread-safe (runtime.hip)
template <typename Kern> static void band_set_lds(Kern kern, size_t lds) { (void)vitk_lds_optin((const void*)kern, lds); }

template <int DW, bool BIAS> static void band_fwd_launch(const BandParams& p, int NT, int cmax, hipStream_t st) {
    const size_t lds = BandCfg<DW, BIAS>::fwd_bytes(NT, cmax);
    band_set_lds(attn_fwd_band_kernel<DW, BIAS>, lds);
    const int chunks = (NT + BAND_CH - 1) / BAND_CH;
    hipLaunchKernelGGL((attn_fwd_band_kernel<DW, BIAS>), dim3(p.H, p.B, chunks), dim3(BAND_FWD_THREADS), lds, st, p);
}
template <int DW, bool BIAS> static void band_bwd_launch(const BandParams& p, int NT, int cmax, hipStream_t st) {
    const size_t lds_kv = BandCfg<DW, BIAS>::bwd_kv_bytes(NT, cmax), lds_q = BandCfg<DW, BIAS>::bwd_q_bytes(NT, cmax);
    band_set_lds(attn_bwd_band_kv_kernel<DW, BIAS>, lds_kv);
    band_set_lds(attn_bwd_band_q_kernel<DW, BIAS>, lds_q);
    const int chunks = (NT + BAND_CH - 1) / BAND_CH;
    if (chunks == 1) {
        const size_t lds = BandCfg<DW, BIAS>::bwd_bytes(NT, cmax);
        band_set_lds(attn_bwd_band_kernel<DW, BIAS>, lds);
        hipLaunchKernelGGL((attn_bwd_band_kernel<DW, BIAS>), dim3(p.H, p.B), dim3(BAND_THREADS), lds, st, p);
        return;
    }
    hipLaunchKernelGGL((attn_bwd_band_kv_kernel<DW, BIAS>), dim3(p.H, p.B, chunks), dim3(BAND_THREADS), lds_kv, st, p);
    hipLaunchKernelGGL((attn_bwd_band_q_kernel<DW, BIAS>), dim3(p.H, p.B, chunks), dim3(BAND_FWD_THREADS), lds_q, st, p);
}

bool lai_eligible(const BandParams& p, bool backward);
void lai_launch(const BandParams& p, bool backward, hipStream_t st);
bool full_eligible(const BandParams& p, bool backward);
void full_launch(const BandParams& p, bool backward, hipStream_t st);

// Returns true when a backward launch has already produced the table gradients (p.dpk / p.dpq).
bool band_launch(const BandParams& p, bool backward, hipStream_t st) {
    if (full_eligible(p, backward)) {    // unmasked + disentangled: whole score rows per lane, Bias / P / dS in one LDS array (attention_lai.hip)
        full_launch(p, backward, st);
        return false;                    // (backward: the table gradients are left to attn_bwd_relds_kernel, from the dS tiles)
    }
    if (lai_eligible(p, backward)) {     // banded (with or without the disentangled tables), one chunk, head width 64 (attention_lai.hip)
        lai_launch(p, backward, st);
        return backward;
    }
    const int NT = (p.N + 15) / 16;
    int cmax = p.w >= 0 ? (p.w + 15) / 16 : NT - 1;
    if (cmax > NT - 1) cmax = NT - 1;
    const bool bias = p.pk != nullptr;
#define VITK_BAND(D, BI)                                          \
    do {                                                          \
        if (backward) band_bwd_launch<D, BI>(p, NT, cmax, st);    \
        else band_fwd_launch<D, BI>(p, NT, cmax, st);             \
    } while (0)
    if (p.dh <= 32) {
        if (bias) VITK_BAND(32, true); else VITK_BAND(32, false);
    } else if (p.dh <= 64) {
        if (bias) VITK_BAND(64, true); else VITK_BAND(64, false);
    } else {      // heads of 72 / 80 (ViT-L/16's 1280 / 16): 80-wide row images, the last K block half empty
        if (bias) VITK_BAND(80, true); else VITK_BAND(80, false);
    }
#undef VITK_BAND
    return false;
}

}  // namespace vitk
