// Banded ("locality-aware") disentangled attention for one-chunk heads of width 64: the kernels of the headline model
// (ViT-B/16-224 LAI + disentangled: 197 tokens, 16 heads of 64, LAI half widths 8 / 12 / 18 / 20) and of every other
// DisLAI stage with a band -- DisentangledLAISelfAttention.forward + disentangled_attention_bias
// (disentangled_transformer/disentangled_transformer.py:851-998), mask = get_attn_mask (:1000-1010).
//
//   raw[i][j] = Q_i.K_j + Q_i.Pk[i-j] + K_j.Pq[i-j],   P = softmax_j(scale * raw | |i-j| <= w)
//
// With a band of half width w only the 2w+1 relative rows t = i - j + w in [0, 2w] are ever touched, so the two
// position terms are two SMALL matrices per head,
//     C2P[i][t] = Q_i . Pk_b[t]      P2C[j][t] = K_j . Pq_b[t]          (Pk_b[t] = Pk[t - w], [N x (2w+1)] each),
// made once per head on the matrix cores by the wave that owns the row tile (3 MFMA tiles of 16 table rows), kept in LDS
// in fp32, and read back along the diagonals: score (i, j) adds C2P[i][i-j+w] + P2C[j][i-j+w].  The general kernels
// (attention_band.hip) form these products per 16 x 16 tile PAIR through a per-wave scratch with four intra-wave LDS
// hand-offs per pair; here every score tile of a row is independent work (5 tiles of a row in flight at once).
// Backward: dS along the same diagonals IS the gradient of the two small matrices,
//     dC2P[i][t] = dS[i][i-t+w]       dP2C[j][t] = dS[j+t-w][j]
// (a one-to-one map inside the band: no accumulation), kept in LDS as bf16 rows of 48; then
//     dQ_i  += sum_t dC2P[i][t] Pk_b[t]      dK_j  += sum_t dP2C[j][t] Pq_b[t]           (own rows, per wave)
//     dPk_b[t] = sum_i dC2P[i][t] Q_i        dPq_b[t] = sum_j dP2C[j][t] K_j             (12 output tiles over 12 waves)
// are plain MFMA products over K-major LDS reads -- the table-gradient kernel (attn_bwd_relds_kernel) and the dS
// workspace in global memory are not used.  A workgroup walks several batch elements of ONE head and keeps its table
// gradients in registers across them: one flush of 2 (2w+1) x 64 atomics per workgroup.
//
// 16 waves; wave < NT owns query tile `wave` (lane = query, scores transposed) in the query-stationary phase and key tile
// `wave` in the key-stationary one (which recomputes P and dS from the LDS-resident C2P / P2C: 4 MFMAs per tile).
// Data movement: a workgroup belongs to ONE head h and walks batch elements; the two 48-row table images are staged once.
// Per batch element every wave loads the 16 rows of its own tile as MFMA row fragments (16 bytes per lane) and writes the
// row images the other waves read (K, V; later Q, dO; later K again) FROM those registers: every global byte is read once
// per phase and nothing is staged twice.  (First version: each wave fetched its twelve table fragments from L2 and the
// images were staged by separate loads -- 256 KB through a CU's vector-memory pipe per head, 7.5 of a head's 13 us.)
// LDS at 197 tokens, w = 20 (backward): C2P + P2C 64.6 KiB, two row images 58.5 KiB (K, V; then Q, dO; then K again),
// dC2P / dP2C 19.5 KiB (one slot, used in turn), two table images 13.5 KiB, lse / delta rows, column sums: 158.5 KiB.
#include "attn_common.h"

namespace vitk {

constexpr int LAI_WAVES = 16;
constexpr int LAI_THREADS = 64 * LAI_WAVES;
constexpr int LAI_MAXW = 20;          // 2w+1 <= 41 table rows: three 16-row MFMA tiles, 48-element rows of the dS copies
constexpr int LAI_MAXNT = 13;         // one chunk (208 rows)
constexpr int LAI_ASTR = 72;          // row stride of a 64-wide bf16 image (ACfg<__bf16, 2, 64>::ASTR)
constexpr int LAI_TLD = 48;           // elements per row of dC2P / dP2C; rows of the staged table image
static_assert(LAI_ASTR == ACfg<__bf16, 2, 64>::ASTR, "row images are staged with stage_rows<__bf16, 2, 64>");
// Timing experiments (tools/lai_phases.sh): a build with -DVITK_LAI_STOP=k leaves every head after phase k (wrong results)
#ifndef VITK_LAI_STOP
#define VITK_LAI_STOP 0
#endif

// Workgroup barrier that orders LDS traffic only.  __syncthreads() also waits for every global load and store of the
// wave (s_waitcnt vmcnt(0)): the next element's prefetched rows and the output stores would be waited for at each barrier
// -- a memory round trip per phase.  Nothing in these kernels passes through global memory between waves.
__device__ __forceinline__ void lai_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

constexpr int LAI_LDB = 42;          // floats per C2P / P2C row, whatever the band (2w+1 <= 41): a compile-time row stride
                                      // makes every diagonal read an LDS instruction with an immediate offset; even, so that
                                      // both diagonal walks (strides LDB + 1 and LDB - 1) step through the banks oddly
struct LaiGeom {
    int NT, NR, TW;
};
__host__ __device__ inline LaiGeom lai_geom(int N, int w) {
    LaiGeom G;
    G.NT = (N + 15) / 16;
    G.NR = 16 * G.NT;
    G.TW = 2 * w + 1;                 // live table rows
    return G;
}
// C2P / P2C hold N rows (not NR: the 11 rows saved per array at 197 tokens are what lets both table images stay resident)
// (without tables -- plain banded attention, AttentionLAI of timm_locality_aware_transformer.py:25-79 -- only the two row
// images and the lse / delta rows are there)
static size_t lai_fwd_bytes(int N, int w, bool bias) {
    const LaiGeom G = lai_geom(N, w);
    return (size_t)2 * G.NR * LAI_ASTR * 2 + (bias ? (size_t)2 * LAI_TLD * LAI_ASTR * 2 + (size_t)2 * N * LAI_LDB * 4 : 0);
}
static size_t lai_bwd_bytes(int N, int w, bool bias) {
    const LaiGeom G = lai_geom(N, w);
    return (size_t)2 * G.NR * LAI_ASTR * 2 + (size_t)2 * G.NR * 4 + 3 * 64 * 4 +
           (bias ? (size_t)G.NR * LAI_TLD * 2 + (size_t)2 * LAI_TLD * LAI_ASTR * 2 + (size_t)2 * N * LAI_LDB * 4 : 0);
}

// 16 bytes of row `row` of a [nrows][ld] matrix at column d; rows outside [0, nrows) read as zero.  The load itself is
// UNCONDITIONAL (clamped address, select afterwards): a load under a branch gets a wait of its own, and a phase of a dozen
// such loads becomes a dozen memory round trips in a row.
__device__ __forceinline__ u32x4 lai_load16(const __bf16* __restrict__ gm, long ld, int row, int nrows, int d) {
    const int rc = min(max(row, 0), nrows - 1);
    const u32x4 v = *reinterpret_cast<const u32x4*>(gm + (long)rc * ld + d);
    const bool ok = row >= 0 && row < nrows;
    return u32x4{ok ? v[0] : 0u, ok ? v[1] : 0u, ok ? v[2] : 0u, ok ? v[3] : 0u};
}
// the two K blocks of row `row` as MFMA row fragments (lane (g, li): its row, columns 32 kb + 8 g .. + 7)
__device__ __forceinline__ void lai_row_frags(Frag<__bf16> (&f)[2], const __bf16* __restrict__ gm, long ld, int row,
                                              int nrows, int lane) {
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
        f[kb].v = __builtin_bit_cast(bf16x8, lai_load16(gm, ld, row, nrows, 32 * kb + 8 * (lane >> 4)));
}
// ... and into the row image the other waves read (row = this lane's row of the wave's own tile)
__device__ __forceinline__ void lai_put_rows(__bf16* img, const Frag<__bf16> (&f)[2], int row, int lane) {
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) *reinterpret_cast<bf16x8*>(img + row * LAI_ASTR + 32 * kb + 8 * (lane >> 4)) = f[kb].v;
}
// Row-image reads from per-lane constants (made once per kernel, not per head: the address arithmetic of the generic
// row_frag / col_frag helpers, redone per head and tile, was half of these kernels' VALU work -- and they are VALU-bound):
//   rowb = li * ASTR + 8 g                      row fragment of tile T, block kb:  img[rowb + T * 16 ASTR + 32 kb]
//   colb = (4 g + (li >> 2)) * ASTR + 4 (li & 3)  K-major fragment over the rows 16 Ta + 4g.. (first half) and 16 Tb + 4g..
//                                                 (second half), columns 16 dt..:   img[colb + T * 16 ASTR + 16 dt]
__device__ __forceinline__ void lai_rfrag(Frag<__bf16>& f, const __bf16* img, int rowb, int tile, int kb) {
    frag_load(f, img + rowb + tile * (16 * LAI_ASTR) + 32 * kb);
}
__device__ __forceinline__ void lai_cfrag(Frag<__bf16>& f, const __bf16* img, int colb, int ta, int tb, int dt) {
    const __bf16* a0 = img + colb + ta * (16 * LAI_ASTR) + 16 * dt;
    const __bf16* a1 = img + colb + tb * (16 * LAI_ASTR) + 16 * dt;
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(uintptr_t)a0);
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(uintptr_t)a1);
    const s16x8 r = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    f.v = __builtin_bit_cast(bf16x8, r);
}

// both 48-row table images, once per workgroup: row t <-> table row tab_r0 + t for t < TW, zero rows behind
__device__ __forceinline__ void lai_stage_tables(__bf16* tabk, __bf16* tabq, const __bf16* PK, const __bf16* PQ, long ld_p,
                                                 int tab_r0, int TW, int nrows, int tid) {
    if (tid < 2 * LAI_TLD * 8) {
        const bool second = tid >= LAI_TLD * 8;
        const int c = second ? tid - LAI_TLD * 8 : tid;
        const int r = c >> 3, d = (c & 7) * 8;
        const u32x4 v = lai_load16(second ? PQ : PK, ld_p, r < TW ? tab_r0 + r : -1, nrows, d);
        *reinterpret_cast<u32x4*>((second ? tabq : tabk) + r * LAI_ASTR + d) = v;
    }
}

// C2P / P2C rows of the tile this wave owns: X^T[t][i] = Pk_b[t] . Q_i and Y^T[t][j] = Pq_b[t] . K_j, three 16-row table
// tiles each, table fragments from the resident LDS images.
__device__ __forceinline__ void lai_bias_rows(const __bf16* tabk, const __bf16* tabq, const LaiGeom& G,
                                              const Frag<__bf16> (&qf)[2], const Frag<__bf16> (&kf)[2], float* c2p_row,
                                              float* p2c_row, bool row_ok, int lane) {
    const int g = lane >> 4;
#pragma unroll
    for (int u = 0; u < 3; ++u) {
        f32x4 x = {0.f, 0.f, 0.f, 0.f}, y = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
            Frag<__bf16> a;
            row_frag<__bf16, 2, 64>(a, tabk, 16 * u, kb, lane);
            x = mma(a, qf[kb], x);                    // X^T[t = 16u + 4g + r][i = li]
            row_frag<__bf16, 2, 64>(a, tabq, 16 * u, kb, lane);
            y = mma(a, kf[kb], y);                    // Y^T[t][j = li]
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int tt = 16 * u + 4 * g + r;
            if (row_ok && tt < LAI_LDB) {
                c2p_row[tt] = x[r];
                p2c_row[tt] = y[r];
            }
        }
        __builtin_amdgcn_sched_barrier(0);            // one table tile at a time: with all 24 fragment reads hoisted to the
                                                      // front the backward kernel spills 100 registers in this phase
    }
}

// ------------------------------------------------------------------------------------------ forward
// Element (ci, r) of a lane in the QUERY-stationary layout (lane = query i = 16 qt + li; key j = 16 (qt + CM - ci) + 4g + r):
//   t = i - j + w = T0 + 16 ci - r,  T0 = li - 4g - 16 CM + w
//   C2P[i][t]  = pc[16 ci + 3 - r]                           pc = C2P + i LDB + T0 - 3
//   P2C[j][t]  = pp[(NC-1-ci) (16 LDB - 16) + r (LDB - 1)]   pp = P2C + (16 (qt - CM) + 4g) LDB + T0 + 16 (NC-1)
// -- two per-lane pointers and compile-time offsets.  Elements outside the band, the sequence or the live tiles are read
// all the same (the addresses stay inside the workgroup's LDS: P2C has the images in front of it and C2P behind it) and
// discarded through a per-lane bit mask made once per kernel (bit 4 ci + r).
template <int NC>
__device__ __forceinline__ uint32_t lai_mask_q(int qt, int li, int g, int N, int NT, int w) {
    constexpr int CM = (NC - 1) / 2;
    uint32_t m = 0;
    const int i = 16 * qt + li;
#pragma unroll
    for (int ci = 0; ci < NC; ++ci)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int kt = qt + CM - ci, j = 16 * kt + 4 * g + r, t = i - j + w;
            const bool ok = kt >= 0 && kt < NT && j < N && i < N && t >= 0 && t <= 2 * w;
            m |= ok ? (1u << (4 * ci + r)) : 0u;
        }
    return m;
}
// ... and in the KEY-stationary layout (lane = key j = 16 kt + li; query i = 16 (kt + ci - CM) + 4g + r):
//   t = T0c + 16 ci + r,  T0c = 4g - li - 16 CM + w
//   P2C[j][t] = pp[16 ci + r]                           pp = P2C + j LDB + T0c
//   C2P[i][t] = pc[ci (16 LDB + 16) + r (LDB + 1)]      pc = C2P + (16 (kt - CM) + 4g) LDB + T0c
template <int NC>
__device__ __forceinline__ uint32_t lai_mask_k(int kt, int li, int g, int N, int NT, int w) {
    constexpr int CM = (NC - 1) / 2;
    uint32_t m = 0;
    const int j = 16 * kt + li;
#pragma unroll
    for (int ci = 0; ci < NC; ++ci)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int qt = kt + ci - CM, i = 16 * qt + 4 * g + r, t = i - j + w;
            const bool ok = qt >= 0 && qt < NT && i < N && j < N && t >= 0 && t <= 2 * w;
            m |= ok ? (1u << (4 * ci + r)) : 0u;
        }
    return m;
}

// grid (H, groups): the workgroup walks batch elements b = group, group + groups, ... of head h; the next element's rows
// are requested before the current one is worked on.  NC = live tile diagonals of a row: 3 (w <= 16) or 5.
// BIAS = false: no relative tables (plain banded attention): the same kernel without C2P / P2C.
template <int NC, bool DROP, bool BIAS>
__global__ __launch_bounds__(LAI_THREADS) void attn_fwd_lai_kernel(const BandParams p, int groups) {
    using T = __bf16;
    constexpr int ASTR = LAI_ASTR, LDB = LAI_LDB, CM = (NC - 1) / 2, NP = (NC + 1) / 2;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g = lane >> 4, li = lane & 15;
    const int N = p.N, w = p.w;
    const LaiGeom G = lai_geom(N, w);
    const int h = blockIdx.x;

    T* Ks = reinterpret_cast<T*>(smem);
    T* Vs = Ks + G.NR * ASTR;
    T* tabk = Vs + G.NR * ASTR;
    T* tabq = tabk + LAI_TLD * ASTR;
    float* P2C = reinterpret_cast<float*>(tabq + LAI_TLD * ASTR);
    float* C2P = P2C + N * LDB;

    const T* PK = reinterpret_cast<const T*>(p.pk) + h * 64;
    const T* PQ = reinterpret_cast<const T*>(p.pq) + h * 64;
    const long hb = (long)N * p.ld;                  // elements between batch elements in q / k / v
    const T* Qh = reinterpret_cast<const T*>(p.q) + h * 64;
    const T* Kh = reinterpret_cast<const T*>(p.k) + h * 64;
    const T* Vh = reinterpret_cast<const T*>(p.v) + h * 64;

    const int qt = wave;
    const bool own = qt < G.NT;
    const int qi = 16 * qt + li;
    const int fr = own ? qi : -1;
    const int qin = min(qi, N - 1);
    constexpr float LOG2E = 1.4426950408889634f;
    const float sc2 = p.scale * LOG2E;
    // per-lane constants of the loop below
    const int rowb = li * ASTR + 8 * g, colb = (4 * g + (li >> 2)) * ASTR + 4 * (li & 3);
    const int T0 = li - 4 * g - 16 * CM + w;
    const float* pc = C2P + qin * LDB + T0 - 3;                  // (BIAS only: never dereferenced without tables)
    const float* pp = P2C + (16 * (qt - CM) + 4 * g) * LDB + T0 + 16 * (NC - 1);
    const uint32_t okm = own ? lai_mask_q<NC>(qt, li, g, N, G.NT, w) : 0u;
    int ktc[NC];
#pragma unroll
    for (int ci = 0; ci < NC; ++ci) ktc[ci] = min(max(qt + CM - ci, 0), G.NT - 1);

    if (BIAS) lai_stage_tables(tabk, tabq, PK, PQ, p.ld_p, N - 1 - w, G.TW, 2 * N - 1, tid);
    Frag<T> nq[2], nk[2], nv[2];
    {
        const long o = (long)blockIdx.y * hb;
        lai_row_frags(nq, Qh + o, p.ld, fr, N, lane);
        lai_row_frags(nk, Kh + o, p.ld, fr, N, lane);
        lai_row_frags(nv, Vh + o, p.ld, fr, N, lane);
    }
    lai_barrier();
    if (VITK_LAI_STOP == 1) return;

    for (int b = blockIdx.y; b < p.B; b += groups) {
        Frag<T> qf[2];
        if (own) {
            qf[0] = nq[0];
            qf[1] = nq[1];
            lai_put_rows(Ks, nk, qi, lane);
            lai_put_rows(Vs, nv, qi, lane);
            if (BIAS) lai_bias_rows(tabk, tabq, G, qf, nk, C2P + qin * LDB, P2C + qin * LDB, qi < N, lane);
        }
        auto prefetch = [&]() {   // the next batch element's rows (the last one asks for its own again: no branch around the loads)
            const long o = (long)min(b + groups, p.B - 1) * hb;
            lai_row_frags(nq, Qh + o, p.ld, fr, N, lane);
            lai_row_frags(nk, Kh + o, p.ld, fr, N, lane);
            lai_row_frags(nv, Vh + o, p.ld, fr, N, lane);
        };
        if (!DROP) prefetch();    // (with dropout the 24 registers they land in are needed for the mask hashes: asked for below)
        lai_barrier();
        if (own && VITK_LAI_STOP != 2) {
            const uint64_t dhead = (uint64_t)(b * p.H + h) * (uint64_t)N;
            f32x4 sc[NC];
            float m = -INFINITY;
#pragma unroll
            for (int ci = 0; ci < NC; ++ci) {
                f32x4 s = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int kb = 0; kb < 2; ++kb) {
                    Frag<T> kf;
                    lai_rfrag(kf, Ks, rowb, ktc[ci], kb);
                    s = mma(kf, qf[kb], s);      // raw[query li][key 4g + r]
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float v = (BIAS ? s[r] + pc[16 * ci + 3 - r] + pp[(NC - 1 - ci) * (16 * LDB - 16) + r * (LDB - 1)] : s[r]) * sc2;
                    sc[ci][r] = (okm >> (4 * ci + r)) & 1u ? v : -INFINITY;
                    m = fmaxf(m, sc[ci][r]);
                }
            }
            m = fmaxf(m, __shfl_xor(m, 16, 64));
            m = fmaxf(m, __shfl_xor(m, 32, 64));
            const float mref = (m == -INFINITY) ? 0.f : m;
            float ls = 0.f;
            const uint64_t drow = DROP ? (dhead + (uint64_t)qin) * (uint64_t)N : 0;      // this query's row of the dropout stream
#pragma unroll
            for (int ci = 0; ci < NC; ++ci) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float pv = __builtin_amdgcn_exp2f(sc[ci][r] - mref);
                    ls += pv;
                    // dropout (reference :915): the normaliser keeps every probability, P V sees the dropped ones; element
                    // ((b H + h) N + i) N + j of the stream as in the other attention kernels (a masked element is 0 anyway)
                    sc[ci][r] = DROP ? pv * drop_mul(p.drop, drow + (uint64_t)(uint32_t)(16 * (qt + CM - ci) + 4 * g + r)) : pv;
                }
                if (DROP) __builtin_amdgcn_sched_barrier(0);      // one tile's hashes at a time (20 interleaved chains spill)
            }
            ls += __shfl_xor(ls, 16, 64);
            ls += __shfl_xor(ls, 32, 64);
            if (VITK_LAI_STOP == 3) {
                if (qi < N && g == 0) p.lse[((long)b * p.H + h) * N + qi] = m + ls;
            } else {
                f32x4 oacc[4];
#pragma unroll
                for (int dt = 0; dt < 4; ++dt) oacc[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int pi = 0; pi < NP; ++pi) {
                    constexpr f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
                    const int ca = 2 * pi, cb = 2 * pi + 1 < NC ? 2 * pi + 1 : 2 * pi;
                    Frag<T> pf;
                    acc_to_frag(pf, sc[ca], 2 * pi + 1 < NC ? sc[cb] : zero4);
#pragma unroll
                    for (int dt = 0; dt < 4; ++dt) {
                        Frag<T> vf;
                        lai_cfrag(vf, Vs, colb, ktc[ca], ktc[cb], dt);
                        oacc[dt] = mma(vf, pf, oacc[dt]);     // O^T[d = 16 dt + 4g + r][query li]
                    }
                }
                if (qi < N) {
                    const float inv = __builtin_amdgcn_rcpf(ls);
                    T* O = reinterpret_cast<T*>(p.out) + ((long)b * N + qi) * p.ldo + h * 64;
#pragma unroll
                    for (int dt = 0; dt < 4; ++dt)
                        *reinterpret_cast<bf16x4*>(O + 16 * dt + 4 * g) =
                            bf16x4{(__bf16)(oacc[dt][0] * inv), (__bf16)(oacc[dt][1] * inv), (__bf16)(oacc[dt][2] * inv),
                                   (__bf16)(oacc[dt][3] * inv)};
                    if (g == 0)
                        p.lse[((long)b * p.H + h) * N + qi] = (m + __builtin_amdgcn_logf(ls)) * 0.6931471805599453f;
                }
            }
        }
        if (DROP) prefetch();
        lai_barrier();            // everybody is done with the images and P2C before the next element's rows land in them
    }
}

// ---------------------------------------------------------------------------- forward, unmasked stage
// The whole head without a band (w < 0: config 3, and the last LAI stage of every DisLAI run): every query sees all N
// keys and all 2N-1 relative rows are live.  Same ownership as above (wave = query tile, lane = query, a workgroup walks the
// batch elements of one head), but
//   * a whole score ROW lives in registers: 13 tiles x 4 scores per lane, one softmax, no online rescaling;
//   * the two position terms are summed ONCE per head into one bf16 array  Bias[i][j] = C2P[i][i-j] + P2C[j][i-j]
//     ([N][212], 83.5 KiB at 197 tokens) that the score loop reads with ONE 8-byte LDS load per tile and lane.
// Bias is made on the matrix cores in strips: query tile qt needs the table rows t = i - j + N - 1 in [16 qt, 16 qt + N + 14]
// -- 14 ALIGNED 16-row table tiles -- and key tile kt a 14-tile strip of the other table; X^T[t][i] = Pk[t] . Q_i (lane =
// query i, registers = 4 consecutive t) is scattered to Bias[i][i - t + N - 1] as four 2-byte LDS stores, then, after a
// barrier, Y^T[t][j] = Pq[t] . K_j is added at Bias[t + j - N + 1][j] (each element has exactly one writer per phase).
// The tables do not fit beside it (2 x 56 KiB): their tiles pass through two 4-tile buffers, 7 rounds per table, every wave
// taking the tiles of its strip as they come by (one workgroup barrier per round; the next round's rows are in flight
// from L2 meanwhile).  attention_band.hip forms the same products per 16 x 16 tile PAIR (8 MFMAs, 4 LDS stores, 8 LDS
// loads and two intra-wave hand-offs per pair): 264 us per layer at the headline shape against 196 here -- of which the Bias
// strips are ~90 (their 2-byte scatter / read-modify-write LDS traffic, 2184 wave instructions per head, not the MFMAs),
// the score loop + P V ~80, row loads, staging and the 16 barriers the rest (tools/lai_phases.sh with
// VITK_STOP_MACRO=VITK_FULL_STOP).
// Balancing the rounds and dropping the range tests of interior tiles changed nothing: the strips are bound by the
// latency chain load - MFMA - MFMA - convert - (load - add) - store of each tile with four waves per SIMD.
#ifndef VITK_FULL_STOP
#define VITK_FULL_STOP 0             // timing experiments (tools/lai_phases.sh): bit 0 skips the Bias products, bit 1 the score loop, bit 2 P V
#endif
constexpr int FULL_LDJ = 212;                  // Bias row: 13 x 16 key columns + pad (8-byte aligned rows, odd multiple of 4 banks)
constexpr int FULL_TPR = 4;                    // table tiles per round
constexpr int FULL_TBUF = FULL_TPR * 16 * LAI_ASTR;      // elements of one table buffer
static size_t full_fwd_bytes(int N) {
    const LaiGeom G = lai_geom(N, 0);
    return (size_t)2 * G.NR * LAI_ASTR * 2 + (size_t)2 * FULL_TBUF * 2 + (size_t)N * FULL_LDJ * 2;
}

// BIAS = false: no relative tables (the unmasked stage of the timm LAI ViT): no Bias array, no table rounds.
template <bool DROP, bool BIAS>
__global__ __launch_bounds__(LAI_THREADS) void attn_fwd_full_kernel(const BandParams p, int groups) {
    using T = __bf16;
    constexpr int ASTR = LAI_ASTR, LDJ = FULL_LDJ, NTC = LAI_MAXNT, NP = (NTC + 1) / 2;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 4, li = lane & 15;
    const int N = p.N;
    const LaiGeom G = lai_geom(N, 0);
    const int NT = G.NT;
    const int h = blockIdx.x;

    T* Ks = reinterpret_cast<T*>(smem);
    T* Vs = Ks + G.NR * ASTR;
    T* tbuf = Vs + G.NR * ASTR;                     // two table buffers
    T* Bias = tbuf + 2 * FULL_TBUF;                 // [N][LDJ]

    const T* PK = reinterpret_cast<const T*>(p.pk) + h * 64;
    const T* PQ = reinterpret_cast<const T*>(p.pq) + h * 64;
    const long hb = (long)N * p.ld;
    const T* Qh = reinterpret_cast<const T*>(p.q) + h * 64;
    const T* Kh = reinterpret_cast<const T*>(p.k) + h * 64;
    const T* Vh = reinterpret_cast<const T*>(p.v) + h * 64;

    const int qt = wave;                             // query tile AND key tile this wave owns
    const bool own = qt < NT;
    const int qi = 16 * qt + li;
    const int fr = own ? qi : -1;
    const int qin = min(qi, N - 1);
    constexpr float LOG2E = 1.4426950408889634f;
    const float sc2 = p.scale * LOG2E;
    const int rowb = li * ASTR + 8 * g, colb = (4 * g + (li >> 2)) * ASTR + 4 * (li & 3);
    const int R = 2 * N - 1;                         // table rows
    constexpr int NRND = (2 * LAI_MAXNT + FULL_TPR - 1) / FULL_TPR;      // 7 rounds per table (tiles past the table: zero rows)
    // table tiles of this wave's strips: t = i - j + N - 1 over its 16 queries and all keys / its 16 keys and all queries
    const int lo_q = qt, hi_q = (16 * qt + 14 + N) >> 4;
    const int lo_k = max(0, N - 16 - 16 * qt) >> 4, hi_k = max(0, 2 * N - 2 - 16 * qt) >> 4;
    // Staging: a round's 4 x 16 table rows are 512 pieces of 16 bytes.  Waves 0-7 carry the even rounds, waves 8-15 the odd
    // ones; a thread's seven pieces of a head are the SAME for every batch element, so it walks them as a cycle and keeps
    // the next FOUR in flight / in registers at all times (also through the score loop): a round never waits for L2 -- with
    // one round in flight every round took a memory round trip (21 of a head's 30 us).
    const int grp = wave >> 3;
    const int st_row = (tid & 511) >> 3, st_d = (tid & 7) * 8;   // row 0..63 of the round, column
    auto table_piece = [&](int k) -> u32x4 {                     // piece k of this thread: round 2k + grp
        const int rnd = 2 * k + grp;                             // rounds 0 .. NRND-1: Pk, NRND .. 2 NRND-1: Pq
        const bool second = rnd >= NRND;
        // round rho of a table carries its tiles rho, rho + 7, rho + 14, rho + 21: every 14-tile strip has exactly two
        // of them, so every wave has two tiles of work in every round (with four CONSECUTIVE tiles per round the middle
        // rounds had all waves on four tiles and the outer ones almost none: 49 against 3 wave-tiles)
        const int r0 = 16 * ((second ? rnd - NRND : rnd) + NRND * (st_row >> 4)) + (st_row & 15);
        // rows past the table: the last row again, NOT zeros -- a select on the loaded value makes the compiler wait for the
        // load where it is issued (a memory round trip per round: 0.65 us x 28); those rows only ever meet dS / Bias
        // entries that the range tests leave out
        return *reinterpret_cast<const u32x4*>((second ? PQ : PK) + (long)min(r0, R - 1) * p.ld_p + st_d);
    };
    T* const st_dst = tbuf + grp * FULL_TBUF + st_row * ASTR + st_d;      // even rounds -> buffer 0, odd rounds -> buffer 1

    Frag<T> nq[2], nk[2], nv[2];
    {
        const long o = (long)blockIdx.y * hb;
        lai_row_frags(nq, Qh + o, p.ld, fr, N, lane);
        lai_row_frags(nk, Kh + o, p.ld, fr, N, lane);
        lai_row_frags(nv, Vh + o, p.ld, fr, N, lane);
    }
    u32x4 q0 = {}, q1 = {}, q2 = {}, q3 = {};
    if (BIAS) {
        q0 = table_piece(0);
        q1 = table_piece(1);
        q2 = table_piece(2);
        q3 = table_piece(3);
    }
    int knext = 4;               // the piece asked for next
    auto stage_next = [&]() {    // this thread's next piece -> its buffer; ask for the one four ahead
        *reinterpret_cast<u32x4*>(st_dst) = q0;
        q0 = q1;
        q1 = q2;
        q2 = q3;
        q3 = table_piece(knext);
        knext = knext == NRND - 1 ? 0 : knext + 1;
    };

    for (int b = blockIdx.y; b < p.B; b += groups) {
        Frag<T> qf[2], kf[2];
        qf[0] = nq[0];
        qf[1] = nq[1];
        kf[0] = nk[0];
        kf[1] = nk[1];
        if (own) {
            lai_put_rows(Ks, nk, qi, lane);
            lai_put_rows(Vs, nv, qi, lane);
        }
        if (BIAS && grp == 0) stage_next();          // round 0 -> buffer 0
        lai_barrier();
        // ---- Bias: 2 NRND rounds of 4 table tiles ---------------------------------------------------------------
        // per-lane bases; every element of a round is at a compile-time offset from them.  A query / key past the sequence
        // gets a poisoned index: none of its range tests passes, and its writes go to a pad column of its own row
        const int jb = qi < N ? qi + N - 1 - 4 * g : -(1 << 20);           // j of (tt = 0, r = 0)
        const int ib = qi < N ? qi - (N - 1) + 4 * g : -(1 << 20);         // i of (tt = 0, r = 0)
        T* const c2p_b = Bias + qin * LDJ + (qin + N - 1 - 4 * g) - 3;     // &Bias[i][j of (tt = 0, r = 3)]
        T* const p2c_b = Bias + (qin - (N - 1) + 4 * g) * LDJ + qin;       // &Bias[i of (tt = 0, r = 0)][j]
        T* const padp = Bias + qin * LDJ + 16 * LAI_MAXNT + g;
        auto bias_round = [&](int rnd, const T* tb) {
            if (!own || (VITK_FULL_STOP & 1)) return;
            const bool second = rnd >= NRND;
            const int rho = second ? rnd - NRND : rnd;           // tiles rho + 7 u
            if (!second) {
                T* cb = c2p_b - 16 * rho;
                const int j0 = jb - 16 * rho;
#pragma unroll
                for (int u = 0; u < FULL_TPR; ++u) {
                    const int tt = rho + NRND * u;
                    if (tt < lo_q || tt > hi_q) continue;        // wave-uniform
                    f32x4 x = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int kb = 0; kb < 2; ++kb) {
                        Frag<T> a;
                        frag_load(a, tb + u * (16 * ASTR) + rowb + 32 * kb);
                        x = mma(a, qf[kb], x);           // X^T[t = 16 tt + 4g + r][query li]: Bias[i][i - t + N - 1] = C2P
                    }
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        T* e = (unsigned)(j0 - 16 * NRND * u - r) < (unsigned)N ? cb + (3 - r - 16 * NRND * u) : padp;
                        *e = (T)x[r];
                    }
                }
            } else {
                T* pb = p2c_b + 16 * rho * LDJ;
                const int i0 = ib + 16 * rho;
#pragma unroll
                for (int u = 0; u < FULL_TPR; ++u) {
                    const int tt = rho + NRND * u;
                    if (tt < lo_k || tt > hi_k) continue;
                    f32x4 x = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int kb = 0; kb < 2; ++kb) {
                        Frag<T> a;
                        frag_load(a, tb + u * (16 * ASTR) + rowb + 32 * kb);
                        x = mma(a, kf[kb], x);           // Y^T[t][key li]: Bias[t + j - N + 1][j] += P2C
                    }
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        T* e = (unsigned)(i0 + 16 * NRND * u + r) < (unsigned)N ? pb + (16 * NRND * u + r) * LDJ : padp;
                        *e = (T)((float)*e + x[r]);
                    }
                }
            }
        };
#pragma nounroll
        for (int it = 0; BIAS && it < NRND; ++it) {
            bias_round(2 * it, tbuf);
            if (grp == 1) stage_next();                          // round 2 it + 1 -> buffer 1 (last read in round 2 it - 1)
            lai_barrier();
            bias_round(2 * it + 1, tbuf + FULL_TBUF);
            if (grp == 0 && it + 1 < NRND) stage_next();         // round 2 it + 2 -> buffer 0
            lai_barrier();
        }
        // ---- scores of the whole row, softmax, P V -----------------------------------------------------------------
        if (own && !(VITK_FULL_STOP & 2)) {
            const uint64_t dhead = (uint64_t)(b * p.H + h) * (uint64_t)N;
            const T* brow = Bias + qin * LDJ + 4 * g;
            f32x4 sc[NTC];
            float m = -INFINITY;
#pragma unroll
            for (int kt = 0; kt < NTC; ++kt) {
                const int ktc = min(kt, NT - 1);
                f32x4 s = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int kb = 0; kb < 2; ++kb) {
                    Frag<T> kr;
                    lai_rfrag(kr, Ks, rowb, ktc, kb);
                    s = mma(kr, qf[kb], s);          // raw[query li][key 16 kt + 4g + r]
                }
                bf16x4 b4 = {};
                if (BIAS) b4 = *reinterpret_cast<const bf16x4*>(brow + 16 * kt);
                if (16 * kt + 16 <= N) {             // (scalar branch) a tile inside the sequence: no mask
#pragma unroll
                    for (int r = 0; r < 4; ++r) sc[kt][r] = (s[r] + (float)b4[r]) * sc2;
                } else {
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        sc[kt][r] = 16 * kt + 4 * g + r < N ? (s[r] + (float)b4[r]) * sc2 : -INFINITY;
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) m = fmaxf(m, sc[kt][r]);
            }
            m = fmaxf(m, __shfl_xor(m, 16, 64));
            m = fmaxf(m, __shfl_xor(m, 32, 64));
            float ls = 0.f;
            const uint64_t drow = DROP ? (dhead + (uint64_t)qin) * (uint64_t)N : 0;
            // (made per head: as invariants of the batch loop the 52 clamped key indices of the dropout stream are hoisted
            // out of it, held through everything and spilled -- 125 registers)
            int g4 = 4 * g;
            if (DROP) asm volatile("" : "+v"(g4));
#pragma unroll
            for (int kt = 0; kt < NTC; ++kt) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float pv = __builtin_amdgcn_exp2f(sc[kt][r] - m);
                    ls += pv;
                    sc[kt][r] = pv;
                }
            }
            ls += __shfl_xor(ls, 16, 64);
            ls += __shfl_xor(ls, 32, 64);
            f32x4 oacc[4];
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) oacc[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (VITK_FULL_STOP & 4) {
#pragma unroll
                for (int kt = 0; kt < NTC; ++kt) oacc[kt & 3] += sc[kt];
            } else
#pragma unroll
            for (int pi = 0; pi < NP; ++pi) {
                constexpr f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
                const int ca = 2 * pi, cb = 2 * pi + 1 < NTC ? 2 * pi + 1 : 2 * pi;
                f32x4 pa = sc[ca], pb = 2 * pi + 1 < NTC ? sc[cb] : zero4;
                if (DROP) {
                    // dropout (reference :915) on the way into P V: the normaliser above keeps every probability; element
                    // ((b H + h) N + i) N + j of the stream as in the other attention kernels.  One tile pair's hashes at a
                    // time (hashed in the exp loop, beside 52 live scores, the kernel spilled 153 registers)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        pa[r] *= drop_mul(p.drop, drow + (uint64_t)(uint32_t)min(16 * ca + g4 + r, N - 1));
                        pb[r] *= drop_mul(p.drop, drow + (uint64_t)(uint32_t)min(16 * cb + g4 + r, N - 1));
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
                Frag<T> pf;
                acc_to_frag(pf, pa, pb);
                const int ka = min(ca, NT - 1), kb2 = min(cb, NT - 1);
#pragma unroll
                for (int dt = 0; dt < 4; ++dt) {
                    Frag<T> vf;
                    lai_cfrag(vf, Vs, colb, ka, kb2, dt);
                    oacc[dt] = mma(vf, pf, oacc[dt]);     // O^T[d = 16 dt + 4g + r][query li]
                }
            }
            // the next element's rows: asked for once the score registers are free
            {
                const long o = (long)min(b + groups, p.B - 1) * hb;
                lai_row_frags(nq, Qh + o, p.ld, fr, N, lane);
                lai_row_frags(nk, Kh + o, p.ld, fr, N, lane);
                lai_row_frags(nv, Vh + o, p.ld, fr, N, lane);
            }
            if (qi < N) {
                const float inv = __builtin_amdgcn_rcpf(ls);
                T* O = reinterpret_cast<T*>(p.out) + ((long)b * N + qi) * p.ldo + h * 64;
#pragma unroll
                for (int dt = 0; dt < 4; ++dt)
                    *reinterpret_cast<bf16x4*>(O + 16 * dt + 4 * g) =
                        bf16x4{(__bf16)(oacc[dt][0] * inv), (__bf16)(oacc[dt][1] * inv), (__bf16)(oacc[dt][2] * inv),
                               (__bf16)(oacc[dt][3] * inv)};
                if (g == 0)
                    p.lse[((long)b * p.H + h) * N + qi] = (m + __builtin_amdgcn_logf(ls)) * 0.6931471805599453f;
            }
        }
        lai_barrier();            // everybody is done with the images before the next element's rows land in them
    }
}

// --------------------------------------------------------------------------- backward, unmasked stage
// Same layout.  The [N][N] bf16 array is used three times per head, in place:
//   1. Bias, built as in the forward;
//   2. phase B (wave = query tile, lane = query): scores from K / Bias, P = exp2(s - lse) -- stored OVER the Bias row it
//      came from (one 8-byte store per tile), dP = dO V^T, dS = P (dP scale - delta scale), dQ += dS K (registers);
//   3. phase C (wave = key tile, lane = key; Q and dO images): P read back column-wise, dP again (2 MFMAs: no exp, no
//      score product), dS -- stored over P, and as a 16 x 16 tile to the workspace attn_bwd_relds_kernel reads -- ,
//      dV += P^T dO, dK += dS^T Q;
//   then one more pass of the table tiles for the position terms, read from the dS array along its diagonals
//      dQ_i += sum_t dS[i][i - t + N - 1] Pk[t]   (first 7 rounds)      dK_j += sum_t dS[j + t - N + 1][j] Pq[t]   (last 7):
//   every round holds exactly two tiles of each wave's strip = one 32-deep MFMA block (8 two-byte LDS reads for the dS
//   fragment, 4 K-major table fragments, 4 MFMAs).
// The table gradients stay with attn_bwd_relds_kernel (its 2 x 104 accumulator tiles per head do not fit this kernel's
// registers); dropout on the probabilities stays with attention_band.hip.
// Measured at the headline shape (128 x 16 heads x 197 tokens): 462 us + 186 us of attn_bwd_relds_kernel against 885 us for
// attn_bwd_band_kernel + attn_bwd_relds_kernel; by leaving one phase out (VITK_FULLB_STOP): Bias 58, phase B 69, phase C 113
// (its column-wise two-byte reads and writes of the array), position pass 62 us.
__device__ __forceinline__ void lai_colsum(const f32x4 (&acc)[4], float* dst, int g, int li);
#ifndef VITK_FULLB_STOP
#define VITK_FULLB_STOP 0            // timing experiments: bit 0 skips the Bias products, 1 phase B, 2 phase C, 3 the position pass
#endif
constexpr int FULLB_LDJ = 208;                  // 13 x 16 key columns, no pad (the 1.5 KiB are the lse / delta rows and column sums)
static size_t full_bwd_bytes(int N) {
    const LaiGeom G = lai_geom(N, 0);
    return (size_t)2 * G.NR * LAI_ASTR * 2 + (size_t)2 * FULL_TBUF * 2 + (size_t)N * FULLB_LDJ * 2 + (size_t)G.NR * 4 +
           3 * 64 * 4 + 64 * 2;
}

template <bool BIAS, bool DROP>
__global__ __launch_bounds__(LAI_THREADS) void attn_bwd_full_kernel(const BandParams p, int groups) {
    using T = __bf16;
    constexpr int ASTR = LAI_ASTR, LDJ = FULLB_LDJ, NTC = LAI_MAXNT, NP = (NTC + 1) / 2;
    constexpr int NRND = (2 * LAI_MAXNT + FULL_TPR - 1) / FULL_TPR;      // 7 rounds per table
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int N = p.N;
    const LaiGeom G = lai_geom(N, 0);
    const int NT = G.NT, NR = G.NR;
    const int h = blockIdx.x;

    T* img0 = reinterpret_cast<T*>(smem);                 // K, then Q
    T* img1 = img0 + NR * ASTR;                           // V, then dO
    T* tbuf = img1 + NR * ASTR;                           // two table buffers
    T* Pm = tbuf + 2 * FULL_TBUF;                         // [N][LDJ]: Bias, then P, then dS
    float* dls = reinterpret_cast<float*>(Pm + N * LDJ);  // [NR]: delta_i * scale
    float* colacc = dls + NR;                             // [3][64]
    T* pad = reinterpret_cast<T*>(colacc + 3 * 64);       // [64]: where masked two-byte writes go

    const T* PK = reinterpret_cast<const T*>(p.pk) + h * 64;
    const T* PQ = reinterpret_cast<const T*>(p.pq) + h * 64;
    constexpr float LOG2E = 1.4426950408889634f;
    const float sc2 = p.scale * LOG2E;
    const bool own = wave < NT;
    const int R = 2 * N - 1;
    const int qt = wave;                                  // query tile (phase B) and key tile (phase C) of this wave
    const int lo_q = qt;
    const int lo_k = max(0, N - 16 - 16 * qt) >> 4;
    const int hi_q = (16 * qt + 14 + N) >> 4, hi_k = max(0, 2 * N - 2 - 16 * qt) >> 4;

    // table staging: as in the forward (a cycle of seven pieces per thread, four in flight)
    const int grp = wave >> 3;
    const int st_row = (tid & 511) >> 3, st_d = (tid & 7) * 8;
    auto table_piece = [&](int k) -> u32x4 {
        const int rnd = 2 * k + grp;
        const bool second = rnd >= NRND;
        const int r0 = 16 * ((second ? rnd - NRND : rnd) + NRND * (st_row >> 4)) + (st_row & 15);
        // rows past the table: the last row again, NOT zeros -- a select on the loaded value makes the compiler wait for the
        // load where it is issued (a memory round trip per round: 0.65 us x 28); those rows only ever meet dS / Bias
        // entries that the range tests leave out
        return *reinterpret_cast<const u32x4*>((second ? PQ : PK) + (long)min(r0, R - 1) * p.ld_p + st_d);
    };
    T* const st_dst = tbuf + grp * FULL_TBUF + st_row * ASTR + st_d;
    u32x4 q0 = {}, q1 = {}, q2 = {}, q3 = {};
    if (BIAS) {
        q0 = table_piece(0);
        q1 = table_piece(1);
        q2 = table_piece(2);
        q3 = table_piece(3);
    }
    int knext = 4;
    auto stage_next = [&]() {
        *reinterpret_cast<u32x4*>(st_dst) = q0;
        q0 = q1;
        q1 = q2;
        q2 = q3;
        q3 = table_piece(knext);
        knext = knext == NRND - 1 ? 0 : knext + 1;
    };
    if (tid < 3 * 64) colacc[tid] = 0.f;

    for (int b = blockIdx.y; b < p.B; b += groups) {
        const T* Q = reinterpret_cast<const T*>(p.q) + (long)b * N * p.ld + h * 64;
        const T* K = reinterpret_cast<const T*>(p.k) + (long)b * N * p.ld + h * 64;
        const T* V = reinterpret_cast<const T*>(p.v) + (long)b * N * p.ld + h * 64;
        const T* dO = reinterpret_cast<const T*>(p.dout) + (long)b * N * p.ldo + h * 64;
        const T* O = reinterpret_cast<const T*>(p.out) + (long)b * N * p.ldo + h * 64;
        const float* lse = p.lse + ((long)b * p.H + h) * N;
        __bf16* ds_head = p.ds_ws + ((long)b * p.H + h) * (2 * NT - 1) * NT * 256;       // this head's dS tiles (cmax = NT - 1)
        const uint64_t dhead = (uint64_t)(b * p.H + h) * (uint64_t)N;      // first row of this head in the dropout stream
        int stid = tid;
        asm volatile("" : "+v"(stid));                    // per-thread offsets are made per head (as in attn_bwd_lai_kernel)
        const int lane = stid & 63, g = lane >> 4, li = lane & 15;
        const int row = 16 * wave + li;                   // the query (phase B) / key (phase C) of this lane
        const int fr = own ? row : -1;
        const int rown = min(row, N - 1);
        const int rowb = li * ASTR + 8 * g, colb = (4 * g + (li >> 2)) * ASTR + 4 * (li & 3);

        // ---- own rows; K, V images; lse and delta rows ---------------------------------------------------------
        Frag<T> qf[2], dof[2], kfo[2], vfo[2];
        float dlt = 0.f, l2 = 0.f;
        {
            Frag<T> of[2];
            lai_row_frags(kfo, K, p.ld, fr, N, lane);
            lai_row_frags(vfo, V, p.ld, fr, N, lane);
            lai_row_frags(qf, Q, p.ld, fr, N, lane);
            lai_row_frags(dof, dO, p.ldo, fr, N, lane);
            lai_row_frags(of, O, p.ldo, fr, N, lane);
            const float lraw = lse[rown];
            if (own) {
                lai_put_rows(img0, kfo, row, lane);
                lai_put_rows(img1, vfo, row, lane);
#pragma unroll
                for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                    for (int e = 0; e < 8; ++e) dlt += (float)dof[kb].v[e] * (float)of[kb].v[e];
                dlt += __shfl_xor(dlt, 16, 64);
                dlt += __shfl_xor(dlt, 32, 64);           // delta_i = dO_i . O_i
                l2 = row < N ? lraw * LOG2E : INFINITY;     // (a query past the sequence: every probability 0)
                if (g == 0) dls[row] = dlt * p.scale;
            }
        }
        const int jb = row < N ? row + N - 1 - 4 * g : -(1 << 20);           // j of (t = 4g, r = 0) for this query
        const int ib = row < N ? row - (N - 1) + 4 * g : -(1 << 20);         // i of (t = 4g, r = 0) for this key
        T* const c2p_b = Pm + rown * LDJ + (rown + N - 1 - 4 * g) - 3;
        T* const p2c_b = Pm + (rown - (N - 1) + 4 * g) * LDJ + rown;
        T* const padp = pad + lane;

        // ---- Bias (as in the forward) ------------------------------------------------------------------------------
        auto bias_round = [&](int rnd, const T* tb) {
            if (!own || (VITK_FULLB_STOP & 1)) return;
            const bool second = rnd >= NRND;
            const int rho = second ? rnd - NRND : rnd;
            if (!second) {
                T* cb = c2p_b - 16 * rho;
                const int j0 = jb - 16 * rho;
#pragma unroll
                for (int u = 0; u < FULL_TPR; ++u) {
                    const int tt = rho + NRND * u;
                    if (tt < lo_q || tt > hi_q) continue;
                    f32x4 x = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int kb = 0; kb < 2; ++kb) {
                        Frag<T> a;
                        frag_load(a, tb + u * (16 * ASTR) + rowb + 32 * kb);
                        x = mma(a, qf[kb], x);
                    }
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        T* e = (unsigned)(j0 - 16 * NRND * u - r) < (unsigned)N ? cb + (3 - r - 16 * NRND * u) : padp;
                        *e = (T)x[r];
                    }
                }
            } else {
                T* pb = p2c_b + 16 * rho * LDJ;
                const int i0 = ib + 16 * rho;
#pragma unroll
                for (int u = 0; u < FULL_TPR; ++u) {
                    const int tt = rho + NRND * u;
                    if (tt < lo_k || tt > hi_k) continue;
                    f32x4 x = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int kb = 0; kb < 2; ++kb) {
                        Frag<T> a;
                        frag_load(a, tb + u * (16 * ASTR) + rowb + 32 * kb);
                        x = mma(a, kfo[kb], x);
                    }
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        T* e = (unsigned)(i0 + 16 * NRND * u + r) < (unsigned)N ? pb + (16 * NRND * u + r) * LDJ : padp;
                        *e = (T)((float)*e + x[r]);
                    }
                }
            }
        };
        if (BIAS && grp == 0) stage_next();
        lai_barrier();
#pragma nounroll
        for (int it = 0; BIAS && it < NRND; ++it) {
            bias_round(2 * it, tbuf);
            if (grp == 1) stage_next();
            lai_barrier();
            bias_round(2 * it + 1, tbuf + FULL_TBUF);
            if (grp == 0 && it + 1 < NRND) stage_next();
            lai_barrier();
        }

        // ---- phase B: query-stationary: P over Bias, dQ (content part) --------------------------------------------
        f32x4 dq[4];
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) dq[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (own && !(VITK_FULLB_STOP & 2)) {
            const float dls_own = dlt * p.scale;
            T* brow = Pm + rown * LDJ + 4 * g;
#pragma unroll
            for (int pi = 0; pi < NP; ++pi) {
                f32x4 dsv[2];
                int ktc[2] = {0, 0};
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const int kt = 2 * pi + u;
                    dsv[u] = f32x4{0.f, 0.f, 0.f, 0.f};
                    if (kt >= NTC) {
                        ktc[u] = ktc[0];
                        continue;
                    }
                    ktc[u] = min(kt, NT - 1);
                    f32x4 s = {0.f, 0.f, 0.f, 0.f}, dp = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int kb = 0; kb < 2; ++kb) {
                        Frag<T> kr, vr;
                        lai_rfrag(kr, img0, rowb, ktc[u], kb);
                        lai_rfrag(vr, img1, rowb, ktc[u], kb);
                        s = mma(kr, qf[kb], s);           // raw[query li][key 16 kt + 4g + r]
                        dp = mma(vr, dof[kb], dp);        // dP[query li][key ..]
                    }
                    bf16x4 b4 = {};
                    if (BIAS) b4 = *reinterpret_cast<const bf16x4*>(brow + 16 * kt);
                    bf16x4 p4;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const bool ok = 16 * kt + 4 * g + r < N;
                        const float pe = __builtin_amdgcn_exp2f(ok ? fmaf(s[r] + (float)b4[r], sc2, -l2) : -INFINITY);
                        p4[r] = (T)pe;
                        // dropout: dP = (dO V^T) * keep / (1 - p); delta is unchanged (O was made from the dropped P)
                        const float dm = DROP ? drop_mul(p.drop, (dhead + (uint64_t)rown) * (uint64_t)N +
                                                                     (uint64_t)(uint32_t)min(16 * kt + 4 * g + r, N - 1)) : 1.f;
                        dsv[u][r] = pe * fmaf(dp[r] * dm, p.scale, -dls_own);
                    }
                    if (row < N) *reinterpret_cast<bf16x4*>(brow + 16 * kt) = p4;      // P over the Bias it was made from
                }
                Frag<T> bs;
                acc_to_frag(bs, dsv[0], dsv[1]);
#pragma unroll
                for (int dt = 0; dt < 4; ++dt) {          // dQ^T[d][q] += K^T[d][key] dS^T[key][q]
                    Frag<T> a;
                    lai_cfrag(a, img0, colb, ktc[0], ktc[1], dt);
                    dq[dt] = mma(a, bs, dq[dt]);
                }
            }
        }
        lai_barrier();
        // ---- Q, dO replace K, V (from the registers that hold them) ---------------------------------------------
        if (own) {
            lai_put_rows(img0, qf, row, lane);
            lai_put_rows(img1, dof, row, lane);
        }
        lai_barrier();

        // ---- phase C: key-stationary: dS over P (and to the workspace), dV, dK (content part) ----------------------
        f32x4 dk[4];
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) dk[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (own && !(VITK_FULLB_STOP & 4)) {
            const int kt = wave;
            f32x4 dv[4];
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) dv[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
            T* pcol = Pm + (4 * g) * LDJ + row;           // P[i = 4g (+ 16 qt + r)][j = row]; row <= 207 < LDJ
#pragma unroll
            for (int pi = 0; pi < NP; ++pi) {
                f32x4 pt[2], dsv[2];
                int qtc[2] = {0, 0};
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const int qq = 2 * pi + u;
                    pt[u] = f32x4{0.f, 0.f, 0.f, 0.f};
                    dsv[u] = f32x4{0.f, 0.f, 0.f, 0.f};
                    if (qq >= NTC) {
                        qtc[u] = qtc[0];
                        continue;
                    }
                    qtc[u] = min(qq, NT - 1);
                    if (qq >= NT) continue;               // (wave-uniform) no such query tile
                    f32x4 dp = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int kb = 0; kb < 2; ++kb) {
                        Frag<T> dor;
                        lai_rfrag(dor, img1, rowb, qq, kb);
                        dp = mma(dor, vfo[kb], dp);       // dP[query 16 qq + 4g + r][key li]
                    }
                    const f32x4 d4 = *reinterpret_cast<const f32x4*>(dls + 16 * qq + 4 * g);
                    const int i0 = 16 * qq + 4 * g;
                    bf16x4 ds4;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const bool ok = i0 + r < N;
                        T* e = ok ? pcol + (16 * qq + r) * LDJ : padp;
                        const float pe = ok ? (float)*e : 0.f;
                        const float dm = DROP ? drop_mul(p.drop, (dhead + (uint64_t)(uint32_t)min(i0 + r, N - 1)) * (uint64_t)N +
                                                                     (uint64_t)rown) : 1.f;
                        const float d = pe * fmaf(dp[r] * dm, p.scale, -d4[r]);
                        *e = (T)d;                        // dS over P
                        pt[u][r] = pe * dm;               // dV sees the dropped probabilities
                        dsv[u][r] = d;
                        ds4[r] = (T)d;
                    }
                    // the tile attn_bwd_relds_kernel reads: entry (c + cmax, kt), lane l holds dS[4 (l >> 4) + r][l & 15]
                    if (BIAS) *reinterpret_cast<bf16x4*>(ds_head + ((long)(qq - kt + NT - 1) * NT + kt) * 256 + lane * 4) = ds4;
                }
                Frag<T> bp, bs;
                acc_to_frag(bp, pt[0], pt[1]);
                acc_to_frag(bs, dsv[0], dsv[1]);
#pragma unroll
                for (int dt = 0; dt < 4; ++dt) {
                    Frag<T> a;
                    lai_cfrag(a, img1, colb, qtc[0], qtc[1], dt);
                    dv[dt] = mma(a, bp, dv[dt]);          // dV^T[d][key] += dO^T[d][q] P[q][key]
                    lai_cfrag(a, img0, colb, qtc[0], qtc[1], dt);
                    dk[dt] = mma(a, bs, dk[dt]);          // dK^T[d][key] += Q^T[d][q] dS[q][key]
                }
            }
            if (row < N) {
                T* dV = reinterpret_cast<T*>(p.dv) + ((long)b * N + row) * p.ld_d + h * 64;
#pragma unroll
                for (int dt = 0; dt < 4; ++dt)
                    *reinterpret_cast<bf16x4*>(dV + 16 * dt + 4 * g) =
                        bf16x4{(__bf16)dv[dt][0], (__bf16)dv[dt][1], (__bf16)dv[dt][2], (__bf16)dv[dt][3]};
            }
            if (p.dbias) lai_colsum(dv, colacc + 128, g, li);
        }

        // ---- position terms: one more cycle of the table tiles against the dS array ---------------------------------
        auto pos_round = [&](int rnd, const T* tb) {
            if (!own || (VITK_FULLB_STOP & 8)) return;
            const bool second = rnd >= NRND;
            const int rho = second ? rnd - NRND : rnd;
            const int lo = second ? lo_k : lo_q;
            // the two tiles of this round inside the strip: rho + 7 ua, rho + 7 (ua + 1)
            const int ua = lo > rho ? (lo - rho + NRND - 1) / NRND : 0;
            const int ta = rho + NRND * ua;
            Frag<T> bf;
            if (!second) {
                // dC2P[i][t] = dS[i][i - t + N - 1]: k = 4g + e of tile ta (e < 4), of tile ta + 7 (e >= 4)
                const T* rowp = Pm + rown * LDJ;
                const int ja = jb - 16 * ta;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const int j = ja - (e < 4 ? e : 16 * NRND + e - 4);
                    const T v = rowp[min(max(j, 0), LDJ - 1)];
                    bf.v[e] = (unsigned)j < (unsigned)N ? v : (T)0.f;
                }
#pragma unroll
                for (int dt = 0; dt < 4; ++dt) {
                    Frag<T> a;
                    lai_cfrag(a, tb, colb, ua, ua + 1, dt);
                    dq[dt] = mma(a, bf, dq[dt]);          // dQ^T[d][query] += Pk^T[d][t] dC2P^T[t][query]
                }
            } else {
                const T* colp = Pm + rown;
                const int ia = ib + 16 * ta;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const int i = ia + (e < 4 ? e : 16 * NRND + e - 4);
                    const T v = colp[min(max(i, 0), N - 1) * LDJ];
                    bf.v[e] = (unsigned)i < (unsigned)N ? v : (T)0.f;
                }
#pragma unroll
                for (int dt = 0; dt < 4; ++dt) {
                    Frag<T> a;
                    lai_cfrag(a, tb, colb, ua, ua + 1, dt);
                    dk[dt] = mma(a, bf, dk[dt]);          // dK^T[d][key] += Pq^T[d][t] dP2C^T[t][key]
                }
            }
        };
        if (BIAS && grp == 0) stage_next();
        lai_barrier();            // (also: every wave's dS is in the array; without tables: everybody is done with the images)
#pragma nounroll
        for (int it = 0; BIAS && it < NRND; ++it) {
            pos_round(2 * it, tbuf);
            if (grp == 1) stage_next();
            lai_barrier();
            pos_round(2 * it + 1, tbuf + FULL_TBUF);
            if (grp == 0 && it + 1 < NRND) stage_next();
            lai_barrier();
        }
        if (own) {
            if (row < N) {
                T* dQ = reinterpret_cast<T*>(p.dq) + ((long)b * N + row) * p.ld_d + h * 64;
                T* dK = reinterpret_cast<T*>(p.dk) + ((long)b * N + row) * p.ld_d + h * 64;
#pragma unroll
                for (int dt = 0; dt < 4; ++dt) {
                    *reinterpret_cast<bf16x4*>(dQ + 16 * dt + 4 * g) =
                        bf16x4{(__bf16)dq[dt][0], (__bf16)dq[dt][1], (__bf16)dq[dt][2], (__bf16)dq[dt][3]};
                    *reinterpret_cast<bf16x4*>(dK + 16 * dt + 4 * g) =
                        bf16x4{(__bf16)dk[dt][0], (__bf16)dk[dt][1], (__bf16)dk[dt][2], (__bf16)dk[dt][3]};
                }
            }
            if (p.dbias) {
                lai_colsum(dq, colacc, g, li);
                lai_colsum(dk, colacc + 64, g, li);
            }
        }
        // (the next head's first LDS writes -- K, V images, lse rows -- come after this loop's last barrier; its Bias
        // writes after two more)
    }
    lai_barrier();
    if (p.dbias && tid < 3 * 64) {
        const int which = tid >> 6, d = tid & 63;
        atomicAdd(p.dbias + (long)which * p.H * 64 + h * 64 + d, colacc[tid]);
    }
}

// ----------------------------------------------------------------------------------------- backward
// column sums of a [64 d][16 rows] accumulator set over its 16 row lanes, += dst[d] (LDS)
__device__ __forceinline__ void lai_colsum(const f32x4 (&acc)[4], float* dst, int g, int li) {
#pragma unroll
    for (int dt = 0; dt < 4; ++dt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float t = row16_sum(acc[dt][r]);
            if (li == 0) atomicAdd(dst + 16 * dt + 4 * g + r, t);
        }
}

// acc[dt] += sum_t Tab^T[d][t] * S[row][t]: the own-row product with a skewed dS copy (dcp_row = this lane's row of 48)
__device__ __forceinline__ void lai_table_product(f32x4 (&acc)[4], const __bf16* tab, const __bf16* dcp_row, int KBT,
                                                  int lane) {
    const int g = lane >> 4;
    for (int kb = 0; kb < KBT; ++kb) {
        const int t0 = 32 * kb + 8 * g;
        const bool inb = t0 < LAI_TLD;
        const int tc = inb ? t0 : 0;
        Frag<__bf16> bf;
        frag_load(bf, dcp_row + tc);
        if (!inb) frag_zero(bf);
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
            Frag<__bf16> a;
            col_frag(a, tab, LAI_ASTR, tc, tc + 4, 16 * dt, lane);
            acc[dt] = mma(a, bf, acc[dt]);
        }
    }
}

// acc += sum_rows S[row][16 u + .]^T * X[row][16 dt + .] over all NR rows: one 16 x 16 tile of a table gradient
__device__ __forceinline__ f32x4 lai_table_grad(f32x4 acc, const __bf16* dcp, const __bf16* img, int NR, int u, int dt,
                                                int lane) {
    const int g = lane >> 4;
    for (int kb = 0; 32 * kb < NR; ++kb) {
        const int row0 = 32 * kb + 8 * g;
        const bool valid = row0 < NR;
        const int rc = valid ? row0 : 0;
        Frag<__bf16> a, bq;
        col_frag(a, dcp, LAI_TLD, rc, rc + 4, 16 * u, lane);        // A[t = li][k] = S[row0 + e][16 u + li]
        col_frag(bq, img, LAI_ASTR, rc, rc + 4, 16 * dt, lane);     // B[k][d = li] = X[row0 + e][16 dt + li]
        if (!valid) frag_zero(a);
        acc = mma(a, bq, acc);           // D[t = 16 u + 4g + r][d = 16 dt + li]
    }
    return acc;
}

// grid (H, groups): the workgroup walks batch elements b = group, group + groups, ... of head h.
template <int NC, bool DROP, bool BIAS>
__global__ __launch_bounds__(LAI_THREADS) void attn_bwd_lai_kernel(const BandParams p, int groups) {
    using T = __bf16;
    constexpr int ASTR = LAI_ASTR, TLD = LAI_TLD, CM = (NC - 1) / 2, NP = (NC + 1) / 2;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, wave = tid >> 6;
    const int N = p.N, w = p.w;
    const LaiGeom G = lai_geom(N, w);
    const int NR = G.NR;
    constexpr int LDB = LAI_LDB;
    const int KBT = (G.TW + 31) / 32, NDT = (G.TW + 15) / 16;
    const int h = blockIdx.x;

    // P2C sits between the images and C2P, C2P in front of dC2P: the diagonal walks below read every element of a tile,
    // also those past either end of the sequence, and discard them -- those reads land in the workgroup's own LDS
    T* img0 = reinterpret_cast<T*>(smem);                 // K, then Q, then K
    T* img1 = img0 + NR * ASTR;                           // V, then dO
    T* tabk = img1 + NR * ASTR;                           // Pk_b, Pq_b: [48][ASTR], rows past the band zero
    T* tabq = tabk + TLD * ASTR;
    float* P2C = reinterpret_cast<float*>(tabq + TLD * ASTR);    // [N][LDB]
    float* C2P = P2C + N * LDB;
    T* dcp = reinterpret_cast<T*>(C2P + N * LDB);         // dC2P, then dP2C: [NR][48]  (2 N LDB floats: a multiple of 16 bytes)
    float* lse2s = BIAS ? reinterpret_cast<float*>(dcp + NR * TLD) : reinterpret_cast<float*>(tabk);   // (no tables: behind the images)
    float* dls = lse2s + NR;
    float* colacc = dls + NR;                             // [3][64]: column sums of dQ | dK | dV over this workgroup's heads

    const T* PK = reinterpret_cast<const T*>(p.pk) + h * 64;
    const T* PQ = reinterpret_cast<const T*>(p.pq) + h * 64;
    const int tab_r0 = N - 1 - w;                         // table row of t = 0
    constexpr float LOG2E = 1.4426950408889634f;
    const float sc2 = p.scale * LOG2E;
    const bool own = wave < G.NT;
    const bool tgw = BIAS && wave < 4 * NDT;              // owns one tile of each table gradient
    const int tg_u = wave >> 2, tg_dt = wave & 3;
    f32x4 apk = {0.f, 0.f, 0.f, 0.f}, apq = {0.f, 0.f, 0.f, 0.f};
    if (tid < 3 * 64) colacc[tid] = 0.f;
    if (BIAS) lai_stage_tables(tabk, tabq, PK, PQ, p.ld_p, tab_r0, G.TW, 2 * N - 1, tid);
    lai_barrier();

    for (int b = blockIdx.y; b < p.B; b += groups) {
        const T* Q = reinterpret_cast<const T*>(p.q) + (long)b * N * p.ld + h * 64;
        const T* K = reinterpret_cast<const T*>(p.k) + (long)b * N * p.ld + h * 64;
        const T* V = reinterpret_cast<const T*>(p.v) + (long)b * N * p.ld + h * 64;
        const T* dO = reinterpret_cast<const T*>(p.dout) + (long)b * N * p.ldo + h * 64;
        const T* O = reinterpret_cast<const T*>(p.out) + (long)b * N * p.ldo + h * 64;
        const float* lse = p.lse + ((long)b * p.H + h) * N;
        const uint64_t dhead = (uint64_t)(b * p.H + h) * (uint64_t)N;      // first row of this head in the dropout stream
        // per-thread offsets are invariants of this loop: hoisted out of it they alone overflow the 128 registers of a
        // 16-wave workgroup; recomputed per head they are a few VALU instructions
        int stid = tid;
        asm volatile("" : "+v"(stid));
        const int lane = stid & 63, g = lane >> 4, li = lane & 15;
        const int row = 16 * wave + li;                   // the query (phase B) / key (phase C) of this lane
        const int fr = own ? row : -1;
        const int rown = min(row, N - 1);
        if (VITK_LAI_STOP == 1) continue;

        // ---- phase A: own rows of Q, K, V, dO, O; K, V images; C2P / P2C rows, lse and delta ----------------------
        Frag<T> qf[2], dof[2];
        float dlt = 0.f, l2 = 0.f;
        {
            Frag<T> kfo[2], vfo[2], of[2];
            lai_row_frags(kfo, K, p.ld, fr, N, lane);
            lai_row_frags(vfo, V, p.ld, fr, N, lane);
            lai_row_frags(qf, Q, p.ld, fr, N, lane);
            lai_row_frags(dof, dO, p.ldo, fr, N, lane);
            lai_row_frags(of, O, p.ldo, fr, N, lane);
            const float lraw = lse[rown];
            if (own) {
                lai_put_rows(img0, kfo, row, lane);
                lai_put_rows(img1, vfo, row, lane);
                if (BIAS) lai_bias_rows(tabk, tabq, G, qf, kfo, C2P + rown * LDB, P2C + rown * LDB, row < N, lane);
#pragma unroll
                for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                    for (int e = 0; e < 8; ++e) dlt += (float)dof[kb].v[e] * (float)of[kb].v[e];
                dlt += __shfl_xor(dlt, 16, 64);
                dlt += __shfl_xor(dlt, 32, 64);           // delta_i = dO_i . O_i
                l2 = row < N ? lraw * LOG2E : 0.f;
                if (g == 0) {
                    lse2s[row] = l2;
                    dls[row] = dlt * p.scale;             // pre-scaled: dS = P (dP scale - delta scale)
                }
                if (BIAS)
                    for (int i = lane; i < 16 * TLD * 2 / 16; i += 64)   // own dC2P rows: zero (entries outside [0, N) stay so)
                        reinterpret_cast<u32x4*>(dcp + 16 * wave * TLD)[i] = u32x4{0, 0, 0, 0};
            }
        }
        lai_barrier();
        if (VITK_LAI_STOP == 2) continue;

        // ---- phase B: query-stationary (lane = query `row`): dQ, dC2P --------------------------------------------
        if (own) {
            const int qt = wave;
            const int tq0 = row < N ? li - 4 * g - 16 * CM + w : 1 << 20;                // t of element (ci = 0, r = 0)
            const float* pcB = C2P + rown * LDB + (li - 4 * g - 16 * CM + w) - 3;
            const float* ppB = P2C + (16 * (qt - CM) + 4 * g) * LDB + (li - 4 * g - 16 * CM + w) + 16 * (NC - 1);
            T* dcprow = dcp + row * TLD;
            const float dls_own = dlt * p.scale;
            f32x4 dq[4];
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) dq[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int pi = 0; pi < NP; ++pi) {
                f32x4 dsv[2];
                int ktc[2] = {0, 0};
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const int ci = 2 * pi + u;
                    dsv[u] = f32x4{0.f, 0.f, 0.f, 0.f};
                    if (ci >= NC) {
                        ktc[u] = ktc[0];
                        continue;
                    }
                    const int kt = qt + CM - ci;
                    ktc[u] = min(max(kt, 0), G.NT - 1);
                    f32x4 s = {0.f, 0.f, 0.f, 0.f}, dp = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int kb = 0; kb < 2; ++kb) {
                        Frag<T> kf, vf;
                        row_frag<T, 2, 64>(kf, img0, 16 * ktc[u], kb, lane);
                        row_frag<T, 2, 64>(vf, img1, 16 * ktc[u], kb, lane);
                        s = mma(kf, qf[kb], s);           // raw[query li][key 4g + r]
                        dp = mma(vf, dof[kb], dp);        // dP[query li][key 4g + r]
                    }
                    // element r: key j = j0 + r, band row t = t0 - r; (unsigned) compares cover t < 0, j < 0 (tile in front
                    // of the sequence) and, through the poisoned t of a query past the sequence, i >= N
                    const int j0 = 16 * kt + 4 * g, t0 = tq0 + 16 * ci;
                    const float* pc = pcB + 16 * ci;                                     // C2P[i][t0 - r]  = pc[3 - r]
                    const float* pp = ppB + (NC - 1 - ci) * (16 * LDB - 16);             // P2C[j0 + r][t0 - r] = pp[r (LDB - 1)]
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const bool ok = (unsigned)(t0 - r) < (unsigned)G.TW && (unsigned)(j0 + r) < (unsigned)N;
                        const float raw = BIAS ? s[r] + pc[3 - r] + pp[r * (LDB - 1)] : s[r];
                        const float pe = __builtin_amdgcn_exp2f(ok ? fmaf(raw, sc2, -l2) : -INFINITY);
                        // dropout: dP = (dO V^T) * keep / (1 - p); delta is unchanged (O was made from the dropped P)
                        const float dm = DROP ? drop_mul(p.drop, (dhead + (uint64_t)rown) * (uint64_t)N + (uint64_t)(uint32_t)(j0 + r))
                                              : 1.f;
                        const float d = pe * fmaf(dp[r] * dm, p.scale, -dls_own);
                        dsv[u][r] = d;
                        // outside the band d = 0: written to a pad column (47) of the own row instead of branching
                        if (BIAS) dcprow[ok ? t0 - r : TLD - 1] = (__bf16)d;
                    }
                }
                Frag<T> bs;
                acc_to_frag(bs, dsv[0], dsv[1]);
#pragma unroll
                for (int dt = 0; dt < 4; ++dt) {          // dQ^T[d][q] += K^T[d][key] dS^T[key][q]
                    Frag<T> a;
                    col_frag(a, img0, ASTR, 16 * ktc[0] + 4 * g, 16 * ktc[1] + 4 * g, 16 * dt, lane);
                    dq[dt] = mma(a, bs, dq[dt]);
                }
            }
            if (BIAS) {
                wave_lds_sync();
                lai_table_product(dq, tabk, dcprow, KBT, lane);     // dQ_i += sum_t dC2P[i][t] Pk_b[t]
            }
            if (row < N) {
                T* dQ = reinterpret_cast<T*>(p.dq) + ((long)b * N + row) * p.ld_d + h * 64;
#pragma unroll
                for (int dt = 0; dt < 4; ++dt)
                    *reinterpret_cast<bf16x4*>(dQ + 16 * dt + 4 * g) =
                        bf16x4{(__bf16)dq[dt][0], (__bf16)dq[dt][1], (__bf16)dq[dt][2], (__bf16)dq[dt][3]};
            }
            if (p.dbias) lai_colsum(dq, colacc, g, li);
        }
        lai_barrier();
        if (VITK_LAI_STOP == 3) continue;

        // ---- Q, dO replace K, V (from the registers that hold them); dPk_b += dC2P^T Q ---------------------------
        Frag<T> kfo[2], vfo[2];
        lai_row_frags(kfo, K, p.ld, fr, N, lane);         // for phase C: back from L2 while the table gradient runs
        lai_row_frags(vfo, V, p.ld, fr, N, lane);
        if (own) {
            lai_put_rows(img0, qf, row, lane);
            lai_put_rows(img1, dof, row, lane);
        }
        lai_barrier();
        if (BIAS) {
            if (tgw) apk = lai_table_grad(apk, dcp, img0, NR, tg_u, tg_dt, lane);
            lai_barrier();
        }
        if (VITK_LAI_STOP == 4) continue;

        // ---- phase C: key-stationary (lane = key `row`): P and dS again, dK, dV, dP2C --------------------------
        if (own) {
            const int kt = wave;
            const int tk0 = row < N ? 4 * g - li - 16 * CM + w : 1 << 20;                // t of element (ci = 0, r = 0)
            const float* ppC = P2C + rown * LDB + (4 * g - li - 16 * CM + w);
            const float* pcC = C2P + (16 * (kt - CM) + 4 * g) * LDB + (4 * g - li - 16 * CM + w);
            T* dcprow = dcp + row * TLD;
            if (BIAS)
                for (int i = lane; i < 16 * TLD * 2 / 16; i += 64)
                    reinterpret_cast<u32x4*>(dcp + 16 * wave * TLD)[i] = u32x4{0, 0, 0, 0};
            f32x4 dk[4], dv[4];
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
                dk[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
                dv[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
            }
#pragma unroll
            for (int pi = 0; pi < NP; ++pi) {
                f32x4 pt[2], dsv[2];
                int qtc[2] = {0, 0};
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const int ci = 2 * pi + u;
                    pt[u] = f32x4{0.f, 0.f, 0.f, 0.f};
                    dsv[u] = f32x4{0.f, 0.f, 0.f, 0.f};
                    if (ci >= NC) {
                        qtc[u] = qtc[0];
                        continue;
                    }
                    const int qt = kt + ci - CM;
                    qtc[u] = min(max(qt, 0), G.NT - 1);
                    f32x4 s = {0.f, 0.f, 0.f, 0.f}, dp = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int kb = 0; kb < 2; ++kb) {
                        Frag<T> qr, dor;
                        row_frag<T, 2, 64>(qr, img0, 16 * qtc[u], kb, lane);
                        row_frag<T, 2, 64>(dor, img1, 16 * qtc[u], kb, lane);
                        s = mma(qr, kfo[kb], s);          // raw[query 4g + r][key li]
                        dp = mma(dor, vfo[kb], dp);
                    }
                    const f32x4 l4 = *reinterpret_cast<const f32x4*>(lse2s + 16 * qtc[u] + 4 * g);   // queries 4g .. 4g+3
                    const f32x4 d4 = *reinterpret_cast<const f32x4*>(dls + 16 * qtc[u] + 4 * g);
                    const int i0 = 16 * qt + 4 * g, t0 = tk0 + 16 * ci;                  // query i0 + r, band row t0 + r
                    const float* pc = pcC + ci * (16 * LDB + 16);                        // C2P[i0 + r][t0 + r] = pc[r (LDB + 1)]
                    const float* pp = ppC + 16 * ci;                                     // P2C[j][t0 + r]      = pp[r]
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const bool ok = (unsigned)(t0 + r) < (unsigned)G.TW && (unsigned)(i0 + r) < (unsigned)N;
                        const float raw = BIAS ? s[r] + pc[r * (LDB + 1)] + pp[r] : s[r];
                        const float pe = __builtin_amdgcn_exp2f(ok ? fmaf(raw, sc2, -l4[r]) : -INFINITY);
                        const float dm = DROP ? drop_mul(p.drop, (dhead + (uint64_t)(uint32_t)(i0 + r)) * (uint64_t)N + (uint64_t)rown)
                                              : 1.f;
                        const float d = pe * fmaf(dp[r] * dm, p.scale, -d4[r]);
                        pt[u][r] = pe * dm;           // dV sees the dropped probabilities
                        dsv[u][r] = d;
                        if (BIAS) dcprow[ok ? t0 + r : TLD - 1] = (__bf16)d;
                    }
                }
                Frag<T> bp, bs;
                acc_to_frag(bp, pt[0], pt[1]);
                acc_to_frag(bs, dsv[0], dsv[1]);
#pragma unroll
                for (int dt = 0; dt < 4; ++dt) {
                    Frag<T> a;
                    col_frag(a, img1, ASTR, 16 * qtc[0] + 4 * g, 16 * qtc[1] + 4 * g, 16 * dt, lane);
                    dv[dt] = mma(a, bp, dv[dt]);          // dV^T[d][key] += dO^T[d][q] P[q][key]
                    col_frag(a, img0, ASTR, 16 * qtc[0] + 4 * g, 16 * qtc[1] + 4 * g, 16 * dt, lane);
                    dk[dt] = mma(a, bs, dk[dt]);          // dK^T[d][key] += Q^T[d][q] dS[q][key]
                }
            }
            if (BIAS) {
                wave_lds_sync();
                lai_table_product(dk, tabq, dcprow, KBT, lane);     // dK_j += sum_t dP2C[j][t] Pq_b[t]
            }
            if (row < N) {
                T* dK = reinterpret_cast<T*>(p.dk) + ((long)b * N + row) * p.ld_d + h * 64;
                T* dV = reinterpret_cast<T*>(p.dv) + ((long)b * N + row) * p.ld_d + h * 64;
#pragma unroll
                for (int dt = 0; dt < 4; ++dt) {
                    *reinterpret_cast<bf16x4*>(dK + 16 * dt + 4 * g) =
                        bf16x4{(__bf16)dk[dt][0], (__bf16)dk[dt][1], (__bf16)dk[dt][2], (__bf16)dk[dt][3]};
                    *reinterpret_cast<bf16x4*>(dV + 16 * dt + 4 * g) =
                        bf16x4{(__bf16)dv[dt][0], (__bf16)dv[dt][1], (__bf16)dv[dt][2], (__bf16)dv[dt][3]};
                }
            }
            if (p.dbias) {
                lai_colsum(dk, colacc + 64, g, li);
                lai_colsum(dv, colacc + 128, g, li);
            }
        }
        lai_barrier();
        if (VITK_LAI_STOP == 5) continue;

        // ---- K again (from the registers); dPq_b += dP2C^T K ------------------------------------------------------
        if (BIAS) {
            if (own) lai_put_rows(img0, kfo, row, lane);
            lai_barrier();
            if (tgw) apq = lai_table_grad(apq, dcp, img0, NR, tg_u, tg_dt, lane);
            lai_barrier();
        }
    }

    // ---- flush: table gradients (one 16 x 16 tile of each per wave), bias-gradient column sums -------------------
    const int lane = tid & 63, g = lane >> 4, li = lane & 15;
    if (tgw) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int t = 16 * tg_u + 4 * g + r;
            if (t < G.TW) {
                const long off = (long)(tab_r0 + t) * p.ld_dp + h * 64 + 16 * tg_dt + li;
                atomicAdd(p.dpk + off, apk[r]);
                atomicAdd(p.dpq + off, apq[r]);
            }
        }
    }
    if (p.dbias && tid < 3 * 64) {
        const int which = tid >> 6, d = tid & 63;
        atomicAdd(p.dbias + (long)which * p.H * 64 + h * 64 + d, colacc[tid]);
    }
}

// ------------------------------------------------------------------------------------------- host
bool lai_eligible(const BandParams& p, bool backward) {
    static const bool enabled = [] {
        const char* e = getenv("VITK_ATTN_LAI");
        return !(e && e[0] == '0');
    }();
    if (!(enabled && p.dh == 64 && p.N <= 16 * LAI_MAXNT && p.w >= 0 && p.w <= LAI_MAXW && p.w <= p.N - 1)) return false;
    const bool bias = p.pk != nullptr;
    return (backward ? lai_bwd_bytes(p.N, p.w, bias) : lai_fwd_bytes(p.N, p.w, bias)) <= 160 * 1024;     // e.g. 200+ tokens at w = 20: no
}
static_assert(2 * 208 * LAI_ASTR * 2 + 208 * LAI_TLD * 2 + 2 * LAI_TLD * LAI_ASTR * 2 + 2 * 208 * 4 + 768 + 2 * 197 * LAI_LDB * 4 <=
                  160 * 1024, "backward at 197 tokens, w = 20");

// unmasked stage, forward: heads of 64 with tables, one chunk, and the Bias array must fit beside the images
bool full_eligible(const BandParams& p, bool backward) {
    static const bool enabled = [] {
        const char* e = getenv("VITK_ATTN_FULL");
        return !(e && e[0] == '0');
    }();
    // (short heads: the 13 score tiles a lane carries would be mostly padding: they stay on attention_band.hip)
    if (!enabled || p.dh != 64 || p.N > 16 * LAI_MAXNT || p.N <= 128) return false;
    if (p.w >= 0 && p.w < p.N - 1) return false;
    if (backward) {
        static const bool bwd_enabled = [] {
            const char* e = getenv("VITK_ATTN_FULL_BWD");
            return !(e && e[0] == '0');
        }();
        // a key tile's table strip (rows N - 16 - 16 kt .. 2N - 2 - 16 kt) must span at most 14 tiles: the position
        // pass pairs exactly two tiles of a strip per round (201+ tokens span 15: attention_band.hip)
        return bwd_enabled && p.ds_ws != nullptr && (p.N & 15) + p.N <= 209 && full_bwd_bytes(p.N) <= 160 * 1024;
    }
    return full_fwd_bytes(p.N) <= 160 * 1024;          // up to 201 tokens
}
void full_launch(const BandParams& p, bool backward, hipStream_t st) {
    if (backward) {
        int groups = 256 / p.H;
        if (groups < 1) groups = 1;
        if (groups > p.B) groups = p.B;
        if (groups > 65535) groups = 65535;
        const size_t lds = full_bwd_bytes(p.N);
#define VITK_FULLB(BI, DR)                                                                                         \
    do {                                                                                                           \
        (void)vitk_lds_optin((const void*)attn_bwd_full_kernel<BI, DR>, lds);                                      \
        hipLaunchKernelGGL((attn_bwd_full_kernel<BI, DR>), dim3(p.H, groups), dim3(LAI_THREADS), lds, st, p, groups); \
    } while (0)
        const bool drop = p.drop.thresh != 0;
        if (p.pk != nullptr) {
            if (drop) VITK_FULLB(true, true); else VITK_FULLB(true, false);
        } else {
            if (drop) VITK_FULLB(false, true); else VITK_FULLB(false, false);
        }
#undef VITK_FULLB
        return;
    }
    int groups = 256 / p.H;
    if (groups < 1) groups = 1;
    if (groups > p.B) groups = p.B;
    if (groups > 65535) groups = 65535;
    const size_t lds = full_fwd_bytes(p.N);
#define VITK_FULLF(DR, BI)                                                                                         \
    do {                                                                                                           \
        (void)vitk_lds_optin((const void*)attn_fwd_full_kernel<DR, BI>, lds);                                      \
        hipLaunchKernelGGL((attn_fwd_full_kernel<DR, BI>), dim3(p.H, groups), dim3(LAI_THREADS), lds, st, p, groups); \
    } while (0)
    const bool drop = p.drop.thresh != 0;
    if (p.pk != nullptr) {
        if (drop) VITK_FULLF(true, true); else VITK_FULLF(false, true);
    } else {
        if (drop) VITK_FULLF(true, false); else VITK_FULLF(false, false);
    }
#undef VITK_FULLF
}

void lai_launch(const BandParams& p, bool backward, hipStream_t st) {
    const bool wide = p.w > 16;
    // One workgroup per CU for the whole launch: it stages the tables once and (backward) keeps its table gradients in
    // registers across its batch elements.  Measured against shorter workgroups that the hardware could deal out as CUs come
    // free beside the side stream's persistent dW GEMM (VITK_LAI_HEADS_PER_WG = 2 / 1 at B = 128, H = 16: 1024 / 2048
    // workgroups): alone 57 / 72 / 85 us forward and 287 / 330 / 391 us backward, in the training step 2002 / 1978 / 1949
    // images/s -- the tables and the flush cost more than the balance gains.
    static const int per_wg = [] {
        const char* e = getenv("VITK_LAI_HEADS_PER_WG");
        return e ? atoi(e) : 0;
    }();
    int groups = per_wg > 0 ? (p.B + per_wg - 1) / per_wg : 256 / p.H;
    if (groups < 1) groups = 1;
    if (groups > p.B) groups = p.B;
    if (groups > 65535) groups = 65535;
    const bool drop = p.drop.thresh != 0;
    const bool bias = p.pk != nullptr;
#define VITK_LAI2(KERN, NCV, DR, BI)                                                                                \
    do {                                                                                                            \
        (void)vitk_lds_optin((const void*)KERN<NCV, DR, BI>, lds);                                                  \
        hipLaunchKernelGGL((KERN<NCV, DR, BI>), dim3(p.H, groups), dim3(LAI_THREADS), lds, st, p, groups);          \
    } while (0)
#define VITK_LAI(KERN, NCV, DR)                                                                                     \
    do {                                                                                                            \
        if (bias) VITK_LAI2(KERN, NCV, DR, true); else VITK_LAI2(KERN, NCV, DR, false);                             \
    } while (0)
    if (!backward) {
        const size_t lds = lai_fwd_bytes(p.N, p.w, bias);
        if (wide) {
            if (drop) VITK_LAI(attn_fwd_lai_kernel, 5, true); else VITK_LAI(attn_fwd_lai_kernel, 5, false);
        } else {
            if (drop) VITK_LAI(attn_fwd_lai_kernel, 3, true); else VITK_LAI(attn_fwd_lai_kernel, 3, false);
        }
        return;
    }
    const size_t lds = lai_bwd_bytes(p.N, p.w, bias);
    if (wide) {
        if (drop) VITK_LAI(attn_bwd_lai_kernel, 5, true); else VITK_LAI(attn_bwd_lai_kernel, 5, false);
    } else {
        if (drop) VITK_LAI(attn_bwd_lai_kernel, 3, true); else VITK_LAI(attn_bwd_lai_kernel, 3, false);
    }
#undef VITK_LAI
#undef VITK_LAI2
}

}  // namespace vitk
