// Fused attention backward: dQ, dK, dV and the gradients of the projected relative tables
// (dPk, dPq), recomputing the probabilities from Q, K, the tables and the forward's LSE.
//
// Replaces the autograd backward of AttentionLAI.forward
// (timm_classes/timm_locality_aware_transformer.py:51-76) and of
// Disentangled[LAI]SelfAttention.forward + disentangled_attention_bias
// (disentangled_transformer/disentangled_transformer.py:324-347, 357-426, 875-919, 929-998):
// bmm / gather / masked_fill / softmax backward chains over [B*H, N, N] and [B*H, N, 2S] tensors.
// Here nothing of size N x N (or N x 2S) reaches HBM.
//
// With s = scale * raw, raw[i][j] = Q_i.K_j + Q_i.Pk[i-j] + K_j.Pq[i-j], P = softmax(s | band):
//   delta_i = dO_i . O_i                       dP_ij = dO_i . V_j
//   dS_ij   = scale * P_ij * (dP_ij - delta_i)                      (= d loss / d raw)
//   dQ_i = sum_j dS_ij (K_j + Pk[i-j])         dK_j = sum_i dS_ij (Q_i + Pq[i-j])
//   dV_j = sum_i P_ij dO_i
//   dPk[r] = sum_{b, i-j=r} dS_ij Q_i          dPq[r] = sum_{b, i-j=r} dS_ij K_j
//
// Three kernels:
//   attn_delta_kernel : delta[b][h][i]  (HBM-bound row dot)
//   attn_bwd_kernel   : key-stationary.  A workgroup owns 64*KTW keys of one (b, h) (a wave owns
//                       16*KTW of them: K/V fragments and the dK^T/dV^T accumulators stay in
//                       registers) and sweeps the queries in blocks of 32.  Scores are computed
//                       with the KEY on the MFMA lane, so P and dS tiles are directly the B
//                       operands of dV^T += dO^T P and dK^T += Q^T dS.  dS crosses LDS once
//                       (per-wave scratch) for dQ and for the Toeplitz (relative-position) terms,
//                       which are MFMA products against skewed reads of that scratch:
//                         dK^T[d][jj] += sum_t Pq^T[d][t] Sy[t][jj],  Sy[t][jj] = dS[jj+t-15][jj]
//                         dQ^T[d][ii] += sum_t Pk^T[d][t] Sx[t][ii],  Sx[t][ii] = dS[ii][ii+15-t]
//                       dQ is reduced over the 4 waves in LDS, then added to an fp32 accumulator
//                       in HBM (one 256-byte row segment per wave instruction).
//   attn_bwd_rel_kernel: dPk / dPq.  A wave owns one 16x16-tile DIAGONAL of one head and walks
//                       it for a whole group of batches: along a diagonal the 31 relative rows a
//                       tile touches never change, so the table fragments and the dPk/dPq
//                       accumulators stay in registers and are flushed once (fp32 atomics, tiny).
#include "attn_common.h"

namespace vitk {

constexpr int DSLD = 20;  // floats per row of a 16x16 dS scratch tile (16-byte aligned rows)

struct AttnBwdParams {
    const void *q, *k, *v;
    long ld;
    const void *pk, *pq;
    long ld_p;
    const void* out;
    const void* dout;
    long ldo;
    const float* lse;   // [B][H][N]
    float* delta;       // [B][H][N] workspace
    float* dq_acc;      // fp32 accumulator, element (b, n, h, d) at (b*N + n) * ld_acc + h*dh + d
    long ld_acc;
    void *dk, *dv;      // T, element at (b*N + n) * ld_d + h*dh + d
    long ld_d;
    float *dpk, *dpq;   // fp32 [2N-1][ld_dp] (+ h*dh + d), accumulated into
    long ld_dp;
    int B, H, N, dh;
    float scale;
    int w;
    int bgroup;         // rel kernel: batches per wave
    const __bf16* ds_ws; // whole-head path: dS tiles written by attn_bwd_band_kernel (else null)
    DropParams drop;     // dropout on the probabilities (attention.hip)
};

__device__ __forceinline__ void frag_set(Frag<__bf16>& f, int e, float v) { f.v[e] = (__bf16)v; }
__device__ __forceinline__ void frag_set(Frag<float>& f, int e, float v) { f.v[e] = v; }

// ------------------------------------------------------------------ delta = rowsum(dO * O)
template <typename T>
__global__ __launch_bounds__(256) void attn_delta_kernel(const T* __restrict__ out, const T* __restrict__ dout,
                                                         long ldo, float* __restrict__ delta, int B, int H, int N,
                                                         int dh) {
    constexpr int EPC = 16 / sizeof(T);
    const long total = (long)B * N * H;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const int h = idx % H;
        const long bn = idx / H;
        const int n = bn % N;
        const int b = bn / N;
        const T* o = out + bn * ldo + h * dh;
        const T* d = dout + bn * ldo + h * dh;
        float s = 0.f;
        for (int c = 0; c < dh; c += EPC) {
            const u32x4 a = *reinterpret_cast<const u32x4*>(o + c);
            const u32x4 g = *reinterpret_cast<const u32x4*>(d + c);
            const T* ae = reinterpret_cast<const T*>(&a);
            const T* ge = reinterpret_cast<const T*>(&g);
#pragma unroll
            for (int e = 0; e < EPC; ++e) s += (float)ae[e] * (float)ge[e];
        }
        delta[((long)b * H + h) * N + n] = s;
    }
}

// Skewed fragments of a 16x16 dS tile held in LDS as ds[ii][jj] (row stride DSLD floats).
//   Sx : element e <- dS[li][li + 15 - (8g+e)]     (B operand, k = t = 8g+e, column ii = li)
//   Sy : element e <- dS[li + (8g+e) - 15][li]     (B operand, k = t = 8g+e, column jj = li)
template <typename T>
__device__ __forceinline__ void skew_x(Frag<T>& f, const float* ds, int g, int li) {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int jj = li + 15 - 8 * g - e;
        const float v = ds[li * DSLD + min(max(jj, 0), 15)];
        frag_set(f, e, (jj >= 0 && jj < 16) ? v : 0.f);
    }
}
template <typename T>
__device__ __forceinline__ void skew_y(Frag<T>& f, const float* ds, int g, int li) {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int ii = li + 8 * g + e - 15;
        const float v = ds[min(max(ii, 0), 15) * DSLD + li];
        frag_set(f, e, (ii >= 0 && ii < 16) ? v : 0.f);
    }
}

template <typename T, int DKB, int KTW, int NW, bool BIAS> struct BwdCfg {
    using C = ACfg<T, DKB>;
    static constexpr int NT = 64 * NW;                // threads per workgroup
    static constexpr int KB = 16 * KTW * NW;          // keys per workgroup
    static constexpr int WIN = KB + 32;               // staged relative rows (KB + 31 used)
    static constexpr int DQLD = 32 * DKB + 4;         // floats per row of the dQ reduction tile
    static constexpr int WSCR = (BIAS ? 2 * SCR_FLOATS : 0) + KTW * 16 * DSLD;  // floats per wave
    static constexpr size_t lds_bytes() {
        return (size_t)(64 + KB + (BIAS ? 2 * WIN : 0)) * C::ASTR * sizeof(T) +
               (size_t)(64 + 32 * DQLD + NW * WSCR) * sizeof(float);
    }
};

template <typename T, int DKB, int KTW, int NW, bool BIAS>
__global__ __launch_bounds__(64 * NW) void attn_bwd_kernel(const AttnBwdParams p) {
    using C = ACfg<T, DKB>;
    using BC = BwdCfg<T, DKB, KTW, NW, BIAS>;
    constexpr int NTH = BC::NT;
    constexpr int ASTR = C::ASTR;
    constexpr int KB = BC::KB, WIN = BC::WIN, DQLD = BC::DQLD;
    constexpr int ND = 2 * DKB;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    T* Qs = reinterpret_cast<T*>(smem);
    T* dOs = Qs + 32 * ASTR;
    T* Ks = dOs + 32 * ASTR;
    T* Pks = Ks + KB * ASTR;
    T* Pqs = Pks + (BIAS ? WIN * ASTR : 0);
    float* lse2s = reinterpret_cast<float*>(Pqs + (BIAS ? WIN * ASTR : 0));
    float* dls = lse2s + 32;
    float* dQs = dls + 32;
    float* scr = dQs + 32 * DQLD;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g = lane >> 4, li = lane & 15;
    const int N = p.N, dh = p.dh, H = p.H;
    const int kb0 = blockIdx.x * KB, h = blockIdx.y, b = blockIdx.z;
    const int kw0 = 16 * KTW * wave;   // wave's first key inside the block
    const int jw = kb0 + kw0;

    float* xs = scr + wave * BC::WSCR;
    float* ys = xs + (BIAS ? SCR_FLOATS : 0);
    float* dss = ys + (BIAS ? SCR_FLOATS : 0);

    const T* Q = reinterpret_cast<const T*>(p.q) + (long)b * N * p.ld + h * dh;
    const T* K = reinterpret_cast<const T*>(p.k) + (long)b * N * p.ld + h * dh;
    const T* V = reinterpret_cast<const T*>(p.v) + (long)b * N * p.ld + h * dh;
    const T* dO = reinterpret_cast<const T*>(p.dout) + (long)b * N * p.ldo + h * dh;
    const T* PK = BIAS ? reinterpret_cast<const T*>(p.pk) + h * dh : nullptr;
    const T* PQ = BIAS ? reinterpret_cast<const T*>(p.pq) + h * dh : nullptr;
    const float* lse = p.lse + ((long)b * H + h) * N;
    const float* delta = p.delta + ((long)b * H + h) * N;

    constexpr float LOG2E = 1.4426950408889634f;
    const float sc2 = p.scale * LOG2E;

    int ilo = 0, ihi = N;
    if (p.w >= 0) {
        ilo = max(0, kb0 - p.w);
        ihi = min(N, kb0 + KB + p.w);
        ilo = (ilo / 32) * 32;
    }

    // LAI band narrow enough: every live tile's relative rows fit ONE window, staged once
    // (rows delta in [-16 cmax - 15, 16 cmax + 15], cmax = last live tile diagonal).
    const int cmax = p.w >= 0 ? (p.w + 15) / 16 : 0;
    const bool win_fixed = BIAS && p.w >= 0 && 32 * cmax + 32 <= WIN;
    stage_rows<T, DKB>(Ks, K, p.ld, kb0, KB, 0, N, dh, tid, NTH);
    if (win_fixed) {
        const int r_lo = -16 * cmax - 15 + (N - 1);
        stage_rows<T, DKB>(Pks, PK, p.ld_p, r_lo, 32 * cmax + 32, 0, 2 * N - 1, dh, tid, NTH);
        stage_rows<T, DKB>(Pqs, PQ, p.ld_p, r_lo, 32 * cmax + 32, 0, 2 * N - 1, dh, tid, NTH);
    }
    __syncthreads();

    Frag<T> kf[KTW][DKB], vf[KTW][DKB];
#pragma unroll
    for (int kt = 0; kt < KTW; ++kt)
#pragma unroll
        for (int kb = 0; kb < DKB; ++kb) {
            row_frag<T, DKB>(kf[kt][kb], Ks, kw0 + 16 * kt, kb, lane);
            row_frag_global<T>(vf[kt][kb], V, p.ld, jw + 16 * kt + li, N, dh, kb, lane);
        }
    f32x4 dkacc[KTW][ND], dvacc[KTW][ND];
#pragma unroll
    for (int kt = 0; kt < KTW; ++kt)
#pragma unroll
        for (int dt = 0; dt < ND; ++dt) {
            dkacc[kt][dt] = f32x4{0.f, 0.f, 0.f, 0.f};
            dvacc[kt][dt] = f32x4{0.f, 0.f, 0.f, 0.f};
        }

    for (int i0 = ilo; i0 < ihi; i0 += 32) {
        __syncthreads();
        stage_rows<T, DKB>(Qs, Q, p.ld, i0, 32, 0, N, dh, tid, NTH);
        stage_rows<T, DKB>(dOs, dO, p.ldo, i0, 32, 0, N, dh, tid, NTH);
        if (BIAS && !win_fixed) {
            const int r_lo = i0 - kb0 - KB + 1 + (N - 1);  // table row of window row 0
            stage_rows<T, DKB>(Pks, PK, p.ld_p, r_lo, WIN, 0, 2 * N - 1, dh, tid, NTH);
            stage_rows<T, DKB>(Pqs, PQ, p.ld_p, r_lo, WIN, 0, 2 * N - 1, dh, tid, NTH);
        }
        if (tid < 32) {
            const int qi = i0 + tid;
            lse2s[tid] = qi < N ? lse[qi] * LOG2E : 0.f;
            dls[tid] = qi < N ? delta[qi] : 0.f;
        }
        for (int i = tid; i < 32 * DQLD; i += NTH) dQs[i] = 0.f;
        __syncthreads();

        const bool wave_live =
            jw < N && (p.w < 0 || !(jw > i0 + 31 + p.w || jw + 16 * KTW - 1 < i0 - p.w));
        if (wave_live) {
            f32x4 pt[2][KTW], dst[2][KTW];
#pragma unroll
            for (int qt = 0; qt < 2; ++qt) {
                const int qi0 = i0 + 16 * qt;
                Frag<T> qf[DKB], dof[DKB];
#pragma unroll
                for (int kb = 0; kb < DKB; ++kb) {
                    row_frag<T, DKB>(qf[kb], Qs, 16 * qt, kb, lane);
                    row_frag<T, DKB>(dof[kb], dOs, 16 * qt, kb, lane);
                }
                f32x4 dqa[ND];
#pragma unroll
                for (int dt = 0; dt < ND; ++dt) dqa[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int kt = 0; kt < KTW; ++kt) {
                    const int kj0 = jw + 16 * kt;
                    float* ds = dss + kt * 16 * DSLD;
                    const bool dead = kj0 >= N || qi0 >= N ||
                                      (p.w >= 0 && (kj0 > qi0 + 15 + p.w || kj0 + 15 < qi0 - p.w));
                    const int rel = win_fixed ? (qi0 - kj0 + 16 * cmax) : (16 * qt - (kw0 + 16 * kt) + KB - 16);
                    f32x4 pv = {0.f, 0.f, 0.f, 0.f}, dsv = {0.f, 0.f, 0.f, 0.f};
                    if (!dead) {
                        const f32x4 raw = score_tile<T, DKB, BIAS, false>(qf, kf[kt], Pks, Pqs, rel, xs, ys, lane);
                        f32x4 dp = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                        for (int kb = 0; kb < DKB; ++kb) dp = mma(dof[kb], vf[kt][kb], dp);
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const int qr = 16 * qt + 4 * g + r;
                            const int qi = i0 + qr, kj = kj0 + li;
                            const bool ok = qi < N && kj < N && (p.w < 0 || abs(qi - kj) <= p.w);
                            const float pe = ok ? exp2f(raw[r] * sc2 - lse2s[qr]) : 0.f;
                            const float dm = (p.drop.thresh != 0 && ok)
                                                 ? drop_mul(p.drop, ((uint64_t)(b * p.H + h) * (uint64_t)N + (uint64_t)qi) * (uint64_t)N + (uint64_t)kj)
                                                 : 1.f;
                            pv[r] = pe * dm;
                            dsv[r] = pe * (dp[r] * dm - dls[qr]) * p.scale;
                        }
                    }
                    pt[qt][kt] = pv;
                    dst[qt][kt] = dsv;
#pragma unroll
                    for (int r = 0; r < 4; ++r) ds[(4 * g + r) * DSLD + li] = dsv[r];
                    wave_lds_sync();
                    if (BIAS && !dead) {
                        Frag<T> sx, sy;
                        skew_x<T>(sx, ds, g, li);
                        skew_y<T>(sy, ds, g, li);
#pragma unroll
                        for (int dt = 0; dt < ND; ++dt) {
                            Frag<T> a;
                            col_frag(a, Pks, ASTR, rel + 8 * g, rel + 8 * g + 4, 16 * dt, lane);
                            dqa[dt] = mma(a, sx, dqa[dt]);
                            col_frag(a, Pqs, ASTR, rel + 8 * g, rel + 8 * g + 4, 16 * dt, lane);
                            dkacc[kt][dt] = mma(a, sy, dkacc[kt][dt]);
                        }
                    }
                }
                // content part of dQ^T[d][ii] += sum_key K^T[d][key] dS^T[key][ii]
                {
                    Frag<T> bs;
                    const int kt_of_g = (KTW == 2) ? (g >> 1) : 0;
                    const float* ds = dss + kt_of_g * 16 * DSLD + li * DSLD + 8 * (g & 1);
                    const f32x4 lo = *reinterpret_cast<const f32x4*>(ds);
                    const f32x4 hi = *reinterpret_cast<const f32x4*>(ds + 4);
                    const bool use = (KTW == 2) || g < 2;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        frag_set(bs, e, use ? lo[e] : 0.f);
                        frag_set(bs, 4 + e, use ? hi[e] : 0.f);
                    }
                    const int krow = kw0 + ((KTW == 2) ? 8 * g : 8 * (g & 1));
#pragma unroll
                    for (int dt = 0; dt < ND; ++dt) {
                        Frag<T> a;
                        col_frag(a, Ks, ASTR, krow, krow + 4, 16 * dt, lane);
                        dqa[dt] = mma(a, bs, dqa[dt]);
                    }
                }
                // dqa[dt][r] = dQ^T[d = 16dt+4g+r][ii = li]: reduce over the workgroup's waves in LDS
#pragma unroll
                for (int dt = 0; dt < ND; ++dt)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        atomicAdd(&dQs[(16 * qt + li) * DQLD + 16 * dt + 4 * g + r], dqa[dt][r]);
                wave_lds_sync();
            }
            // content parts of dV^T[d][key] += dO^T[d][q] P[q][key], dK^T[d][key] += Q^T[d][q] dS[q][key]
#pragma unroll
            for (int kt = 0; kt < KTW; ++kt) {
                Frag<T> bp, bs;
                acc_to_frag(bp, pt[0][kt], pt[1][kt]);
                acc_to_frag(bs, dst[0][kt], dst[1][kt]);
#pragma unroll
                for (int dt = 0; dt < ND; ++dt) {
                    Frag<T> a;
                    col_frag(a, dOs, ASTR, 4 * g, 16 + 4 * g, 16 * dt, lane);
                    dvacc[kt][dt] = mma(a, bp, dvacc[kt][dt]);
                    col_frag(a, Qs, ASTR, 4 * g, 16 + 4 * g, 16 * dt, lane);
                    dkacc[kt][dt] = mma(a, bs, dkacc[kt][dt]);
                }
            }
        }
        __syncthreads();
        // dQ block -> fp32 accumulator in HBM (a wave instruction covers contiguous d of one row)
        for (int idx = tid; idx < 32 * dh; idx += NTH) {
            const int r = idx / dh, d = idx - r * dh;
            const int qi = i0 + r;
            if (qi < N) {
                const float v = dQs[r * DQLD + d];
                atomicAdd(p.dq_acc + ((long)b * N + qi) * p.ld_acc + h * dh + d, v);
            }
        }
    }

    // dK / dV of the wave's keys: acc[dt][r] = X^T[d = 16dt+4g+r][key = li]
#pragma unroll
    for (int kt = 0; kt < KTW; ++kt) {
        const int kj = jw + 16 * kt + li;
        if (kj < N) {
            T* dK = reinterpret_cast<T*>(p.dk) + ((long)b * N + kj) * p.ld_d + h * dh;
            T* dV = reinterpret_cast<T*>(p.dv) + ((long)b * N + kj) * p.ld_d + h * dh;
#pragma unroll
            for (int dt = 0; dt < ND; ++dt) {
                const int d = 16 * dt + 4 * g;
                if (d < dh) {
                    if (sizeof(T) == 2) {
                        const f32x4 a = dkacc[kt][dt], c = dvacc[kt][dt];
                        *reinterpret_cast<bf16x4*>(dK + d) = bf16x4{(__bf16)a[0], (__bf16)a[1], (__bf16)a[2], (__bf16)a[3]};
                        *reinterpret_cast<bf16x4*>(dV + d) = bf16x4{(__bf16)c[0], (__bf16)c[1], (__bf16)c[2], (__bf16)c[3]};
                    } else {
                        *reinterpret_cast<f32x4*>(dK + d) = dkacc[kt][dt];
                        *reinterpret_cast<f32x4*>(dV + d) = dvacc[kt][dt];
                    }
                }
            }
        }
    }
}

// ------------------------------------------------------------------ dPk / dPq
// Raw score tile with the relative-table fragments already in registers (the 32 table rows of
// a tile diagonal): same math as score_tile<.., BIAS = true, TRANS = false>.
template <typename T, int DKB>
__device__ __forceinline__ f32x4 score_tile_diag(const Frag<T> (&qf)[DKB], const Frag<T> (&kf)[DKB],
                                                 const Frag<T> (&pkf)[2][DKB], const Frag<T> (&pqf)[2][DKB],
                                                 float* xs, float* ys, int lane) {
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kb = 0; kb < DKB; ++kb) s = mma(qf[kb], kf[kb], s);
    const int g = lane >> 4, li = lane & 15;
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        f32x4 x = {0.f, 0.f, 0.f, 0.f}, y = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kb = 0; kb < DKB; ++kb) {
            x = mma(pkf[u][kb], qf[kb], x);
            y = mma(pqf[u][kb], kf[kb], y);
        }
        *reinterpret_cast<f32x4*>(xs + li * SCR_LD + 16 * u + 4 * g) = x;
        *reinterpret_cast<f32x4*>(ys + li * SCR_LD + 16 * u + 4 * g) = y;
    }
    wave_lds_sync();
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int ii = 4 * g + r, jj = li;
        const int t = ii - jj + 15;
        s[r] += xs[ii * SCR_LD + t] + ys[jj * SCR_LD + t];
    }
    wave_lds_sync();
    return s;
}

template <typename T, int DKB> struct RelCfg {
    using C = ACfg<T, DKB>;
    // per wave: Q pair + K pair images (32 rows each), xs, ys, two dS tiles
    static constexpr size_t wave_bytes() {
        return (size_t)64 * C::ASTR * sizeof(T) + (size_t)2 * SCR_FLOATS * sizeof(float) +
               (sizeof(T) == 2 ? (size_t)2 * DSXY_ELEMS * 2 : (size_t)2 * 16 * DSLD * sizeof(float));
    }
};

// Everything one tile pair needs from HBM, fetched one pair ahead of its use (the loads stay in
// flight under the previous pair's MFMAs and LDS round trips).
template <typename T, int DKB> struct RelPrefetch {
    static constexpr int CPR = ACfg<T, DKB>::CPR;
    static constexpr int CH = 32 * CPR / 64;     // 16-byte chunks per lane for a 32-row image
    u32x4 q[CH], k[CH];
};

template <typename T, int DKB>
__device__ __forceinline__ void rel_fetch(RelPrefetch<T, DKB>& pf, const AttnBwdParams& p, int h, int b, int i0,
                                          int j0, int ilim, int jlim, int lane) {
    using C = ACfg<T, DKB>;
    const int N = p.N, dh = p.dh;
    const T* Q = reinterpret_cast<const T*>(p.q) + (long)b * N * p.ld + h * dh;
    const T* K = reinterpret_cast<const T*>(p.k) + (long)b * N * p.ld + h * dh;
#pragma unroll
    for (int c = 0; c < RelPrefetch<T, DKB>::CH; ++c) {
        const int idx = lane + 64 * c;
        const int r = idx / C::CPR, d = (idx - r * C::CPR) * C::EPC;
        pf.q[c] = u32x4{0, 0, 0, 0};
        pf.k[c] = u32x4{0, 0, 0, 0};
        if (i0 + r < ilim && d < dh) pf.q[c] = *reinterpret_cast<const u32x4*>(Q + (long)(i0 + r) * p.ld + d);
        if (j0 + r < jlim && d < dh) pf.k[c] = *reinterpret_cast<const u32x4*>(K + (long)(j0 + r) * p.ld + d);
    }
}

template <typename T, int DKB>
__global__ __launch_bounds__(256, 2) void attn_bwd_rel_kernel(const AttnBwdParams p) {
    using C = ACfg<T, DKB>;
    using PF = RelPrefetch<T, DKB>;
    constexpr int ASTR = C::ASTR;
    constexpr int ND = 2 * DKB;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g = lane >> 4, li = lane & 15;
    char* wbase = smem + wave * RelCfg<T, DKB>::wave_bytes();
    T* Qw = reinterpret_cast<T*>(wbase);
    T* Kw = Qw + 32 * ASTR;
    float* xs = reinterpret_cast<float*>(Kw + 32 * ASTR);
    float* ys = xs + SCR_FLOATS;
    float* dss = ys + SCR_FLOATS;

    const int N = p.N, dh = p.dh;
    const int NT = (N + 15) / 16;
    // Work items.  Narrow LAI band: one live tile diagonal each (c = item - cmax).  Otherwise the
    // diagonals are paired (c, c - NT) so every item walks NT tiles.
    const int cmax = p.w >= 0 ? min((p.w + 15) / 16, NT - 1) : NT - 1;
    const bool banded = 2 * cmax + 1 < NT;
    const int nitems = banded ? 2 * cmax + 1 : NT;
    // (diagonal item, head) pairs are packed 4 to a workgroup: every wave of a workgroup has work
    const int wid = blockIdx.x * 4 + wave;
    const int h = wid / nitems;
    const int item = wid - h * nitems;
    const int b_lo = blockIdx.z * p.bgroup, b_hi = min(p.B, b_lo + p.bgroup);
    if (h >= p.H || b_lo >= b_hi) return;   // whole wave leaves; no workgroup barriers below
    constexpr bool PADDED = sizeof(T) == 2;   // bf16: zero-padded dS copies (attn_common.h)
    __bf16* dsx = reinterpret_cast<__bf16*>(dss);
    if (PADDED) {
        for (int i = lane; i < DSXY_ELEMS; i += 64) reinterpret_cast<float*>(dss)[i] = 0.f;   // 2 tiles x 1568 bf16
        wave_lds_sync();
    }

    const float sc2 = p.scale * 1.4426950408889634f;
    const T* PK = reinterpret_cast<const T*>(p.pk) + h * dh;
    const T* PQ = reinterpret_cast<const T*>(p.pq) + h * dh;

    for (int pass = 0; pass < (banded ? 1 : 2); ++pass) {
        const int c = banded ? item - cmax : (pass == 0 ? item : item - NT);
        if (!banded && pass == 1 && item == 0) break;
        const int ac = c < 0 ? -c : c;
        if (p.w >= 0 && 16 * ac - 15 > p.w) continue;   // whole diagonal outside the LAI band
        const int ntile = NT - ac;
        const int npair = (ntile + 1) / 2;
        const int ibase = 16 * (c > 0 ? c : 0), jbase = 16 * (c < 0 ? -c : 0);
        const int ilim = min(N, ibase + 16 * ntile), jlim = min(N, jbase + 16 * ntile);
        const int R0 = 16 * c - 15 + (N - 1);            // table row of t = 0

        Frag<T> pkf[2][DKB], pqf[2][DKB];
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int kb = 0; kb < DKB; ++kb) {
                const int row = R0 + 16 * u + li;
                const int d = 32 * kb + 8 * g;
                if (row >= 0 && row < 2 * N - 1 && d < dh) {
                    frag_load(pkf[u][kb], PK + (long)row * p.ld_p + d);
                    frag_load(pqf[u][kb], PQ + (long)row * p.ld_p + d);
                } else {
                    frag_zero(pkf[u][kb]);
                    frag_zero(pqf[u][kb]);
                }
            }
        f32x4 apk[2][ND], apq[2][ND];
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int dt = 0; dt < ND; ++dt) {
                apk[u][dt] = f32x4{0.f, 0.f, 0.f, 0.f};
                apq[u][dt] = f32x4{0.f, 0.f, 0.f, 0.f};
            }

        const int total = (b_hi - b_lo) * npair;
        PF pf;
        rel_fetch<T, DKB>(pf, p, h, b_lo, ibase, jbase, ilim, jlim, lane);
        for (int it = 0; it < total; ++it) {
            const int m = 2 * (it % npair);
            const int i0 = ibase + 16 * m, j0 = jbase + 16 * m;
            // prefetched rows -> this wave's LDS images (rows past the diagonal's end are zero)
#pragma unroll
            for (int cch = 0; cch < PF::CH; ++cch) {
                const int idx = lane + 64 * cch;
                const int r = idx / C::CPR, d = (idx - r * C::CPR) * C::EPC;
                *reinterpret_cast<u32x4*>(Qw + r * ASTR + d) = pf.q[cch];
                *reinterpret_cast<u32x4*>(Kw + r * ASTR + d) = pf.k[cch];
            }
            const int b = b_lo + it / npair;
            const T* V = reinterpret_cast<const T*>(p.v) + (long)b * N * p.ld + h * dh;
            const T* dO = reinterpret_cast<const T*>(p.dout) + (long)b * N * p.ldo + h * dh;
            const float* lse = p.lse + ((long)b * p.H + h) * N;
            const float* delta = p.delta + ((long)b * p.H + h) * N;
            Frag<T> vf[2][DKB], dof[2][DKB];
            float lse2[2][4], dl[2][4];
#pragma unroll
            for (int u = 0; u < 2; ++u) {
#pragma unroll
                for (int kb = 0; kb < DKB; ++kb) {
                    row_frag_global<T>(vf[u][kb], V, p.ld, j0 + 16 * u + li, N, dh, kb, lane);
                    row_frag_global<T>(dof[u][kb], dO, p.ldo, i0 + 16 * u + li, N, dh, kb, lane);
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int qc = min(i0 + 16 * u + 4 * g + r, N - 1);
                    lse2[u][r] = lse[qc] * 1.4426950408889634f;
                    dl[u][r] = delta[qc];
                }
            }
            if (it + 1 < total) {
                const int nb = b_lo + (it + 1) / npair, nm = 2 * ((it + 1) % npair);
                rel_fetch<T, DKB>(pf, p, h, nb, ibase + 16 * nm, jbase + 16 * nm, ilim, jlim, lane);
            }
            wave_lds_sync();
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int qi0 = i0 + 16 * u, kj0 = j0 + 16 * u;
                float* ds = dss + u * 16 * DSLD;
                f32x4 dsv = {0.f, 0.f, 0.f, 0.f};
                if (m + u < ntile) {
                    Frag<T> qf[DKB], kf[DKB];
#pragma unroll
                    for (int kb = 0; kb < DKB; ++kb) {
                        row_frag<T, DKB>(qf[kb], Qw, 16 * u, kb, lane);
                        row_frag<T, DKB>(kf[kb], Kw, 16 * u, kb, lane);
                    }
                    const f32x4 raw = score_tile_diag<T, DKB>(qf, kf, pkf, pqf, xs, ys, lane);
                    f32x4 dp = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int kb = 0; kb < DKB; ++kb) dp = mma(dof[u][kb], vf[u][kb], dp);
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int qi = qi0 + 4 * g + r, kj = kj0 + li;
                        const bool ok = qi < N && kj < N && (p.w < 0 || abs(qi - kj) <= p.w);
                        const float pe = ok ? exp2f(raw[r] * sc2 - lse2[u][r]) : 0.f;
                        const float dm = (p.drop.thresh != 0 && ok)
                                             ? drop_mul(p.drop, ((uint64_t)(b * p.H + h) * (uint64_t)N + (uint64_t)qi) * (uint64_t)N + (uint64_t)kj)
                                             : 1.f;
                        dsv[r] = pe * (dp[r] * dm - dl[u][r]) * p.scale;
                    }
                }
                if (PADDED) {
                    __bf16* X = dsx + u * DSXY_ELEMS;
                    __bf16* Y = X + DSX_ELEMS;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const __bf16 v = (__bf16)dsv[r];
                        X[(4 * g + r) * DSX_LD + 16 + li] = v;
                        Y[(15 + 4 * g + r) * DSY_LD + li] = v;
                    }
                } else {
#pragma unroll
                    for (int r = 0; r < 4; ++r) ds[(4 * g + r) * DSLD + li] = dsv[r];
                }
            }
            wave_lds_sync();
            // dPk[t][d] += sum_ii Sx[t][ii] Q[ii][d],  dPq[t][d] += sum_jj Sy[t][jj] K[jj][d]
            // K dimension = the pair's 32 rows: k = 8g+e -> tile (g >> 1), local row 8(g&1)+e
            {
                const float* ds = dss + (g >> 1) * 16 * DSLD;
                const int lbase = 8 * (g & 1);
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const int t = li + 16 * u;
                    Frag<T> ax, ay;
                    if (PADDED) {
                        const __bf16* X = dsx + (g >> 1) * DSXY_ELEMS;
                        const __bf16* Y = X + DSX_ELEMS;
#pragma unroll
                        for (int e = 0; e < 8; ++e) {
                            const int l = lbase + e;
                            frag_set(ax, e, (float)X[l * (DSX_LD + 1) + 31 - t]);   // dS[l][l + 15 - t]
                            frag_set(ay, e, (float)Y[l * (DSY_LD + 1) + t * DSY_LD]);   // dS[l + t - 15][l]
                        }
                    } else {
#pragma unroll
                        for (int e = 0; e < 8; ++e) {
                            const int l = lbase + e;
                            const int jj = l + 15 - t;   // Sx[t][ii = l] = dS[l][l + 15 - t]
                            const float vx = ds[l * DSLD + min(max(jj, 0), 15)];
                            frag_set(ax, e, (jj >= 0 && jj < 16) ? vx : 0.f);
                            const int ii = l + t - 15;   // Sy[t][jj = l] = dS[l + t - 15][l]
                            const float vy = ds[min(max(ii, 0), 15) * DSLD + l];
                            frag_set(ay, e, (ii >= 0 && ii < 16) ? vy : 0.f);
                        }
                    }
#pragma unroll
                    for (int dt = 0; dt < ND; ++dt) {
                        Frag<T> bq, bk;
                        col_frag(bq, Qw, ASTR, 8 * g, 8 * g + 4, 16 * dt, lane);
                        col_frag(bk, Kw, ASTR, 8 * g, 8 * g + 4, 16 * dt, lane);
                        apk[u][dt] = mma(ax, bq, apk[u][dt]);
                        apq[u][dt] = mma(ay, bk, apq[u][dt]);
                    }
                }
            }
            wave_lds_sync();
        }
        // flush: acc[u][dt][r] = dP*[t = 16u+4g+r][d = 16dt+li]
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int t = 16 * u + 4 * g + r;
                const int row = R0 + t;
                if (t < 31 && row >= 0 && row < 2 * N - 1) {
#pragma unroll
                    for (int dt = 0; dt < ND; ++dt) {
                        const int d = 16 * dt + li;
                        if (d < dh) {
                            atomicAdd(p.dpk + (long)row * p.ld_dp + h * dh + d, apk[u][dt][r]);
                            atomicAdd(p.dpq + (long)row * p.ld_dp + h * dh + d, apq[u][dt][r]);
                        }
                    }
                }
            }
    }
}

// ------------------------------------------------------------------ dPk / dPq from stored dS tiles
// Whole-head path: attn_bwd_band_kernel leaves every live 16 x 16 dS tile in a workspace
// (bf16, entry (b, h, c + cmax, kt), 512 bytes, lane l holds dS[4 (l >> 4) + r][l & 15], r = 0..3),
// so this kernel neither recomputes scores nor touches V / dO / LSE: per tile pair it reads the Q / K
// rows and two dS tiles, builds the skewed fragments from zero-padded LDS copies and runs the
// 16 table-gradient MFMAs.  Work decomposition and flush as attn_bwd_rel_kernel.
template <int DKB> struct RelDsCfg {
    using C = ACfg<__bf16, DKB>;
    static constexpr size_t wave_bytes() { return (size_t)64 * C::ASTR * 2 + (size_t)2 * DSXY_ELEMS * 2; }
};

template <int DKB>
__global__ __launch_bounds__(256, 2) void attn_bwd_relds_kernel(const AttnBwdParams p) {
    using T = __bf16;
    using C = ACfg<T, DKB>;
    constexpr int ASTR = C::ASTR;
    constexpr int ND = 2 * DKB;
    constexpr int CH = 32 * C::CPR / 64;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g = lane >> 4, li = lane & 15;
    char* wbase = smem + wave * RelDsCfg<DKB>::wave_bytes();
    T* Qw = reinterpret_cast<T*>(wbase);
    T* Kw = Qw + 32 * ASTR;
    T* dsx = Kw + 32 * ASTR;   // two tiles x (X copy, Y copy), attn_common.h

    const int N = p.N, dh = p.dh;
    const int NT = (N + 15) / 16;
    const int cmax = p.w >= 0 ? min((p.w + 15) / 16, NT - 1) : NT - 1;
    const int ndiag = 2 * cmax + 1;
    const bool banded = ndiag < NT;
    const int nitems = banded ? ndiag : NT;
    const int wid = blockIdx.x * 4 + wave;
    const int h = wid / nitems;
    const int item = wid - h * nitems;
    const int b_lo = blockIdx.z * p.bgroup, b_hi = min(p.B, b_lo + p.bgroup);
    if (h >= p.H || b_lo >= b_hi) return;   // whole wave leaves; no workgroup barriers below
    for (int i = lane; i < DSXY_ELEMS; i += 64) reinterpret_cast<float*>(dsx)[i] = 0.f;
    wave_lds_sync();

    for (int pass = 0; pass < (banded ? 1 : 2); ++pass) {
        const int c = banded ? item - cmax : (pass == 0 ? item : item - NT);
        if (!banded && pass == 1 && item == 0) break;
        const int ac = c < 0 ? -c : c;
        if (ac > cmax) continue;                         // diagonal outside the band: no tiles stored
        const int ntile = NT - ac;
        const int npair = (ntile + 1) / 2;
        const int ibase = 16 * (c > 0 ? c : 0), jbase = 16 * (c < 0 ? -c : 0);
        const int ilim = min(N, ibase + 16 * ntile), jlim = min(N, jbase + 16 * ntile);
        const int R0 = 16 * c - 15 + (N - 1);            // table row of t = 0
        const int kt0 = jbase >> 4;

        f32x4 apk[2][ND], apq[2][ND];
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int dt = 0; dt < ND; ++dt) {
                apk[u][dt] = f32x4{0.f, 0.f, 0.f, 0.f};
                apq[u][dt] = f32x4{0.f, 0.f, 0.f, 0.f};
            }

        const int total = (b_hi - b_lo) * npair;
        u32x4 pq_[CH], pk_[CH];
        bf16x4 pds[2];
        auto fetch = [&](int it) {
            const int b = b_lo + it / npair, m = 2 * (it % npair);
            const int i0 = ibase + 16 * m, j0 = jbase + 16 * m;
            const T* Q = reinterpret_cast<const T*>(p.q) + (long)b * N * p.ld + h * dh;
            const T* K = reinterpret_cast<const T*>(p.k) + (long)b * N * p.ld + h * dh;
#pragma unroll
            for (int cc = 0; cc < CH; ++cc) {
                const int idx = lane + 64 * cc;
                const int r = idx / C::CPR, d = (idx - r * C::CPR) * C::EPC;
                pq_[cc] = u32x4{0, 0, 0, 0};
                pk_[cc] = u32x4{0, 0, 0, 0};
                if (i0 + r < ilim && d < dh) pq_[cc] = *reinterpret_cast<const u32x4*>(Q + (long)(i0 + r) * p.ld + d);
                if (j0 + r < jlim && d < dh) pk_[cc] = *reinterpret_cast<const u32x4*>(K + (long)(j0 + r) * p.ld + d);
            }
            const __bf16* ws = p.ds_ws + ((((long)b * p.H + h) * ndiag + (c + cmax)) * NT + kt0 + m) * 256 + lane * 4;
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                pds[u] = bf16x4{};
                if (m + u < ntile) pds[u] = *reinterpret_cast<const bf16x4*>(ws + u * 256);
            }
        };
        fetch(0);
        for (int it = 0; it < total; ++it) {
#pragma unroll
            for (int cc = 0; cc < CH; ++cc) {
                const int idx = lane + 64 * cc;
                const int r = idx / C::CPR, d = (idx - r * C::CPR) * C::EPC;
                *reinterpret_cast<u32x4*>(Qw + r * ASTR + d) = pq_[cc];
                *reinterpret_cast<u32x4*>(Kw + r * ASTR + d) = pk_[cc];
            }
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                __bf16* X = dsx + u * DSXY_ELEMS;
                __bf16* Y = X + DSX_ELEMS;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    X[(4 * g + r) * DSX_LD + 16 + li] = pds[u][r];
                    Y[(15 + 4 * g + r) * DSY_LD + li] = pds[u][r];
                }
            }
            if (it + 1 < total) fetch(it + 1);
            wave_lds_sync();
            // dPk[t][d] += sum_ii Sx[t][ii] Q[ii][d],  dPq[t][d] += sum_jj Sy[t][jj] K[jj][d]
            // K dimension = the pair's 32 rows: k = 8g+e -> tile (g >> 1), local row 8(g&1)+e
            {
                const __bf16* X = dsx + (g >> 1) * DSXY_ELEMS;
                const __bf16* Y = X + DSX_ELEMS;
                const int lbase = 8 * (g & 1);
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const int t = li + 16 * u;
                    Frag<T> ax, ay;
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const int l = lbase + e;
                        ax.v[e] = X[l * (DSX_LD + 1) + 31 - t];        // Sx[t][l] = dS[l][l + 15 - t]
                        ay.v[e] = Y[l * (DSY_LD + 1) + t * DSY_LD];    // Sy[t][l] = dS[l + t - 15][l]
                    }
#pragma unroll
                    for (int dt = 0; dt < ND; ++dt) {
                        Frag<T> bq, bk;
                        col_frag(bq, Qw, ASTR, 8 * g, 8 * g + 4, 16 * dt, lane);
                        col_frag(bk, Kw, ASTR, 8 * g, 8 * g + 4, 16 * dt, lane);
                        apk[u][dt] = mma(ax, bq, apk[u][dt]);
                        apq[u][dt] = mma(ay, bk, apq[u][dt]);
                    }
                }
            }
            wave_lds_sync();
        }
        // flush: acc[u][dt][r] = dP*[t = 16u+4g+r][d = 16dt+li]
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int t = 16 * u + 4 * g + r;
                const int row = R0 + t;
                if (t < 31 && row >= 0 && row < 2 * N - 1) {
#pragma unroll
                    for (int dt = 0; dt < ND; ++dt) {
                        const int d = 16 * dt + li;
                        if (d < dh) {
                            atomicAdd(p.dpk + (long)row * p.ld_dp + h * dh + d, apk[u][dt][r]);
                            atomicAdd(p.dpq + (long)row * p.ld_dp + h * dh + d, apq[u][dt][r]);
                        }
                    }
                }
            }
    }
}

// fp32 accumulator rows -> T rows (bf16 path's dQ)
__global__ void cast_rows_kernel(const float* __restrict__ src, long lds_, __bf16* __restrict__ dst, long ldd, long rows,
                                 int cols) {
    const int c4 = cols >> 2;
    const long total = rows * c4;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long r = i / c4;
        const int c = (int)(i - r * c4) * 4;
        const f32x4 v = *reinterpret_cast<const f32x4*>(src + r * lds_ + c);
        *reinterpret_cast<bf16x4*>(dst + r * ldd + c) = bf16x4{(__bf16)v[0], (__bf16)v[1], (__bf16)v[2], (__bf16)v[3]};
    }
}

template <typename T, int DKB, int KTW, int NW, bool BIAS>
static void bwd_launch(const AttnBwdParams& p, hipStream_t st) {
    using BC = BwdCfg<T, DKB, KTW, NW, BIAS>;
    static_assert(BC::lds_bytes() <= 160 * 1024, "attention backward tile does not fit the LDS");
    const size_t lds = BC::lds_bytes();
    (void)vitk_lds_optin((const void*)attn_bwd_kernel<T, DKB, KTW, NW, BIAS>, lds);
    dim3 grid((p.N + BC::KB - 1) / BC::KB, p.H, p.B);
    hipLaunchKernelGGL((attn_bwd_kernel<T, DKB, KTW, NW, BIAS>), grid, dim3(BC::NT), lds, st, p);
}

template <typename T, int DKB> static void rel_launch(const AttnBwdParams& p, int zgroups, hipStream_t st) {
    const size_t lds = 4 * RelCfg<T, DKB>::wave_bytes();
    (void)vitk_lds_optin((const void*)attn_bwd_rel_kernel<T, DKB>, lds);
    const int NT = (p.N + 15) / 16;
    int cmax = p.w >= 0 ? (p.w + 15) / 16 : NT - 1;
    if (cmax > NT - 1) cmax = NT - 1;
    const int nitems = (2 * cmax + 1 < NT) ? 2 * cmax + 1 : NT;
    dim3 grid((nitems * p.H + 3) / 4, 1, zgroups);
    hipLaunchKernelGGL((attn_bwd_rel_kernel<T, DKB>), grid, dim3(256), lds, st, p);
}

template <int DKB> static void relds_launch(const AttnBwdParams& p, int zgroups, hipStream_t st) {
    const size_t lds = 4 * RelDsCfg<DKB>::wave_bytes();
    (void)vitk_lds_optin((const void*)attn_bwd_relds_kernel<DKB>, lds);
    const int NT = (p.N + 15) / 16;
    int cmax = p.w >= 0 ? (p.w + 15) / 16 : NT - 1;
    if (cmax > NT - 1) cmax = NT - 1;
    const int nitems = (2 * cmax + 1 < NT) ? 2 * cmax + 1 : NT;
    dim3 grid((nitems * p.H + 3) / 4, 1, zgroups);
    hipLaunchKernelGGL((attn_bwd_relds_kernel<DKB>), grid, dim3(256), lds, st, p);
}

template <typename T, int KTW> static void bwd_dispatch(const AttnBwdParams& p, int zgroups, hipStream_t st) {
    const int dkb = (p.dh + 31) / 32;
    const bool bias = p.pk != nullptr;
    // fp32 tiles of wide heads (and dh > 96 in bf16) only fit the LDS with a 2-wave workgroup
#define VITK_CASE(D)                                                               \
    case D: {                                                                      \
        constexpr int NW = ((sizeof(T) == 4 && D >= 3) || D >= 4) ? 2 : 4;         \
        if (bias) {                                                                \
            bwd_launch<T, D, KTW, NW, true>(p, st);                                \
            rel_launch<T, D>(p, zgroups, st);                                      \
        } else {                                                                   \
            bwd_launch<T, D, KTW, NW, false>(p, st);                               \
        }                                                                          \
    } break;
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

extern "C" int vitk_colsum(int dtype, int M, int N, const void* X, long ldx, float* out, hipStream_t stream);

static bool b16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

// Bytes of workspace vitk_attn_bwd needs: delta [B*H*N] fp32, then the larger of the fp32 dQ
// accumulator of the generic bf16 path and the dS tiles of the whole-head path (worst case: every
// tile diagonal live, 512 bytes per 16 x 16 tile).
extern "C" long vitk_attn_bwd_workspace_bytes(int dtype, int B, int H, int N, int dh) {
    long n = ((long)B * H * N + 63) / 64 * 64 * 4;
    if (dtype == VITK_BF16) {
        long extra = (long)B * N * H * dh * 4;
        if (band_eligible(dtype, N, dh, true, -1)) {
            const long NT = (N + 15) / 16;
            const long ds = (long)B * H * (2 * NT - 1) * NT * 512;
            if (ds > extra) extra = ds;
        }
        n += extra;
    }
    return n;
}

extern "C" int vitk_attn_bwd(int dtype, int B, int H, int N, int dh, const void* q, const void* k, const void* v,
                             long ld, const void* pk, const void* pq, long ld_p, float scale, int band_w,
                             const void* out, const void* dout, long ldo, const float* lse, void* dq, void* dk,
                             void* dv, long ld_d, float* dpk, float* dpq, long ld_dp, void* workspace,
                             float drop_p, unsigned long long drop_seed, unsigned long long drop_offset, float* dbias,
                             hipStream_t stream) {
    VITK_CHECK_ARG(dtype == VITK_F32 || dtype == VITK_BF16, "vitk_attn_bwd: bad dtype");
    VITK_CHECK_ARG(drop_p >= 0.f && drop_p < 1.f, "vitk_attn_bwd: dropout p must be in [0, 1)");
    VITK_CHECK_ARG(B > 0 && H > 0 && N > 0, "vitk_attn_bwd: empty problem");
    const int esz = dtype == VITK_BF16 ? 2 : 4;
    VITK_CHECK_ARG(dh > 0 && dh <= 128 && (dh * esz) % 16 == 0, "vitk_attn_bwd: head dim %d unsupported", dh);
    VITK_CHECK_ARG(q && k && v && out && dout && lse && dq && dk && dv && workspace, "vitk_attn_bwd: null pointer");
    VITK_CHECK_ARG((pk == nullptr) == (pq == nullptr), "vitk_attn_bwd: pk and pq go together");
    VITK_CHECK_ARG(!pk || (dpk && dpq), "vitk_attn_bwd: dpk/dpq required with pk/pq");
    const long HD = (long)H * dh;
    VITK_CHECK_ARG(b16(q) && b16(k) && b16(v) && b16(out) && b16(dout) && b16(dq) && b16(dk) && b16(dv) &&
                       (ld * esz) % 16 == 0 && (ldo * esz) % 16 == 0 && (ld_d * esz) % 16 == 0 && ld >= HD &&
                       ldo >= HD && ld_d >= HD,
                   "vitk_attn_bwd: rows must be 16-byte aligned and at least H*dh wide");
    VITK_CHECK_ARG(!pk || (b16(pk) && b16(pq) && (ld_p * esz) % 16 == 0 && ld_p >= HD && ld_dp >= HD),
                   "vitk_attn_bwd: band tables must be 16-byte aligned rows of at least H*dh");
    VITK_CHECK_ARG(b16(workspace), "vitk_attn_bwd: workspace must be 16-byte aligned");
    VITK_CHECK_ARG(H <= 65535 && B <= 65535, "vitk_attn_bwd: grid too large");

    AttnBwdParams p;
    p.q = q; p.k = k; p.v = v; p.ld = ld;
    p.pk = pk; p.pq = pq; p.ld_p = ld_p;
    p.out = out; p.dout = dout; p.ldo = ldo;
    p.lse = lse;
    p.delta = reinterpret_cast<float*>(workspace);
    p.dk = dk; p.dv = dv; p.ld_d = ld_d;
    p.dpk = dpk; p.dpq = dpq; p.ld_dp = ld_dp;
    p.B = B; p.H = H; p.N = N; p.dh = dh;
    p.scale = scale; p.w = band_w;
    p.drop = vitk_drop_params(drop_p, drop_seed, drop_offset);
    // batches per wave in the table-gradient kernel: enough waves to fill the chip (the kernel is
    // LDS-limited to ~8 waves per CU), few atomic flushes (one per wave)
    const int NT = (N + 15) / 16;
    int cmax = band_w >= 0 ? (band_w + 15) / 16 : NT - 1;
    if (cmax > NT - 1) cmax = NT - 1;
    const int nitems = (2 * cmax + 1 < NT) ? 2 * cmax + 1 : NT;
    // the kernel runs 2 workgroups (8 waves) per CU: aim at ONE full round of 512 workgroups of 4
    // (item, head) waves each, so no wave slot idles behind a partial second round
    const int blocks_per_group = (nitems * H + 3) / 4;
    int zgroups = 512 / blocks_per_group;
    if (zgroups > B) zgroups = B;
    if (zgroups < 1) zgroups = 1;
    p.bgroup = (B + zgroups - 1) / zgroups;
    zgroups = (B + p.bgroup - 1) / p.bgroup;

    const long rows = (long)B * N;
    if (band_eligible(dtype, N, dh, pk != nullptr, band_w)) {
        // whole-head kernels (attention_band.hip): dQ / dK / dV without atomics or accumulators;
        // delta is formed while dO is staged and left in the workspace for the table-gradient kernel
        BandParams bp = {};
        bp.q = q; bp.k = k; bp.v = v; bp.ld = ld;
        bp.pk = pk; bp.pq = pq; bp.ld_p = ld_p;
        bp.out = const_cast<void*>(out); bp.ldo = ldo; bp.lse = const_cast<float*>(lse);
        bp.dout = dout; bp.delta = p.delta;
        bp.dq = dq; bp.dk = dk; bp.dv = dv; bp.ld_d = ld_d;
        bp.B = B; bp.H = H; bp.N = N; bp.dh = dh;
        bp.scale = scale; bp.w = band_w;
        bp.drop = p.drop;
        bp.dbias = dbias;
        // pass 1 leaves its dS tiles behind delta: for pass 2 and for the table-gradient kernel
        bp.ds_ws = reinterpret_cast<__bf16*>(p.delta + ((long)B * H * N + 63) / 64 * 64);
        p.ds_ws = bp.ds_ws;
        bp.dpk = dpk; bp.dpq = dpq; bp.ld_dp = ld_dp;
        const bool tables_done = band_launch(bp, true, stream);     // attention_lai.hip sums the table gradients itself
        if (pk && !tables_done) {
            const int dkb = (dh + 31) / 32;
            if (dkb == 1) relds_launch<1>(p, zgroups, stream);
            else if (dkb == 2) relds_launch<2>(p, zgroups, stream);
            else relds_launch<3>(p, zgroups, stream);
        }
        VITK_CHECK_LAUNCH("vitk_attn_bwd");
        return VITK_OK;
    }
    float* delta_end = p.delta + ((long)B * H * N + 63) / 64 * 64;
    if (dtype == VITK_BF16) {
        p.dq_acc = delta_end;
        p.ld_acc = HD;
        if (vitk_zero_async(p.dq_acc, (size_t)HD * 4, (size_t)HD * 4, (size_t)rows, stream) != VITK_OK) return VITK_ERR_LAUNCH;
    } else {
        p.dq_acc = reinterpret_cast<float*>(dq);
        p.ld_acc = ld_d;
        if (vitk_zero_async(dq, (size_t)ld_d * 4, (size_t)HD * 4, (size_t)rows, stream) != VITK_OK) return VITK_ERR_LAUNCH;
    }
    {
        const long total = rows * H;
        long blocks = (total + 255) / 256;
        if (blocks > 8192) blocks = 8192;
        if (dtype == VITK_BF16)
            hipLaunchKernelGGL(attn_delta_kernel<__bf16>, dim3((int)blocks), dim3(256), 0, stream, (const __bf16*)out,
                               (const __bf16*)dout, ldo, p.delta, B, H, N, dh);
        else
            hipLaunchKernelGGL(attn_delta_kernel<float>, dim3((int)blocks), dim3(256), 0, stream, (const float*)out,
                               (const float*)dout, ldo, p.delta, B, H, N, dh);
    }
    if (dtype == VITK_BF16) {
        bwd_dispatch<__bf16, 2>(p, zgroups, stream);
        long blocks = (rows * (HD / 4) + 255) / 256;
        if (blocks > 8192) blocks = 8192;
        hipLaunchKernelGGL(cast_rows_kernel, dim3((int)blocks), dim3(256), 0, stream, p.dq_acc, HD, (__bf16*)dq, ld_d,
                           rows, (int)HD);
    } else {
        bwd_dispatch<float, 1>(p, zgroups, stream);
    }
    VITK_CHECK_LAUNCH("vitk_attn_bwd");
    if (dbias) {     // the generic kernels keep no per-head sums: three column-sum passes over dq / dk / dv
        const void* parts[3] = {dq, dk, dv};
        for (int i = 0; i < 3; ++i) {
            const int st = vitk_colsum(dtype, (int)rows, (int)HD, parts[i], ld_d, dbias + (long)i * HD, stream);
            if (st != VITK_OK) return st;
        }
    }
    return VITK_OK;
}
