// FAVOR+ (Performer) random-feature attention: feature maps and the linear-attention contraction,
// forward and backward, as MFMA products with wave-reduced normalisers.
//
// Replaces PerformerAttention.forward / softmax_kernel_transformation / relu_kernel_transformation
// (timm_classes/timm_vision_performer.py:26-101) and MultiheadAttentionPerformer.forward
// (model.py:257-352) between the qkv projection and the output projection, quirk for quirk:
//   dd   = data . w^T                      (data is NOT pre-scaled by dh^-1/4)
//   diag = (sum_d data_d) / 2 * dh^-1/2    (plain sum: the reference overwrites its square, :76-78)
//   q'   = ratio * (exp(dd - diag - max_m dd) + eps)                       per row
//   k'   = ratio * (exp(dd - diag - max over the WHOLE [B,L,H,m] tensor) + eps)
//   relu : relu(ratio * dd) + eps
//   KV[b,h] = sum_l k'_l^T v_l,  ksum[b,h] = sum_l k'_l,  out_l = (q'_l KV) / (q'_l . ksum)
// and the gradients of all of it, including the gradient through BOTH stabilising maxima (the
// argmax element receives minus the sum of the row's / the tensor's exp-gradients).
//
// q, k, v, out, dout, dq, dk, dv: element (b, l, h, d) at ((b*L + l) * ld + h*dh + d) -- views into
// packed [B*L, 3D] / [B*L, D] buffers, as for the softmax attention kernels.
// One workgroup = one (b, h) for every kernel that reduces over l; its 4 waves split l.
//
// Orientation: most products run with the token l on the MFMA LANE (dd^T[m][l] = w . data^T), so
// per-token scalars (row max, diag, den) are lane-local and the feature tiles are directly the B
// operand of the next product (sum over m).  Products that sum over l (KV, dKV) recompute dd with
// l in the accumulator ROWS and take V / dO through a wave-private LDS image (hardware transposed
// read for bf16).
#include "favor_common.h"

namespace vitk {

// ------------------------------------------------------------------ global key max (+ argmax)
template <typename T, int DKB, int MT>
__global__ __launch_bounds__(256) void favor_kmax_kernel(const FavorParams p) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int g = lane >> 4, li = lane & 15;
    const int h = blockIdx.x, b = blockIdx.y;
    const int L = p.L, dh = p.dh, m = p.m;
    const T* K = reinterpret_cast<const T*>(p.k) + (long)b * L * p.ld + h * dh;
    const T* W = reinterpret_cast<const T*>(p.w);
    unsigned long long best = 0ull;
    for (int l0 = 16 * wave; l0 < L; l0 += 64) {
        Frag<T> xf[DKB];
#pragma unroll
        for (int kb = 0; kb < DKB; ++kb) row_frag_global<T>(xf[kb], K, p.ld, l0 + li, L, dh, kb, lane);
        f32x4 dd[MT];
        ddT_tiles<T, DKB, MT>(dd, xf, W, m, dh, lane);
        const int l = l0 + li;
        if (l < L) {
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int mi = 16 * mt + 4 * g + r;
                    if (mi < m) {
                        const unsigned long long idx = (((unsigned long long)b * L + l) * p.H + h) * m + mi;
                        // ties: the smaller flat index wins (stored inverted)
                        const unsigned long long key = ((unsigned long long)ordered_bits(dd[mt][r]) << 32) |
                                                       (0xffffffffull - idx);
                        best = key > best ? key : best;
                    }
                }
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const unsigned long long other = __shfl_xor(best, o, 64);
        best = other > best ? other : best;
    }
    // most waves lose against the running maximum: read it first (a stale value only costs a spare atomic)
    if (lane == 0 && best > __atomic_load_n(p.kmax, __ATOMIC_RELAXED)) atomicMax(p.kmax, best);
}

// ------------------------------------------------------------------ KV, ksum
// grid (m chunks of 16*MTC features, H, B).  dd is computed with tokens in the accumulator rows:
// acc[r] = dd[token 4g + r][feature li]; k' tiles of two token tiles are the B operand of
// KV^T[d][m] += V^T[d][l] k'[l][m] (A = transposed read of the wave's V image).
template <typename T, int DKB, int MTC>
__global__ __launch_bounds__(256) void favor_kv_kernel(const FavorParams p) {
    using C = ACfg<T, DKB>;
    constexpr int ASTR = C::ASTR;
    constexpr int ND = 2 * DKB;
    constexpr int RLD = 16 * MTC + 4;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g = lane >> 4, li = lane & 15;
    const int mc0 = blockIdx.x * 16 * MTC, h = blockIdx.y, b = blockIdx.z;
    const int L = p.L, dh = p.dh, m = p.m;
    T* Vw = reinterpret_cast<T*>(smem) + wave * 32 * ASTR;
    float* red = reinterpret_cast<float*>(reinterpret_cast<T*>(smem) + 4 * 32 * ASTR);   // [32*DKB][RLD] + ksum[16*MTC]
    float* ksr = red + 32 * DKB * RLD;
    for (int i = tid; i < 32 * DKB * RLD + 16 * MTC; i += 256) red[i] = 0.f;
    __syncthreads();

    const T* K = reinterpret_cast<const T*>(p.k) + (long)b * L * p.ld + h * dh;
    const T* V = reinterpret_cast<const T*>(p.v) + (long)b * L * p.ld + h * dh;
    const T* W = reinterpret_cast<const T*>(p.w);
    float stab = 0.f;
    if (p.kind == FAVOR_SOFTMAX) stab = unordered_bits((unsigned int)(*p.kmax >> 32));

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
    for (int l0 = 32 * wave; l0 < L; l0 += 128) {
        stage_rows<T, DKB>(Vw, V, p.ld, l0, 32, 0, L, dh, lane, 64);
        f32x4 kp[2][MTC];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            Frag<T> xf[DKB];
            f32x4 dg = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int kb = 0; kb < DKB; ++kb) {
                row_frag_global<T>(xf[kb], K, p.ld, l0 + 16 * u + li, L, dh, kb, lane);
                dg = mma(xf[kb], ones, dg);            // row sums of K: dg[r] = sum_d K[4g+r][d]
            }
#pragma unroll
            for (int mt = 0; mt < MTC; ++mt) {
                f32x4 s = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int kb = 0; kb < DKB; ++kb) s = mma(xf[kb], wf[mt][kb], s);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int lt = l0 + 16 * u + 4 * g + r;
                    const bool ok = (lt < L) && (mc0 + 16 * mt + li < m);
                    float v = ok ? feat(p.kind, s[r], dg[r] * p.diag_c, stab, p.ratio, p.eps) : 0.f;
                    if (p.pk && ok) v += p.pk[((long)lt * p.H + h) * p.mp + mc0 + 16 * mt + li];
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
    }
    // reduce the 4 waves, one after the other (a fixed order: the forward is bit-reproducible; LDS atomics would add
    // in arrival order): acc[dt][mt][r] = KV^T[d = 16dt+4g+r][feature 16mt+li]
    for (int wv = 0; wv < 4; ++wv) {
        if (wave == wv) {
#pragma unroll
            for (int mt = 0; mt < MTC; ++mt) {
                float s = ks[mt];
                s += __shfl_xor(s, 16, 64);
                s += __shfl_xor(s, 32, 64);
                if (g == 0) ksr[16 * mt + li] += s;
#pragma unroll
                for (int dt = 0; dt < ND; ++dt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) red[(16 * dt + 4 * g + r) * RLD + 16 * mt + li] += acc[dt][mt][r];
            }
        }
        __syncthreads();
    }
    T* KVT = reinterpret_cast<T*>(p.kvt) + ((long)b * p.H + h) * dh * p.mp;
    T* KV = reinterpret_cast<T*>(p.kv) + ((long)b * p.H + h) * p.mp * p.dhp;
    float* KS = p.ksum + ((long)b * p.H + h) * p.mp;
    for (int i = tid; i < p.dhp * 16 * MTC; i += 256) {
        const int d = i / (16 * MTC), mm = i - d * 16 * MTC;
        const float v = d < dh ? red[d * RLD + mm] : 0.f;
        if (d < dh) KVT[(long)d * p.mp + mc0 + mm] = from_f32<T>(v);
        KV[(long)(mc0 + mm) * p.dhp + d] = from_f32<T>(v);
        p.kv32[(((long)b * p.H + h) * p.mp + mc0 + mm) * p.dhp + d] = v;
    }
    for (int i = tid; i < 16 * MTC; i += 256) KS[mc0 + i] = ksr[i];
}

// ------------------------------------------------------------------ out = (q' KV) / (q' . ksum)
// grid (ceil(L / 64), H, B); a wave owns 16 tokens, lane = token.
template <typename T, int DKB, int MT>
__global__ __launch_bounds__(256) void favor_out_kernel(const FavorParams p) {
    constexpr int ND = 2 * DKB;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int g = lane >> 4, li = lane & 15;
    const int l0 = blockIdx.x * 64 + 16 * wave, h = blockIdx.y, b = blockIdx.z;
    const int L = p.L, dh = p.dh, m = p.m;
    if (l0 >= L) return;
    const int l = l0 + li;
    const T* Q = reinterpret_cast<const T*>(p.q) + (long)b * L * p.ld + h * dh;
    const T* W = reinterpret_cast<const T*>(p.w);
    const T* KVT = reinterpret_cast<const T*>(p.kvt) + ((long)b * p.H + h) * dh * p.mp;
    const float* KS = p.ksum + ((long)b * p.H + h) * p.mp;

    Frag<T> xf[DKB];
#pragma unroll
    for (int kb = 0; kb < DKB; ++kb) row_frag_global<T>(xf[kb], Q, p.ld, l, L, dh, kb, lane);
    f32x4 dd[MT];
    ddT_tiles<T, DKB, MT>(dd, xf, W, m, dh, lane);
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
    const float* PQ = (p.pq && l < L) ? p.pq + ((long)l * p.H + h) * p.mp : nullptr;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        const f32x4 ksv = *reinterpret_cast<const f32x4*>(KS + 16 * mt + 4 * g);
        f32x4 pqv = {0.f, 0.f, 0.f, 0.f};
        if (PQ) pqv = *reinterpret_cast<const f32x4*>(PQ + 16 * mt + 4 * g);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const bool ok = 16 * mt + 4 * g + r < m;
            const float v = ok ? feat(p.kind, dd[mt][r], diag, mx, p.ratio, p.eps) + pqv[r] : 0.f;
            dd[mt][r] = v;
            den += v * ksv[r];
        }
    }
    den += __shfl_xor(den, 16, 64);
    den += __shfl_xor(den, 32, 64);
    const float inv = 1.f / den;
    T* O = reinterpret_cast<T*>(p.out) + ((long)b * L + l) * p.ldo + h * dh;
#pragma unroll
    for (int dt = 0; dt < ND; ++dt) {
        f32x4 num = {0.f, 0.f, 0.f, 0.f};
        const int d = 16 * dt + li;              // A row: KV^T[d][.]
#pragma unroll
        for (int mb = 0; mb < (MT + 1) / 2; ++mb) {
            Frag<T> a, bq;
            const bool has2 = 2 * mb + 1 < MT;
            if (d < dh) frag_load_split<T>(a, KVT + (long)d * p.mp + 32 * mb + 4 * g, KVT + (long)d * p.mp + 32 * mb + 16 + 4 * g, has2);
            else frag_zero(a);
            acc_to_frag(bq, dd[2 * mb], has2 ? dd[2 * mb + 1] : f32x4{0.f, 0.f, 0.f, 0.f});
            num = mma(a, bq, num);
        }
        const int dd0 = 16 * dt + 4 * g;
        if (l < L && dd0 < dh) store4<T>(O + dd0, num * inv);
    }
}

// ------------------------------------------------------------------ backward, query side
// One workgroup per (b, h); waves split the tokens in blocks of 16.
//   given dout: dnum = dout / den, dden = -(dout . out) / den
//   dq'[l][m] = KV[m] . dnum[l] + dden[l] ksum[m];   e = dq' * d feat / d dd
//   d dd = e - [m = argmax] sum_m e (softmax kernel);  d diag = -sum_m e
//   dQ[l] = d dd[l] . w + d diag * diag_c
//   dKV^T[d][m] += dO^T[d][l] q'[l][m] / den[l];  dksum[m] += dden[l] q'[l][m]
// launch bounds: two waves per SIMD.  Left alone the compiler took 200 VGPRs + 64 AGPRs -- eight over half the file, so ONE
// workgroup per CU: 768 workgroups (ViT-S/16-384, batch 64) ran in three rounds of one wave per SIMD (725 -> 685 us for
// the whole backward).  m = 256 keeps the whole file: at half of it that instance spills 1.4 KB per lane.
template <typename T, int DKB, int MT>
__global__ __launch_bounds__(256, MT <= 4 ? 2 : 1) void favor_bwd_q_kernel(const FavorParams p) {
    using C = ACfg<T, DKB>;
    constexpr int ASTR = C::ASTR;
    constexpr int ND = 2 * DKB;
    constexpr int MP = 16 * MT;
    constexpr int RLD = MP + 4;
    constexpr int WTLD = MP + 8;                  // elements per row of the transposed w image
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g = lane >> 4, li = lane & 15;
    const int h = blockIdx.x, b = blockIdx.y;
    const int L = p.L, dh = p.dh, m = p.m;
    T* wT = reinterpret_cast<T*>(smem);                          // [32*DKB][WTLD]: w^T, rows d
    T* dOw = wT + 32 * DKB * WTLD + wave * 16 * ASTR;            // wave-private dO image (16 rows)
    float* red = reinterpret_cast<float*>(wT + 32 * DKB * WTLD + 4 * 16 * ASTR);   // [32*DKB][RLD]
    float* dksr = red + 32 * DKB * RLD;                          // [MP]
    float* rowv = dksr + MP + wave * 64;                         // per wave: inv_den[16], dden[16], mx[16], diag[16]

    const T* W = reinterpret_cast<const T*>(p.w);
    for (int i = tid; i < 32 * DKB * WTLD; i += 256) {
        const int d = i / WTLD, mm = i - d * WTLD;
        wT[i] = (d < dh && mm < m) ? W[(long)mm * dh + d] : from_f32<T>(0.f);
    }
    for (int i = tid; i < 32 * DKB * RLD + MP; i += 256) red[i] = 0.f;
    __syncthreads();

    const T* Q = reinterpret_cast<const T*>(p.q) + (long)b * L * p.ld + h * dh;
    const T* OUT = reinterpret_cast<const T*>(p.out) + (long)b * L * p.ldo + h * dh;
    const T* DO = reinterpret_cast<const T*>(p.dout) + (long)b * L * p.ldo + h * dh;
    const T* KV = reinterpret_cast<const T*>(p.kv) + ((long)b * p.H + h) * p.mp * p.dhp;
    const float* KS = p.ksum + ((long)b * p.H + h) * p.mp;
    const float* KV32 = p.kv32 + ((long)b * p.H + h) * p.mp * p.dhp;

    f32x4 dkvt[ND][MT];
    float dks[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        dks[mt] = 0.f;
#pragma unroll
        for (int dt = 0; dt < ND; ++dt) dkvt[dt][mt] = f32x4{0.f, 0.f, 0.f, 0.f};
    }

    for (int l0 = 16 * wave; l0 < L; l0 += 64) {
        const int l = l0 + li;
        stage_rows<T, DKB>(dOw, DO, p.ldo, l0, 16, 0, L, dh, lane, 64);
        Frag<T> xf[DKB];
#pragma unroll
        for (int kb = 0; kb < DKB; ++kb) row_frag_global<T>(xf[kb], Q, p.ld, l, L, dh, kb, lane);
        // ---- phase A: lane = token -------------------------------------------------------------
        f32x4 dd[MT];
        ddT_tiles<T, DKB, MT>(dd, xf, W, m, dh, lane);
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
        f32x4 qp[MT];
        const float* PQ = (p.pq && l < L) ? p.pq + ((long)l * p.H + h) * p.mp : nullptr;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const f32x4 ksv = *reinterpret_cast<const f32x4*>(KS + 16 * mt + 4 * g);
            f32x4 pqv = {0.f, 0.f, 0.f, 0.f};
            if (PQ) pqv = *reinterpret_cast<const f32x4*>(PQ + 16 * mt + 4 * g);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const bool ok = 16 * mt + 4 * g + r < m;
                qp[mt][r] = ok ? feat(p.kind, dd[mt][r], diag, mx, p.ratio, p.eps) + pqv[r] : 0.f;
                den += qp[mt][r] * ksv[r];
            }
        }
        den += __shfl_xor(den, 16, 64);
        den += __shfl_xor(den, 32, 64);
        const float inv = (l < L) ? 1.f / den : 0.f;
        // dnum^T tiles (acc layout: rows d = 16dt+4g+r, column = token) and dden
        f32x4 dnum[ND];
        float dden = 0.f;
#pragma unroll
        for (int dt = 0; dt < ND; ++dt) {
            const int d0 = 16 * dt + 4 * g;
            f32x4 go = {0.f, 0.f, 0.f, 0.f}, oo = {0.f, 0.f, 0.f, 0.f};
            if (l < L && d0 < dh) {
                go = load4<T>(DO + (long)l * p.ldo + d0);
                oo = load4<T>(OUT + (long)l * p.ldo + d0);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) dden -= go[r] * oo[r];
            dnum[dt] = go * inv;
        }
        dden += __shfl_xor(dden, 16, 64);
        dden += __shfl_xor(dden, 32, 64);
        dden *= inv;
        // dq'^T[m][l] = sum_d KV[m][d] dnum^T[d][l] + dden ksum[m];  e = dq' * feat'
        float esum = 0.f;
        f32x4 e[MT];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            f32x4 s = {0.f, 0.f, 0.f, 0.f};
            const int mrow = 16 * mt + li;
#pragma unroll
            for (int kb = 0; kb < DKB; ++kb) {
                if constexpr (sizeof(T) == 2) {
                    const float* kr = KV32 + (long)mrow * p.dhp + 32 * kb + 4 * g;
                    Frag<__bf16> ah, alo, bh, blo;
                    acc_to_frag_hilo(ah, alo, *reinterpret_cast<const f32x4*>(kr), *reinterpret_cast<const f32x4*>(kr + 16));
                    acc_to_frag_hilo(bh, blo, dnum[2 * kb], dnum[2 * kb + 1]);
                    s = mma(ah, bh, s);
                    s = mma(ah, blo, s);
                    s = mma(alo, bh, s);
                } else {
                    Frag<T> a, bn;
                    frag_load_split<T>(a, KV + (long)mrow * p.dhp + 32 * kb + 4 * g, KV + (long)mrow * p.dhp + 32 * kb + 16 + 4 * g, true);
                    acc_to_frag(bn, dnum[2 * kb], dnum[2 * kb + 1]);
                    s = mma(a, bn, s);
                }
            }
            const f32x4 ksv = *reinterpret_cast<const f32x4*>(KS + 16 * mt + 4 * g);
            f32x4 dnew;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const bool ok = 16 * mt + 4 * g + r < m;
                const float dqp = s[r] + dden * ksv[r];
                dnew[r] = ok ? dqp : 0.f;
                e[mt][r] = ok ? dqp * feat_grad(p.kind, dd[mt][r], diag, mx, p.ratio) : 0.f;
                esum += e[mt][r];
            }
            if (p.dnq && l < L)
                *reinterpret_cast<f32x4*>(p.dnq + (((long)b * L + l) * p.H + h) * p.mp + 16 * mt + 4 * g) = dnew;
        }
        esum += __shfl_xor(esum, 16, 64);
        esum += __shfl_xor(esum, 32, 64);
        if (p.kind == FAVOR_SOFTMAX) {
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (16 * mt + 4 * g + r < m && dd[mt][r] == mx) e[mt][r] -= esum;
        }
        const float ddiag = p.kind == FAVOR_SOFTMAX ? -esum * p.diag_c : 0.f;
        // dQ^T[d][l] = sum_m w^T[d][m] ddd^T[m][l] + ddiag
        T* DQ = reinterpret_cast<T*>(p.dq) + ((long)b * L + l) * p.ld_d + h * dh;
#pragma unroll
        for (int dt = 0; dt < ND; ++dt) {
            f32x4 s = {ddiag, ddiag, ddiag, ddiag};
            const T* wrow = wT + (16 * dt + li) * WTLD;
#pragma unroll
            for (int mb = 0; mb < (MT + 1) / 2; ++mb) {
                const bool has2 = 2 * mb + 1 < MT;
                Frag<T> a, be;
                frag_load_split<T>(a, wrow + 32 * mb + 4 * g, wrow + 32 * mb + 16 + 4 * g, has2);
                acc_to_frag(be, e[2 * mb], has2 ? e[2 * mb + 1] : f32x4{0.f, 0.f, 0.f, 0.f});
                s = mma(a, be, s);
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
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            f32x4 s = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int kb = 0; kb < DKB; ++kb) {
                Frag<T> wf;
                row_frag_global<T>(wf, W, dh, 16 * mt + li, m, dh, kb, lane);
                s = mma(xf[kb], wf, s);                 // dd[token 4g+r][feature 16mt+li]
            }
            f32x4 qn;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int lr = 4 * g + r;
                const bool ok = (l0 + lr < L) && (16 * mt + li < m);
                float v = ok ? feat(p.kind, s[r], rowv[48 + lr], rowv[32 + lr], p.ratio, p.eps) : 0.f;
                if (p.pq && ok) v += p.pq[((long)(l0 + lr) * p.H + h) * p.mp + 16 * mt + li];
                dks[mt] += v * rowv[16 + lr];
                qn[r] = v * rowv[lr];
            }
            Frag<T> bq;
            acc_to_frag(bq, qn, f32x4{0.f, 0.f, 0.f, 0.f});     // 16 tokens: second half of K is zero
#pragma unroll
            for (int dt = 0; dt < ND; ++dt) {
                Frag<T> a;
                col_frag(a, dOw, ASTR, 4 * g, 4 * g, 16 * dt, lane);
                dkvt[dt][mt] = mma(a, bq, dkvt[dt][mt]);
            }
        }
        wave_lds_sync();
    }
    // reduce over the 4 waves and write dKV in both layouts (+ fp32 transposed copy) and dksum
    // the four waves add in turn, with plain LDS reads and writes: LDS float atomics from four waves onto the same
    // 4 K addresses took ~40 us per workgroup (most of this kernel's fixed cost)
    for (int wv = 0; wv < 4; ++wv) {
        if (wave == wv) {
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                float s = dks[mt];
                s += __shfl_xor(s, 16, 64);
                s += __shfl_xor(s, 32, 64);
                if (g == 0) dksr[16 * mt + li] += s;
#pragma unroll
                for (int dt = 0; dt < ND; ++dt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) red[(16 * dt + 4 * g + r) * RLD + 16 * mt + li] += dkvt[dt][mt][r];
            }
        }
        __syncthreads();
    }
    T* DKVT = reinterpret_cast<T*>(p.dkvt_t) + ((long)b * p.H + h) * dh * p.mp;
    T* DKV = reinterpret_cast<T*>(p.dkv_t) + ((long)b * p.H + h) * p.mp * p.dhp;
    float* DKS = p.dksum + ((long)b * p.H + h) * p.mp;
    for (int i = tid; i < p.dhp * MP; i += 256) {
        const int d = i / MP, mm = i - d * MP;
        const float v = d < dh ? red[d * RLD + mm] : 0.f;
        if (d < dh) DKVT[(long)d * p.mp + mm] = from_f32<T>(v);
        DKV[(long)mm * p.dhp + d] = from_f32<T>(v);
        p.dkv32[(((long)b * p.H + h) * p.mp + mm) * p.dhp + d] = v;
    }
    for (int i = tid; i < MP; i += 256) DKS[i] = dksr[i];
}

// ------------------------------------------------------------------ backward, key / value side
// grid (ceil(L / 64), H, B); lane = token.
//   dk'[l][m] = dKV[m] . v[l] + dksum[m];  e = dk' * feat';  esum(global) += sum e
//   dK[l] = e[l] . w - (sum_m e[l][m]) diag_c      (the global-max term is applied by favor_fix)
//   dV[l] = k'[l] . dKV
template <typename T, int DKB, int MT>
__global__ __launch_bounds__(256) void favor_bwd_kv_kernel(const FavorParams p) {
    constexpr int ND = 2 * DKB;
    constexpr int MP = 16 * MT;
    constexpr int WTLD = MP + 8;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g = lane >> 4, li = lane & 15;
    const int l0 = blockIdx.x * 64 + 16 * wave, h = blockIdx.y, b = blockIdx.z;
    const int L = p.L, dh = p.dh, m = p.m;
    T* wT = reinterpret_cast<T*>(smem);
    const T* W = reinterpret_cast<const T*>(p.w);
    for (int i = tid; i < 32 * DKB * WTLD; i += 256) {
        const int d = i / WTLD, mm = i - d * WTLD;
        wT[i] = (d < dh && mm < m) ? W[(long)mm * dh + d] : from_f32<T>(0.f);
    }
    __syncthreads();
    if (l0 >= L) return;
    const int l = l0 + li;
    const T* K = reinterpret_cast<const T*>(p.k) + (long)b * L * p.ld + h * dh;
    const T* V = reinterpret_cast<const T*>(p.v) + (long)b * L * p.ld + h * dh;
    const T* DKV = reinterpret_cast<const T*>(p.dkv_t) + ((long)b * p.H + h) * p.mp * p.dhp;
    const T* DKVT = reinterpret_cast<const T*>(p.dkvt_t) + ((long)b * p.H + h) * dh * p.mp;
    const float* DKS = p.dksum + ((long)b * p.H + h) * p.mp;
    const float* DKV32 = p.dkv32 + ((long)b * p.H + h) * p.mp * p.dhp;
    float stab = 0.f;
    if (p.kind == FAVOR_SOFTMAX) stab = unordered_bits((unsigned int)(*p.kmax >> 32));

    Frag<T> xf[DKB], vf[DKB];
#pragma unroll
    for (int kb = 0; kb < DKB; ++kb) {
        row_frag_global<T>(xf[kb], K, p.ld, l, L, dh, kb, lane);
        row_frag_global<T>(vf[kb], V, p.ld, l, L, dh, kb, lane);
    }
    f32x4 dd[MT];
    ddT_tiles<T, DKB, MT>(dd, xf, W, m, dh, lane);
    const float diag = row_sum<T, DKB>(xf) * p.diag_c;
    f32x4 kp[MT], e[MT];
    float esum = 0.f;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        f32x4 s = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kb = 0; kb < DKB; ++kb) {
            if constexpr (sizeof(T) == 2) {
                const float* dr = DKV32 + (long)(16 * mt + li) * p.dhp + 32 * kb + 8 * g;
                Frag<__bf16> ah, alo;
                acc_to_frag_hilo(ah, alo, *reinterpret_cast<const f32x4*>(dr), *reinterpret_cast<const f32x4*>(dr + 4));
                s = mma(ah, vf[kb], s);
                s = mma(alo, vf[kb], s);
            } else {
                Frag<T> a;
                row_frag_global<T>(a, DKV, p.dhp, 16 * mt + li, p.mp, p.dhp, kb, lane);   // rows m of dKV
                s = mma(a, vf[kb], s);                                                    // dk'^T[m][l]
            }
        }
        const f32x4 dksv = *reinterpret_cast<const f32x4*>(DKS + 16 * mt + 4 * g);
        f32x4 pkv = {0.f, 0.f, 0.f, 0.f}, dnew;
        if (p.pk && l < L) pkv = *reinterpret_cast<const f32x4*>(p.pk + ((long)l * p.H + h) * p.mp + 16 * mt + 4 * g);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const bool ok = (16 * mt + 4 * g + r < m) && (l < L);
            kp[mt][r] = ok ? feat(p.kind, dd[mt][r], diag, stab, p.ratio, p.eps) + pkv[r] : 0.f;
            dnew[r] = ok ? s[r] + dksv[r] : 0.f;
            e[mt][r] = ok ? dnew[r] * feat_grad(p.kind, dd[mt][r], diag, stab, p.ratio) : 0.f;
            esum += e[mt][r];
        }
        if (p.dnk && l < L)
            *reinterpret_cast<f32x4*>(p.dnk + (((long)b * L + l) * p.H + h) * p.mp + 16 * mt + 4 * g) = dnew;
    }
    float esum_l = esum;
    esum_l += __shfl_xor(esum_l, 16, 64);
    esum_l += __shfl_xor(esum_l, 32, 64);
    if (p.kind == FAVOR_SOFTMAX) {
        float tot = wave_sum(esum);
        if (lane == 0) atomicAdd(p.esum + (long)b * p.H + h, tot);
    }
    const float ddiag = p.kind == FAVOR_SOFTMAX ? -esum_l * p.diag_c : 0.f;
    T* DK = reinterpret_cast<T*>(p.dk) + ((long)b * L + l) * p.ld_d + h * dh;
    T* DV = reinterpret_cast<T*>(p.dv) + ((long)b * L + l) * p.ld_d + h * dh;
#pragma unroll
    for (int dt = 0; dt < ND; ++dt) {
        f32x4 sk = {ddiag, ddiag, ddiag, ddiag}, sv = {0.f, 0.f, 0.f, 0.f};
        const T* wrow = wT + (16 * dt + li) * WTLD;
        const int d = 16 * dt + li;
#pragma unroll
        for (int mb = 0; mb < (MT + 1) / 2; ++mb) {
            const bool has2 = 2 * mb + 1 < MT;
            const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
            Frag<T> a, be, a2, bk;
            frag_load_split<T>(a, wrow + 32 * mb + 4 * g, wrow + 32 * mb + 16 + 4 * g, has2);
            acc_to_frag(be, e[2 * mb], has2 ? e[2 * mb + 1] : zero);
            sk = mma(a, be, sk);
            if (d < dh) frag_load_split<T>(a2, DKVT + (long)d * p.mp + 32 * mb + 4 * g, DKVT + (long)d * p.mp + 32 * mb + 16 + 4 * g, has2);
            else frag_zero(a2);
            acc_to_frag(bk, kp[2 * mb], has2 ? kp[2 * mb + 1] : zero);
            sv = mma(a2, bk, sv);
        }
        const int d0 = 16 * dt + 4 * g;
        if (l < L && d0 < dh) {
            store4<T>(DK + d0, sk);
            store4<T>(DV + d0, sv);
        }
    }
}

// ------------------------------------------------------------------ position feature tables
// The feature map on its own, applied to the [L, H*dh] projections of pos_embed (one "batch"):
// P[l][h][.] fp32 with pitch mp (DisentangledSelfAttentionPerformer.disentangled_attention_perf_bias,
// disentangled_transformer.py:612-644).  One wave per (l, h); lane j owns features j, j+64, ...
// L*H*m elements in all: plain FMAs.
constexpr int POS_MJ = 4;   // 256 features / 64 lanes

template <typename T>
__device__ __forceinline__ void pos_dd(float (&dd)[POS_MJ], float& rsum, const T* __restrict__ x, const T* __restrict__ W,
                                       int dh, int m, int lane) {
    rsum = 0.f;
#pragma unroll
    for (int j = 0; j < POS_MJ; ++j) dd[j] = 0.f;
    // four elements per step (dh % 4 == 0, rows 8- / 16-byte aligned: host-checked): 4x fewer dependent loads
#pragma unroll 4
    for (int d = 0; d < dh; d += 4) {
        const f32x4 xv = load4<T>(x + d);
        rsum += (xv[0] + xv[1]) + (xv[2] + xv[3]);
#pragma unroll
        for (int j = 0; j < POS_MJ; ++j) {
            const int mi = lane + 64 * j;
            if (mi < m) {
                const f32x4 wv = load4<T>(W + (long)mi * dh + d);
                dd[j] += xv[0] * wv[0] + xv[1] * wv[1] + xv[2] * wv[2] + xv[3] * wv[3];
            }
        }
    }
}

// row max of the valid dd (query stabiliser) or the stored global max (key stabiliser)
__device__ __forceinline__ float pos_stab(const FavorParams& p, const float (&dd)[POS_MJ], int is_query, int lane) {
    if (p.kind != FAVOR_SOFTMAX) return 0.f;
    if (!is_query) return unordered_bits((unsigned int)(*p.kmax >> 32));
    float mx = -INFINITY;
#pragma unroll
    for (int j = 0; j < POS_MJ; ++j)
        if (lane + 64 * j < p.m) mx = fmaxf(mx, dd[j]);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
    return mx;
}

// pass 0: global max (+ argmax) of the key-side dd into p.kmax;  pass 1: the features
template <typename T>
__global__ __launch_bounds__(64) void favor_pos_fwd_kernel(const FavorParams p, int is_query, int pass, float* feat_out) {
    const int lane = threadIdx.x, h = blockIdx.x, l = blockIdx.y;
    const T* x = reinterpret_cast<const T*>(p.q) + (long)l * p.ld + h * p.dh;
    float dd[POS_MJ], rsum;
    pos_dd<T>(dd, rsum, x, reinterpret_cast<const T*>(p.w), p.dh, p.m, lane);
    if (pass == 0) {
        unsigned long long best = 0ull;
#pragma unroll
        for (int j = 0; j < POS_MJ; ++j) {
            const int mi = lane + 64 * j;
            if (mi < p.m) {
                const unsigned long long idx = ((unsigned long long)l * p.H + h) * p.m + mi;
                const unsigned long long key = ((unsigned long long)ordered_bits(dd[j]) << 32) | (0xffffffffull - idx);
                best = key > best ? key : best;
            }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const unsigned long long other = __shfl_xor(best, o, 64);
            best = other > best ? other : best;
        }
        if (lane == 0 && best > __atomic_load_n(p.kmax, __ATOMIC_RELAXED)) atomicMax(p.kmax, best);
        return;
    }
    const float diag = rsum * p.diag_c;
    const float stab = pos_stab(p, dd, is_query, lane);
    float* out = feat_out + ((long)l * p.H + h) * p.mp;
#pragma unroll
    for (int j = 0; j < POS_MJ; ++j) {
        const int mi = lane + 64 * j;
        if (mi < p.mp) out[mi] = mi < p.m ? feat(p.kind, dd[j], diag, stab, p.ratio, p.eps) : 0.f;
    }
}

// dx[l][h][.] from G[l][h][.] = sum_b dfeat[b][l][h][.]; the key side's global-max term is applied by favor_fix.
// One workgroup of 4 waves per (l, h): all lanes share the sum over b (with mp = 16 a wave covers 4 values of b
// per pass), wave 0 goes back through the feature map, every thread then takes columns of dx.
template <typename T>
__global__ __launch_bounds__(256) void favor_pos_bwd_kernel(const FavorParams p, int is_query, const float* __restrict__ dfeat,
                                                            int nsum) {
    __shared__ float gs[4][64 * POS_MJ];
    __shared__ float es[64 * POS_MJ];
    __shared__ float ddiag_s;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, h = blockIdx.x, l = blockIdx.y;
    const T* W = reinterpret_cast<const T*>(p.w);
    {
        const int fp = p.mp < 64 ? p.mp : 64;          // features a wave covers per pass (mp is 16, 64 or 256)
        const int per = 64 / fp;                       // values of b per wave and pass
        const int mi0 = lane % fp, bsub = lane / fp;
        float acc[POS_MJ];
#pragma unroll
        for (int j = 0; j < POS_MJ; ++j) acc[j] = 0.f;
        const float* src = dfeat + ((long)l * p.H + h) * p.mp;
        const long bstride = (long)p.L * p.H * p.mp;
        for (int b = wave * per + bsub; b < nsum; b += 4 * per) {
#pragma unroll
            for (int j = 0; j < POS_MJ; ++j)
                if (mi0 + 64 * j < p.mp) acc[j] += src[b * bstride + mi0 + 64 * j];
        }
        if (fp == 16) {
            acc[0] += __shfl_xor(acc[0], 16, 64);
            acc[0] += __shfl_xor(acc[0], 32, 64);
        }
        if (bsub == 0) {
#pragma unroll
            for (int j = 0; j < POS_MJ; ++j)
                if (mi0 + 64 * j < p.mp) gs[wave][mi0 + 64 * j] = acc[j];
        }
    }
    __syncthreads();
    if (wave == 0) {
        const T* x = reinterpret_cast<const T*>(p.q) + (long)l * p.ld + h * p.dh;
        float dd[POS_MJ], rsum;
        pos_dd<T>(dd, rsum, x, W, p.dh, p.m, lane);
        const float diag = rsum * p.diag_c;
        const float stab = pos_stab(p, dd, is_query, lane);
        float e[POS_MJ];
        float esum = 0.f;
#pragma unroll
        for (int j = 0; j < POS_MJ; ++j) {
            const int mi = lane + 64 * j;
            e[j] = 0.f;
            if (mi < p.m) {
                const float g = gs[0][mi] + gs[1][mi] + gs[2][mi] + gs[3][mi];
                e[j] = g * feat_grad(p.kind, dd[j], diag, stab, p.ratio);
                esum += e[j];
            }
        }
        esum = wave_sum(esum);
        float ddiag = 0.f;
        if (p.kind == FAVOR_SOFTMAX) {
            ddiag = -esum * p.diag_c;
            if (is_query) {
#pragma unroll
                for (int j = 0; j < POS_MJ; ++j)
                    if (lane + 64 * j < p.m && dd[j] == stab) e[j] -= esum;
            } else if (lane == 0) {
                p.esum[(long)l * p.H + h] = esum;
            }
        }
#pragma unroll
        for (int j = 0; j < POS_MJ; ++j) es[lane + 64 * j] = e[j];
        if (lane == 0) ddiag_s = ddiag;
    }
    __syncthreads();
    const float ddiag = ddiag_s;
    T* DX = reinterpret_cast<T*>(p.dk) + (long)l * p.ld_d + h * p.dh;
    for (int d = tid; d < p.dh; d += 256) {
        float sacc = ddiag;
        for (int mi = 0; mi < p.m; ++mi) sacc += es[mi] * to_f32<T>(W[(long)mi * p.dh + d]);
        DX[d] = from_f32<T>(sacc);
    }
}

}  // namespace vitk

using namespace vitk;

namespace {

bool f16al(const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; }

int pick_mt(int m) { return m <= 16 ? 1 : (m <= 64 ? 4 : 16); }

template <typename T, int DKB, int MT> void launch_fwd(const FavorParams& p, hipStream_t st) {
    using C = ACfg<T, DKB>;
    constexpr int MTC = MT >= 4 ? 4 : 1;
    if (p.kind == FAVOR_SOFTMAX) {
        (void)vitk_zero_async(p.kmax, 8, 8, 1, st);
        hipLaunchKernelGGL((favor_kmax_kernel<T, DKB, MT>), dim3(p.H, p.B), dim3(256), 0, st, p);
    }
    const size_t lds = (size_t)4 * 32 * C::ASTR * sizeof(T) + (size_t)(32 * DKB * (16 * MTC + 4) + 16 * MTC) * 4;
    (void)vitk_lds_optin((const void*)favor_kv_kernel<T, DKB, MTC>, lds);
    hipLaunchKernelGGL((favor_kv_kernel<T, DKB, MTC>), dim3(MT / MTC, p.H, p.B), dim3(256), lds, st, p);
    hipLaunchKernelGGL((favor_out_kernel<T, DKB, MT>), dim3((p.L + 63) / 64, p.H, p.B), dim3(256), 0, st, p);
}

template <typename T, int DKB, int MT> void launch_bwd(const FavorParams& p, hipStream_t st) {
    using C = ACfg<T, DKB>;
    constexpr int MP = 16 * MT;
    if (p.kind == FAVOR_SOFTMAX) (void)vitk_zero_async(p.esum, 4 * (size_t)p.nesum, 4 * (size_t)p.nesum, 1, st);
    const size_t lds_q = (size_t)(32 * DKB * (MP + 8) + 4 * 16 * C::ASTR) * sizeof(T) +
                         (size_t)(32 * DKB * (MP + 4) + MP + 4 * 64) * 4;
    const size_t lds_k = (size_t)(32 * DKB * (MP + 8)) * sizeof(T);
    (void)vitk_lds_optin((const void*)favor_bwd_q_kernel<T, DKB, MT>, lds_q);
    (void)vitk_lds_optin((const void*)favor_bwd_kv_kernel<T, DKB, MT>, lds_k);
    hipLaunchKernelGGL((favor_bwd_q_kernel<T, DKB, MT>), dim3(p.H, p.B), dim3(256), lds_q, st, p);
    hipLaunchKernelGGL((favor_bwd_kv_kernel<T, DKB, MT>), dim3((p.L + 63) / 64, p.H, p.B), dim3(256), lds_k, st, p);
    if (p.kind == FAVOR_SOFTMAX) hipLaunchKernelGGL(favor_fix_kernel<T>, dim3(1), dim3(64), 0, st, p);
}

template <typename T, int DKB> void dispatch_mt(const FavorParams& p, bool bwd, hipStream_t st) {
    const int mt = pick_mt(p.m);
#define VITK_FV(M)                                  \
    do {                                            \
        if (bwd) launch_bwd<T, DKB, M>(p, st);      \
        else launch_fwd<T, DKB, M>(p, st);          \
    } while (0)
    VITK_FV(16);               // mt == 1 and 4 never get here (favor_fast.hip)
#undef VITK_FV
}

void dispatch(const FavorParams& p, int dtype, bool bwd, hipStream_t st) {
    if (favor_fast_launch(p, dtype, bwd, st)) return;          // up to 64 features: favor_fast.hip
    const int dkb = (p.dh + 31) / 32;
    if (dtype == VITK_BF16) {
        if (dkb == 1) dispatch_mt<__bf16, 1>(p, bwd, st);
        else dispatch_mt<__bf16, 2>(p, bwd, st);
    } else {
        if (dkb == 1) dispatch_mt<float, 1>(p, bwd, st);
        else dispatch_mt<float, 2>(p, bwd, st);
    }
}

}  // namespace

// Workspace layout (bytes, all 256-byte aligned), esz = element size of T, mp = 16 * MT, dhp = 32 * DKB:
//   kmax (8) | ksum fp32 [B*H*mp] | dksum fp32 [B*H*mp] | KVT T [B*H*dh*mp] | KV T [B*H*mp*dhp]
//   | dKVT T | dKV T | KV fp32 [B*H*mp*dhp] | dKV fp32 | esum fp32 [B*H]
// The forward fills kmax, ksum, KVT, KV; the backward reads them, so the SAME workspace must be
// passed to both calls of one step (it is the saved state of the op).
extern "C" long vitk_favor_workspace_bytes(int dtype, int B, int H, int dh, int m) {
    const long esz = dtype == VITK_BF16 ? 2 : 4;
    const long mp = 16 * pick_mt(m), dhp = 32 * ((dh + 31) / 32);
    const long bh = (long)B * H;
    auto al = [](long x) { return (x + 255) / 256 * 256; };
    return 256 + 2 * al(bh * mp * 4) + 2 * al(bh * dh * mp * esz) + 2 * al(bh * mp * dhp * esz) + 2 * al(bh * mp * dhp * 4) +
           al(bh * 4);
}

static int favor_setup(FavorParams& p, int dtype, int kind, int B, int H, int L, int dh, int m, float ratio, float eps,
                       void* workspace) {
    const long esz = dtype == VITK_BF16 ? 2 : 4;
    const long mp = 16 * pick_mt(m), dhp = 32 * ((dh + 31) / 32);
    const long bh = (long)B * H;
    auto al = [](long x) { return (x + 255) / 256 * 256; };
    char* ws = reinterpret_cast<char*>(workspace);
    p.kmax = reinterpret_cast<unsigned long long*>(ws);
    p.kmax_slots = 1;
    ws += 256;
    p.ksum = reinterpret_cast<float*>(ws); ws += al(bh * mp * 4);
    p.dksum = reinterpret_cast<float*>(ws); ws += al(bh * mp * 4);
    p.kvt = ws; ws += al(bh * dh * mp * esz);
    p.kv = ws; ws += al(bh * mp * dhp * esz);
    p.dkvt_t = ws; ws += al(bh * dh * mp * esz);
    p.dkv_t = ws; ws += al(bh * mp * dhp * esz);
    p.kv32 = reinterpret_cast<float*>(ws); ws += al(bh * mp * dhp * 4);
    p.dkv32 = reinterpret_cast<float*>(ws); ws += al(bh * mp * dhp * 4);
    p.esum = reinterpret_cast<float*>(ws);
    p.nesum = (int)bh;
    p.dkvt = nullptr;
    p.B = B; p.H = H; p.L = L; p.dh = dh; p.m = m; p.mp = (int)mp; p.dhp = (int)dhp;
    p.kind = kind; p.ratio = ratio; p.eps = eps;
    p.diag_c = 0.5f / sqrtf((float)dh);
    return 0;
}

#define FAVOR_COMMON_CHECKS(name)                                                                                  \
    VITK_CHECK_ARG(dtype == VITK_F32 || dtype == VITK_BF16, name ": bad dtype");                                   \
    VITK_CHECK_ARG(kind == FAVOR_SOFTMAX || kind == FAVOR_RELU, name ": kernel must be 0 (softmax) or 1 (relu)");  \
    VITK_CHECK_ARG(B > 0 && H > 0 && L > 0, name ": empty problem");                                               \
    VITK_CHECK_ARG(dh > 0 && dh <= 64 && (dh * (dtype == VITK_BF16 ? 2 : 4)) % 16 == 0, name ": head dim %d unsupported", dh); \
    VITK_CHECK_ARG(m > 0 && m <= 256, name ": 1..256 random features supported, got %d", m);                       \
    VITK_CHECK_ARG((long)B * L * H * m < 0xffffffffL, name ": feature tensor too large for the argmax index");     \
    VITK_CHECK_ARG(workspace && f16al(workspace), name ": workspace must be 16-byte aligned");                     \
    VITK_CHECK_ARG(H <= 65535 && B <= 65535, name ": grid too large")

// out = FAVOR+ attention(q, k, v; w).  kind 0 = softmax kernel, 1 = relu kernel; ratio / eps as the
// reference computes them (m^-1/4 or m^-1/2; 1e-6 or 1e-3).  w: [m][dh] in T, contiguous.
static int favor_fwd_impl(int dtype, int kind, int B, int H, int L, int dh, int m, const void* q, const void* k,
                          const void* v, long ld, const void* w, float ratio, float eps, const float* pq, const float* pk,
                          void* out, long ldo, void* workspace, hipStream_t stream) {
    FAVOR_COMMON_CHECKS("vitk_favor_fwd");
    VITK_CHECK_ARG((!pq || f16al(pq)) && (!pk || f16al(pk)), "vitk_favor_fwd: position tables must be 16-byte aligned");
    const long esz = dtype == VITK_BF16 ? 2 : 4;
    VITK_CHECK_ARG(q && k && v && w && out, "vitk_favor_fwd: null pointer");
    VITK_CHECK_ARG(f16al(q) && f16al(k) && f16al(v) && f16al(out) && f16al(w) && (ld * esz) % 16 == 0 &&
                       (ldo * esz) % 16 == 0 && ld >= (long)H * dh && ldo >= (long)H * dh,
                   "vitk_favor_fwd: rows must be 16-byte aligned and at least H*dh wide");
    FavorParams p = {};
    favor_setup(p, dtype, kind, B, H, L, dh, m, ratio, eps, workspace);
    p.q = q; p.k = k; p.v = v; p.ld = ld; p.w = w; p.out = out; p.ldo = ldo;
    p.pq = pq; p.pk = pk;
    dispatch(p, dtype, false, stream);
    VITK_CHECK_LAUNCH("vitk_favor_fwd");
    return VITK_OK;
}

extern "C" int vitk_favor_fwd(int dtype, int kind, int B, int H, int L, int dh, int m, const void* q, const void* k,
                              const void* v, long ld, const void* w, float ratio, float eps, void* out, long ldo,
                              void* workspace, hipStream_t stream) {
    return favor_fwd_impl(dtype, kind, B, H, L, dh, m, q, k, v, ld, w, ratio, eps, nullptr, nullptr, out, ldo, workspace,
                          stream);
}

// FAVOR+ attention with position features added to both feature maps (q' + pq, k' + pk): pq / pk are fp32
// [L][H][mp] tables made by vitk_favor_pos_fwd, mp = vitk_favor_feature_pitch(m).
extern "C" int vitk_favor_pos_attn_fwd(int dtype, int kind, int B, int H, int L, int dh, int m, const void* q,
                                       const void* k, const void* v, long ld, const void* w, float ratio, float eps,
                                       const float* pq, const float* pk, void* out, long ldo, void* workspace,
                                       hipStream_t stream) {
    VITK_CHECK_ARG(pq && pk, "vitk_favor_pos_attn_fwd: null position table");
    return favor_fwd_impl(dtype, kind, B, H, L, dh, m, q, k, v, ld, w, ratio, eps, pq, pk, out, ldo, workspace, stream);
}

static int favor_bwd_impl(int dtype, int kind, int B, int H, int L, int dh, int m, const void* q, const void* k,
                          const void* v, long ld, const void* w, float ratio, float eps, const float* pq, const float* pk,
                          const void* out, const void* dout, long ldo, void* dq, void* dk, void* dv, long ld_d, float* dnq,
                          float* dnk, void* workspace, hipStream_t stream) {
    FAVOR_COMMON_CHECKS("vitk_favor_bwd");
    VITK_CHECK_ARG((!pq || f16al(pq)) && (!pk || f16al(pk)) && (!dnq || f16al(dnq)) && (!dnk || f16al(dnk)),
                   "vitk_favor_bwd: position tables / gradients must be 16-byte aligned");
    const long esz = dtype == VITK_BF16 ? 2 : 4;
    VITK_CHECK_ARG(q && k && v && w && out && dout && dq && dk && dv, "vitk_favor_bwd: null pointer");
    VITK_CHECK_ARG(f16al(q) && f16al(k) && f16al(v) && f16al(out) && f16al(dout) && f16al(w) && f16al(dq) && f16al(dk) &&
                       f16al(dv) && (ld * esz) % 16 == 0 && (ldo * esz) % 16 == 0 && (ld_d * esz) % 16 == 0 &&
                       ld >= (long)H * dh && ldo >= (long)H * dh && ld_d >= (long)H * dh,
                   "vitk_favor_bwd: rows must be 16-byte aligned and at least H*dh wide");
    FavorParams p = {};
    favor_setup(p, dtype, kind, B, H, L, dh, m, ratio, eps, workspace);
    p.q = q; p.k = k; p.v = v; p.ld = ld; p.w = w;
    p.out = const_cast<void*>(out); p.dout = dout; p.ldo = ldo;
    p.dq = dq; p.dk = dk; p.dv = dv; p.ld_d = ld_d;
    p.pq = pq; p.pk = pk; p.dnq = dnq; p.dnk = dnk;
    dispatch(p, dtype, true, stream);
    VITK_CHECK_LAUNCH("vitk_favor_bwd");
    return VITK_OK;
}

extern "C" int vitk_favor_bwd(int dtype, int kind, int B, int H, int L, int dh, int m, const void* q, const void* k,
                              const void* v, long ld, const void* w, float ratio, float eps, const void* out,
                              const void* dout, long ldo, void* dq, void* dk, void* dv, long ld_d, void* workspace,
                              hipStream_t stream) {
    return favor_bwd_impl(dtype, kind, B, H, L, dh, m, q, k, v, ld, w, ratio, eps, nullptr, nullptr, out, dout, ldo, dq,
                          dk, dv, ld_d, nullptr, nullptr, workspace, stream);
}

// Backward of vitk_favor_pos_attn_fwd.  dnq / dnk (fp32 [B][L][H][mp], fully written) receive the gradient of the
// summed features per batch element; vitk_favor_pos_bwd reduces them over b and goes back through the tables.
extern "C" int vitk_favor_pos_attn_bwd(int dtype, int kind, int B, int H, int L, int dh, int m, const void* q,
                                       const void* k, const void* v, long ld, const void* w, float ratio, float eps,
                                       const float* pq, const float* pk, const void* out, const void* dout, long ldo,
                                       void* dq, void* dk, void* dv, long ld_d, float* dnq, float* dnk, void* workspace,
                                       hipStream_t stream) {
    VITK_CHECK_ARG(pq && pk && dnq && dnk, "vitk_favor_pos_attn_bwd: null position table / gradient");
    return favor_bwd_impl(dtype, kind, B, H, L, dh, m, q, k, v, ld, w, ratio, eps, pq, pk, out, dout, ldo, dq, dk, dv,
                          ld_d, dnq, dnk, workspace, stream);
}

// Row pitch (floats) of the position-feature tables and of dnq / dnk for m random features.
extern "C" int vitk_favor_feature_pitch(int m) { return 16 * pick_mt(m); }
extern "C" long vitk_favor_pos_state_bytes(int H, int L) { return 64 + 4 * (long)H * L; }

#define FAVOR_POS_CHECKS(name)                                                                                     \
    VITK_CHECK_ARG(dtype == VITK_F32 || dtype == VITK_BF16, name ": bad dtype");                                   \
    VITK_CHECK_ARG(kind == FAVOR_SOFTMAX || kind == FAVOR_RELU, name ": kernel must be 0 (softmax) or 1 (relu)");  \
    VITK_CHECK_ARG(H > 0 && L > 0 && H <= 65535 && L <= 65535, name ": empty or oversized problem");               \
    VITK_CHECK_ARG(dh > 0 && dh <= 64, name ": head dim %d unsupported", dh);                                      \
    VITK_CHECK_ARG(m > 0 && m <= 256, name ": 1..256 random features supported, got %d", m);                       \
    VITK_CHECK_ARG(x && w && state && f16al(state) && ld >= (long)H * dh, name ": bad x / w / state");             \
    VITK_CHECK_ARG(dh % 4 == 0 && f16al(x) && f16al(w) && (ld * (dtype == VITK_BF16 ? 2 : 4)) % 16 == 0 &&         \
                       (dh * (dtype == VITK_BF16 ? 2 : 4)) % 8 == 0,                                               \
                   name ": x / w rows must be 16-byte aligned, head dim a multiple of 4")

static void favor_pos_params(FavorParams& p, int kind, int H, int L, int dh, int m, const void* x, long ld, const void* w,
                             float ratio, float eps, void* state) {
    p.q = x; p.ld = ld; p.w = w;
    p.kmax = reinterpret_cast<unsigned long long*>(state);
    p.kmax_slots = 1;
    p.esum = reinterpret_cast<float*>(reinterpret_cast<char*>(state) + 64);
    p.nesum = L * H;
    p.B = 1; p.H = H; p.L = L; p.dh = dh; p.m = m; p.mp = 16 * pick_mt(m);
    p.kind = kind; p.ratio = ratio; p.eps = eps;
    p.diag_c = 0.5f / sqrtf((float)dh);
}

// Position-feature table: feat[l][h][.] = feature map (query form: row max; key form: max over the whole
// [L][H][m] tensor) of x[l][h*dh ..], x = pos_query_proj / pos_key_proj of pos_embed ([L, >= H*dh] in T).
// Replaces softmax_kernel_transformation / relu_kernel_transformation as called at
// disentangled_transformer.py:623-641.  state: vitk_favor_pos_state_bytes(H, L) bytes, 16-byte aligned: the key
// max / argmax for vitk_favor_pos_bwd, and room for its per-(l, h) partial sums.
extern "C" int vitk_favor_pos_fwd(int dtype, int kind, int is_query, int H, int L, int dh, int m, const void* x, long ld,
                                  const void* w, float ratio, float eps, float* feat_out, void* state,
                                  hipStream_t stream) {
    FAVOR_POS_CHECKS("vitk_favor_pos_fwd");
    VITK_CHECK_ARG(feat_out && f16al(feat_out), "vitk_favor_pos_fwd: feat must be 16-byte aligned");
    FavorParams p = {};
    favor_pos_params(p, kind, H, L, dh, m, x, ld, w, ratio, eps, state);
    const dim3 grid(H, L);
    if (dtype == VITK_BF16) {
        if (kind == FAVOR_SOFTMAX && !is_query) {
            (void)vitk_zero_async(state, 16, 16, 1, stream);
            hipLaunchKernelGGL(favor_pos_fwd_kernel<__bf16>, grid, dim3(64), 0, stream, p, is_query, 0, feat_out);
        }
        hipLaunchKernelGGL(favor_pos_fwd_kernel<__bf16>, grid, dim3(64), 0, stream, p, is_query, 1, feat_out);
    } else {
        if (kind == FAVOR_SOFTMAX && !is_query) {
            (void)vitk_zero_async(state, 16, 16, 1, stream);
            hipLaunchKernelGGL(favor_pos_fwd_kernel<float>, grid, dim3(64), 0, stream, p, is_query, 0, feat_out);
        }
        hipLaunchKernelGGL(favor_pos_fwd_kernel<float>, grid, dim3(64), 0, stream, p, is_query, 1, feat_out);
    }
    VITK_CHECK_LAUNCH("vitk_favor_pos_fwd");
    return VITK_OK;
}

// dx[l][h*dh ..] (T, [L, ld_d]) from dfeat (fp32 [nsum][L][H][mp], summed over its first dim here).
extern "C" int vitk_favor_pos_bwd(int dtype, int kind, int is_query, int nsum, int H, int L, int dh, int m, const void* x,
                                  long ld, const void* w, float ratio, float eps, const float* dfeat, void* dx, long ld_d,
                                  void* state, hipStream_t stream) {
    FAVOR_POS_CHECKS("vitk_favor_pos_bwd");
    VITK_CHECK_ARG(nsum > 0 && dfeat && dx && ld_d >= (long)H * dh, "vitk_favor_pos_bwd: bad dfeat / dx");
    FavorParams p = {};
    favor_pos_params(p, kind, H, L, dh, m, x, ld, w, ratio, eps, state);
    p.dk = dx; p.ld_d = ld_d;
    const bool fix = kind == FAVOR_SOFTMAX && !is_query;      // every (l, h) stores its partial sum: nothing to zero
    const dim3 grid(H, L);
    if (dtype == VITK_BF16) {
        hipLaunchKernelGGL(favor_pos_bwd_kernel<__bf16>, grid, dim3(256), 0, stream, p, is_query, dfeat, nsum);
        if (fix) hipLaunchKernelGGL(favor_fix_kernel<__bf16>, dim3(1), dim3(64), 0, stream, p);
    } else {
        hipLaunchKernelGGL(favor_pos_bwd_kernel<float>, grid, dim3(256), 0, stream, p, is_query, dfeat, nsum);
        if (fix) hipLaunchKernelGGL(favor_fix_kernel<float>, dim3(1), dim3(64), 0, stream, p);
    }
    VITK_CHECK_LAUNCH("vitk_favor_pos_bwd");
    return VITK_OK;
}
