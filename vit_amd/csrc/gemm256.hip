// bf16 MFMA GEMM, 256 x 256 output tile, direct-to-LDS staging, persistent workgroups with a
// two-group ping-pong schedule: the fast path behind vitk_gemm for the large linears of the
// encoder (qkv / proj / fc1 / fc2 forward, dX, dW).
//
// Geometry (one workgroup = 512 threads = 8 waves as 2 (M) x 4 (N), one workgroup per CU):
//   * 64 K-elements per K tile; A and B tiles (256 x 64 bf16 = 32 KiB each) are written straight
//     into LDS by global_load_lds_dwordx4 (1 KiB per wave instruction, 4 per wave per operand),
//     two LDS stages (128 KiB): K tile t+1 is in flight while tile t is consumed.
//   * a wave owns a 128 x 64 output block: 8 x 4 MFMA 16x16x32 tiles, 128 accumulator registers.
//     A K tile is consumed in four phases, one 64 x 32 quadrant of the block (16 MFMAs) each; a
//     phase is [LDS fragment reads + LDS-DMA issue] barrier [16 MFMAs] barrier.  The two wave
//     groups (M halves; one wave of each per SIMD) run ONE barrier apart, so on every SIMD one
//     wave issues MFMAs while the other reads LDS / issues loads.
//   * K-contiguous operands (rows contiguous along K: activations, nn.Linear weights) use an
//     image of 128-byte rows whose 16-byte chunks are XOR-swizzled so that every ds_read_b128 lane
//     group covers all 16 slots of the 256-byte bank row.  LDS-DMA writes are lane-linear, so the
//     swizzle is applied to each lane's SOURCE address.
//   * K-major operands (dX = dY W reads W as [K][N]; dW = dY^T X reads both as [K][.]) use an
//     image of 512-byte k-rows with chunks XOR-swizzled by 2 * ((k & 3) | ((k >> 3) & 1) << 2);
//     fragments come from ds_read_b64_tr_b16 (hardware transpose).  No transposed copy of any
//     operand exists in HBM.
//   * operands are swapped in the MFMA (D = Bfrag x Afrag) so a lane owns 4 consecutive output
//     columns per tile, and the B rows are fed to the MFMA column slots in a permuted order
//     (slot 4u + e of tile j <- column 32 (j >> 1) + 8 u + 4 (j & 1) + e of the wave's 64): the
//     accumulators of tiles (0,1) / (2,3) of one lane are then 8 CONSECUTIVE output columns, and
//     the fused epilogue stores straight from registers, 16 bytes per lane, 64 contiguous bytes
//     per row and instruction -- no LDS round trip.
//   * a workgroup walks work items (K split, tile) blockIdx.x, + gridDim.x, ...; the K-tile
//     pipeline runs across item boundaries (the next item's first tile is prefetched under the
//     last K tile and the epilogue).  XCD-aware grouped tile order; split-K (fp32 atomics shaped
//     through LDS) for the dW shapes.
// Requirements checked by the host: K % 64 == 0, 16-byte aligned rows, row counts of K-major
// operands multiples of 8.  Rows past M / N are clamped to the last valid row on load (their
// products land in output rows / columns that are never stored).
#include "gemm_common.h"

namespace vitk {

constexpr int T256 = 256;
constexpr int OPER_BYTES = 256 * 64 * 2;       // one operand tile
constexpr int STAGE256 = 2 * OPER_BYTES;       // A + B
constexpr int STAGES_BYTES = 2 * STAGE256;     // 128 KiB of staging
constexpr int EPI_BYTES = 8 * 4096;            // + 4 KiB per wave (split-K atomic shaping) = 160 KiB

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

__device__ __forceinline__ int km_swz(int k) { return 2 * ((k & 3) | (((k >> 3) & 1) << 2)); }

// Chunk swizzle of row r of a K-contiguous image.  A rows are read 16 consecutive rows per
// fragment; B rows in the permuted order above, where the slot index of row r inside its
// fragment is 4 ((r >> 3) & 3) + (r & 3): both read patterns see swizzle (slot >> 1) & 7.
template <bool PERM> __device__ __forceinline__ int kc_swz(int r) {
    return PERM ? (2 * ((r >> 3) & 3) + ((r >> 1) & 1)) : ((r >> 1) & 7);
}

// Staging pieces.  An operand tile (32 KiB) is staged as two 16 KiB pieces, each 16 LDS-DMA
// instructions of 1 KiB (two per wave):
//   K-contiguous image: piece q = the rows the waves read in their quadrant-row half q
//       (A: rows 128 wm + 64 q + [0,64); B: rows 64 wn + 32 q + [0,32)), in groups of 8 rows;
//   K-major image: piece q = k-rows [32 q, 32 q + 32), in groups of 2 k-rows.
// off[2 q + jj] is the lane's source offset (elements, without the K advance) of its jj-th group.
template <bool KM, bool IS_B> __device__ __forceinline__ int piece_row(int q, int gi) {
    if (KM) return 32 * q + 2 * gi;
    return IS_B ? (gi >> 2) * 64 + q * 32 + (gi & 3) * 8 : (gi >> 3) * 128 + q * 64 + (gi & 7) * 8;
}

template <bool KM, bool IS_B>
__device__ __forceinline__ void stage_offsets(long (&off)[4], long ld, int r0, int R, int wave, int lane) {
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
        for (int jj = 0; jj < 2; ++jj) {
            const int row = piece_row<KM, IS_B>(q, 2 * wave + jj);
            if (!KM) {
                const int r = row + (lane >> 3);
                const int c = (lane & 7) ^ kc_swz<IS_B>(r);
                off[2 * q + jj] = (long)min(r0 + r, R - 1) * ld + 8 * c;
            } else {
                const int k = row + (lane >> 5);
                const int cc = (lane & 31) ^ km_swz(k);
                off[2 * q + jj] = (long)k * ld + min(r0 + 8 * cc, R - 8);
            }
        }
}

template <bool KM, bool IS_B>
__device__ __forceinline__ void stage_piece(const __bf16* __restrict__ X, const long (&off)[4], long kadv, char* img,
                                            int q, int wave) {
#pragma unroll
    for (int jj = 0; jj < 2; ++jj) {
        const __bf16* src = X + off[2 * q + jj] + kadv;
        char* dst = img + piece_row<KM, IS_B>(q, 2 * wave + jj) * (KM ? 512 : 128);
        __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)dst, 16, 0, 0);
    }
}

__device__ __forceinline__ void tr_pair(Frag<__bf16>& f, const char* a0) {
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(uintptr_t)a0);
    const s16x4 hi =
        __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(uintptr_t)(a0 + 4 * 512));
    const s16x8 r = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    f.v = __builtin_bit_cast(bf16x8, r);
}

// A fragment: image rows [row_base, +16) (row_base a multiple of 16), MFMA K block kb.
template <bool KM> __device__ __forceinline__ void frag_a(Frag<__bf16>& f, const char* lds, int row_base, int kb, int lane) {
    const int g = lane >> 4, i = lane & 15;
    if (!KM) {
        const int c = (4 * kb + g) ^ (i >> 1);
        f.v = *reinterpret_cast<const bf16x8*>(lds + (row_base + i) * 128 + (c << 4));
    } else {
        const int q = i >> 2, pp = i & 3;
        const int k = 32 * kb + 8 * g + q;
        const int cc = ((row_base >> 3) + (pp >> 1)) ^ km_swz(k);
        tr_pair(f, lds + k * 512 + (cc << 4) + 8 * (pp & 1));
    }
}

// B fragment of the wave's tile j (0..3): MFMA slot 4u + e <- image row nb + 32 (j >> 1) + 8 u + 4 (j & 1) + e.
template <bool KM> __device__ __forceinline__ void frag_b(Frag<__bf16>& f, const char* lds, int nb, int j, int kb, int lane) {
    const int g = lane >> 4, i = lane & 15;
    if (!KM) {
        const int row = nb + 32 * (j >> 1) + 8 * (i >> 2) + 4 * (j & 1) + (i & 3);
        const int c = (4 * kb + g) ^ (i >> 1);
        f.v = *reinterpret_cast<const bf16x8*>(lds + row * 128 + (c << 4));
    } else {
        const int q = i >> 2, u = i & 3;
        const int k = 32 * kb + 8 * g + q;
        const int cc = ((nb >> 3) + 4 * (j >> 1) + u) ^ km_swz(k);
        tr_pair(f, lds + k * 512 + (cc << 4) + 8 * (j & 1));
    }
}

// ---------------------------------------------------------------------------------- epilogue
// Epilogue loads are hidden from the compiler (inline asm) and waited for by hand with counted
// vmcnt: beside LDS-DMA in flight, hipcc waits vmcnt(0) for every ordinary load, and vmcnt retires
// in issue order, so each such wait would also sit out the write acknowledgements of all stores
// issued before it (~1-2 us each under load; it made the epilogue ~15k cycles per tile).
__device__ __forceinline__ bf16x8 asm_load_bf16x8(const __bf16* ptr) {
    bf16x8 r;
    asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(r) : "v"(ptr) : "memory");
    return r;
}
__device__ __forceinline__ f32x4 asm_load_f32x4(const float* ptr) {
    f32x4 r;
    asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(r) : "v"(ptr) : "memory");
    return r;
}
#define VITK_WAIT_VM(N)                                         \
    do {                                                        \
        asm volatile("s_waitcnt vmcnt(" #N ")" ::: "memory");   \
        __builtin_amdgcn_sched_barrier(0);                      \
    } while (0)

// Fast form of the fused epilogue: one output row x 8 consecutive columns from registers (16-byte
// accesses legal, all columns < N, no row remap / broadcast residual, at most one side operand:
// the residual, or the saved activation of dgelu / drelu).
template <typename TC, bool SIDE>
__device__ __forceinline__ void epi8_regs(const GemmParams& p, long row, int n8, const f32x4 lo, const f32x4 hi,
                                          const float (&bias8)[8], const bf16x8 side) {
    float v[8] = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = v[e] * p.alpha + bias8[e];
    if (p.act == ACT_GELU) {
        if (!SIDE && p.aux_out) {
            bf16x8 o;
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = (__bf16)v[e];
            *reinterpret_cast<bf16x8*>(reinterpret_cast<__bf16*>(p.aux_out) + row * p.ld_aux + n8) = o;
        }
#pragma unroll
        for (int e = 0; e < 8; e += 2) {
            const f32x2 y = gelu_poly2(f32x2{v[e], v[e + 1]});
            v[e] = y[0];
            v[e + 1] = y[1];
        }
    } else if (p.act == ACT_RELU) {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = fmaxf(v[e], 0.f);
    } else if (SIDE && p.act == ACT_DGELU) {
#pragma unroll
        for (int e = 0; e < 8; e += 2) {
            const f32x2 d = gelu_grad_poly2(f32x2{(float)side[e], (float)side[e + 1]});
            v[e] *= d[0];
            v[e + 1] *= d[1];
        }
    } else if (SIDE && p.act == ACT_DRELU) {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = (float)side[e] > 0.f ? v[e] : 0.f;
    }
    if (SIDE && p.residual) {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] += (float)side[e];
    }
    TC* dst = reinterpret_cast<TC*>(p.C) + row * p.ldc + n8;
    if (sizeof(TC) == 4) {
        *reinterpret_cast<f32x4*>(dst) = f32x4{v[0], v[1], v[2], v[3]};
        *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(dst) + 4) = f32x4{v[4], v[5], v[6], v[7]};
    } else {
        bf16x8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = (__bf16)v[e];
        *reinterpret_cast<bf16x8*>(dst) = o;
    }
}

#define VITK_PICK_ROW(IT)            \
    case IT:                         \
        a0 = acc[IT][0];             \
        a1 = acc[IT][1];             \
        a2 = acc[IT][2];             \
        a3 = acc[IT][3];             \
        asm volatile("; acc row tile " #IT : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3)); \
        break;
#define VITK_PICK_ROWS()                                                                      \
    f32x4 a0, a1, a2, a3;                                                                     \
    switch (it) {                                                                             \
        VITK_PICK_ROW(0) VITK_PICK_ROW(1) VITK_PICK_ROW(2) VITK_PICK_ROW(3) VITK_PICK_ROW(4)  \
        VITK_PICK_ROW(5) VITK_PICK_ROW(6)                                                     \
        default:                                                                              \
            a0 = acc[7][0];                                                                   \
            a1 = acc[7][1];                                                                   \
            a2 = acc[7][2];                                                                   \
            a3 = acc[7][3];                                                                   \
            asm volatile("; acc row tile 7" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));        \
            break;                                                                            \
    }

// The wave's 128 x 64 block.  ONE rolled loop over its eight 16-row tiles (only the register
// selection is per tile): unrolled, the fused epilogue is ~0.5 MiB of straight-line code per kernel.
// ALLOW_FAST = false: only the split-K and the rolled generic forms (no hand-counted asynchronous loads).
template <typename TC, bool ALLOW_FAST = true>
__device__ __forceinline__ void epilogue256(const GemmParams& p, const f32x4 (&acc)[8][4], float* wl, int mrow0, int n0w,
                                            int split, int lane) {
    const int g = lane >> 4, lm = lane & 15;
    const bool atomic = p.splitk > 1;
    if (atomic) {
        // fp32 atomics shaped through the wave's LDS scratch (16 x 64 fp32, slot s of row r at
        // s ^ r) into one contiguous 256-byte row segment per wave instruction (lane = column):
        // the shape the memory-side atomic units run at full rate.
        const int col = n0w + lane;
        const float bcol = (p.bias != nullptr && split == 0 && col < p.N) ? p.bias[col] : 0.f;
        float* __restrict__ C = reinterpret_cast<float*>(p.C);
#pragma nounroll
        for (int it = 0; it < 8; ++it) {
            VITK_PICK_ROWS()
            // tile j holds columns 32 (j >> 1) + 8 g + 4 (j & 1) + e -> 16-byte slot 8 (j >> 1) + 2 g + (j & 1)
            *reinterpret_cast<f32x4*>(wl + lm * 64 + (((2 * g) ^ lm) << 2)) = a0;
            *reinterpret_cast<f32x4*>(wl + lm * 64 + (((2 * g + 1) ^ lm) << 2)) = a1;
            *reinterpret_cast<f32x4*>(wl + lm * 64 + (((8 + 2 * g) ^ lm) << 2)) = a2;
            *reinterpret_cast<f32x4*>(wl + lm * 64 + (((9 + 2 * g) ^ lm) << 2)) = a3;
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            const int mbase = mrow0 + it * 16;
#pragma unroll 8
            for (int row = 0; row < 16; ++row) {
                const float v = wl[row * 64 + ((((lane >> 2) ^ row) << 2) | (lane & 3))];
                if (mbase + row < p.M && col < p.N) atomicAdd(C + (long)(mbase + row) * p.ldc + col, v * p.alpha + bcol);
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
        return;
    }
    const int ncol = n0w + 8 * g;   // this lane's columns: ncol + 32 jp + 0..7, jp = 0, 1
    if (mrow0 >= p.M) return;
    const bool has_side = p.residual != nullptr || p.act == ACT_DGELU || p.act == ACT_DRELU;
    const bool odd = (p.residual != nullptr && (p.act == ACT_DGELU || p.act == ACT_DRELU)) ||   // two side operands
                     (p.residual != nullptr && p.act == ACT_GELU && p.aux_out != nullptr) || p.accumulate;
    // fast form: the wave's whole 128 x 64 block is inside the matrix, 16-byte accesses are legal
    const bool fast = ALLOW_FAST && p.c_vec8_ok && n0w + 64 <= p.N && mrow0 + 128 <= p.M && p.grp_rows == 0 && p.res_period == 0 && !odd;
    if (ALLOW_FAST && fast) {
        // Straight-line code with hand-counted waits.  Issue order per wave:
        //   [4 bias loads] S(0) | S(1) wait(2) stores(0) | S(2) wait(ns + 2) stores(1) | ... | wait(ns) stores(7)
        // S(i) = the two 16-byte side loads of row tile i, ns = stores per row tile: every wait leaves
        // exactly the operations YOUNGER than the loads it needs in flight, never a store ahead of them.
        constexpr int NS = sizeof(TC) == 4 ? 4 : 2;
        f32x4 b0 = {0.f, 0.f, 0.f, 0.f}, b1 = b0, b2 = b0, b3 = b0;
        if (p.bias != nullptr) {
            b0 = asm_load_f32x4(p.bias + ncol);
            b1 = asm_load_f32x4(p.bias + ncol + 4);
            b2 = asm_load_f32x4(p.bias + ncol + 32);
            b3 = asm_load_f32x4(p.bias + ncol + 36);
        }
        float bias8[2][8];
        if (has_side) {
            const __bf16* side = reinterpret_cast<const __bf16*>(p.residual ? p.residual : p.aux_in);
            const long ld_side = p.residual ? p.ldr : p.ld_aux;
            const __bf16* s = side + (long)(mrow0 + lm) * ld_side + ncol;
            bf16x8 sd[2][2];
            sd[0][0] = asm_load_bf16x8(s);
            sd[0][1] = asm_load_bf16x8(s + 32);
#pragma unroll
            for (int it = 0; it < 8; ++it) {
                if (it < 7) {
                    s += 16 * ld_side;
                    sd[(it + 1) & 1][0] = asm_load_bf16x8(s);
                    sd[(it + 1) & 1][1] = asm_load_bf16x8(s + 32);
                }
                if (it == 0) {
                    VITK_WAIT_VM(2);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {   // the bias loads are older than S(0): landed too
                        bias8[0][e] = b0[e];
                        bias8[0][4 + e] = b1[e];
                        bias8[1][e] = b2[e];
                        bias8[1][4 + e] = b3[e];
                    }
                } else if (it < 7) {
                    if (NS == 2) VITK_WAIT_VM(4);
                    else VITK_WAIT_VM(6);
                } else {
                    if (NS == 2) VITK_WAIT_VM(2);
                    else VITK_WAIT_VM(4);
                }
                const int m = mrow0 + it * 16 + lm;
                epi8_regs<TC, true>(p, m, ncol, acc[it][0], acc[it][1], bias8[0], sd[it & 1][0]);
                epi8_regs<TC, true>(p, m, ncol + 32, acc[it][2], acc[it][3], bias8[1], sd[it & 1][1]);
                __builtin_amdgcn_sched_barrier(0);
            }
            return;
        }
        if (p.bias != nullptr) VITK_WAIT_VM(0);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            bias8[0][e] = b0[e];
            bias8[0][4 + e] = b1[e];
            bias8[1][e] = b2[e];
            bias8[1][4 + e] = b3[e];
        }
#pragma unroll
        for (int it = 0; it < 8; ++it) {
            const int m = mrow0 + it * 16 + lm;
            epi8_regs<TC, false>(p, m, ncol, acc[it][0], acc[it][1], bias8[0], bf16x8{});
            epi8_regs<TC, false>(p, m, ncol + 32, acc[it][2], acc[it][3], bias8[1], bf16x8{});
            __builtin_amdgcn_sched_barrier(0);
        }
        return;
    }
    float bias8[2][8];
#pragma unroll
    for (int jp = 0; jp < 2; ++jp)
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int n = ncol + 32 * jp + e;
            bias8[jp][e] = (p.bias != nullptr && n < p.N) ? p.bias[n] : 0.f;
        }
    // generic form (ragged N, unaligned pointers, patch-embed row remap / broadcast residual)
#pragma nounroll
    for (int it = 0; it < 8; ++it) {
        VITK_PICK_ROWS()
        const int m = mrow0 + it * 16 + lm;
#pragma nounroll
        for (int jp = 0; jp < 2; ++jp) {
            const f32x4 lo = jp ? a2 : a0, hi = jp ? a3 : a1;
            const float v[8] = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
            gemm_epilogue8<__bf16, TC>(p, split, m, ncol + 32 * jp, v, bias8[jp]);
        }
    }
}
#undef VITK_PICK_ROWS
#undef VITK_PICK_ROW

// ------------------------------------------------------------------------------------ kernel
// Work item = (K split, output tile).  The XCD remap runs over the whole (split, tile) list, split
// major: the 32 items one XCD has in flight are a compact block of tiles of ONE K range (8 x 4
// tiles read 12 operand panels through that XCD's L2; spread over the splits they read 36, and
// the split-K weight-gradient GEMMs ran into the fabric bandwidth: 3.6 TB/s of L2 misses).
struct Item256 {
    int m0, n0, ks0, ks1, split;
};
__device__ __forceinline__ Item256 item256(const GemmParams& p, int item) {
    const int ntiles = p.tiles_m * p.tiles_n;
    Item256 it;
    item = gemm_xcd_remap(item, ntiles * p.splitk);
    it.split = item / ntiles;
    int tile_m, tile_n;
    gemm_grouped_tile(item - it.split * ntiles, p.tiles_m, p.tiles_n, tile_m, tile_n);
    it.m0 = tile_m * T256;
    it.n0 = tile_n * T256;
    const int ksteps_total = p.K / 64;
    const int steps_per = (ksteps_total + p.splitk - 1) / p.splitk;
    it.ks0 = it.split * steps_per;
    it.ks1 = min(ksteps_total, it.ks0 + steps_per);   // host guarantees ks0 < ks1
    return it;
}

#define VITK_FENCE() __builtin_amdgcn_sched_barrier(0)
// In-kernel cycle stamps (tools/gemm_stamps.py): build with -DVITK_GEMM_STAMPS, run with VITK_GEMM_DEBUG=32.
#ifdef VITK_GEMM_STAMPS
#define VITK_STAMP(I) do { if (stamp) stamps[I] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define VITK_STAMP(I) do { } while (0)
#endif
// A phase: [fragment reads, 2 LDS-DMA, waits] barrier [16 MFMAs of one quadrant] barrier.
// The fragment reads are retired after the first barrier (their latency overlaps the barrier
// wait), so a piece may be restaged two phases after its last read -- except where LGKM_FIRST
// retires them before the barrier, which allows the restage in the very next phase.
#define VITK_QUADRANT(IT0, JT0, FB, SI, LGKM_FIRST)                                              \
    do {                                                                                         \
        VITK_STAMP(SI + 1);                                                                      \
        if (LGKM_FIRST) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                       \
        VITK_STAMP(SI + 2);                                                                      \
        VITK_FENCE();                                                                            \
        __builtin_amdgcn_s_barrier();                                                            \
        if (!(LGKM_FIRST)) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                    \
        VITK_FENCE();                                                                            \
        VITK_STAMP(SI + 3);                                                                      \
        __builtin_amdgcn_s_setprio(1);                                                           \
        _Pragma("unroll") for (int kb = 0; kb < 2; ++kb)                                         \
            _Pragma("unroll") for (int i_ = 0; i_ < 4; ++i_)                                     \
                _Pragma("unroll") for (int jt = 0; jt < 2; ++jt)                                 \
                    acc[IT0 + i_][JT0 + jt] = mma(FB[jt][kb], fa[i_][kb], acc[IT0 + i_][JT0 + jt]); \
        __builtin_amdgcn_s_setprio(0);                                                           \
        VITK_FENCE();                                                                            \
        VITK_STAMP(SI + 4);                                                                      \
        __builtin_amdgcn_s_barrier();                                                            \
        VITK_FENCE();                                                                            \
        VITK_STAMP(SI + 5);                                                                      \
    } while (0)
// Two-phase schedule: one phase = a row half of the wave's block against all four column tiles (32 MFMAs).
#define VITK_HALF(IT0, LGKM_FIRST)                                                               \
    do {                                                                                         \
        if (LGKM_FIRST) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                       \
        VITK_FENCE();                                                                            \
        __builtin_amdgcn_s_barrier();                                                            \
        if (!(LGKM_FIRST)) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                    \
        VITK_FENCE();                                                                            \
        __builtin_amdgcn_s_setprio(1);                                                           \
        _Pragma("unroll") for (int kb = 0; kb < 2; ++kb)                                         \
            _Pragma("unroll") for (int i_ = 0; i_ < 4; ++i_) {                                   \
                acc[IT0 + i_][0] = mma(fb0[0][kb], fa[i_][kb], acc[IT0 + i_][0]);                \
                acc[IT0 + i_][1] = mma(fb0[1][kb], fa[i_][kb], acc[IT0 + i_][1]);                \
                acc[IT0 + i_][2] = mma(fb1[0][kb], fa[i_][kb], acc[IT0 + i_][2]);                \
                acc[IT0 + i_][3] = mma(fb1[1][kb], fa[i_][kb], acc[IT0 + i_][3]);                \
            }                                                                                    \
        __builtin_amdgcn_s_setprio(0);                                                           \
        VITK_FENCE();                                                                            \
        __builtin_amdgcn_s_barrier();                                                            \
        VITK_FENCE();                                                                            \
    } while (0)
#define VITK_BARRIER()                \
    do {                              \
        VITK_FENCE();                 \
        __builtin_amdgcn_s_barrier(); \
        VITK_FENCE();                 \
    } while (0)
#define VITK_VMCNT(N) asm volatile("s_waitcnt vmcnt(" #N ")" ::: "memory")

// The K-tile pipeline.  Compute tile t lives in LDS stage t & 1; its four phases also issue the
// four pieces of the tiles ahead, two LDS-DMA per wave and phase, each piece as soon as its target
// was read for the last time:
//     phase 2, 3 of tile t      -> pieces 0, 1 of tile t + 2   (into stage t & 1, the one in use)
//     phase 0, 1 of tile t + 1  -> pieces 2, 3 of tile t + 2
//   piece order: A K-contiguous: [A q0, B 0, B 1, A q1]      A K-major: [B 0, B 1, A 0, A 1]
// (A q0 / B q0 are read in phase 0 only, B q1 in phase 1, A q1 in phase 2; K-major tiles are read
// whole: A in phases 0 and 2, B in phases 0 and 1.)  Every piece is in flight for 3-6 phases.
// vmcnt is counted (loads, stores and LDS-DMA retire in issue order): the wait of phase p, placed
// before p's first barrier, covers what phase p + 1 reads -- for both wave groups, which run one
// barrier apart.  Epilogue loads / stores of an item only make the counted waits conservative.
template <typename TC, bool A_KM, bool B_KM>
__global__ __launch_bounds__(512, 2) void gemm256_kernel(const GemmParams p) {
    // K-major operands (twice the LDS read instructions per fragment) run the two-phase schedule: half as
    // many barrier hand-offs per K tile (+10 % on the dW GEMMs); K-contiguous NT measures the same either way
    // and keeps the four-phase one.
    constexpr bool TWO_PHASE = A_KM || B_KM;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;
    const int nitems = p.tiles_m * p.tiles_n * p.splitk;
    int item = blockIdx.x;
    if (item >= nitems) return;
    // Stagger: when every CU of an XCD reaches its epilogue at the same time, the 32 of them share that XCD's L2
    // write path (128 KiB per CU take 4.6 k cycles, against 1.9 k for a CU storing alone: tools/micro/store_rate.hip)
    // and each CU's next loads queue behind its own stores.  Workgroups that walk one item FEWER than the busiest
    // ones have a whole tile time of slack: the later quarters of each XCD's workgroups (blockIdx >> 3 >= 8), as far as
    // it has that slack, starts late -- in `groups` phases spread over `pct` % of a tile time -- so that only a fraction of
    // an XCD stores at any time.  Same-box A/B on the layer's NT GEMMs: -2.9 % (dX fc2 -6 %, dX qkv -4 %, fc2 -3 %).
    // That was the GEMMs alone, with 75 %.  In the training step the waiting workgroups hold CUs the second stream's dW
    // GEMMs could use: same-box sweep of the headline step, 0 / 25 / 50 / 75 %: 1898 / 1907 / 1889 / 1872 img/s, NT launch
    // 184 / 179 / 179 / 178 us -- the default is 25 %.
    if (p.stagger > 0) {
        const int rem = nitems % (int)gridDim.x;           // workgroups below `rem` walk one item more
        const int groups = p.stagger >> 24, ticks = p.stagger & 0xffffff;      // phase g of `groups` starts g * ticks late
        const int ph = (int)(blockIdx.x >> 3) * groups / 32;
        if (nitems > (int)gridDim.x && rem != 0 && (int)blockIdx.x >= rem && ph > 0) {
            const long t0 = __builtin_amdgcn_s_memtime();
            while (__builtin_amdgcn_s_memtime() - t0 < (long)ticks * ph) __builtin_amdgcn_s_sleep(32);
        }
    }

    const __bf16* __restrict__ A = reinterpret_cast<const __bf16*>(p.A);
    const __bf16* __restrict__ B = reinterpret_cast<const __bf16*>(p.B);
    const long a_kstep = A_KM ? 64 * p.lda : 64;
    const long b_kstep = B_KM ? 64 * p.ldb : 64;
    float* wl = reinterpret_cast<float*>(smem + STAGES_BYTES + wave * 4096);

    // prefetch cursor: the K tile whose pieces are being issued
    Item256 it = item256(p, item);
    Item256 pf = it;
    int pf_item = item, pf_ks = pf.ks0, pf_buf = 0;
    bool pf_live = true;
    long offa[4], offb[4];
    stage_offsets<A_KM, false>(offa, p.lda, pf.m0, p.M, wave, lane);
    stage_offsets<B_KM, true>(offb, p.ldb, pf.n0, p.N, wave, lane);

    auto issue_slot = [&](int slot) {   // slot 0..3 of the cursor tile (slot is a literal at every call)
        if (!pf_live) return;
        char* img = smem + pf_buf * STAGE256;
        // two-phase schedule and K-major A: [B 0, B 1, A 0, A 1]; four-phase NT: [A q0, B 0, B 1, A q1]
        const bool is_b = (TWO_PHASE || A_KM) ? slot < 2 : (slot == 1 || slot == 2);
        const int q = (TWO_PHASE || A_KM) ? (slot & 1) : (slot == 0 || slot == 1 ? 0 : 1);
        if (is_b) stage_piece<B_KM, true>(B, offb, pf_ks * b_kstep, img + OPER_BYTES, q, wave);
        else stage_piece<A_KM, false>(A, offa, pf_ks * a_kstep, img, q, wave);
    };
    auto advance = [&]() {
        if (!pf_live) return;
        pf_buf ^= 1;
        if (++pf_ks < pf.ks1) return;
        pf_item += gridDim.x;
        if (pf_item >= nitems) {
            pf_live = false;
            return;
        }
        pf = item256(p, pf_item);
        pf_ks = pf.ks0;
        stage_offsets<A_KM, false>(offa, p.lda, pf.m0, p.M, wave, lane);
        stage_offsets<B_KM, true>(offb, p.ldb, pf.n0, p.N, wave, lane);
    };

#ifdef VITK_GEMM_STAMPS
    // timeline mode (VITK_GEMM_DEBUG=64): [0] kernel start, then per item r < 10: main loop start / end, epilogue end
    const bool tl = (p.debug & 64) && lane == 0 && (wave & 3) == 0;
    long* tls = reinterpret_cast<long*>(p.aux_out) + blockIdx.x * 64 + wm * 32;
    int tl_r = 0;
    if (tl) tls[0] = __builtin_amdgcn_s_memtime();
#endif
    issue_slot(0);
    issue_slot(1);
    issue_slot(2);
    issue_slot(3);
    advance();
    issue_slot(0);
    issue_slot(1);
    // in flight behind what the first phase reads.  Two-phase: B and A q0 (the whole A when it is K-major) of
    // tile 0 -> [A q1 (2)] + B of tile 1 (4) may fly; four-phase: the two pieces of tile 1.
    if (!pf_live) VITK_VMCNT(0);
    else if (TWO_PHASE && !A_KM) VITK_VMCNT(6);
    else VITK_VMCNT(4);
    VITK_BARRIER();
    if (wm == 1) VITK_BARRIER();   // the second wave group runs one barrier behind the first
    int cur = 0;

    while (true) {
        f32x4 acc[8][4];
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        const int next = item + gridDim.x;
        const Item256 me = it;
#ifdef VITK_GEMM_STAMPS
        if (tl && tl_r < 10) tls[1 + 3 * tl_r] = __builtin_amdgcn_s_memtime();
#endif

        for (int ks = me.ks0; ks < me.ks1; ++ks) {
            const char* la = smem + cur * STAGE256;
            const char* lb = la + OPER_BYTES;
            Frag<__bf16> fa[4][2], fb0[2][2], fb1[2][2];
#ifdef VITK_GEMM_STAMPS
            const bool stamp = (p.debug & 32) && lane == 0 && (wave & 3) == 0 && ks == me.ks0 + 5 && item == (int)blockIdx.x;
            long* stamps = reinterpret_cast<long*>(p.aux_out) + blockIdx.x * 64 + wm * 32;
#endif
            VITK_STAMP(0);

            if constexpr (TWO_PHASE) {
            // ---- phase A: rows 0-63 against the four column tiles.  Issues the A pieces of the cursor tile
            // (their stage was last read in phase B of the previous K tile).  The B fragments read here are
            // retired BEFORE the barrier: the other wave group restages those pieces in its next phase.
#pragma unroll
            for (int jt = 0; jt < 2; ++jt)
#pragma unroll
                for (int kb = 0; kb < 2; ++kb) {
                    frag_b<B_KM>(fb0[jt][kb], lb, wn * 64, jt, kb, lane);
                    frag_b<B_KM>(fb1[jt][kb], lb, wn * 64, 2 + jt, kb, lane);
                }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int kb = 0; kb < 2; ++kb) frag_a<A_KM>(fa[i][kb], la, wm * 128 + i * 16, kb, lane);
            VITK_FENCE();
            issue_slot(2);
            issue_slot(3);
            if (pf_live) VITK_VMCNT(8);     // phase B reads A q1 of this tile: behind it B (4) and A (4) of the tiles ahead
            else VITK_VMCNT(0);
            advance();
            VITK_HALF(0, true);

            // ---- phase B: rows 64-127.  Issues the B pieces of the (new) cursor tile into this tile's stage.
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int kb = 0; kb < 2; ++kb) frag_a<A_KM>(fa[i][kb], la, wm * 128 + 64 + i * 16, kb, lane);
            VITK_FENCE();
            issue_slot(0);
            issue_slot(1);
            if (!pf_live) VITK_VMCNT(0);    // next phase A reads B and A q0 (K-major: all of A) of the next tile
            else if (A_KM) VITK_VMCNT(4);
            else VITK_VMCNT(6);
            VITK_HALF(4, false);
            } else {
            // ---- phase 0: quadrant (rows 0-63, tiles 0,1)
#pragma unroll
            for (int jt = 0; jt < 2; ++jt)
#pragma unroll
                for (int kb = 0; kb < 2; ++kb) frag_b<B_KM>(fb0[jt][kb], lb, wn * 64, jt, kb, lane);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int kb = 0; kb < 2; ++kb) frag_a<A_KM>(fa[i][kb], la, wm * 128 + i * 16, kb, lane);
            VITK_FENCE();
            issue_slot(2);
            if (pf_live) VITK_VMCNT(8);
            else VITK_VMCNT(0);
            VITK_QUADRANT(0, 0, fb0, 0, false);

            // ---- phase 1: quadrant (rows 0-63, tiles 2,3)
#pragma unroll
            for (int jt = 0; jt < 2; ++jt)
#pragma unroll
                for (int kb = 0; kb < 2; ++kb) frag_b<B_KM>(fb1[jt][kb], lb, wn * 64, 2 + jt, kb, lane);
            VITK_FENCE();
            issue_slot(3);
            if (pf_live) VITK_VMCNT(8);
            else VITK_VMCNT(0);
            advance();
            VITK_QUADRANT(0, 2, fb1, 5, A_KM);   // K-major A: piece B 0 is restaged in the next phase

            // ---- phase 2: quadrant (rows 64-127, tiles 2,3)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int kb = 0; kb < 2; ++kb) frag_a<A_KM>(fa[i][kb], la, wm * 128 + 64 + i * 16, kb, lane);
            VITK_FENCE();
            issue_slot(0);
            VITK_QUADRANT(4, 2, fb1, 10, false);

            // ---- phase 3: quadrant (rows 64-127, tiles 0,1); the next K tile must have landed
            issue_slot(1);
            if (!pf_live) VITK_VMCNT(0);
            else if (A_KM) VITK_VMCNT(4);
            else if (B_KM) VITK_VMCNT(6);
            else VITK_VMCNT(8);
            VITK_QUADRANT(4, 0, fb0, 15, false);
            }
            cur ^= 1;
        }

        {
#ifdef VITK_GEMM_STAMPS
            const bool stamp = (p.debug & 32) && lane == 0 && (wave & 3) == 0 && item == (int)blockIdx.x;
            long* stamps = reinterpret_cast<long*>(p.aux_out) + blockIdx.x * 64 + wm * 32;
#endif
            VITK_STAMP(21);
#ifdef VITK_GEMM_STAMPS
            if (tl && tl_r < 10) tls[2 + 3 * tl_r] = __builtin_amdgcn_s_memtime();
#endif
            // The fast form's loads come from inline asm and are waited for by hand: the compiler takes their registers for
            // valid at once, and a spill in between stores data that has not landed and hands the register on (the load
            // then lands on the new owner -- seen as a wild pointer and a GPU fault when unrelated code raised the
            // pressure).  tools/check_async_loads.py reads the generated assembly for exactly that (a CPU test runs it);
            // it found the bias loads of the K-major-A instances spilled that way.  Those are the dW GEMMs, always
            // split-K in the layer: they do not carry the fast form (which also takes them from ~28 spilled VGPRs to 2).
            epilogue256<TC, !A_KM>(p, acc, wl, me.m0 + wm * 128, me.n0 + wn * 64, me.split, lane);
            VITK_STAMP(22);
#ifdef VITK_GEMM_STAMPS
            if (tl && tl_r < 10) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                tls[3 + 3 * tl_r] = __builtin_amdgcn_s_memtime();
                ++tl_r;
            }
#endif
        }
        if (next >= nitems) break;
        item = next;
        it = item256(p, item);
    }
    if (wm == 0) VITK_BARRIER();
}
#undef VITK_QUADRANT
#undef VITK_BARRIER
#undef VITK_VMCNT
#undef VITK_FENCE

extern "C" int vitk_get_reserved_cus(void);
static int cu_count() {
    static int n = 0;
    if (n == 0) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess)
            n = prop.multiProcessorCount;
        if (n <= 0) n = 256;
    }
    return n;
}

template <typename TC> static void launch256(const GemmParams& p, int transA, int transB, hipStream_t stream) {
    const int nitems = p.tiles_m * p.tiles_n * p.splitk;
    // a multiple of 8 keeps item -> XCD fixed across rounds; vitk_set_reserved_cus() leaves CUs to other kernels (RCCL)
    int grid = (cu_count() - vitk_get_reserved_cus()) & ~7;
    if (grid < 8) grid = 8;
    if (grid > nitems) grid = nitems;
    const size_t lds = STAGES_BYTES + EPI_BYTES;
    const bool a_km = transA != 0, b_km = transB == 0;
#define VITK_L256(AK, BK_)                                                                           \
    do {                                                                                             \
        (void)vitk_lds_optin((const void*)gemm256_kernel<TC, AK, BK_>, lds);                         \
        hipLaunchKernelGGL((gemm256_kernel<TC, AK, BK_>), dim3(grid), dim3(512), lds, stream, p);    \
    } while (0)
    if (!a_km && !b_km) VITK_L256(false, false);
    else if (!a_km && b_km) VITK_L256(false, true);
    else if (a_km && !b_km) VITK_L256(true, false);
    else VITK_L256(true, true);
#undef VITK_L256
}

// Shapes the 256 x 256 kernel accepts (vitk_gemm falls back to the generic 128 x 128 kernel otherwise).
bool gemm256_eligible(const GemmParams& p, int transA, int transB) {
    if (p.K % 64 != 0 || p.K < 128 || !p.a_vec_ok || !p.b_vec_ok) return false;
    if ((long)p.M * p.N < 512L * 512L) return false;   // small problems: finer tiles fill the chip better
    if (p.M < 8 || p.N < 8) return false;
    if (transA && p.M % 8 != 0) return false;
    if (!transB && p.N % 8 != 0) return false;
    return true;
}

void gemm256_launch(const GemmParams& p0, int c_dtype, int transA, int transB, hipStream_t stream) {
    GemmParams p = p0;
    // half a tile time in shader clock ticks (3.6 k per 64-deep K tile + epilogue); only plain-store epilogues
    static int pct = -1, groups = 4;       // VITK_GEMM_STAGGER = total spread in % of a tile time (0: off), _GROUPS = phases
    if (pct < 0) {
        const char* e = getenv("VITK_GEMM_STAGGER");
        pct = e ? atoi(e) : 25;
        const char* g = getenv("VITK_GEMM_STAGGER_GROUPS");
        groups = g ? atoi(g) : 4;
        if (groups < 2 || groups > 32) groups = 4;
    }
    const long tile_ticks = (long)p.K / 64 * 3600 + 10000;
    long ticks = tile_ticks * pct / 100 / (groups - 1);    // spacing between consecutive phases: the last one starts pct % of a tile late
    if (ticks > 0xffffff) ticks = 0xffffff;
    p.stagger = (pct > 0 && p.splitk == 1) ? (int)((groups << 24) | ticks) : 0;
    if (c_dtype == VITK_BF16) launch256<__bf16>(p, transA, transB, stream);
    else launch256<float>(p, transA, transB, stream);
}

}  // namespace vitk
