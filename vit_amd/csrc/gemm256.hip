// bf16 MFMA GEMM, 256 x 256 output tile, direct-to-LDS staging: the fast path behind vitk_gemm
// for the large linears of the encoder (qkv / proj / fc1 / fc2 forward, dX, dW).
//
// Geometry (one workgroup = 512 threads = 8 waves as 2 (M) x 4 (N), one workgroup per CU):
//   * 64 K-elements per step; A and B tiles (256 x 64 bf16 = 32 KiB each) are written straight
//     into LDS by global_load_lds_dwordx4 (1 KiB per wave instruction, 4 per wave per operand),
//     two LDS stages (128 KiB): the loads of step t+1 are in flight under the MFMAs of step t.
//   * a wave owns a 128 x 64 output block: 8 x 4 MFMA 16x16x32 tiles, 128 accumulator registers,
//     64 MFMAs per K step.
//   * K-contiguous operands (rows contiguous along K: activations, nn.Linear weights) use an
//     image of 128-byte rows whose 16-byte chunks are XOR-swizzled by (row >> 1) & 7: every
//     ds_read_b128 lane group then covers all 16 slots of the 256-byte bank row.  LDS-DMA writes
//     are lane-linear, so the swizzle is applied to each lane's SOURCE address.
//   * K-major operands (dX = dY W reads W as [K][N]; dW = dY^T X reads both as [K][.]) use an
//     image of 512-byte k-rows with chunks XOR-swizzled by 2 * ((k & 3) | ((k >> 3) & 1) << 2);
//     fragments come from ds_read_b64_tr_b16 (hardware transpose), conflict-free per 32-lane half.
//     No transposed copy of any operand exists in HBM.
//   * operands are swapped in the MFMA (D = Bfrag x Afrag) so a lane owns 4 consecutive output
//     columns: the fused epilogue (gemm_common.h) moves bias / aux / residual / C as vectors.
//   * XCD-aware grouped tile order; split-K (fp32 atomics) for the dW shapes.
// Requirements checked by the host: K % 64 == 0, 16-byte aligned rows, row counts of K-major
// operands multiples of 8.  Rows past M / N are clamped to the last valid row on load (their
// products land in output rows / columns that are never stored).
#include "gemm_common.h"

namespace vitk {

constexpr int T256 = 256;
constexpr int OPER_BYTES = 256 * 64 * 2;       // one operand tile
constexpr int STAGE256 = 2 * OPER_BYTES;       // A + B

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

__device__ __forceinline__ int km_swz(int k) { return 2 * ((k & 3) | (((k >> 3) & 1) << 2)); }

// Per-lane source offsets (in elements, without the K advance) of the 4 pieces a wave stages.
template <bool KM>
__device__ __forceinline__ void stage_offsets(long (&off)[4], long ld, int r0, int R, int wave, int lane) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int piece = wave * 4 + j;
        if (!KM) {
            const int r = piece * 8 + (lane >> 3);
            const int c = (lane & 7) ^ ((r >> 1) & 7);
            off[j] = (long)min(r0 + r, R - 1) * ld + 8 * c;
        } else {
            const int k = piece * 2 + (lane >> 5);
            const int cc = (lane & 31) ^ km_swz(k);
            off[j] = (long)k * ld + min(r0 + 8 * cc, R - 8);
        }
    }
}

template <bool KM>
__device__ __forceinline__ void stage_issue(const __bf16* __restrict__ X, const long (&off)[4], long kadv, char* lds,
                                            int wave) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const __bf16* src = X + off[j] + kadv;
        __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(lds + (wave * 4 + j) * 1024), 16, 0, 0);
    }
}

// Fragment of tile rows [row_base, +16) for MFMA K block kb (32 deep) of the staged image.
template <bool KM>
__device__ __forceinline__ void frag256(Frag<__bf16>& f, const char* lds, int row_base, int kb, int lane) {
    const int g = lane >> 4, i = lane & 15;
    if (!KM) {
        const int row = row_base + i;
        const int c = (4 * kb + g) ^ ((row >> 1) & 7);
        f.v = *reinterpret_cast<const bf16x8*>(lds + row * 128 + (c << 4));
    } else {
        const int q = i >> 2, pp = i & 3;
        const int k = 32 * kb + 8 * g + q;
        const int cc = ((row_base >> 3) + (pp >> 1)) ^ km_swz(k);
        const char* a0 = lds + k * 512 + (cc << 4) + 8 * (pp & 1);
        const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(uintptr_t)a0);
        const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
            (__attribute__((address_space(3))) s16x4*)(uintptr_t)(a0 + 4 * 512));
        const s16x8 r = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        f.v = __builtin_bit_cast(bf16x8, r);
    }
}

constexpr int ELD = 68;   // floats per staged epilogue row (64 + pad: conflict-free b128 writes)

// Rows [32 QD, 32 QD + 32) of a wave's 128 x 64 accumulator block: registers -> the wave's LDS
// scratch -> row-major read back (8 lanes per row, 8 columns per lane) -> fused epilogue.
template <typename TC, int QD>
__device__ __forceinline__ void epi_quarter(const GemmParams& p, const f32x4 (&acc)[8][4], float* wl, int mrow0,
                                            int ncol, const float (&bias8)[8], int lane) {
    const int g = lane >> 4, lm = lane & 15;
    const int r8 = lane >> 3, c8 = lane & 7;
#pragma unroll
    for (int ii = 0; ii < 2; ++ii)
#pragma unroll
        for (int j = 0; j < 4; ++j)
            *reinterpret_cast<f32x4*>(wl + (ii * 16 + lm) * ELD + j * 16 + 4 * g) = acc[2 * QD + ii][j];
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
    for (int ps = 0; ps < 4; ++ps) {
        const int row = ps * 8 + r8;
        const f32x4 lo = *reinterpret_cast<const f32x4*>(wl + row * ELD + 8 * c8);
        const f32x4 hi = *reinterpret_cast<const f32x4*>(wl + row * ELD + 8 * c8 + 4);
        const float v[8] = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        gemm_epilogue8<__bf16, TC>(p, 0, mrow0 + QD * 32 + row, ncol, v, bias8);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
}

template <int QD>
__device__ __forceinline__ void epi_quarter_atomic(const GemmParams& p, const f32x4 (&acc)[8][4], float* wl, int mrow0,
                                                   int col, float bcol, int lane) {
    const int g = lane >> 4, lm = lane & 15;
#pragma unroll
    for (int ii = 0; ii < 2; ++ii)
#pragma unroll
        for (int j = 0; j < 4; ++j)
            *reinterpret_cast<f32x4*>(wl + (ii * 16 + lm) * ELD + j * 16 + 4 * g) = acc[2 * QD + ii][j];
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    float* __restrict__ C = reinterpret_cast<float*>(p.C);
#pragma unroll 8
    for (int row = 0; row < 32; ++row) {
        const int m = mrow0 + QD * 32 + row;
        if (m < p.M && col < p.N) atomicAdd(C + (long)m * p.ldc + col, wl[row * ELD + lane] * p.alpha + bcol);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
}

template <typename TC, bool A_KM, bool B_KM>
__global__ __launch_bounds__(512, 2) void gemm256_kernel(const GemmParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;

    const int ntiles = p.tiles_m * p.tiles_n;
    const int split = blockIdx.x / ntiles;
    int tile_m, tile_n;
    gemm_tile_of(blockIdx.x - split * ntiles, p.tiles_m, p.tiles_n, tile_m, tile_n);
    const int m0 = tile_m * T256, n0 = tile_n * T256;

    const int ksteps_total = p.K / 64;
    const int steps_per = (ksteps_total + p.splitk - 1) / p.splitk;
    const int ks0 = split * steps_per;
    const int ks1 = min(ksteps_total, ks0 + steps_per);

    const __bf16* __restrict__ A = reinterpret_cast<const __bf16*>(p.A);
    const __bf16* __restrict__ B = reinterpret_cast<const __bf16*>(p.B);
    long offa[4], offb[4];
    stage_offsets<A_KM>(offa, p.lda, m0, p.M, wave, lane);
    stage_offsets<B_KM>(offb, p.ldb, n0, p.N, wave, lane);
    const long a_kstep = A_KM ? 64 * p.lda : 64;
    const long b_kstep = B_KM ? 64 * p.ldb : 64;

    f32x4 acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    if (ks0 < ks1) {
        stage_issue<A_KM>(A, offa, ks0 * a_kstep, smem, wave);
        stage_issue<B_KM>(B, offb, ks0 * b_kstep, smem + OPER_BYTES, wave);
    }
    __syncthreads();   // waits for the LDS-DMA (vmcnt(0)) and publishes it

    for (int ks = ks0; ks < ks1; ++ks) {
        const int cur = (ks - ks0) & 1;
        char* la = smem + cur * STAGE256;
        char* lb = la + OPER_BYTES;
        if (ks + 1 < ks1) {
            char* na = smem + (cur ^ 1) * STAGE256;
            stage_issue<A_KM>(A, offa, (ks + 1) * a_kstep, na, wave);
            stage_issue<B_KM>(B, offb, (ks + 1) * b_kstep, na + OPER_BYTES, wave);
        }
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
            Frag<__bf16> fb[4], fa[8];
#pragma unroll
            for (int j = 0; j < 4; ++j) frag256<B_KM>(fb[j], lb, wn * 64 + j * 16, kb, lane);
#pragma unroll
            for (int i = 0; i < 8; ++i) frag256<A_KM>(fa[i], la, wm * 128 + i * 16, kb, lane);
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int i = 0; i < 8; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = mma(fb[j], fa[i], acc[i][j]);
            __builtin_amdgcn_s_setprio(0);
        }
        __syncthreads();
    }

    const int g = lane >> 4, lm = lane & 15;
    float* wl = reinterpret_cast<float*>(smem + wave * 16384);
    if (p.splitk > 1) {
        // split-K: fp32 atomics, shaped as one contiguous 256-byte row segment per wave
        // instruction (lane = column) -- the shape the memory-side atomic units run at full rate.
        const int col = n0 + wn * 64 + lane;
        const float bcol = (p.bias != nullptr && split == 0 && col < p.N) ? p.bias[col] : 0.f;
        epi_quarter_atomic<0>(p, acc, wl, m0 + wm * 128, col, bcol, lane);
        epi_quarter_atomic<1>(p, acc, wl, m0 + wm * 128, col, bcol, lane);
        epi_quarter_atomic<2>(p, acc, wl, m0 + wm * 128, col, bcol, lane);
        epi_quarter_atomic<3>(p, acc, wl, m0 + wm * 128, col, bcol, lane);
        return;
    }
    // Row-major epilogue: the wave's 128 x 64 accumulator block goes through its private 16 KiB
    // of LDS (free now: the loop's last barrier has passed) 32 rows at a time, and is read back
    // with 8 lanes per row, 8 consecutive columns per lane: every global access of the fused
    // epilogue (C, aux, residual) is then a full 128-byte line per row instead of 32-byte pieces.
    const int c8 = lane & 7;
    const int ncol = n0 + wn * 64 + 8 * c8;
    float bias8[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) bias8[e] = (p.bias != nullptr && ncol + e < p.N) ? p.bias[ncol + e] : 0.f;
    epi_quarter<TC, 0>(p, acc, wl, m0 + wm * 128, ncol, bias8, lane);
    epi_quarter<TC, 1>(p, acc, wl, m0 + wm * 128, ncol, bias8, lane);
    epi_quarter<TC, 2>(p, acc, wl, m0 + wm * 128, ncol, bias8, lane);
    epi_quarter<TC, 3>(p, acc, wl, m0 + wm * 128, ncol, bias8, lane);
}

template <typename TC> static void launch256(const GemmParams& p, int transA, int transB, hipStream_t stream) {
    const int grid = p.tiles_m * p.tiles_n * p.splitk;
    const size_t lds = 2 * STAGE256;
    const bool a_km = transA != 0, b_km = transB == 0;
#define VITK_L256(AK, BK_)                                                                           \
    do {                                                                                             \
        static bool attr_set = false;                                                                \
        if (!attr_set) {                                                                             \
            (void)hipFuncSetAttribute((const void*)gemm256_kernel<TC, AK, BK_>,                       \
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);         \
            attr_set = true;                                                                         \
        }                                                                                            \
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

void gemm256_launch(const GemmParams& p, int c_dtype, int transA, int transB, hipStream_t stream) {
    if (c_dtype == VITK_BF16) launch256<__bf16>(p, transA, transB, stream);
    else launch256<float>(p, transA, transB, stream);
}

}  // namespace vitk
