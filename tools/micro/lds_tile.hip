// How much of the MFMA rate does a 256 x 256 x 64 K tile keep when its fragments come from LDS, as a function of the
// WAVE tile?  No global memory in the loop: the A / B images sit in LDS (two 64 KiB stages, rewritten by the waves with
// ds_write_b128 when WRITES is set, to stand in for the LDS-DMA stream), every wave reads its fragments with
// ds_read_b128 on the swizzled K-contiguous image of gemm256.hip and runs the 16x16x32 bf16 MFMAs of its block.
//   NW = 8: waves 2 (M) x 4 (N), wave tile 128 x 64  (what gemm256_kernel does: 24 fragment reads per 64 MFMAs)
//   NW = 4: waves 2 x 2, wave tile 128 x 128         (32 reads per 128 MFMAs; 256 accumulator registers)
// One barrier per K tile in both.  Prints TFLOP/s over 256 workgroups (one per CU).
//   hipcc --offload-arch=gfx950 -O3 lds_tile.hip -o lds_tile
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

constexpr int OPER = 256 * 64 * 2;     // one operand tile, bytes
constexpr int STAGE = 2 * OPER;

// rows of 128 bytes, 16-byte chunks XOR-swizzled by (row >> 1) & 7 (conflict-free for 16 consecutive rows per fragment)
__device__ __forceinline__ bf16x8 frag(const char* img, int row_base, int kb, int lane) {
    const int g = lane >> 4, i = lane & 15;
    const int c = (4 * kb + g) ^ (i >> 1);
    return *reinterpret_cast<const bf16x8*>(img + (row_base + i) * 128 + (c << 4));
}

template <int NW, int IT, int JT, bool WRITES, bool SWAP>
__global__ __launch_bounds__(NW * 64) void tile_loop(float* out, int ktiles) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    constexpr int WN = NW == 8 ? 4 : 2;                 // waves along N
    const int wm = wave / WN, wn = wave % WN;
    const int arow = wm * 128, brow = wn * (256 / WN);
    for (int i = tid; i < 2 * STAGE / 16; i += NW * 64)
        reinterpret_cast<u32x4*>(smem)[i] = u32x4{0x3c003c00u + i, 0x3c003c00u, 0x3c003c01u, 0x3c003c02u};
    __syncthreads();
    f32x4 acc[IT][JT];
#pragma unroll
    for (int i = 0; i < IT; ++i)
#pragma unroll
        for (int j = 0; j < JT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    u32x4 wv = {1u, 2u, 3u, (unsigned)tid};
    for (int kt = 0; kt < ktiles; ++kt) {
        const char* la = smem + (kt & 1) * STAGE;
        const char* lb = la + OPER;
        if (WRITES) {      // the other stage is rewritten meanwhile: 64 KiB per K tile, 16 bytes per lane and instruction
            char* dst = smem + ((kt + 1) & 1) * STAGE;
#pragma unroll
            for (int q = 0; q < STAGE / 16 / (NW * 64); ++q)
                *reinterpret_cast<u32x4*>(dst + (q * NW * 64 + tid) * 16) = wv;
        }
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
            bf16x8 fa[IT], fb[JT];
#pragma unroll
            for (int i = 0; i < IT; ++i) fa[i] = frag(la, arow + 16 * i, kb, lane);
#pragma unroll
            for (int j = 0; j < JT; ++j) fb[j] = frag(lb, brow + 16 * j, kb, lane);
#pragma unroll
            for (int i = 0; i < IT; ++i)
#pragma unroll
                for (int j = 0; j < JT; ++j)
                    acc[i][j] = SWAP ? __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[j], fa[i], acc[i][j], 0, 0, 0)
                                     : __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[i], fb[j], acc[i][j], 0, 0, 0);
        }
        __syncthreads();
    }
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < IT; ++i)
#pragma unroll
        for (int j = 0; j < JT; ++j) s += acc[i][j];
    out[blockIdx.x * NW * 64 + tid] = s[0] + s[1] + s[2] + s[3];
}

template <int NW, int IT, int JT, bool WRITES> static void run(const char* name, float* out) {
    const int blocks = 256, ktiles = 4000;
    const size_t lds = 2 * STAGE;
    auto kern = tile_loop<NW, IT, JT, WRITES, true>;
    (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    float ms = 0;
    for (int rep = 0; rep < 3; ++rep) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(kern, dim3(blocks), dim3(NW * 64), lds, 0, out, ktiles);
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        (void)hipEventElapsedTime(&ms, e0, e1);
    }
    const hipError_t err = hipGetLastError();
    const double flop = 2.0 * 256 * 256 * 64 * (double)ktiles * blocks;
    printf("%-44s %8.3f ms  %7.1f TFLOP/s%s\n", name, ms, flop / ms / 1e9, err == hipSuccess ? "" : "  (launch error)");
}

int main() {
    float* out;
    if (hipMalloc(&out, 256 * 512 * sizeof(float)) != hipSuccess) return 1;
    run<8, 8, 4, false>("8 waves, 128 x 64 wave tiles, reads only", out);
    run<8, 8, 4, true>("8 waves, 128 x 64 wave tiles, reads + writes", out);
    run<4, 8, 8, false>("4 waves, 128 x 128 wave tiles, reads only", out);
    run<4, 8, 8, true>("4 waves, 128 x 128 wave tiles, reads + writes", out);
    (void)hipDeviceSynchronize();
    (void)hipFree(out);
    return 0;
}
