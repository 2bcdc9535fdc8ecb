// Store-issue microbenchmark: cycles for a workgroup to issue (and retire) a 256 x 256 bf16 tile's
// worth of 16-byte stores in different shapes.  hipcc --offload-arch=gfx950 -O3 store_rate.hip -o store_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

// mode 0: per instr 16 rows x 64 B (4 lanes per row)      [the register-direct epilogue]
// mode 1: per instr  8 rows x 128 B (8 lanes per row)     [row-major through LDS]
// mode 2: per instr  4 rows x 256 B (16 lanes per row)
// mode 3: per instr  1 row x 1024 B
template <int MODE>
__global__ void k(char* out, long ld, int waves_active, long* cyc, int wait) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (wave >= waves_active) return;
    // tile of 256 rows x 512 B per workgroup; a wave owns 128 rows x 128 B (wave = wm*4 + wn)
    const int wm = wave >> 2, wn = wave & 3;
    char* base = out + (long)blockIdx.x * 256 * ld + (long)wm * 128 * ld + wn * 128;
    u32x4 v = {1u, 2u, 3u, (unsigned)lane};
    long t0 = __builtin_amdgcn_s_memtime();
    constexpr int LPR = MODE == 0 ? 4 : MODE == 1 ? 8 : MODE == 2 ? 16 : 64;   // lanes per row
    constexpr int RPI = 64 / LPR;                                              // rows per instruction
    // the wave's 128 x 128 B block = 16 KiB = 16 instructions
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        // instruction i covers bytes: rows and columns chosen so that all 16 instructions tile 128 rows x 128 B
        const int chunk = i * 64 + lane;            // 16-byte chunk index in some order
        int row, col;
        if (MODE == 0) { const int half = i & 1; row = (i >> 1) * 16 + (lane >> 2); col = half * 64 + (lane & 3) * 16; }
        else if (MODE == 1) { row = i * 8 + (lane >> 3); col = (lane & 7) * 16; }
        else { row = chunk >> 3; col = (chunk & 7) * 16; }
        *reinterpret_cast<u32x4*>(base + (long)row * ld + col) = v;
    }
    if (wait) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    long t1 = __builtin_amdgcn_s_memtime();
    if (lane == 0) cyc[blockIdx.x * 8 + wave] = t1 - t0;
}

int main(int argc, char** argv) {
    const long ld = 8192;   // bytes per row (N = 4096 bf16)
    const int blocks = argc > 1 ? atoi(argv[1]) : 256;   // 8: one workgroup per XCD; 64: eight per XCD
    char* out; long* cyc;
    hipMalloc(&out, (size_t)blocks * 256 * ld);
    hipMalloc(&cyc, blocks * 8 * sizeof(long));
    std::vector<long> h(blocks * 8);
    for (int wait = 0; wait < 2; ++wait)
        for (int wa : {4, 8})
            for (int mode = 0; mode < 3; ++mode) {
                for (int rep = 0; rep < 3; ++rep) {
                    hipMemset(cyc, 0, blocks * 8 * sizeof(long));
                    if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(512), 0, 0, out, ld, wa, cyc, wait);
                    if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(512), 0, 0, out, ld, wa, cyc, wait);
                    if (mode == 2) hipLaunchKernelGGL(k<2>, dim3(blocks), dim3(512), 0, 0, out, ld, wa, cyc, wait);
                    hipDeviceSynchronize();
                }
                hipMemcpy(h.data(), cyc, blocks * 8 * sizeof(long), hipMemcpyDeviceToHost);
                double s = 0, mx = 0; int n = 0;
                for (int b = 0; b < blocks; ++b) for (int w = 0; w < wa; ++w) { s += h[b * 8 + w]; if (h[b*8+w] > mx) mx = h[b*8+w]; ++n; }
                printf("blocks=%d wait=%d waves=%d mode=%d: mean %.0f cycles, max %.0f per wave for 16 stores (%d KB per CU)\n", blocks, wait, wa, mode, s / n, mx, wa * 16);
            }
    return 0;
}
