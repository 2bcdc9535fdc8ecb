// MFMA issue-rate microbenchmark: what the matrix pipes deliver with no memory traffic at all, at
// short (one CU busy) and sustained (whole chip busy) load -- the practical ceiling the GEMM main
// loop is measured against, and the s_memtime tick rate against wall time.
//   hipcc --offload-arch=gfx950 -O3 mfma_rate.hip -o mfma_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

template <int NACC>
__global__ __launch_bounds__(512) void k(float* out, long* ticks, int iters) {
    const int lane = threadIdx.x & 63;
    bf16x8 a, b;
    for (int e = 0; e < 8; ++e) { a[e] = (__bf16)(0.001f * (lane + e)); b[e] = (__bf16)(0.002f * (lane - e)); }
    f32x4 acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    const long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[i], 0, 0, 0);
    }
    const long t1 = __builtin_amdgcn_s_memtime();
    f32x4 s = acc[0];
    for (int i = 1; i < NACC; ++i) s += acc[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s[0] + s[1] + s[2] + s[3];
    if (lane == 0) ticks[blockIdx.x * 8 + (threadIdx.x >> 6)] = t1 - t0;
}

int main() {
    float* out; long* ticks;
    hipMalloc(&out, 4096 * 512 * sizeof(float));
    hipMalloc(&ticks, 4096 * 8 * sizeof(long));
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    std::vector<long> h(4096 * 8);
    const int NACC = 16;
    for (int blocks : {1, 8, 64, 256, 512}) {
        for (int threads : {256, 512}) {
            for (int iters : {2000, 200000}) {
                float ms = 0;
                for (int rep = 0; rep < 2; ++rep) {
                    hipEventRecord(e0);
                    hipLaunchKernelGGL(k<16>, dim3(blocks), dim3(threads), 0, 0, out, ticks, iters);
                    hipEventRecord(e1);
                    hipEventSynchronize(e1);
                    hipEventElapsedTime(&ms, e0, e1);
                }
                hipMemcpy(h.data(), ticks, blocks * 8 * sizeof(long), hipMemcpyDeviceToHost);
                const int waves = threads / 64;
                double s = 0; int n = 0;
                for (int b = 0; b < blocks; ++b) for (int w = 0; w < waves; ++w) { s += h[b * 8 + w]; ++n; }
                const double mf = (double)iters * NACC;                      // MFMAs per wave
                const double flops = mf * 16384.0 * waves * blocks;
                printf("blocks %4d x %d waves, %7d iters: %8.3f ms, %7.1f TFLOP/s, %.2f ticks per MFMA per wave, %.1f ticks/us\n",
                       blocks, waves, iters, ms, flops / ms / 1e9, s / n / mf, s / n / (ms * 1e3));
            }
        }
    }
    return 0;
}
