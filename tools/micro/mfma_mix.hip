// One wave per SIMD issues MFMAs back to back while a second wave on the same SIMD does something
// else (nothing / ds_read_b128 / VALU / s_sleep): how much does the neighbour slow the MFMA wave?
//   hipcc --offload-arch=gfx950 -O3 mfma_mix.hip -o mfma_mix
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

template <int MODE, int PRIO>
__global__ __launch_bounds__(512) void k(float* out, long* ticks, int iters) {
    __shared__ u32x4 lds[4096];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 4096; i += 512) lds[i] = u32x4{(unsigned)i, 1u, 2u, 3u};
    __syncthreads();
    float res = 0.f;
    long t0 = 0, t1 = 0;
    if (wave < 4) {
        bf16x8 a[4], b[4];   // distinct operand registers per MFMA, as in a GEMM quadrant
        for (int q = 0; q < 4; ++q)
            for (int e = 0; e < 8; ++e) { a[q][e] = (__bf16)(0.001f * (lane + e + q)); b[q][e] = (__bf16)(0.002f * (lane - e - q)); }
        f32x4 acc[16];
        for (int i = 0; i < 16; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (PRIO) __builtin_amdgcn_s_setprio(1);
        t0 = __builtin_amdgcn_s_memtime();
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i & 3], b[i >> 2], acc[i], 0, 0, 0);
        }
        t1 = __builtin_amdgcn_s_memtime();
        f32x4 s = acc[0];
        for (int i = 1; i < 16; ++i) s += acc[i];
        res = s[0] + s[1] + s[2] + s[3];
    } else if (MODE == 1) {          // LDS reads, 16 B per lane, conflict-free
        u32x4 s = {0, 0, 0, 0};
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 8; ++i) s += lds[(lane + 64 * i + it) & 4095];
        }
        res = (float)(s[0] + s[1] + s[2] + s[3]);
    } else if (MODE == 2) {          // VALU
        float x = lane * 0.5f, y = 1.0001f;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 32; ++i) x = x * y + 0.25f;
        }
        res = x;
    } else if (MODE == 3) {          // asleep most of the time
        for (int it = 0; it < iters; ++it) __builtin_amdgcn_s_sleep(8);
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = res;
    if (lane == 0 && wave < 4) ticks[blockIdx.x * 4 + wave] = t1 - t0;
}

template <int MODE, int PRIO> static void run(const char* name, float* out, long* ticks, std::vector<long>& h) {
    const int blocks = 256, iters = 20000;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    float ms = 0;
    for (int rep = 0; rep < 2; ++rep) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL((k<MODE, PRIO>), dim3(blocks), dim3(512), 0, 0, out, ticks, iters);
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        (void)hipEventElapsedTime(&ms, e0, e1);
    }
    (void)hipMemcpy(h.data(), ticks, blocks * 4 * sizeof(long), hipMemcpyDeviceToHost);
    double s = 0;
    for (int i = 0; i < blocks * 4; ++i) s += h[i];
    s /= blocks * 4;
    printf("%-28s prio %d: MFMA wave %.2f ticks per MFMA, its loop took %.3f of the kernel's %.3f ms (at 2.4 ticks/ns)\n", name, PRIO,
           s / (iters * 16.0), s / 2.4e6, ms);
}

int main() {
    float* out; long* ticks;
    (void)hipMalloc(&out, 256 * 512 * sizeof(float));
    (void)hipMalloc(&ticks, 256 * 4 * sizeof(long));
    std::vector<long> h(256 * 4);
    run<0, 0>("neighbour: none", out, ticks, h);
    run<1, 0>("neighbour: ds_read_b128", out, ticks, h);
    run<1, 1>("neighbour: ds_read_b128", out, ticks, h);
    run<2, 0>("neighbour: VALU fma", out, ticks, h);
    run<2, 1>("neighbour: VALU fma", out, ticks, h);
    run<3, 0>("neighbour: s_sleep", out, ticks, h);
    return 0;
}
