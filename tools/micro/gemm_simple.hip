// A deliberately SIMPLE 256 x 256 bf16 GEMM (C = A B^T, both K-contiguous), to measure what the plain structure gives:
// eight waves (2 x 4, wave tile 128 x 64), 64-deep K tiles written to LDS by global_load_lds_dwordx4 (the swizzled
// image of gemm256.hip), NSTAGE LDS stages, ONE workgroup barrier pair per K tile, the two waves of a SIMD overlapping
// each other's LDS reads and MFMAs by themselves (no explicit ping-pong phases).  tools/micro/lds_tile.hip says the
// LDS + MFMA side of this sustains 1.6-1.9 PFLOP/s; gemm256_kernel (eight barriers per K tile) runs 1.35 on 4096^3.
// One tile per workgroup, M, N multiples of 256, K a multiple of 64; no epilogue beyond the bf16 store.
//   hipcc --offload-arch=gfx950 -O3 gemm_simple.hip -o gemm_simple ; ./gemm_simple [M N K]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <vector>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

constexpr int OPER = 256 * 64 * 2;
constexpr int STAGE = 2 * OPER;

__device__ __forceinline__ bf16x8 frag(const char* img, int row_base, int kb, int lane) {
    const int g = lane >> 4, i = lane & 15;
    const int c = (4 * kb + g) ^ (i >> 1);
    return *reinterpret_cast<const bf16x8*>(img + (row_base + i) * 128 + (c << 4));
}

#define FENCE() __builtin_amdgcn_sched_barrier(0)

template <int NSTAGE>
__global__ __launch_bounds__(512) void gemm_simple(const __bf16* __restrict__ A, const __bf16* __restrict__ B, __bf16* __restrict__ C,
                                                   int M, int N, int K, int tiles_n) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;
    const int tm = blockIdx.x / tiles_n, tn = blockIdx.x - tm * tiles_n;
    const int m0 = tm * 256, n0 = tn * 256;
    const int KT = K / 64;

    // this wave's share of a K tile: 4 LDS-DMA instructions per operand, 8 rows x 128 bytes each
    // instruction j of the wave covers image rows 32 * wave + 8 * j + (lane >> 3); lane's chunk (lane & 7), swizzled at the source
    long offa[4], offb[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int r = 32 * wave + 8 * j + (lane >> 3);
        const int c = (lane & 7) ^ ((r >> 1) & 7);
        offa[j] = (long)(m0 + r) * K + 8 * c;
        offb[j] = (long)(n0 + r) * K + 8 * c;
    }
    auto issue = [&](int kt) {
        char* img = smem + (kt % NSTAGE) * STAGE;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            __builtin_amdgcn_global_load_lds((gptr_t)(A + offa[j] + (long)kt * 64), (lptr_t)(img + (32 * wave + 8 * j) * 128), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((gptr_t)(B + offb[j] + (long)kt * 64), (lptr_t)(img + OPER + (32 * wave + 8 * j) * 128), 16, 0, 0);
        }
    };

    f32x4 acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // NSTAGE - 1 tiles in flight
#pragma unroll
    for (int s = 0; s < NSTAGE - 1; ++s)
        if (s < KT) issue(s);
    for (int kt = 0; kt < KT; ++kt) {
        // tile kt has landed once at most the younger tiles' loads of this wave are in flight
        const int ahead = min(NSTAGE - 2, KT - 1 - kt);        // tiles issued after kt so far
        if (ahead >= 2) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
        else if (ahead == 1) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        FENCE();
        __builtin_amdgcn_s_barrier();       // everybody's share of tile kt is in LDS; everybody is done with tile kt - 1
        FENCE();
        if (kt + NSTAGE - 1 < KT) issue(kt + NSTAGE - 1);     // into the stage tile kt - 1 was read from
        FENCE();
        const char* la = smem + (kt % NSTAGE) * STAGE;
        const char* lb = la + OPER;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
            bf16x8 fa[8], fb[4];
#pragma unroll
            for (int i = 0; i < 8; ++i) fa[i] = frag(la, wm * 128 + 16 * i, kb, lane);
#pragma unroll
            for (int j = 0; j < 4; ++j) fb[j] = frag(lb, wn * 64 + 16 * j, kb, lane);
#pragma unroll
            for (int i = 0; i < 8; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[j], fa[i], acc[i][j], 0, 0, 0);
        }
        FENCE();
    }
    // D = Bfrag x Afrag: acc[i][j][r] = C[row 16 i + (lane & 15)][col 16 j + 4 (lane >> 4) + r]
    const int g = lane >> 4, li = lane & 15;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const long row = m0 + wm * 128 + 16 * i + li;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const f32x4 a = acc[i][j];
            *reinterpret_cast<bf16x4*>(C + row * N + n0 + wn * 64 + 16 * j + 4 * g) = bf16x4{(__bf16)a[0], (__bf16)a[1], (__bf16)a[2], (__bf16)a[3]};
        }
    }
}

// The same with 32-deep K tiles and four stages (4 x 32 KiB): three tiles in flight instead of one, a barrier per 32
// MFMAs of a wave.  Image rows are 64 bytes; chunk c of row r sits at c ^ tbl[(r >> 2) & 3], tbl = {0, 2, 3, 1}: the
// sixteen lanes of every ds_read_b128 group then hit sixteen different 16-byte slots of the bank row.
constexpr int OPER32 = 256 * 32 * 2;
constexpr int STAGE32 = 2 * OPER32;
__device__ __forceinline__ int swz32(int r) { return (0x78 >> (2 * ((r >> 2) & 3))) & 3; }   // {0, 2, 3, 1}
__device__ __forceinline__ bf16x8 frag32(const char* img, int row_base, int lane) {
    const int g = lane >> 4, r = row_base + (lane & 15);
    return *reinterpret_cast<const bf16x8*>(img + r * 64 + ((g ^ swz32(r)) << 4));
}

template <int NSTAGE>
__global__ __launch_bounds__(512) void gemm_simple32(const __bf16* __restrict__ A, const __bf16* __restrict__ B, __bf16* __restrict__ C,
                                                     int M, int N, int K, int tiles_n) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;
    const int tm = blockIdx.x / tiles_n, tn = blockIdx.x - tm * tiles_n;
    const int m0 = tm * 256, n0 = tn * 256;
    const int KT = K / 32;
    // per operand and stage a wave issues 2 LDS-DMA instructions of 16 rows x 64 bytes: rows 32 * wave + 16 * j + (lane >> 2)
    long offa[2], offb[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int r = 32 * wave + 16 * j + (lane >> 2);
        const int c = (lane & 3) ^ swz32(r);
        offa[j] = (long)(m0 + r) * K + 8 * c;
        offb[j] = (long)(n0 + r) * K + 8 * c;
    }
    auto issue = [&](int kt) {
        char* img = smem + (kt % NSTAGE) * STAGE32;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            __builtin_amdgcn_global_load_lds((gptr_t)(A + offa[j] + (long)kt * 32), (lptr_t)(img + (32 * wave + 16 * j) * 64), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((gptr_t)(B + offb[j] + (long)kt * 32), (lptr_t)(img + OPER32 + (32 * wave + 16 * j) * 64), 16, 0, 0);
        }
    };
    f32x4 acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < NSTAGE - 1; ++s)
        if (s < KT) issue(s);
    for (int kt = 0; kt < KT; ++kt) {
        const int ahead = min(NSTAGE - 2, KT - 1 - kt);        // tiles issued after kt so far (4 loads each)
        if (ahead >= 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else if (ahead == 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        FENCE();
        __builtin_amdgcn_s_barrier();
        FENCE();
        if (kt + NSTAGE - 1 < KT) issue(kt + NSTAGE - 1);     // into the stage tile kt - 1 was read from
        FENCE();
        const char* la = smem + (kt % NSTAGE) * STAGE32;
        const char* lb = la + OPER32;
        bf16x8 fa[8], fb[4];
#pragma unroll
        for (int i = 0; i < 8; ++i) fa[i] = frag32(la, wm * 128 + 16 * i, lane);
#pragma unroll
        for (int j = 0; j < 4; ++j) fb[j] = frag32(lb, wn * 64 + 16 * j, lane);
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[j], fa[i], acc[i][j], 0, 0, 0);
        FENCE();
    }
    const int g = lane >> 4, li = lane & 15;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const long row = m0 + wm * 128 + 16 * i + li;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const f32x4 a = acc[i][j];
            *reinterpret_cast<bf16x4*>(C + row * N + n0 + wn * 64 + 16 * j + 4 * g) = bf16x4{(__bf16)a[0], (__bf16)a[1], (__bf16)a[2], (__bf16)a[3]};
        }
    }
}

static float bf(unsigned short h) {
    unsigned u = (unsigned)h << 16;
    float f;
    memcpy(&f, &u, 4);
    return f;
}

template <int NSTAGE, bool K32> static void run(int M, int N, int K, const __bf16* A, const __bf16* B, __bf16* C, const std::vector<unsigned short>& hA,
                                      const std::vector<unsigned short>& hB) {
    const int tiles_m = M / 256, tiles_n = N / 256;
    const size_t lds = (size_t)NSTAGE * (K32 ? STAGE32 : STAGE);
    auto kern = K32 ? gemm_simple32<NSTAGE> : gemm_simple<K32 ? 2 : NSTAGE>;
    (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    float ms = 0, best = 1e9f;
    for (int rep = 0; rep < 6; ++rep) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(kern, dim3(tiles_m * tiles_n), dim3(512), lds, 0, A, B, C, M, N, K, tiles_n);
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        (void)hipEventElapsedTime(&ms, e0, e1);
        if (rep > 0 && ms < best) best = ms;
    }
    const hipError_t err = hipGetLastError();
    std::vector<unsigned short> hC((size_t)M * N);
    (void)hipMemcpy(hC.data(), C, hC.size() * 2, hipMemcpyDeviceToHost);
    double worst = 0;
    for (int t = 0; t < 200; ++t) {
        const int r = (int)((1103515245u * (t + 7) + 12345u) % (unsigned)M), c = (int)((22695477u * (t + 3) + 1u) % (unsigned)N);
        double s = 0;
        for (int k = 0; k < K; ++k) s += (double)bf(hA[(size_t)r * K + k]) * bf(hB[(size_t)c * K + k]);
        const double e = fabs(bf(hC[(size_t)r * N + c]) - s) / (fabs(s) + 1.0);
        if (e > worst) worst = e;
    }
    printf("M %d N %d K %d  K tile %d, stages %d: %8.3f ms  %7.1f TFLOP/s   worst rel err of 200 samples %.2e%s\n", M, N, K, K32 ? 32 : 64, NSTAGE, best,
           2.0 * M * N * K / best / 1e9, worst, err == hipSuccess ? "" : "  (launch error)");
}

int main(int argc, char** argv) {
    int shapes[][3] = {{4096, 4096, 4096}, {25088, 1024, 4096}, {25088, 3072, 1024}, {25088, 1024, 1024}, {25088, 4096, 1024}};
    int ns = 5;
    if (argc == 4) {
        shapes[0][0] = atoi(argv[1]);
        shapes[0][1] = atoi(argv[2]);
        shapes[0][2] = atoi(argv[3]);
        ns = 1;
    }
    for (int s = 0; s < ns; ++s) {
        const int M = shapes[s][0], N = shapes[s][1], K = shapes[s][2];
        if (M % 256 || N % 256 || K % 64 || K < 128) {
            printf("shape %d %d %d not supported\n", M, N, K);
            continue;
        }
        std::vector<unsigned short> hA((size_t)M * K), hB((size_t)N * K);
        unsigned x = 12345u + s;
        auto rnd = [&]() {   // bf16 in roughly [-1, 1)
            x = x * 1664525u + 1013904223u;
            const float f = ((x >> 8) & 0xffff) / 32768.0f - 1.0f;
            unsigned u;
            memcpy(&u, &f, 4);
            return (unsigned short)(u >> 16);
        };
        for (auto& v : hA) v = rnd();
        for (auto& v : hB) v = rnd();
        __bf16 *A, *B, *C;
        if (hipMalloc(&A, hA.size() * 2) != hipSuccess || hipMalloc(&B, hB.size() * 2) != hipSuccess ||
            hipMalloc(&C, (size_t)M * N * 2) != hipSuccess)
            return 1;
        (void)hipMemcpy(A, hA.data(), hA.size() * 2, hipMemcpyHostToDevice);
        (void)hipMemcpy(B, hB.data(), hB.size() * 2, hipMemcpyHostToDevice);
        run<2, false>(M, N, K, A, B, C, hA, hB);
        run<4, true>(M, N, K, A, B, C, hA, hB);
        run<3, true>(M, N, K, A, B, C, hA, hB);
        (void)hipFree(A);
        (void)hipFree(B);
        (void)hipFree(C);
    }
    return 0;
}
