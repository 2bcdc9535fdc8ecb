// Does a captured graph with hipMemsetAsync / hipMemset2DAsync nodes in front of accumulating kernels replay
// correctly?  (with memset nodes in the captured training step, the second replay accumulated onto
// stale data; the library zero-fills with a kernel of its own since.)  P (buffer, memset, accumulate) triples per
// graph, some buffers zeroed and accumulated twice with a reader in between; every replay must leave 1.0 everywhere.
// build: hipcc --offload-arch=gfx950 -O2 tools/micro/memset_graph.hip -o tools/micro/memset_graph
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); return 2; } } while (0)

__global__ void accum(float* p, long n, float v) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) atomicAdd(p + i, v);
}
__global__ void check(const float* p, long n, float want, int* bad) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
        if (p[i] != want) atomicAdd(bad, 1);
}

int main(int argc, char** argv) {
    const int P = argc > 1 ? atoi(argv[1]) : 64, R = 4;
    const int mode = argc > 2 ? atoi(argv[2]) : 0;     // 0: hipMemsetAsync, 1: hipMemset2DAsync (pitched rows), 2: hipMemsetD32Async, 3: hipMemsetAsync, tiny buffers
    hipStream_t st;
    CK(hipStreamCreate(&st));
    std::vector<float*> bufs(P);
    std::vector<long> ns(P);
    for (int i = 0; i < P; ++i) {
        ns[i] = (i % 3 == 0) ? 1024L * 1024 * 4 : (i % 3 == 1 ? 3 * 1024 : 1000 * 1024);
        if (mode == 3) ns[i] = (i % 3 == 0) ? 2 : (i % 3 == 1 ? 4 : 48);       // 8 / 16 / 192 bytes: the small accumulators
        CK(hipMalloc(&bufs[i], ns[i] * 4));
        CK(hipMemset(bufs[i], 0x7f, ns[i] * 4));
    }
    int* bad;
    CK(hipMalloc(&bad, 4));
    CK(hipMemset(bad, 0, 4));
    hipGraph_t g;
    hipGraphExec_t ge;
    CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
    for (int i = 0; i < P; ++i) {
        for (int twice = 0; twice < (i % 4 == 0 ? 2 : 1); ++twice) {
            if (mode == 0 || mode == 3) CK(hipMemsetAsync(bufs[i], 0, ns[i] * 4, st));
            else if (mode == 1) CK(hipMemset2DAsync(bufs[i], 1024 * 4, 0, 1000 * 4, ns[i] / 1024, st));
            else CK(hipMemsetD32Async((hipDeviceptr_t)bufs[i], 0, ns[i], st));
            const long n = ns[i];
            hipLaunchKernelGGL(accum, dim3(256), dim3(256), 0, st, bufs[i], n, 0.25f);
            hipLaunchKernelGGL(accum, dim3(512), dim3(256), 0, st, bufs[i], n, 0.75f);
            // mode 1 zeroes 1000 of every 1024 floats of a row: the other 24 keep growing and are not checked
            if (mode != 1) hipLaunchKernelGGL(check, dim3(256), dim3(256), 0, st, bufs[i], n, 1.0f, bad);
            else for (long r = 0; r < ns[i] / 1024; r += 97) hipLaunchKernelGGL(check, dim3(1), dim3(256), 0, st, bufs[i] + r * 1024, 1000L, 1.0f, bad);
        }
    }
    CK(hipStreamEndCapture(st, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    for (int r = 0; r < R; ++r) {
        CK(hipGraphLaunch(ge, st));
        CK(hipStreamSynchronize(st));
        int hb = 0;
        CK(hipMemcpy(&hb, bad, 4, hipMemcpyDeviceToHost));
        printf("mode %d, %d triples, replay %d: %d wrong elements so far\n", mode, P, r, hb);
    }
    return 0;
}
