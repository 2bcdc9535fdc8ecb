// Library-wide runtime bits of the C-ABI: version, per-thread error text.
#include <cstdlib>
#include <map>
#include <mutex>
#include <utility>
#include <cstring>
#include <stdarg.h>
#include <stdio.h>

#include "common.h"

static thread_local char g_err[512] = "";

extern "C" void vitk_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char* vitk_last_error(void) { return g_err; }

// ---- dynamic-LDS opt-in, per (kernel, device) --------------------------------------------------------------------
static std::mutex g_optin_mu;
static std::map<std::pair<const void*, int>, size_t> g_optin;      // largest size already granted
static thread_local bool g_optin_bad = false;

extern "C" int vitk_lds_optin(const void* kernel, size_t bytes) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) dev = 0;
    std::lock_guard<std::mutex> lock(g_optin_mu);
    size_t& have = g_optin[std::make_pair(kernel, dev)];
    if (bytes <= have) return VITK_OK;
    const hipError_t e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    if (e != hipSuccess) {
        vitk_set_error("dynamic LDS opt-in of %zu bytes failed on device %d: %s", bytes, dev, hipGetErrorString(e));
        g_optin_bad = true;
        (void)hipGetLastError();
        return VITK_ERR_LAUNCH;
    }
    have = bytes;
    return VITK_OK;
}
extern "C" int vitk_optin_failed(void) {
    const bool bad = g_optin_bad;
    g_optin_bad = false;
    return bad ? 1 : 0;
}

// Bumped whenever an entry point's argument list changes (round 2 added arguments to vitk_attn_fwd / _bwd without a bump:
// a stale library then misaligns arguments silently).  include/vitk.h carries the same number as VITK_ABI_VERSION and
// vit_amd/_lib.py refuses a library whose version differs from the one it was written for.
extern "C" int vitk_version(void) { return 3; }

// CUs the persistent launches (one 160 KiB-LDS workgroup per CU: gemm256) leave free -- for RCCL's reduction kernels
// while gradient buckets are in flight (a workgroup that owns a CU's whole LDS keeps every other kernel off that CU
// until it exits).  Process-wide; read at launch time.
static int g_reserved_cus = -1;
extern "C" int vitk_set_reserved_cus(int n) {
    VITK_CHECK_ARG(n >= 0 && n <= 128, "vitk_set_reserved_cus: %d is outside [0, 128]", n);
    g_reserved_cus = n;
    return VITK_OK;
}
extern "C" int vitk_get_reserved_cus(void) {
    if (g_reserved_cus < 0) {
        const char* e = getenv("VITK_RESERVED_CUS");
        int n = e ? atoi(e) : 0;
        g_reserved_cus = n < 0 ? 0 : (n > 128 ? 128 : n);
    }
    return g_reserved_cus;
}

namespace vitk {
// rows x row_bytes at a row pitch; 4-byte granularity (every caller zeroes fp32 / 8-byte counters)
__global__ __launch_bounds__(256) void zero_kernel(char* __restrict__ ptr, size_t pitch, size_t row_words, size_t rows) {
    const size_t total = row_words * rows;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const size_t r = i / row_words, c = i - r * row_words;
        *reinterpret_cast<unsigned*>(ptr + r * pitch + 4 * c) = 0u;
    }
}
struct UploadChunk {
    unsigned long long w[256];      // 2 KiB of kernel arguments
};
__global__ __launch_bounds__(256) void upload_kernel(const UploadChunk c, unsigned long long* __restrict__ dst, int n) {
    if ((int)threadIdx.x < n) dst[threadIdx.x] = c.w[threadIdx.x];
}
}  // namespace vitk

// Host bytes -> device memory THROUGH KERNEL ARGUMENTS (2 KiB per launch): stream-ordered, legal inside a stream
// capture (a replay rewrites the same bytes), and the host buffer may be reused or freed as soon as the call
// returns.  For small tables (descriptor lists) only.
extern "C" int vitk_upload_args(void* dst, const void* src_host, size_t bytes, hipStream_t stream) {
    if (bytes == 0) return VITK_OK;
    VITK_CHECK_ARG(dst && src_host && bytes % 8 == 0 && (reinterpret_cast<uintptr_t>(dst) & 7) == 0,
                   "vitk_upload_args: 8-byte granularity");
    VITK_CHECK_ARG(bytes <= (1u << 20), "vitk_upload_args: %zu bytes is not a small table", bytes);
    const char* src = static_cast<const char*>(src_host);
    char* d = static_cast<char*>(dst);
    for (size_t off = 0; off < bytes; off += sizeof(vitk::UploadChunk)) {
        const size_t nb = bytes - off < sizeof(vitk::UploadChunk) ? bytes - off : sizeof(vitk::UploadChunk);
        vitk::UploadChunk c;
        memcpy(c.w, src + off, nb);
        hipLaunchKernelGGL(vitk::upload_kernel, dim3(1), dim3(256), 0, stream, c,
                           reinterpret_cast<unsigned long long*>(d + off), (int)(nb / 8));
    }
    VITK_CHECK_LAUNCH("vitk_upload_args");
    return VITK_OK;
}

extern "C" int vitk_zero_async(void* ptr, size_t pitch, size_t row_bytes, size_t rows, hipStream_t stream) {
    if (row_bytes == 0 || rows == 0) return VITK_OK;
    VITK_CHECK_ARG(ptr && row_bytes % 4 == 0 && (reinterpret_cast<uintptr_t>(ptr) & 3) == 0 && pitch % 4 == 0,
                   "vitk_zero_async: 4-byte granularity");
    const size_t words = row_bytes / 4 * rows;
    size_t blocks = (words + 1023) / 1024;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(vitk::zero_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, (char*)ptr, pitch, row_bytes / 4, rows);
    VITK_CHECK_LAUNCH("vitk_zero_async");
    return VITK_OK;
}
