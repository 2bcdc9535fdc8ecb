// Library-wide runtime bits of the C-ABI: version, per-thread error text.
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

extern "C" int vitk_version(void) { return 1; }

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
