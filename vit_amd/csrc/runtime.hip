// Library-wide runtime bits of the C-ABI: version, per-thread error text.
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
}  // namespace vitk

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
