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
