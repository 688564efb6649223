// Version + error plumbing of the C ABI (include/afhip.h).
#include "common.h"
#include <stdarg.h>
#include <stdio.h>

static thread_local char g_err[512] = "";

void afhip_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" int afhip_version(void) { return 100; }
extern "C" const char* afhip_last_error(void) { return g_err; }
