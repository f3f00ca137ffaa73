// api.cpp — error plumbing shared by every entry point of libsgnrast.so (host only).
#include <stdarg.h>
#include <stdio.h>

#include "sgn_rast.h"

static thread_local char g_err[512] = "";

void sgn_set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" __attribute__((visibility("default"))) const char *sgn_last_error(void) { return g_err; }
extern "C" __attribute__((visibility("default"))) int sgn_version(void) { return 100; }
