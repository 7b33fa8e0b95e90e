// last-error string + version for libsliders_hip.so
#include <stdarg.h>
#include <stdio.h>
#include "../../include/sliders_hip.h"

static thread_local char g_err[512] = "";

void slh_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char* slh_last_error(void) { return g_err; }
extern "C" int slh_version(void) { return 1; }
