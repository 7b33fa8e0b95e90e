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

// ---- kernel-name query (slh_gemm_kernel_name) ----------------------------------------------------------------------------------
// While a sink is set on this thread the GEMM launchers (common.h: slh_launch<Kern>) record the instantiation they WOULD launch
// instead of launching it: the name is formatted at the launch site from the template arguments themselves, in rocprofv3's spelling.
#include <string.h>
static thread_local char* g_name_sink = nullptr;
static thread_local int g_name_cap = 0;

bool slh_name_mode() { return g_name_sink != nullptr; }
void slh_name_sink_set(char* buf, int cap) { g_name_sink = buf; g_name_cap = cap; if (buf && cap > 0) buf[0] = 0; }

void slh_name_record(const char* fmt, ...) {
    if (!g_name_sink || g_name_cap <= 0) return;
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_name_sink, g_name_cap, fmt, ap);
    va_end(ap);
}
