// error plumbing + version for libanyv2v_hip.so
#include "common.h"
#include <stdarg.h>

static thread_local char g_err[512] = "";

void anyv2v_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char* anyv2v_last_error(void) { return g_err; }
extern "C" int anyv2v_version(void) { return 100; }
