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
extern "C" int anyv2v_version(void) { return ANYV2V_ABI_VERSION; }

// Batch hint: launch heuristics (kernel family, split-K factor, GroupNorm chunking) are functions of the row count.  A step that runs
// a SUBSET of the branches of another step (the PnP edit's [negative, editing] steps vs its three-branch steps) must make the same
// choices, or its fp32 summation orders -- and with them single fp16 results -- differ.  With a hint num / den the heuristics see
// rows * num / den; grids and bounds always use the true row count.  Host-side state, read at launch (= capture) time.
static int g_hint_num = 1, g_hint_den = 1;
extern "C" int anyv2v_set_batch_hint(int32_t num, int32_t den) {
    AV_CHECK(num > 0 && den > 0 && num <= 64 && den <= 64, "batch hint: need 0 < num, den <= 64");
    g_hint_num = num;
    g_hint_den = den;
    return ANYV2V_OK;
}
int av_hint_rows(int rows) { return (int)(((long long)rows * g_hint_num) / g_hint_den); }
