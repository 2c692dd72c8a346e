// Error reporting shared by every entry point: thread-local last-error string, never throws across the ABI.
#include "common.h"
#include <stdarg.h>
#include <stdio.h>

static thread_local char g_err[512] = "";

void rgqa_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
int rgqa_check_hip(hipError_t e, const char* what) {
    if (e == hipSuccess) return RGQA_OK;
    rgqa_set_error("HIP error %d (%s) at %s", (int)e, hipGetErrorString(e), what);
    return RGQA_ERR_HIP;
}
extern "C" const char* rgqa_last_error_string(void) { return g_err; }
