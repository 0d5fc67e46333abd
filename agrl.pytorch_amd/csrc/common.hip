// Error plumbing + version of libagrl_hip.so.
#include "agrl_common.h"

static thread_local char g_err[512] = "";

void agrl_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" int agrl_version(void) { return 100; }
extern "C" const char* agrl_last_error(void) { return g_err; }
