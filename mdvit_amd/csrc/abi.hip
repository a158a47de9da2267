// Error plumbing and version of libmdvit_hip.so.
#include <stdarg.h>

#include "common.h"

thread_local char g_mdvit_err[512] = {0};

int mdvit_set_error(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_mdvit_err, sizeof(g_mdvit_err), fmt, ap);
    va_end(ap);
    return code;
}

extern "C" const char* mdvit_last_error(void) { return g_mdvit_err; }
extern "C" int mdvit_version(void) { return MDVIT_ABI_VERSION; }
