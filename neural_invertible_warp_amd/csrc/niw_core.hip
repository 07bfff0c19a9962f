// Library bookkeeping: version and the per-thread error string of the C ABI.
#include <stdarg.h>
#include "niw_common.h"

static thread_local char g_err[512] = "";

void niw_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" int niw_version(void) { return 100; }   // 0.1.0

extern "C" const char* niw_last_error_string(void) { return g_err; }
