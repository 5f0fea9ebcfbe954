// ABI bookkeeping: version symbol and the thread-local error string (include/fpcdr.h).
#include <stdarg.h>

#include "common.h"

static thread_local char g_err[512] = "";

void fpcdr_set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" int fpcdr_abi_version(void) { return FPCDR_ABI_VERSION; }
extern "C" const char *fpcdr_last_error(void) { return g_err; }
