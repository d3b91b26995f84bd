#include "sdfr_common.h"
#include <stdarg.h>
#include <string.h>

static thread_local char g_err[512] = "";

void sdfr_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char* sdfr_last_error(void) { return g_err; }
extern "C" int sdfr_version(void) { return 200; }
