// Error plumbing shared by every entry point of libdtt_hip.so.
#include <stdarg.h>
#include <stdio.h>

static thread_local char g_err[512] = "";

extern "C" const char* dtt_last_error() { return g_err; }
extern "C" int dtt_version() { return 1; }

void dtt_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
