#include <stdarg.h>
#include <stdio.h>

#include "../../include/ralf_hip.h"

namespace ralf {
static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
}  // namespace ralf

extern "C" const char* ralf_last_error(void) { return ralf::g_err; }
extern "C" int ralf_abi_version(void) { return RALF_ABI_VERSION; }
