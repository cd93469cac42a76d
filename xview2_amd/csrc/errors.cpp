// Thread-local error string of the C ABI (include/xv2.h: xv2_last_error).
#include <stdarg.h>
#include <stdio.h>
#include "../../include/xv2.h"
namespace xv2 {
static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
}  // namespace xv2
extern "C" const char* xv2_last_error(void) { return xv2::g_err; }
extern "C" int xv2_version(void) { return 1; }
