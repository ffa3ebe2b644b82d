// Error reporting and version for libr4r_hip.so (see include/r4r.h).
#include <stdarg.h>
#include <stdio.h>

#include "../../include/r4r.h"

namespace r4r {
static thread_local char g_err[512] = "";

void set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
}  // namespace r4r

extern "C" int r4r_version(void) { return 1; }
extern "C" const char *r4r_last_error(void) { return r4r::g_err; }
