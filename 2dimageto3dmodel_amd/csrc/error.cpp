// Thread-local last-error string of the C-ABI (include/m355.h).
#include <stdarg.h>
#include <stdio.h>

#include "../../include/m355.h"

namespace m355 {
static thread_local char g_err[512] = "";

void set_error(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
}  // namespace m355

extern "C" const char *m355_last_error(void) { return m355::g_err; }
extern "C" int m355_abi_version(void) { return 1; }
