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

namespace m355 {
static thread_local const char *g_kernel = "";
void note_kernel(const char *name) { g_kernel = name; }
}  // namespace m355

extern "C" const char *m355_last_error(void) { return m355::g_err; }
/* kernel family the calling thread's last conv2d entry point dispatched to (what rocprofv3 will list) */
extern "C" const char *m355_last_kernel(void) { return m355::g_kernel; }
extern "C" int m355_abi_version(void) { return 4; }
/* bytes of an ACTIVATION element the conv / elementwise entry points of this build read and write: 2 = bf16 (the product), 4 = fp32
 * (the EXACT build, -DM355_EXACT: csrc/conv_exact.hip) */
extern "C" int m355_act_bytes(void)
{
#ifdef M355_EXACT
    return 4;
#else
    return 2;
#endif
}
