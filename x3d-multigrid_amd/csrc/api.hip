// Error reporting and version of the C ABI (include/x3dhip.h).
#include "common.h"
#include <string.h>

static thread_local char g_err[512] = "";

void x3d_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char* x3d_last_error(void) { return g_err; }
extern "C" int x3d_abi_version(void) { return X3D_ABI_VERSION; }
