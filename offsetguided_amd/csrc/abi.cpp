// Error channel and version of libog_decoder.so (see include/og_decoder.h).
#include <stdarg.h>
#include <stdio.h>

#include "og_common.h"

static thread_local char g_err[512] = "";

void og_set_error(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

OG_API int og_abi_version(void) { return OG_ABI_VERSION; }

OG_API const char *og_last_error(void) { return g_err; }

OG_API int og_device_count(void)
{
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) {
        og_set_error("og_device_count: %s", hipGetErrorString(e));
        return OG_EHIP;
    }
    return n;
}
