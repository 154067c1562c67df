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

// ---- "warm the next layer's weights" hint (og_common.h) ----
static thread_local OgWarm g_warm = {nullptr, 0};

OG_API void og_conv_next_weights_hint(const void *w_next, size_t bytes)
{
    g_warm.ptr = w_next;
    g_warm.bytes = bytes > 0xffffff80u ? 0xffffff80u : (unsigned)bytes;
}

OgWarm og_take_warm_hint()
{
    const OgWarm w = g_warm;
    g_warm = OgWarm{nullptr, 0};
    return w;
}

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
