// Error reporting / version / device probe for libctxnerf.so.
#include "common.h"
#include <string.h>

static thread_local char g_err[512] = "";

void ctx_set_error(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char *ctx_last_error(void) { return g_err; }
extern "C" int32_t ctx_version(void) { return 100; }

extern "C" int32_t ctx_device_check(void)
{
    int dev = -1;
    if (hipGetDevice(&dev) != hipSuccess) {
        ctx_set_error("no HIP device");
        return CTX_E_STATE;
    }
    hipDeviceProp_t p;
    if (hipGetDeviceProperties(&p, dev) != hipSuccess) {
        ctx_set_error("hipGetDeviceProperties failed");
        return CTX_E_STATE;
    }
    if (strncmp(p.gcnArchName, "gfx950", 6) != 0) {
        ctx_set_error("device %d is %s, this library is built for gfx950 only", dev, p.gcnArchName);
        return CTX_E_STATE;
    }
    return CTX_OK;
}
