// Error reporting / version / device probe for libctxnerf.so.
#include "common.h"
#include <string.h>
#include <vector>

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

// ---- per-kernel-class timing with dispatch-tight HIP events (bench.py's live roofline measurement) -------
struct ProfRec { hipEvent_t a, b; int klass; };
static std::vector<ProfRec> g_prof;
static bool g_prof_on = false;

bool ctx_prof_on(void) { return g_prof_on; }
void ctx_prof_events(int klass, hipEvent_t *a, hipEvent_t *b)
{
    ProfRec r; r.klass = klass;
    (void)hipEventCreate(&r.a); (void)hipEventCreate(&r.b);
    g_prof.push_back(r);
    *a = r.a; *b = r.b;
}

extern "C" int32_t ctx_profile_begin(void)
{
    for (auto &r : g_prof) { (void)hipEventDestroy(r.a); (void)hipEventDestroy(r.b); }
    g_prof.clear();
    g_prof_on = true;
    return CTX_OK;
}

extern "C" int32_t ctx_profile_end(int32_t klass, double *total_ms, int64_t *count)
{
    CTX_REQUIRE(total_ms && count, "profile_end: null pointer");
    g_prof_on = false;
    double t = 0; int64_t n = 0;
    for (auto &r : g_prof) {
        if (r.klass != klass) continue;
        if (hipEventSynchronize(r.b) != hipSuccess) continue;
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, r.a, r.b) == hipSuccess) { t += ms; ++n; }
    }
    *total_ms = t; *count = n;
    return CTX_OK;
}
