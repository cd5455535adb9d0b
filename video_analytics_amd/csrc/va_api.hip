// Context, error reporting and version of libva_hip.so.
#include "va_internal.h"
#include <cstring>

static thread_local char g_err[512] = "";

void va_set_error(const char* fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char* va_last_error(void) { return g_err; }

extern "C" int va_version(void) { return 3 | (kVaExperiments ? VA_VERSION_EXPERIMENTS : 0); }

extern "C" int va_ctx_create(int device, va_ctx** out)
{
    VA_CHECK_ARG(out != nullptr, "va_ctx_create: out is NULL");
    *out = nullptr;
    int n = 0;
    VA_HIP(hipGetDeviceCount(&n));
    VA_CHECK_ARG(device >= 0 && device < n, "va_ctx_create: device %d out of range (%d visible)", device, n);
    va_device_guard guard(device);  // (the caller's current device is restored on return)
    VA_HIP(guard.err);
    hipDeviceProp_t prop;
    VA_HIP(hipGetDeviceProperties(&prop, device));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
        va_set_error("va_ctx_create: device %d is %s; this library is built for gfx950 (MI355X) only", device, prop.gcnArchName);
        return VA_ERR_HIP;
    }
    va_ctx* c = new va_ctx();
    c->device = device;
    c->n_cu = prop.multiProcessorCount;
    c->prof_on = false;
    c->prof_ref = nullptr;
    c->prof_ms = c->prof_union_ms = c->prof_launches = c->prof_pxiters = c->prof_pxwarps = 0.0;
    for (int i = 0; i < kVaProfLevels; ++i) c->prof_level_ms[i] = c->prof_level_pxiters[i] = c->prof_level_launches[i] = 0.0;
    *out = c;
    return VA_OK;
}

extern "C" void va_ctx_destroy(va_ctx* ctx)
{
    if (!ctx) return;
    for (va_prof_span& s : ctx->prof_spans) {
        (void)hipEventDestroy(s.beg);
        (void)hipEventDestroy(s.end);
    }
    for (va_prof_span& s : ctx->prof_pool) {
        (void)hipEventDestroy(s.beg);
        (void)hipEventDestroy(s.end);
    }
    if (ctx->prof_ref) (void)hipEventDestroy(ctx->prof_ref);
    delete ctx;
}
