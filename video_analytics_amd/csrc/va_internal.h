// Internal definitions shared by the translation units of libva_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdarg>
#include <cstdio>
#include <vector>
#include "../../include/va.h"

struct va_prof_span {
    hipEvent_t beg, end;
    int level;  // pyramid level the bracketed inner-iteration launches belong to
};
constexpr int kVaProfLevels = 16;

struct va_ctx {
    int device;
    int n_cu;
    // TV-L1 measurement hooks (va_tvl1_profile_*)
    bool prof_on;
    std::vector<va_prof_span> prof_spans;  // recorded, not yet read
    std::vector<va_prof_span> prof_pool;   // reusable events
    hipEvent_t prof_ref;                   // time origin for the union of spans (several streams)
    double prof_ms, prof_union_ms, prof_launches, prof_pxiters, prof_pxwarps;
    double prof_level_ms[kVaProfLevels], prof_level_pxiters[kVaProfLevels], prof_level_launches[kVaProfLevels];
};

void va_set_error(const char* fmt, ...);

#define VA_CHECK_ARG(cond, ...)          \
    do {                                 \
        if (!(cond)) {                   \
            va_set_error(__VA_ARGS__);   \
            return VA_ERR_INVALID;       \
        }                                \
    } while (0)

#define VA_HIP(call)                                                                     \
    do {                                                                                 \
        hipError_t e_ = (call);                                                          \
        if (e_ != hipSuccess) {                                                          \
            va_set_error("%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__); \
            return VA_ERR_HIP;                                                           \
        }                                                                                \
    } while (0)

#define VA_LAUNCH_CHECK() VA_HIP(hipGetLastError())

// Every entry point runs on its context's device whatever the calling thread's current device is (launches,
// events and allocations follow hipSetDevice; the caller's stream and pointers must belong to that device).
#define VA_USE_DEVICE(ctx_) VA_HIP(hipSetDevice((ctx_)->device))

static inline size_t va_align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }
static inline int va_cdiv(int a, int b) { return (a + b - 1) / b; }
