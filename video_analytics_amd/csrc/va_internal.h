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
// events and allocations follow hipSetDevice; the caller's stream and pointers must belong to that device), and hands
// the thread back with the device it came with: a process that holds several devices (torch) keeps its current device.
struct va_device_guard {
    int prev = -1, want = -1;
    hipError_t err = hipSuccess;
    explicit va_device_guard(int device) : want(device)
    {
        err = hipGetDevice(&prev);
        if (err == hipSuccess && prev != want) err = hipSetDevice(want);
    }
    ~va_device_guard()
    {
        if (prev >= 0 && prev != want) (void)hipSetDevice(prev);
    }
    va_device_guard(const va_device_guard&) = delete;
    va_device_guard& operator=(const va_device_guard&) = delete;
};
#define VA_USE_DEVICE(ctx_)                      \
    va_device_guard va_dev_guard_((ctx_)->device); \
    VA_HIP(va_dev_guard_.err)

// va_tvl1_params.tuning[]: the library's own switches (include/va.h keeps them anonymous)
enum {
    VA_TUNE_STREAM_LEVELS = 0,  // -1: the library decides per level; else bit s = pyramid level s iterates with the row pipeline
    VA_TUNE_STREAM_WAVES = 1,   // 0: default shape per level (four waves x 4 or 5 levels where they fit); 1: one-wave pipeline everywhere;
                                // 2: the two-wave form (2 x 8 levels); 7 / 8 / 9: four waves x 4 / 5 / 3 levels wherever they fit;
                                // (VA_EXPERIMENTS) 3: one deep wave, 4: four jobs per workgroup, 5 / 6: two interleaved chains of levels
                                // per wave (one deep wave / two waves), 10 ... 12: 3 x 5, 3 x 6, 4 x 6 levels
    VA_TUNE_STREAM_CHUNKS = 2,  // 0: rows cut into as many chunks as fill the GPU; n > 0: n chunks (capped at h / 32)
    VA_TUNE_STREAM_SLOTS = 3,   // 0: default target number of strip x chunk x pair jobs per call
    VA_TUNE_ROWS_LEVELS = 4,    // (VA_EXPERIMENTS) -1 / bit set: levels iterated by the persistent row pipeline k_iter_rows
    VA_TUNE_STREAM_PPL = 5,     // 0 / 2: two pixels per lane; 3 (VA_EXPERIMENTS): 192-column strips
    VA_TUNE_STREAM_QUEUE = 6,   // (VA_EXPERIMENTS) 1: all passes of a warp step in one launch (k_iter_stream_q)
    VA_TUNE_ROWS_CFG = 7,       // (VA_EXPERIMENTS) k_iter_rows shape: waves * 16 + levels per wave
};
#ifdef VA_EXPERIMENTS
constexpr bool kVaExperiments = true;
#else
constexpr bool kVaExperiments = false;
#endif

static inline size_t va_align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }
static inline int va_cdiv(int a, int b) { return (a + b - 1) / b; }
