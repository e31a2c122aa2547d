// lz_common.h -- shared plumbing for the gfx950 kernels: error capture, launch geometry helpers.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include <atomic>

#include "lzzx_nerf_hip.h"

#define LZ_WAVE 64

void lz_set_error(const char* fmt, ...);

static inline hipStream_t lz_st(lz_stream_t s) { return reinterpret_cast<hipStream_t>(s); }

static inline uint32_t lz_div_up(uint64_t a, uint64_t b) { return (uint32_t)((a + b - 1) / b); }

// Compute units of the CURRENT device -- the one the caller's stream lives on (every entry point launches on the current device).
// Looked up per call and cached per device ordinal: the persistent kernels size their grid, the samples-per-pass rule and the
// two-slot-row switch from it, so a process that drives several GPUs (or two threads with one GPU each) must not share one value.
inline int lz_cu_count() {
    static std::atomic<int> cache[64];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 256;
    int n = cache[dev].load(std::memory_order_relaxed);
    if (n <= 0) {
        if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
        cache[dev].store(n, std::memory_order_relaxed);
    }
    return n;
}

// every launch funnels through here so a bad launch configuration is reported, not swallowed
#define LZ_CHECK_LAUNCH(name)                                                   \
    do {                                                                        \
        hipError_t e__ = hipGetLastError();                                     \
        if (e__ != hipSuccess) {                                                \
            lz_set_error("%s: launch failed: %s", name, hipGetErrorString(e__)); \
            return (int)e__;                                                    \
        }                                                                       \
    } while (0)

#define LZ_REQUIRE(cond, code, ...)   \
    do {                              \
        if (!(cond)) {                \
            lz_set_error(__VA_ARGS__); \
            return (code);            \
        }                             \
    } while (0)
