// lz_common.h -- shared plumbing for the gfx950 kernels: error capture, launch geometry helpers.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "lzzx_nerf_hip.h"

#define LZ_WAVE 64

void lz_set_error(const char* fmt, ...);

static inline hipStream_t lz_st(lz_stream_t s) { return reinterpret_cast<hipStream_t>(s); }

static inline uint32_t lz_div_up(uint64_t a, uint64_t b) { return (uint32_t)((a + b - 1) / b); }

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
