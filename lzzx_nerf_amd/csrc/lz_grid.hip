// lz_grid.hip -- multi-resolution hash / tiled grid encoder for gfx950.
//
// Replaces the reference's gridencoder back-end (gridencoder/src/gridencoder.cu:75-342, entry points
// :424-479).  What is kept: the arithmetic (so table indices are bit-exact and values match the CPU
// checker bit for bit): per-level scale/resolution, pos = fma(x, scale, 0.5), corner weights in the
// same multiplication order, fma accumulation over corners.  What is not: the launch shape.
//
// MI355X design notes
//   * Per-level constants (scale, resolution) are computed once on the host and travel in kernel
//     arguments (SGPRs), not recomputed with exp2f per thread.
//   * Two thread->work mappings, picked by the output layout:
//       level-major   (out_layout 0, [L,B,C]): blockIdx.y = level, one lane per sample.  A wave touches
//                     one level's table only, so its 64 x 2^D gathers land in one small address window
//                     (dense levels: a few hundred bytes; hashed levels: one 64 KB table that sits in L2).
//       sample-major  (out_layout 1, [B,L*C]): one lane per (sample, level), level fastest.  Stores are
//                     perfectly coalesced in the layout the MLP wants, and the permute+reshape copy of
//                     grid.py:52 disappears (that copy is 2 x 48 B/sample/plane of pure HBM traffic).
//   * 256-thread blocks (4 waves), grids >> 256 workgroups; no LDS: the working set is the table, which
//     L2 (4 MB/XCD) holds entirely for the triplane (654 KB/plane).
//   * f16 tables follow at::Half semantics (round to half after every operation), see oracle/grid_oracle.c.
#include "lz_common.h"
#include "lzzx_detmath.h"
#include <hip/hip_fp16.h>
#include <math.h>
#include <stdlib.h>
#include <type_traits>

#define LZ_MAX_LEVELS 32

struct LzGridLevels {
    float scale[LZ_MAX_LEVELS];
    uint32_t res[LZ_MAX_LEVELS];
};

static int lz_fill_levels(LzGridLevels& lv, uint32_t L, float S, uint32_t H) {
    if (L > LZ_MAX_LEVELS) return -1;
    for (uint32_t l = 0; l < L; l++) {
        // gridencoder.cu:125-126, evaluated on the host with the same libm call the CPU checker uses
        const float sc = exp2f((float)l * S) * (float)H - 1.0f;
        lv.scale[l] = sc;
        lv.res[l] = (uint32_t)ceilf(sc) + 1u;
    }
    return 0;
}

template <uint32_t D>
__device__ __forceinline__ uint32_t lz_grid_index(uint32_t C, uint32_t gridtype, bool align_corners, uint32_t hashmap_size,
                                                  uint32_t resolution, const uint32_t (&pos_grid)[D]) {
    constexpr uint32_t primes[7] = {1u, 2654435761u, 805459861u, 3674653429u, 2097192037u, 1434869437u, 2165219737u};
    uint32_t stride = 1, index = 0;
#pragma unroll
    for (uint32_t d = 0; d < D; d++) {
        if (stride <= hashmap_size) {
            index += pos_grid[d] * stride;
            stride *= align_corners ? resolution : (resolution + 1);
        }
    }
    if (gridtype == 0 && stride > hashmap_size) {
        uint32_t h = 0;
#pragma unroll
        for (uint32_t d = 0; d < D; d++) h ^= pos_grid[d] * primes[d];
        index = h;
    }
    return (index % hashmap_size) * C;
}

template <typename T> struct LzElem;
template <> struct LzElem<float> {
    static __device__ __forceinline__ float ld(const float* p) { return *p; }
    static __device__ __forceinline__ float acc(float r, float w, float g) { return lz_fmaf(w, g, r); }
    static __device__ __forceinline__ float accd(float r, float w, float gr, float gl) { return lz_fmaf(w, gr - gl, r); }
    static __device__ __forceinline__ float st(float v) { return v; }
};
template <> struct LzElem<__half> {
    static __device__ __forceinline__ float rh(float v) { return __half2float(__float2half_rn(v)); }
    static __device__ __forceinline__ float ld(const __half* p) { return __half2float(*p); }
    // at::Half semantics: the f32 product is rounded to f32 FIRST, then to half.  The opaque asm keeps the compiler from
    // selecting v_fma_mixlo_f16 for cvt(mul(cvt(g), w)), which rounds the exact product once and differs in ~2^-13 of cases.
    static __device__ __forceinline__ float mul32(float a, float b) { float p = a * b; asm("" : "+v"(p)); return p; }
    static __device__ __forceinline__ float sum32(float a, float b) { float p = a + b; asm("" : "+v"(p)); return p; }
    static __device__ __forceinline__ float acc(float r, float w, float g) { return rh(sum32(r, rh(mul32(w, g)))); }
    static __device__ __forceinline__ float accd(float r, float w, float gr, float gl) { return rh(sum32(r, rh(mul32(w, rh(sum32(gr, -gl)))))); }
    static __device__ __forceinline__ __half st(float v) { return __float2half_rn(v); }
};

template <typename T, uint32_t D, uint32_t C>
__global__ void __launch_bounds__(256)
lz_k_grid_forward(const float* __restrict__ inputs, const T* __restrict__ grid, const int* __restrict__ offsets,
                  T* __restrict__ outputs, uint32_t B, uint32_t L, LzGridLevels lv, T* __restrict__ dy_dx,
                  uint32_t gridtype, bool align_corners, bool sample_major) {
    uint32_t b, level;
    if (sample_major) {
        const uint64_t gid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
        if (gid >= (uint64_t)B * L) return;
        b = (uint32_t)(gid / L);
        level = (uint32_t)(gid - (uint64_t)b * L);
    } else {
        b = blockIdx.x * blockDim.x + threadIdx.x;
        level = blockIdx.y;
        if (b >= B) return;
    }
    const size_t oidx = sample_major ? ((size_t)b * L + level) * C : ((size_t)level * B + b) * C;
    T* out = outputs + oidx;

    float x[D];
    bool oob = false;
#pragma unroll
    for (uint32_t d = 0; d < D; d++) {
        x[d] = inputs[(size_t)b * D + d];
        if (x[d] < 0 || x[d] > 1) oob = true;
    }
    if (oob) {  // gridencoder.cu:98-122
#pragma unroll
        for (uint32_t ch = 0; ch < C; ch++) out[ch] = LzElem<T>::st(0.0f);
        if (dy_dx) {
            T* dd = dy_dx + (size_t)b * D * L * C + (size_t)level * D * C;
#pragma unroll
            for (uint32_t i = 0; i < D * C; i++) dd[i] = LzElem<T>::st(0.0f);
        }
        return;
    }
    const uint32_t off0 = (uint32_t)offsets[level];
    const uint32_t hashmap_size = (uint32_t)offsets[level + 1] - off0;
    const float scale = lv.scale[level];
    const uint32_t resolution = lv.res[level];
    const T* g = grid + (size_t)off0 * C;

    float pos[D];
    uint32_t pg[D];
#pragma unroll
    for (uint32_t d = 0; d < D; d++) {
        pos[d] = lz_fmaf(x[d], scale, align_corners ? 0.0f : 0.5f);
        pg[d] = (uint32_t)floorf(pos[d]);
        pos[d] -= (float)pg[d];
    }
    float res[C];
#pragma unroll
    for (uint32_t ch = 0; ch < C; ch++) res[ch] = 0.0f;
#pragma unroll
    for (uint32_t idx = 0; idx < (1u << D); idx++) {
        float w = 1.0f;
        uint32_t pl[D];
#pragma unroll
        for (uint32_t d = 0; d < D; d++) {
            if ((idx & (1u << d)) == 0) { w *= 1 - pos[d]; pl[d] = pg[d]; }
            else { w *= pos[d]; pl[d] = pg[d] + 1; }
        }
        const uint32_t index = lz_grid_index<D>(C, gridtype, align_corners, hashmap_size, resolution, pl);
#pragma unroll
        for (uint32_t ch = 0; ch < C; ch++) res[ch] = LzElem<T>::acc(res[ch], w, LzElem<T>::ld(g + index + ch));
    }
#pragma unroll
    for (uint32_t ch = 0; ch < C; ch++) out[ch] = LzElem<T>::st(res[ch]);

    if (dy_dx) {  // gridencoder.cu:179-222, layout [B, L, D, C]
        T* dd = dy_dx + (size_t)b * D * L * C + (size_t)level * D * C;
#pragma unroll
        for (uint32_t gd = 0; gd < D; gd++) {
            float rg[C];
#pragma unroll
            for (uint32_t ch = 0; ch < C; ch++) rg[ch] = 0.0f;
#pragma unroll
            for (uint32_t idx = 0; idx < (1u << (D - 1)); idx++) {
                float w = scale;
                uint32_t pl[D];
#pragma unroll
                for (uint32_t nd = 0; nd < D - 1; nd++) {
                    const uint32_t d = (nd >= gd) ? (nd + 1) : nd;
                    if ((idx & (1u << nd)) == 0) { w *= 1 - pos[d]; pl[d] = pg[d]; }
                    else { w *= pos[d]; pl[d] = pg[d] + 1; }
                }
                pl[gd] = pg[gd];
                const uint32_t il = lz_grid_index<D>(C, gridtype, align_corners, hashmap_size, resolution, pl);
                pl[gd] = pg[gd] + 1;
                const uint32_t ir = lz_grid_index<D>(C, gridtype, align_corners, hashmap_size, resolution, pl);
#pragma unroll
                for (uint32_t ch = 0; ch < C; ch++)
                    rg[ch] = LzElem<T>::accd(rg[ch], w, LzElem<T>::ld(g + ir + ch), LzElem<T>::ld(g + il + ch));
            }
#pragma unroll
            for (uint32_t ch = 0; ch < C; ch++) dd[gd * C + ch] = LzElem<T>::st(rg[ch]);
        }
    }
}

// Sample-major forward, the hot variant ([B, L*C] output, no dy_dx): 2-D workgroup (x = level, y = sample) so that
// neither the lane -> (sample, level) map nor the output address needs an integer division, per-level constants
// staged once per workgroup in LDS (each lane reads its own level's record), `% size` replaced by a mask / skipped
// where that is exact: dense levels have index < size by construction, hashed levels have size = 2^T (grid.py:116).
// A wave-uniform test falls back to the generic modulo when some level of the table does not satisfy this.
template <typename T, uint32_t D, uint32_t C>
__global__ void __launch_bounds__(256)
lz_k_grid_forward_sm(const float* __restrict__ inputs, const T* __restrict__ grid, const int* __restrict__ offsets,
                     T* __restrict__ outputs, uint32_t B, uint32_t L, LzGridLevels lv, uint32_t gridtype, bool align_corners) {
    __shared__ float s_scale[LZ_MAX_LEVELS];
    __shared__ uint32_t s_res[LZ_MAX_LEVELS], s_off[LZ_MAX_LEVELS], s_hs[LZ_MAX_LEVELS], s_mode[LZ_MAX_LEVELS];
    const uint32_t tid = threadIdx.y * blockDim.x + threadIdx.x;
    if (tid < L) {
        const uint32_t off0 = (uint32_t)offsets[tid], hs = (uint32_t)offsets[tid + 1] - off0;
        const uint32_t res = lv.res[tid];
        // replay the stride loop of get_grid_index (gridencoder.cu:56-69) once per level
        uint32_t stride = 1;
        uint64_t stride_exact = 1;   // the same product without 32-bit wrap-around
        for (uint32_t d = 0; d < D; d++)
            if (stride <= hs) {
                stride *= align_corners ? res : (res + 1);
                stride_exact *= align_corners ? res : (res + 1);
            }
        const bool wrapped = stride_exact != (uint64_t)stride;   // very fine levels: the reference's uint32 stride wraps; keep
                                                                  // its exact (index % size) behaviour via the generic path
        const bool hashed = gridtype == 0 && stride > hs;
        const bool dense = stride <= hs && !wrapped && !align_corners;   // every stride fitted: index < hs, modulo is the identity --
                                                                  // not with align_corners: side = res, and the +1 corner of x = 1
                                                                  // lands one stride past the level (gridencoder.cu:71 wraps it)
        const bool pow2 = (hs & (hs - 1u)) == 0u;
        s_scale[tid] = lv.scale[tid];
        s_res[tid] = res; s_off[tid] = off0; s_hs[tid] = hs;
        // 0 none, 1 mask, 2 modulo; bit 2: hash
        s_mode[tid] = dense ? 0u : ((pow2 && !wrapped) ? 1u : 2u) | (hashed ? 4u : 0u);
    }
    __syncthreads();
    const uint32_t level = threadIdx.x;
    const uint32_t b = blockIdx.x * blockDim.y + threadIdx.y;
    if (b >= B) return;
    T* out = outputs + ((size_t)b * L + level) * C;
    float x[D];
    bool oob = false;
#pragma unroll
    for (uint32_t d = 0; d < D; d++) {
        x[d] = inputs[(size_t)b * D + d];
        if (x[d] < 0 || x[d] > 1) oob = true;
    }
    const uint32_t hashmap_size = s_hs[level], resolution = s_res[level], mode = s_mode[level];
    const float scale = s_scale[level];
    const T* g = grid + (size_t)s_off[level] * C;
    float pos[D];
    uint32_t pg[D];
#pragma unroll
    for (uint32_t d = 0; d < D; d++) {
        const float xc = lz_fminf(lz_fmaxf(x[d], 0.0f), 1.0f);   // clamp for addressing only; result zeroed below
        pos[d] = lz_fmaf(xc, scale, align_corners ? 0.0f : 0.5f);
        pg[d] = (uint32_t)floorf(pos[d]);
        pos[d] -= (float)pg[d];
    }
    float res[C];
#pragma unroll
    for (uint32_t ch = 0; ch < C; ch++) res[ch] = 0.0f;
    const bool slow = __any((mode & 3u) == 2u);
    auto corners = [&](auto slow_tag) {
        constexpr bool SLOW = decltype(slow_tag)::value;
#pragma unroll
        for (uint32_t idx = 0; idx < (1u << D); idx++) {
            float w = 1.0f;
            uint32_t pl[D];
#pragma unroll
            for (uint32_t d = 0; d < D; d++) {
                if ((idx & (1u << d)) == 0) { w *= 1 - pos[d]; pl[d] = pg[d]; }
                else { w *= pos[d]; pl[d] = pg[d] + 1; }
            }
            uint32_t index;
            if constexpr (SLOW) {
                index = lz_grid_index<D>(C, gridtype, align_corners, hashmap_size, resolution, pl);
            } else {
                constexpr uint32_t primes[7] = {1u, 2654435761u, 805459861u, 3674653429u, 2097192037u, 1434869437u, 2165219737u};
                uint32_t lin = 0, h = 0, stride = 1;
#pragma unroll
                for (uint32_t d = 0; d < D; d++) {
                    lin += pl[d] * stride;            // only used when every stride fitted (mode 0)
                    stride *= align_corners ? resolution : (resolution + 1);
                    h ^= pl[d] * primes[d];
                }
                // mode 0: dense.  mode 1|4: hashed, power-of-two table.  (mode 1 without hash = tiled wrap of a partial
                // sum: routed to the generic path by `slow` below)
                index = ((mode & 4u) ? (h & (hashmap_size - 1u)) : lin) * C;
            }
#pragma unroll
            for (uint32_t ch = 0; ch < C; ch++) res[ch] = LzElem<T>::acc(res[ch], w, LzElem<T>::ld(g + index + ch));
        }
    };
    // generic path when some level needs a true modulo, or wraps a partial (tiled) sum
    if (slow || __any(mode != 0u && (mode & 4u) == 0u)) corners(std::true_type{});
    else corners(std::false_type{});
    if constexpr (C == 2 && sizeof(T) == 4) {
        *reinterpret_cast<float2*>(out) = oob ? make_float2(0.f, 0.f) : make_float2(res[0], res[1]);
    } else {
#pragma unroll
        for (uint32_t ch = 0; ch < C; ch++) out[ch] = LzElem<T>::st(oob ? 0.0f : res[ch]);
    }
}

template <uint32_t D>
__global__ void __launch_bounds__(256)
lz_k_grid_corner_indices(const float* __restrict__ inputs, const int* __restrict__ offsets, int* __restrict__ out,
                         uint32_t B, uint32_t C, uint32_t L, LzGridLevels lv, uint32_t gridtype, bool align_corners) {
    const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t level = blockIdx.y;
    if (b >= B) return;
    int* o = out + ((size_t)level * B + b) * (1u << D);
    float x[D];
    bool oob = false;
#pragma unroll
    for (uint32_t d = 0; d < D; d++) {
        x[d] = inputs[(size_t)b * D + d];
        if (x[d] < 0 || x[d] > 1) oob = true;
    }
    if (oob) {
#pragma unroll
        for (uint32_t i = 0; i < (1u << D); i++) o[i] = -1;
        return;
    }
    const uint32_t off0 = (uint32_t)offsets[level];
    const uint32_t hashmap_size = (uint32_t)offsets[level + 1] - off0;
    uint32_t pg[D];
#pragma unroll
    for (uint32_t d = 0; d < D; d++) pg[d] = (uint32_t)floorf(lz_fmaf(x[d], lv.scale[level], align_corners ? 0.0f : 0.5f));
#pragma unroll
    for (uint32_t idx = 0; idx < (1u << D); idx++) {
        uint32_t pl[D];
#pragma unroll
        for (uint32_t d = 0; d < D; d++) pl[d] = pg[d] + ((idx >> d) & 1u);
        o[idx] = (int)(off0 * C + lz_grid_index<D>(C, gridtype, align_corners, hashmap_size, lv.res[level], pl));
    }
}

// ---- backward: scatter-add of w * grad into the table (gridencoder.cu:226-313) ----
__device__ __forceinline__ void lz_atomic_add(float* p, float v) { atomicAdd(p, v); }

template <typename T, uint32_t D, uint32_t C>
__global__ void __launch_bounds__(256)
lz_k_grid_backward(const T* __restrict__ grad, const float* __restrict__ inputs, const int* __restrict__ offsets,
                   T* __restrict__ grad_grid, uint32_t B, uint32_t L, LzGridLevels lv, uint32_t gridtype,
                   bool align_corners, bool sample_major) {
    uint32_t b, level;
    if (sample_major) {
        const uint64_t gid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
        if (gid >= (uint64_t)B * L) return;
        b = (uint32_t)(gid / L);
        level = (uint32_t)(gid - (uint64_t)b * L);
    } else {
        b = blockIdx.x * blockDim.x + threadIdx.x;
        level = blockIdx.y;
        if (b >= B) return;
    }
    float x[D];
#pragma unroll
    for (uint32_t d = 0; d < D; d++) {
        x[d] = inputs[(size_t)b * D + d];
        if (x[d] < 0 || x[d] > 1) return;
    }
    const uint32_t off0 = (uint32_t)offsets[level];
    const uint32_t hashmap_size = (uint32_t)offsets[level + 1] - off0;
    const float scale = lv.scale[level];
    const uint32_t resolution = lv.res[level];
    T* gg = grad_grid + (size_t)off0 * C;
    const T* gsrc = grad + (sample_major ? ((size_t)b * L + level) * C : ((size_t)level * B + b) * C);
    float gcur[C];
#pragma unroll
    for (uint32_t ch = 0; ch < C; ch++) gcur[ch] = LzElem<T>::ld(gsrc + ch);

    float pos[D];
    uint32_t pg[D];
#pragma unroll
    for (uint32_t d = 0; d < D; d++) {
        pos[d] = lz_fmaf(x[d], scale, align_corners ? 0.0f : 0.5f);
        pg[d] = (uint32_t)floorf(pos[d]);
        pos[d] -= (float)pg[d];
    }
#pragma unroll
    for (uint32_t idx = 0; idx < (1u << D); idx++) {
        float w = 1.0f;
        uint32_t pl[D];
#pragma unroll
        for (uint32_t d = 0; d < D; d++) {
            if ((idx & (1u << d)) == 0) { w *= 1 - pos[d]; pl[d] = pg[d]; }
            else { w *= pos[d]; pl[d] = pg[d] + 1; }
        }
        const uint32_t index = lz_grid_index<D>(C, gridtype, align_corners, hashmap_size, resolution, pl);
        if constexpr (sizeof(T) == 4) {
#pragma unroll
            for (uint32_t ch = 0; ch < C; ch++) lz_atomic_add(reinterpret_cast<float*>(gg) + index + ch, w * gcur[ch]);
        } else {
            if constexpr (C % 2 == 0) {
#pragma unroll
                for (uint32_t ch = 0; ch < C; ch += 2) {
                    const __half2 v = __halves2half2(__float2half_rn(LzElem<__half>::mul32(w, gcur[ch])), __float2half_rn(LzElem<__half>::mul32(w, gcur[ch + 1])));
                    unsafeAtomicAdd(reinterpret_cast<__half2*>(gg + index + ch), v);
                }
            } else {
                // C == 1 with half tables never happens through the Python wrapper (grid.py:38); keep it correct anyway
                // with a 32-bit CAS on the containing word.
                __half* addr = reinterpret_cast<__half*>(gg) + index;
                unsigned int* word = reinterpret_cast<unsigned int*>(reinterpret_cast<uintptr_t>(addr) & ~(uintptr_t)3);
                const bool hi = (reinterpret_cast<uintptr_t>(addr) & 2) != 0;
                unsigned int old = *word, assumed;
                do {
                    assumed = old;
                    const unsigned short cur = hi ? (unsigned short)(assumed >> 16) : (unsigned short)(assumed & 0xffffu);
                    const __half nv = __float2half_rn(LzElem<__half>::sum32(__half2float(__ushort_as_half(cur)), __half2float(__float2half_rn(LzElem<__half>::mul32(w, gcur[0])))));
                    const unsigned int nb = __half_as_ushort(nv);
                    const unsigned int repl = hi ? ((assumed & 0xffffu) | (nb << 16)) : ((assumed & 0xffff0000u) | nb);
                    old = atomicCAS(word, assumed, repl);
                } while (old != assumed);
            }
        }
    }
}

// Scatter-add for tables too big for LDS (cfg2: 4 MB per level), f32.  Scattered float atomics retire at the memory side per LINE
// touched (~2e10 lane-adds/s when every lane of an instruction hits its own line, a sixteenth of the contiguous rate).  So the lanes
// are arranged to share lines: lane = (sample, x bit of the corner, channel) -- the x and x+1 corners of a cell are adjacent entries
// on dense levels and sit in the same 128-byte line on hashed ones (prime_x = 1) -- and an instruction covers one (y, z, ...) corner
// combination of 64 / (2 C) samples: 2 C lanes per line instead of one.  Levels are the slow launch dimension (one 4 MB gradient
// level L2-resident at a time).  Same terms as the plain kernel; the order of the atomics is free in both.
template <uint32_t D, uint32_t C>
__global__ void __launch_bounds__(256)
lz_k_grid_backward_xc(const float* __restrict__ grad, const float* __restrict__ inputs, const int* __restrict__ offsets,
                      float* __restrict__ grad_grid, uint32_t B, uint32_t L, LzGridLevels lv, uint32_t gridtype, bool align_corners,
                      bool sample_major) {
    constexpr uint32_t LPS = 2 * C;                     // lanes per sample
    const uint32_t level = blockIdx.y;
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t b = t / LPS, sub = t - b * LPS, xb = sub / C, ch = sub - xb * C;
    if (b >= B) return;
    float x[D];
#pragma unroll
    for (uint32_t d = 0; d < D; d++) {
        x[d] = inputs[(size_t)b * D + d];
        if (x[d] < 0 || x[d] > 1) return;
    }
    const uint32_t off0 = (uint32_t)offsets[level], hs = (uint32_t)offsets[level + 1] - off0;
    const float scale = lv.scale[level];
    const uint32_t resolution = lv.res[level];
    float* gg = grad_grid + (size_t)off0 * C;
    const float g = grad[(sample_major ? ((size_t)b * L + level) * C : ((size_t)level * B + b) * C) + ch];
    float pos[D];
    uint32_t pg[D];
#pragma unroll
    for (uint32_t d = 0; d < D; d++) {
        pos[d] = lz_fmaf(x[d], scale, align_corners ? 0.0f : 0.5f);
        pg[d] = (uint32_t)floorf(pos[d]);
        pos[d] -= (float)pg[d];
    }
#pragma unroll
    for (uint32_t h = 0; h < (1u << (D - 1)); h++) {
        const uint32_t idx = (h << 1) | xb;
        float w = 1.0f;
        uint32_t pl[D];
#pragma unroll
        for (uint32_t d = 0; d < D; d++) {
            if ((idx & (1u << d)) == 0) { w *= 1 - pos[d]; pl[d] = pg[d]; }
            else { w *= pos[d]; pl[d] = pg[d] + 1; }
        }
        const uint32_t index = lz_grid_index<D>(C, gridtype, align_corners, hs, resolution, pl);
        lz_atomic_add(gg + index + ch, w * g);
    }
}

template <typename T, uint32_t D, uint32_t C>
__global__ void __launch_bounds__(256)
lz_k_grid_input_backward(const T* __restrict__ grad, const T* __restrict__ dy_dx, T* __restrict__ grad_inputs, uint32_t B,
                         uint32_t L, bool sample_major) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= B * D) return;
    const uint32_t b = t / D, d = t - b * D;
    const T* dd = dy_dx + (size_t)b * L * D * C;
    float r = 0.0f;
    for (uint32_t l = 0; l < L; l++) {
#pragma unroll
        for (uint32_t ch = 0; ch < C; ch++) {
            const size_t gi = sample_major ? ((size_t)b * L + l) * C + ch : ((size_t)l * B + b) * C + ch;
            const float gv = LzElem<T>::ld(grad + gi), jv = LzElem<T>::ld(dd + (size_t)l * D * C + d * C + ch);
            if constexpr (sizeof(T) == 4) r = lz_fmaf(gv, jv, r);
            else r = LzElem<__half>::rh(LzElem<__half>::sum32(r, LzElem<__half>::rh(LzElem<__half>::mul32(gv, jv))));
        }
    }
    grad_inputs[t] = LzElem<T>::st(r);
}

// ---- level-resident forward: the level's whole table lives in LDS ------------------------------------------------
// A random 4-byte gather costs the texture-address path ~1 lane/clk/CU (measured: 2^24 samples x 48 gathers in 1.8 ms),
// which caps the sample-major kernel at ~2.3 TB/s of algorithmic traffic however well L2 hits.  The triplane tables are
// tiny (<= 16384 entries = 64 KB per level), so here a workgroup owns ONE (level, sample chunk): it copies the level's
// table into LDS once (coalesced 16 B/lane), then streams its chunk: x in (coalesced, re-used by the L workgroups of the
// chunk through L2), 2^D corner reads from LDS (~8 lanes/clk even with bank conflicts), one strided store to [B, L*C].
// blockIdx -> (chunk, level) puts the L workgroups of a chunk on ONE XCD (blocks b, b+8, b+16, ... share an XCD), next
// to each other in dispatch order, so their partial-line stores merge in that XCD's L2 before write-back.
// A level that does not fit LDS falls back to global gathers inside the same kernel (wave-uniform branch).
#define LZ_GRID_LDS_BYTES 65536

// Index terms per dimension for the dense (mode 0) and power-of-two hashed (mode 1) levels: term[d][0] belongs to the lower cell
// coordinate pg[d], term[d][1] to pg[d] + 1 -- the lower one plus a constant modulo 2^32, exactly the reference's uint32 arithmetic
// (gridencoder.cu:60-98: index += pos * stride / result ^= pos * prime).  A corner's index is then the sum (dense) or the xor (hashed)
// of D terms: one 32-bit multiply per dimension and sample, spelled out (the compiler found the same common subexpressions in the
// per-corner form: measured, no change in the triplane plane or the cfg2 gather -- neither is bound by the index arithmetic).
template <uint32_t D>
__device__ __forceinline__ void lz_grid_terms(const uint32_t (&pg)[D], uint32_t mode, uint32_t resolution, bool align_corners, uint32_t (&term)[D][2]) {
    constexpr uint32_t primes[7] = {1u, 2654435761u, 805459861u, 3674653429u, 2097192037u, 1434869437u, 2165219737u};
    uint32_t stride = 1;
#pragma unroll
    for (uint32_t d = 0; d < D; d++) {
        const uint32_t k = mode == 1u ? primes[d] : stride;
        term[d][0] = d == 0 ? pg[d] : pg[d] * k;      // primes[0] == 1 and the first stride is 1
        term[d][1] = term[d][0] + k;
        stride *= align_corners ? resolution : (resolution + 1);
    }
}
template <uint32_t D>
__device__ __forceinline__ uint32_t lz_grid_corner(const uint32_t (&term)[D][2], uint32_t idx, uint32_t mode, uint32_t hs) {
    uint32_t lin = 0, h = 0;
#pragma unroll
    for (uint32_t d = 0; d < D; d++) {
        const uint32_t t = term[d][(idx >> d) & 1u];
        lin += t;
        h ^= t;
    }
    return mode == 1u ? (h & (hs - 1u)) : lin;
}

template <typename T, uint32_t D, uint32_t C, bool IN_LDS>
__device__ __forceinline__ void lz_grid_level_stream(const float* __restrict__ inputs, const T* __restrict__ tab, T* __restrict__ outputs,
                                                     uint32_t b0, uint32_t b1, uint32_t L, uint32_t level, float scale,
                                                     uint32_t resolution, uint32_t hashmap_size, uint32_t mode, uint32_t gridtype,
                                                     bool align_corners) {
    for (uint32_t b = b0 + threadIdx.x; b < b1; b += blockDim.x) {
        float x[D];
        bool oob = false;
#pragma unroll
        for (uint32_t d = 0; d < D; d++) {
            x[d] = inputs[(size_t)b * D + d];
            if (x[d] < 0 || x[d] > 1) oob = true;
        }
        float pos[D];
        uint32_t pg[D];
#pragma unroll
        for (uint32_t d = 0; d < D; d++) {
            const float xc = lz_fminf(lz_fmaxf(x[d], 0.0f), 1.0f);
            pos[d] = lz_fmaf(xc, scale, align_corners ? 0.0f : 0.5f);
            pg[d] = (uint32_t)floorf(pos[d]);
            pos[d] -= (float)pg[d];
        }
        float res[C];
#pragma unroll
        for (uint32_t ch = 0; ch < C; ch++) res[ch] = 0.0f;
        // per-dimension index terms (lz_grid_terms)
        uint32_t term[D][2];
        if (mode != 2u) lz_grid_terms<D>(pg, mode, resolution, align_corners, term);
#pragma unroll
        for (uint32_t idx = 0; idx < (1u << D); idx++) {
            float w = 1.0f;
            uint32_t pl[D];
#pragma unroll
            for (uint32_t d = 0; d < D; d++) {
                if ((idx & (1u << d)) == 0) { w *= 1 - pos[d]; pl[d] = pg[d]; }
                else { w *= pos[d]; pl[d] = pg[d] + 1; }
            }
            uint32_t index;
            if (mode == 2u) index = lz_grid_index<D>(C, gridtype, align_corners, hashmap_size, resolution, pl);   // workgroup-uniform: true modulo / wrapped strides
            else index = lz_grid_corner<D>(term, idx, mode, hashmap_size) * C;
#pragma unroll
            for (uint32_t ch = 0; ch < C; ch++) res[ch] = LzElem<T>::acc(res[ch], w, LzElem<T>::ld(tab + index + ch));
        }
        T* out = outputs + ((size_t)b * L + level) * C;
        if constexpr (C == 2 && sizeof(T) == 4) {
            *reinterpret_cast<float2*>(out) = oob ? make_float2(0.f, 0.f) : make_float2(res[0], res[1]);
        } else {
#pragma unroll
            for (uint32_t ch = 0; ch < C; ch++) out[ch] = LzElem<T>::st(oob ? 0.0f : res[ch]);
        }
    }
}

#ifndef LZ_GRID_LDS_WG
#define LZ_GRID_LDS_WG 512
#endif
#ifndef LZ_GRID_LDS_CHUNK
#define LZ_GRID_LDS_CHUNK 32768
#endif
template <typename T, uint32_t D, uint32_t C>
__global__ void __launch_bounds__(LZ_GRID_LDS_WG)
lz_k_grid_forward_lds(const float* __restrict__ inputs, const T* __restrict__ grid, const int* __restrict__ offsets,
                      T* __restrict__ outputs, uint32_t B, uint32_t L, LzGridLevels lv, uint32_t gridtype, bool align_corners,
                      uint32_t chunk, uint32_t n_chunks) {
    extern __shared__ __align__(16) unsigned char lz_grid_smem[];
    T* tab = reinterpret_cast<T*>(lz_grid_smem);
    const uint32_t r = blockIdx.x & 7u, t = blockIdx.x >> 3;
    const uint32_t level = t % L, g = t / L;
    const uint32_t c = 8u * g + r;
    if (c >= n_chunks) return;
    const uint32_t off0 = (uint32_t)offsets[level], hs = (uint32_t)offsets[level + 1] - off0;
    const uint32_t res = lv.res[level];
    const float scale = lv.scale[level];
    // classify the level exactly like lz_k_grid_forward_sm: 0 dense (index < size), 1 hashed + power-of-two size (mask),
    // 2 anything else (generic modulo path)
    uint32_t stride = 1;
    uint64_t stride_exact = 1;
    for (uint32_t d = 0; d < D; d++)
        if (stride <= hs) {
            stride *= align_corners ? res : (res + 1);
            stride_exact *= align_corners ? res : (res + 1);
        }
    const bool wrapped = stride_exact != (uint64_t)stride;
    const bool hashed = gridtype == 0 && stride > hs;
    const bool dense = stride <= hs && !wrapped && !align_corners;   // align_corners: the +1 corner of x = 1 needs the wrap
    const bool pow2 = (hs & (hs - 1u)) == 0u;
    const uint32_t mode = dense ? 0u : ((hashed && pow2 && !wrapped) ? 1u : 2u);
    const T* gsrc = grid + (size_t)off0 * C;
    const size_t bytes = (size_t)hs * C * sizeof(T);
    const uint32_t b0 = c * chunk, b1 = (b0 + chunk < B) ? b0 + chunk : B;
    if (bytes <= LZ_GRID_LDS_BYTES) {
        // (off0 * C * sizeof(T)) is a multiple of 16: level sizes are multiples of 8 entries (grid.py:117)
        const uint4* s4 = reinterpret_cast<const uint4*>(gsrc);
        uint4* d4 = reinterpret_cast<uint4*>(tab);
        const uint32_t n16 = (reinterpret_cast<uintptr_t>(gsrc) & 15u) ? 0u : (uint32_t)(bytes >> 4);
        for (uint32_t i = threadIdx.x; i < n16; i += blockDim.x) d4[i] = s4[i];
        for (uint32_t i = (n16 << 4) / sizeof(T) + threadIdx.x; i < hs * C; i += blockDim.x) tab[i] = gsrc[i];
        __syncthreads();
        lz_grid_level_stream<T, D, C, true>(inputs, tab, outputs, b0, b1, L, level, scale, res, hs, mode, gridtype, align_corners);
    } else {
        lz_grid_level_stream<T, D, C, false>(inputs, gsrc, outputs, b0, b1, L, level, scale, res, hs, mode, gridtype, align_corners);
    }
}

template <typename T, uint32_t D, uint32_t C>
static void lz_grid_lds_launch(const float* inputs, const T* emb, const int* offsets, T* out, uint32_t B, uint32_t L,
                               const LzGridLevels& lv, uint32_t gridtype, bool ac, hipStream_t st) {
    // chunk: large enough to amortise the table copy (<= 64 KB per chunk per level), small enough for >= ~2 waves of CUs
    uint32_t chunk = LZ_GRID_LDS_CHUNK;
    while (chunk > 2048 && (uint64_t)lz_div_up(B, chunk) * L < 1024) chunk >>= 1;
    const uint32_t n_chunks = lz_div_up(B, chunk);
    const uint32_t groups = lz_div_up(n_chunks, 8);
    hipLaunchKernelGGL((lz_k_grid_forward_lds<T, D, C>), dim3(8u * groups * L), dim3(LZ_GRID_LDS_WG), LZ_GRID_LDS_BYTES, st, inputs, emb, offsets, out, B, L,
                       lv, gridtype, ac, chunk, n_chunks);
}

// ---- big tables (a level does not fit LDS): level-major pass over sample tiles + in-place untile --------------------
// The sample-major kernel above makes every wave touch all L levels at once, so the L2 working set is the whole table
// (49 MB for D3/L16/C2/T19) and most gathers miss to the Infinity Cache.  Here blockIdx.y = level is the SLOW launch
// dimension: at any moment the whole chip gathers from one or two levels (<= 4 MB each, L2-resident), the level's mode
// (dense / mask / modulo) is workgroup-uniform, and a corner is ONE vector load of C elements.  To keep stores whole
// lines without a scratch buffer the pass writes the caller's [B, L*C] buffer in a TILED order
//     [tile][level][sample in tile][C]          (tile = Tn samples; same bytes as the tile's final [sample][level][C])
// and lz_k_grid_untile then transposes every tile in place through LDS (one workgroup owns one tile: load all, barrier,
// store all).  The untile pass costs one read + one write of the output (~0.3 ms per GB), far less than it saves.
template <uint32_t D>
__device__ __forceinline__ uint32_t lz_grid_level_mode(uint32_t hs, uint32_t res, uint32_t gridtype, bool align_corners) {
    uint32_t stride = 1;
    uint64_t stride_exact = 1;
    for (uint32_t d = 0; d < D; d++)
        if (stride <= hs) {
            stride *= align_corners ? res : (res + 1);
            stride_exact *= align_corners ? res : (res + 1);
        }
    const bool wrapped = stride_exact != (uint64_t)stride;
    const bool hashed = gridtype == 0 && stride > hs;
    const bool dense = stride <= hs && !wrapped && !align_corners;   // align_corners: the +1 corner of x = 1 needs the wrap
    const bool pow2 = (hs & (hs - 1u)) == 0u;
    return dense ? 0u : ((hashed && pow2 && !wrapped) ? 1u : 2u);   // 0 identity, 1 hash + mask, 2 generic modulo
}

template <typename T, uint32_t C> struct LzVec {
    T v[C];
};

template <typename T, uint32_t D, uint32_t C>
__global__ void __launch_bounds__(256)
lz_k_grid_forward_lm(const float* __restrict__ inputs, const T* __restrict__ grid, const int* __restrict__ offsets,
                     T* __restrict__ outputs, uint32_t B, uint32_t L, LzGridLevels lv, uint32_t gridtype, bool align_corners) {
    const uint32_t Tn = blockDim.x, tile = blockIdx.x, level = blockIdx.y, t = threadIdx.x;
    const uint32_t b0 = tile * Tn;
    const uint32_t n = (B - b0 < Tn) ? B - b0 : Tn;
    if (t >= n) return;
    const uint32_t b = b0 + t;
    const uint32_t off0 = (uint32_t)offsets[level], hs = (uint32_t)offsets[level + 1] - off0;
    const uint32_t resolution = lv.res[level];
    const float scale = lv.scale[level];
    const uint32_t mode = lz_grid_level_mode<D>(hs, resolution, gridtype, align_corners);
    const T* g = grid + (size_t)off0 * C;
    float x[D];
    bool oob = false;
#pragma unroll
    for (uint32_t d = 0; d < D; d++) {
        x[d] = inputs[(size_t)b * D + d];
        if (x[d] < 0 || x[d] > 1) oob = true;
    }
    float pos[D];
    uint32_t pg[D];
#pragma unroll
    for (uint32_t d = 0; d < D; d++) {
        const float xc = lz_fminf(lz_fmaxf(x[d], 0.0f), 1.0f);   // clamp for addressing only; result zeroed below
        pos[d] = lz_fmaf(xc, scale, align_corners ? 0.0f : 0.5f);
        pg[d] = (uint32_t)floorf(pos[d]);
        pos[d] -= (float)pg[d];
    }
    uint32_t index[1u << D];
    float w[1u << D];
    uint32_t term[D][2];
    if (mode != 2u) lz_grid_terms<D>(pg, mode, resolution, align_corners, term);
#pragma unroll
    for (uint32_t idx = 0; idx < (1u << D); idx++) {
        float wc = 1.0f;
        uint32_t pl[D];
#pragma unroll
        for (uint32_t d = 0; d < D; d++) {
            if ((idx & (1u << d)) == 0) { wc *= 1 - pos[d]; pl[d] = pg[d]; }
            else { wc *= pos[d]; pl[d] = pg[d] + 1; }
        }
        w[idx] = wc;
        if (mode == 2u) index[idx] = lz_grid_index<D>(C, gridtype, align_corners, hs, resolution, pl);   // workgroup-uniform
        else index[idx] = lz_grid_corner<D>(term, idx, mode, hs) * C;
    }
    // all 2^D corner reads in flight before the first use; (index * sizeof(T)) is a multiple of the vector size
    LzVec<T, C> cv[1u << D];
#pragma unroll
    for (uint32_t idx = 0; idx < (1u << D); idx++)
        __builtin_memcpy(&cv[idx], __builtin_assume_aligned(g + index[idx], sizeof(T) * C), sizeof(T) * C);
    float res[C];
#pragma unroll
    for (uint32_t ch = 0; ch < C; ch++) res[ch] = 0.0f;
#pragma unroll
    for (uint32_t idx = 0; idx < (1u << D); idx++)
#pragma unroll
        for (uint32_t ch = 0; ch < C; ch++) res[ch] = LzElem<T>::acc(res[ch], w[idx], LzElem<T>::ld(&cv[idx].v[ch]));
    LzVec<T, C> o;
#pragma unroll
    for (uint32_t ch = 0; ch < C; ch++) o.v[ch] = LzElem<T>::st(oob ? 0.0f : res[ch]);
    T* out = outputs + ((size_t)b0 * L + (size_t)level * n + t) * C;
    __builtin_memcpy(__builtin_assume_aligned(out, sizeof(T) * C), &o, sizeof(T) * C);
}

// The same pass with TWO lanes per sample: lane pair (2t, 2t+1) splits the 2^D corners by their x bit.  The x and x+1 corners of
// a cell sit in the same 128-byte line almost always (dense levels: adjacent entries; hashed levels: prime_x = 1, so the two
// indices differ by x ^ (x+1), within 16 entries 94 % of the time), and the texture-address unit processes one line per clock:
// putting both corners in ONE load instruction halves the lines per instruction (64 -> ~34).  The partner's values come back
// through a DPP lane swap and the even lane runs the fma chain in the reference's corner order, so results stay bit-identical.
template <typename T, uint32_t D, uint32_t C>
__global__ void __launch_bounds__(512)
lz_k_grid_forward_lmp(const float* __restrict__ inputs, const T* __restrict__ grid, const int* __restrict__ offsets,
                      T* __restrict__ outputs, uint32_t B, uint32_t L, LzGridLevels lv, uint32_t gridtype, bool align_corners,
                      const int* __restrict__ count = nullptr, float bound = 0.0f) {
    // count (lz_grid_encode_forward_tiled): rows that hold samples this launch, read on the device (the render loop's n_alive * n_step); tiles
    // behind it do nothing.  bound > 0: inputs arrive in [-bound, bound] and are mapped like GridEncoder.forward, (x + bound) / (2 bound) (grid.py:143)
    constexpr uint32_t NC = 1u << (D - 1), WORDS = sizeof(T) * C / 4;
    static_assert(sizeof(T) * C % 4 == 0, "pair kernel moves whole dwords");
    const uint32_t Tn = blockDim.x >> 1, tile = blockIdx.x, level = blockIdx.y, t = threadIdx.x >> 1, xb = threadIdx.x & 1u;
    const uint32_t b0 = tile * Tn;
    if (count && b0 >= (uint32_t)*count) return;
    const uint32_t n = (B - b0 < Tn) ? B - b0 : Tn;
    if (t >= n) return;   // pair-uniform
    const uint32_t b = b0 + t;
    const uint32_t off0 = (uint32_t)offsets[level], hs = (uint32_t)offsets[level + 1] - off0;
    const uint32_t resolution = lv.res[level];
    const float scale = lv.scale[level];
    const uint32_t mode = lz_grid_level_mode<D>(hs, resolution, gridtype, align_corners);
    const T* g = grid + (size_t)off0 * C;
    float x[D];
    bool oob = false;
#pragma unroll
    for (uint32_t d = 0; d < D; d++) {
        x[d] = inputs[(size_t)b * D + d];
        if (bound > 0.0f) x[d] = (x[d] + bound) / (2.0f * bound);
        if (x[d] < 0 || x[d] > 1) oob = true;
    }
    float pos[D];
    uint32_t pg[D];
#pragma unroll
    for (uint32_t d = 0; d < D; d++) {
        const float xc = lz_fminf(lz_fmaxf(x[d], 0.0f), 1.0f);
        pos[d] = lz_fmaf(xc, scale, align_corners ? 0.0f : 0.5f);
        pg[d] = (uint32_t)floorf(pos[d]);
        pos[d] -= (float)pg[d];
    }
    uint32_t own[NC][WORDS], oth[NC][WORDS];
    uint32_t term[D][2];
    if (mode != 2u) lz_grid_terms<D>(pg, mode, resolution, align_corners, term);
#pragma unroll
    for (uint32_t h = 0; h < NC; h++) {
        const uint32_t idx = (h << 1) | xb;
        uint32_t pl[D];
#pragma unroll
        for (uint32_t d = 0; d < D; d++) pl[d] = pg[d] + ((idx >> d) & 1u);
        uint32_t index;
        if (mode == 2u) index = lz_grid_index<D>(C, gridtype, align_corners, hs, resolution, pl);
        else index = lz_grid_corner<D>(term, idx, mode, hs) * C;
        __builtin_memcpy(own[h], __builtin_assume_aligned(g + index, sizeof(T) * C), sizeof(T) * C);
    }
#pragma unroll
    for (uint32_t h = 0; h < NC; h++)
#pragma unroll
        for (uint32_t k = 0; k < WORDS; k++) oth[h][k] = (uint32_t)__shfl_xor((int)own[h][k], 1, 64);
    if (xb != 0) return;
    float res[C];
#pragma unroll
    for (uint32_t ch = 0; ch < C; ch++) res[ch] = 0.0f;
#pragma unroll
    for (uint32_t idx = 0; idx < (1u << D); idx++) {
        float wc = 1.0f;
#pragma unroll
        for (uint32_t d = 0; d < D; d++) wc *= ((idx & (1u << d)) == 0) ? 1 - pos[d] : pos[d];
        LzVec<T, C> cvv;
        __builtin_memcpy(&cvv, (idx & 1u) ? oth[idx >> 1] : own[idx >> 1], sizeof(T) * C);
#pragma unroll
        for (uint32_t ch = 0; ch < C; ch++) res[ch] = LzElem<T>::acc(res[ch], wc, LzElem<T>::ld(&cvv.v[ch]));
    }
    LzVec<T, C> o;
#pragma unroll
    for (uint32_t ch = 0; ch < C; ch++) o.v[ch] = LzElem<T>::st(oob ? 0.0f : res[ch]);
    T* out = outputs + ((size_t)b0 * L + (size_t)level * n + t) * C;
    __builtin_memcpy(__builtin_assume_aligned(out, sizeof(T) * C), &o, sizeof(T) * C);
}

// one workgroup per tile; W = dwords per (sample, level) group = C * sizeof(T) / 4; the tile is [L][n][W] -> [n][L][W].
// Whole tiles move as 16-byte vectors (8 loads in flight per thread, then 8 stores): the pass is pure streaming and a quarter of the
// big-table forward's time, so it has to run at copy bandwidth (4-byte accesses, few in flight, got 2.9 TB/s: 0.74 ms per 2^23 x 128 B).
__global__ void __launch_bounds__(256)
lz_k_grid_untile(uint32_t* __restrict__ out, uint32_t B, uint32_t L, uint32_t W, uint32_t Tn, uint32_t pitch) {
    extern __shared__ __align__(16) uint32_t lz_untile_smem[];
    typedef uint32_t lz_u4 __attribute__((ext_vector_type(4)));
    const uint32_t b0 = blockIdx.x * Tn;
    const uint32_t n = (B - b0 < Tn) ? B - b0 : Tn;
    const uint32_t LW = L * W, nW = n * W;
    uint32_t* reg = out + (size_t)b0 * LW;
    const bool vec = (n == Tn) && (nW % 4u == 0u) && (LW % 4u == 0u) && (pitch % 4u == 0u) && (((size_t)b0 * LW) % 4u == 0u) &&
                     ((reinterpret_cast<uintptr_t>(out) & 15u) == 0u);   // workgroup-uniform; a caller may hand over a sliced / offset buffer
    if (vec) {
        const uint32_t per_level = nW / 4u, chunks = L * per_level;             // 16-byte chunks of the tile
        for (uint32_t c0 = 0; c0 < chunks; c0 += 8u * blockDim.x) {
            lz_u4 v[8];
#pragma unroll
            for (uint32_t k = 0; k < 8; k++) {
                const uint32_t c = c0 + k * blockDim.x + threadIdx.x;
                if (c < chunks) v[k] = __builtin_nontemporal_load(reinterpret_cast<const lz_u4*>(reg) + c);
            }
#pragma unroll
            for (uint32_t k = 0; k < 8; k++) {
                const uint32_t c = c0 + k * blockDim.x + threadIdx.x;
                if (c < chunks) {
                    const uint32_t l = c / per_level, i = (c - l * per_level) * 4u;
                    *reinterpret_cast<lz_u4*>(lz_untile_smem + l * pitch + i) = v[k];
                }
            }
        }
        __syncthreads();
        const uint32_t per_row = LW / 4u, rows_out = n * per_row;               // 16-byte chunks of the output rows
        for (uint32_t c = threadIdx.x; c < rows_out; c += blockDim.x) {
            const uint32_t t = c / per_row, r = (c - t * per_row) * 4u;
            lz_u4 v;
#pragma unroll
            for (uint32_t k = 0; k < 4; k++) {
                const uint32_t rr = r + k, l = rr / W, wq = rr - l * W;
                v[k] = lz_untile_smem[l * pitch + t * W + wq];
            }
            *(reinterpret_cast<lz_u4*>(reg) + c) = v;
        }
        return;
    }
    for (uint32_t l = 0; l < L; l++)
        for (uint32_t i = threadIdx.x; i < nW; i += blockDim.x) lz_untile_smem[l * pitch + i] = reg[(size_t)l * nW + i];
    __syncthreads();
    const uint32_t rows = blockDim.x / LW;          // >= 1: LW <= 256
    const uint32_t dt = threadIdx.x / LW, r = threadIdx.x - dt * LW;
    if (dt >= rows) return;
    const uint32_t l = r / W, wq = r - l * W;
    const uint32_t* src = lz_untile_smem + l * pitch + wq;
    for (uint32_t t = dt; t < n; t += rows) reg[(size_t)t * LW + r] = src[t * W];
}

template <typename T, uint32_t D, uint32_t C>
static void lz_grid_lm_launch(const float* inputs, const T* emb, const int* offsets, T* out, uint32_t B, uint32_t L,
                              const LzGridLevels& lv, uint32_t gridtype, bool ac, hipStream_t st) {
    constexpr uint32_t W = C * sizeof(T) / 4;
    const uint32_t LW = L * W;
    const uint32_t Tn = LW <= 64 ? 256u : (LW <= 128 ? 128u : 64u);      // tile <= 64 KB
    const uint32_t tiles = lz_div_up(B, Tn);
    if constexpr (sizeof(T) * C % 4 == 0 && D >= 3)   // two lanes per sample (x corner pairs share a line); measured: D3/C2 f32
                                                      // 3.62 -> 3.29 ms, f16 3.05 -> 2.36 ms per 2^23 samples; D2/C1 gets slower
        hipLaunchKernelGGL((lz_k_grid_forward_lmp<T, D, C>), dim3(tiles, L), dim3(2 * Tn), 0, st, inputs, emb, offsets, out, B, L, lv, gridtype, ac,
                           (const int*)nullptr, 0.0f);
    else
        hipLaunchKernelGGL((lz_k_grid_forward_lm<T, D, C>), dim3(tiles, L), dim3(Tn), 0, st, inputs, emb, offsets, out, B, L, lv, gridtype, ac);
    // LDS row pitch: level l starts at bank (l * 64/L) so the 64 lanes of a store (64/LW samples x L levels x W) hit 64 banks
    const uint32_t want = L < 64 ? 64u / L : 1u;
    uint32_t pitch = Tn * W + ((want + 64u - (Tn * W) % 64u) % 64u);
    pitch = (pitch + 3u) & ~3u;      // 16-byte rows for the vector path of lz_k_grid_untile
    hipLaunchKernelGGL(lz_k_grid_untile, dim3(tiles), dim3(256), (size_t)L * pitch * 4, st, reinterpret_cast<uint32_t*>(out), B, L, W, Tn, pitch);
}

// ---- level-resident backward: the level's gradient table is accumulated in LDS -------------------------------------
// Scattered global float atomics execute at the memory side at ~2e10 lane-adds/s chip-wide (64 lanes -> 64 lines), which
// is what the plain scatter kernel above gets (a triplane plane: 0.37 Gsample/s).  When a level's table fits LDS a
// workgroup owns ONE (level, sample chunk): it zeroes a private copy, accumulates its chunk with LDS atomics (see below),
// then flushes the non-zero entries with CONTIGUOUS global atomics (whole-line wave instructions: full atomic rate).
// Summation order differs from the plain kernel (as it does between two runs of the reference); results agree to rounding.
template <uint32_t D, uint32_t C, bool IN_LDS>
__device__ __forceinline__ void lz_grid_level_scatter(const float* __restrict__ grad, const float* __restrict__ inputs, float* dst,
                                                      uint32_t b0, uint32_t b1, uint32_t B, uint32_t L, uint32_t level, float scale,
                                                      uint32_t resolution, uint32_t hs, uint32_t mode, uint32_t gridtype,
                                                      bool align_corners, bool sample_major) {
    for (uint32_t b = b0 + threadIdx.x; b < b1; b += blockDim.x) {
        float x[D];
        bool oob = false;
#pragma unroll
        for (uint32_t d = 0; d < D; d++) {
            x[d] = inputs[(size_t)b * D + d];
            if (x[d] < 0 || x[d] > 1) oob = true;
        }
        if (oob) continue;
        const float* gsrc = grad + (sample_major ? ((size_t)b * L + level) * C : ((size_t)level * B + b) * C);
        float gcur[C];
#pragma unroll
        for (uint32_t ch = 0; ch < C; ch++) gcur[ch] = gsrc[ch];
        float pos[D];
        uint32_t pg[D];
#pragma unroll
        for (uint32_t d = 0; d < D; d++) {
            pos[d] = lz_fmaf(x[d], scale, align_corners ? 0.0f : 0.5f);
            pg[d] = (uint32_t)floorf(pos[d]);
            pos[d] -= (float)pg[d];
        }
        uint32_t term[D][2];
        if (mode != 2u) lz_grid_terms<D>(pg, mode, resolution, align_corners, term);
#pragma unroll
        for (uint32_t idx = 0; idx < (1u << D); idx++) {
            float w = 1.0f;
            uint32_t pl[D];
#pragma unroll
            for (uint32_t d = 0; d < D; d++) {
                if ((idx & (1u << d)) == 0) { w *= 1 - pos[d]; pl[d] = pg[d]; }
                else { w *= pos[d]; pl[d] = pg[d] + 1; }
            }
            uint32_t index;
            if (mode == 2u) index = lz_grid_index<D>(C, gridtype, align_corners, hs, resolution, pl);
            else index = lz_grid_corner<D>(term, idx, mode, hs) * C;
#pragma unroll
            for (uint32_t ch = 0; ch < C; ch++) {
                if constexpr (IN_LDS) __hip_atomic_fetch_add(dst + index + ch, w * gcur[ch], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                else lz_atomic_add(dst + index + ch, w * gcur[ch]);
            }
        }
    }
}

// 64-bit FIXED-POINT accumulators in LDS.  Why not float: on gfx950 ds_add_f32 runs ~50x slower than the integer LDS atomics (measured:
// 0.27 ms vs 1.02 ms per 2^22-sample plane with everything else equal); with float accumulators the atomics were the whole cost.  A workgroup first finds max|grad| over its (level, chunk), picks the power-of-two scale that leaves headroom for
// 4 * chunk additions in 51 bits (>= 2^-31 of that maximum as resolution; see to_fixed below), accumulates round(w * g * scale) with ds_add_u64, and flushes
// float(acc) / scale with contiguous global float atomics.  Within a workgroup the sum is exact up to the per-term rounding
// (<= 1.2e-10 of the chunk's largest gradient) and independent of the order -- tighter than f32 atomics for all but terms ~1e-6 of the
// maximum.  Levels whose 8-byte table does not fit 128 KB take the global float-atomic path inside the same kernel.
#define LZ_GRID_FX_LDS_BYTES 131072
// |g| for the max pass, with NaN mapped to +inf: fmaxf drops NaN operands and __float2ll_rn(NaN) adds 0, so a NaN gradient would vanish
// in the fixed-point path, while the reference's float atomicAdd propagates it into grad_embeddings (what GradScaler's non-finite check
// looks at).  A (level, chunk) with any non-finite gradient takes the float-atomic path, like inf always did.
__device__ __forceinline__ float lz_abs_nan_inf(float g) {
    const float a = fabsf(g);
    return a == a ? a : INFINITY;
}
#define LZ_GRID_FX_KEEP 72   // gradients a thread keeps in registers between the two passes (C == 1)
template <uint32_t D, uint32_t C>
__global__ void __launch_bounds__(1024)
lz_k_grid_backward_lds_fx(const float* __restrict__ grad, const float* __restrict__ inputs, const int* __restrict__ offsets,
                          float* __restrict__ grad_grid, uint32_t B, uint32_t L, LzGridLevels lv, uint32_t gridtype, bool align_corners,
                          bool sample_major, uint32_t chunk) {
    extern __shared__ __align__(16) unsigned long long lz_grid_acc64[];
    __shared__ float wmax[16];
    const uint32_t level = blockIdx.x % L, c = blockIdx.x / L;
    const uint32_t off0 = (uint32_t)offsets[level], hs = (uint32_t)offsets[level + 1] - off0;
    const uint32_t res = lv.res[level];
    const float scale_l = lv.scale[level];
    const uint32_t mode = lz_grid_level_mode<D>(hs, res, gridtype, align_corners);
    const uint32_t b0 = c * chunk, b1 = (B - b0 < chunk) ? B : b0 + chunk;
    float* gg = grad_grid + (size_t)off0 * C;
    const uint32_t n = hs * C;
    if ((size_t)n * 8 > LZ_GRID_FX_LDS_BYTES) {
        lz_grid_level_scatter<D, C, false>(grad, inputs, gg, b0, b1, B, L, level, scale_l, res, hs, mode, gridtype, align_corners, sample_major);
        return;
    }
    // pass 1: largest |grad| of this (level, chunk).  For C == 1 (the triplane planes) a thread keeps its <= LZ_GRID_FX_KEEP gradients in
    // registers, so pass 2 reads only the inputs again (12 instead of 16 bytes per sample and level).
    constexpr bool kKeep = (C == 1);
    float gkeep[kKeep ? LZ_GRID_FX_KEEP : 1];
    const bool kept = kKeep && D == 2 && blockDim.x == 1024 && (b1 - b0) <= LZ_GRID_FX_KEEP * 1024u;   // workgroup-uniform
    float gm = 0.0f;
    // kept path, WHICH rows a thread takes: LANE-MAJOR segments.  The 64 lanes of a wave-instruction must not sit in the same cell -- same-
    // address LDS atomics retire one lane at a time -- and consecutive rows do exactly that: ray-major rows are one ray's consecutive steps
    // (the same cell of a plane the ray is normal to: the xy plane of a camera looking along z ran 0.44 ms against 0.29 for the others),
    // step-major rows (lz_march_rays_train_grouped) are neighbouring rays at one step (every plane, every coarse level: 0.57 - 0.70 ms).
    // So the chunk is cut into 64 segments of SEG = 64 T rows (T <= 18), lane l walks segment l, and inside it the 16 waves x 4 rows x T
    // trips cover the segment:  row = b0 + l SEG + (16 k + w) 4 + i.  The lanes of an instruction are SEG rows apart -- different rays AND
    // different steps under either layout -- while a thread's four rows per trip are consecutive (one 16-byte load of gradients, two of
    // inputs) and the 16 waves of the workgroup read the neighbouring pieces of the same lines in the same trip.  (Rounds 3-4 voted per
    // workgroup between whole-line loads and a 4-row spread inside 1 024-row tiles; that could not separate the lanes of a step-major tile.)
    const uint32_t fx_T = kept ? (b1 - b0 + 4095u) / 4096u : 0u, fx_seg = 64u * fx_T;
    const uint32_t fx_row0 = b0 + (threadIdx.x & 63u) * fx_seg + (threadIdx.x >> 6) * 4u;       // + 64 k + i
    const float* gbase = grad + (sample_major ? (size_t)level * C : (size_t)level * B * C);     // C == 1 on this path
    const bool g_vec = !sample_major && (((uintptr_t)(gbase + b0) & 15u) == 0u) && (fx_seg % 4u == 0u);
    if (kept) {
#pragma unroll
        for (uint32_t k = 0; k < (kKeep ? LZ_GRID_FX_KEEP / 4 : 0); k++) {
            const uint32_t r = fx_row0 + 64u * k;
            float4 g4 = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
            if (k < fx_T) {
                if (g_vec && r + 4u <= b1) g4 = *reinterpret_cast<const float4*>(gbase + r);
                else {
                    if (r < b1) g4.x = gbase[sample_major ? (size_t)r * L : r];
                    if (r + 1 < b1) g4.y = gbase[sample_major ? (size_t)(r + 1) * L : r + 1];
                    if (r + 2 < b1) g4.z = gbase[sample_major ? (size_t)(r + 2) * L : r + 2];
                    if (r + 3 < b1) g4.w = gbase[sample_major ? (size_t)(r + 3) * L : r + 3];
                }
            }
            gkeep[4 * k] = g4.x; gkeep[4 * k + 1] = g4.y; gkeep[4 * k + 2] = g4.z; gkeep[4 * k + 3] = g4.w;
            gm = fmaxf(fmaxf(gm, lz_abs_nan_inf(g4.x)), fmaxf(lz_abs_nan_inf(g4.y), fmaxf(lz_abs_nan_inf(g4.z), lz_abs_nan_inf(g4.w))));
        }
    } else {
        for (uint32_t b = b0 + threadIdx.x; b < b1; b += blockDim.x) {
            const float* gsrc = grad + (sample_major ? ((size_t)b * L + level) * C : ((size_t)level * B + b) * C);
#pragma unroll
            for (uint32_t ch = 0; ch < C; ch++) gm = fmaxf(gm, lz_abs_nan_inf(gsrc[ch]));
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) gm = fmaxf(gm, __shfl_xor(gm, off, 64));
    if ((threadIdx.x & 63) == 0) wmax[threadIdx.x >> 6] = gm;
    for (uint32_t i = threadIdx.x; i < n; i += blockDim.x) lz_grid_acc64[i] = 0ull;
    __syncthreads();
    gm = wmax[0];
    for (uint32_t w = 1; w < (blockDim.x >> 6); w++) gm = fmaxf(gm, wmax[w]);
    if (!(gm > 0.0f) || !(gm < INFINITY)) {   // nothing to add (or non-finite gradients: leave them to the float path's semantics)
        if (gm > 0.0f) lz_grid_level_scatter<D, C, false>(grad, inputs, gg, b0, b1, B, L, level, scale_l, res, hs, mode, gridtype, align_corners, sample_major);
        return;
    }
    int ex;
    (void)frexpf(gm, &ex);                                    // gm < 2^ex
    int hb = 2;                                               // 4 corners... (2^D of them) per sample
    while ((1u << hb) < (b1 - b0) * (1u << D)) hb++;
    // round(v * 2^e) as a 64-bit integer WITHOUT the f32 -> i64 conversion (a dozen vector instructions, four times per sample: half of
    // this kernel's issue slots): in double, v * 2^e + 1.5 * 2^52 is rounded to an integer by the addition itself (round to nearest even,
    // |v * 2^e| < 2^51) and the integer sits in the low mantissa bits -- one convert, one fma, one 32-bit subtract of the magic's high word.
    // The scale leaves 51 bits instead of 62: >= 2^-31 of the chunk's largest gradient as resolution, far below an f32 ulp of the sum.
    int e = 51 - hb - ex;
    e = e > 100 ? 100 : (e < -100 ? -100 : e);
    const float inv = ldexpf(1.0f, -e);
    const double fxd = (double)ldexpf(1.0f, e);
    auto to_fixed = [&](float v) -> unsigned long long {
        const double t = __builtin_fma((double)v, fxd, 6755399441055744.0);
        return (unsigned long long)(__double_as_longlong(t) - 0x4338000000000000ll);
    };
    auto scatter_at = [&](const float (&x)[D], const float (&gcur)[C]) {
        bool oob = false;
#pragma unroll
        for (uint32_t d = 0; d < D; d++)
            if (x[d] < 0 || x[d] > 1) oob = true;
        if (oob) return;
        float pos[D];
        uint32_t pg[D];
#pragma unroll
        for (uint32_t d = 0; d < D; d++) {
            pos[d] = lz_fmaf(x[d], scale_l, align_corners ? 0.0f : 0.5f);
            pg[d] = (uint32_t)floorf(pos[d]);
            pos[d] -= (float)pg[d];
        }
        if constexpr (D == 2) {
            if (mode != 2u) {
                // the planes of the triplane head: the row term of the two upper corners is the lower one plus a constant (mod 2^32) and the
                // dense product fits the 24-bit multiplier -- one quarter-rate multiply per sample instead of up to eight; same indices,
                // same weights (1 * a * b == a * b), same order of the four additions
                const uint32_t s1 = align_corners ? res : res + 1;
                const uint32_t r0 = mode == 1u ? pg[1] * 2654435761u : __umul24(pg[1], s1);
                const uint32_t r1 = r0 + (mode == 1u ? 2654435761u : s1);
                const float wx[2] = {1 - pos[0], pos[0]}, wy[2] = {1 - pos[1], pos[1]};
#pragma unroll
                for (uint32_t idx = 0; idx < 4; idx++) {
                    const uint32_t c0 = pg[0] + (idx & 1u), rr = (idx >> 1) ? r1 : r0;
                    const uint32_t index = (mode == 1u ? ((c0 ^ rr) & (hs - 1u)) : c0 + rr) * C;
                    const float w = wx[idx & 1u] * wy[idx >> 1];
#pragma unroll
                    for (uint32_t ch = 0; ch < C; ch++) {
                        __hip_atomic_fetch_add(lz_grid_acc64 + index + ch, to_fixed(w * gcur[ch]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    }
                }
                return;
            }
        }
        uint32_t gterm[D][2];
        if (mode != 2u) lz_grid_terms<D>(pg, mode, res, align_corners, gterm);
#pragma unroll
        for (uint32_t idx = 0; idx < (1u << D); idx++) {
            float w = 1.0f;
            uint32_t pl[D];
#pragma unroll
            for (uint32_t d = 0; d < D; d++) {
                if ((idx & (1u << d)) == 0) { w *= 1 - pos[d]; pl[d] = pg[d]; }
                else { w *= pos[d]; pl[d] = pg[d] + 1; }
            }
            uint32_t index;
            if (mode == 2u) index = lz_grid_index<D>(C, gridtype, align_corners, hs, res, pl);
            else index = lz_grid_corner<D>(gterm, idx, mode, hs) * C;
#pragma unroll
            for (uint32_t ch = 0; ch < C; ch++) {
                __hip_atomic_fetch_add(lz_grid_acc64 + index + ch, to_fixed(w * gcur[ch]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
        }
    };
    auto scatter = [&](uint32_t b, const float (&gcur)[C]) {
        float x[D];
#pragma unroll
        for (uint32_t d = 0; d < D; d++) x[d] = inputs[(size_t)b * D + d];
        scatter_at(x, gcur);
    };
    if (kept) {
        if constexpr (D == 2) {
            const bool x_vec = (((uintptr_t)(inputs + (size_t)b0 * 2) & 15u) == 0u) && (fx_seg % 2u == 0u);
#pragma unroll
            for (uint32_t k = 0; k < (kKeep ? LZ_GRID_FX_KEEP / 4 : 0); k++) {
                const uint32_t r = fx_row0 + 64u * k;
                if (k < fx_T && r < b1) {
                    float xy[8];
                    if (x_vec && r + 4u <= b1) {
                        const float4 p = *reinterpret_cast<const float4*>(inputs + (size_t)r * 2), q = *reinterpret_cast<const float4*>(inputs + (size_t)r * 2 + 4);
                        xy[0] = p.x; xy[1] = p.y; xy[2] = p.z; xy[3] = p.w; xy[4] = q.x; xy[5] = q.y; xy[6] = q.z; xy[7] = q.w;
                    } else {
#pragma unroll
                        for (uint32_t i = 0; i < 4; i++) {
                            const bool in = r + i < b1;
                            xy[2 * i] = in ? inputs[(size_t)(r + i) * 2] : -1.0f;          // out of range: skipped like an out-of-bounds input
                            xy[2 * i + 1] = in ? inputs[(size_t)(r + i) * 2 + 1] : -1.0f;
                        }
                    }
#pragma unroll
                    for (uint32_t i = 0; i < 4; i++) {
                        if (r + i < b1) {
                            const float x[D] = {xy[2 * i], xy[2 * i + 1]};
                            float gcur[C];
                            gcur[0] = gkeep[4 * k + i];
                            scatter_at(x, gcur);
                        }
                    }
                }
            }
        }
    } else {
        for (uint32_t b = b0 + threadIdx.x; b < b1; b += blockDim.x) {
            const float* gsrc = grad + (sample_major ? ((size_t)b * L + level) * C : ((size_t)level * B + b) * C);
            float gcur[C];
#pragma unroll
            for (uint32_t ch = 0; ch < C; ch++) gcur[ch] = gsrc[ch];
            scatter(b, gcur);
        }
    }
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < n; i += blockDim.x) {
        const long long a = (long long)lz_grid_acc64[i];
        if (a != 0) lz_atomic_add(gg + i, __ll2float_rn(a) * inv);
    }
}

template <uint32_t D, uint32_t C>
static void lz_grid_bwd_lds_launch(const float* grad, const float* inputs, const int* offsets, float* gemb, uint32_t B, uint32_t L,
                                   const LzGridLevels& lv, uint32_t gridtype, bool ac, bool sm, hipStream_t st) {
    // ~1024 workgroups (2 per CU: 64 KB of LDS each), never less than 4096 samples per workgroup
    uint32_t n_chunks = lz_div_up(B, 4096);
    const uint32_t cap = 1024 / L > 0 ? 1024 / L : 1;
    if (n_chunks > cap) n_chunks = cap;
    const uint32_t fit = lz_div_up(B, LZ_GRID_FX_KEEP * 1024u);   // chunks small enough for the register-kept gradients (C == 1)
    if (C == 1 && n_chunks < fit && fit * L <= 4096) n_chunks = fit;
    {   // whole rounds of workgroups (one per CU: 128 KB of LDS each) when that keeps the chunks register-sized: fewer chunks = fewer flushes
        const uint32_t per_round = 256, down = (n_chunks * L / per_round) * per_round / L;
        if (down >= 1 && down >= fit && down * L >= per_round) n_chunks = down;
    }
    const uint32_t chunk = (lz_div_up(B, n_chunks) + 63u) & ~63u;      // whole 64-row blocks: the kept path's 16-byte loads stay aligned
    n_chunks = lz_div_up(B, chunk);
    static bool attr_set = false;   // per instantiation: more than 64 KB of dynamic LDS has to be requested once
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&lz_k_grid_backward_lds_fx<D, C>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  LZ_GRID_FX_LDS_BYTES);
        attr_set = true;
    }
    hipLaunchKernelGGL((lz_k_grid_backward_lds_fx<D, C>), dim3(n_chunks * L), dim3(1024), LZ_GRID_FX_LDS_BYTES, st, grad, inputs, offsets, gemb,
                       B, L, lv, gridtype, ac, sm, chunk);
}

// ---- host dispatch ----
template <typename T, uint32_t D>
static int lz_grid_fwd_c(const float* inputs, const T* emb, const int* offsets, T* out, uint32_t B, uint32_t C, uint32_t L,
                         const LzGridLevels& lv, T* dy_dx, uint32_t gridtype, bool ac, bool sm, bool resident, hipStream_t st) {
    dim3 grid, block(256);
    if (sm && !dy_dx && resident) {  // large batches: level-resident tables in LDS, see lz_k_grid_forward_lds
        if constexpr (D <= 3) {
            if (C == 1) { lz_grid_lds_launch<T, D, 1>(inputs, emb, offsets, out, B, L, lv, gridtype, ac, st); return LZ_OK; }
            if (C == 2) { lz_grid_lds_launch<T, D, 2>(inputs, emb, offsets, out, B, L, lv, gridtype, ac, st); return LZ_OK; }
        }
    }
    if (sm && !dy_dx && B >= 65536) {  // large batches over big tables: level-major tiles + in-place untile
        if constexpr (D == 2 || D == 3) {
            if constexpr (sizeof(T) == 4)
                if (C == 1) { lz_grid_lm_launch<T, D, 1>(inputs, emb, offsets, out, B, L, lv, gridtype, ac, st); return LZ_OK; }
            if (C == 2) { lz_grid_lm_launch<T, D, 2>(inputs, emb, offsets, out, B, L, lv, gridtype, ac, st); return LZ_OK; }
            if (C == 4) { lz_grid_lm_launch<T, D, 4>(inputs, emb, offsets, out, B, L, lv, gridtype, ac, st); return LZ_OK; }
            if (C == 8) { lz_grid_lm_launch<T, D, 8>(inputs, emb, offsets, out, B, L, lv, gridtype, ac, st); return LZ_OK; }
        }
    }
    if (sm && !dy_dx) {  // 2-D workgroup (level, sample), see lz_k_grid_forward_sm
        const uint32_t ny = 256 / L;
        const dim3 b2(L, ny, 1), g2(lz_div_up(B, ny), 1, 1);
        switch (C) {
            case 1: hipLaunchKernelGGL((lz_k_grid_forward_sm<T, D, 1>), g2, b2, 0, st, inputs, emb, offsets, out, B, L, lv, gridtype, ac); break;
            case 2: hipLaunchKernelGGL((lz_k_grid_forward_sm<T, D, 2>), g2, b2, 0, st, inputs, emb, offsets, out, B, L, lv, gridtype, ac); break;
            case 4: hipLaunchKernelGGL((lz_k_grid_forward_sm<T, D, 4>), g2, b2, 0, st, inputs, emb, offsets, out, B, L, lv, gridtype, ac); break;
            case 8: hipLaunchKernelGGL((lz_k_grid_forward_sm<T, D, 8>), g2, b2, 0, st, inputs, emb, offsets, out, B, L, lv, gridtype, ac); break;
            default: lz_set_error("GridEncoding: C must be 1, 2, 4, or 8."); return LZ_ERR_UNSUPPORTED;
        }
        return LZ_OK;
    }
    if (sm) grid = dim3(lz_div_up((uint64_t)B * L, 256), 1, 1);
    else grid = dim3(lz_div_up(B, 256), L, 1);
    switch (C) {
        case 1: hipLaunchKernelGGL((lz_k_grid_forward<T, D, 1>), grid, block, 0, st, inputs, emb, offsets, out, B, L, lv, dy_dx, gridtype, ac, sm); break;
        case 2: hipLaunchKernelGGL((lz_k_grid_forward<T, D, 2>), grid, block, 0, st, inputs, emb, offsets, out, B, L, lv, dy_dx, gridtype, ac, sm); break;
        case 4: hipLaunchKernelGGL((lz_k_grid_forward<T, D, 4>), grid, block, 0, st, inputs, emb, offsets, out, B, L, lv, dy_dx, gridtype, ac, sm); break;
        case 8: hipLaunchKernelGGL((lz_k_grid_forward<T, D, 8>), grid, block, 0, st, inputs, emb, offsets, out, B, L, lv, dy_dx, gridtype, ac, sm); break;
        default: lz_set_error("GridEncoding: C must be 1, 2, 4, or 8."); return LZ_ERR_UNSUPPORTED;
    }
    return LZ_OK;
}

template <typename T>
static int lz_grid_fwd_d(const float* inputs, const T* emb, const int* offsets, T* out, uint32_t B, uint32_t D, uint32_t C,
                         uint32_t L, const LzGridLevels& lv, T* dy_dx, uint32_t gridtype, bool ac, bool sm, bool resident,
                         hipStream_t st) {
    switch (D) {
        case 1: return lz_grid_fwd_c<T, 1>(inputs, emb, offsets, out, B, C, L, lv, dy_dx, gridtype, ac, sm, resident, st);
        case 2: return lz_grid_fwd_c<T, 2>(inputs, emb, offsets, out, B, C, L, lv, dy_dx, gridtype, ac, sm, resident, st);
        case 3: return lz_grid_fwd_c<T, 3>(inputs, emb, offsets, out, B, C, L, lv, dy_dx, gridtype, ac, sm, resident, st);
        case 4: return lz_grid_fwd_c<T, 4>(inputs, emb, offsets, out, B, C, L, lv, dy_dx, gridtype, ac, sm, resident, st);
        case 5: return lz_grid_fwd_c<T, 5>(inputs, emb, offsets, out, B, C, L, lv, dy_dx, gridtype, ac, sm, resident, st);
        default: lz_set_error("GridEncoding: D must be 1, 2, 3, 4, or 5"); return LZ_ERR_UNSUPPORTED;
    }
}

extern "C" int lz_grid_encode_forward(const float* inputs, const void* embeddings, const int32_t* offsets, void* outputs,
                                      uint32_t B, uint32_t D, uint32_t C, uint32_t L, float S, uint32_t H, void* dy_dx,
                                      uint32_t gridtype, int align_corners, int emb_f16, int out_layout, lz_stream_t stream) {
    if (B == 0) return LZ_OK;
    LZ_REQUIRE(inputs && embeddings && offsets && outputs, LZ_ERR_BAD_ARGUMENT, "grid_encode_forward: null tensor");
    LZ_REQUIRE(L >= 1 && H >= 1 && gridtype <= 1u, LZ_ERR_BAD_ARGUMENT, "grid_encode_forward: num_levels and base_resolution must be >= 1, gridtype 0 (hash) or 1 (tiled)");
    LZ_REQUIRE(out_layout >= 0 && out_layout <= 2, LZ_ERR_BAD_ARGUMENT, "grid_encode_forward: out_layout must be 0 (level-major [L, B, C]), 1 or 2 (sample-major [B, L*C])");
    LzGridLevels lv;
    LZ_REQUIRE(lz_fill_levels(lv, L, S, H) == 0, LZ_ERR_UNSUPPORTED, "grid_encode_forward: at most %d levels", LZ_MAX_LEVELS);
    if (B == 0) return LZ_OK;
    int rc;
    const bool sm = out_layout == 1 || out_layout == 2, resident = out_layout == 2;
    if (emb_f16)
        rc = lz_grid_fwd_d<__half>(inputs, (const __half*)embeddings, offsets, (__half*)outputs, B, D, C, L, lv, (__half*)dy_dx,
                                   gridtype, align_corners != 0, sm, resident, lz_st(stream));
    else
        rc = lz_grid_fwd_d<float>(inputs, (const float*)embeddings, offsets, (float*)outputs, B, D, C, L, lv, (float*)dy_dx,
                                  gridtype, align_corners != 0, sm, resident, lz_st(stream));
    if (rc != LZ_OK) return rc;
    LZ_CHECK_LAUNCH("grid_encode_forward");
    return LZ_OK;
}

// The level-major pass alone, for consumers that read the TILED layout themselves (lz_ngp.hip: the hash-grid NeRF head gathers its MFMA
// operands straight from it, so the [B, L*C] matrix is never untiled): outputs = [tile][level][sample in tile][C], tiles of
// LZ_GRID_TILE_ROWS samples, a last partial tile of n rows packed as [level][n][C].  `count` (device int32, may be NULL): rows in use.
extern "C" int lz_grid_encode_forward_tiled(const float* inputs, const void* embeddings, const int32_t* offsets, void* outputs, uint32_t B,
                                            const int32_t* count, float bound, uint32_t D, uint32_t C, uint32_t L, float S, uint32_t H,
                                            uint32_t gridtype, int align_corners, int emb_f16, lz_stream_t stream) {
    if (B == 0) return LZ_OK;
    LZ_REQUIRE(inputs && embeddings && offsets && outputs, LZ_ERR_BAD_ARGUMENT, "grid_encode_forward_tiled: null tensor");
    LZ_REQUIRE(D == 3 && C == 2, LZ_ERR_UNSUPPORTED, "grid_encode_forward_tiled: input_dim 3, level_dim 2 (get_encoder('hashgrid') defaults, encoding.py:6-8)");
    LZ_REQUIRE(L >= 1 && H >= 1 && gridtype <= 1u, LZ_ERR_BAD_ARGUMENT, "grid_encode_forward_tiled: num_levels and base_resolution must be >= 1, gridtype 0 (hash) or 1 (tiled)");
    LZ_REQUIRE(bound >= 0.0f, LZ_ERR_BAD_ARGUMENT, "grid_encode_forward_tiled: bound must be >= 0 (0 = inputs already in [0, 1])");
    LzGridLevels lv;
    LZ_REQUIRE(lz_fill_levels(lv, L, S, H) == 0, LZ_ERR_UNSUPPORTED, "grid_encode_forward_tiled: at most %d levels", LZ_MAX_LEVELS);
    const uint32_t Tn = LZ_GRID_TILE_ROWS, tiles = lz_div_up(B, Tn);
    const bool ac = align_corners != 0;
    hipStream_t st = lz_st(stream);
    if (emb_f16)
        hipLaunchKernelGGL((lz_k_grid_forward_lmp<__half, 3, 2>), dim3(tiles, L), dim3(2 * Tn), 0, st, inputs, (const __half*)embeddings, offsets,
                           (__half*)outputs, B, L, lv, gridtype, ac, count, bound);
    else
        hipLaunchKernelGGL((lz_k_grid_forward_lmp<float, 3, 2>), dim3(tiles, L), dim3(2 * Tn), 0, st, inputs, (const float*)embeddings, offsets,
                           (float*)outputs, B, L, lv, gridtype, ac, count, bound);
    LZ_CHECK_LAUNCH("grid_encode_forward_tiled");
    return LZ_OK;
}

extern "C" int lz_grid_corner_indices(const float* inputs, const int32_t* offsets, int32_t* corner_idx, uint32_t B, uint32_t D,
                                      uint32_t C, uint32_t L, float S, uint32_t H, uint32_t gridtype, int align_corners,
                                      lz_stream_t stream) {
    LzGridLevels lv;
    LZ_REQUIRE(lz_fill_levels(lv, L, S, H) == 0, LZ_ERR_UNSUPPORTED, "grid_corner_indices: at most %d levels", LZ_MAX_LEVELS);
    if (B == 0) return LZ_OK;
    LZ_REQUIRE(inputs && offsets && corner_idx, LZ_ERR_BAD_ARGUMENT, "grid_corner_indices: null tensor");
    LZ_REQUIRE(L >= 1 && H >= 1 && gridtype <= 1u, LZ_ERR_BAD_ARGUMENT, "grid_corner_indices: num_levels and base_resolution must be >= 1, gridtype 0 (hash) or 1 (tiled)");
    dim3 grid(lz_div_up(B, 256), L, 1), block(256);
    const bool ac = align_corners != 0;
    hipStream_t st = lz_st(stream);
    switch (D) {
        case 1: hipLaunchKernelGGL((lz_k_grid_corner_indices<1>), grid, block, 0, st, inputs, offsets, corner_idx, B, C, L, lv, gridtype, ac); break;
        case 2: hipLaunchKernelGGL((lz_k_grid_corner_indices<2>), grid, block, 0, st, inputs, offsets, corner_idx, B, C, L, lv, gridtype, ac); break;
        case 3: hipLaunchKernelGGL((lz_k_grid_corner_indices<3>), grid, block, 0, st, inputs, offsets, corner_idx, B, C, L, lv, gridtype, ac); break;
        case 4: hipLaunchKernelGGL((lz_k_grid_corner_indices<4>), grid, block, 0, st, inputs, offsets, corner_idx, B, C, L, lv, gridtype, ac); break;
        case 5: hipLaunchKernelGGL((lz_k_grid_corner_indices<5>), grid, block, 0, st, inputs, offsets, corner_idx, B, C, L, lv, gridtype, ac); break;
        default: lz_set_error("GridEncoding: D must be 1, 2, 3, 4, or 5"); return LZ_ERR_UNSUPPORTED;
    }
    LZ_CHECK_LAUNCH("grid_corner_indices");
    return LZ_OK;
}

template <typename T, uint32_t D, uint32_t C>
static void lz_grid_bwd_launch(const T* grad, const float* inputs, const int* offsets, T* gemb, uint32_t B, uint32_t L,
                               const LzGridLevels& lv, const T* dy_dx, T* ginp, uint32_t gridtype, bool ac, bool sm, bool resident,
                               hipStream_t st) {
    dim3 grid, block(256);
    bool done = false;
    if constexpr (sizeof(T) == 4 && D <= 3 && C <= 2) {
        if (resident) {  // level tables fit LDS: private fixed-point accumulation, see lz_k_grid_backward_lds_fx
            lz_grid_bwd_lds_launch<D, C>(grad, inputs, offsets, gemb, B, L, lv, gridtype, ac, sm, st);
            done = true;
        }
    }
    if constexpr (sizeof(T) == 4 && C <= 2 && D >= 2) {
        static const bool plain = getenv("LZ_GRID_BWD_PLAIN") != nullptr;   // diagnostic: the one-lane-per-(sample, level) kernel
        if (!done && !plain && B >= 4096) {  // big tables: lanes share lines (lz_k_grid_backward_xc)
            hipLaunchKernelGGL((lz_k_grid_backward_xc<D, C>), dim3(lz_div_up((uint64_t)B * 2 * C, 256), L, 1), block, 0, st, grad, inputs, offsets,
                               gemb, B, L, lv, gridtype, ac, sm);
            done = true;
        }
    }
    if (!done) {
        if (sm) grid = dim3(lz_div_up((uint64_t)B * L, 256), 1, 1);
        else grid = dim3(lz_div_up(B, 256), L, 1);
        hipLaunchKernelGGL((lz_k_grid_backward<T, D, C>), grid, block, 0, st, grad, inputs, offsets, gemb, B, L, lv, gridtype, ac, sm);
    }
    if (dy_dx && ginp)
        hipLaunchKernelGGL((lz_k_grid_input_backward<T, D, C>), dim3(lz_div_up((uint64_t)B * D, 256)), block, 0, st, grad, dy_dx, ginp, B, L, sm);
}

template <typename T, uint32_t D>
static int lz_grid_bwd_c(const T* grad, const float* inputs, const int* offsets, T* gemb, uint32_t B, uint32_t C, uint32_t L,
                         const LzGridLevels& lv, const T* dy_dx, T* ginp, uint32_t gridtype, bool ac, bool sm, bool resident, hipStream_t st) {
    switch (C) {
        case 1: lz_grid_bwd_launch<T, D, 1>(grad, inputs, offsets, gemb, B, L, lv, dy_dx, ginp, gridtype, ac, sm, resident, st); break;
        case 2: lz_grid_bwd_launch<T, D, 2>(grad, inputs, offsets, gemb, B, L, lv, dy_dx, ginp, gridtype, ac, sm, resident, st); break;
        case 4: lz_grid_bwd_launch<T, D, 4>(grad, inputs, offsets, gemb, B, L, lv, dy_dx, ginp, gridtype, ac, sm, resident, st); break;
        case 8: lz_grid_bwd_launch<T, D, 8>(grad, inputs, offsets, gemb, B, L, lv, dy_dx, ginp, gridtype, ac, sm, resident, st); break;
        default: lz_set_error("GridEncoding: C must be 1, 2, 4, or 8."); return LZ_ERR_UNSUPPORTED;
    }
    return LZ_OK;
}

template <typename T>
static int lz_grid_bwd_d(const T* grad, const float* inputs, const int* offsets, T* gemb, uint32_t B, uint32_t D, uint32_t C,
                         uint32_t L, const LzGridLevels& lv, const T* dy_dx, T* ginp, uint32_t gridtype, bool ac, bool sm, bool resident,
                         hipStream_t st) {
    switch (D) {
        case 1: return lz_grid_bwd_c<T, 1>(grad, inputs, offsets, gemb, B, C, L, lv, dy_dx, ginp, gridtype, ac, sm, resident, st);
        case 2: return lz_grid_bwd_c<T, 2>(grad, inputs, offsets, gemb, B, C, L, lv, dy_dx, ginp, gridtype, ac, sm, resident, st);
        case 3: return lz_grid_bwd_c<T, 3>(grad, inputs, offsets, gemb, B, C, L, lv, dy_dx, ginp, gridtype, ac, sm, resident, st);
        case 4: return lz_grid_bwd_c<T, 4>(grad, inputs, offsets, gemb, B, C, L, lv, dy_dx, ginp, gridtype, ac, sm, resident, st);
        case 5: return lz_grid_bwd_c<T, 5>(grad, inputs, offsets, gemb, B, C, L, lv, dy_dx, ginp, gridtype, ac, sm, resident, st);
        default: lz_set_error("GridEncoding: D must be 1, 2, 3, 4, or 5"); return LZ_ERR_UNSUPPORTED;
    }
}

extern "C" int lz_grid_encode_backward(const void* grad, const float* inputs, const void* embeddings, const int32_t* offsets,
                                       void* grad_embeddings, uint32_t B, uint32_t D, uint32_t C, uint32_t L, float S, uint32_t H,
                                       const void* dy_dx, void* grad_inputs, uint32_t gridtype, int align_corners, int emb_f16,
                                       int grad_layout, lz_stream_t stream) {
    (void)embeddings;
    if (B == 0) return LZ_OK;
    LZ_REQUIRE(grad && inputs && offsets && grad_embeddings, LZ_ERR_BAD_ARGUMENT, "grid_encode_backward: null tensor");
    LZ_REQUIRE(L >= 1 && H >= 1 && gridtype <= 1u, LZ_ERR_BAD_ARGUMENT, "grid_encode_backward: num_levels and base_resolution must be >= 1, gridtype 0 (hash) or 1 (tiled)");
    LZ_REQUIRE(grad_layout >= 0 && grad_layout <= 3, LZ_ERR_BAD_ARGUMENT, "grid_encode_backward: grad_layout must be 0 .. 3 (see the header)");
    LZ_REQUIRE(!grad_inputs || dy_dx, LZ_ERR_BAD_ARGUMENT, "grid_encode_backward: grad_inputs needs dy_dx");
    LzGridLevels lv;
    LZ_REQUIRE(lz_fill_levels(lv, L, S, H) == 0, LZ_ERR_UNSUPPORTED, "grid_encode_backward: at most %d levels", LZ_MAX_LEVELS);
    if (B == 0) return LZ_OK;
    int rc;
    const bool sm = grad_layout == 1 || grad_layout == 2, resident = grad_layout >= 2;
    if (emb_f16)
        rc = lz_grid_bwd_d<__half>((const __half*)grad, inputs, offsets, (__half*)grad_embeddings, B, D, C, L, lv, (const __half*)dy_dx,
                                   (__half*)grad_inputs, gridtype, align_corners != 0, sm, resident, lz_st(stream));
    else
        rc = lz_grid_bwd_d<float>((const float*)grad, inputs, offsets, (float*)grad_embeddings, B, D, C, L, lv, (const float*)dy_dx,
                                  (float*)grad_inputs, gridtype, align_corners != 0, sm, resident, lz_st(stream));
    if (rc != LZ_OK) return rc;
    LZ_CHECK_LAUNCH("grid_encode_backward");
    return LZ_OK;
}
