/*
 * oracle/grid_oracle.c -- TEST INFRASTRUCTURE ONLY (CPU checker, never shipped, never timed as product).
 *
 * CPU restatement of the reference's multi-resolution grid encoder:
 *   forward  kernel_grid            /root/reference/gridencoder/src/gridencoder.cu:75-223
 *   index    get_grid_index         gridencoder.cu:54-72   hash fast_hash gridencoder.cu:35-51
 *   backward kernel_grid_backward   gridencoder.cu:226-313 kernel_input_backward gridencoder.cu:316-342
 *
 * Parity status: the reference ships no tests or golden vectors for this path and its CUDA kernels
 * cannot be built here (no nvcc) -- "parity unpinned" against the CUDA binary; pinned instead by
 * (a) the reference's importable Python (table layout / offsets, tests/golden/) and (b) analytic
 * known-answer tests (dense level == bilinear interpolation).
 *
 * Floating-point conventions: nvcc contracts `a*b+c` inside one expression into an FMA (the
 * reference builds with default -fmad=true, gridencoder/setup.py), so those spots are explicit
 * lz_fmaf() here; everything else is one IEEE operation per C operator (-ffp-contract=off).
 * Embedding element type: 0 = f32, 1 = f16 (at::Half semantics: every `half op float` promotes to
 * float and the result is rounded back to half when stored in a half variable, gridencoder.cu:142,165).
 */
#include <math.h>
#include <stdint.h>
#include <string.h>
#include "lzzx_detmath.h"
#include "lzzx_half.h"

#define LZO_MAX_D 5
#define LZO_MAX_C 8

static const uint32_t k_primes[7] = {1u, 2654435761u, 805459861u, 3674653429u, 2097192037u, 1434869437u, 2165219737u};

/* gridencoder.cu:54-72 */
static uint32_t grid_index(uint32_t D, uint32_t C, uint32_t gridtype, int align_corners, uint32_t ch,
                           uint32_t hashmap_size, uint32_t resolution, const uint32_t* pos_grid) {
    uint32_t stride = 1, index = 0;
    for (uint32_t d = 0; d < D && stride <= hashmap_size; d++) {
        index += pos_grid[d] * stride;
        stride *= align_corners ? resolution : (resolution + 1);
    }
    if (gridtype == 0 && stride > hashmap_size) {
        uint32_t h = 0;
        for (uint32_t d = 0; d < D; d++) h ^= pos_grid[d] * k_primes[d];
        index = h;
    }
    return (index % hashmap_size) * C + ch;
}

/* per-level constants, gridencoder.cu:124-126. exp2f is the host libm here; the product computes the
 * same two numbers on the host with the same call and hands them to the kernel as arguments. */
void lzo_grid_level_params(uint32_t level, float S, uint32_t H, float* scale, uint32_t* resolution) {
    const float sc = exp2f((float)level * S) * (float)H - 1.0f;
    *scale = sc;
    *resolution = (uint32_t)ceilf(sc) + 1u;
}

static inline float emb_load(const void* emb, int emb_f16, size_t i) {
    return emb_f16 ? lz_half_to_float(((const uint16_t*)emb)[i]) : ((const float*)emb)[i];
}

/* outputs: [L, B, C] level-major (gridencoder.cu:95); dy_dx: [B, L, D, C] or NULL (gridencoder.cu:181) */
void lzo_grid_encode_forward(const float* inputs, const void* emb, const int32_t* offsets, void* outputs,
                             uint32_t B, uint32_t D, uint32_t C, uint32_t L, float S, uint32_t H,
                             void* dy_dx, uint32_t gridtype, int align_corners, int emb_f16) {
    for (uint32_t level = 0; level < L; level++) {
        const size_t goff = (size_t)(uint32_t)offsets[level] * C;
        const uint32_t hashmap_size = (uint32_t)(offsets[level + 1] - offsets[level]);
        float scale; uint32_t resolution;
        lzo_grid_level_params(level, S, H, &scale, &resolution);
#pragma omp parallel for schedule(static)
        for (uint32_t b = 0; b < B; b++) {
            const float* x = inputs + (size_t)b * D;
            const size_t oidx = ((size_t)level * B + b) * C;
            int oob = 0;
            for (uint32_t d = 0; d < D; d++) if (x[d] < 0 || x[d] > 1) oob = 1;
            if (oob) {
                for (uint32_t ch = 0; ch < C; ch++) {
                    if (emb_f16) ((uint16_t*)outputs)[oidx + ch] = 0; else ((float*)outputs)[oidx + ch] = 0;
                }
                if (dy_dx) {
                    const size_t didx = (size_t)b * D * L * C + (size_t)level * D * C;
                    for (uint32_t i = 0; i < D * C; i++) {
                        if (emb_f16) ((uint16_t*)dy_dx)[didx + i] = 0; else ((float*)dy_dx)[didx + i] = 0;
                    }
                }
                continue;
            }
            float pos[LZO_MAX_D]; uint32_t pg[LZO_MAX_D];
            for (uint32_t d = 0; d < D; d++) {
                pos[d] = lz_fmaf(x[d], scale, align_corners ? 0.0f : 0.5f);
                pg[d] = (uint32_t)floorf(pos[d]);
                pos[d] -= (float)pg[d];
            }
            float res[LZO_MAX_C] = {0};
            for (uint32_t idx = 0; idx < (1u << D); idx++) {
                float w = 1; uint32_t pl[LZO_MAX_D];
                for (uint32_t d = 0; d < D; d++) {
                    if ((idx & (1u << d)) == 0) { w *= 1 - pos[d]; pl[d] = pg[d]; }
                    else { w *= pos[d]; pl[d] = pg[d] + 1; }
                }
                const uint32_t index = grid_index(D, C, gridtype, align_corners, 0, hashmap_size, resolution, pl);
                for (uint32_t ch = 0; ch < C; ch++) {
                    const float g = emb_load(emb, emb_f16, goff + index + ch);
                    if (emb_f16) res[ch] = lz_round_to_half(res[ch] + lz_round_to_half(w * g)); /* Half += Half(float*Half) */
                    else res[ch] = lz_fmaf(w, g, res[ch]);
                }
            }
            for (uint32_t ch = 0; ch < C; ch++) {
                if (emb_f16) ((uint16_t*)outputs)[oidx + ch] = lz_float_to_half(res[ch]);
                else ((float*)outputs)[oidx + ch] = res[ch];
            }
            if (dy_dx) {
                const size_t didx = (size_t)b * D * L * C + (size_t)level * D * C;
                for (uint32_t gd = 0; gd < D; gd++) {
                    float rg[LZO_MAX_C] = {0};
                    for (uint32_t idx = 0; idx < (1u << (D - 1)); idx++) {
                        float w = scale; uint32_t pl[LZO_MAX_D];
                        for (uint32_t nd = 0; nd < D - 1; nd++) {
                            const uint32_t d = (nd >= gd) ? (nd + 1) : nd;
                            if ((idx & (1u << nd)) == 0) { w *= 1 - pos[d]; pl[d] = pg[d]; }
                            else { w *= pos[d]; pl[d] = pg[d] + 1; }
                        }
                        pl[gd] = pg[gd];
                        const uint32_t il = grid_index(D, C, gridtype, align_corners, 0, hashmap_size, resolution, pl);
                        pl[gd] = pg[gd] + 1;
                        const uint32_t ir = grid_index(D, C, gridtype, align_corners, 0, hashmap_size, resolution, pl);
                        for (uint32_t ch = 0; ch < C; ch++) {
                            const float gr = emb_load(emb, emb_f16, goff + ir + ch);
                            const float gl = emb_load(emb, emb_f16, goff + il + ch);
                            if (emb_f16) rg[ch] = lz_round_to_half(rg[ch] + lz_round_to_half(w * lz_round_to_half(gr - gl)));
                            else rg[ch] = lz_fmaf(w, gr - gl, rg[ch]);
                        }
                    }
                    for (uint32_t ch = 0; ch < C; ch++) {
                        if (emb_f16) ((uint16_t*)dy_dx)[didx + gd * C + ch] = lz_float_to_half(rg[ch]);
                        else ((float*)dy_dx)[didx + gd * C + ch] = rg[ch];
                    }
                }
            }
        }
    }
}

/* Flat table indices of the 2^D corners for every (level, sample): [L, B, 2^D] int32, -1 when the
 * sample is out of range.  Not a reference entry point: it exposes get_grid_index so tests can
 * assert index parity bit-for-bit (SURVEY 8d "grid indices bit-exact"). */
void lzo_grid_corner_indices(const float* inputs, const int32_t* offsets, int32_t* corner_idx,
                             uint32_t B, uint32_t D, uint32_t C, uint32_t L, float S, uint32_t H,
                             uint32_t gridtype, int align_corners) {
    const uint32_t NC = 1u << D;
    for (uint32_t level = 0; level < L; level++) {
        const uint32_t hashmap_size = (uint32_t)(offsets[level + 1] - offsets[level]);
        float scale; uint32_t resolution;
        lzo_grid_level_params(level, S, H, &scale, &resolution);
#pragma omp parallel for schedule(static)
        for (uint32_t b = 0; b < B; b++) {
            const float* x = inputs + (size_t)b * D;
            int32_t* out = corner_idx + ((size_t)level * B + b) * NC;
            int oob = 0;
            for (uint32_t d = 0; d < D; d++) if (x[d] < 0 || x[d] > 1) oob = 1;
            if (oob) { for (uint32_t i = 0; i < NC; i++) out[i] = -1; continue; }
            uint32_t pg[LZO_MAX_D];
            for (uint32_t d = 0; d < D; d++) {
                const float p = lz_fmaf(x[d], scale, align_corners ? 0.0f : 0.5f);
                pg[d] = (uint32_t)floorf(p);
            }
            for (uint32_t idx = 0; idx < NC; idx++) {
                uint32_t pl[LZO_MAX_D];
                for (uint32_t d = 0; d < D; d++) pl[d] = pg[d] + ((idx >> d) & 1u);
                out[idx] = (int32_t)((uint32_t)offsets[level] * C +
                                     grid_index(D, C, gridtype, align_corners, 0, hashmap_size, resolution, pl));
            }
        }
    }
}

/* grad: [L, B, C]; grad_embeddings [sO, C] must be zero-filled by the caller (grid.py:72).
 * Sequential accumulation in (level, sample, corner) order; the reference's atomicAdd order is
 * unspecified, so float parity on this entry is tolerance-based.  grad_inputs: [B, D] or NULL. */
void lzo_grid_encode_backward(const float* grad, const float* inputs, const int32_t* offsets, float* grad_emb,
                              uint32_t B, uint32_t D, uint32_t C, uint32_t L, float S, uint32_t H,
                              const float* dy_dx, float* grad_inputs, uint32_t gridtype, int align_corners) {
    for (uint32_t level = 0; level < L; level++) {
        const size_t goff = (size_t)(uint32_t)offsets[level] * C;
        const uint32_t hashmap_size = (uint32_t)(offsets[level + 1] - offsets[level]);
        float scale; uint32_t resolution;
        lzo_grid_level_params(level, S, H, &scale, &resolution);
        for (uint32_t b = 0; b < B; b++) {
            const float* x = inputs + (size_t)b * D;
            int oob = 0;
            for (uint32_t d = 0; d < D; d++) if (x[d] < 0 || x[d] > 1) oob = 1;
            if (oob) continue;
            float pos[LZO_MAX_D]; uint32_t pg[LZO_MAX_D];
            for (uint32_t d = 0; d < D; d++) {
                pos[d] = lz_fmaf(x[d], scale, align_corners ? 0.0f : 0.5f);
                pg[d] = (uint32_t)floorf(pos[d]);
                pos[d] -= (float)pg[d];
            }
            const float* g = grad + ((size_t)level * B + b) * C;
            for (uint32_t idx = 0; idx < (1u << D); idx++) {
                float w = 1; uint32_t pl[LZO_MAX_D];
                for (uint32_t d = 0; d < D; d++) {
                    if ((idx & (1u << d)) == 0) { w *= 1 - pos[d]; pl[d] = pg[d]; }
                    else { w *= pos[d]; pl[d] = pg[d] + 1; }
                }
                const uint32_t index = grid_index(D, C, gridtype, align_corners, 0, hashmap_size, resolution, pl);
                for (uint32_t ch = 0; ch < C; ch++) grad_emb[goff + index + ch] += w * g[ch];
            }
        }
    }
    if (dy_dx && grad_inputs) { /* gridencoder.cu:316-342 */
        for (uint32_t b = 0; b < B; b++)
            for (uint32_t d = 0; d < D; d++) {
                float r = 0;
                for (uint32_t l = 0; l < L; l++)
                    for (uint32_t ch = 0; ch < C; ch++)
                        r = lz_fmaf(grad[((size_t)l * B + b) * C + ch],
                                    dy_dx[(size_t)b * L * D * C + (size_t)l * D * C + d * C + ch], r);
                grad_inputs[(size_t)b * D + d] = r;
            }
    }
}

/* The same backward with half tables (scalar_t = at::Half: the autocast branch of grid.py:38-39 when C is even; the torso encoder
 * D2 L16 C2 under `-O`), gridencoder.cu:296-311: the gradient is half, each term is `(__half)(w * grad_cur[c])` -- an f32 product of
 * the float weight and the promoted half, rounded to half -- and every atomicAdd (`__half2` pairs when N_C is even, at::Half CAS
 * otherwise) is a half + half addition rounded to nearest even INTO the half table.  grad: [L, B, C] half; grad_emb [sO, C] half,
 * zero-filled by the caller.  The order of the atomics is unspecified in the reference; this restatement accumulates in (level,
 * sample, corner) order, so a table entry with one or two terms is order-free (half addition commutes) and one with more is
 * compared to tolerance.  For that comparison `exact` (double, or NULL) receives the exact sum of the SAME half-rounded terms,
 * `absum` (double, or NULL) the sum of their magnitudes and `terms` (int32, or NULL) their count per entry.
 * grad_inputs (gridencoder.cu:316-342 with scalar_t = at::Half): `result += grad * dy_dx` is Half += Half * Half -- the product
 * and the running sum are each rounded to half -- sequential over (level, channel): deterministic. */
void lzo_grid_encode_backward_f16(const uint16_t* grad, const float* inputs, const int32_t* offsets, uint16_t* grad_emb,
                                  double* exact, double* absum, int32_t* terms,
                                  uint32_t B, uint32_t D, uint32_t C, uint32_t L, float S, uint32_t H,
                                  const uint16_t* dy_dx, uint16_t* grad_inputs, uint32_t gridtype, int align_corners) {
    for (uint32_t level = 0; level < L; level++) {
        const size_t goff = (size_t)(uint32_t)offsets[level] * C;
        const uint32_t hashmap_size = (uint32_t)(offsets[level + 1] - offsets[level]);
        float scale; uint32_t resolution;
        lzo_grid_level_params(level, S, H, &scale, &resolution);
        for (uint32_t b = 0; b < B; b++) {
            const float* x = inputs + (size_t)b * D;
            int oob = 0;
            for (uint32_t d = 0; d < D; d++) if (x[d] < 0 || x[d] > 1) oob = 1;
            if (oob) continue;
            float pos[LZO_MAX_D]; uint32_t pg[LZO_MAX_D];
            for (uint32_t d = 0; d < D; d++) {
                pos[d] = lz_fmaf(x[d], scale, align_corners ? 0.0f : 0.5f);
                pg[d] = (uint32_t)floorf(pos[d]);
                pos[d] -= (float)pg[d];
            }
            const uint16_t* g = grad + ((size_t)level * B + b) * C;
            for (uint32_t idx = 0; idx < (1u << D); idx++) {
                float w = 1; uint32_t pl[LZO_MAX_D];
                for (uint32_t d = 0; d < D; d++) {
                    if ((idx & (1u << d)) == 0) { w *= 1 - pos[d]; pl[d] = pg[d]; }
                    else { w *= pos[d]; pl[d] = pg[d] + 1; }
                }
                const uint32_t index = grid_index(D, C, gridtype, align_corners, 0, hashmap_size, resolution, pl);
                for (uint32_t ch = 0; ch < C; ch++) {
                    const size_t e = goff + index + ch;
                    const float term = lz_round_to_half(w * lz_half_to_float(g[ch]));
                    grad_emb[e] = lz_float_to_half(lz_half_to_float(grad_emb[e]) + term);
                    if (exact) exact[e] += (double)term;
                    if (absum) absum[e] += fabs((double)term);
                    if (terms) terms[e] += 1;
                }
            }
        }
    }
    if (dy_dx && grad_inputs) {
        for (uint32_t b = 0; b < B; b++)
            for (uint32_t d = 0; d < D; d++) {
                float r = 0;
                for (uint32_t l = 0; l < L; l++)
                    for (uint32_t ch = 0; ch < C; ch++) {
                        const float gv = lz_half_to_float(grad[((size_t)l * B + b) * C + ch]);
                        const float jv = lz_half_to_float(dy_dx[(size_t)b * L * D * C + (size_t)l * D * C + d * C + ch]);
                        r = lz_round_to_half(r + lz_round_to_half(gv * jv));
                    }
                grad_inputs[(size_t)b * D + d] = lz_float_to_half(r);
            }
    }
}

/* Table layout of GridEncoder.__init__ (gridencoder/grid.py:108-121): float64 resolution, cap at
 * 2^log2_hashmap_size, round up to a multiple of 8.  offsets must hold L+1 entries. */
void lzo_grid_offsets(uint32_t D, uint32_t L, double per_level_scale, uint32_t H, uint32_t log2_hashmap_size,
                      int align_corners, int32_t* offsets) {
    const double max_params = ldexp(1.0, (int)log2_hashmap_size);
    int64_t offset = 0;
    for (uint32_t i = 0; i < L; i++) {
        const double resolution = ceil((double)H * pow(per_level_scale, (double)i));
        const double r = align_corners ? resolution : resolution + 1;
        double p = pow(r, (double)D);
        if (p > max_params) p = max_params;
        const int64_t n = (int64_t)(ceil(p / 8.0) * 8.0);
        offsets[i] = (int32_t)offset;
        offset += n;
    }
    offsets[L] = (int32_t)offset;
}
