// lz_march.h -- the per-ray pieces of raymarching.cu shared by the operator kernels (lz_raymarch.hip) and the fused frame kernel
// (lz_frame.hip): Morton helpers, mip levels, the slab test of near_far_from_aabb and the occupancy-grid DDA step (LzMarch).
#ifndef LZ_MARCH_H
#define LZ_MARCH_H
#include <float.h>
#include <math.h>

#include "lz_common.h"
#include "lzzx_detmath.h"

#define LZ_SQRT3F 1.7320508075688772f
#define LZ_RPIF 0.3183098861837907f

// ------------------------------------------------------------------------------------------------
// helpers (raymarching.cu:42-81)
// ------------------------------------------------------------------------------------------------
__host__ __device__ __forceinline__ uint32_t lz_expand_bits(uint32_t v) {
    v = (v * 0x00010001u) & 0xFF0000FFu;
    v = (v * 0x00000101u) & 0x0F00F00Fu;
    v = (v * 0x00000011u) & 0xC30C30C3u;
    v = (v * 0x00000005u) & 0x49249249u;
    return v;
}
__host__ __device__ __forceinline__ uint32_t lz_morton3(uint32_t x, uint32_t y, uint32_t z) {
    return lz_expand_bits(x) | (lz_expand_bits(y) << 1) | (lz_expand_bits(z) << 2);
}
__host__ __device__ __forceinline__ uint32_t lz_morton3_inv(uint32_t x) {
    x = x & 0x49249249u;
    x = (x | (x >> 2)) & 0xc30c30c3u;
    x = (x | (x >> 4)) & 0x0f00f00fu;
    x = (x | (x >> 8)) & 0xff0000ffu;
    x = (x | (x >> 16)) & 0x0000ffffu;
    return x;
}
__device__ __forceinline__ int lz_mip_from_pos(float x, float y, float z, float max_cascade) {
    const float mx = lz_fmaxf(lz_fabsf(x), lz_fmaxf(lz_fabsf(y), lz_fabsf(z)));
    return (int)lz_fminf(max_cascade - 1, lz_fmaxf(0, (float)lz_frexp_exp(mx)));
}
__device__ __forceinline__ int lz_mip_from_dt(float dt, float H, float max_cascade) {
    const float mx = (dt * H) * 0.5f;   // raymarching.cu:50 narrows (double)(dt * H) * 0.5: halving a float is exact in either type
    return (int)lz_fminf(max_cascade - 1, lz_fmaxf(0, (float)lz_frexp_exp(mx)));
}

// bit-spread table for the Morton code of the march (LzMarch::morton_lut): every thread of the workgroup writes its share, caller barriers
#define LZ_MORTON_LUT 256
__device__ __forceinline__ void lz_morton_lut_stage(uint32_t* lut) {
    for (uint32_t i = threadIdx.x; i < LZ_MORTON_LUT; i += blockDim.x) lut[i] = lz_expand_bits(i);
}

// slab intersection with the aabb (raymarching.cu:91-145); a miss yields near = far = FLT_MAX
__device__ __forceinline__ void lz_near_far_ray(float ox, float oy, float oz, float dx, float dy, float dz, const float* __restrict__ aabb,
                                                float min_near, float& near_out, float& far_out) {
    const float rdx = 1 / dx, rdy = 1 / dy, rdz = 1 / dz;
    float near = (aabb[0] - ox) * rdx, far = (aabb[3] - ox) * rdx;
    if (near > far) { const float c = near; near = far; far = c; }
    float near_y = (aabb[1] - oy) * rdy, far_y = (aabb[4] - oy) * rdy;
    if (near_y > far_y) { const float c = near_y; near_y = far_y; far_y = c; }
    if (near > far_y || near_y > far) { near_out = far_out = FLT_MAX; return; }
    if (near_y > near) near = near_y;
    if (far_y < far) far = far_y;
    float near_z = (aabb[2] - oz) * rdz, far_z = (aabb[5] - oz) * rdz;
    if (near_z > far_z) { const float c = near_z; near_z = far_z; far_z = c; }
    if (near > far_z || near_z > far) { near_out = far_out = FLT_MAX; return; }
    if (near_z > near) near = near_z;
    if (far_z < far) far = far_z;
    if (near < min_near) near = min_near;
    near_out = near;
    far_out = far;
}

// ------------------------------------------------------------------------------------------------
// marching
// ------------------------------------------------------------------------------------------------
// the divisions of LzMarch::init, which depend on the frame only (raymarching.cu:380-381, 866-867): a persistent kernel that re-initialises its
// per-ray march every pass takes them from the host instead of running four IEEE division sequences (~11 vector instructions each) per pass
struct LzMarchFrame {
    float rbound, rH, dt_max, dt_min;
};
__host__ __device__ __forceinline__ LzMarchFrame lz_march_frame(float bound, uint32_t max_steps, uint32_t C, uint32_t H) {
    LzMarchFrame f;
    f.rbound = 1 / bound;
    f.rH = 1 / (float)H;
    f.dt_max = 2 * LZ_SQRT3F * (float)(1 << (C - 1)) / (float)H;
    const float dmin = 2 * LZ_SQRT3F / (float)max_steps;
    f.dt_min = f.dt_max < dmin ? f.dt_max : dmin;       // fminf(dt_max, 2 sqrt(3) / max_steps); max_steps = 0 gives +inf: dt_max
    return f;
}

struct LzMarch {
    float ox, oy, oz, dx, dy, dz, rdx, rdy, rdz;
    float bound, rbound, dt_gamma, dt_min, dt_max, rH, H3, fC, fH, halfH;
    uint32_t H;
    bool pow2H, one_cascade;
    const uint8_t* grid;
    const uint32_t* morton_lut;   // optional (LDS): lz_expand_bits(v) for v < H, set by kernels that stage one (lz_frame.hip); null = compute

    __device__ __forceinline__ void init(const float* o, const float* d, float bound_, float dt_gamma_, uint32_t max_steps,
                                         uint32_t C, uint32_t H_, const uint8_t* grid_) {
        init(o, d, 1 / d[0], 1 / d[1], 1 / d[2], bound_, dt_gamma_, max_steps, C, H_, grid_);
    }
    // the same with the three reciprocals of the direction supplied (the fused frame kernel computes them once per ray, when a slot takes
    // the ray, not once per pass: an IEEE division is ~11 vector instructions)
    __device__ __forceinline__ void init(const float* o, const float* d, float rdx_, float rdy_, float rdz_, float bound_, float dt_gamma_,
                                         uint32_t max_steps, uint32_t C, uint32_t H_, const uint8_t* grid_) {
        init(o, d, rdx_, rdy_, rdz_, bound_, dt_gamma_, lz_march_frame(bound_, max_steps, C, H_), C, H_, grid_);
    }
    // ... and with the frame's quotients supplied as well (lz_march_frame on the host: the same IEEE divisions, the same bits)
    __device__ __forceinline__ void init(const float* o, const float* d, float rdx_, float rdy_, float rdz_, float bound_, float dt_gamma_,
                                         const LzMarchFrame& fr, uint32_t C, uint32_t H_, const uint8_t* grid_) {
        ox = o[0]; oy = o[1]; oz = o[2];
        dx = d[0]; dy = d[1]; dz = d[2];
        rdx = rdx_; rdy = rdy_; rdz = rdz_;
        morton_lut = nullptr;
        bound = bound_; rbound = fr.rbound; dt_gamma = dt_gamma_;
        rH = fr.rH;
        H3 = (float)(H_ * H_ * H_);
        H = H_; fC = (float)C; fH = (float)H_; grid = grid_;
        one_cascade = C == 1;      // then both mip rules clamp to level 0 (raymarching.cu:42-54: min(max_cascade - 1, .)): nothing to evaluate
        pow2H = (H_ & (H_ - 1u)) == 0u && H_ >= 2u;
        halfH = 0.5f * (float)H_;
        dt_max = fr.dt_max;
        dt_min = fr.dt_min;
    }

    // the step of the march at t -- sample step and empty-space skip alike (raymarching.cu:907, 919-926)
    __device__ __forceinline__ float step_at(float t) const { return lz_clampf(t * dt_gamma, dt_min, dt_max); }

    // where the march stands at t: clamped position, step, and the cell it tests (raymarching.cu:876-895)
    struct Cell {
        float x, y, z, dt, mip_bound;
        int nx, ny, nz;
        uint32_t index;      // bit index into the density bitfield
    };
    __device__ __forceinline__ void locate(float tt0, Cell& c) const {
        c.x = lz_clampf(lz_fmaf(tt0, dx, ox), -bound, bound);
        c.y = lz_clampf(lz_fmaf(tt0, dy, oy), -bound, bound);
        c.z = lz_clampf(lz_fmaf(tt0, dz, oz), -bound, bound);
        c.dt = lz_clampf(tt0 * dt_gamma, dt_min, dt_max);
        int level = 0;
        float mip_rbound;
        if (one_cascade) {          // wave-uniform; the same values the general branch yields for level = 0, without its ~20 instructions
            c.mip_bound = lz_fminf(1.0f, bound);
            mip_rbound = (1.0f <= bound) ? 1.0f : rbound;
        } else {
            const int lp = lz_mip_from_pos(c.x, c.y, c.z, fC), ld = lz_mip_from_dt(c.dt, fH, fC);
            level = lp > ld ? lp : ld;
            const float lb = lz_scalbnf(1.0f, level);
            c.mip_bound = lz_fminf(lb, bound);
            // 1 / mip_bound without a division per probe: the reciprocal of 2^level is exact, the one of `bound` is hoisted to init()
            mip_rbound = (lb <= bound) ? lz_scalbnf(1.0f, -level) : rbound;
        }
        const float hm1 = (float)(H - 1);
        // raymarching.cu:415-417 evaluates 0.5 * (x * mip_rbound + 1) * H in double and narrows.  For a power-of-two H (the reference
        // hard-codes 128, renderer.py:94) the two multiplications are exact in float as well, so both evaluations give the same bits
        // and the f64 pipe (half rate, plus four conversions per axis) stays out of the march; any other H takes the double path.
        if (pow2H) {
            c.nx = (int)lz_clampf(lz_fmaf(c.x, mip_rbound, 1.0f) * halfH, 0.0f, hm1);
            c.ny = (int)lz_clampf(lz_fmaf(c.y, mip_rbound, 1.0f) * halfH, 0.0f, hm1);
            c.nz = (int)lz_clampf(lz_fmaf(c.z, mip_rbound, 1.0f) * halfH, 0.0f, hm1);
        } else {
            c.nx = (int)lz_clampf((float)(0.5 * (double)lz_fmaf(c.x, mip_rbound, 1.0f) * (double)H), 0.0f, hm1);
            c.ny = (int)lz_clampf((float)(0.5 * (double)lz_fmaf(c.y, mip_rbound, 1.0f) * (double)H), 0.0f, hm1);
            c.nz = (int)lz_clampf((float)(0.5 * (double)lz_fmaf(c.z, mip_rbound, 1.0f) * (double)H), 0.0f, hm1);
        }
        const uint32_t mort = morton_lut ? (morton_lut[c.nx] | (morton_lut[c.ny] << 1) | (morton_lut[c.nz] << 2))      // three LDS reads for 24 vector instructions
                                         : lz_morton3((uint32_t)c.nx, (uint32_t)c.ny, (uint32_t)c.nz);
        c.index = (uint32_t)((float)level * H3 + (float)mort);
    }
    __device__ __forceinline__ int occupied(const Cell& c) const { return grid[c.index / 8] & (1 << (c.index % 8)); }
    // t at which the ray leaves the (empty) cell it is in at tt0 (raymarching.cu:919-924)
    __device__ __forceinline__ float exit_t(float tt0, const Cell& c) const {
        const float tx = lz_fmaf(lz_fmaf(((float)c.nx + 0.5f + 0.5f * lz_signf(dx)) * rH, 2.0f, -1.0f), c.mip_bound, -c.x) * rdx;
        const float ty = lz_fmaf(lz_fmaf(((float)c.ny + 0.5f + 0.5f * lz_signf(dy)) * rH, 2.0f, -1.0f), c.mip_bound, -c.y) * rdy;
        const float tz = lz_fmaf(lz_fmaf(((float)c.nz + 0.5f + 0.5f * lz_signf(dz)) * rH, 2.0f, -1.0f), c.mip_bound, -c.z) * rdz;
        return tt0 + lz_fmaxf(0.0f, lz_fminf(tx, lz_fminf(ty, tz)));
    }

    // (kept as ONE function next to locate / occupied / exit_t, which restate it in pieces for the batched march of lz_frame.hip: split into
    // calls the f32 frame kernel, 130 instructions of whose every pass this is, measured 1 % slower)
    // 1: cell occupied (x, y, z, dt describe the sample, caller advances t by dt); 0: t advanced past the empty cell
    __device__ __forceinline__ int probe(float& t, float& x, float& y, float& z, float& dt) const {
        const float tt0 = t;
        x = lz_clampf(lz_fmaf(tt0, dx, ox), -bound, bound);
        y = lz_clampf(lz_fmaf(tt0, dy, oy), -bound, bound);
        z = lz_clampf(lz_fmaf(tt0, dz, oz), -bound, bound);
        dt = lz_clampf(tt0 * dt_gamma, dt_min, dt_max);
        int level = 0;
        float mip_bound, mip_rbound;
        if (one_cascade) {          // wave-uniform; the same values the general branch yields for level = 0, without its ~20 instructions
            mip_bound = lz_fminf(1.0f, bound);
            mip_rbound = (1.0f <= bound) ? 1.0f : rbound;
        } else {
            const int lp = lz_mip_from_pos(x, y, z, fC), ld = lz_mip_from_dt(dt, fH, fC);
            level = lp > ld ? lp : ld;
            const float lb = lz_scalbnf(1.0f, level);
            mip_bound = lz_fminf(lb, bound);
            // 1 / mip_bound without a division per probe: the reciprocal of 2^level is exact, the one of `bound` is hoisted to init()
            mip_rbound = (lb <= bound) ? lz_scalbnf(1.0f, -level) : rbound;
        }
        const float hm1 = (float)(H - 1);
        // raymarching.cu:415-417 evaluates 0.5 * (x * mip_rbound + 1) * H in double and narrows.  For a power-of-two H (the reference
        // hard-codes 128, renderer.py:94) the two multiplications are exact in float as well, so both evaluations give the same bits
        // and the f64 pipe (half rate, plus four conversions per axis) stays out of the march; any other H takes the double path.
        int nx, ny, nz;
        if (pow2H) {
            nx = (int)lz_clampf(lz_fmaf(x, mip_rbound, 1.0f) * halfH, 0.0f, hm1);
            ny = (int)lz_clampf(lz_fmaf(y, mip_rbound, 1.0f) * halfH, 0.0f, hm1);
            nz = (int)lz_clampf(lz_fmaf(z, mip_rbound, 1.0f) * halfH, 0.0f, hm1);
        } else {
            nx = (int)lz_clampf((float)(0.5 * (double)lz_fmaf(x, mip_rbound, 1.0f) * (double)H), 0.0f, hm1);
            ny = (int)lz_clampf((float)(0.5 * (double)lz_fmaf(y, mip_rbound, 1.0f) * (double)H), 0.0f, hm1);
            nz = (int)lz_clampf((float)(0.5 * (double)lz_fmaf(z, mip_rbound, 1.0f) * (double)H), 0.0f, hm1);
        }
        const uint32_t mort = morton_lut ? (morton_lut[nx] | (morton_lut[ny] << 1) | (morton_lut[nz] << 2))      // three LDS reads for 24 vector instructions
                                         : lz_morton3((uint32_t)nx, (uint32_t)ny, (uint32_t)nz);
        const uint32_t index = (uint32_t)((float)level * H3 + (float)mort);
        const int occ = grid[index / 8] & (1 << (index % 8));
        if (occ) return 1;
        const float tx = lz_fmaf(lz_fmaf(((float)nx + 0.5f + 0.5f * lz_signf(dx)) * rH, 2.0f, -1.0f), mip_bound, -x) * rdx;
        const float ty = lz_fmaf(lz_fmaf(((float)ny + 0.5f + 0.5f * lz_signf(dy)) * rH, 2.0f, -1.0f), mip_bound, -y) * rdy;
        const float tz = lz_fmaf(lz_fmaf(((float)nz + 0.5f + 0.5f * lz_signf(dz)) * rH, 2.0f, -1.0f), mip_bound, -z) * rdz;
        const float tt = tt0 + lz_fmaxf(0.0f, lz_fminf(tx, lz_fminf(ty, tz)));
        float tc = tt0;
        do { tc += lz_clampf(tc * dt_gamma, dt_min, dt_max); } while (tc < tt);
        t = tc;
        return 0;
    }
};

#endif
