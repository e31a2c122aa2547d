/*
 * oracle/raymarch_oracle.c -- TEST INFRASTRUCTURE ONLY (CPU checker).
 *
 * CPU restatement of the reference's ray-marching extension, /root/reference/raymarching/src/raymarching.cu:
 *   helpers            :19-81    near_far_from_aabb :91-145   sph_from_ray :162-198
 *   morton3D(+invert)  :214-254  packbits :267-289            morton3D_dilation :304-335
 *   march_rays_train   :352-518  (+backward :535-583)         march_rays (inference) :827-929
 *   compositing, five channel variants sharing one loop shape:
 *     train fwd/bwd  plain :603-809, sigma :1161-1367, uncertainty :1506-1734, triplane :1877-2122
 *     inference      plain :942-1029, ambient :1042-1136, ambient_sigma :1386-1480,
 *                    uncertainty :1753-1854, triplane :2141-2249
 *
 * Parity status: "parity unpinned" against the CUDA binary (no nvcc, no tests or vectors in the
 * reference); pinned by analytic known-answer tests (tests/test_oracle_*.py).
 *
 * Floating point: FMA where nvcc (-fmad=true default, raymarching/setup.py) contracts `a*b+c` within
 * one expression, written as lz_fmaf(); the double-typed sub-expressions of the reference
 * (`0.5 * (...) * H` :415, `dt * H * 0.5` :50) are kept in double; `level * H3` is float arithmetic
 * (:380,:419).  exp() is lz_expf (shared deterministic implementation, see lzzx_detmath.h) where the
 * reference calls __expf.
 *
 * Variant descriptor for compositing: n_amb in {0,1,2} ambient channels, amb_weighted (the *_sigma
 * variants weight the ambient channel by w, all others add it unweighted), has_unc (uncertainty
 * channel, always weighted).
 */
#include <float.h>
#include <math.h>
#include <stdint.h>
#include <string.h>
#include "lzzx_detmath.h"

/* `weight = alpha * T; weight_sum += weight;` (raymarching.cu:2203-2206 and siblings): two statements.  The checker reads nvcc's default
 * -fmad=true as "a * b + c inside ONE expression becomes an fma" and keeps these two roundings; a compiler that fuses across statements
 * (NVPTX-style) may emit fma(alpha, T, weight_sum) for the sum while still materialising `weight`.  That reading is undecidable without
 * nvcc, so the SECOND checker library (oracle/_build/liblzzx_oracle_fast.so: -DLZO_CONTRACT_ACROSS_STATEMENTS -ffp-contract=fast -mfma)
 * takes the other side of every such choice, and tests/test_fma_contraction_bound.py measures how far apart the two land. */
#ifdef LZO_CONTRACT_ACROSS_STATEMENTS
#define LZO_ACC_PRODUCT(a, b, product, acc) lz_fmaf((a), (b), (acc))
#else
#define LZO_ACC_PRODUCT(a, b, product, acc) ((acc) + (product))
#endif

#define SQRT3F 1.7320508075688772f
#define RPIF 0.3183098861837907f

static inline uint32_t expand_bits(uint32_t v) { /* :56-63 */
    v = (v * 0x00010001u) & 0xFF0000FFu;
    v = (v * 0x00000101u) & 0x0F00F00Fu;
    v = (v * 0x00000011u) & 0xC30C30C3u;
    v = (v * 0x00000005u) & 0x49249249u;
    return v;
}
static inline uint32_t morton3(uint32_t x, uint32_t y, uint32_t z) {
    return expand_bits(x) | (expand_bits(y) << 1) | (expand_bits(z) << 2);
}
static inline uint32_t morton3_inv(uint32_t x) { /* :73-81 */
    x = x & 0x49249249u;
    x = (x | (x >> 2)) & 0xc30c30c3u;
    x = (x | (x >> 4)) & 0x0f00f00fu;
    x = (x | (x >> 8)) & 0xff0000ffu;
    x = (x | (x >> 16)) & 0x0000ffffu;
    return x;
}

static inline int mip_from_pos(float x, float y, float z, float max_cascade) { /* :42-47 */
    const float mx = lz_fmaxf(lz_fabsf(x), lz_fmaxf(lz_fabsf(y), lz_fabsf(z)));
    const int exponent = lz_frexp_exp(mx);
    return (int)lz_fminf(max_cascade - 1, lz_fmaxf(0, (float)exponent));
}
static inline int mip_from_dt(float dt, float H, float max_cascade) { /* :49-54 */
    const float mx = (float)((double)(dt * H) * 0.5);
    const int exponent = lz_frexp_exp(mx);
    return (int)lz_fminf(max_cascade - 1, lz_fmaxf(0, (float)exponent));
}

/* Ray generation, /root/reference/nerf_triplane/utils.py:226-312, B poses x N pixels.
 * Pixel p = row * W + col (inds[n], or n itself when inds is NULL = the N = -1 branch :287), centre (col + 0.5, row + 0.5) (:243-244);
 * dir = ((i-cx)/fx, (j-cy)/fy, 1) / |.| (:297-301); rays_d[k] = sum_c dir[c] * R[k][c] (`directions @ R^T`, :303) accumulated
 * c = 0,1,2 as an fma chain (torch.matmul fixes no order); rays_o = pose[:3, 3] (:305-306).  poses: row-major 4x4 cam2world.
 * out_i / out_j (optional): results['i'], results['j'] (:290-291). */
void lzo_get_rays(const float* poses, float fx, float fy, float cx, float cy, uint32_t H, uint32_t W, uint32_t B, uint32_t N,
                  const int64_t* inds, float* rays_o, float* rays_d, float* out_i, float* out_j) {
    (void)H;
#pragma omp parallel for schedule(static)
    for (int64_t t = 0; t < (int64_t)B * N; t++) {
        const uint32_t b = (uint32_t)(t / N), n = (uint32_t)(t % N);
        const uint32_t p = inds ? (uint32_t)inds[n] : n;
        const float* pose = poses + (size_t)b * 16;
        const float fi = (float)(p % W) + 0.5f, fj = (float)(p / W) + 0.5f;
        const float xs = (fi - cx) / fx, ys = (fj - cy) / fy, zs = 1.0f;
        const float nrm = sqrtf(lz_fmaf(zs, zs, lz_fmaf(ys, ys, xs * xs)));
        const float d0 = xs / nrm, d1 = ys / nrm, d2 = zs / nrm;
        for (int k = 0; k < 3; k++) {
            rays_d[(size_t)t * 3 + k] = lz_fmaf(d2, pose[k * 4 + 2], lz_fmaf(d1, pose[k * 4 + 1], d0 * pose[k * 4 + 0]));
            rays_o[(size_t)t * 3 + k] = pose[k * 4 + 3];
        }
        if (out_i) out_i[t] = fi;
        if (out_j) out_j[t] = fj;
    }
}

/* get_bg_coords, utils.py:217-223: torch evaluates arange(H) / (H-1) * 2 - 1 in float32, one IEEE operation per step */
void lzo_bg_coords(uint32_t H, uint32_t W, float* out) {
    for (uint32_t p = 0; p < H * W; p++) {
        out[(size_t)p * 2] = (float)(p / W) / (float)(H - 1) * 2.0f - 1.0f;
        out[(size_t)p * 2 + 1] = (float)(p % W) / (float)(W - 1) * 2.0f - 1.0f;
    }
}

void lzo_near_far_from_aabb(const float* rays_o, const float* rays_d, const float* aabb, uint32_t N,
                            float min_near, float* nears, float* fars) {
#pragma omp parallel for schedule(dynamic, 256)
    for (uint32_t n = 0; n < N; n++) {
        const float ox = rays_o[n * 3], oy = rays_o[n * 3 + 1], oz = rays_o[n * 3 + 2];
        const float rdx = 1 / rays_d[n * 3], rdy = 1 / rays_d[n * 3 + 1], rdz = 1 / rays_d[n * 3 + 2];
        float near = (aabb[0] - ox) * rdx, far = (aabb[3] - ox) * rdx;
        if (near > far) { float c = near; near = far; far = c; }
        float near_y = (aabb[1] - oy) * rdy, far_y = (aabb[4] - oy) * rdy;
        if (near_y > far_y) { float c = near_y; near_y = far_y; far_y = c; }
        if (near > far_y || near_y > far) { nears[n] = fars[n] = FLT_MAX; continue; }
        if (near_y > near) near = near_y;
        if (far_y < far) far = far_y;
        float near_z = (aabb[2] - oz) * rdz, far_z = (aabb[5] - oz) * rdz;
        if (near_z > far_z) { float c = near_z; near_z = far_z; far_z = c; }
        if (near > far_z || near_z > far) { nears[n] = fars[n] = FLT_MAX; continue; }
        if (near_z > near) near = near_z;
        if (far_z < far) far = far_z;
        if (near < min_near) near = min_near;
        nears[n] = near; fars[n] = far;
    }
}

/* :162-198; atan2f / sqrtf come from libm here, parity on this entry is tolerance-based */
void lzo_sph_from_ray(const float* rays_o, const float* rays_d, float radius, uint32_t N, float* coords) {
    for (uint32_t n = 0; n < N; n++) {
        const float ox = rays_o[n * 3], oy = rays_o[n * 3 + 1], oz = rays_o[n * 3 + 2];
        const float dx = rays_d[n * 3], dy = rays_d[n * 3 + 1], dz = rays_d[n * 3 + 2];
        const float A = lz_fmaf(dz, dz, lz_fmaf(dy, dy, dx * dx));
        const float B = lz_fmaf(oz, dz, lz_fmaf(oy, dy, ox * dx));
        const float C = lz_fmaf(oz, oz, lz_fmaf(oy, oy, ox * ox)) - radius * radius;
        const float t = (-B + sqrtf(lz_fmaf(B, B, -(A * C)))) / A;
        const float x = lz_fmaf(t, dx, ox), y = lz_fmaf(t, dy, oy), z = lz_fmaf(t, dz, oz);
        const float theta = atan2f(sqrtf(lz_fmaf(z, z, x * x)), y);
        const float phi = atan2f(z, x);
        coords[n * 2] = lz_fmaf(2 * theta, RPIF, -1.0f);
        coords[n * 2 + 1] = phi * RPIF;
    }
}

void lzo_morton3D(const int32_t* coords, uint32_t N, int32_t* indices) {
    for (uint32_t n = 0; n < N; n++)
        indices[n] = (int32_t)morton3((uint32_t)coords[n * 3], (uint32_t)coords[n * 3 + 1], (uint32_t)coords[n * 3 + 2]);
}

void lzo_morton3D_invert(const int32_t* indices, uint32_t N, int32_t* coords) {
    for (uint32_t n = 0; n < N; n++) {
        const int32_t ind = indices[n]; /* arithmetic shifts of a signed int, :249-253 */
        coords[n * 3] = (int32_t)morton3_inv((uint32_t)(ind >> 0));
        coords[n * 3 + 1] = (int32_t)morton3_inv((uint32_t)(ind >> 1));
        coords[n * 3 + 2] = (int32_t)morton3_inv((uint32_t)(ind >> 2));
    }
}

void lzo_packbits(const float* grid, uint32_t N, float density_thresh, uint8_t* bitfield) {
    for (uint32_t n = 0; n < N; n++) {
        uint8_t bits = 0;
        for (uint32_t i = 0; i < 8; i++) bits |= (grid[(size_t)n * 8 + i] > density_thresh) ? (uint8_t)(1u << i) : 0;
        bitfield[n] = bits;
    }
}

void lzo_morton3D_dilation(const float* grid, uint32_t C, uint32_t H, float* out) {
    const uint32_t H3 = H * H * H;
#pragma omp parallel for schedule(dynamic, 256)
    for (uint32_t n = 0; n < C * H3; n++) {
        const uint32_t c = n / H3, ind = n - c * H3;
        const uint32_t x = morton3_inv(ind >> 0), y = morton3_inv(ind >> 1), z = morton3_inv(ind >> 2);
        const float* g = grid + (size_t)c * H3;
        float res = grid[n];
        if (x + 1 < H) res = lz_fmaxf(res, g[morton3(x + 1, y, z)]);
        if (x > 0) res = lz_fmaxf(res, g[morton3(x - 1, y, z)]);
        if (y + 1 < H) res = lz_fmaxf(res, g[morton3(x, y + 1, z)]);
        if (y > 0) res = lz_fmaxf(res, g[morton3(x, y - 1, z)]);
        if (z + 1 < H) res = lz_fmaxf(res, g[morton3(x, y, z + 1)]);
        if (z > 0) res = lz_fmaxf(res, g[morton3(x, y, z - 1)]);
        out[n] = res;
    }
}

/* ---- one marching probe: position, step, occupancy lookup, empty-space skip (:400-440, :875-927) ---- */
typedef struct {
    float ox, oy, oz, dx, dy, dz, rdx, rdy, rdz;
    float bound, dt_gamma, dt_min, dt_max, rH, H3, fC, fH;
    uint32_t H;
    const uint8_t* grid;
} march_ctx;

static void march_ctx_init(march_ctx* m, const float* o, const float* d, float bound, float dt_gamma,
                           uint32_t max_steps, uint32_t C, uint32_t H, const uint8_t* grid) {
    m->ox = o[0]; m->oy = o[1]; m->oz = o[2];
    m->dx = d[0]; m->dy = d[1]; m->dz = d[2];
    m->rdx = 1 / m->dx; m->rdy = 1 / m->dy; m->rdz = 1 / m->dz;
    m->bound = bound; m->dt_gamma = dt_gamma;
    m->rH = 1 / (float)H;
    m->H3 = (float)(H * H * H);
    m->H = H; m->fC = (float)C; m->fH = (float)H; m->grid = grid;
    m->dt_max = 2 * SQRT3F * (float)(1 << (C - 1)) / (float)H;
    m->dt_min = lz_fminf(m->dt_max, 2 * SQRT3F / (float)max_steps);
}

/* returns 1 when the cell at t is occupied (then *x,*y,*z,*dt describe the sample; caller advances t),
 * else 0 after advancing *t past the empty cell. */
static int march_probe(const march_ctx* m, float* t, float* x, float* y, float* z, float* dt) {
    const float tt0 = *t;
    *x = lz_clampf(lz_fmaf(tt0, m->dx, m->ox), -m->bound, m->bound);
    *y = lz_clampf(lz_fmaf(tt0, m->dy, m->oy), -m->bound, m->bound);
    *z = lz_clampf(lz_fmaf(tt0, m->dz, m->oz), -m->bound, m->bound);
    *dt = lz_clampf(tt0 * m->dt_gamma, m->dt_min, m->dt_max);
    const int lp = mip_from_pos(*x, *y, *z, m->fC), ld = mip_from_dt(*dt, m->fH, m->fC);
    const int level = lp > ld ? lp : ld;
    const float mip_bound = lz_fminf(lz_scalbnf(1.0f, level), m->bound);
    const float mip_rbound = 1 / mip_bound;
    const float hm1 = (float)(m->H - 1);
    const int nx = (int)lz_clampf((float)(0.5 * (double)lz_fmaf(*x, mip_rbound, 1.0f) * (double)m->H), 0.0f, hm1);
    const int ny = (int)lz_clampf((float)(0.5 * (double)lz_fmaf(*y, mip_rbound, 1.0f) * (double)m->H), 0.0f, hm1);
    const int nz = (int)lz_clampf((float)(0.5 * (double)lz_fmaf(*z, mip_rbound, 1.0f) * (double)m->H), 0.0f, hm1);
    const uint32_t index = (uint32_t)((float)level * m->H3 + (float)morton3((uint32_t)nx, (uint32_t)ny, (uint32_t)nz));
    const int occ = m->grid[index / 8] & (1 << (index % 8));
    if (occ) return 1;
    const float tx = lz_fmaf(lz_fmaf(((float)nx + 0.5f + 0.5f * lz_signf(m->dx)) * m->rH, 2.0f, -1.0f), mip_bound, -*x) * m->rdx;
    const float ty = lz_fmaf(lz_fmaf(((float)ny + 0.5f + 0.5f * lz_signf(m->dy)) * m->rH, 2.0f, -1.0f), mip_bound, -*y) * m->rdy;
    const float tz = lz_fmaf(lz_fmaf(((float)nz + 0.5f + 0.5f * lz_signf(m->dz)) * m->rH, 2.0f, -1.0f), mip_bound, -*z) * m->rdz;
    const float tt = tt0 + lz_fmaxf(0.0f, lz_fminf(tx, lz_fminf(ty, tz)));
    float tc = tt0;
    do { tc += lz_clampf(tc * m->dt_gamma, m->dt_min, m->dt_max); } while (tc < tt);
    *t = tc;
    return 0;
}

/* :352-518.  Ray rows are emitted in ray order (the one arrival order of the reference's atomics that
 * is reproducible); rays whose samples would overflow M keep their row but write nothing (:457). */
void lzo_march_rays_train(const float* rays_o, const float* rays_d, const uint8_t* grid, float bound, float dt_gamma,
                          uint32_t max_steps, uint32_t N, uint32_t C, uint32_t H, uint32_t M,
                          const float* nears, const float* fars, float* xyzs, float* dirs, float* deltas,
                          int32_t* rays, int32_t* counter, const float* noises) {
    for (uint32_t n = 0; n < N; n++) {
        march_ctx m;
        march_ctx_init(&m, rays_o + n * 3, rays_d + n * 3, bound, dt_gamma, max_steps, C, H, grid);
        const float far = fars[n];
        float t0 = nears[n];
        t0 = lz_fmaf(lz_clampf(t0 * dt_gamma, m.dt_min, m.dt_max), noises[n], t0);
        float t = t0, x, y, z, dt;
        uint32_t num_steps = 0;
        while (t < far && num_steps < max_steps) {
            if (march_probe(&m, &t, &x, &y, &z, &dt)) { num_steps++; t += dt; }
        }
        const uint32_t point_index = (uint32_t)counter[0];
        const uint32_t ray_index = (uint32_t)counter[1];
        counter[0] += (int32_t)num_steps;
        counter[1] += 1;
        rays[ray_index * 3] = (int32_t)n;
        rays[ray_index * 3 + 1] = (int32_t)point_index;
        rays[ray_index * 3 + 2] = (int32_t)num_steps;
        if (num_steps == 0) continue;
        if (point_index + num_steps > M) continue;
        float* px = xyzs + (size_t)point_index * 3;
        float* pd = dirs + (size_t)point_index * 3;
        float* pl = deltas + (size_t)point_index * 2;
        t = t0;
        uint32_t step = 0;
        while (t < far && step < num_steps) {
            if (march_probe(&m, &t, &x, &y, &z, &dt)) {
                px[0] = x; px[1] = y; px[2] = z;
                pd[0] = m.dx; pd[1] = m.dy; pd[2] = m.dz;
                t += dt;
                pl[0] = dt; pl[1] = t;
                px += 3; pd += 3; pl += 2; step++;
            }
        }
    }
}

/* :535-583 */
void lzo_march_rays_train_backward(const float* grad_xyzs, const float* grad_dirs, const int32_t* rays,
                                   const float* deltas, uint32_t N, uint32_t M, float* grad_rays_o, float* grad_rays_d) {
    for (uint32_t n = 0; n < N; n++) {
        const uint32_t offset = (uint32_t)rays[n * 3 + 1], num_steps = (uint32_t)rays[n * 3 + 2];
        if (num_steps == 0 || offset + num_steps > M) continue;
        float* go = grad_rays_o + n * 3; float* gd = grad_rays_d + n * 3;
        for (uint32_t s = 0; s < num_steps; s++) {
            const float* gx = grad_xyzs + (size_t)(offset + s) * 3;
            const float* gdi = grad_dirs + (size_t)(offset + s) * 3;
            const float tt = deltas[(size_t)(offset + s) * 2 + 1];
            for (int k = 0; k < 3; k++) {
                go[k] += gx[k];
                gd[k] += lz_fmaf(gx[k], tt, gdi[k]);
            }
        }
    }
}

/* :827-929.  xyzs/dirs/deltas must be zero-filled by the caller (raymarching.py:384-386). */
void lzo_march_rays(uint32_t n_alive, uint32_t n_step, const int32_t* rays_alive, const float* rays_t,
                    const float* rays_o, const float* rays_d, float bound, float dt_gamma, uint32_t max_steps,
                    uint32_t C, uint32_t H, const uint8_t* grid, const float* nears, const float* fars,
                    float* xyzs, float* dirs, float* deltas, const float* noises) {
    (void)nears;
#pragma omp parallel for schedule(dynamic, 256)
    for (uint32_t n = 0; n < n_alive; n++) {
        const int32_t index = rays_alive[n];
        march_ctx m;
        march_ctx_init(&m, rays_o + (size_t)index * 3, rays_d + (size_t)index * 3, bound, dt_gamma, max_steps, C, H, grid);
        float* px = xyzs + (size_t)n * n_step * 3;
        float* pd = dirs + (size_t)n * n_step * 3;
        float* pl = deltas + (size_t)n * n_step * 2;
        float t = rays_t[index];
        const float far = fars[index];
        t = lz_fmaf(lz_clampf(t * dt_gamma, m.dt_min, m.dt_max), noises[n], t);
        uint32_t step = 0;
        float x, y, z, dt;
        while (t < far && step < n_step) {
            if (march_probe(&m, &t, &x, &y, &z, &dt)) {
                px[0] = x; px[1] = y; px[2] = z;
                pd[0] = m.dx; pd[1] = m.dy; pd[2] = m.dz;
                t += dt;
                pl[0] = dt; pl[1] = t;
                px += 3; pd += 3; pl += 2; step++;
            }
        }
    }
}

/* ---- compositing, training (:603-687 and siblings).  amb0/amb1/unc may be NULL per variant. ---- */
void lzo_composite_rays_train_forward(const float* sigmas, const float* rgbs, const float* amb0, const float* amb1,
                                      const float* unc, const float* deltas, const int32_t* rays,
                                      uint32_t M, uint32_t N, float T_thresh, int n_amb, int amb_weighted, int has_unc,
                                      float* weights_sum, float* amb0_sum, float* amb1_sum, float* unc_sum,
                                      float* depth, float* image) {
#pragma omp parallel for schedule(dynamic, 256)
    for (uint32_t n = 0; n < N; n++) {
        const uint32_t index = (uint32_t)rays[n * 3], offset = (uint32_t)rays[n * 3 + 1], num_steps = (uint32_t)rays[n * 3 + 2];
        float T = 1.0f, r = 0, g = 0, b = 0, ws = 0, d = 0, a0 = 0, a1 = 0, u = 0;
        if (!(num_steps == 0 || offset + num_steps > M)) {
            for (uint32_t step = 0; step < num_steps; step++) {
                const size_t i = (size_t)offset + step;
                const float alpha = 1.0f - lz_expf(-sigmas[i] * deltas[i * 2]);
                const float weight = alpha * T;
                r = lz_fmaf(weight, rgbs[i * 3], r);
                g = lz_fmaf(weight, rgbs[i * 3 + 1], g);
                b = lz_fmaf(weight, rgbs[i * 3 + 2], b);
                d = lz_fmaf(weight, deltas[i * 2 + 1], d);
                ws = LZO_ACC_PRODUCT(alpha, T, weight, ws);
                if (n_amb > 0) a0 = amb_weighted ? lz_fmaf(weight, amb0[i], a0) : a0 + amb0[i];
                if (n_amb > 1) a1 = amb_weighted ? lz_fmaf(weight, amb1[i], a1) : a1 + amb1[i];
                if (has_unc) u = lz_fmaf(weight, unc[i], u);
                T *= 1.0f - alpha;
                if (T < T_thresh) break;
            }
        }
        weights_sum[index] = ws;
        if (n_amb > 0) amb0_sum[index] = a0;
        if (n_amb > 1) amb1_sum[index] = a1;
        if (has_unc) unc_sum[index] = u;
        depth[index] = d;
        image[index * 3] = r; image[index * 3 + 1] = g; image[index * 3 + 2] = b;
    }
}

/* :711-809 and siblings.  grad_* outputs must be zero-filled by the caller (raymarching.py:332-334). */
void lzo_composite_rays_train_backward(const float* grad_weights_sum, const float* grad_amb0_sum, const float* grad_amb1_sum,
                                       const float* grad_unc_sum, const float* grad_image,
                                       const float* sigmas, const float* rgbs, const float* amb0, const float* amb1,
                                       const float* unc, const float* deltas, const int32_t* rays,
                                       const float* weights_sum, const float* amb0_sum, const float* unc_sum, const float* image,
                                       uint32_t M, uint32_t N, float T_thresh, int n_amb, int amb_weighted, int has_unc,
                                       float* grad_sigmas, float* grad_rgbs, float* grad_amb0, float* grad_amb1, float* grad_unc) {
    (void)amb1;
#pragma omp parallel for schedule(dynamic, 256)
    for (uint32_t n = 0; n < N; n++) {
        const uint32_t index = (uint32_t)rays[n * 3], offset = (uint32_t)rays[n * 3 + 1], num_steps = (uint32_t)rays[n * 3 + 2];
        if (num_steps == 0 || offset + num_steps > M) continue;
        const float gi0 = grad_image[index * 3], gi1 = grad_image[index * 3 + 1], gi2 = grad_image[index * 3 + 2];
        const float gws = grad_weights_sum[index];
        const float r_final = image[index * 3], g_final = image[index * 3 + 1], b_final = image[index * 3 + 2];
        const float ws_final = weights_sum[index];
        const float amb_final = (n_amb > 0 && amb_weighted) ? amb0_sum[index] : 0.0f;
        const float unc_final = has_unc ? unc_sum[index] : 0.0f;
        float T = 1.0f, r = 0, g = 0, b = 0, ws = 0, amb = 0, u = 0;
        for (uint32_t step = 0; step < num_steps; step++) {
            const size_t i = (size_t)offset + step;
            const float alpha = 1.0f - lz_expf(-sigmas[i] * deltas[i * 2]);
            const float weight = alpha * T;
            r = lz_fmaf(weight, rgbs[i * 3], r);
            g = lz_fmaf(weight, rgbs[i * 3 + 1], g);
            b = lz_fmaf(weight, rgbs[i * 3 + 2], b);
            if (n_amb > 0 && amb_weighted) amb = lz_fmaf(weight, amb0[i], amb);
            if (has_unc) u = lz_fmaf(weight, unc[i], u);
            ws = LZO_ACC_PRODUCT(alpha, T, weight, ws);
            T *= 1.0f - alpha;
            grad_rgbs[i * 3] = gi0 * weight;
            grad_rgbs[i * 3 + 1] = gi1 * weight;
            grad_rgbs[i * 3 + 2] = gi2 * weight;
            if (n_amb > 0) grad_amb0[i] = amb_weighted ? grad_amb0_sum[index] * weight : grad_amb0_sum[index];
            if (n_amb > 1) grad_amb1[i] = grad_amb1_sum[index];
            if (has_unc) grad_unc[i] = grad_unc_sum[index] * weight;
            float s = gi0 * lz_fmaf(T, rgbs[i * 3], -(r_final - r));
            s = lz_fmaf(gi1, lz_fmaf(T, rgbs[i * 3 + 1], -(g_final - g)), s);
            s = lz_fmaf(gi2, lz_fmaf(T, rgbs[i * 3 + 2], -(b_final - b)), s);
            if (n_amb > 0 && amb_weighted) s = lz_fmaf(grad_amb0_sum[index], lz_fmaf(T, amb0[i], -(amb_final - amb)), s);
            if (has_unc) s = lz_fmaf(grad_unc_sum[index], lz_fmaf(T, unc[i], -(unc_final - u)), s);
            s = lz_fmaf(gws, 1 - ws_final, s);
            grad_sigmas[i] = deltas[i * 2] * s;
            if (T < T_thresh) break;
        }
    }
}

/* ---- compositing, inference (:942-1029 and siblings): in-place on the per-ray accumulators ---- */
void lzo_composite_rays(uint32_t n_alive, uint32_t n_step, float T_thresh, int32_t* rays_alive, float* rays_t,
                        const float* sigmas, const float* rgbs, const float* deltas,
                        const float* amb0, const float* amb1, const float* unc,
                        int n_amb, int amb_weighted, int has_unc,
                        float* weights_sum, float* depth, float* image, float* amb0_sum, float* amb1_sum, float* unc_sum) {
#pragma omp parallel for schedule(dynamic, 256)
    for (uint32_t n = 0; n < n_alive; n++) {
        const int32_t index = rays_alive[n];
        float t = rays_t[index];
        float weight_sum = weights_sum[index], d = depth[index];
        float r = image[index * 3], g = image[index * 3 + 1], b = image[index * 3 + 2];
        float a0 = n_amb > 0 ? amb0_sum[index] : 0, a1 = n_amb > 1 ? amb1_sum[index] : 0, u = has_unc ? unc_sum[index] : 0;
        uint32_t step = 0;
        while (step < n_step) {
            const size_t i = (size_t)n * n_step + step;
            if (deltas[i * 2] == 0) break;
            const float alpha = 1.0f - lz_expf(-sigmas[i] * deltas[i * 2]);
            const float T = 1 - weight_sum;
            const float weight = alpha * T;
            weight_sum = LZO_ACC_PRODUCT(alpha, T, weight, weight_sum);
            t = deltas[i * 2 + 1];
            d = lz_fmaf(weight, t, d);
            r = lz_fmaf(weight, rgbs[i * 3], r);
            g = lz_fmaf(weight, rgbs[i * 3 + 1], g);
            b = lz_fmaf(weight, rgbs[i * 3 + 2], b);
            if (n_amb > 0) a0 = amb_weighted ? lz_fmaf(weight, amb0[i], a0) : a0 + amb0[i];
            if (n_amb > 1) a1 = amb_weighted ? lz_fmaf(weight, amb1[i], a1) : a1 + amb1[i];
            if (has_unc) u = lz_fmaf(weight, unc[i], u);
            if (T < T_thresh) break;
            step++;
        }
        if (step < n_step) rays_alive[n] = -1; else rays_t[index] = t;
        weights_sum[index] = weight_sum;
        depth[index] = d;
        image[index * 3] = r; image[index * 3 + 1] = g; image[index * 3 + 2] = b;
        if (n_amb > 0) amb0_sum[index] = a0;
        if (n_amb > 1) amb1_sum[index] = a1;
        if (has_unc) unc_sum[index] = u;
    }
}
