// lz_torso.hip -- the torso branch of the frame: NeRFRenderer.run_torso (nerf_triplane/renderer.py:572-631)
// + NeRFNetwork.forward_torso (nerf_triplane/network.py:170-205) as ONE kernel, one lane per pixel.
//
// Reference per pixel: 2-D occupancy lookup (F.grid_sample, bilinear, align_corners) -> boolean mask -> for the masked pixels:
// shrink, frequency-encode (2 -> 34), cat with the frame-constant anchor encoding (42) and individual code, deform MLP
// (-> 32 -> 32 -> 2), x + dx clamped, tiled-grid encode (D = 2, L = 16, C = 2 -> 32), torso MLP (-> 32 -> 32 -> 4), sigmoids;
// scatter back into zero-filled [N,1] / [N,3] tensors, then mix with the background.  ~25 launches, a mask.any() sync and two
// boolean-mask gathers / scatters per frame.  Here nothing is materialised: a lane owns a pixel from lookup to alpha/colour.
//
// Arithmetic (restated bit for bit by oracle/torso.py): everything is an explicit f32 fma chain on the VALU (the MLPs are
// 5.4 kMAC per pixel: not worth the matrix cores at 2.6e5 pixels per frame).  Chain order of the two first layers: the
// frame-constant inputs (anchor encoding, individual code) first -- their partial sums are computed once per workgroup -- then
// the per-pixel inputs in natural order; all other layers natural order.
#include "lz_common.h"
#include "lzzx_detmath.h"

#define LZ_TORSO_FREQ 34      // 2 + 2 * 2 * 8   (get_encoder('frequency', input_dim=2, multires=8), network.py:160)
#define LZ_TORSO_ANCHOR 42    // 6 + 2 * 6 * 3   (input_dim=6, multires=3, network.py:162)
#define LZ_TORSO_GRIDF 32     // 16 levels x 2   (tiledgrid, network.py:166)
#define LZ_TORSO_HID 32
#define LZ_PI_F 3.141592653589793f

struct LzTorsoArgs {
    lz_torso_params p;
    float scale[16];
    uint32_t res[16];
};

// get_grid_index (gridencoder.cu:54-72), D = 2, generic form (tiled grids wrap with a true modulo)
__device__ __forceinline__ uint32_t lz_torso_grid_index(uint32_t gridtype, uint32_t hs, uint32_t resolution, uint32_t p0, uint32_t p1) {
    uint32_t stride = 1, index = 0;
    if (stride <= hs) { index += p0 * stride; stride *= resolution + 1; }
    if (stride <= hs) { index += p1 * stride; stride *= resolution + 1; }
    if (gridtype == 0 && stride > hs) index = p0 ^ (p1 * 2654435761u);
    return (index % hs) * 2u;
}

template <int IND>
__global__ void __launch_bounds__(256)
lz_k_torso_forward(LzTorsoArgs A, const float* __restrict__ bg_coords, uint32_t N, float* __restrict__ alpha_out,
                   float* __restrict__ color_out, float* __restrict__ deform_out) {
    constexpr int KC = LZ_TORSO_ANCHOR + IND;              // frame-constant inputs
    constexpr int K0 = LZ_TORSO_FREQ + KC;                 // deform net input width
    constexpr int K1 = LZ_TORSO_GRIDF + K0;                // torso net input width
    constexpr int H = LZ_TORSO_HID;
    // LDS: per-pixel parts of the weights + the constant partial sums
    __shared__ float d0[H * LZ_TORSO_FREQ], d1[H * H], d2[2 * H], t0[H * (LZ_TORSO_GRIDF + LZ_TORSO_FREQ)], t1[H * H], t2[4 * H];
    __shared__ float cd[H], ct[H];
    const lz_torso_params& P = A.p;
    for (int i = threadIdx.x; i < H * LZ_TORSO_FREQ; i += blockDim.x) d0[i] = P.deform_w0[(i / LZ_TORSO_FREQ) * K0 + i % LZ_TORSO_FREQ];
    for (int i = threadIdx.x; i < H * H; i += blockDim.x) { d1[i] = P.deform_w1[i]; t1[i] = P.torso_w1[i]; }
    for (int i = threadIdx.x; i < 2 * H; i += blockDim.x) d2[i] = P.deform_w2[i];
    for (int i = threadIdx.x; i < 4 * H; i += blockDim.x) t2[i] = P.torso_w2[i];
    constexpr int KP = LZ_TORSO_GRIDF + LZ_TORSO_FREQ;
    for (int i = threadIdx.x; i < H * KP; i += blockDim.x) t0[i] = P.torso_w0[(i / KP) * K1 + i % KP];
    if (threadIdx.x < 2 * H) {   // constant partial sums: fma chain over [anchor 42 | ind] in natural order
        const int o = threadIdx.x % H;
        const bool tor = threadIdx.x >= H;
        const float* w = tor ? P.torso_w0 + (size_t)o * K1 + KP : P.deform_w0 + (size_t)o * K0 + LZ_TORSO_FREQ;
        float acc = 0.0f;
        for (int k = 0; k < LZ_TORSO_ANCHOR; k++) acc = lz_fmaf(w[k], P.enc_anchor[k], acc);
        for (int k = 0; k < IND; k++) acc = lz_fmaf(w[LZ_TORSO_ANCHOR + k], P.ind_code[k], acc);
        (tor ? ct : cd)[o] = acc;
    }
    __syncthreads();
    const uint32_t n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N) return;
    const float bx = bg_coords[(size_t)n * 2], by = bg_coords[(size_t)n * 2 + 1];
    // ---- 2-D occupancy (renderer.py:603-606): F.grid_sample(bilinear, zeros padding, align_corners=True) > thresh ----
    bool masked = true;
    if (P.density_grid) {
        const uint32_t G = P.G;
        const float ix = ((bx + 1.0f) / 2.0f) * (float)(G - 1), iy = ((by + 1.0f) / 2.0f) * (float)(G - 1);
        const float x0f = floorf(ix), y0f = floorf(iy);
        const int x0 = (int)x0f, y0 = (int)y0f, x1 = x0 + 1, y1 = y0 + 1;
        // corner weights as torch forms them: nw = (ix_se - ix) * (iy_se - iy), ... (GridSampler.cuh)
        const float x1f = x0f + 1.0f, y1f = y0f + 1.0f;
        const float nw = (x1f - ix) * (y1f - iy), ne = (ix - x0f) * (y1f - iy), sw = (x1f - ix) * (iy - y0f), se = (ix - x0f) * (iy - y0f);
        auto at = [&](int xx, int yy) { return (xx >= 0 && yy >= 0 && xx < (int)G && yy < (int)G) ? P.density_grid[(size_t)yy * G + xx] : 0.0f; };
        float occ = 0.0f;   // `out_acc += value * weight` in nw, ne, sw, se order, contracted to fma by nvcc's default -fmad=true
        occ = lz_fmaf(at(x0, y0), nw, occ);
        occ = lz_fmaf(at(x1, y0), ne, occ);
        occ = lz_fmaf(at(x0, y1), sw, occ);
        occ = lz_fmaf(at(x1, y1), se, occ);
        masked = occ > P.density_thresh;
    }
    if (!masked) {   // torso_alpha / torso_color stay zero for unmasked pixels (renderer.py:609-610)
        alpha_out[n] = 0.0f;
        color_out[(size_t)n * 3] = 0.0f; color_out[(size_t)n * 3 + 1] = 0.0f; color_out[(size_t)n * 3 + 2] = 0.0f;
        if (deform_out) { deform_out[(size_t)n * 2] = 0.0f; deform_out[(size_t)n * 2 + 1] = 0.0f; }
        return;
    }
    // ---- forward_torso (network.py:170-205) ----
    const float x[2] = {bx * P.torso_shrink, by * P.torso_shrink};
    float ex[LZ_TORSO_FREQ];   // freqencoder.cu:30-66: [x, sin(2^f x), cos(2^f x)]_f, cos as sin(. + pi/2)
    ex[0] = x[0]; ex[1] = x[1];
#pragma unroll
    for (int c = 2; c < LZ_TORSO_FREQ; c++) {
        const int col = c / 2 - 1, d = c % 2, freq = col / 2;
        ex[c] = lz_sinf(lz_scalbnf(x[d], freq) + (float)(col % 2) * (LZ_PI_F / 2));
    }
    float h1[H], h2[H];
#pragma unroll
    for (int o = 0; o < H; o++) {
        float acc = cd[o];
#pragma unroll
        for (int k = 0; k < LZ_TORSO_FREQ; k++) acc = lz_fmaf(d0[o * LZ_TORSO_FREQ + k], ex[k], acc);
        h1[o] = acc > 0.0f ? acc : 0.0f;
    }
#pragma unroll
    for (int o = 0; o < H; o++) {
        float acc = 0.0f;
#pragma unroll
        for (int k = 0; k < H; k++) acc = lz_fmaf(d1[o * H + k], h1[k], acc);
        h2[o] = acc > 0.0f ? acc : 0.0f;
    }
    float dx[2];
#pragma unroll
    for (int o = 0; o < 2; o++) {
        float acc = 0.0f;
#pragma unroll
        for (int k = 0; k < H; k++) acc = lz_fmaf(d2[o * H + k], h2[k], acc);
        dx[o] = acc;
    }
    // x = (x + dx).clamp(-1, 1); torso_encoder(x, bound=1) (network.py:193-195, grid.py:143)
    float gx[LZ_TORSO_GRIDF];
    {
        float u[2];
#pragma unroll
        for (int d = 0; d < 2; d++) u[d] = (lz_fminf(lz_fmaxf(x[d] + dx[d], -1.0f), 1.0f) + 1.0f) / 2.0f;
        const bool oob = u[0] < 0 || u[0] > 1 || u[1] < 0 || u[1] > 1;   // cannot happen after the clamp; kept for NaN-free parity
#pragma unroll
        for (int l = 0; l < 16; l++) {
            const uint32_t off0 = (uint32_t)P.offsets[l], hs = (uint32_t)P.offsets[l + 1] - off0;
            const float sc = A.scale[l];
            const uint32_t resolution = A.res[l];
            const float* g = P.emb + (size_t)off0 * 2;
            const float p0 = lz_fmaf(u[0], sc, 0.5f), p1 = lz_fmaf(u[1], sc, 0.5f);
            const uint32_t g0 = (uint32_t)floorf(p0), g1 = (uint32_t)floorf(p1);
            const float f0 = p0 - (float)g0, f1 = p1 - (float)g1;
            float r0 = 0.0f, r1 = 0.0f;
#pragma unroll
            for (int c = 0; c < 4; c++) {
                const float w = ((c & 1) ? f0 : 1 - f0) * ((c >> 1) ? f1 : 1 - f1);
                const uint32_t index = lz_torso_grid_index(P.gridtype, hs, resolution, g0 + (c & 1), g1 + (c >> 1));
                const float2 v = *reinterpret_cast<const float2*>(g + index);
                r0 = lz_fmaf(w, v.x, r0);
                r1 = lz_fmaf(w, v.y, r1);
            }
            gx[2 * l] = oob ? 0.0f : r0;
            gx[2 * l + 1] = oob ? 0.0f : r1;
        }
    }
    // torso net: [grid 32 | enc_x 34 | anchor | ind] -> 32 -> 32 -> 4
#pragma unroll
    for (int o = 0; o < H; o++) {
        float acc = ct[o];
#pragma unroll
        for (int k = 0; k < LZ_TORSO_GRIDF; k++) acc = lz_fmaf(t0[o * KP + k], gx[k], acc);
#pragma unroll
        for (int k = 0; k < LZ_TORSO_FREQ; k++) acc = lz_fmaf(t0[o * KP + LZ_TORSO_GRIDF + k], ex[k], acc);
        h1[o] = acc > 0.0f ? acc : 0.0f;
    }
#pragma unroll
    for (int o = 0; o < H; o++) {
        float acc = 0.0f;
#pragma unroll
        for (int k = 0; k < H; k++) acc = lz_fmaf(t1[o * H + k], h1[k], acc);
        h2[o] = acc > 0.0f ? acc : 0.0f;
    }
    float out4[4];
#pragma unroll
    for (int o = 0; o < 4; o++) {
        float acc = 0.0f;
#pragma unroll
        for (int k = 0; k < H; k++) acc = lz_fmaf(t2[o * H + k], h2[k], acc);
        out4[o] = lz_sigmoidf(acc) * 1.002f - 0.001f;   // network.py:202-203
    }
    alpha_out[n] = out4[0];
    color_out[(size_t)n * 3] = out4[1]; color_out[(size_t)n * 3 + 1] = out4[2]; color_out[(size_t)n * 3 + 2] = out4[3];
    if (deform_out) { deform_out[(size_t)n * 2] = dx[0]; deform_out[(size_t)n * 2 + 1] = dx[1]; }
}

// ---- the frame-constant anchor encoding (network.py:179-183) as one launch -------------------------------------------------------------
// wrapped = anchor_points @ inverse(pose^T): row i is pose^-1 . a_i; (x / w / z, y / w / z) per anchor; frequency encoding of the 6 values,
// degree 3 (lz_k_freq_forward's formula and order).  The 4 x 4 inverse is a Gauss-Jordan elimination with partial pivoting in double by
// one thread -- torch runs an LU factorisation in f32 through a dozen library launches for it; the results agree to a few f32 ulp.
__global__ void __launch_bounds__(64) lz_k_torso_anchor_encode(const float* __restrict__ pose, const float* __restrict__ anchors, float* __restrict__ enc) {
    __shared__ float w6[6];
    if (threadIdx.x == 0) {
        double a[4][8];
        for (int r = 0; r < 4; r++)
            for (int c = 0; c < 4; c++) { a[r][c] = (double)pose[r * 4 + c]; a[r][4 + c] = r == c ? 1.0 : 0.0; }
        for (int col = 0; col < 4; col++) {
            int piv = col;
            for (int r = col + 1; r < 4; r++)
                if (fabs(a[r][col]) > fabs(a[piv][col])) piv = r;
            if (piv != col)
                for (int c = 0; c < 8; c++) { const double t = a[col][c]; a[col][c] = a[piv][c]; a[piv][c] = t; }
            const double inv = 1.0 / a[col][col];
            for (int c = 0; c < 8; c++) a[col][c] *= inv;
            for (int r = 0; r < 4; r++) {
                if (r == col) continue;
                const double f = a[r][col];
                for (int c = 0; c < 8; c++) a[r][c] -= f * a[col][c];
            }
        }
        for (int i = 0; i < 3; i++) {
            double w[4];
            for (int r = 0; r < 4; r++) {
                w[r] = 0.0;
                for (int c = 0; c < 4; c++) w[r] += a[r][4 + c] * (double)anchors[i * 4 + c];
            }
            w6[2 * i] = (float)(w[0] / w[3] / w[2]);
            w6[2 * i + 1] = (float)(w[1] / w[3] / w[2]);
        }
    }
    __syncthreads();
    const uint32_t c = threadIdx.x;
    if (c < 42) {
        float v;
        if (c < 6) v = w6[c];
        else {
            const uint32_t col = c / 6 - 1, d = c % 6, freq = col / 2;
            const float phase = (float)(col % 2) * (3.141592653589793f / 2);
            v = lz_sinf(lz_scalbnf(w6[d], (int)freq) + phase);
        }
        enc[c] = v;
    }
}

extern "C" int lz_torso_anchor_encode(const float* pose, const float* anchor_points, float* enc_anchor, lz_stream_t stream) {
    LZ_REQUIRE(pose && anchor_points && enc_anchor, LZ_ERR_BAD_ARGUMENT, "torso_anchor_encode: null tensor");
    hipLaunchKernelGGL(lz_k_torso_anchor_encode, dim3(1), dim3(64), 0, lz_st(stream), pose, anchor_points, enc_anchor);
    LZ_CHECK_LAUNCH("torso_anchor_encode");
    return LZ_OK;
}

extern "C" int lz_torso_forward(const lz_torso_params* p, const float* bg_coords, uint32_t N, float* alpha, float* color, float* deform,
                                lz_stream_t stream) {
    if (N == 0) return LZ_OK;
    LZ_REQUIRE(p && bg_coords && alpha && color, LZ_ERR_BAD_ARGUMENT, "torso_forward: null tensor");
    LZ_REQUIRE(p->deform_w0 && p->deform_w1 && p->deform_w2 && p->torso_w0 && p->torso_w1 && p->torso_w2 && p->emb && p->offsets &&
                   p->enc_anchor, LZ_ERR_BAD_ARGUMENT, "torso_forward: incomplete lz_torso_params");
    LZ_REQUIRE(p->ind_dim == 0 || p->ind_code, LZ_ERR_BAD_ARGUMENT, "torso_forward: ind_code required when ind_dim > 0");
    LzTorsoArgs a;
    a.p = *p;
    for (int l = 0; l < 16; l++) {   // gridencoder.cu:125-126 on the host, same libm call as the CPU checker
        const float sc = exp2f((float)l * p->S) * (float)p->H - 1.0f;
        a.scale[l] = sc;
        a.res[l] = (uint32_t)ceilf(sc) + 1u;
    }
    const dim3 grid(lz_div_up(N, 256)), block(256);
    hipStream_t st = lz_st(stream);
    switch (p->ind_dim) {
        case 0: hipLaunchKernelGGL((lz_k_torso_forward<0>), grid, block, 0, st, a, bg_coords, N, alpha, color, deform); break;
        case 8: hipLaunchKernelGGL((lz_k_torso_forward<8>), grid, block, 0, st, a, bg_coords, N, alpha, color, deform); break;
        default: lz_set_error("torso_forward: ind_dim_torso must be 0 or 8 (the reference's default)"); return LZ_ERR_UNSUPPORTED;
    }
    LZ_CHECK_LAUNCH("torso_forward");
    return LZ_OK;
}
