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
#include <type_traits>

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

// get_grid_index (gridencoder.cu:54-72), D = 2, generic form (tiled grids wrap with a true modulo).  The modulo itself is ~25
// instructions per corner; it is the identity on a dense level (index < (res + 1)^2 <= hs) and a mask when hs is a power of two (every
// wrapped level of the reference's torso encoder: hs = 2^16), so the division only runs for table sizes that are neither
__device__ __forceinline__ uint32_t lz_torso_grid_index(uint32_t gridtype, uint32_t hs, uint32_t resolution, uint32_t p0, uint32_t p1) {
    uint32_t stride = 1, index = 0;
    if (stride <= hs) { index += p0 * stride; stride *= resolution + 1; }
    if (stride <= hs) { index += p1 * stride; stride *= resolution + 1; }
    if (gridtype == 0 && stride > hs) index = p0 ^ (p1 * 2654435761u);
    if ((hs & (hs - 1u)) == 0u) index &= hs - 1u;
    else if (index >= hs) index %= hs;
    return index * 2u;
}

// ---- the kernel: sixteen pixels per wave pass on v_mfma_f32_16x16x4_f32 ---------------------------------------------------------------
// Lane (s = lane & 15, q = lane >> 4) works on pixel s of the wave's slice.  Orientation D[feature, pixel] = W . X as in the fused head
// (lz_head.hip): A = weights (16 features x 4 k), B = inputs (4 k x 16 pixels; lane (s, q) supplies k = 4 ks + q), and in a D tile lane
// (s, q) register r holds feature 16 t + 4 q + r.  The checker's chains run over the inputs in NATURAL order from the frame-constant
// partial sum, and a k-step of this MFMA is bit for bit that fma chain over its four k -- so the layers keep natural order: the first
// layers' per-pixel inputs are produced by the lane that supplies them (frequency features 4 i + q, grid features 4 i + q = level
// 2 i + (q >> 1), channel q & 1), the accumulators start from the constant partial sums, and between layers a D tile becomes four B
// operands by a 4 x 4 transpose across the pixel's four lanes: two v_permlane32_swap + two v_permlane16_swap (gfx950) per tile.
// Round 1-2's form -- one lane per pixel, 5.4 kMAC of scalar fma chains with a broadcast LDS read per weight, 256 registers, one wave
// per SIMD -- took 0.19 ms per 512^2 frame; k-steps that run past a layer's width multiply zeros (fma(0, 0, acc) = acc).
#define LZT_WG 256
typedef float lzt_f4 __attribute__((ext_vector_type(4)));
enum { LZT_D0 = 0, LZT_D1, LZT_D2, LZT_T0, LZT_T1, LZT_T2, LZT_LAYERS };
//                                   D0  D1  D2  T0  T1  T2
constexpr int LZT_KS[LZT_LAYERS] = {  9,  8,  8, 17,  8,  8 };   // k-steps of 4: 34 -> 36, 32, 32, 66 -> 68, 32, 32
constexpr int LZT_NT[LZT_LAYERS] = {  2,  2,  1,  2,  2,  1 };   // feature tiles of 16: 32, 32, 2 -> 16, 32, 32, 4 -> 16
constexpr int lzt_base(int layer) {
    int b = 0;
    for (int i = 0; i < layer; i++) b += LZT_KS[i] * LZT_NT[i];
    return b;
}
constexpr int LZT_FRAGS = lzt_base(LZT_LAYERS);   // 100 fragments x 64 lanes x 4 B = 25.6 KB of LDS

template <int LAYER>
__device__ __forceinline__ void lzt_layer(const float* __restrict__ wl, int lane, const float (&b)[LZT_KS[LAYER]], lzt_f4 (&acc)[LZT_NT[LAYER]]) {
    constexpr int KS = LZT_KS[LAYER], NT = LZT_NT[LAYER];
    const float* frag = wl + lzt_base(LAYER) * 64 + lane;
#pragma unroll
    for (int ks = 0; ks < KS; ks++)
#pragma unroll
        for (int ft = 0; ft < NT; ft++) acc[ft] = __builtin_amdgcn_mfma_f32_16x16x4f32(frag[(ks * NT + ft) * 64], b[ks], acc[ft], 0, 0, 0);
}
// D tile (register r of lane q = feature 16 t + 4 q + r) -> register j of lane q = feature 16 t + 4 j + q: the B operands of k-steps 4 t + j
__device__ __forceinline__ void lzt_transpose(lzt_f4& v) {
    uint32_t r0 = __float_as_uint(v[0]), r1 = __float_as_uint(v[1]), r2 = __float_as_uint(v[2]), r3 = __float_as_uint(v[3]);
    auto a = __builtin_amdgcn_permlane32_swap(r0, r2, false, false);    // register bit 1 <-> lane bit 5
    auto b = __builtin_amdgcn_permlane32_swap(r1, r3, false, false);
    auto c = __builtin_amdgcn_permlane16_swap(a[0], b[0], false, false);   // register bit 0 <-> lane bit 4
    auto d = __builtin_amdgcn_permlane16_swap(a[1], b[1], false, false);
    v[0] = __uint_as_float(c[0]); v[1] = __uint_as_float(c[1]); v[2] = __uint_as_float(d[0]); v[3] = __uint_as_float(d[1]);
}

template <int IND>
__global__ void __launch_bounds__(LZT_WG)
lz_k_torso_forward(LzTorsoArgs A, const float* __restrict__ bg_coords, uint32_t N, float* __restrict__ alpha_out,
                   float* __restrict__ color_out, float* __restrict__ deform_out) {
    constexpr int KC = LZ_TORSO_ANCHOR + IND;              // frame-constant inputs
    constexpr int K0 = LZ_TORSO_FREQ + KC;                 // deform net input width
    constexpr int K1 = LZ_TORSO_GRIDF + K0;                // torso net input width
    constexpr int H = LZ_TORSO_HID;
    constexpr int KP = LZ_TORSO_GRIDF + LZ_TORSO_FREQ;     // per-pixel inputs of the torso net: [grid 32 | enc_x 34]
    __shared__ float wl[LZT_FRAGS * 64];
    __shared__ __align__(16) float cd[H], ct[H];
    const lz_torso_params& P = A.p;
    // A fragments: lane l of fragment (layer, ks, ft) = W[16 ft + (l & 15)][4 ks + (l >> 4)] over the layer's PER-PIXEL columns, zero outside
    auto pack = [&](auto layer_c, const float* __restrict__ w, int ld, int n_rows, int n_cols) {
        constexpr int LAYER = decltype(layer_c)::value;
        constexpr int NT = LZT_NT[LAYER], CNT = LZT_KS[LAYER] * NT * 64;
        float* dst = wl + lzt_base(LAYER) * 64;
        for (int i = threadIdx.x; i < CNT; i += LZT_WG) {
            const int fr = i >> 6, l = i & 63;
            const int ks = NT == 2 ? fr >> 1 : fr, ft = NT == 2 ? fr & 1 : 0;
            const int row = 16 * ft + (l & 15), col = 4 * ks + (l >> 4);
            dst[i] = (row < n_rows && col < n_cols) ? w[(size_t)row * ld + col] : 0.0f;
        }
    };
    pack(std::integral_constant<int, LZT_D0>{}, P.deform_w0, K0, H, LZ_TORSO_FREQ);
    pack(std::integral_constant<int, LZT_D1>{}, P.deform_w1, H, H, H);
    pack(std::integral_constant<int, LZT_D2>{}, P.deform_w2, H, 2, H);
    pack(std::integral_constant<int, LZT_T0>{}, P.torso_w0, K1, H, KP);
    pack(std::integral_constant<int, LZT_T1>{}, P.torso_w1, H, H, H);
    pack(std::integral_constant<int, LZT_T2>{}, P.torso_w2, H, 4, H);
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
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int s = lane & 15, q = lane >> 4;
    const uint32_t n_slices = (N + 15) / 16, stride = gridDim.x * (LZT_WG / 64);
    for (uint32_t slice = blockIdx.x * (LZT_WG / 64) + wave; slice < n_slices; slice += stride) {
        const uint32_t n_raw = slice * 16 + s;
        const bool valid = n_raw < N;
        const uint32_t n = valid ? n_raw : N - 1;     // clamped lanes compute a real pixel again and store nothing
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
        if (!__ballot(masked && valid)) {     // wave-uniform: nothing to evaluate in this slice (renderer.py:609-610: zeros)
            if (q == 0 && valid) {
                alpha_out[n] = 0.0f;
                color_out[(size_t)n * 3] = 0.0f; color_out[(size_t)n * 3 + 1] = 0.0f; color_out[(size_t)n * 3 + 2] = 0.0f;
                if (deform_out) { deform_out[(size_t)n * 2] = 0.0f; deform_out[(size_t)n * 2 + 1] = 0.0f; }
            }
            continue;
        }
        // ---- forward_torso (network.py:170-205) ----
        const float x[2] = {bx * P.torso_shrink, by * P.torso_shrink};
        // frequency features 4 i + q of [x, sin(2^f x), cos(2^f x)]_f (freqencoder.cu:30-66; cos as sin(. + pi/2)); 34 and 35 are padding
        float ex[9];
#pragma unroll
        for (int i = 0; i < 9; i++) {
            const int c = 4 * i + q;
            float v = 0.0f;
            if (c < 2) v = x[c & 1];
            else if (c < LZ_TORSO_FREQ) {
                const int col = c / 2 - 1, d = c % 2, freq = col / 2;
                v = lz_sinf(lz_scalbnf(x[d], freq) + (float)(col % 2) * (LZ_PI_F / 2));
            }
            ex[i] = v;
        }
        auto relu4 = [](lzt_f4& v) {
#pragma unroll
            for (int r = 0; r < 4; r++) v[r] = v[r] > 0.0f ? v[r] : 0.0f;
        };
        auto init2 = [&](const float* c0, lzt_f4 (&acc)[2]) {
#pragma unroll
            for (int t = 0; t < 2; t++) acc[t] = *reinterpret_cast<const lzt_f4*>(c0 + 16 * t + 4 * q);
        };
        // hidden pair of a net: D tiles -> ReLU -> transpose -> eight B operands
        auto to_b = [&](lzt_f4 (&acc)[2], float (&b)[8]) {
#pragma unroll
            for (int t = 0; t < 2; t++) {
                relu4(acc[t]);
                lzt_transpose(acc[t]);
#pragma unroll
                for (int j = 0; j < 4; j++) b[4 * t + j] = acc[t][j];
            }
        };
        float dx[2];
        {
            lzt_f4 a0[2];
            init2(cd, a0);
            lzt_layer<LZT_D0>(wl, lane, ex, a0);
            float b1[8];
            to_b(a0, b1);
            lzt_f4 a1[2] = {lzt_f4{0, 0, 0, 0}, lzt_f4{0, 0, 0, 0}};
            lzt_layer<LZT_D1>(wl, lane, b1, a1);
            float b2[8];
            to_b(a1, b2);
            lzt_f4 a2[1] = {lzt_f4{0, 0, 0, 0}};
            lzt_layer<LZT_D2>(wl, lane, b2, a2);
            dx[0] = __shfl(a2[0][0], s, 64);      // rows 0, 1 of the tile sit in lane (s, q = 0)
            dx[1] = __shfl(a2[0][1], s, 64);
        }
        // x = (x + dx).clamp(-1, 1); torso_encoder(x, bound=1) (network.py:193-195, grid.py:143): this lane's grid features 4 i + q =
        // channel q & 1 of level 2 i + (q >> 1)
        float gx[8];
        {
            float u[2];
#pragma unroll
            for (int d = 0; d < 2; d++) u[d] = (lz_fminf(lz_fmaxf(x[d] + dx[d], -1.0f), 1.0f) + 1.0f) / 2.0f;
            const bool oob = u[0] < 0 || u[0] > 1 || u[1] < 0 || u[1] > 1;   // cannot happen after the clamp; kept for NaN-free parity
            const int ch = q & 1;
#pragma unroll
            for (int i = 0; i < 8; i++) {
                const int l = 2 * i + (q >> 1);
                const uint32_t off0 = (uint32_t)P.offsets[l], hs = (uint32_t)P.offsets[l + 1] - off0;
                const float sc = A.scale[l];
                const uint32_t resolution = A.res[l];
                const float* g = P.emb + (size_t)off0 * 2 + ch;
                const float p0 = lz_fmaf(u[0], sc, 0.5f), p1 = lz_fmaf(u[1], sc, 0.5f);
                const uint32_t g0 = (uint32_t)floorf(p0), g1 = (uint32_t)floorf(p1);
                const float f0 = p0 - (float)g0, f1 = p1 - (float)g1;
                float r0 = 0.0f;
#pragma unroll
                for (int c = 0; c < 4; c++) {
                    const float w = ((c & 1) ? f0 : 1 - f0) * ((c >> 1) ? f1 : 1 - f1);
                    const uint32_t index = lz_torso_grid_index(P.gridtype, hs, resolution, g0 + (c & 1), g1 + (c >> 1));
                    r0 = lz_fmaf(w, g[index], r0);
                }
                gx[i] = oob ? 0.0f : r0;
            }
        }
        // torso net: [grid 32 | enc_x 34 | anchor | ind] -> 32 -> 32 -> 4
        float out4[4];
        {
            float b0[17];
#pragma unroll
            for (int i = 0; i < 8; i++) b0[i] = gx[i];
#pragma unroll
            for (int i = 0; i < 9; i++) b0[8 + i] = ex[i];
            lzt_f4 a0[2];
            init2(ct, a0);
            lzt_layer<LZT_T0>(wl, lane, b0, a0);
            float b1[8];
            to_b(a0, b1);
            lzt_f4 a1[2] = {lzt_f4{0, 0, 0, 0}, lzt_f4{0, 0, 0, 0}};
            lzt_layer<LZT_T1>(wl, lane, b1, a1);
            float b2[8];
            to_b(a1, b2);
            lzt_f4 a2[1] = {lzt_f4{0, 0, 0, 0}};
            lzt_layer<LZT_T2>(wl, lane, b2, a2);
#pragma unroll
            for (int o = 0; o < 4; o++) out4[o] = lz_sigmoidf(a2[0][o]) * 1.002f - 0.001f;   // network.py:202-203; lanes q == 0
        }
        if (q == 0 && valid) {
            alpha_out[n] = masked ? out4[0] : 0.0f;
            color_out[(size_t)n * 3] = masked ? out4[1] : 0.0f;
            color_out[(size_t)n * 3 + 1] = masked ? out4[2] : 0.0f;
            color_out[(size_t)n * 3 + 2] = masked ? out4[3] : 0.0f;
            if (deform_out) { deform_out[(size_t)n * 2] = masked ? dx[0] : 0.0f; deform_out[(size_t)n * 2 + 1] = masked ? dx[1] : 0.0f; }
        }
    }
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
    // four waves per workgroup, a wave walks 16-pixel slices with the grid's stride; the weights are packed into LDS once per workgroup
    uint32_t nwg = lz_div_up(N, 16 * (LZT_WG / 64));
    const uint32_t cap = (uint32_t)lz_cu_count() * 3u;     // three workgroups per CU are resident (126 registers: three waves per SIMD)
    if (nwg > cap) nwg = cap;
    const dim3 grid(nwg), block(LZT_WG);
    hipStream_t st = lz_st(stream);
    switch (p->ind_dim) {
        case 0: hipLaunchKernelGGL((lz_k_torso_forward<0>), grid, block, 0, st, a, bg_coords, N, alpha, color, deform); break;
        case 8: hipLaunchKernelGGL((lz_k_torso_forward<8>), grid, block, 0, st, a, bg_coords, N, alpha, color, deform); break;
        default: lz_set_error("torso_forward: ind_dim_torso must be 0 or 8 (the reference's default)"); return LZ_ERR_UNSUPPORTED;
    }
    LZ_CHECK_LAUNCH("torso_forward");
    return LZ_OK;
}
