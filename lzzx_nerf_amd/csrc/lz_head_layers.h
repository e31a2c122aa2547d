// lz_head_layers.h -- layer tables, packed-weight layout and the per-layer MFMA / VALU helpers shared by the forward
// (lz_head.hip) and backward (lz_head_bwd.hip) fused-head kernels.
#ifndef LZ_HEAD_LAYERS_H
#define LZ_HEAD_LAYERS_H
#include "lz_common.h"
#include "lzzx_detmath.h"

typedef float lz_f4 __attribute__((ext_vector_type(4)));

// ---- packed layout: layer table (fragment offsets) ----
// Layers on the matrix cores.  The skinny ones (eye_att 16 -> 1, sigma row of sigma_net.2, colour 64 -> 3, unc 32 -> 1, and
// the sum of squares behind ||att||) would each burn whole 16-row MFMA tiles for 1-3 useful rows (44 of 405 MFMAs per slice);
// they run on the VALU instead (lz_lane_dot), concurrently with the other waves' MFMAs.
enum { LZ_L_A1 = 0, LZ_L_A2, LZ_L_E1, LZ_L_S1, LZ_L_S2, LZ_L_S3, LZ_L_C1, LZ_L_U1, LZ_L_COUNT };
//                                    A1  A2  E1  S1  S2  S3  C1  U1
constexpr int LZ_KS[LZ_L_COUNT] = {   9, 16,  9, 18, 16, 16, 21,  9 };
constexpr int LZ_NT[LZ_L_COUNT] = {   4,  2,  1,  4,  4,  4,  4,  2 };
constexpr int lz_frag_base(int layer) {
    int b = 0;
    for (int i = 0; i < layer; i++) b += LZ_KS[i] * LZ_NT[i];
    return b;
}
// backward on the f16 matrix cores: fragments (kt, ft) per layer, kt over the ceil(KS / 4) tiles of 16 input slots; unc_net has no
// data gradient (its input is detached, network.py:241-249)
constexpr int lz_kt(int layer) { return (LZ_KS[layer] + 3) / 4; }
constexpr int lz_bfrag_base(int layer) {
    int b = 0;
    for (int i = 0; i < layer; i++) b += i == LZ_L_U1 ? 0 : lz_kt(i) * LZ_NT[i];
    return b;
}
constexpr int LZ_BFRAGS = lz_bfrag_base(LZ_L_COUNT);     // 99
constexpr int LZ_FRAGS_INFER = lz_frag_base(LZ_L_U1);   // 361
constexpr int LZ_FRAGS_ALL = lz_frag_base(LZ_L_COUNT);  // 379
// VALU-layer weights, plain rows indexed by input feature, after the fragments: colour.1 [3][64], sigma row [64], eye.1 [16], unc.1 [32]
constexpr int LZ_WV_C2 = 0, LZ_WV_SIG = 192, LZ_WV_E2 = 256, LZ_WV_U2 = 272, LZ_WV_FLOATS = 320;
static_assert(LZ_FRAGS_ALL * 64 + LZ_WV_FLOATS == LZ_HEAD_PACKED_FLOATS, "packed size mismatch with the header");


struct LzHeadArgs {
    const float* emb[3];
    const int* offsets;
    const float* packed;
    const float* enc_a;
    const float* ind_code;
    const float* eye;
    float bound;
    float scale[12];
    uint32_t res[12];
    int testing;
};

template <int LAYER, int T>
__device__ __forceinline__ void lz_layer(const float* __restrict__ wl, int lane, const float (&b)[T][LZ_KS[LAYER]],
                                         lz_f4 (&acc)[LZ_NT[LAYER]][T]) {
    constexpr int KS = LZ_KS[LAYER], NT = LZ_NT[LAYER];
    const float* frag = wl + lz_frag_base(LAYER) * 64 + lane;
#pragma unroll
    for (int ks = 0; ks < KS; ks++) {
        float a[NT];
#pragma unroll
        for (int ft = 0; ft < NT; ft++) a[ft] = frag[(ks * NT + ft) * 64];
#pragma unroll
        for (int ft = 0; ft < NT; ft++)
#pragma unroll
            for (int j = 0; j < T; j++) acc[ft][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[ft], b[j][ks], acc[ft][j], 0, 0, 0);
    }
}

__device__ __forceinline__ float lz_relu(float v) { return v > 0.0f ? v : 0.0f; }

// VALU layer: dot product of a weight row with an activation vector held in the chained layout (lane q of a sample holds
// x[4 t + r] = feature 16 t + 4 q + r, t < NTILE).  Each lane runs an fma chain over its features in (t, r) order, then the four
// lanes of the sample are combined as (p0 + p1) + (p2 + p3) (two xor shuffles; f32 addition commutes, so every lane ends with the
// same bits).  This order is what oracle/head.py restates (lzo_linear_lanes).
template <int NTILE>
__device__ __forceinline__ float lz_lane_dot(const float* __restrict__ wrow, int q, const float (&x)[4 * NTILE]) {
    float acc = 0.0f;
#pragma unroll
    for (int t = 0; t < NTILE; t++)
#pragma unroll
        for (int r = 0; r < 4; r++) acc = lz_fmaf(wrow[16 * t + 4 * q + r], x[4 * t + r], acc);
    acc += __shfl_xor(acc, 16, 64);
    acc += __shfl_xor(acc, 32, 64);
    return acc;
}

#endif
