// lz_head_bwd_common.h -- pieces shared by the two backward kernels of the fused head (lz_head_bwd.hip: activations recomputed;
// lz_head_rec.hip: activations recorded by the forward): the transposed-fragment layer, record stores, ReLU masks.
#ifndef LZ_HEAD_BWD_COMMON_H
#define LZ_HEAD_BWD_COMMON_H
#include "lz_head_layers.h"

#define LZ_BWD_WG 512

// partial-tile image of the weight-gradient products (lz_head_gradw.hip; also written by the fused backward of lz_head_rec.hip):
// 95 tiles of 16 x 16, first tile of each product
#define LZ_DW_TILES 95
#define LZ_DW_MAX_PARTS 768
#define LZ_DW_T_X3 0
#define LZ_DW_T_AUD1 21
#define LZ_DW_T_SIG1 29
#define LZ_DW_T_SIG0 45
#define LZ_DW_T_C1H 65
// sums `n_parts` partial images and scatters them into the five row-major matrices (lz_head_gradw.hip)
int lz_head_grad_w_reduce_launch(const float* parts, uint32_t n_parts, bool h16, uint32_t k_sig0, float* dW_x3, float* dW_aud1, float* dW_sig0,
                                 float* dW_sig1, float* dW_c1h, lz_stream_t stream);

// record / state stores: written once, read back once by a later kernel
#if defined(LZ_REC_NO_STORES)   /* experiment: the kernels without their record traffic (results are garbage) */
#define LZ_REC_STORE(v, p) ((void)(p), (void)(v))
#elif defined(LZ_REC_PLAIN_STORES)
#define LZ_REC_STORE(v, p) (*(p) = (v))
#else
#define LZ_REC_STORE(v, p) __builtin_nontemporal_store((v), (p))
#endif

struct LzHeadBwdArgs {
    LzHeadArgs fwd;
    const float *g_sigma, *g_rgb, *g_amb_aud, *g_amb_eye, *g_unc;     // upstream gradients [M], [M,3], [M], [M], [M]
    lz_head_bwd_out o;
    const void* wb16;   // transposed f16 fragments (lz_head_pack_weights_bwd_f16) for the backward on the f16 matrix cores, else null
    // the recomputing arrangement (lz_k_triplane_head_backward_rec<.., RC>): the forward's f16 weight image (lz_head_pack_weights_f16), unc_net's
    // five fragments (lz_head_pack_unc_f16) and the samples' view directions; the kernel's `st` argument is then the forward's encx16
    const void* fw16 = nullptr;
    const void* unc16 = nullptr;
    const float* dirs = nullptr;
};

// dX = W^T dY on the matrix cores from the FORWARD fragments of `LAYER` (see the header comment for the address map)
template <int LAYER>
__device__ __forceinline__ void lz_layer_bwd(const float* __restrict__ wl, int lane, const float (&dy)[4 * LZ_NT[LAYER]], float (&dx)[LZ_KS[LAYER]]) {
    constexpr int KS = LZ_KS[LAYER], NT = LZ_NT[LAYER], KT = (KS + 3) / 4;
    const int m = lane & 15, q = lane >> 4;
    const float* base = wl + lz_frag_base(LAYER) * 64 + 4 * q + 16 * (m >> 2);
    // One 16-byte LDS read feeds four MFMAs (128 cycles); the read of step i + 1 is issued BEFORE the MFMAs of step i and the
    // scheduling barrier keeps it there, so its latency (4-way bank conflict included) hides under them.
    auto frag = [&](int i) -> lz_f4 {   // raw read (rows past KS are clamped to row 0 and zeroed at use)
        const int kt = i / NT, ft = i - kt * NT;
        const int ks = 4 * kt + (m & 3);
        return *reinterpret_cast<const lz_f4*>(base + (size_t)(ks < KS ? ks : 0) * NT * 64 + ft * 64);
    };
    lz_f4 a_cur = frag(0);
    lz_f4 acc = lz_f4{0, 0, 0, 0};
#pragma unroll
    for (int i = 0; i < KT * NT; i++) {
        const int kt = i / NT, ft = i - kt * NT;
        lz_f4 a_nxt = a_cur;
        if (i + 1 < KT * NT) a_nxt = frag(i + 1);
        __builtin_amdgcn_sched_barrier(0);
        if (ft == 0) acc = lz_f4{0, 0, 0, 0};
        const bool ok = 4 * kt + (m & 3) < KS;
#pragma unroll
        for (int r = 0; r < 4; r++) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(ok ? a_cur[r] : 0.0f, dy[4 * ft + r], acc, 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        if (ft == NT - 1) {
#pragma unroll
            for (int r = 0; r < 4; r++)
                if (4 * kt + r < KS) dx[4 * kt + r] = acc[r];
        }
        a_cur = a_nxt;
    }
}

// The same product on v_mfma_f32_16x16x16_f16 from the transposed half fragments (lz_head_pack_weights_bwd_f16): a D tile of dY sits in
// a lane's four registers as the four k values 4 q + j of that tile, so one instruction per (input tile, output tile) replaces the four
// f32 ones; dY and the weights are rounded to half (what the reference's autocast backward multiplies), the sum stays f32.
typedef _Float16 lz_bh4 __attribute__((ext_vector_type(4)));
template <int LAYER>
__device__ __forceinline__ void lz_layer_bwd16(const uint2* __restrict__ wb, int lane, const float (&dy)[4 * LZ_NT[LAYER]], float (&dx)[LZ_KS[LAYER]]) {
    constexpr int KS = LZ_KS[LAYER], NT = LZ_NT[LAYER], KT = (KS + 3) / 4;
    typedef float lz_f2v __attribute__((ext_vector_type(2)));
    typedef _Float16 lz_h2v __attribute__((ext_vector_type(2)));
    lz_bh4 b[NT];
#pragma unroll
    for (int ft = 0; ft < NT; ft++) {
        const lz_f2v lo = {dy[4 * ft], dy[4 * ft + 1]}, hi = {dy[4 * ft + 2], dy[4 * ft + 3]};
        const lz_h2v l = __builtin_convertvector(lo, lz_h2v), h = __builtin_convertvector(hi, lz_h2v);
        b[ft] = lz_bh4{l[0], l[1], h[0], h[1]};
    }
    const uint2* frag = wb + lz_bfrag_base(LAYER) * 64 + lane;
#pragma unroll
    for (int kt = 0; kt < KT; kt++) {
        lz_f4 acc = lz_f4{0, 0, 0, 0};
#pragma unroll
        for (int ft = 0; ft < NT; ft++) {
            const uint2 w = frag[(kt * NT + ft) * 64];
            acc = __builtin_amdgcn_mfma_f32_16x16x16f16(__builtin_bit_cast(lz_bh4, w), b[ft], acc, 0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < 4; r++)
            if (4 * kt + r < KS) dx[4 * kt + r] = acc[r];
    }
}

// Record / state buffers are BLOCKED by 16-sample slice: [slice][tile of 16 dwords][sample 16][16 dwords], so that a store or load
// instruction of a wave (16 samples x 4 lanes x 16 bytes) covers ONE contiguous kilobyte -- eight whole cache lines -- instead of sixteen
// separate 64-byte pieces a row apart.  Column c of a sample (the per-sample layouts of include/lzzx_nerf_hip.h) sits at dword
// lz_tcol(c) of its slot in the slice block; buffers are allocated for whole slices.
__device__ __forceinline__ int lz_tcol(int c) { return ((c >> 4) << 8) + (c & 15); }
__device__ __forceinline__ float* lz_blk(float* base, uint32_t slice, int row_dwords, int s) {
    return base + (size_t)slice * (16u * (uint32_t)row_dwords) + s * 16;
}
__device__ __forceinline__ const float* lz_blk(const float* base, uint32_t slice, int row_dwords, int s) {
    return base + (size_t)slice * (16u * (uint32_t)row_dwords) + s * 16;
}

// chained-layout vector v[4 t + r] = feature 16 t + 4 q + r -> record columns col0 + feature of the sample whose block slot is `rb`:
// one dwordx4 per tile (a lane's four columns never straddle a tile: col0 is a multiple of 4)
template <int NTILE>
__device__ __forceinline__ void lz_dump_chained(float* __restrict__ rb, int q, int col0, const float (&v)[4 * NTILE]) {
    typedef float lz_v4 __attribute__((ext_vector_type(4)));
#pragma unroll
    for (int t = 0; t < NTILE; t++) {
        lz_v4 w = {v[4 * t], v[4 * t + 1], v[4 * t + 2], v[4 * t + 3]};
        LZ_REC_STORE(w, reinterpret_cast<lz_v4*>(rb + lz_tcol(col0 + 16 * t + 4 * q)));   // streamed: read back once
    }
}

// v where bit k of the ReLU mask is set, +0 elsewhere: the bit sign-extended to a word (v_bfe_i32) and one AND -- two instructions per value
// where the compare + select form the compiler makes of `(mk >> k) & 1 ? v : 0` costs three (76 values per slice of the backward kernels,
// which are bound by vector-instruction issue); the same bits in every case (a kept value is untouched, a dropped one is +0)
__device__ __forceinline__ float lz_mask_keep(uint32_t mk, int k, float v) {
    return __uint_as_float(__float_as_uint(v) & (uint32_t)__builtin_amdgcn_sbfe((int)mk, (uint32_t)k, 1u));
}
// (the recomputing backward sits at its register limit and spills nine more values with the AND form: it keeps the select)
#define LZ_MASK_KEEP(SELECT, mk, k, v) ((SELECT) ? ((((mk) >> (k)) & 1u) ? (v) : 0.0f) : lz_mask_keep((mk), (k), (v)))

template <int N>
__device__ __forceinline__ uint32_t lz_mask_pos(const float (&v)[N]) {
    uint32_t mk = 0;
#pragma unroll
    for (int k = 0; k < N; k++) mk |= (v[k] > 0.0f) ? (1u << k) : 0u;
    return mk;
}

#endif
