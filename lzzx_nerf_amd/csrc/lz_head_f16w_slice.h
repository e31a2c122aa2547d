// lz_head_f16w_slice.h -- the f16 fused triplane head (torch-autocast rounding) for ONE 32-sample slice of a wave on
// v_mfma_f32_32x32x16_f16: the inference arrangement since round 5 (stand-alone head lz_head_f16.hip, fused frame lz_frame.hip).
// The 16-sample slice on v_mfma_f32_16x16x32_f16 (lz_head_f16_slice.h) stays with the recording training forward.
//
// Why the wide shape.  The f16 kernels are bound by vector-instruction ISSUE, and an MFMA holds its SIMD's issue port for 8 cycles
// whatever its shape (MI355X_MICROARCH.md, cycle constants): the 16x16x32 slice issued 59 of them per 16 samples (118 per row pair, 944
// issue cycles), the 32x32x16 slice issues 60 per 32 samples (480) for the same matrix-pipe time (60 x 32 vs 118 x 16 cycles).  Rounding
// sequence, what is half and what is f32: lz_head_f16.hip; only the grouping of the f32 accumulation inside a Linear differs (k in steps
// of 16 instead of 32), which the reference leaves to its GEMM library anyway.
//
// Layout.  Lane l = (s = l & 31, h = l >> 5): sample s of the slice, lane half h.
//   * B operand (activations) of k-step ks: element j of lane (s, h) = k slot 16 ks + 8 h + j of sample s.
//   * A operand (weights) of (k-step, 32-row tile): element j of lane (r, h) = W[row r of the tile][that k slot]: one 16-byte LDS read.
//   * D tile: register i of lane (s, h) = output row (i & 3) + 8 (i >> 2) + 4 h of sample s.  Registers 8 u .. 8 u + 7 of tile t, converted
//     pairwise to half, ARE the B operand of k-step 2 t + u of the next layer, k slot j <-> feature 32 t + 16 u + 8 (j >> 2) + 4 h + (j & 3)
//     (w_chain): activations never leave registers or cross lanes, the packer permutes the weights once.
//   * enc_x: lane half h gathers the 18 features of levels l = h (mod 2), two lz_head_gather calls of 9 (36 loads in flight each): slot
//     n = 8 ks + j < 18 of the three enc_x k-steps <-> call n / 9, feature i = n % 9 of it (w_encx); slot 18 of half 0 carries eye * eye_att.
//   * the four transcendentals of a sample sit two per lane: chain A = colour channel h (colour_net.1 rows 0 / 4 = register 0), chain B =
//     colour channel 2 on h = 0 (row 1 = register 1) and sigma on h = 1 (sigma_net.2's sigma row alone in its third tile, at row 4).
#ifndef LZ_HEAD_F16W_SLICE_H
#define LZ_HEAD_F16W_SLICE_H
#include "lz_head_f16_slice.h"   // LzHead16Args / LzHead16Ctx, h_cvt2, h_round*, h_sigmoid, the level table

typedef float lz_f16v __attribute__((ext_vector_type(16)));

enum { W_A1 = 0, W_A2, W_E1, W_E2, W_S1, W_S2, W_S3, W_C1, W_C2, W_COUNT };
//                               A1 A2 E1 E2 S1 S2 S3 C1 C2
constexpr int W_KS[W_COUNT] = {  3, 4, 3, 1, 5, 4, 4, 6, 4 };     // k-steps of 16
constexpr int W_NT[W_COUNT] = {  2, 1, 1, 1, 2, 2, 3, 2, 1 };     // output tiles of 32 rows
constexpr int w_frag_base(int layer) {
    int b = 0;
    for (int i = 0; i < layer; i++) b += W_KS[i] * W_NT[i];
    return b;
}
constexpr int W_FRAGS = w_frag_base(W_COUNT);  // 60
static_assert(W_FRAGS * 64 * 16 == LZ_HEAD_PACKED_F16W_BYTES, "packed size mismatch with the header");
constexpr int LZ_HEAD16W_LDS_H8 = W_FRAGS * 64 + LZ_LVTAB_WORDS / 4;   // lz_h8 elements: fragments, then the level table (enc_a inside)

// input feature held by k slot (k-step ks, lane half h, j) of a chained B operand
__host__ __device__ __forceinline__ int w_chain(int ks, int h, int j, int K) {
    const int f = 16 * ks + 8 * (j >> 2) + 4 * h + (j & 3);
    return f < K ? f : -1;
}
// ... and of the gathered enc_x operand: slot n = 8 ks + j <-> call n / 9 (levels 6 c + h, + 2, + 4), feature i = n % 9 of that call
// (plane i / 3, level record i % 3: lz_head_gather<.., LSTRIDE = 2>)
__host__ __device__ __forceinline__ int w_encx(int ks, int h, int j) {
    const int n = 8 * ks + j;
    if (n >= 18) return -1;
    const int c = n / 9, i = n % 9;
    return 12 * (i / 3) + 6 * c + 2 * (i % 3) + h;
}
constexpr int W_EYE_SLOT = 18;   // k slot (ks 2, j 2) of lane half 0 in sigma_net.0's enc_x operand: eye * eye_att

template <int LAYER>
__device__ __forceinline__ void w_layer(const lz_h8* __restrict__ wl, int lane, const lz_h8 (&b)[W_KS[LAYER]], lz_f16v (&acc)[W_NT[LAYER]]) {
    constexpr int KS = W_KS[LAYER], NT = W_NT[LAYER];
    const lz_h8* frag = wl + w_frag_base(LAYER) * 64 + lane;
#pragma unroll
    for (int ks = 0; ks < KS; ks++)
#pragma unroll
        for (int ft = 0; ft < NT; ft++) acc[ft] = __builtin_amdgcn_mfma_f32_32x32x16_f16(frag[(ks * NT + ft) * 64], b[ks], acc[ft], 0, 0, 0);
}

// registers 8 u .. 8 u + 7 of a D tile -> the B operand of k-step 2 t + u of the next layer (half output of an autocast Linear [+ relu])
__device__ __forceinline__ lz_h8 w_pack(const lz_f16v& d, int u, bool relu) {
    const lz_u4v w = {h_cvt2(d[8 * u], d[8 * u + 1], relu), h_cvt2(d[8 * u + 2], d[8 * u + 3], relu), h_cvt2(d[8 * u + 4], d[8 * u + 5], relu),
                      h_cvt2(d[8 * u + 6], d[8 * u + 7], relu)};
    return __builtin_bit_cast(lz_h8, w);
}
__device__ __forceinline__ lz_f16v w_zero() {
    lz_f16v z;
#pragma unroll
    for (int i = 0; i < 16; i++) z[i] = 0.0f;
    return z;
}
// enc_a * att for the att operand of k-step u (features 16 u + 8 (j >> 2) + 4 h + (j & 3)): half * half -> half, one rounding (h_encw)
__device__ __forceinline__ lz_h8 w_encw(const int* __restrict__ tab, int h, int u, const lz_h8& att) {
    typedef uint32_t lz_u2v __attribute__((ext_vector_type(2)));
    const lz_u2v lo = *reinterpret_cast<const lz_u2v*>(tab + LZ_LVTAB_ENCA16 + 8 * u + 2 * h),
                 hi = *reinterpret_cast<const lz_u2v*>(tab + LZ_LVTAB_ENCA16 + 8 * u + 4 + 2 * h);
    const lz_u4v e = {lo[0], lo[1], hi[0], hi[1]};
    return __builtin_bit_cast(lz_h8, e) * att;
}

struct LzHead16wOut {
    float a, b;                  // the lane's two transcendental chains: h = 0: rgb[0], rgb[2]; h = 1: rgb[1], sigma
    float ambaud, eyeatt, unc;   // eyeatt is valid on lanes h == 0
};

// SH(4) of the sample's direction as eight halves (components 8 h .. 8 h + 7) in four words: from an f32 evaluator (LzShFromDir) ...
template <typename ShFn>
__device__ __forceinline__ void w_sh_pk(const ShFn& f, int h, uint32_t (&w)[4]) {
#pragma unroll
    for (int k = 0; k < 4; k++) w[k] = h_cvt2(f.comp_qj(2 * h + (k >> 1), 2 * (k & 1)), f.comp_qj(2 * h + (k >> 1), 2 * (k & 1) + 1), false);
}

// stage weights + tables into LDS (all threads; caller synchronises afterwards) and fill the context (the 16-sample head's, same fields)
__device__ __forceinline__ void lz_head16w_stage(const LzHead16Args& P, lz_h8* wl, uint32_t n_threads, LzHead16Ctx& hc) {
    float* tabf = reinterpret_cast<float*>(wl + W_FRAGS * 64);
    int* tab = reinterpret_cast<int*>(tabf);
    for (uint32_t i = threadIdx.x; i < (uint32_t)W_FRAGS * 64; i += n_threads) wl[i] = P.packed[i];
    lz_level_table_fill(tab, P.offsets, P.scale, P.res);   // + the slice queue head of the stand-alone kernel
    if (threadIdx.x < 32) tabf[LZ_LVTAB_ENCA + threadIdx.x] = (float)(_Float16)P.enc_a[threadIdx.x];   // enc_a is half under autocast
    if (threadIdx.x < 16) tab[LZ_LVTAB_ENCA16 + threadIdx.x] = (int)h_cvt2(P.enc_a[2 * threadIdx.x], P.enc_a[2 * threadIdx.x + 1], false);
    if (threadIdx.x < 2) tab[LZ_LVTAB_IND16 + threadIdx.x] = P.ind_code ? (int)h_cvt2(P.ind_code[2 * threadIdx.x], P.ind_code[2 * threadIdx.x + 1], false) : 0;
    hc.wl = wl;
    hc.tab = tab;
    hc.lenca = tabf + LZ_LVTAB_ENCA;
    hc.emb[0] = P.emb[0]; hc.emb[1] = P.emb[1]; hc.emb[2] = P.emb[2];
    hc.ind_code = P.ind_code;
    hc.bound = P.bound;
    hc.two_bound = 2.0f * P.bound;
    hc.has_eye = P.eye != nullptr;
    hc.eye_v = hc.has_eye ? P.eye[0] : 0.0f;
    hc.unc_const = lz_softplusf(0.0f);   // test mode (network.py:243-249, 278)
}

#ifndef LZ_F16W_PRIO
#define LZ_F16W_PRIO 1   // the frame kernel runs march + gather addresses at a raised wave priority and drops it once a gather call has issued its loads (same-box A/B, three alternations: 1.789 -> 1.783, 1.800 -> 1.791 ms, cfg5 0.770 -> 0.762; -DLZ_F16W_PRIO=0 switches it off)
#endif
template <bool IN_RANGE = false, bool YIELD = false, typename ShFn>   // SH(4) source: LzShFromDir (lz_head_slice.h) or the frame kernel's per-ray LDS copy
__device__ __forceinline__ void lz_head16w_slice(const LzHead16Ctx& hc, int lane, float px, float py, float pz, ShFn shfn, LzHead16wOut& out) {
    const int h = lane >> 5;
    // ---------------- gather (f32, lz_head_gather.h, the f32 kernels' arithmetic): 18 features of this lane's sample, 2 x 36 loads ----------------
    lz_h8 bx[3];
    {
        float e0[9], e1[9], c01[3];
        lz_head_map01(px, py, pz, hc.bound, hc.two_bound, c01);      // once for both calls
        lz_head_gather<IN_RANGE, LZ_GATHER_PACK16, YIELD, false, 2, true>(hc.emb, hc.tab, c01[0], c01[1], c01[2], h, hc.bound, hc.two_bound, e0);
        // h_round2: every f32 feature exists first, then its half (no v_fma_mixlo_f16 with the interpolation's last fma)
        const lz_u4v w0 = {h_round2(e0[0], e0[1]), h_round2(e0[2], e0[3]), h_round2(e0[4], e0[5]), h_round2(e0[6], e0[7])};
        bx[0] = __builtin_bit_cast(lz_h8, w0);
        if constexpr (YIELD) __builtin_amdgcn_s_setprio(2);
        lz_head_gather<IN_RANGE, LZ_GATHER_PACK16, YIELD, false, 2, true>(hc.emb, hc.tab, c01[0], c01[1], c01[2], 6 + h, hc.bound, hc.two_bound, e1);
        const lz_u4v w1 = {h_round2(e0[8], e1[0]), h_round2(e1[1], e1[2]), h_round2(e1[3], e1[4]), h_round2(e1[5], e1[6])};
        const lz_u4v w2 = {h_round2(e1[7], e1[8]), 0u, 0u, 0u};
        bx[1] = __builtin_bit_cast(lz_h8, w1);
        bx[2] = __builtin_bit_cast(lz_h8, w2);
    }
    // ---------------- audio channel attention: 36 -> 64 -> 32 ----------------
    lz_h8 att16[2];   // [u][j] = feature 16 u + 8 (j >> 2) + 4 h + (j & 3)
    {
        lz_f16v a1[2] = {w_zero(), w_zero()};
        w_layer<W_A1>(hc.wl, lane, bx, a1);
        const lz_h8 b2[4] = {w_pack(a1[0], 0, true), w_pack(a1[0], 1, true), w_pack(a1[1], 0, true), w_pack(a1[1], 1, true)};
        lz_f16v a2[1] = {w_zero()};
        w_layer<W_A2>(hc.wl, lane, b2, a2);
        att16[0] = w_pack(a2[0], 0, false);
        att16[1] = w_pack(a2[0], 1, false);
    }
    {   // ambient_aud = || att ||_2 in f32 (norm is an autocast-to-f32 op): lane partial over its 16 features, then over the halves
        float ss = 0.0f;
#pragma unroll
        for (int u = 0; u < 2; u++)
#pragma unroll
            for (int k = 0; k < 8; k++) ss = lz_fmaf((float)att16[u][k], (float)att16[u][k], ss);
        ss += __shfl_xor(ss, 32, 64);
        out.ambaud = h_sqrt32(ss);
    }
    // ---------------- eye attention: 36 -> 16 -> 1, sigmoid (half) ----------------
    float eyeatt = 0.0f;
    if (hc.has_eye) {
        lz_f16v e1[1] = {w_zero()};
        w_layer<W_E1>(hc.wl, lane, bx, e1);
        const lz_h8 be[1] = {w_pack(e1[0], 0, true)};       // rows 0 .. 15 are registers 0 .. 7
        lz_f16v e2[1] = {w_zero()};
        w_layer<W_E2>(hc.wl, lane, be, e2);
        eyeatt = (float)(_Float16)h_sigmoid((float)(_Float16)e2[0][0]);   // valid on lanes h == 0 (row 0)
    }
    // ---------------- sigma net: [enc_x 36 | enc_a * att 32 | eye * eye_att 1] -> 64 -> 64 -> 65 ----------------
    lz_h8 geo16[4];
    float spre;
    {
        lz_h8 b1[5];
        b1[0] = bx[0]; b1[1] = bx[1]; b1[2] = bx[2];
        b1[2][2] = (hc.has_eye && h == 0) ? h_round(hc.eye_v * eyeatt) : (_Float16)0.0f;
        b1[3] = w_encw(hc.tab, h, 0, att16[0]);
        b1[4] = w_encw(hc.tab, h, 1, att16[1]);
        lz_f16v s1[2] = {w_zero(), w_zero()};
        w_layer<W_S1>(hc.wl, lane, b1, s1);
        const lz_h8 b2[4] = {w_pack(s1[0], 0, true), w_pack(s1[0], 1, true), w_pack(s1[1], 0, true), w_pack(s1[1], 1, true)};
        lz_f16v s2[2] = {w_zero(), w_zero()};
        w_layer<W_S2>(hc.wl, lane, b2, s2);
        const lz_h8 b3[4] = {w_pack(s2[0], 0, true), w_pack(s2[0], 1, true), w_pack(s2[1], 0, true), w_pack(s2[1], 1, true)};
        lz_f16v s3[3] = {w_zero(), w_zero(), w_zero()};
        w_layer<W_S3>(hc.wl, lane, b3, s3);
        geo16[0] = w_pack(s3[0], 0, false);   // geo_feat, no activation (network.py:304)
        geo16[1] = w_pack(s3[0], 1, false);
        geo16[2] = w_pack(s3[1], 0, false);
        geo16[3] = w_pack(s3[1], 1, false);
        spre = (float)(_Float16)s3[2][0];      // the sigma row (half): row 4 of the third tile = register 0 of lanes h == 1
    }
    // ---------------- colour net: [SH 16 | geo 64 | ind 4] -> 64 -> 3 ----------------
    {
        lz_h8 b1[6];
        {
            ShFn f = shfn;
            f.prepare();
            uint32_t shw[4];
            w_sh_pk(f, h, shw);
            const lz_u4v w = {shw[0], shw[1], shw[2], shw[3]};
            b1[0] = __builtin_bit_cast(lz_h8, w);
            const lz_u4v wi = {h == 0 ? (uint32_t)hc.tab[LZ_LVTAB_IND16] : 0u, h == 0 ? (uint32_t)hc.tab[LZ_LVTAB_IND16 + 1] : 0u, 0u, 0u};
            b1[5] = __builtin_bit_cast(lz_h8, wi);
        }
        b1[1] = geo16[0]; b1[2] = geo16[1]; b1[3] = geo16[2]; b1[4] = geo16[3];
        lz_f16v c1[2] = {w_zero(), w_zero()};
        w_layer<W_C1>(hc.wl, lane, b1, c1);
        const lz_h8 b2[4] = {w_pack(c1[0], 0, true), w_pack(c1[0], 1, true), w_pack(c1[1], 0, true), w_pack(c1[1], 1, true)};
        lz_f16v c2[1] = {w_zero()};
        w_layer<W_C2>(hc.wl, lane, b2, c2);
        // The four transcendentals of a sample, two per lane and ONE instruction sequence per chain: sigma = exp(x) (an autocast-to-f32 op:
        // half in, f32 out) and sigmoid(x) = 1 / (1 + exp(-x)) both start with exp2 of x times +-log2(e) -- the same multiply and the same
        // exp2 as h_exp32 / h_sigmoid, so the same bits -- and the colour lanes go on through network.py:275 in half (* 1.002, - 0.001, each
        // rounded).  Chain A: colour channel h.  Chain B: channel 2 on h = 0, sigma on h = 1.
        {
            const float pre = (float)(_Float16)c2[0][0];
            const float e = __builtin_amdgcn_exp2f(pre * -1.44269504088896340736f);
            const _Float16 sg = (_Float16)__builtin_amdgcn_rcpf(1.0f + e);
            const _Float16 t1 = h_round((float)sg * 1.002f);
            out.a = (float)h_round((float)t1 - 0.001f);
        }
        {
            const float pre = h == 1 ? spre : (float)(_Float16)c2[0][1];
            const float e = __builtin_amdgcn_exp2f(pre * (h == 1 ? 1.44269504088896340736f : -1.44269504088896340736f));
            const _Float16 sg = (_Float16)__builtin_amdgcn_rcpf(1.0f + e);
            const _Float16 t1 = h_round((float)sg * 1.002f);
            const float col = (float)h_round((float)t1 - 0.001f);
            out.b = h == 1 ? e : col;
        }
    }
    out.eyeatt = eyeatt;
    out.unc = hc.unc_const;
}
#endif
