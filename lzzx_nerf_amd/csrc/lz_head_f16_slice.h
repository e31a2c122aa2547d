// lz_head_f16_slice.h -- the f16 fused triplane head (v_mfma_f32_16x16x32_f16, torch-autocast rounding) for ONE 16-sample slice of a
// wave, shared by the stand-alone kernel (lz_head_f16.hip) and the fused frame kernel (lz_frame.hip).  See lz_head_f16.hip for
// the rounding sequence it reproduces and the operand layout.
#ifndef LZ_HEAD_F16_SLICE_H
#define LZ_HEAD_F16_SLICE_H
#include <hip/hip_fp16.h>

#include "lz_common.h"
#include "lzzx_detmath.h"
#include "lzzx_sh_eval.h"
#include "lz_head_gather.h"
#include "lz_head_slice.h"   // LzShFromDir

#ifndef LZ_HEAD_LAYERS_H
typedef float lz_f4 __attribute__((ext_vector_type(4)));
#endif
typedef _Float16 lz_h8 __attribute__((ext_vector_type(8)));

enum { H_A1 = 0, H_A2, H_E1, H_E2, H_S1, H_S2, H_S3, H_C1, H_C2, H_COUNT };
//                               A1 A2 E1 E2 S1 S2 S3 C1 C2
constexpr int H_KS[H_COUNT] = {  2, 2, 2, 1, 3, 2, 2, 3, 2 };
constexpr int H_NT[H_COUNT] = {  4, 2, 1, 1, 4, 4, 5, 4, 1 };
constexpr int h_frag_base(int layer) {
    int b = 0;
    for (int i = 0; i < layer; i++) b += H_KS[i] * H_NT[i];
    return b;
}
constexpr int H_FRAGS = h_frag_base(H_COUNT);  // 59
static_assert(H_FRAGS * 64 * 16 == LZ_HEAD_PACKED_F16_BYTES, "packed size mismatch with the header");

struct LzHead16Args {
    const float* emb[3];
    const int* offsets;
    const lz_h8* packed;
    const float* enc_a;
    const float* ind_code;
    const float* eye;
    float bound;
    float scale[12];
    uint32_t res[12];
};

template <int LAYER>
__device__ __forceinline__ void h_layer(const lz_h8* __restrict__ wl, int lane, const lz_h8 (&b)[H_KS[LAYER]], lz_f4 (&acc)[H_NT[LAYER]]) {
    constexpr int KS = H_KS[LAYER], NT = H_NT[LAYER];
    const lz_h8* frag = wl + h_frag_base(LAYER) * 64 + lane;
#pragma unroll
    for (int ks = 0; ks < KS; ks++)
#pragma unroll
        for (int ft = 0; ft < NT; ft++) acc[ft] = __builtin_amdgcn_mfma_f32_16x16x32_f16(frag[(ks * NT + ft) * 64], b[ks], acc[ft], 0, 0, 0);
}

// input feature held by k slot (k-step ks, lane group kg, j) of a B operand: chained from two D tiles / the gathered enc_x
__device__ __forceinline__ int h_chain(int ks, int kg, int j, int K) {   // two D tiles -> one B operand
    const int f = 16 * (2 * ks + (j >> 2)) + 4 * kg + (j & 3);
    return f < K ? f : -1;
}
__device__ __forceinline__ int h_encx(int ks, int kg, int j) {           // lane kg gathers features 4 i + kg, i = 8 ks + j < 9
    const int i = 8 * ks + j;
    return i < 9 ? 4 * i + kg : -1;
}

// round an f32 RESULT to half: the value must exist as an f32 first (torch's half ops with an f32 scalar compute in f32, round to f32, then
// to half).  Without the barrier the compiler may fold the multiply / subtract and the conversion into v_fma_mixlo_f16, which rounds the
// exact result ONCE -- a different value in ~2^-13 of the cases, and whether it does so depends on the surrounding code, so two kernels
// built from this same slice would disagree (the fused frame and the stand-alone head did, in a few pixels by 2e-7).
__device__ __forceinline__ _Float16 h_round(float v) {
    asm volatile("" : "+v"(v));
    return (_Float16)v;
}

// sigmoid whose result the caller rounds to half (autocast: sigmoid runs in half): hardware exp2 / reciprocal, a few f32 ulp, which the
// half rounding absorbs except on a rounding boundary -- 4 VALU instructions where the bit-reproducible lz_sigmoidf (polynomial exp +
// IEEE division) costs about 31, four times per slice in kernels that are bound by VALU issue
__device__ __forceinline__ float h_sigmoid(float x) { return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(x * -1.44269504088896340736f)); }

// exp and sqrt that autocast runs in f32 on a half-valued argument (sigma = exp(h[..., 0]), ||att||: network.py:302,308; both on torch's
// "autocast to float32" list): hardware v_exp_f32 / v_sqrt_f32 -- about 1 ulp, as good a float exp / sqrt as the reference's (CUDA's expf is
// 2 ulp) -- where the bit-reproducible lz_expf polynomial and the IEEE sqrtf expansion cost ~22 and ~10 instructions per slice in kernels
// bound by vector-instruction issue.  Every f16 kernel (inference slice, fused frame, recording forward) calls these, so they agree bit for bit.
__device__ __forceinline__ float h_exp32(float x) { return __builtin_amdgcn_exp2f(x * 1.44269504088896340736f); }
__device__ __forceinline__ float h_sqrt32(float x) { return __builtin_amdgcn_sqrtf(x); }

template <int KS, int NT>
__device__ __forceinline__ void h_layer_at(const lz_h8* __restrict__ frags, int lane, const lz_h8 (&b)[KS], lz_f4 (&acc)[NT]) {
    const lz_h8* frag = frags + lane;
#pragma unroll
    for (int ks = 0; ks < KS; ks++)
#pragma unroll
        for (int ft = 0; ft < NT; ft++) acc[ft] = __builtin_amdgcn_mfma_f32_16x16x32_f16(frag[(ks * NT + ft) * 64], b[ks], acc[ft], 0, 0, 0);
}

__device__ __forceinline__ _Float16 h_relu16(float v) { return (_Float16)(v > 0.0f ? v : 0.0f); }   // relu(half(v)) == half(relu(v))

// two D tiles of a layer -> one B operand of the next (ReLU + round to half = the half output of an autocast Linear + relu).
// The ReLU runs on the packed halves as a signed 16-bit max with zero -- a half is negative exactly when its bit pattern is a negative
// int16, and relu(half(v)) == half(relu(v)) -- one v_pk_max_i16 per two values where the f32 form costs a canonicalising max and the
// max itself per value (the f16 kernels are bound by VALU issue).
typedef _Float16 lz_h2v __attribute__((ext_vector_type(2)));
typedef short lz_s2v __attribute__((ext_vector_type(2)));
typedef uint32_t lz_u4v __attribute__((ext_vector_type(4)));
__device__ __forceinline__ uint32_t h_cvt2(float a, float b, bool relu) {
    typedef float lz_f2v __attribute__((ext_vector_type(2)));
    const lz_f2v f = {a, b};
    const lz_h2v h = __builtin_convertvector(f, lz_h2v);   // one v_cvt_pk_f16_f32 (round to nearest even)
    if (relu) {
        const lz_s2v z = {0, 0};
        return __builtin_bit_cast(uint32_t, __builtin_elementwise_max(__builtin_bit_cast(lz_s2v, h), z));
    }
    return __builtin_bit_cast(uint32_t, h);
}
// two f32 RESULTS -> packed halves, each through its own f32 value first (h_round's reason), one v_cvt_pk_f16_f32
__device__ __forceinline__ uint32_t h_round2(float a, float b) {
    asm volatile("" : "+v"(a), "+v"(b));
    return h_cvt2(a, b, false);
}
// enc_a * att, half * half -> half (network.py:291 under autocast with enc_a half): the exact product of two halves has 22 significant
// bits, so "f32 product, then round to half" is ONE rounding of the exact product = v_pk_mul_f16, two products per instruction.
// enca16: this lane's eight enc_a halves (features 16 (j >> 2) + 4 q + (j & 3)) from the LDS table
__device__ __forceinline__ lz_h8 h_encw(const int* __restrict__ tab, int q, const lz_h8& att) {
    typedef uint32_t lz_u2v __attribute__((ext_vector_type(2)));
    const lz_u2v lo = *reinterpret_cast<const lz_u2v*>(tab + LZ_LVTAB_ENCA16 + 2 * q), hi = *reinterpret_cast<const lz_u2v*>(tab + LZ_LVTAB_ENCA16 + 8 + 2 * q);
    const lz_u4v e = {lo[0], lo[1], hi[0], hi[1]};
    return __builtin_bit_cast(lz_h8, e) * att;
}
// the SH / ind_code half of colour_net.0's first B operand: SH components 4 q .. 4 q + 3 as halves, then ind_code (lanes q == 0)
template <typename ShFn>
__device__ __forceinline__ void h_sh_pk(const ShFn& f, int q, uint32_t (&w)[2]) {
    w[0] = h_cvt2(f.comp_qj(q, 0), f.comp_qj(q, 1), false);
    w[1] = h_cvt2(f.comp_qj(q, 2), f.comp_qj(q, 3), false);
}

__device__ __forceinline__ lz_h8 h_pair(const lz_f4& lo, const lz_f4& hi, bool relu) {
    const lz_u4v w = {h_cvt2(lo[0], lo[1], relu), h_cvt2(lo[2], lo[3], relu), h_cvt2(hi[0], hi[1], relu), h_cvt2(hi[2], hi[3], relu)};
    return __builtin_bit_cast(lz_h8, w);
}

struct LzHead16Ctx {
    const lz_h8* wl;        // LDS: packed A fragments
    const int* tab;         // LDS: the level table (lz_head_gather.h: LZ_LVTAB_*)
    const float* lenca;     // LDS: enc_a rounded to half [32]
    const float* emb[3];
    const float* ind_code;
    float bound, two_bound, eye_v, unc_const;
    bool has_eye;
};

struct LzHead16Out {
    float sigma, rgb[3], ambaud, eyeatt, unc;   // eyeatt is valid on lanes q == 0 only; sigma and rgb on every lane of the sample
    float own;                                  // this lane group's own transcendental: rgb[q] on q < 3, sigma on q == 3
};

constexpr int LZ_HEAD16_LDS_H8 = H_FRAGS * 64 + LZ_LVTAB_WORDS / 4;   // lz_h8 elements: fragments, then the level table (enc_a inside)

// stage weights + tables into LDS (all threads; caller synchronises afterwards) and fill the context
__device__ __forceinline__ void lz_head16_stage(const LzHead16Args& P, lz_h8* wl, uint32_t n_threads, LzHead16Ctx& hc) {
    float* tabf = reinterpret_cast<float*>(wl + H_FRAGS * 64);
    int* tab = reinterpret_cast<int*>(tabf);
    for (uint32_t i = threadIdx.x; i < (uint32_t)H_FRAGS * 64; i += n_threads) wl[i] = P.packed[i];
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

// R row tiles (R x 16 samples) through the head TOGETHER: every A fragment is read from LDS once and feeds R MFMAs (the weights are 59 KB
// per slice: with one row per call the LDS array is busy 56 % of the fused f16 frame kernel's cycles), and the R accumulation chains
// interleave.  The gathers stay one row at a time (36 loads in flight, the row's enc_x parked as two half operands = 8 registers), so
// the register peak is the gather's plus 8 per parked row.  Per row the arithmetic and its order are those of a single-row call.
template <int LAYER, int R>
__device__ __forceinline__ void h_layer_rows(const lz_h8* __restrict__ wl, int lane, const lz_h8 (&b)[R][H_KS[LAYER]], lz_f4 (&acc)[R][H_NT[LAYER]]) {
    constexpr int KS = H_KS[LAYER], NT = H_NT[LAYER];
    const lz_h8* frag = wl + h_frag_base(LAYER) * 64 + lane;
#pragma unroll
    for (int ks = 0; ks < KS; ks++)
#pragma unroll
        for (int ft = 0; ft < NT; ft++) {
            const lz_h8 a = frag[(ks * NT + ft) * 64];
#pragma unroll
            for (int r = 0; r < R; r++) acc[r][ft] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b[r][ks], acc[r][ft], 0, 0, 0);
        }
}

template <bool IN_RANGE = false, int R = 1, typename ShFn>   // SH(4) source: LzShFromDir (lz_head_slice.h) or the frame kernel's per-ray LDS copy
__device__ __forceinline__ void lz_head16_slice_rows(const LzHead16Ctx& hc, int lane, const float (&px)[R], const float (&py)[R], const float (&pz)[R],
                                                     ShFn (&shfn)[R], LzHead16Out (&out)[R]) {
    const int q = lane >> 4;
    const lz_f4 z4 = lz_f4{0, 0, 0, 0};
    // ---------------- gather (f32, the same code as lz_k_triplane_head: lz_head_gather.h): lane q holds enc_x features 4 i + q
    // enc_x as two half B operands (slot j of k-step ks <-> i = 8 ks + j); slot (1, q = 0, 1) is filled in for the sigma net
    lz_h8 bx[R][2];
#pragma unroll
    for (int r = 0; r < R; r++) {
        float encx[9];
        lz_head_gather<IN_RANGE, LZ_GATHER_PACK16>(hc.emb, hc.tab, px[r], py[r], pz[r], q, hc.bound, hc.two_bound, encx);
        // h_round2: every f32 feature exists first, then its half (no v_fma_mixlo_f16 with the interpolation's last fma)
        const lz_u4v w0 = {h_round2(encx[0], encx[1]), h_round2(encx[2], encx[3]), h_round2(encx[4], encx[5]), h_round2(encx[6], encx[7])};
        const lz_u4v w1 = {h_round2(encx[8], 0.0f), 0u, 0u, 0u};
        bx[r][0] = __builtin_bit_cast(lz_h8, w0);
        bx[r][1] = __builtin_bit_cast(lz_h8, w1);
    }

    // ---------------- audio channel attention: 36 -> 64 -> 32 ----------------
    lz_h8 att16[R];   // [4 t + r] = feature 16 t + 4 q + r
    {
        lz_f4 a1[R][4];
#pragma unroll
        for (int r = 0; r < R; r++)
#pragma unroll
            for (int t = 0; t < 4; t++) a1[r][t] = z4;
        h_layer_rows<H_A1, R>(hc.wl, lane, bx, a1);
        lz_h8 b2[R][2];
        lz_f4 a2[R][2];
#pragma unroll
        for (int r = 0; r < R; r++) {
            b2[r][0] = h_pair(a1[r][0], a1[r][1], true);
            b2[r][1] = h_pair(a1[r][2], a1[r][3], true);
            a2[r][0] = z4; a2[r][1] = z4;
        }
        h_layer_rows<H_A2, R>(hc.wl, lane, b2, a2);
#pragma unroll
        for (int r = 0; r < R; r++) att16[r] = h_pair(a2[r][0], a2[r][1], false);
    }
    // ambient_aud = || att ||_2 in f32 (norm is an autocast-to-f32 op): lane partial over its 8 features, then over q
#pragma unroll
    for (int r = 0; r < R; r++) {
        float ss = 0.0f;
#pragma unroll
        for (int k = 0; k < 8; k++) ss = lz_fmaf((float)att16[r][k], (float)att16[r][k], ss);
        ss += __shfl_xor(ss, 16, 64);
        ss += __shfl_xor(ss, 32, 64);
        out[r].ambaud = h_sqrt32(ss);
    }
    // ---------------- eye attention: 36 -> 16 -> 1, sigmoid (half) ----------------
    float eyeatt[R];
#pragma unroll
    for (int r = 0; r < R; r++) eyeatt[r] = 0.0f;
    if (hc.has_eye) {
        lz_f4 e1[R][1];
#pragma unroll
        for (int r = 0; r < R; r++) e1[r][0] = z4;
        h_layer_rows<H_E1, R>(hc.wl, lane, bx, e1);
        lz_h8 be[R][1];
        lz_f4 e2[R][1];
#pragma unroll
        for (int r = 0; r < R; r++) {
            be[r][0] = h_pair(e1[r][0], z4, true);
            e2[r][0] = z4;
        }
        h_layer_rows<H_E2, R>(hc.wl, lane, be, e2);
#pragma unroll
        for (int r = 0; r < R; r++) eyeatt[r] = (float)(_Float16)h_sigmoid((float)(_Float16)e2[r][0][0]);   // valid on lanes q == 0
    }
    // ---------------- sigma net: [enc_x 36 | enc_a * att 32 | eye * eye_att 1] -> 64 -> 64 -> 65 ----------------
    lz_h8 geo16[R][2];
    float spre[R];
    {
        lz_h8 b1[R][3];
        lz_f4 s1[R][4];
#pragma unroll
        for (int r = 0; r < R; r++) {
            b1[r][0] = bx[r][0];
            b1[r][1] = bx[r][1];
            b1[r][1][1] = (hc.has_eye && q == 0) ? h_round(hc.eye_v * eyeatt[r]) : (_Float16)0.0f;
            b1[r][2] = h_encw(hc.tab, q, att16[r]);
#pragma unroll
            for (int t = 0; t < 4; t++) s1[r][t] = z4;
        }
        h_layer_rows<H_S1, R>(hc.wl, lane, b1, s1);
        lz_h8 b2[R][2];
        lz_f4 s2[R][4];
#pragma unroll
        for (int r = 0; r < R; r++) {
            b2[r][0] = h_pair(s1[r][0], s1[r][1], true);
            b2[r][1] = h_pair(s1[r][2], s1[r][3], true);
#pragma unroll
            for (int t = 0; t < 4; t++) s2[r][t] = z4;
        }
        h_layer_rows<H_S2, R>(hc.wl, lane, b2, s2);
        lz_h8 b3[R][2];
        lz_f4 s3[R][5];
#pragma unroll
        for (int r = 0; r < R; r++) {
            b3[r][0] = h_pair(s2[r][0], s2[r][1], true);
            b3[r][1] = h_pair(s2[r][2], s2[r][3], true);
#pragma unroll
            for (int t = 0; t < 5; t++) s3[r][t] = z4;
        }
        h_layer_rows<H_S3, R>(hc.wl, lane, b3, s3);
#pragma unroll
        for (int r = 0; r < R; r++) {
            geo16[r][0] = h_pair(s3[r][0], s3[r][1], false);   // geo_feat, no activation (network.py:304)
            geo16[r][1] = h_pair(s3[r][2], s3[r][3], false);
            spre[r] = (float)(_Float16)s3[r][4][0];               // the sigma row (half): lanes q == 3 (row 12 of tile 4, lz_k_head_pack_f16)
        }
    }
    // ---------------- colour net: [SH 16 | geo 64 | ind 4] -> 64 -> 3 ----------------
    {
        lz_h8 b1[R][3];
        lz_f4 c1[R][4];
#pragma unroll
        for (int r = 0; r < R; r++) {
            ShFn f = shfn[r];              // a local copy: the polynomial values of LzShFromDir stay in registers (an array of them would not)
            f.prepare();
            uint32_t shw[2];
            h_sh_pk(f, q, shw);                                                        // SH 4 q + j, j < 4
            const lz_u4v w = {shw[0], shw[1], q == 0 ? (uint32_t)hc.tab[LZ_LVTAB_IND16] : 0u, q == 0 ? (uint32_t)hc.tab[LZ_LVTAB_IND16 + 1] : 0u};
            b1[r][0] = __builtin_bit_cast(lz_h8, w);
            b1[r][1] = geo16[r][0];
            b1[r][2] = geo16[r][1];
#pragma unroll
            for (int t = 0; t < 4; t++) c1[r][t] = z4;
        }
        h_layer_rows<H_C1, R>(hc.wl, lane, b1, c1);
        lz_h8 b2[R][2];
        lz_f4 c2[R][1];
#pragma unroll
        for (int r = 0; r < R; r++) {
            b2[r][0] = h_pair(c1[r][0], c1[r][1], true);
            b2[r][1] = h_pair(c1[r][2], c1[r][3], true);
            c2[r][0] = z4;
        }
        h_layer_rows<H_C2, R>(hc.wl, lane, b2, c2);
        // The four transcendentals of a sample, one per lane group and ONE instruction sequence for all of them: lanes q < 3 hold colour channel
        // q's pre-activation (register 0 of the colour_net.1 tile), lanes q == 3 the sigma row.  sigma = exp(x) (an autocast-to-f32 op: half in,
        // f32 out) and sigmoid(x) = 1 / (1 + exp(-x)) both start with exp2 of x times +-log2(e) -- the same multiply and the same exp2 as
        // h_exp32 / h_sigmoid, so the same bits -- and the colour lanes go on through network.py:275 in half (* 1.002, - 0.001, each rounded).
        // Then the sample's lanes fetch each other's result.  (Round 4; before: three sigmoid chains + one exp on every lane.)
#pragma unroll
        for (int r = 0; r < R; r++) {
            const float pre = q == 3 ? spre[r] : (float)(_Float16)c2[r][0][0];
            const float e = __builtin_amdgcn_exp2f(pre * (q == 3 ? 1.44269504088896340736f : -1.44269504088896340736f));
            const _Float16 sg = (_Float16)__builtin_amdgcn_rcpf(1.0f + e);
            const _Float16 t1 = h_round((float)sg * 1.002f);
            const float col = (float)h_round((float)t1 - 0.001f);
            const float own = q == 3 ? e : col;
            const int s0 = lane & 15;
            out[r].own = own;
            out[r].sigma = __shfl(own, s0 + 48, 64);
#pragma unroll
            for (int c = 0; c < 3; c++) out[r].rgb[c] = __shfl(own, s0 + 16 * c, 64);
        }
    }
#pragma unroll
    for (int r = 0; r < R; r++) {
        out[r].eyeatt = eyeatt[r];
        out[r].unc = hc.unc_const;
    }
}

template <bool IN_RANGE = false, typename ShFn>
__device__ __forceinline__ void lz_head16_slice(const LzHead16Ctx& hc, int lane, float px, float py, float pz, ShFn shfn, LzHead16Out& out) {
    const float x[1] = {px}, y[1] = {py}, z[1] = {pz};
    ShFn f[1] = {shfn};
    LzHead16Out o[1];
    lz_head16_slice_rows<IN_RANGE, 1>(hc, lane, x, y, z, f, o);
    out = o[0];
}
#endif
