// lz_head_fwd16_chain.h -- the MLP part of the f16 TRAINING forward (autocast arithmetic on v_mfma_f32_16x16x32_f16, 16-sample slices;
// rounding sequence: lz_head_f16_slice.h / lz_head_f16.hip) as ONE function of the slice's enc_x operand, shared by
//   * the recording forward (lz_head_rec16.hip): its sink stores the layer inputs and the state row to memory, and
//   * the recomputing backward (lz_head_rec.hip, RC): its sink keeps the same words in registers -- the backward then needs no record,
//     only the 80 bytes per sample of enc_x halves the light forward left behind (round 5).
// One source for both, so the values the backward differentiates are bit for bit the values the forward produced.
//
// Sink interface (every call with compile-time constant indices):
//   x_pair_h8(pair, b)      X-record pair `pair` (include/lzzx_nerf_hip.h LZ_R16_*: dword r = {low tile register r, high tile register r})
//                           from a B operand (slots 0..3 = low tile, 4..7 = high tile)
//   x_pair_f(pair, l0..l3, h0..h3)   the same from eight f32 values, each rounded to half on its own (h_round)
//   s_pair_h8(pair16, b)    state-row pair (LZ_S16_E1 / U1 / C1)
//   s_att(w0, w1)           att as f32 (LZ_ST_ATT)
#ifndef LZ_HEAD_FWD16_CHAIN_H
#define LZ_HEAD_FWD16_CHAIN_H
#include "lz_head_f16_slice.h"

typedef float lz_v4 __attribute__((ext_vector_type(4)));
typedef uint32_t lz_u4 __attribute__((ext_vector_type(4)));
#define LZ_UNC16_FRAGS 5   // unc_net.0: 2 k-steps x 2 feature tiles; unc_net.1: 1 x 1

// a B operand -> the record's pair layout: dword r = {low tile register r, high tile register r}
__device__ __forceinline__ lz_v4 lz_pair_words_h8(const lz_h8& b) {
    const lz_u4 p = __builtin_bit_cast(lz_u4, b);
    const lz_u4 w = {__builtin_amdgcn_perm(p[2], p[0], 0x05040100u), __builtin_amdgcn_perm(p[2], p[0], 0x07060302u),
                     __builtin_amdgcn_perm(p[3], p[1], 0x05040100u), __builtin_amdgcn_perm(p[3], p[1], 0x07060302u)};
    return __builtin_bit_cast(lz_v4, w);
}
__device__ __forceinline__ float lz_pack_h2f(float lo, float hi) {
    typedef _Float16 lz_h2 __attribute__((ext_vector_type(2)));
    const lz_h2 v = {h_round(lo), h_round(hi)};   // the same halves the B operands hold (no fused single rounding)
    return __builtin_bit_cast(float, v);
}
__device__ __forceinline__ lz_v4 lz_pair_words_f(float l0, float l1, float l2, float l3, float h0, float h1, float h2, float h3) {
    return lz_v4{lz_pack_h2f(l0, h0), lz_pack_h2f(l1, h1), lz_pack_h2f(l2, h2), lz_pack_h2f(l3, h3)};
}
// bit 8 p + j of a layer's mask <-> slot j of its B operand p <-> chained index 4 t + r (lz_head_bwd_common.h: lz_mask_pos).  The operand is
// what ReLU left: halves >= +0, so "positive" is "bit pattern not zero" -- an unsigned 16-bit min with 1 per packed pair, then the
// eight 0 / 1 halves are folded into one byte (10 instructions; a compare + select + or per half costs 17)
__device__ __forceinline__ uint32_t lz_mask_h8(const lz_h8& b) {
    // v_pk_min_u16 spelled out: the compiler expands the vector min against the constant into a compare, a select and a pack per half
    // (76 compares and 96 selects per slice in the ISA of round 2), four times the instructions of the packed form
    const lz_u4 w = __builtin_bit_cast(lz_u4, b);   // whole-vector cast only: a bit_cast of a single vector ELEMENT to a 2-vector was miscompiled here (see lz_head_rec.hip)
    uint32_t q[4];
#pragma unroll
    for (int d = 0; d < 4; d++) asm("v_pk_min_u16 %0, %1, %2" : "=v"(q[d]) : "v"(w[d]), "v"(0x00010001u));       // halves 2 d, 2 d + 1 of dword d -> 0 / 1
    const uint32_t m = q[0] | (q[1] << 2) | (q[2] << 4) | (q[3] << 6);                                          // bits 2 d and 16 + 2 d
    return (m | (m >> 15)) & 0xffu;
}

struct LzFwd16Out {
    float norm, eyeatt, upre, sigma, cpre[3];                  // ||att||, eye attention, pre-activations of unc / (sigma = exp) / colour: on every lane of the sample
    uint32_t mk_a1, mk_s1, mk_s2, mk_c1, mk_u1, mk_e1;         // ReLU masks of this lane's values (bit 8 p + j <-> slot j of operand p)
};

// bx: the slice's enc_x operand (lane (s, q): features 4 i + q as halves, slot j of k-step ks <-> i = 8 ks + j; bx[1] slots 1.. are zero);
// (dx, dy, dz): the sample's view direction; indq = ind_code[q] (0 without one); wl_unc: the five unc_net fragments in LDS
template <typename Sink>
__device__ __forceinline__ void lz_fwd16_chain(const LzHead16Ctx& hc, const lz_h8* __restrict__ wl_unc, int lane, const lz_h8 (&bx)[2], float dx, float dy,
                                               float dz, float indq, Sink& sink, LzFwd16Out& out) {
    const int s = lane & 15, q = lane >> 4;
    // ---------------- audio channel attention: 36 -> 64 -> 32 ----------------
    lz_h8 att16;
    {
        lz_f4 a1[4] = {lz_f4{0, 0, 0, 0}, lz_f4{0, 0, 0, 0}, lz_f4{0, 0, 0, 0}, lz_f4{0, 0, 0, 0}};
        h_layer<H_A1>(hc.wl, lane, bx, a1);
        const lz_h8 b2[2] = {h_pair(a1[0], a1[1], true), h_pair(a1[2], a1[3], true)};
        out.mk_a1 = lz_mask_h8(b2[0]) | (lz_mask_h8(b2[1]) << 8);
        sink.x_pair_h8(LZ_R16_X_A1 / 2, b2[0]);
        sink.x_pair_h8(LZ_R16_X_A1 / 2 + 1, b2[1]);
        lz_f4 a2[2] = {lz_f4{0, 0, 0, 0}, lz_f4{0, 0, 0, 0}};
        h_layer<H_A2>(hc.wl, lane, b2, a2);
        att16 = h_pair(a2[0], a2[1], false);
    }
    sink.s_att(lz_v4{(float)att16[0], (float)att16[1], (float)att16[2], (float)att16[3]},
               lz_v4{(float)att16[4], (float)att16[5], (float)att16[6], (float)att16[7]});
    float ss = 0.0f;
#pragma unroll
    for (int k = 0; k < 8; k++) ss = lz_fmaf((float)att16[k], (float)att16[k], ss);
    ss += __shfl_xor(ss, 16, 64);
    ss += __shfl_xor(ss, 32, 64);
    out.norm = h_sqrt32(ss);
    // ---------------- eye attention ----------------
    float eyeatt = 0.0f;
    out.mk_e1 = 0;
    if (hc.has_eye) {
        lz_f4 e1[1] = {lz_f4{0, 0, 0, 0}};
        h_layer<H_E1>(hc.wl, lane, bx, e1);
        const lz_f4 z = lz_f4{0, 0, 0, 0};
        const lz_h8 be[1] = {h_pair(e1[0], z, true)};
        out.mk_e1 = lz_mask_h8(be[0]) & 0xfu;
        sink.s_pair_h8(LZ_S16_E1 / 16, be[0]);
        lz_f4 e2[1] = {lz_f4{0, 0, 0, 0}};
        h_layer<H_E2>(hc.wl, lane, be, e2);
        eyeatt = (float)(_Float16)h_sigmoid((float)(_Float16)e2[0][0]);   // lanes q == 0
        eyeatt = __shfl(eyeatt, s, 64);
    }
    out.eyeatt = eyeatt;
    // ---------------- uncertainty (training): 36 -> 32 -> 1, softplus in f32 on the half pre-activation ----------------
    {
        lz_f4 u1[2] = {lz_f4{0, 0, 0, 0}, lz_f4{0, 0, 0, 0}};
        h_layer_at<2, 2>(wl_unc, lane, bx, u1);
        const lz_h8 bu[1] = {h_pair(u1[0], u1[1], true)};
        out.mk_u1 = lz_mask_h8(bu[0]);
        sink.s_pair_h8(LZ_S16_U1 / 16, bu[0]);
        lz_f4 u2[1] = {lz_f4{0, 0, 0, 0}};
        h_layer_at<1, 1>(wl_unc + 4 * 64, lane, bu, u2);
        out.upre = __shfl((float)(_Float16)u2[0][0], s, 64);
    }
    // ---------------- sigma net ----------------
    lz_h8 geo16[2];
    float spre;
    {
        lz_h8 b1[3];
        b1[0] = bx[0];
        b1[1] = bx[1];
        b1[1][1] = (hc.has_eye && q == 0) ? h_round(hc.eye_v * eyeatt) : (_Float16)0.0f;
        b1[2] = h_encw(hc.tab, q, att16);
        // sigma_net.0 input in the record's arrangement (lz_head_rec.hip: tiles 0, 1 enc_x, tile 2 feature 32 + q and the eye term,
        // tiles 3, 4 enc_a * att); the conversions to half repeat the operands' halves, value for value
        sink.x_pair_f(LZ_R16_X_SIG0 / 2, (float)bx[0][0], (float)bx[0][2], (float)bx[0][4], (float)bx[0][6], (float)bx[0][1], (float)bx[0][3],
                      (float)bx[0][5], (float)bx[0][7]);
        sink.x_pair_f(LZ_R16_X_SIG0 / 2 + 1, (float)bx[1][0], (float)b1[1][1], 0.0f, 0.0f, (float)b1[2][0], (float)b1[2][1], (float)b1[2][2], (float)b1[2][3]);
        sink.x_pair_f(LZ_R16_X_SIG0 / 2 + 2, (float)b1[2][4], (float)b1[2][5], (float)b1[2][6], (float)b1[2][7], 0.0f, 0.0f, 0.0f, 0.0f);
        lz_f4 s1[4] = {lz_f4{0, 0, 0, 0}, lz_f4{0, 0, 0, 0}, lz_f4{0, 0, 0, 0}, lz_f4{0, 0, 0, 0}};
        h_layer<H_S1>(hc.wl, lane, b1, s1);
        const lz_h8 b2[2] = {h_pair(s1[0], s1[1], true), h_pair(s1[2], s1[3], true)};
        out.mk_s1 = lz_mask_h8(b2[0]) | (lz_mask_h8(b2[1]) << 8);
        sink.x_pair_h8(LZ_R16_X_S1 / 2, b2[0]);
        sink.x_pair_h8(LZ_R16_X_S1 / 2 + 1, b2[1]);
        lz_f4 s2[4] = {lz_f4{0, 0, 0, 0}, lz_f4{0, 0, 0, 0}, lz_f4{0, 0, 0, 0}, lz_f4{0, 0, 0, 0}};
        h_layer<H_S2>(hc.wl, lane, b2, s2);
        const lz_h8 b3[2] = {h_pair(s2[0], s2[1], true), h_pair(s2[2], s2[3], true)};
        out.mk_s2 = lz_mask_h8(b3[0]) | (lz_mask_h8(b3[1]) << 8);
        sink.x_pair_h8(LZ_R16_X_S2C / 2, b3[0]);
        sink.x_pair_h8(LZ_R16_X_S2C / 2 + 1, b3[1]);
        lz_f4 s3[5] = {lz_f4{0, 0, 0, 0}, lz_f4{0, 0, 0, 0}, lz_f4{0, 0, 0, 0}, lz_f4{0, 0, 0, 0}, lz_f4{0, 0, 0, 0}};
        h_layer<H_S3>(hc.wl, lane, b3, s3);
        geo16[0] = h_pair(s3[0], s3[1], false);
        geo16[1] = h_pair(s3[2], s3[3], false);
        spre = __shfl((float)(_Float16)s3[4][0], s + 48, 64);      // the sigma row sits at row 12 of tile 4: lanes q == 3 (lz_k_head_pack_f16)
    }
    out.sigma = h_exp32(spre);
    // ---------------- colour net ----------------
    {
        auto shfn = lz_sh_from_dir([&](float& ox, float& oy, float& oz) { ox = dx; oy = dy; oz = dz; });
        shfn.prepare();
        lz_h8 b1[3];
        {
            uint32_t shw[2];
            h_sh_pk(shfn, q, shw);
            const lz_u4v w = {shw[0], shw[1], q == 0 ? (uint32_t)hc.tab[LZ_LVTAB_IND16] : 0u, q == 0 ? (uint32_t)hc.tab[LZ_LVTAB_IND16 + 1] : 0u};
            b1[0] = __builtin_bit_cast(lz_h8, w);
        }
        b1[1] = geo16[0];
        b1[2] = geo16[1];
        // colour_net.0's SH / ind columns as the record keeps them: SH component 4 r + q at column 4 q + r, ind_code[q] at column 4 q
        sink.x_pair_f(LZ_R16_X_S2C / 2 + 2, shfn.comp_iq(0, q), shfn.comp_iq(1, q), shfn.comp_iq(2, q), shfn.comp_iq(3, q), indq, 0.0f, 0.0f, 0.0f);
        lz_f4 c1[4] = {lz_f4{0, 0, 0, 0}, lz_f4{0, 0, 0, 0}, lz_f4{0, 0, 0, 0}, lz_f4{0, 0, 0, 0}};
        h_layer<H_C1>(hc.wl, lane, b1, c1);
        const lz_h8 b2[2] = {h_pair(c1[0], c1[1], true), h_pair(c1[2], c1[3], true)};
        out.mk_c1 = lz_mask_h8(b2[0]) | (lz_mask_h8(b2[1]) << 8);
        sink.s_pair_h8(LZ_S16_C1 / 16, b2[0]);
        sink.s_pair_h8(LZ_S16_C1 / 16 + 1, b2[1]);
        lz_f4 c2[1] = {lz_f4{0, 0, 0, 0}};
        h_layer<H_C2>(hc.wl, lane, b2, c2);
#pragma unroll
        for (int c = 0; c < 3; c++) out.cpre[c] = __shfl((float)(_Float16)c2[0][0], s + 16 * c, 64);   // channel c sits at row 4 c: register 0 of lanes q == c
    }
}
#endif
