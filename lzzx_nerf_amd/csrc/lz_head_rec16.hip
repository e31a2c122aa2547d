// lz_head_rec16.hip -- the recording forward of the training head on the f16 matrix cores: the forward of the reference's usual training
// mode (`-O` = --fp16: torch.cuda.amp.autocast around NeRFNetwork.forward, nerf_triplane/network.py:252-311, TrainerUtil.py:865) with
// the rounding sequence of lz_head_f16_slice.h, writing the f16 records and the state row that lz_triplane_head_backward_recorded
// (record_f16 = 1) and lz_triplane_head_grad_w_f16 consume (layouts: include/lzzx_nerf_hip.h LZ_R16_*, LZ_S16_*).
//
// The layer inputs a half Linear sees ARE the halves this kernel holds as MFMA B operands, so the record costs two byte-permutes per
// dword on top of the inference slice; 64 MFMAs per 16 samples instead of 379 f32 ones.  The uncertainty net (training only:
// network.py:241-249) adds five fragments that are packed separately (lz_head_pack_unc_f16), so the inference image and kernels stay
// as they are.  sigma / rgb / ambient outputs have the bits of lz_k_triplane_head_f16 on the same inputs.
#include "lz_head_f16_slice.h"
#include "lz_head_bwd_common.h"   // lz_blk / lz_tcol: the blocked record layout

typedef float lz_v4 __attribute__((ext_vector_type(4)));
typedef uint32_t lz_u4 __attribute__((ext_vector_type(4)));

#define LZ_FREC16_WG 1024
#define LZ_UNC16_FRAGS 5   // unc_net.0: 2 k-steps x 2 feature tiles; unc_net.1: 1 x 1

extern "C" uint32_t lz_head_packed_unc_size_f16(void) { return (uint32_t)LZ_UNC16_FRAGS * 64u * 16u; }

__global__ void __launch_bounds__(64) lz_k_head_pack_unc_f16(const float* __restrict__ unc0, const float* __restrict__ unc1,
                                                              _Float16* __restrict__ packed) {
    const int frag = blockIdx.x, lane = threadIdx.x;
    const int kg = lane >> 4;
    for (int j = 0; j < 8; j++) {
        float v = 0.0f;
        if (frag < 4) {   // unc_net.0 [32, 36]: fragment (ks, ft)
            const int ks = frag >> 1, ft = frag & 1;
            const int kf = h_encx(ks, kg, j);
            if (kf >= 0) v = unc0[(16 * ft + (lane & 15)) * 36 + kf];
        } else {          // unc_net.1 [1, 32]: row 0 only
            const int kf = h_chain(0, kg, j, 32);
            if (kf >= 0 && (lane & 15) == 0) v = unc1[kf];
        }
        packed[((size_t)frag * 64 + lane) * 8 + j] = (_Float16)v;
    }
}

extern "C" int lz_head_pack_unc_f16(const float* unc0, const float* unc1, void* packed_unc, lz_stream_t stream) {
    LZ_REQUIRE(unc0 && unc1 && packed_unc, LZ_ERR_BAD_ARGUMENT, "head_pack_unc_f16: null argument");
    hipLaunchKernelGGL(lz_k_head_pack_unc_f16, dim3(LZ_UNC16_FRAGS), dim3(64), 0, lz_st(stream), unc0, unc1, reinterpret_cast<_Float16*>(packed_unc));
    LZ_CHECK_LAUNCH("head_pack_unc_f16");
    return LZ_OK;
}

// a B operand (two tiles: slots 0..3 = low tile registers 0..3, slots 4..7 = high tile) -> the record's pair layout (dword r = {low
// tile register r, high tile register r}); rowq = the sample's row in dwords + 4 q
__device__ __forceinline__ void lz_dump_pair_h8(float* __restrict__ rowq, int pair, const lz_h8& b) {
    const lz_u4 p = __builtin_bit_cast(lz_u4, b);
    const lz_u4 w = {__builtin_amdgcn_perm(p[2], p[0], 0x05040100u), __builtin_amdgcn_perm(p[2], p[0], 0x07060302u),
                     __builtin_amdgcn_perm(p[3], p[1], 0x05040100u), __builtin_amdgcn_perm(p[3], p[1], 0x07060302u)};
    LZ_REC_STORE(__builtin_bit_cast(lz_v4, w), reinterpret_cast<lz_v4*>(rowq + 256 * pair));
}
__device__ __forceinline__ float lz_pack_h2f(float lo, float hi) {
    typedef _Float16 lz_h2 __attribute__((ext_vector_type(2)));
    const lz_h2 v = {h_round(lo), h_round(hi)};   // the same halves the B operands hold (no fused single rounding)
    return __builtin_bit_cast(float, v);
}
__device__ __forceinline__ void lz_dump_pair_f(float* __restrict__ rowq, int pair, float l0, float l1, float l2, float l3, float h0, float h1,
                                               float h2, float h3) {
    lz_v4 w = {lz_pack_h2f(l0, h0), lz_pack_h2f(l1, h1), lz_pack_h2f(l2, h2), lz_pack_h2f(l3, h3)};
    LZ_REC_STORE(w, reinterpret_cast<lz_v4*>(rowq + 256 * pair));
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

__global__ void __launch_bounds__(LZ_FREC16_WG, 1)
lz_k_triplane_head_forward_rec16(LzHead16Args P, const lz_h8* __restrict__ packed_unc, const float* __restrict__ xyzs,
                                 const float* __restrict__ dirs, uint32_t M, float* __restrict__ sigmas, float* __restrict__ rgbs,
                                 float* __restrict__ amb_aud, float* __restrict__ amb_eye, float* __restrict__ unc_out, float* __restrict__ rec,
                                 float* __restrict__ st) {
    __shared__ lz_h8 wl[LZ_HEAD16_LDS_H8 + LZ_UNC16_FRAGS * 64];
    const uint32_t n_slices = (M + 15) / 16;
    const uint32_t slice_lo = (uint32_t)(((uint64_t)n_slices * blockIdx.x) / gridDim.x);
    const uint32_t slice_hi = (uint32_t)(((uint64_t)n_slices * (blockIdx.x + 1)) / gridDim.x);
    if (slice_lo >= slice_hi) return;
    LzHead16Ctx hc;
    lz_head16_stage(P, wl, LZ_FREC16_WG, hc);
    lz_h8* wl_unc = wl + LZ_HEAD16_LDS_H8;
    for (uint32_t i = threadIdx.x; i < (uint32_t)LZ_UNC16_FRAGS * 64; i += LZ_FREC16_WG) wl_unc[i] = packed_unc[i];
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const int s = lane & 15, q = lane >> 4;
    const float indq = hc.ind_code ? hc.ind_code[q] : 0.0f;
    int* queue = reinterpret_cast<int*>(wl + H_FRAGS * 64) + LZ_LVTAB_QUEUE;
    auto grab = [&]() -> int {
        int sl = 0;
        if (lane == 0) sl = atomicAdd(queue, 1);
        return __builtin_amdgcn_readfirstlane(sl);
    };
    auto row_of = [&](int sl) -> uint32_t {
        uint32_t gs = slice_lo + (uint32_t)sl;
        if (gs >= slice_hi) gs = slice_hi - 1;
        const uint32_t b = gs * 16 + s;
        return b < M ? b : M - 1;
    };
    // position / direction of slice n + 1 are requested at the top of slice n, before its stores (lz_head_rec.hip)
    int slice = grab();
    float px, py, pz, dx, dy, dz;
    {
        const size_t r0 = row_of(slice);
        px = xyzs[r0 * 3]; py = xyzs[r0 * 3 + 1]; pz = xyzs[r0 * 3 + 2];
        dx = dirs[r0 * 3]; dy = dirs[r0 * 3 + 1]; dz = dirs[r0 * 3 + 2];
    }
    for (;;) {
        if (slice_lo + (uint32_t)slice >= slice_hi) break;
        const size_t row = row_of(slice);   // lanes past the end repeat the last row: the same values are stored again
        float* rb = lz_blk(rec, slice_lo + (uint32_t)slice, LZ_BWD_REC16 / 2, s);
        float* sb = lz_blk(st, slice_lo + (uint32_t)slice, LZ_FWD_STATE16, s);

        float encx[9];
        lz_head_gather(hc.emb, hc.tab, px, py, pz, q, hc.bound, hc.two_bound, encx);
        const float cdx = dx, cdy = dy, cdz = dz;
        const int next = grab();
        {
            const size_t r1 = row_of(next);
            px = xyzs[r1 * 3]; py = xyzs[r1 * 3 + 1]; pz = xyzs[r1 * 3 + 2];
            dx = dirs[r1 * 3]; dy = dirs[r1 * 3 + 1]; dz = dirs[r1 * 3 + 2];
        }
        __builtin_amdgcn_sched_barrier(0);
        lz_h8 bx[2];
        {   // as in lz_head16_slice
            const lz_u4v w0 = {h_round2(encx[0], encx[1]), h_round2(encx[2], encx[3]), h_round2(encx[4], encx[5]), h_round2(encx[6], encx[7])};
            const lz_u4v w1 = {h_round2(encx[8], 0.0f), 0u, 0u, 0u};
            bx[0] = __builtin_bit_cast(lz_h8, w0);
            bx[1] = __builtin_bit_cast(lz_h8, w1);
        }

        // ---------------- audio channel attention: 36 -> 64 -> 32 ----------------
        lz_h8 att16;
        uint32_t mk_a1;
        {
            lz_f4 a1[4] = {lz_f4{0, 0, 0, 0}, lz_f4{0, 0, 0, 0}, lz_f4{0, 0, 0, 0}, lz_f4{0, 0, 0, 0}};
            h_layer<H_A1>(hc.wl, lane, bx, a1);
            const lz_h8 b2[2] = {h_pair(a1[0], a1[1], true), h_pair(a1[2], a1[3], true)};
            mk_a1 = lz_mask_h8(b2[0]) | (lz_mask_h8(b2[1]) << 8);
            lz_dump_pair_h8(rb + 4 * q, LZ_R16_X_A1 / 2, b2[0]);
            lz_dump_pair_h8(rb + 4 * q, LZ_R16_X_A1 / 2 + 1, b2[1]);
            lz_f4 a2[2] = {lz_f4{0, 0, 0, 0}, lz_f4{0, 0, 0, 0}};
            h_layer<H_A2>(hc.wl, lane, b2, a2);
            att16 = h_pair(a2[0], a2[1], false);
        }
        {
            lz_v4 w0 = {(float)att16[0], (float)att16[1], (float)att16[2], (float)att16[3]};
            lz_v4 w1 = {(float)att16[4], (float)att16[5], (float)att16[6], (float)att16[7]};
            LZ_REC_STORE(w0, reinterpret_cast<lz_v4*>(sb + lz_tcol(LZ_ST_ATT + 4 * q)));
            LZ_REC_STORE(w1, reinterpret_cast<lz_v4*>(sb + lz_tcol(LZ_ST_ATT + 16 + 4 * q)));
        }
        float ss = 0.0f;
#pragma unroll
        for (int k = 0; k < 8; k++) ss = lz_fmaf((float)att16[k], (float)att16[k], ss);
        ss += __shfl_xor(ss, 16, 64);
        ss += __shfl_xor(ss, 32, 64);
        const float norm = h_sqrt32(ss);
        // ---------------- eye attention ----------------
        float eyeatt = 0.0f;
        uint32_t mk_e1 = 0;
        if (hc.has_eye) {
            lz_f4 e1[1] = {lz_f4{0, 0, 0, 0}};
            h_layer<H_E1>(hc.wl, lane, bx, e1);
            const lz_f4 z = lz_f4{0, 0, 0, 0};
            const lz_h8 be[1] = {h_pair(e1[0], z, true)};
            mk_e1 = lz_mask_h8(be[0]) & 0xfu;
            lz_dump_pair_h8(sb + 4 * q, LZ_S16_E1 / 16, be[0]);
            lz_f4 e2[1] = {lz_f4{0, 0, 0, 0}};
            h_layer<H_E2>(hc.wl, lane, be, e2);
            eyeatt = (float)(_Float16)h_sigmoid((float)(_Float16)e2[0][0]);   // lanes q == 0
            eyeatt = __shfl(eyeatt, s, 64);
        }
        // ---------------- uncertainty (training): 36 -> 32 -> 1, softplus in f32 on the half pre-activation ----------------
        float upre;
        uint32_t mk_u1;
        {
            lz_f4 u1[2] = {lz_f4{0, 0, 0, 0}, lz_f4{0, 0, 0, 0}};
            h_layer_at<2, 2>(wl_unc, lane, bx, u1);
            const lz_h8 bu[1] = {h_pair(u1[0], u1[1], true)};
            mk_u1 = lz_mask_h8(bu[0]);
            lz_dump_pair_h8(sb + 4 * q, LZ_S16_U1 / 16, bu[0]);
            lz_f4 u2[1] = {lz_f4{0, 0, 0, 0}};
            h_layer_at<1, 1>(wl_unc + 4 * 64, lane, bu, u2);
            upre = __shfl((float)(_Float16)u2[0][0], s, 64);
        }
        // ---------------- sigma net ----------------
        lz_h8 geo16[2];
        float spre;
        uint32_t mk_s1, mk_s2;
        {
            lz_h8 b1[3];
            b1[0] = bx[0];
            b1[1] = bx[1];
            b1[1][1] = (hc.has_eye && q == 0) ? h_round(hc.eye_v * eyeatt) : (_Float16)0.0f;
            b1[2] = h_encw(hc.tab, q, att16);
            // sigma_net.0 input in the record's arrangement (lz_head_rec.hip: tiles 0, 1 enc_x, tile 2 feature 32 + q and the eye term,
            // tiles 3, 4 enc_a * att); the conversions to half repeat the ones above, value for value
            lz_dump_pair_f(rb + 4 * q, LZ_R16_X_SIG0 / 2, encx[0], encx[2], encx[4], encx[6], encx[1], encx[3], encx[5], encx[7]);
            lz_dump_pair_f(rb + 4 * q, LZ_R16_X_SIG0 / 2 + 1, encx[8], (float)b1[1][1], 0.0f, 0.0f, (float)b1[2][0], (float)b1[2][1], (float)b1[2][2],
                           (float)b1[2][3]);
            lz_dump_pair_f(rb + 4 * q, LZ_R16_X_SIG0 / 2 + 2, (float)b1[2][4], (float)b1[2][5], (float)b1[2][6], (float)b1[2][7], 0.0f, 0.0f, 0.0f, 0.0f);
            lz_f4 s1[4] = {lz_f4{0, 0, 0, 0}, lz_f4{0, 0, 0, 0}, lz_f4{0, 0, 0, 0}, lz_f4{0, 0, 0, 0}};
            h_layer<H_S1>(hc.wl, lane, b1, s1);
            const lz_h8 b2[2] = {h_pair(s1[0], s1[1], true), h_pair(s1[2], s1[3], true)};
            mk_s1 = lz_mask_h8(b2[0]) | (lz_mask_h8(b2[1]) << 8);
            lz_dump_pair_h8(rb + 4 * q, LZ_R16_X_S1 / 2, b2[0]);
            lz_dump_pair_h8(rb + 4 * q, LZ_R16_X_S1 / 2 + 1, b2[1]);
            lz_f4 s2[4] = {lz_f4{0, 0, 0, 0}, lz_f4{0, 0, 0, 0}, lz_f4{0, 0, 0, 0}, lz_f4{0, 0, 0, 0}};
            h_layer<H_S2>(hc.wl, lane, b2, s2);
            const lz_h8 b3[2] = {h_pair(s2[0], s2[1], true), h_pair(s2[2], s2[3], true)};
            mk_s2 = lz_mask_h8(b3[0]) | (lz_mask_h8(b3[1]) << 8);
            lz_dump_pair_h8(rb + 4 * q, LZ_R16_X_S2C / 2, b3[0]);
            lz_dump_pair_h8(rb + 4 * q, LZ_R16_X_S2C / 2 + 1, b3[1]);
            lz_f4 s3[5] = {lz_f4{0, 0, 0, 0}, lz_f4{0, 0, 0, 0}, lz_f4{0, 0, 0, 0}, lz_f4{0, 0, 0, 0}, lz_f4{0, 0, 0, 0}};
            h_layer<H_S3>(hc.wl, lane, b3, s3);
            geo16[0] = h_pair(s3[0], s3[1], false);
            geo16[1] = h_pair(s3[2], s3[3], false);
            spre = __shfl((float)(_Float16)s3[4][0], s + 48, 64);      // the sigma row sits at row 12 of tile 4: lanes q == 3 (lz_k_head_pack_f16)
        }
        const float sigma = h_exp32(spre);
        // ---------------- colour net ----------------
        float cpre[3];
        uint32_t mk_c1;
        {
            auto shfn = lz_sh_from_dir([&](float& ox, float& oy, float& oz) { ox = cdx; oy = cdy; oz = cdz; });
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
            lz_dump_pair_f(rb + 4 * q, LZ_R16_X_S2C / 2 + 2, shfn.comp_iq(0, q), shfn.comp_iq(1, q), shfn.comp_iq(2, q), shfn.comp_iq(3, q), indq, 0.0f, 0.0f,
                           0.0f);
            lz_f4 c1[4] = {lz_f4{0, 0, 0, 0}, lz_f4{0, 0, 0, 0}, lz_f4{0, 0, 0, 0}, lz_f4{0, 0, 0, 0}};
            h_layer<H_C1>(hc.wl, lane, b1, c1);
            const lz_h8 b2[2] = {h_pair(c1[0], c1[1], true), h_pair(c1[2], c1[3], true)};
            mk_c1 = lz_mask_h8(b2[0]) | (lz_mask_h8(b2[1]) << 8);
            lz_dump_pair_h8(sb + 4 * q, LZ_S16_C1 / 16, b2[0]);
            lz_dump_pair_h8(sb + 4 * q, LZ_S16_C1 / 16 + 1, b2[1]);
            lz_f4 c2[1] = {lz_f4{0, 0, 0, 0}};
            h_layer<H_C2>(hc.wl, lane, b2, c2);
#pragma unroll
            for (int c = 0; c < 3; c++) cpre[c] = __shfl((float)(_Float16)c2[0][0], s + 16 * c, 64);   // channel c sits at row 4 c: register 0 of lanes q == c
        }
        // ---------------- state words, outputs (the four lanes of a sample store the same values) ----------------
        {
            const float sc = q == 0 ? norm : (q == 1 ? eyeatt : (q == 2 ? upre : sigma));
            lz_v4 w = {__uint_as_float(mk_a1 | (mk_s1 << 16)), __uint_as_float(mk_s2 | (mk_c1 << 16)), __uint_as_float(mk_u1 | (mk_e1 << 8)), sc};
            LZ_REC_STORE(w, reinterpret_cast<lz_v4*>(sb + lz_tcol(LZ_S16_MK + 4 * q)));
            lz_v4 cw = {cpre[0], cpre[1], cpre[2], 0.0f};
            LZ_REC_STORE(cw, reinterpret_cast<lz_v4*>(sb + lz_tcol(LZ_S16_CLR)));
            sigmas[row] = sigma;
            amb_aud[row] = norm;
            if (amb_eye) amb_eye[row] = eyeatt;
            unc_out[row] = lz_softplusf(upre);
            const int qc = q < 2 ? q : 2;
            const float cv = q == 0 ? cpre[0] : (q == 1 ? cpre[1] : cpre[2]);
            const _Float16 sg = (_Float16)h_sigmoid(cv);   // network.py:275 in half: sigmoid, * 1.002, - 0.001, each rounded to half
            const _Float16 t1 = h_round((float)sg * 1.002f);
            rgbs[row * 3 + qc] = (float)h_round((float)t1 - 0.001f);
        }
        slice = next;
    }
}

extern "C" int lz_triplane_head_forward_record_f16(const lz_head_params* p, const void* packed_unc, const float* xyzs, const float* dirs,
                                                   uint32_t M, float* sigmas, float* rgbs, float* amb_aud, float* amb_eye, float* unc,
                                                   void* rec16, float* state16, lz_stream_t stream) {
    LZ_REQUIRE(p && packed_unc && xyzs && dirs && sigmas && rgbs && amb_aud && unc && rec16 && state16, LZ_ERR_BAD_ARGUMENT,
               "triplane_head_forward_record_f16: null tensor");
    LZ_REQUIRE(p->emb_xy && p->emb_yz && p->emb_xz && p->offsets && p->packed && p->enc_a, LZ_ERR_BAD_ARGUMENT,
               "triplane_head_forward_record_f16: incomplete lz_head_params");
    LZ_REQUIRE(p->precision == 1 && !p->testing, LZ_ERR_UNSUPPORTED, "triplane_head_forward_record_f16: precision 1 (f16 packed weights), training mode");
    LZ_REQUIRE((((uintptr_t)rec16 | (uintptr_t)state16 | (uintptr_t)packed_unc | (uintptr_t)p->packed) & 15u) == 0, LZ_ERR_BAD_ARGUMENT,
               "triplane_head_forward_record_f16: rec / state / packed weights must be 16-byte aligned");
    if (M == 0) return LZ_OK;
    LzHead16Args a;
    a.emb[0] = p->emb_xy; a.emb[1] = p->emb_yz; a.emb[2] = p->emb_xz;
    a.offsets = p->offsets; a.packed = reinterpret_cast<const lz_h8*>(p->packed); a.enc_a = p->enc_a; a.ind_code = p->ind_code; a.eye = p->eye;
    a.bound = p->bound;
    for (int l = 0; l < 12; l++) {
        const float sc = exp2f((float)l * p->S) * (float)p->H - 1.0f;
        a.scale[l] = sc;
        a.res[l] = (uint32_t)ceilf(sc) + 1u;
    }
    const int n_cu = lz_cu_count();   // of the current device, per call (cached per device)
    const uint32_t want = lz_div_up(lz_div_up(M, 16), LZ_FREC16_WG / 64);
    const uint32_t grid = want < (uint32_t)n_cu ? want : (uint32_t)n_cu;
    hipLaunchKernelGGL(lz_k_triplane_head_forward_rec16, dim3(grid), dim3(LZ_FREC16_WG), 0, lz_st(stream), a,
                       reinterpret_cast<const lz_h8*>(packed_unc), xyzs, dirs, M, sigmas, rgbs, amb_aud, amb_eye, unc, static_cast<float*>(rec16), state16);
    LZ_CHECK_LAUNCH("triplane_head_forward_record_f16");
    return LZ_OK;
}
