// lz_head_rec.hip -- the fused triplane head for TRAINING without the recompute: a forward that records what the backward needs,
// and a backward that starts from that record (NeRFNetwork.forward, nerf_triplane/network.py:252-311, training mode).
//
// lz_head_bwd.hip recomputes the forward inside the backward kernel: 379 of its 759 MFMAs per 16-sample slice, on a kernel that is
// bound by the matrix pipe.  Here the forward kernel (the same instruction sequence as lz_k_triplane_head<true>, so the five outputs
// have the same bits) also writes
//   * the X half of the per-sample record (the wide layers' inputs, which the weight-gradient pass reads: the recomputing backward
//     wrote exactly these columns itself), and
//   * a state row per sample (LZ_FWD_STATE floats): att, the inputs of the three skinny output layers, the ReLU masks as bits and
//     seven scalars (||att||, eye_att, the pre-activations of unc / sigma / rgb),
// and the backward kernel reads the state row instead of xyz / dirs / the tables: no gather, no SH, no forward matrix work.  The
// arithmetic of every gradient is unchanged (same values, same order), only where the forward values come from.
#include <type_traits>

#include "lz_head_bwd_common.h"
#include "lz_head_fwd16_chain.h"   // RC: the f16 forward chain, recomputed in the backward
#include "lz_head_gather.h"
#include "lz_head_slice.h"
#include "lzzx_sh_eval.h"

typedef float lz_v4 __attribute__((ext_vector_type(4)));
typedef uint32_t lz_v2u __attribute__((ext_vector_type(2)));

#ifndef LZ_FREC_WG
#define LZ_FREC_WG 768   // forward: 143 VGPRs -> three waves per SIMD
#endif

__device__ __forceinline__ uint32_t lz_fbits(float v) { return __float_as_uint(v); }

// ---- f16 records (LZ_BWD_REC16 halves per sample; include/lzzx_nerf_hip.h LZ_R16_*): 16-column tiles interleaved in pairs, dword j of
// pair g = {tile 2 g column j, tile 2 g + 1 column j}.  Lane (s, q) holds columns 4 q + r of a tile in registers r = 0..3, so one
// dwordx4 store per lane covers its share of a pair (64 bytes per sample), values rounded to nearest even.
typedef _Float16 lz_h2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float lz_pack_h2(float lo, float hi) {
    asm volatile("" : "+v"(lo), "+v"(hi));   // the f32 values first, then their halves: no fused single rounding (v_fma_mixlo_f16)
    const lz_h2 v = {(_Float16)lo, (_Float16)hi};
    return __builtin_bit_cast(float, v);
}
// rowq = this sample's slot in the slice block (lz_blk) + 4 q; `pair` = tile / 2: a pair is one 16-dword tile of the block
__device__ __forceinline__ void lz_dump_pair(float* __restrict__ rowq, int pair, float l0, float l1, float l2, float l3, float h0, float h1,
                                             float h2, float h3) {
    lz_v4 w = {lz_pack_h2(l0, h0), lz_pack_h2(l1, h1), lz_pack_h2(l2, h2), lz_pack_h2(l3, h3)};
    LZ_REC_STORE(w, reinterpret_cast<lz_v4*>(rowq + 256 * pair));
}
// tiles t0, t0 + 1 of a chained-layout vector -> pair
template <int N>
__device__ __forceinline__ void lz_dump_pair_chained(float* __restrict__ rowq, int pair, const float (&v)[N], int t0) {
    lz_dump_pair(rowq, pair, v[4 * t0], v[4 * t0 + 1], v[4 * t0 + 2], v[4 * t0 + 3], v[4 * t0 + 4], v[4 * t0 + 5], v[4 * t0 + 6], v[4 * t0 + 7]);
}
__device__ __forceinline__ void lz_unpack_pair(const lz_v4& w, float* lo4, float* hi4) {
#pragma unroll
    for (int r = 0; r < 4; r++) {
        const uint32_t u = __float_as_uint(w[r]);
        lo4[r] = (float)__builtin_bit_cast(_Float16, (uint16_t)(u & 0xffffu));
        hi4[r] = (float)__builtin_bit_cast(_Float16, (uint16_t)(u >> 16));
    }
}

// ------------------------------------------------------------------------------------------------
// forward, recording
// ------------------------------------------------------------------------------------------------
template <bool H16>
__global__ void __launch_bounds__(LZ_FREC_WG, 1)
lz_k_triplane_head_forward_rec(LzHeadArgs P, const float* __restrict__ xyzs, const float* __restrict__ dirs, uint32_t M,
                               float* __restrict__ sigmas, float* __restrict__ rgbs, float* __restrict__ amb_aud,
                               float* __restrict__ amb_eye, float* __restrict__ unc_out, float* __restrict__ rec, float* __restrict__ st) {
    __shared__ float wl[LzHeadLds<true>::FLOATS];
    const uint32_t n_slices = (M + 15) / 16;
    const uint32_t slice_lo = (uint32_t)(((uint64_t)n_slices * blockIdx.x) / gridDim.x);
    const uint32_t slice_hi = (uint32_t)(((uint64_t)n_slices * (blockIdx.x + 1)) / gridDim.x);
    if (slice_lo >= slice_hi) return;
    const int lane = threadIdx.x & 63;
    const int s = lane & 15, q = lane >> 4;
    LzHeadCtx hc;
    lz_head_stage<true>(P, wl, LZ_FREC_WG, q, hc);
    __syncthreads();
    const float* wv = wl + LzHeadLds<true>::WV;
    int* queue = reinterpret_cast<int*>(wl + LzHeadLds<true>::TAB) + LZ_LVTAB_QUEUE;
    for (int w = threadIdx.x >> 8; w > 0; w--) {   // see lz_k_triplane_head_backward: the waves that share a SIMD start a part of a slice apart
        __builtin_amdgcn_s_sleep(127);
        __builtin_amdgcn_s_sleep(127);
    }
    // The position / direction of slice n + 1 are requested at the top of slice n, before its stores: a load issued behind stores can
    // only be waited for once those stores are acknowledged (one in-order counter), and the gather needs the position first.
    auto grab = [&]() -> int {
        int sl = 0;
        if (lane == 0) sl = atomicAdd(queue, 1);
        return __builtin_amdgcn_readfirstlane(sl);
    };
    auto row_of = [&](int sl) -> uint32_t {
        const uint32_t b = (slice_lo + (uint32_t)sl) * 16 + s;
        return b < M ? b : M - 1;
    };
    int slice = grab();
    float px = 0.0f, py = 0.0f, pz = 0.0f, dir0 = 0.0f, dir1 = 0.0f, dir2 = 0.0f;
    if (slice_lo + (uint32_t)slice < slice_hi) {
        const size_t r0 = row_of(slice);
        px = xyzs[r0 * 3]; py = xyzs[r0 * 3 + 1]; pz = xyzs[r0 * 3 + 2];
        dir0 = dirs[r0 * 3]; dir1 = dirs[r0 * 3 + 1]; dir2 = dirs[r0 * 3 + 2];
    }
    for (;;) {
        if (slice_lo + (uint32_t)slice >= slice_hi) break;
        const uint32_t base = (slice_lo + (uint32_t)slice) * 16;
        const bool valid = base + s < M;
        const uint32_t m = valid ? base + s : M - 1;   // clamped lanes repeat the last row: the same values are stored again
        const size_t row = m;
        float* rb = lz_blk(rec, slice_lo + (uint32_t)slice, H16 ? LZ_BWD_REC16 / 2 : LZ_BWD_REC, s);   // f16: rows counted in dwords
        float* sb = lz_blk(st, slice_lo + (uint32_t)slice, H16 ? LZ_FWD_STATE16 : LZ_FWD_STATE, s);

        float encx[9];
        lz_head_gather(hc.emb, hc.tab, px, py, pz, q, hc.bound, hc.two_bound, encx);
        const float cd0 = dir0, cd1 = dir1, cd2 = dir2;
        const int next = grab();
        if (slice_lo + (uint32_t)next < slice_hi) {
            const size_t r1 = row_of(next);
            px = xyzs[r1 * 3]; py = xyzs[r1 * 3 + 1]; pz = xyzs[r1 * 3 + 2];
            dir0 = dirs[r1 * 3]; dir1 = dirs[r1 * 3 + 1]; dir2 = dirs[r1 * 3 + 2];
        }
        __builtin_amdgcn_sched_barrier(0);
        const float bx[1][9] = {{encx[0], encx[1], encx[2], encx[3], encx[4], encx[5], encx[6], encx[7], encx[8]}};
        // audio channel attention
        float att[8];
        uint32_t mk_a1;
        {
            lz_f4 acc1[4][1] = {{lz_f4{0, 0, 0, 0}}, {lz_f4{0, 0, 0, 0}}, {lz_f4{0, 0, 0, 0}}, {lz_f4{0, 0, 0, 0}}};
            lz_layer<LZ_L_A1, 1>(wl, lane, bx, acc1);
            float a1[1][16];
#pragma unroll
            for (int ft = 0; ft < 4; ft++)
#pragma unroll
                for (int r = 0; r < 4; r++) a1[0][4 * ft + r] = lz_relu(acc1[ft][0][r]);
            mk_a1 = lz_mask_pos(a1[0]);
            {
                if constexpr (H16) {
                    lz_dump_pair_chained(rb + 4 * q, LZ_R16_X_A1 / 2, a1[0], 0);
                    lz_dump_pair_chained(rb + 4 * q, LZ_R16_X_A1 / 2 + 1, a1[0], 2);
                } else {
                    lz_dump_chained<4>(rb, q, LZ_BWD_X_A1, a1[0]);
                }
            }
            lz_f4 acc2[2][1] = {{lz_f4{0, 0, 0, 0}}, {lz_f4{0, 0, 0, 0}}};
            lz_layer<LZ_L_A2, 1>(wl, lane, a1, acc2);
#pragma unroll
            for (int ft = 0; ft < 2; ft++)
#pragma unroll
                for (int r = 0; r < 4; r++) att[4 * ft + r] = acc2[ft][0][r];
        }
        lz_dump_chained<2>(sb, q, LZ_ST_ATT, att);   // f32 in both layouts: the data gradient uses it
        float norm;
        {
            float acc = 0.0f;
#pragma unroll
            for (int k = 0; k < 8; k++) acc = lz_fmaf(att[k], att[k], acc);
            acc += __shfl_xor(acc, 16, 64);
            acc += __shfl_xor(acc, 32, 64);
            norm = sqrtf(acc);
        }
        // eye attention
        float eyeatt = 0.0f;
        uint32_t mk_e1 = 0;
        if (hc.has_eye) {
            lz_f4 acce[1][1] = {{lz_f4{0, 0, 0, 0}}};
            lz_layer<LZ_L_E1, 1>(wl, lane, bx, acce);
            float e1[4];
#pragma unroll
            for (int r = 0; r < 4; r++) e1[r] = lz_relu(acce[0][0][r]);
            mk_e1 = lz_mask_pos(e1);
            {
                if constexpr (H16) lz_dump_pair(sb + 4 * q, LZ_S16_E1 / 16, e1[0], e1[1], e1[2], e1[3], 0.0f, 0.0f, 0.0f, 0.0f);
                else lz_dump_chained<1>(sb, q, LZ_ST_E1, e1);
            }
            eyeatt = lz_sigmoidf(lz_lane_dot<1>(wv + LZ_WV_E2, q, e1));
        }
        // uncertainty
        float upre;
        uint32_t mk_u1;
        {
            lz_f4 accu[2][1] = {{lz_f4{0, 0, 0, 0}}, {lz_f4{0, 0, 0, 0}}};
            lz_layer<LZ_L_U1, 1>(wl, lane, bx, accu);
            float u1[8];
#pragma unroll
            for (int ft = 0; ft < 2; ft++)
#pragma unroll
                for (int r = 0; r < 4; r++) u1[4 * ft + r] = lz_relu(accu[ft][0][r]);
            mk_u1 = lz_mask_pos(u1);
            {
                if constexpr (H16) lz_dump_pair_chained(sb + 4 * q, LZ_S16_U1 / 16, u1, 0);
                else lz_dump_chained<2>(sb, q, LZ_ST_U1, u1);
            }
            upre = lz_lane_dot<2>(wv + LZ_WV_U2, q, u1);
        }
        // sigma net
        float spre;
        uint32_t mk_s1, mk_s2;
        float geo[1][16];
        {
            float b1[1][18];
#pragma unroll
            for (int i = 0; i < 9; i++) b1[0][i] = encx[i];
            float encw[8];
#pragma unroll
            for (int t = 0; t < 2; t++)
#pragma unroll
                for (int r = 0; r < 4; r++) encw[4 * t + r] = hc.lenca[16 * t + 4 * q + r] * att[4 * t + r];
#pragma unroll
            for (int k = 0; k < 8; k++) b1[0][9 + k] = encw[k];
            b1[0][17] = (hc.has_eye && q == 0) ? hc.eye_v * eyeatt : 0.0f;
            {   // sigma_net.0 input [enc_x 36 | enc_a * att 32 | eye * eye_att 1]
                if constexpr (H16) {
                    // tiles 0, 1: enc_x features 4 i + q, i < 8 (half 2 r + p of the lane's eight = i); tile 2: feature 32 + q at column
                    // 4 q, the eye term at column 1; tiles 3, 4: enc_a * att; tile 5: padding
                    lz_dump_pair(rb + 4 * q, LZ_R16_X_SIG0 / 2, encx[0], encx[2], encx[4], encx[6], encx[1], encx[3], encx[5], encx[7]);
                    lz_dump_pair(rb + 4 * q, LZ_R16_X_SIG0 / 2 + 1, encx[8], b1[0][17], 0.0f, 0.0f, encw[0], encw[1], encw[2], encw[3]);
                    lz_dump_pair(rb + 4 * q, LZ_R16_X_SIG0 / 2 + 2, encw[4], encw[5], encw[6], encw[7], 0.0f, 0.0f, 0.0f, 0.0f);
                } else {
#pragma unroll
                    for (int i = 0; i < 9; i++) rb[lz_tcol(LZ_BWD_X_SIG0 + 4 * i + q)] = encx[i];
                    lz_dump_chained<2>(rb, q, LZ_BWD_X_SIG0 + 36, encw);
                    rb[lz_tcol(LZ_BWD_X_SIG0 + 68 + q)] = b1[0][17];   // lanes q > 0 write zeros into the padding columns 69..71
                }
            }
            lz_f4 acc1[4][1] = {{lz_f4{0, 0, 0, 0}}, {lz_f4{0, 0, 0, 0}}, {lz_f4{0, 0, 0, 0}}, {lz_f4{0, 0, 0, 0}}};
            lz_layer<LZ_L_S1, 1>(wl, lane, b1, acc1);
            float s1[1][16];
#pragma unroll
            for (int ft = 0; ft < 4; ft++)
#pragma unroll
                for (int r = 0; r < 4; r++) s1[0][4 * ft + r] = lz_relu(acc1[ft][0][r]);
            mk_s1 = lz_mask_pos(s1[0]);
            {
                if constexpr (H16) {
                    lz_dump_pair_chained(rb + 4 * q, LZ_R16_X_S1 / 2, s1[0], 0);
                    lz_dump_pair_chained(rb + 4 * q, LZ_R16_X_S1 / 2 + 1, s1[0], 2);
                } else {
                    lz_dump_chained<4>(rb, q, LZ_BWD_X_S1, s1[0]);
                }
            }
            lz_f4 acc2[4][1] = {{lz_f4{0, 0, 0, 0}}, {lz_f4{0, 0, 0, 0}}, {lz_f4{0, 0, 0, 0}}, {lz_f4{0, 0, 0, 0}}};
            lz_layer<LZ_L_S2, 1>(wl, lane, s1, acc2);
            float s2[1][16];
#pragma unroll
            for (int ft = 0; ft < 4; ft++)
#pragma unroll
                for (int r = 0; r < 4; r++) s2[0][4 * ft + r] = lz_relu(acc2[ft][0][r]);
            mk_s2 = lz_mask_pos(s2[0]);
            {
                if constexpr (H16) {
                    lz_dump_pair_chained(rb + 4 * q, LZ_R16_X_S2C / 2, s2[0], 0);
                    lz_dump_pair_chained(rb + 4 * q, LZ_R16_X_S2C / 2 + 1, s2[0], 2);
                } else {
                    lz_dump_chained<4>(rb, q, LZ_BWD_X_S2C, s2[0]);
                }
            }
            lz_f4 acc3[4][1] = {{lz_f4{0, 0, 0, 0}}, {lz_f4{0, 0, 0, 0}}, {lz_f4{0, 0, 0, 0}}, {lz_f4{0, 0, 0, 0}}};
            lz_layer<LZ_L_S3, 1>(wl, lane, s2, acc3);
#pragma unroll
            for (int ft = 0; ft < 4; ft++)
#pragma unroll
                for (int r = 0; r < 4; r++) geo[0][4 * ft + r] = acc3[ft][0][r];
            spre = lz_lane_dot<4>(wv + LZ_WV_SIG, q, s2[0]);
        }
        // colour net
        float cpre[3];
        uint32_t mk_c1;
        {
            float o[16];
            lz_sh_eval(cd0, cd1, cd2, 4, o, nullptr, nullptr, nullptr);
            float b1[1][21];
#pragma unroll
            for (int i = 0; i < 4; i++) b1[0][i] = q == 0 ? o[4 * i] : (q == 1 ? o[4 * i + 1] : (q == 2 ? o[4 * i + 2] : o[4 * i + 3]));
#pragma unroll
            for (int k = 0; k < 16; k++) b1[0][4 + k] = geo[0][k];
            b1[0][20] = hc.indq;
            {   // colour_net.0 input [SH 16 | geo 64 | ind 4]; geo = s2 . Wg^T is not stored (lz_head_bwd.hip)
                if constexpr (H16) {   // tile 4: SH component 4 r + q at column 4 q + r; tile 5: ind_code[q] at column 4 q
                    lz_dump_pair(rb + 4 * q, LZ_R16_X_S2C / 2 + 2, b1[0][0], b1[0][1], b1[0][2], b1[0][3], hc.indq, 0.0f, 0.0f, 0.0f);
                } else {
#pragma unroll
                    for (int i = 0; i < 4; i++) rb[lz_tcol(LZ_BWD_X_S2C + 64 + 4 * i + q)] = b1[0][i];
                    rb[lz_tcol(LZ_BWD_X_S2C + 80 + q)] = hc.indq;
                }
            }
            lz_f4 acc1[4][1] = {{lz_f4{0, 0, 0, 0}}, {lz_f4{0, 0, 0, 0}}, {lz_f4{0, 0, 0, 0}}, {lz_f4{0, 0, 0, 0}}};
            lz_layer<LZ_L_C1, 1>(wl, lane, b1, acc1);
            float c1[16];
#pragma unroll
            for (int ft = 0; ft < 4; ft++)
#pragma unroll
                for (int r = 0; r < 4; r++) c1[4 * ft + r] = lz_relu(acc1[ft][0][r]);
            mk_c1 = lz_mask_pos(c1);
            {
                if constexpr (H16) {
                    lz_dump_pair_chained(sb + 4 * q, LZ_S16_C1 / 16, c1, 0);
                    lz_dump_pair_chained(sb + 4 * q, LZ_S16_C1 / 16 + 1, c1, 2);
                } else {
                    lz_dump_chained<4>(sb, q, LZ_ST_C1, c1);
                }
            }
#pragma unroll
            for (int c = 0; c < 3; c++) cpre[c] = lz_lane_dot<4>(wv + LZ_WV_C2 + 64 * c, q, c1);
        }
        const float sigma = lz_expf(spre);
        {
            // masks + one scalar per lane: q = 0 ||att||, 1 eye_att, 2 unc pre-activation, 3 sigma
            const float sc = q == 0 ? norm : (q == 1 ? eyeatt : (q == 2 ? upre : sigma));
            lz_v4 w = {__uint_as_float(mk_a1 | (mk_s1 << 16)), __uint_as_float(mk_s2 | (mk_c1 << 16)), __uint_as_float(mk_u1 | (mk_e1 << 8)), sc};
            LZ_REC_STORE(w, reinterpret_cast<lz_v4*>(sb + lz_tcol((H16 ? LZ_S16_MK : LZ_ST_MK) + 4 * q)));
            // the four lanes of a sample hold the same bits: all of them store (same address, same value), no lane-dependent branch
            lz_v4 cw = {cpre[0], cpre[1], cpre[2], 0.0f};
            LZ_REC_STORE(cw, reinterpret_cast<lz_v4*>(sb + lz_tcol(H16 ? LZ_S16_CLR : LZ_ST_CLR)));
            sigmas[m] = sigma;
            amb_aud[m] = norm;
            if (amb_eye) amb_eye[m] = eyeatt;
            unc_out[m] = lz_softplusf(upre);
            const int qc = q < 2 ? q : 2;
            const float cv = q == 0 ? cpre[0] : (q == 1 ? cpre[1] : cpre[2]);
            rgbs[(size_t)m * 3 + qc] = lz_sigmoidf(cv) * 1.002f - 0.001f;
        }
        slice = next;
    }
}

// ------------------------------------------------------------------------------------------------
// backward from the recorded state
// ------------------------------------------------------------------------------------------------
// B16: the matrix work on the f16 cores from the transposed half fragments (99 x 512 B instead of the 379 f32 forward fragments)
//
// FUSE (with H16 and B16: the whole step in the reference's autocast arithmetic): the weight gradients of the wide layers are reduced
// INSIDE this kernel, so the G half of the record is never written and the separate pass over the records (lz_k_head_grad_w16: 8.4 GB
// read per cfg3 step) disappears.  dW = sum over samples of G^T X needs the SAMPLE index as the contraction index of the matrix core,
// while the chain holds G with the sample on the lane: a transpose, done through LDS with gfx950's transposing read
// (ds_read_b64_tr_b16).  95 accumulator tiles do not fit one wave, so the 8 waves of the workgroup share them: the waves advance in
// lockstep ROUNDS of 8 slices (one per wave); after each of five chain segments every wave drops the G tiles it just produced and the
// matching X tiles of its slice (loaded from the X half of the record the forward wrote) into its AREA of an LDS buffer, a workgroup
// barrier follows, and every wave multiplies the tiles it OWNS of that product over all 8 areas.  Two buffers alternate, so one
// barrier per segment suffices (a wave that writes buffer b at segment g has passed barrier g - 1, which every wave reaches only after
// its reads of segment g - 2).  Products and owners (G row tiles t x X column tiles u; w = wave):
//     after dc1, dh0   c1h   5 x 6 = 30   w < 6: column u = w, rows 0..4
//     after ds2        sig1  4 x 4 = 16   all:   column u = w & 3, rows 2 (w >> 2), + 1
//     after ds1        sig0  4 x 5 = 20   w in {6, 7, 0, 1, 2}: column u = (w + 2) & 7, rows 0..3
//     after datt       aud1  2 x 4 =  8   all:   row t = w >> 2, column u = w & 3
//     after da1        x3    7 x 3 = 21   w >= 1: row t = w - 1, columns 0..2
// -- at most 15 tiles = 60 accumulator registers per wave.  At the end a workgroup writes its partial tiles in the layout of
// lz_k_head_grad_w16 and the same lz_k_head_grad_w_reduce sums them over the workgroups.
// With the f32 data-gradient chain (B16 = false: the 97 KB f32 weight image stays in LDS) only ONE buffer fits: every segment then
// ends with a second barrier (all reads done) before the next one may write.
#define LZ_FUSE_AREA_TILES 11                                 // largest segment: c1h, 5 G + 6 X tiles
#define LZ_FUSE_LDS_FLOATS(NBUF) ((NBUF) * 8 * LZ_FUSE_AREA_TILES * 128)   // buffers x 8 areas x 11 tiles x 512 bytes = 44 KB each
// RC (round 5; with H16, B16, FUSE): NOTHING is read back from the forward but the enc_x operand (80 bytes per sample, `st` = encx16 of
// lz_triplane_head_forward_encx_f16) and the view directions: the wave recomputes the f16 forward chain of its slice (lz_fwd16_chain, the very
// function the forward ran: 64 MFMAs) with a sink that keeps the layer inputs, the state pairs and the masks in registers, and goes on as
// FUSE does.  The f16 forward fragments (59 + 5 KB) take the place of the second G / X buffer in LDS, so every segment pays the second
// barrier of the single-buffer arrangement.  Per sample and step: 80 B written by the forward and 80 + 40 B read here, instead of
// 1 216 B of record + state written and read.
template <bool H16, bool B16, bool FUSE = false, bool RC = false>
__global__ void __launch_bounds__(LZ_BWD_WG, LZ_BWD_WG / 256)
lz_k_triplane_head_backward_rec(LzHeadBwdArgs A, const float* __restrict__ st, uint32_t M, float* __restrict__ parts) {
    static_assert(!FUSE || H16, "the fused weight-gradient products run on half operands (f16 records)");
    static_assert(!RC || (H16 && B16 && FUSE), "the recomputing arrangement is the all-f16 one with fused weight gradients");
    constexpr int NFRAG = LZ_FRAGS_ALL;
    constexpr int WV = B16 ? LZ_BFRAGS * 128 : NFRAG * 64, TAB = WV + LZ_WV_FLOATS;
    constexpr uint32_t NBUF = (B16 && !RC) ? 2u : 1u;          // FUSE: LDS buffers of G / X tiles
    constexpr int FUSE_FLOATS = FUSE ? LZ_FUSE_LDS_FLOATS(NBUF) : 0;
    constexpr int FW16 = TAB + LZ_LVTAB_WORDS + FUSE_FLOATS;                        // RC: f16 forward fragments, then unc_net's five
    constexpr int FW16_FLOATS = RC ? (H_FRAGS + LZ_UNC16_FRAGS) * 64 * 4 : 0;
    static_assert((FW16 + FW16_FLOATS) * 4 <= 163840, "LDS budget of one workgroup per CU");
    __shared__ __align__(16) float wl[FW16 + FW16_FLOATS];
    const LzHeadArgs& P = A.fwd;
    const lz_head_bwd_out& O = A.o;
    const uint32_t n_slices = (M + 15) / 16;
    const uint32_t slice_lo = (uint32_t)(((uint64_t)n_slices * blockIdx.x) / gridDim.x);
    const uint32_t slice_hi = (uint32_t)(((uint64_t)n_slices * (blockIdx.x + 1)) / gridDim.x);
    if (slice_lo >= slice_hi) return;
    {
        const float4* src = reinterpret_cast<const float4*>(B16 ? A.wb16 : P.packed);
        float4* dst = reinterpret_cast<float4*>(wl);
        for (int i = threadIdx.x; i < WV / 4; i += LZ_BWD_WG) dst[i] = src[i];
        if (threadIdx.x < LZ_WV_FLOATS) wl[WV + threadIdx.x] = P.packed[LZ_FRAGS_ALL * 64 + threadIdx.x];
        if (threadIdx.x < 32) wl[TAB + LZ_LVTAB_ENCA + threadIdx.x] = P.enc_a[threadIdx.x];
        if (threadIdx.x == 0) reinterpret_cast<int*>(wl + TAB)[LZ_LVTAB_QUEUE] = 0;
        if constexpr (RC) {   // the forward's weight images and the half tables its chain reads (lz_head16_stage)
            const float4* f16 = reinterpret_cast<const float4*>(A.fw16);
            const float4* u16 = reinterpret_cast<const float4*>(A.unc16);
            float4* d16 = reinterpret_cast<float4*>(wl + FW16);
            for (int i = threadIdx.x; i < H_FRAGS * 64; i += LZ_BWD_WG) d16[i] = f16[i];
            for (int i = threadIdx.x; i < LZ_UNC16_FRAGS * 64; i += LZ_BWD_WG) d16[H_FRAGS * 64 + i] = u16[i];
            int* tab = reinterpret_cast<int*>(wl + TAB);
            if (threadIdx.x < 16) tab[LZ_LVTAB_ENCA16 + threadIdx.x] = (int)h_cvt2(P.enc_a[2 * threadIdx.x], P.enc_a[2 * threadIdx.x + 1], false);
            if (threadIdx.x < 2) tab[LZ_LVTAB_IND16 + threadIdx.x] = P.ind_code ? (int)h_cvt2(P.ind_code[2 * threadIdx.x], P.ind_code[2 * threadIdx.x + 1], false) : 0;
        }
    }
    __syncthreads();
    // one spelling for both matrix paths
    auto layer_bwd = [&](auto layer_tag, const auto& dy, auto& dx) {
        constexpr int LAYER = decltype(layer_tag)::value;
        if constexpr (B16) lz_layer_bwd16<LAYER>(reinterpret_cast<const uint2*>(wl), (int)(threadIdx.x & 63), dy, dx);
        else lz_layer_bwd<LAYER>(wl, (int)(threadIdx.x & 63), dy, dx);
    };
    const float* lenca = wl + TAB + LZ_LVTAB_ENCA;
    const float* wv = wl + WV;
    const int lane = threadIdx.x & 63;
    const int s = lane & 15, q = lane >> 4;
    const bool has_eye = P.eye != nullptr;
    const float eye_v = has_eye ? P.eye[0] : 0.0f;
    int* queue = reinterpret_cast<int*>(wl + TAB) + LZ_LVTAB_QUEUE;
    if ((threadIdx.x >> 6) >= 4) {
        __builtin_amdgcn_s_sleep(127);
        __builtin_amdgcn_s_sleep(127);
    }
    float acc_enca[8], acc_ind = 0.0f;   // d(enc_a)[16 t + 4 q + r], d(ind_code)[q], summed over this lane's samples
#pragma unroll
    for (int k = 0; k < 8; k++) acc_enca[k] = 0.0f;
    float acc_e2[4], acc_u2[8], acc_c2[3][16];   // weight gradients of the skinny output layers (lz_head_bwd.hip)
#pragma unroll
    for (int k = 0; k < 4; k++) acc_e2[k] = 0.0f;
#pragma unroll
    for (int k = 0; k < 8; k++) acc_u2[k] = 0.0f;
#pragma unroll
    for (int c = 0; c < 3; c++)
#pragma unroll
        for (int k = 0; k < 16; k++) acc_c2[c][k] = 0.0f;

    // Everything a slice reads (its state row and upstream gradients) is requested one slice ahead, at the top of the previous slice and
    // before that slice's stores: one in-order counter covers loads and stores, so a load issued behind the record stores could only be
    // waited for once those were acknowledged.  The last prefetch of a workgroup reads a clamped row and is dropped.
    struct In {
        lz_v4 att0, att1, c0, c1, c2, c3, u0, u1, e, mk, clr;
        float g_sig, g_aa, g_ae, g_un, g_r0, g_r1, g_r2;
    };
    auto ld4 = [](const float* p) -> lz_v4 { return __builtin_nontemporal_load(reinterpret_cast<const lz_v4*>(p)); };
    auto grab = [&]() -> int {
        int sl = 0;
        if (lane == 0) sl = atomicAdd(queue, 1);
        return __builtin_amdgcn_readfirstlane(sl);
    };
    auto fetch = [&](int sl) -> In {
        uint32_t gs = slice_lo + (uint32_t)sl;
        if (gs >= slice_hi) gs = slice_hi - 1;
        const uint32_t b = gs * 16 + s;
        const size_t row = b < M ? b : M - 1;
        const float* sb = lz_blk(st, gs, H16 ? LZ_FWD_STATE16 : LZ_FWD_STATE, s);   // lanes past the end read their (unwritten-or-stale) padding slot: never accumulated
        In in;
        const lz_v4 z = {0.0f, 0.0f, 0.0f, 0.0f};
        in.att0 = ld4(sb + lz_tcol(LZ_ST_ATT + 4 * q)); in.att1 = ld4(sb + lz_tcol(LZ_ST_ATT + 16 + 4 * q));
        in.c2 = z; in.c3 = z; in.u1 = z; in.e = z;
        if constexpr (H16) {
            in.c0 = ld4(sb + lz_tcol(LZ_S16_C1 + 4 * q)); in.c1 = ld4(sb + lz_tcol(LZ_S16_C1 + 16 + 4 * q));
            in.u0 = ld4(sb + lz_tcol(LZ_S16_U1 + 4 * q));
            if (has_eye) in.e = ld4(sb + lz_tcol(LZ_S16_E1 + 4 * q));
        } else {
            in.c0 = ld4(sb + lz_tcol(LZ_ST_C1 + 4 * q)); in.c1 = ld4(sb + lz_tcol(LZ_ST_C1 + 16 + 4 * q)); in.c2 = ld4(sb + lz_tcol(LZ_ST_C1 + 32 + 4 * q)); in.c3 = ld4(sb + lz_tcol(LZ_ST_C1 + 48 + 4 * q));
            in.u0 = ld4(sb + lz_tcol(LZ_ST_U1 + 4 * q)); in.u1 = ld4(sb + lz_tcol(LZ_ST_U1 + 16 + 4 * q));
            if (has_eye) in.e = ld4(sb + lz_tcol(LZ_ST_E1 + 4 * q));
        }
        in.mk = ld4(sb + lz_tcol((H16 ? LZ_S16_MK : LZ_ST_MK) + 4 * q));
        in.clr = ld4(sb + lz_tcol(H16 ? LZ_S16_CLR : LZ_ST_CLR));
        in.g_sig = A.g_sigma[row]; in.g_aa = A.g_amb_aud[row]; in.g_ae = A.g_amb_eye ? A.g_amb_eye[row] : 0.0f; in.g_un = A.g_unc[row];
        in.g_r0 = A.g_rgb[row * 3]; in.g_r1 = A.g_rgb[row * 3 + 1]; in.g_r2 = A.g_rgb[row * 3 + 2];
        return in;
    };
    // RC: what a slice needs from memory -- its enc_x operand (five dwords per lane), the view direction and the upstream gradients
    struct InRc {
        lz_v4 bx0;
        float bx1, d0, d1, d2;
        float g_sig, g_aa, g_ae, g_un, g_r0, g_r1, g_r2;
    };
    auto fetch_rc = [&](int sl) -> InRc {
        uint32_t gs = slice_lo + (uint32_t)sl;
        if (gs >= slice_hi) gs = slice_hi - 1;
        const uint32_t b = gs * 16 + s;
        const size_t row = b < M ? b : M - 1;
        const float* eb = st + (size_t)gs * (5 * 64) + lane;      // encx16 [slice][5][lane]
        InRc in;
        in.bx0 = lz_v4{eb[0], eb[64], eb[128], eb[192]};
        in.bx1 = eb[256];
        in.d0 = A.dirs[row * 3]; in.d1 = A.dirs[row * 3 + 1]; in.d2 = A.dirs[row * 3 + 2];
        in.g_sig = A.g_sigma[row]; in.g_aa = A.g_amb_aud[row]; in.g_ae = A.g_amb_eye ? A.g_amb_eye[row] : 0.0f; in.g_un = A.g_unc[row];
        in.g_r0 = A.g_rgb[row * 3]; in.g_r1 = A.g_rgb[row * 3 + 1]; in.g_r2 = A.g_rgb[row * 3 + 2];
        return in;
    };
    // ... and the sink of the recomputed chain: the words the recording forward stored, kept in registers (constant indices after inlining)
    // Registers are the budget (256 at two waves per SIMD, 60 of them weight-gradient accumulators): the sink keeps only what cannot be had
    // again cheaply -- the inputs of sigma_net.1 / .2 and color_net.0's SH columns (X_S1, X_S2C), color_net.1's / unc_net.1's / eye_att_net.1's
    // inputs and att as packed halves.  sigma_net.0's input is rebuilt from the enc_x operand, att and the eye term where its segment needs it
    // (a permutation of halves, see rc_sig0_pair), and aud_ch_att_net.1's input by running aud_ch_att_net.0 once more (8 MFMAs).
    struct RegSink {
        lz_v4 xp[10];    // X pairs; kept: LZ_R16_X_S1 / 2 .. LZ_R16_X_S2C / 2 + 2 (5 .. 9)
        lz_v4 sp[6];     // state pairs LZ_S16_C1 / 16 (2, 3), LZ_S16_U1 / 16 (4), LZ_S16_E1 / 16 (5)
        lz_v2u att16[2]; // att as eight halves (exact: the values are halves)
        __device__ __forceinline__ void x_pair_h8(int pair, const lz_h8& b) { if (pair >= LZ_R16_X_S1 / 2) xp[pair] = lz_pair_words_h8(b); }
        __device__ __forceinline__ void x_pair_f(int pair, float l0, float l1, float l2, float l3, float h0, float h1, float h2, float h3) {
            if (pair >= LZ_R16_X_S1 / 2) xp[pair] = lz_pair_words_f(l0, l1, l2, l3, h0, h1, h2, h3);
        }
        __device__ __forceinline__ void s_pair_h8(int pair, const lz_h8& b) { sp[pair] = lz_pair_words_h8(b); }
        __device__ __forceinline__ void s_att(const lz_v4& w0, const lz_v4& w1) {
            att16[0] = lz_v2u{h_cvt2(w0[0], w0[1], false), h_cvt2(w0[2], w0[3], false)};
            att16[1] = lz_v2u{h_cvt2(w1[0], w1[1], false), h_cvt2(w1[2], w1[3], false)};
        }
    };
    LzHead16Ctx hc16;
    if constexpr (RC) {
        hc16.wl = reinterpret_cast<const lz_h8*>(wl + FW16);
        hc16.tab = reinterpret_cast<const int*>(wl + TAB);
        hc16.lenca = wl + TAB + LZ_LVTAB_ENCA;
        hc16.emb[0] = hc16.emb[1] = hc16.emb[2] = nullptr;
        hc16.ind_code = P.ind_code;
        hc16.bound = P.bound; hc16.two_bound = 2.0f * P.bound;
        hc16.has_eye = has_eye; hc16.eye_v = eye_v; hc16.unc_const = 0.0f;
    }
    const float indq = (RC && P.ind_code) ? P.ind_code[q] : 0.0f;
    // ---- FUSE: accumulators of the tiles this wave owns, the two LDS buffers, the helpers of the five segments --------------------
    const int wave = threadIdx.x >> 6;
    const uint32_t n_local = slice_hi - slice_lo;
    float* fuse = wl + TAB + LZ_LVTAB_WORDS;
    constexpr int AREA = LZ_FUSE_AREA_TILES * 128;          // floats per area
    uint32_t seg = 0;                                       // running segment number: buffer = seg & 1
    lz_f4 acc_c1h[5], acc_sig1[2], acc_sig0[4], acc_aud1[1], acc_x3[3];
#pragma unroll
    for (int t = 0; t < 5; t++) acc_c1h[t] = lz_f4{0, 0, 0, 0};
#pragma unroll
    for (int t = 0; t < 4; t++) acc_sig0[t] = lz_f4{0, 0, 0, 0};
#pragma unroll
    for (int t = 0; t < 3; t++) acc_x3[t] = lz_f4{0, 0, 0, 0};
    acc_sig1[0] = acc_sig1[1] = acc_aud1[0] = lz_f4{0, 0, 0, 0};
    lz_f4 acc_c2t = lz_f4{0, 0, 0, 0};   // FUSE, waves 4 .. 7: color_net.1's weight gradient, columns 16 (wave & 3) .. + 15 (rows 0 .. 2 of the tile)
    // a 16 x 16 half tile in LDS: [row][16 halves], row = rho(sample) so that the four rows one lane group of the transposing read
    // takes are samples kg, 4 + kg, 8 + kg, 12 + kg -- the k-slot order both operands of lz_k_head_grad_w16 use
    const int rho = 4 * (s & 3) + (s >> 2);
    // ... and inside a row the 8-byte column block q sits at q ^ (rho / 4) (= q ^ (s & 3)): a tile row is 8 dwords, so rows r, r + 4, r + 8, r + 12
    // start at the same bank and the 16 lanes of a ds_write_b64 group (q fixed, 16 rows) collided four ways -- 589 LDS conflict cycles per
    // slice (round 5, SQ_LDS_BANK_CONFLICT: 31 % of the LDS-active cycles of the backward).  With the swizzle the four rows of a bank group
    // write four different blocks; the transposing read takes 4 consecutive rows per lane group, conflict-free under any block order
    const int qsw = q ^ (s & 3);
    int slice = FUSE ? wave : grab();       // FUSE: static rounds of 8 slices, slice = 8 round + wave
    In nx;
    InRc nxr;
    if constexpr (RC) nxr = fetch_rc(slice);
    else nx = fetch(slice);
    for (;;) {
        if constexpr (FUSE) {
            if ((uint32_t)(slice - wave) >= n_local) break;                     // workgroup-uniform: the round's first slice
        } else {
            if (slice_lo + (uint32_t)slice >= slice_hi) break;
        }
        // FUSE: a wave without a slice in the last round repeats the workgroup's last slice with every lane invalid -- zeros in its area,
        // nothing accumulated, the same denc values stored again -- so that all waves pass the same barriers
        const bool have = !FUSE || (uint32_t)slice < n_local;
        const uint32_t gslice = slice_lo + (have ? (uint32_t)slice : n_local - 1);
        const uint32_t base = gslice * 16;
        const bool valid = have && base + s < M;
        const uint32_t m = base + s < M ? base + s : M - 1;   // clamped lanes repeat the last row: same values stored again, nothing accumulated
        const size_t row = m;
        float* rb = lz_blk(O.rec, gslice, H16 ? LZ_BWD_REC16 / 2 : LZ_BWD_REC, s);   // f16: rows counted in dwords
        float* dencq = O.denc + (size_t)q * M + row;
        In in;
        RegSink rs;
        lz_v4 rc_bx0 = {0.0f, 0.0f, 0.0f, 0.0f};
        float rc_bx1 = 0.0f, rc_eyeatt = 0.0f;
        if constexpr (RC) {
            // the forward of this slice again, from its enc_x operand: the layer inputs / state pairs land in `rs`, the scalars and masks in `fo`
            const InRc ir = nxr;
            lz_h8 bx[2];
            bx[0] = __builtin_bit_cast(lz_h8, ir.bx0);
            bx[1] = __builtin_bit_cast(lz_h8, lz_v4{ir.bx1, 0.0f, 0.0f, 0.0f});
            LzFwd16Out fo;
            lz_fwd16_chain(hc16, hc16.wl + H_FRAGS * 64, lane, bx, ir.d0, ir.d1, ir.d2, indq, rs, fo);
            const lz_v4 z = {0.0f, 0.0f, 0.0f, 0.0f};
            in.att0 = z; in.att1 = z;      // RC: att comes from rs.att16 where it is used
            rc_bx0 = ir.bx0; rc_bx1 = ir.bx1; rc_eyeatt = fo.eyeatt;
            in.c0 = rs.sp[LZ_S16_C1 / 16]; in.c1 = rs.sp[LZ_S16_C1 / 16 + 1]; in.c2 = z; in.c3 = z;
            in.u0 = rs.sp[LZ_S16_U1 / 16]; in.u1 = z;
            in.e = has_eye ? rs.sp[LZ_S16_E1 / 16] : z;
            const float sc = q == 0 ? fo.norm : (q == 1 ? fo.eyeatt : (q == 2 ? fo.upre : fo.sigma));   // (the state row's arrangement)
            in.mk = lz_v4{__uint_as_float(fo.mk_a1 | (fo.mk_s1 << 16)), __uint_as_float(fo.mk_s2 | (fo.mk_c1 << 16)), __uint_as_float(fo.mk_u1 | (fo.mk_e1 << 8)), sc};
            in.clr = lz_v4{fo.cpre[0], fo.cpre[1], fo.cpre[2], 0.0f};
            in.g_sig = ir.g_sig; in.g_aa = ir.g_aa; in.g_ae = ir.g_ae; in.g_un = ir.g_un; in.g_r0 = ir.g_r0; in.g_r1 = ir.g_r1; in.g_r2 = ir.g_r2;
        } else {
            in = nx;
        }
        const int next = FUSE ? slice + 8 : grab();
        if constexpr (!FUSE) nx = fetch(next);   // FUSE fetches later in the slice (after the sig0 segment): 51 registers less across the chain
        // FUSE helpers (all lanes take part in every LDS access: the transposing read needs EXEC all ones)
        float* const my_area0 = fuse + wave * AREA;
        auto g_put = [&](int tile, float v0, float v1, float v2, float v3) {     // this lane's four columns 4 q .. 4 q + 3 of a G tile
            float* ar = my_area0 + (seg & (NBUF - 1u)) * (8 * AREA);
            lz_v2u w = {__float_as_uint(lz_pack_h2(v0, v1)), __float_as_uint(lz_pack_h2(v2, v3))};
            if (!valid) w = lz_v2u{0u, 0u};
            *reinterpret_cast<lz_v2u*>(ar + tile * 128 + rho * 8 + 2 * qsw) = w;
        };
        auto x_load = [&](int pair) -> lz_v4 {     // X half of the record: dword j = {tile 2 p, tile 2 p + 1} column 4 q + j (RC: the recomputed words)
            if constexpr (RC) return rs.xp[pair];
            else return ld4(rb + 256 * pair + 4 * q);
        };
        // RC: pair i of sigma_net.0's input as the forward's sink would have written it (lz_fwd16_chain: x_pair_f(LZ_R16_X_SIG0 / 2 + i, ...)):
        //   pair 0 = {enc_x 2 r | enc_x 2 r + 1} = the enc_x operand's first four dwords as they are
        //   pair 1 = {enc_x 8 | encw 0}, {eye term | encw 1}, {0 | encw 2}, {0 | encw 3};   pair 2 = {encw 4 + r | 0}   (encw = enc_a * att, halves)
        auto rc_sig0_pair = [&](int i) -> lz_v4 {
            if (i == 0) return rc_bx0;
            const lz_u4 a16 = {rs.att16[0][0], rs.att16[0][1], rs.att16[1][0], rs.att16[1][1]};
            const lz_u4 ew = __builtin_bit_cast(lz_u4, h_encw(hc16.tab, q, __builtin_bit_cast(lz_h8, a16)));
            if (i == 2) return __builtin_bit_cast(lz_v4, lz_u4{ew[2] & 0xffffu, ew[2] >> 16, ew[3] & 0xffffu, ew[3] >> 16});
            const _Float16 et = (has_eye && q == 0) ? h_round(eye_v * rc_eyeatt) : (_Float16)0.0f;
            const uint32_t etw = (uint32_t)__builtin_bit_cast(uint16_t, et);
            return __builtin_bit_cast(lz_v4, lz_u4{(__float_as_uint(rc_bx1) & 0xffffu) | (ew[0] << 16), etw | (ew[0] & 0xffff0000u), ew[1] << 16, ew[1] & 0xffff0000u});
        };
        // RC: aud_ch_att_net.1's input (two pairs) by running aud_ch_att_net.0 on the enc_x operand again
        auto rc_a1_pairs = [&](lz_v4& p0, lz_v4& p1) {
            lz_h8 bx[2];
            bx[0] = __builtin_bit_cast(lz_h8, rc_bx0);
            bx[1] = __builtin_bit_cast(lz_h8, lz_v4{rc_bx1, 0.0f, 0.0f, 0.0f});
            lz_f4 a1[4] = {lz_f4{0, 0, 0, 0}, lz_f4{0, 0, 0, 0}, lz_f4{0, 0, 0, 0}, lz_f4{0, 0, 0, 0}};
            h_layer<H_A1>(hc16.wl, lane, bx, a1);
            p0 = lz_pair_words_h8(h_pair(a1[0], a1[1], true));
            p1 = lz_pair_words_h8(h_pair(a1[2], a1[3], true));
        };
        auto x_put = [&](int tile, const lz_v4& d, bool odd) {                               // one tile of the pair, de-interleaved
            float* ar = my_area0 + (seg & (NBUF - 1u)) * (8 * AREA);
            const uint32_t sel = odd ? 0x07060302u : 0x05040100u;
            lz_v2u w = {__builtin_amdgcn_perm(__float_as_uint(d[1]), __float_as_uint(d[0]), sel),
                        __builtin_amdgcn_perm(__float_as_uint(d[3]), __float_as_uint(d[2]), sel)};
            if (!valid) w = lz_v2u{0u, 0u};                                                  // padding rows of the record are never written: not even NaN may pass
            *reinterpret_cast<lz_v2u*>(ar + tile * 128 + rho * 8 + 2 * qsw) = w;
        };
        auto tr = [&](const float* ar, int tile) -> lz_bh4 {     // lane (i, kg): column i, rows 4 kg .. 4 kg + 3 of the tile image
            const int l16 = lane & 15;
            const _Float16* ph = reinterpret_cast<const _Float16*>(ar + tile * 128) + (4 * (lane >> 4) + (l16 >> 2)) * 16 + 4 * ((l16 & 3) ^ (lane >> 4));   // (column block swizzled by row / 4: see qsw)
            typedef short lz_s4 __attribute__((ext_vector_type(4)));
            return __builtin_bit_cast(lz_bh4, __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) lz_s4*)ph));
        };
        // one segment: barrier (every area of this buffer is written), then this wave's tiles of the product over the 8 areas
        // NG = G tiles of the segment (the X tiles follow them in the area)
        // v_mfma_f32_16x16x32_f16: the contraction runs over the 32 samples of TWO areas (k slots 0..3 of a lane from area a, 4..7 from
        // area a + 1: any assignment of samples to k slots works as long as both operands use the same one)
        typedef _Float16 lz_bh8 __attribute__((ext_vector_type(8)));
        auto tr2 = [&](const float* ar, int tile) -> lz_bh8 {
            const lz_bh4 lo = tr(ar, tile), hi = tr(ar + AREA, tile);
            return lz_bh8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        };
        auto mma32 = [](const lz_bh8& g, const lz_bh8& x, lz_f4& acc) { acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(g, x, acc, 0, 0, 0); };
        auto seg_c1h = [&]() {
            __syncthreads();
            if (wave < 6) {
                const float* buf = fuse + (seg & (NBUF - 1u)) * (8 * AREA);
#pragma unroll 2
                for (int a = 0; a < 8; a += 2) {
                    const float* ar = buf + a * AREA;
                    const lz_bh8 x = tr2(ar, 5 + wave);
#pragma unroll
                    for (int t = 0; t < 5; t++) mma32(tr2(ar, t), x, acc_c1h[t]);
                }
            }
            seg++;
            if constexpr (NBUF == 1) __syncthreads();   // single buffer: every wave has read before anyone writes the next segment (in FRONT of the next segment's first write instead: the -O step 5.30 -> 5.36 ms, round 5)
        };
        auto seg_sig1 = [&]() {
            __syncthreads();
            const float* buf = fuse + (seg & (NBUF - 1u)) * (8 * AREA);
            const int u = wave & 3, t0 = 2 * (wave >> 2);
#pragma unroll 2
            for (int a = 0; a < 8; a += 2) {
                const float* ar = buf + a * AREA;
                const lz_bh8 x = tr2(ar, 4 + u);
                mma32(tr2(ar, t0), x, acc_sig1[0]);
                mma32(tr2(ar, t0 + 1), x, acc_sig1[1]);
            }
            seg++;
            if constexpr (NBUF == 1) __syncthreads();   // single buffer: every wave has read before anyone writes the next segment
        };
        auto seg_sig0 = [&]() {
            __syncthreads();
            const int u = (wave + 2) & 7;                    // waves 6, 7, 0, 1, 2 own columns 0 .. 4
            if (u < 5) {
                const float* buf = fuse + (seg & (NBUF - 1u)) * (8 * AREA);
#pragma unroll 2
                for (int a = 0; a < 8; a += 2) {
                    const float* ar = buf + a * AREA;
                    const lz_bh8 x = tr2(ar, 4 + u);
#pragma unroll
                    for (int t = 0; t < 4; t++) mma32(tr2(ar, t), x, acc_sig0[t]);
                }
            }
            seg++;
            if constexpr (NBUF == 1) __syncthreads();   // single buffer: every wave has read before anyone writes the next segment
        };
        auto seg_aud1 = [&]() {
            __syncthreads();
            const float* buf = fuse + (seg & (NBUF - 1u)) * (8 * AREA);
#pragma unroll 2
            for (int a = 0; a < 8; a += 2) {
                const float* ar = buf + a * AREA;
                mma32(tr2(ar, wave >> 2), tr2(ar, 2 + (wave & 3)), acc_aud1[0]);
                if (wave >= 4) mma32(tr2(ar, 6), tr2(ar, 7 + (wave & 3)), acc_c2t);
            }
            seg++;
            if constexpr (NBUF == 1) __syncthreads();   // single buffer: every wave has read before anyone writes the next segment
        };
        auto seg_x3 = [&]() {
            __syncthreads();
            if (wave >= 1) {
                const float* buf = fuse + (seg & (NBUF - 1u)) * (8 * AREA);
#pragma unroll 2
                for (int a = 0; a < 8; a += 2) {
                    const float* ar = buf + a * AREA;
                    const lz_bh8 g = tr2(ar, wave - 1);
#pragma unroll
                    for (int u = 0; u < 3; u++) mma32(g, tr2(ar, 7 + u), acc_x3[u]);
                }
            }
            seg++;
            if constexpr (NBUF == 1) __syncthreads();   // single buffer: every wave has read before anyone writes the next segment
        };
        lz_v4 xa[3];
        if constexpr (FUSE) {
            xa[0] = x_load(LZ_R16_X_S2C / 2); xa[1] = x_load(LZ_R16_X_S2C / 2 + 1); xa[2] = x_load(LZ_R16_X_S2C / 2 + 2);
        }
        __builtin_amdgcn_sched_barrier(0);
        const lz_v4 l_att0 = in.att0, l_att1 = in.att1, l_c0 = in.c0, l_c1 = in.c1, l_c2 = in.c2, l_c3 = in.c3, l_u0 = in.u0, l_u1 = in.u1,
                    l_e = in.e, l_mk = in.mk, l_clr = in.clr;
        const float g_sig = in.g_sig, g_aa = in.g_aa, g_ae = in.g_ae, g_un = in.g_un, g_r0 = in.g_r0, g_r1 = in.g_r1, g_r2 = in.g_r2;
        float att[8] = {l_att0[0], l_att0[1], l_att0[2], l_att0[3], l_att1[0], l_att1[1], l_att1[2], l_att1[3]};
        float c1[16], u1[8], e1[4];
        if constexpr (H16) {
            lz_unpack_pair(l_c0, c1, c1 + 4);
            lz_unpack_pair(l_c1, c1 + 8, c1 + 12);
            lz_unpack_pair(l_u0, u1, u1 + 4);
            float pad[4];
            lz_unpack_pair(l_e, e1, pad);
        } else {
#pragma unroll
            for (int r = 0; r < 4; r++) {
                c1[r] = l_c0[r]; c1[4 + r] = l_c1[r]; c1[8 + r] = l_c2[r]; c1[12 + r] = l_c3[r];
                u1[r] = l_u0[r]; u1[4 + r] = l_u1[r];
                e1[r] = l_e[r];
            }
        }
        const uint32_t w0 = lz_fbits(l_mk[0]), w1 = lz_fbits(l_mk[1]), w2 = lz_fbits(l_mk[2]);
        const uint32_t mk_a1 = w0 & 0xffffu, mk_s1 = w0 >> 16, mk_s2 = w1 & 0xffffu, mk_c1 = w1 >> 16, mk_u1 = w2 & 0xffu, mk_e1 = (w2 >> 8) & 0xfu;
        // the four scalars sit one per q lane of the sample
        const float norm = __shfl(l_mk[3], s, 64), eyeatt = __shfl(l_mk[3], s + 16, 64), upre = __shfl(l_mk[3], s + 32, 64),
                    sigma = __shfl(l_mk[3], s + 48, 64);

        // uncertainty / colour heads: d loss / d pre-activation, and the skinny layers' weight gradients
        const float du = g_un * lz_sigmoidf(upre);
        if (valid) {
#pragma unroll
            for (int k = 0; k < 8; k++) acc_u2[k] = lz_fmaf(du, u1[k], acc_u2[k]);
        }
        float dc[3];
#pragma unroll
        for (int c = 0; c < 3; c++) {
            const float sg = lz_sigmoidf(l_clr[c]);
            dc[c] = (c == 0 ? g_r0 : (c == 1 ? g_r1 : g_r2)) * 1.002f * sg * (1.0f - sg);
            if constexpr (!FUSE) {   // FUSE: color_net.1's weight gradient is one more product on the matrix cores (aud1 segment)
                if (valid) {
#pragma unroll
                    for (int k = 0; k < 16; k++) acc_c2[c][k] = lz_fmaf(dc[c], c1[k], acc_c2[c][k]);
                }
            }
        }

        float dgeo[16];
        float dind;
        {
            float dc1[16];
#pragma unroll
            for (int t = 0; t < 4; t++)
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    const int f = 16 * t + 4 * q + r, k = 4 * t + r;
                    float v = wv[LZ_WV_C2 + f] * dc[0];
                    v = lz_fmaf(wv[LZ_WV_C2 + 64 + f], dc[1], v);
                    v = lz_fmaf(wv[LZ_WV_C2 + 128 + f], dc[2], v);
                    dc1[k] = LZ_MASK_KEEP(RC, mk_c1, k, v);
                }
            if constexpr (FUSE) {
#pragma unroll
                for (int t = 0; t < 4; t++) g_put(t, dc1[4 * t], dc1[4 * t + 1], dc1[4 * t + 2], dc1[4 * t + 3]);
            } else {
                if constexpr (H16) {
                    lz_dump_pair_chained(rb + 4 * q, LZ_R16_G_C1H / 2, dc1, 0);
                    lz_dump_pair_chained(rb + 4 * q, LZ_R16_G_C1H / 2 + 1, dc1, 2);
                } else {
                    lz_dump_chained<4>(rb, q, LZ_BWD_G_C1H, dc1);
                }
            }
            float dxc[21];
            layer_bwd(std::integral_constant<int, LZ_L_C1>{}, dc1, dxc);
#pragma unroll
            for (int k = 0; k < 16; k++) dgeo[k] = dxc[4 + k];
            dind = dxc[20];
        }
        if (valid) acc_ind += dind;
        const float dh0 = g_sig * sigma;
        if constexpr (FUSE) {
            g_put(4, dh0, 0.0f, 0.0f, 0.0f);                 // tile 4: the sigma row (columns 4 q are copies in padding rows of dW)
#pragma unroll
            for (int pp = 0; pp < 3; pp++) { x_put(5 + 2 * pp, xa[pp], false); x_put(6 + 2 * pp, xa[pp], true); }   // X_S2C: 6 tiles
            xa[0] = x_load(LZ_R16_X_S1 / 2); xa[1] = x_load(LZ_R16_X_S1 / 2 + 1);                                    // for the next segment
            seg_c1h();
        } else {   // every q lane writes: columns 1..3 (f32) / 4 q (f16) of this tile are padding rows of dW
            if constexpr (H16) lz_dump_pair(rb + 4 * q, LZ_R16_G_C1H / 2 + 2, dh0, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f);   // tile 4, column 0
            else rb[lz_tcol(LZ_BWD_G_C1H + 64 + q)] = dh0;
        }
        float dencx[9], dencw[8], determ;
        {
            float ds2[16];
            layer_bwd(std::integral_constant<int, LZ_L_S3>{}, dgeo, ds2);
#pragma unroll
            for (int t = 0; t < 4; t++)
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    const int k = 4 * t + r;
                    const float v = lz_fmaf(wv[LZ_WV_SIG + 16 * t + 4 * q + r], dh0, ds2[k]);
                    ds2[k] = LZ_MASK_KEEP(RC, mk_s2, k, v);
                }
            if constexpr (FUSE) {
#pragma unroll
                for (int t = 0; t < 4; t++) g_put(t, ds2[4 * t], ds2[4 * t + 1], ds2[4 * t + 2], ds2[4 * t + 3]);
                x_put(4, xa[0], false); x_put(5, xa[0], true); x_put(6, xa[1], false); x_put(7, xa[1], true);             // X_S1: 4 tiles
                if constexpr (RC) { xa[0] = rc_sig0_pair(0); xa[1] = rc_sig0_pair(1); xa[2] = rc_sig0_pair(2); }
                else { xa[0] = x_load(LZ_R16_X_SIG0 / 2); xa[1] = x_load(LZ_R16_X_SIG0 / 2 + 1); xa[2] = x_load(LZ_R16_X_SIG0 / 2 + 2); }
                seg_sig1();
            } else {
                if constexpr (H16) {
                    lz_dump_pair_chained(rb + 4 * q, LZ_R16_G_S2 / 2, ds2, 0);
                    lz_dump_pair_chained(rb + 4 * q, LZ_R16_G_S2 / 2 + 1, ds2, 2);
                } else {
                    lz_dump_chained<4>(rb, q, LZ_BWD_G_S2, ds2);
                }
            }
            float ds1[16];
            layer_bwd(std::integral_constant<int, LZ_L_S2>{}, ds2, ds1);
#pragma unroll
            for (int k = 0; k < 16; k++) ds1[k] = LZ_MASK_KEEP(RC, mk_s1, k, ds1[k]);
            if constexpr (FUSE) {
#pragma unroll
                for (int t = 0; t < 4; t++) g_put(t, ds1[4 * t], ds1[4 * t + 1], ds1[4 * t + 2], ds1[4 * t + 3]);
                x_put(4, xa[0], false); x_put(5, xa[0], true); x_put(6, xa[1], false); x_put(7, xa[1], true); x_put(8, xa[2], false);   // X_SIG0: 5 tiles
                if constexpr (RC) rc_a1_pairs(xa[0], xa[1]);
                else { xa[0] = x_load(LZ_R16_X_A1 / 2); xa[1] = x_load(LZ_R16_X_A1 / 2 + 1); }
                seg_sig0();
                if constexpr (RC) nxr = fetch_rc(next);   // the next slice's enc_x operand, direction and upstream gradients
                else nx = fetch(next);               // the next slice's state row and upstream gradients: in flight during the last two segments
            } else {
                if constexpr (H16) {
                    lz_dump_pair_chained(rb + 4 * q, LZ_R16_G_S1 / 2, ds1, 0);
                    lz_dump_pair_chained(rb + 4 * q, LZ_R16_G_S1 / 2 + 1, ds1, 2);
                } else {
                    lz_dump_chained<4>(rb, q, LZ_BWD_G_S1, ds1);
                }
            }
            float dxs[18];
            layer_bwd(std::integral_constant<int, LZ_L_S1>{}, ds1, dxs);
#pragma unroll
            for (int i = 0; i < 9; i++) dencx[i] = dxs[i];
#pragma unroll
            for (int k = 0; k < 8; k++) dencw[k] = dxs[9 + k];
            determ = dxs[17];   // meaningful on lanes q == 0
        }
        float datt[8];
        {
            if constexpr (RC) {   // att from the packed halves the sink kept
                const lz_u4 a16 = {rs.att16[0][0], rs.att16[0][1], rs.att16[1][0], rs.att16[1][1]};
                const lz_h8 ah = __builtin_bit_cast(lz_h8, a16);
#pragma unroll
                for (int k = 0; k < 8; k++) att[k] = (float)ah[k];
            }
            const float inv = norm > 0.0f ? g_aa / norm : 0.0f;
#pragma unroll
            for (int t = 0; t < 2; t++)
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    const int k = 4 * t + r;
                    datt[k] = lz_fmaf(lenca[16 * t + 4 * q + r], dencw[k], inv * att[k]);
                    if (valid) acc_enca[k] = lz_fmaf(att[k], dencw[k], acc_enca[k]);
                }
            if constexpr (FUSE) {
                g_put(0, datt[0], datt[1], datt[2], datt[3]);
                g_put(1, datt[4], datt[5], datt[6], datt[7]);
                x_put(2, xa[0], false); x_put(3, xa[0], true); x_put(4, xa[1], false); x_put(5, xa[1], true);             // X_A1: 4 tiles
                // color_net.1 (3 x 64): G = d loss / d (colour pre-activation) in columns 0 .. 2 of a tile, X = its input c1 (the state row's pairs)
                g_put(6, q == 0 ? dc[0] : 0.0f, q == 0 ? dc[1] : 0.0f, q == 0 ? dc[2] : 0.0f, 0.0f);
                x_put(7, l_c0, false); x_put(8, l_c0, true); x_put(9, l_c1, false); x_put(10, l_c1, true);
                if constexpr (RC) { xa[0] = rc_sig0_pair(0); xa[1] = rc_sig0_pair(1); }
                else { xa[0] = x_load(LZ_R16_X_SIG0 / 2); xa[1] = x_load(LZ_R16_X_SIG0 / 2 + 1); }                         // enc_x again, for x3
                seg_aud1();
            } else {
                if constexpr (H16) lz_dump_pair_chained(rb + 4 * q, LZ_R16_G_ATT / 2, datt, 0);
                else lz_dump_chained<2>(rb, q, LZ_BWD_G_ATT, datt);
            }
        }
        // uncertainty: unc = softplus(u); its input is detached (network.py:241-249): weight gradients only
        float du1[8];
#pragma unroll
        for (int t = 0; t < 2; t++)
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const int k = 4 * t + r;
                du1[k] = LZ_MASK_KEEP(RC, mk_u1, k, wv[LZ_WV_U2 + 16 * t + 4 * q + r] * du);
            }
        float de1[4] = {0.0f, 0.0f, 0.0f, 0.0f};   // stays zero without an eye input: the stacked reduction over G_x reads these columns
        if (has_eye) {
            const float det0 = __shfl(determ, s, 64);   // from lane (s, q = 0)
            const float deye = lz_fmaf(eye_v, det0, g_ae);
            const float de2 = deye * eyeatt * (1.0f - eyeatt);
            if (valid) {
#pragma unroll
                for (int r = 0; r < 4; r++) acc_e2[r] = lz_fmaf(de2, e1[r], acc_e2[r]);
            }
#pragma unroll
            for (int r = 0; r < 4; r++) de1[r] = LZ_MASK_KEEP(RC, mk_e1, r, wv[LZ_WV_E2 + 4 * q + r] * de2);
            float dxe[9];
            layer_bwd(std::integral_constant<int, LZ_L_E1>{}, de1, dxe);
#pragma unroll
            for (int i = 0; i < 9; i++) dencx[i] += dxe[i];
        }
        if constexpr (FUSE) {   // G_X tiles 4 .. 6 (the first four follow the A2 product below); same buffer, same segment
            g_put(4, de1[0], de1[1], de1[2], de1[3]);
            g_put(5, du1[0], du1[1], du1[2], du1[3]);
            g_put(6, du1[4], du1[5], du1[6], du1[7]);
        } else {   // G_X = [aud_ch_att_net.0 64 | eye_att_net.0 16 | unc_net.0 32]: the last three tiles
            if constexpr (H16) {
                lz_dump_pair(rb + 4 * q, LZ_R16_G_X / 2 + 2, de1[0], de1[1], de1[2], de1[3], du1[0], du1[1], du1[2], du1[3]);
                lz_dump_pair(rb + 4 * q, LZ_R16_G_X / 2 + 3, du1[4], du1[5], du1[6], du1[7], 0.0f, 0.0f, 0.0f, 0.0f);
            } else {
                lz_dump_chained<1>(rb, q, LZ_BWD_G_X + 64, de1);
                lz_dump_chained<2>(rb, q, LZ_BWD_G_X + 80, du1);
            }
        }
        {
            float da1[16];
            layer_bwd(std::integral_constant<int, LZ_L_A2>{}, datt, da1);
#pragma unroll
            for (int k = 0; k < 16; k++) da1[k] = LZ_MASK_KEEP(RC, mk_a1, k, da1[k]);
            if constexpr (FUSE) {
#pragma unroll
                for (int t = 0; t < 4; t++) g_put(t, da1[4 * t], da1[4 * t + 1], da1[4 * t + 2], da1[4 * t + 3]);
                x_put(7, xa[0], false); x_put(8, xa[0], true); x_put(9, xa[1], false);                                     // X_SIG0 tiles 0 .. 2 (enc_x and the slot of feature 32 + q)
                seg_x3();
            } else {
                if constexpr (H16) {
                    lz_dump_pair_chained(rb + 4 * q, LZ_R16_G_X / 2, da1, 0);
                    lz_dump_pair_chained(rb + 4 * q, LZ_R16_G_X / 2 + 1, da1, 2);
                } else {
                    lz_dump_chained<4>(rb, q, LZ_BWD_G_X, da1);
                }
            }
            float dxa[9];
            layer_bwd(std::integral_constant<int, LZ_L_A1>{}, da1, dxa);
#pragma unroll
            for (int i = 0; i < 9; i++) dencx[i] += dxa[i];
        }
        {
#pragma unroll
            for (int i = 0; i < 9; i++) dencq[(size_t)(4 * i) * M] = dencx[i];   // [3 planes][12 levels][M], level-major; feature 4 i + q
        }
        slice = next;
    }
    if constexpr (FUSE) {
        // this workgroup's partial tiles, in the image lz_k_head_grad_w16 writes (tile = product base + t * KBT + u, fragment order):
        // lz_k_head_grad_w_reduce sums the images of all workgroups
        float* part = parts + (size_t)blockIdx.x * (LZ_DW_TILES * 256);
        auto put_tile = [&](int tile, const lz_f4& a) {
#pragma unroll
            for (int r = 0; r < 4; r++) part[(tile * 4 + r) * 64 + lane] = a[r];
        };
        if (wave < 6) {
#pragma unroll
            for (int t = 0; t < 5; t++) put_tile(LZ_DW_T_C1H + t * 6 + wave, acc_c1h[t]);
        }
        put_tile(LZ_DW_T_SIG1 + (2 * (wave >> 2)) * 4 + (wave & 3), acc_sig1[0]);
        put_tile(LZ_DW_T_SIG1 + (2 * (wave >> 2) + 1) * 4 + (wave & 3), acc_sig1[1]);
        if (((wave + 2) & 7) < 5) {
#pragma unroll
            for (int t = 0; t < 4; t++) put_tile(LZ_DW_T_SIG0 + t * 5 + ((wave + 2) & 7), acc_sig0[t]);
        }
        put_tile(LZ_DW_T_AUD1 + (wave >> 2) * 4 + (wave & 3), acc_aud1[0]);
        if (wave >= 1) {
#pragma unroll
            for (int u = 0; u < 3; u++) put_tile(LZ_DW_T_X3 + (wave - 1) * 3 + u, acc_x3[u]);
        }
        if (wave >= 4 && lane < 16) {   // rows 0 .. 2 of the tile sit in registers 0 .. 2 of lanes 0 .. 15 (column = lane): dW_color1[c][16 u + lane]
#pragma unroll
            for (int c = 0; c < 3; c++) atomicAdd(O.small + 84 + 64 * c + 16 * (wave & 3) + lane, acc_c2t[c]);
        }
    }
    // per-lane sums -> 16 sample lanes -> the workgroup's waves in LDS -> one atomic per value (layout of lz_head_bwd_out.small)
    constexpr int NRED = 32 + 4 + 16 + 32 + 192;
    __syncthreads();
    float* red = wl;
    float* mine = red + (threadIdx.x >> 6) * NRED;
    auto put = [&](float v, int slot) {
#pragma unroll
        for (int off = 8; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
        if (s == 0) mine[slot] = v;
    };
#pragma unroll
    for (int k = 0; k < 8; k++) put(acc_enca[k], 16 * (k >> 2) + 4 * q + (k & 3));
    put(acc_ind, 32 + q);
#pragma unroll
    for (int k = 0; k < 4; k++) put(acc_e2[k], 36 + 4 * q + k);
#pragma unroll
    for (int k = 0; k < 8; k++) put(acc_u2[k], 52 + 16 * (k >> 2) + 4 * q + (k & 3));
    if constexpr (!FUSE) {
#pragma unroll
        for (int c = 0; c < 3; c++)
#pragma unroll
            for (int k = 0; k < 16; k++) put(acc_c2[c][k], 84 + 64 * c + 16 * (k >> 2) + 4 * q + (k & 3));
    }
    __syncthreads();
    if (threadIdx.x < (FUSE ? 84 : NRED)) {
        float v = 0.0f;
        for (int w = 0; w < LZ_BWD_WG / 64; w++) v += red[w * NRED + threadIdx.x];
        if (v != 0.0f) atomicAdd(O.small + threadIdx.x, v);
    }
}

// ------------------------------------------------------------------------------------------------
// plane coordinates of the three table scatters: [3][M][2] = ((x, y) | (y, z) | (x, z)) mapped like the forward (lz_head_gather.h)
// ------------------------------------------------------------------------------------------------
__global__ void lz_k_plane_coords(const float* __restrict__ xyzs, uint32_t M, float bound, float* __restrict__ out) {
    const uint32_t m = blockIdx.x * blockDim.x + threadIdx.x;
    if (m >= M) return;
    const float two_bound = 2.0f * bound;
    const uint32_t tb_bits = __float_as_uint(two_bound);
    const bool pow2 = (tb_bits & 0x007fffffu) == 0u && tb_bits > 0x00800000u && tb_bits < 0x7f000000u;
    const float px = xyzs[(size_t)m * 3], py = xyzs[(size_t)m * 3 + 1], pz = xyzs[(size_t)m * 3 + 2];
    float x01, y01, z01;
    if (pow2) {
        const float inv = __uint_as_float(0x7f000000u - tb_bits);
        x01 = (px + bound) * inv; y01 = (py + bound) * inv; z01 = (pz + bound) * inv;
    } else {
        x01 = (px + bound) / two_bound; y01 = (py + bound) / two_bound; z01 = (pz + bound) / two_bound;
    }
    float2* o = reinterpret_cast<float2*>(out);
    o[m] = make_float2(x01, y01);
    o[(size_t)M + m] = make_float2(y01, z01);
    o[2 * (size_t)M + m] = make_float2(x01, z01);
}

extern "C" int lz_triplane_plane_coords(const float* xyzs, uint32_t M, float bound, float* out, lz_stream_t stream) {
    LZ_REQUIRE(M == 0 || (xyzs && out), LZ_ERR_BAD_ARGUMENT, "triplane_plane_coords: null tensor");
    LZ_REQUIRE(((uintptr_t)out & 7u) == 0, LZ_ERR_BAD_ARGUMENT, "triplane_plane_coords: out must be 8-byte aligned");
    if (M == 0) return LZ_OK;
    hipLaunchKernelGGL(lz_k_plane_coords, dim3(lz_div_up(M, 256)), dim3(256), 0, lz_st(stream), xyzs, M, bound, out);
    LZ_CHECK_LAUNCH("triplane_plane_coords");
    return LZ_OK;
}

// ------------------------------------------------------------------------------------------------
// host
// ------------------------------------------------------------------------------------------------
static void lz_fill_head_args(const lz_head_params* p, LzHeadArgs& a) {
    a.emb[0] = p->emb_xy; a.emb[1] = p->emb_yz; a.emb[2] = p->emb_xz;
    a.offsets = p->offsets; a.packed = reinterpret_cast<const float*>(p->packed); a.enc_a = p->enc_a;
    a.ind_code = p->ind_code; a.eye = p->eye; a.bound = p->bound; a.testing = 0;
    for (int l = 0; l < 12; l++) {
        const float sc = exp2f((float)l * p->S) * (float)p->H - 1.0f;
        a.scale[l] = sc;
        a.res[l] = (uint32_t)ceilf(sc) + 1u;
    }
}

static uint32_t lz_rec_grid(uint32_t M, uint32_t wg) {
    const int n_cu = lz_cu_count();   // of the current device, per call (cached per device)
    const uint32_t want = lz_div_up(lz_div_up(M, 16), wg / 64);
    return want < (uint32_t)n_cu ? want : (uint32_t)n_cu;
}

extern "C" int lz_triplane_head_forward_record(const lz_head_params* p, const float* xyzs, const float* dirs, uint32_t M, float* sigmas,
                                               float* rgbs, float* amb_aud, float* amb_eye, float* unc, void* rec_v, float* state,
                                               int record_f16, lz_stream_t stream) {
    float* rec = static_cast<float*>(rec_v);
    LZ_REQUIRE(p && xyzs && dirs && sigmas && rgbs && amb_aud && unc && rec && state, LZ_ERR_BAD_ARGUMENT, "triplane_head_forward_record: null tensor");
    LZ_REQUIRE(p->emb_xy && p->emb_yz && p->emb_xz && p->offsets && p->packed && p->enc_a, LZ_ERR_BAD_ARGUMENT,
               "triplane_head_forward_record: incomplete lz_head_params");
    LZ_REQUIRE(p->precision == 0 && !p->testing, LZ_ERR_UNSUPPORTED, "triplane_head_forward_record: f32 training mode only");
    LZ_REQUIRE((((uintptr_t)rec | (uintptr_t)state) & 15u) == 0, LZ_ERR_BAD_ARGUMENT, "triplane_head_forward_record: rec / state must be 16-byte aligned");
    if (M == 0) return LZ_OK;
    LzHeadArgs a;
    lz_fill_head_args(p, a);
    if (record_f16)
        hipLaunchKernelGGL(lz_k_triplane_head_forward_rec<true>, dim3(lz_rec_grid(M, LZ_FREC_WG)), dim3(LZ_FREC_WG), 0, lz_st(stream), a, xyzs, dirs, M,
                           sigmas, rgbs, amb_aud, amb_eye, unc, rec, state);
    else
        hipLaunchKernelGGL(lz_k_triplane_head_forward_rec<false>, dim3(lz_rec_grid(M, LZ_FREC_WG)), dim3(LZ_FREC_WG), 0, lz_st(stream), a, xyzs, dirs, M,
                           sigmas, rgbs, amb_aud, amb_eye, unc, rec, state);
    LZ_CHECK_LAUNCH("triplane_head_forward_record");
    return LZ_OK;
}

extern "C" int lz_triplane_head_backward_recorded(const lz_head_params* p, const float* state, uint32_t M, const float* g_sigma,
                                                  const float* g_rgb, const float* g_amb_aud, const float* g_amb_eye, const float* g_unc,
                                                  const lz_head_bwd_out* out, int record_f16, const void* packed_bwd16,
                                                  lz_stream_t stream) {
    LZ_REQUIRE(p && state && g_sigma && g_rgb && g_amb_aud && g_unc && out, LZ_ERR_BAD_ARGUMENT, "triplane_head_backward_recorded: null tensor");
    LZ_REQUIRE(p->packed && p->enc_a, LZ_ERR_BAD_ARGUMENT, "triplane_head_backward_recorded: incomplete lz_head_params");
    LZ_REQUIRE(p->precision == 0 && !p->testing, LZ_ERR_UNSUPPORTED, "triplane_head_backward_recorded: f32 training mode only");
    const lz_head_bwd_out& o = *out;
    LZ_REQUIRE(o.denc && o.small && o.rec, LZ_ERR_BAD_ARGUMENT, "triplane_head_backward_recorded: incomplete lz_head_bwd_out");
    LZ_REQUIRE((((uintptr_t)o.rec | (uintptr_t)state) & 15u) == 0, LZ_ERR_BAD_ARGUMENT, "triplane_head_backward_recorded: rec / state must be 16-byte aligned");
    if (M == 0) return LZ_OK;
    LzHeadBwdArgs a;
    lz_fill_head_args(p, a.fwd);
    a.g_sigma = g_sigma; a.g_rgb = g_rgb; a.g_amb_aud = g_amb_aud; a.g_amb_eye = g_amb_eye; a.g_unc = g_unc;
    a.o = o;
    a.wb16 = packed_bwd16;
    LZ_REQUIRE(!packed_bwd16 || (record_f16 && ((uintptr_t)packed_bwd16 & 15u) == 0), LZ_ERR_BAD_ARGUMENT,
               "triplane_head_backward_recorded: the f16 matrix path goes with f16 records and 16-byte aligned fragments");
    const dim3 grid(lz_rec_grid(M, LZ_BWD_WG)), block(LZ_BWD_WG);
    if (packed_bwd16) hipLaunchKernelGGL((lz_k_triplane_head_backward_rec<true, true>), grid, block, 0, lz_st(stream), a, state, M, (float*)nullptr);
    else if (record_f16) hipLaunchKernelGGL((lz_k_triplane_head_backward_rec<true, false>), grid, block, 0, lz_st(stream), a, state, M, (float*)nullptr);
    else hipLaunchKernelGGL((lz_k_triplane_head_backward_rec<false, false>), grid, block, 0, lz_st(stream), a, state, M, (float*)nullptr);
    LZ_CHECK_LAUNCH("triplane_head_backward_recorded");
    return LZ_OK;
}

// The all-f16 arrangement with the weight gradients of the wide layers reduced inside the backward kernel (FUSE above): reads the state
// rows and the X half of the f16 records the forward wrote, writes denc / small and the five weight-gradient matrices; the G half of
// the records is never touched.  workspace: lz_triplane_head_grad_w_workspace() bytes.
extern "C" int lz_triplane_head_backward_recorded_dw16(const lz_head_params* p, const float* state, const void* rec16, uint32_t M,
                                                       const float* g_sigma, const float* g_rgb, const float* g_amb_aud, const float* g_amb_eye,
                                                       const float* g_unc, const lz_head_bwd_out* out, const void* packed_bwd16, uint32_t k_sig0,
                                                       float* dW_x3, float* dW_aud1, float* dW_sig0, float* dW_sig1, float* dW_c1h, void* workspace,
                                                       lz_stream_t stream) {
    LZ_REQUIRE(p && state && rec16 && g_sigma && g_rgb && g_amb_aud && g_unc && out && workspace, LZ_ERR_BAD_ARGUMENT,
               "triplane_head_backward_recorded_dw16: null tensor");
    LZ_REQUIRE(dW_x3 && dW_aud1 && dW_sig0 && dW_sig1 && dW_c1h, LZ_ERR_BAD_ARGUMENT, "triplane_head_backward_recorded_dw16: null weight-gradient output");
    LZ_REQUIRE(p->packed && p->enc_a, LZ_ERR_BAD_ARGUMENT, "triplane_head_backward_recorded_dw16: incomplete lz_head_params");
    LZ_REQUIRE(p->precision == 0 && !p->testing, LZ_ERR_UNSUPPORTED, "triplane_head_backward_recorded_dw16: training mode, f32 packed weights for the skinny rows");
    LZ_REQUIRE(k_sig0 == 68 || k_sig0 == 69, LZ_ERR_BAD_ARGUMENT, "triplane_head_backward_recorded_dw16: sigma_net.0 takes 68 or 69 inputs");
    const lz_head_bwd_out& o = *out;
    LZ_REQUIRE(o.denc && o.small, LZ_ERR_BAD_ARGUMENT, "triplane_head_backward_recorded_dw16: incomplete lz_head_bwd_out");
    LZ_REQUIRE((((uintptr_t)rec16 | (uintptr_t)state | (uintptr_t)packed_bwd16) & 15u) == 0, LZ_ERR_BAD_ARGUMENT,
               "triplane_head_backward_recorded_dw16: rec / state / fragments must be 16-byte aligned");
    if (M == 0) {   // no sample: the weight gradients are zero
        hipStream_t st = lz_st(stream);
        (void)hipMemsetAsync(dW_x3, 0, 112 * 36 * 4, st); (void)hipMemsetAsync(dW_aud1, 0, 32 * 64 * 4, st); (void)hipMemsetAsync(dW_sig0, 0, 64 * k_sig0 * 4, st);
        (void)hipMemsetAsync(dW_sig1, 0, 64 * 64 * 4, st); (void)hipMemsetAsync(dW_c1h, 0, 65 * 84 * 4, st);
        return LZ_OK;
    }
    LzHeadBwdArgs a;
    lz_fill_head_args(p, a.fwd);
    a.g_sigma = g_sigma; a.g_rgb = g_rgb; a.g_amb_aud = g_amb_aud; a.g_amb_eye = g_amb_eye; a.g_unc = g_unc;
    a.o = o;
    a.o.rec = const_cast<float*>(static_cast<const float*>(rec16));   // read only: the X half
    a.wb16 = packed_bwd16;
    const uint32_t grid = lz_rec_grid(M, LZ_BWD_WG);
    LZ_REQUIRE(grid <= LZ_DW_MAX_PARTS, LZ_ERR_UNSUPPORTED, "triplane_head_backward_recorded_dw16: more workgroups than partial images");
    if (packed_bwd16)   // data gradient on the f16 matrix cores: two LDS buffers of G / X tiles
        hipLaunchKernelGGL((lz_k_triplane_head_backward_rec<true, true, true>), dim3(grid), dim3(LZ_BWD_WG), 0, lz_st(stream), a, state, M,
                           static_cast<float*>(workspace));
    else                // f32 data-gradient chain (its 97 KB weight image leaves room for one buffer)
        hipLaunchKernelGGL((lz_k_triplane_head_backward_rec<true, false, true>), dim3(grid), dim3(LZ_BWD_WG), 0, lz_st(stream), a, state, M,
                           static_cast<float*>(workspace));
    LZ_CHECK_LAUNCH("triplane_head_backward_recorded_dw16");
    return lz_head_grad_w_reduce_launch(static_cast<const float*>(workspace), grid, true, k_sig0, dW_x3, dW_aud1, dW_sig0, dW_sig1, dW_c1h, stream);
}

// The recomputing -O arrangement (round 5): lz_triplane_head_backward_recorded_dw16 without record and state -- the forward
// (lz_triplane_head_forward_encx_f16) left only the enc_x halves (encx16, 80 bytes per sample); this kernel recomputes the f16 forward chain
// of every slice from them (packed_f16 / packed_unc: the forward's weight images, dirs: the view directions for SH) and then runs the fused
// backward: data gradient on the f16 matrix cores (packed_bwd16, required), weight gradients of the wide layers reduced in the kernel.
// Same outputs as the recorded pair, bit for bit (tests/test_gpu_train_step.py).  p: f32 packed weights (the skinny rows), training mode.
extern "C" int lz_triplane_head_backward_encx_dw16(const lz_head_params* p, const void* packed_f16, const void* packed_unc, const void* encx16,
                                                   const float* dirs, uint32_t M, const float* g_sigma, const float* g_rgb, const float* g_amb_aud,
                                                   const float* g_amb_eye, const float* g_unc, const lz_head_bwd_out* out, const void* packed_bwd16,
                                                   uint32_t k_sig0, float* dW_x3, float* dW_aud1, float* dW_sig0, float* dW_sig1, float* dW_c1h, void* workspace,
                                                   lz_stream_t stream) {
    LZ_REQUIRE(p && packed_f16 && packed_unc && encx16 && dirs && g_sigma && g_rgb && g_amb_aud && g_unc && out && packed_bwd16 && workspace, LZ_ERR_BAD_ARGUMENT,
               "triplane_head_backward_encx_dw16: null tensor");
    LZ_REQUIRE(dW_x3 && dW_aud1 && dW_sig0 && dW_sig1 && dW_c1h, LZ_ERR_BAD_ARGUMENT, "triplane_head_backward_encx_dw16: null weight-gradient output");
    LZ_REQUIRE(p->packed && p->enc_a, LZ_ERR_BAD_ARGUMENT, "triplane_head_backward_encx_dw16: incomplete lz_head_params");
    LZ_REQUIRE(p->precision == 0 && !p->testing, LZ_ERR_UNSUPPORTED, "triplane_head_backward_encx_dw16: training mode, f32 packed weights for the skinny rows");
    LZ_REQUIRE(k_sig0 == 68 || k_sig0 == 69, LZ_ERR_BAD_ARGUMENT, "triplane_head_backward_encx_dw16: sigma_net.0 takes 68 or 69 inputs");
    const lz_head_bwd_out& o = *out;
    LZ_REQUIRE(o.denc && o.small, LZ_ERR_BAD_ARGUMENT, "triplane_head_backward_encx_dw16: incomplete lz_head_bwd_out");
    LZ_REQUIRE((((uintptr_t)encx16 | (uintptr_t)packed_f16 | (uintptr_t)packed_unc | (uintptr_t)packed_bwd16) & 15u) == 0, LZ_ERR_BAD_ARGUMENT,
               "triplane_head_backward_encx_dw16: encx16 / fragments must be 16-byte aligned");
    if (M == 0) {   // no sample: the weight gradients are zero
        hipStream_t st = lz_st(stream);
        (void)hipMemsetAsync(dW_x3, 0, 112 * 36 * 4, st); (void)hipMemsetAsync(dW_aud1, 0, 32 * 64 * 4, st); (void)hipMemsetAsync(dW_sig0, 0, 64 * k_sig0 * 4, st);
        (void)hipMemsetAsync(dW_sig1, 0, 64 * 64 * 4, st); (void)hipMemsetAsync(dW_c1h, 0, 65 * 84 * 4, st);
        return LZ_OK;
    }
    LzHeadBwdArgs a;
    lz_fill_head_args(p, a.fwd);
    a.g_sigma = g_sigma; a.g_rgb = g_rgb; a.g_amb_aud = g_amb_aud; a.g_amb_eye = g_amb_eye; a.g_unc = g_unc;
    a.o = o;
    a.o.rec = nullptr;
    a.wb16 = packed_bwd16;
    a.fw16 = packed_f16; a.unc16 = packed_unc; a.dirs = dirs;
    const uint32_t grid = lz_rec_grid(M, LZ_BWD_WG);
    LZ_REQUIRE(grid <= LZ_DW_MAX_PARTS, LZ_ERR_UNSUPPORTED, "triplane_head_backward_encx_dw16: more workgroups than partial images");
    hipLaunchKernelGGL((lz_k_triplane_head_backward_rec<true, true, true, true>), dim3(grid), dim3(LZ_BWD_WG), 0, lz_st(stream), a,
                       static_cast<const float*>(encx16), M, static_cast<float*>(workspace));
    LZ_CHECK_LAUNCH("triplane_head_backward_encx_dw16");
    return lz_head_grad_w_reduce_launch(static_cast<const float*>(workspace), grid, true, k_sig0, dW_x3, dW_aud1, dW_sig0, dW_sig1, dW_c1h, stream);
}
