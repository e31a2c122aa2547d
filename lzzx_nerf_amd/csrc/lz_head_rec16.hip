// lz_head_rec16.hip -- the recording forward of the training head on the f16 matrix cores: the forward of the reference's usual training
// mode (`-O` = --fp16: torch.cuda.amp.autocast around NeRFNetwork.forward, nerf_triplane/network.py:252-311, TrainerUtil.py:865) with
// the rounding sequence of lz_head_f16_slice.h, writing the f16 records and the state row that lz_triplane_head_backward_recorded
// (record_f16 = 1) and lz_triplane_head_grad_w_f16 consume (layouts: include/lzzx_nerf_hip.h LZ_R16_*, LZ_S16_*).
//
// The layer inputs a half Linear sees ARE the halves this kernel holds as MFMA B operands, so the record costs two byte-permutes per
// dword on top of the inference slice; 64 MFMAs per 16 samples instead of 379 f32 ones.  The uncertainty net (training only:
// network.py:241-249) adds five fragments that are packed separately (lz_head_pack_unc_f16), so the inference image and kernels stay
// as they are.  sigma / rgb / ambient outputs have the bits of lz_k_triplane_head_f16 on the same inputs.
#include "lz_head_fwd16_chain.h"   // the MLP chain itself, shared with the recomputing backward (lz_head_rec.hip, RC)
#include "lz_head_bwd_common.h"   // lz_blk / lz_tcol: the blocked record layout

#define LZ_FREC16_WG 1024

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

// the chain's sink of the recording forward: layer inputs into the X half of the sample's record, the state row next to it
struct LzRec16Sink {
    float* rbq;      // the sample's record row (dwords) + 4 q
    float* sbq;      // ... state row + 4 q
    float* sb;
    int q;
    __device__ __forceinline__ void x_pair_h8(int pair, const lz_h8& b) const { LZ_REC_STORE(lz_pair_words_h8(b), reinterpret_cast<lz_v4*>(rbq + 256 * pair)); }
    __device__ __forceinline__ void x_pair_f(int pair, float l0, float l1, float l2, float l3, float h0, float h1, float h2, float h3) const {
        LZ_REC_STORE(lz_pair_words_f(l0, l1, l2, l3, h0, h1, h2, h3), reinterpret_cast<lz_v4*>(rbq + 256 * pair));
    }
    __device__ __forceinline__ void s_pair_h8(int pair, const lz_h8& b) const { LZ_REC_STORE(lz_pair_words_h8(b), reinterpret_cast<lz_v4*>(sbq + 256 * pair)); }
    __device__ __forceinline__ void s_att(const lz_v4& w0, const lz_v4& w1) const {
        LZ_REC_STORE(w0, reinterpret_cast<lz_v4*>(sb + lz_tcol(LZ_ST_ATT + 4 * q)));
        LZ_REC_STORE(w1, reinterpret_cast<lz_v4*>(sb + lz_tcol(LZ_ST_ATT + 16 + 4 * q)));
    }
};
// ... and of the LIGHT forward (RC): nothing is kept but the enc_x operand itself (below)
struct LzNoSink {
    __device__ __forceinline__ void x_pair_h8(int, const lz_h8&) const {}
    __device__ __forceinline__ void x_pair_f(int, float, float, float, float, float, float, float, float) const {}
    __device__ __forceinline__ void s_pair_h8(int, const lz_h8&) const {}
    __device__ __forceinline__ void s_att(const lz_v4&, const lz_v4&) const {}
};

// RC = false: the recording forward (records + state row per sample).  RC = true: the LIGHT forward of the recomputing arrangement -- the
// same gather, chain and outputs, and instead of 1 216 bytes of record and state per sample only the enc_x operand the chain started from:
// encx16 [slice][5][64 lanes] dwords (80 bytes per sample; dwords 0 .. 3 = bx[0], dword 4 = bx[1]'s first), from which
// lz_triplane_head_backward_encx_dw16 recomputes everything else.
// Workgroups: the recording forward one of 1024 threads per CU; the light one (82 registers without the record's address and mask work, 66 KB of
// LDS) TWO of 768 per CU -- six waves per SIMD instead of four to cover its gathers.
#define LZ_FENCX16_WG 768
template <bool RC>
__global__ void __launch_bounds__(RC ? LZ_FENCX16_WG : LZ_FREC16_WG, RC ? 6 : 1)
lz_k_triplane_head_forward_rec16(LzHead16Args P, const lz_h8* __restrict__ packed_unc, const float* __restrict__ xyzs,
                                 const float* __restrict__ dirs, uint32_t M, float* __restrict__ sigmas, float* __restrict__ rgbs,
                                 float* __restrict__ amb_aud, float* __restrict__ amb_eye, float* __restrict__ unc_out, float* __restrict__ rec,
                                 float* __restrict__ st) {
    __shared__ lz_h8 wl[LZ_HEAD16_LDS_H8 + LZ_UNC16_FRAGS * 64];
    const uint32_t n_slices = (M + 15) / 16;
    const uint32_t slice_lo = (uint32_t)(((uint64_t)n_slices * blockIdx.x) / gridDim.x);
    const uint32_t slice_hi = (uint32_t)(((uint64_t)n_slices * (blockIdx.x + 1)) / gridDim.x);
    if (slice_lo >= slice_hi) return;
    constexpr uint32_t WG = RC ? LZ_FENCX16_WG : LZ_FREC16_WG;
    LzHead16Ctx hc;
    lz_head16_stage(P, wl, WG, hc);
    lz_h8* wl_unc = wl + LZ_HEAD16_LDS_H8;
    for (uint32_t i = threadIdx.x; i < (uint32_t)LZ_UNC16_FRAGS * 64; i += WG) wl_unc[i] = packed_unc[i];
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
        const uint32_t gslice = slice_lo + (uint32_t)slice;
        const size_t row = row_of(slice);   // lanes past the end repeat the last row: the same values are stored again

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
        LzFwd16Out o;
        if constexpr (RC) {
            float* eb = rec + (size_t)gslice * (5 * 64) + lane;      // [slice][5][lane]: five coalesced 256-byte rows
            const lz_u4v w0 = __builtin_bit_cast(lz_u4v, bx[0]);
            const lz_u4v w1 = __builtin_bit_cast(lz_u4v, bx[1]);
#pragma unroll
            for (int k = 0; k < 4; k++) eb[64 * k] = __uint_as_float(w0[k]);
            eb[64 * 4] = __uint_as_float(w1[0]);
            LzNoSink sink;
            lz_fwd16_chain(hc, wl_unc, lane, bx, cdx, cdy, cdz, indq, sink, o);
        } else {
            float* rb = lz_blk(rec, gslice, LZ_BWD_REC16 / 2, s);
            float* sb = lz_blk(st, gslice, LZ_FWD_STATE16, s);
            LzRec16Sink sink{rb + 4 * q, sb + 4 * q, sb, q};
            lz_fwd16_chain(hc, wl_unc, lane, bx, cdx, cdy, cdz, indq, sink, o);
            // ---------------- state words (the four lanes of a sample store the same values) ----------------
            const float sc = q == 0 ? o.norm : (q == 1 ? o.eyeatt : (q == 2 ? o.upre : o.sigma));
            lz_v4 w = {__uint_as_float(o.mk_a1 | (o.mk_s1 << 16)), __uint_as_float(o.mk_s2 | (o.mk_c1 << 16)), __uint_as_float(o.mk_u1 | (o.mk_e1 << 8)), sc};
            LZ_REC_STORE(w, reinterpret_cast<lz_v4*>(sb + lz_tcol(LZ_S16_MK + 4 * q)));
            lz_v4 cw = {o.cpre[0], o.cpre[1], o.cpre[2], 0.0f};
            LZ_REC_STORE(cw, reinterpret_cast<lz_v4*>(sb + lz_tcol(LZ_S16_CLR)));
        }
        // ---------------- outputs ----------------
        {
            sigmas[row] = o.sigma;
            amb_aud[row] = o.norm;
            if (amb_eye) amb_eye[row] = o.eyeatt;
            unc_out[row] = lz_softplusf(o.upre);
            const int qc = q < 2 ? q : 2;
            const float cv = q == 0 ? o.cpre[0] : (q == 1 ? o.cpre[1] : o.cpre[2]);
            const _Float16 sg = (_Float16)h_sigmoid(cv);   // network.py:275 in half: sigmoid, * 1.002, - 0.001, each rounded to half
            const _Float16 t1 = h_round((float)sg * 1.002f);
            rgbs[row * 3 + qc] = (float)h_round((float)t1 - 0.001f);
        }
        slice = next;
    }
}

static int lz_fwd16_launch(bool rc, const char* who, const lz_head_params* p, const void* packed_unc, const float* xyzs, const float* dirs, uint32_t M,
                           float* sigmas, float* rgbs, float* amb_aud, float* amb_eye, float* unc, float* rec_or_encx, float* state16, hipStream_t st) {
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
    const uint32_t wg = rc ? LZ_FENCX16_WG : LZ_FREC16_WG, per_cu = rc ? 2u : 1u;
    const uint32_t want = lz_div_up(lz_div_up(M, 16), wg / 64);
    const uint32_t grid = want < per_cu * (uint32_t)n_cu ? want : per_cu * (uint32_t)n_cu;
    if (rc) hipLaunchKernelGGL(lz_k_triplane_head_forward_rec16<true>, dim3(grid), dim3(LZ_FENCX16_WG), 0, st, a, reinterpret_cast<const lz_h8*>(packed_unc),
                               xyzs, dirs, M, sigmas, rgbs, amb_aud, amb_eye, unc, rec_or_encx, state16);
    else hipLaunchKernelGGL(lz_k_triplane_head_forward_rec16<false>, dim3(grid), dim3(LZ_FREC16_WG), 0, st, a, reinterpret_cast<const lz_h8*>(packed_unc),
                            xyzs, dirs, M, sigmas, rgbs, amb_aud, amb_eye, unc, rec_or_encx, state16);
    (void)who;
    return LZ_OK;
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
    lz_fwd16_launch(false, "triplane_head_forward_record_f16", p, packed_unc, xyzs, dirs, M, sigmas, rgbs, amb_aud, amb_eye, unc, static_cast<float*>(rec16), state16,
                    lz_st(stream));
    LZ_CHECK_LAUNCH("triplane_head_forward_record_f16");
    return LZ_OK;
}

// The LIGHT forward of the recomputing -O arrangement (round 5): the same arithmetic and outputs as lz_triplane_head_forward_record_f16, but
// instead of the records and the state row it leaves only encx16 -- the enc_x halves the MLP started from, LZ_ENCX16_BYTES(M) bytes,
// 80 per sample -- for lz_triplane_head_backward_encx_dw16, which recomputes every layer input and mask from them.
extern "C" int lz_triplane_head_forward_encx_f16(const lz_head_params* p, const void* packed_unc, const float* xyzs, const float* dirs, uint32_t M,
                                                 float* sigmas, float* rgbs, float* amb_aud, float* amb_eye, float* unc, void* encx16, lz_stream_t stream) {
    if (M == 0) return LZ_OK;
    LZ_REQUIRE(p && packed_unc && xyzs && dirs && sigmas && rgbs && amb_aud && unc && encx16, LZ_ERR_BAD_ARGUMENT, "triplane_head_forward_encx_f16: null tensor");
    LZ_REQUIRE(p->emb_xy && p->emb_yz && p->emb_xz && p->offsets && p->packed && p->enc_a, LZ_ERR_BAD_ARGUMENT,
               "triplane_head_forward_encx_f16: incomplete lz_head_params");
    LZ_REQUIRE(p->precision == 1 && !p->testing, LZ_ERR_UNSUPPORTED, "triplane_head_forward_encx_f16: precision 1 (lz_head_pack_weights_f16 image), training mode");
    LZ_REQUIRE((((uintptr_t)encx16 | (uintptr_t)packed_unc | (uintptr_t)p->packed) & 15u) == 0, LZ_ERR_BAD_ARGUMENT,
               "triplane_head_forward_encx_f16: encx16 / packed weights must be 16-byte aligned");
    lz_fwd16_launch(true, "triplane_head_forward_encx_f16", p, packed_unc, xyzs, dirs, M, sigmas, rgbs, amb_aud, amb_eye, unc, static_cast<float*>(encx16), nullptr,
                    lz_st(stream));
    LZ_CHECK_LAUNCH("triplane_head_forward_encx_f16");
    return LZ_OK;
}
