// lz_head_slice.h -- the f32 fused triplane head for ONE 16-sample slice of a wave (v_mfma_f32_16x16x4_f32), shared by the stand-alone
// head kernel (lz_head.hip: lz_k_triplane_head) and the fused frame kernel (lz_frame.hip: lz_k_frame), so that both evaluate
// NeRFNetwork.forward (nerf_triplane/network.py:252-311) with the same instruction sequence, bit for bit.  Lane (s = lane & 15,
// q = lane >> 4) owns sample s and a quarter of its features; see lz_head.hip for the layer chaining and the summation orders.
#ifndef LZ_HEAD_SLICE_H
#define LZ_HEAD_SLICE_H
#include "lz_common.h"
#include "lzzx_detmath.h"
#include "lzzx_sh_eval.h"
#include "lz_head_gather.h"
#include "lz_head_layers.h"

#define LZ_T 1             // sample tiles (of 16) per wave pass

// per-workgroup context: the packed weights and small tables staged in LDS (lz_head_stage), per-launch constants
struct LzHeadCtx {
    const float* wl;        // LDS: packed A fragments, then the VALU-layer rows
    const int* tab;         // LDS: the level table (lz_head_gather.h: LZ_LVTAB_*)
    const float* lenca;     // LDS: enc_a [32]
    const float* emb[3];    // the three planes' tables (global)
    float bound, two_bound, eye_v, indq;
    bool has_eye;
};

struct LzHeadOut {
    float sigma, rgb[3], ambaud, eyeatt, unc;   // every lane of a sample ends with the same bits
};

// LDS floats the stage needs: fragments + VALU rows + the level table (lz_head_gather.h: LZ_LVTAB_WORDS, enc_a inside)
template <bool TRAIN_UNC>
struct LzHeadLds {
    static constexpr int NFRAG = TRAIN_UNC ? LZ_FRAGS_ALL : LZ_FRAGS_INFER;
    static constexpr int WV = NFRAG * 64, TAB = WV + LZ_WV_FLOATS, FLOATS = TAB + LZ_LVTAB_WORDS;
};

// stage weights + tables into LDS (all threads of the workgroup; caller synchronises afterwards) and fill the context
template <bool TRAIN_UNC>
__device__ __forceinline__ void lz_head_stage(const LzHeadArgs& P, float* wl, uint32_t n_threads, int q, LzHeadCtx& hc) {
    using L = LzHeadLds<TRAIN_UNC>;
    const float4* src = reinterpret_cast<const float4*>(P.packed);
    float4* dst = reinterpret_cast<float4*>(wl);
    for (uint32_t i = threadIdx.x; i < (uint32_t)L::NFRAG * 16; i += n_threads) dst[i] = src[i];   // 16 B per lane per step, coalesced
    if (threadIdx.x < LZ_WV_FLOATS) wl[L::WV + threadIdx.x] = P.packed[LZ_FRAGS_ALL * 64 + threadIdx.x];
    // per-level table (indexed per lane in the gather) + the slice queue head of the stand-alone kernel
    int* tab = reinterpret_cast<int*>(wl + L::TAB);
    lz_level_table_fill(tab, P.offsets, P.scale, P.res);
    if (threadIdx.x < 32) wl[L::TAB + LZ_LVTAB_ENCA + threadIdx.x] = P.enc_a[threadIdx.x];
    hc.wl = wl;
    hc.tab = tab;
    hc.lenca = wl + L::TAB + LZ_LVTAB_ENCA;
    hc.emb[0] = P.emb[0]; hc.emb[1] = P.emb[1]; hc.emb[2] = P.emb[2];
    hc.bound = P.bound;
    hc.two_bound = 2.0f * P.bound;
    hc.has_eye = P.eye != nullptr;
    hc.eye_v = hc.has_eye ? P.eye[0] : 0.0f;
    hc.indq = P.ind_code ? P.ind_code[q] : 0.0f;
}

// SH(4) source that evaluates the polynomials from a direction fetched on demand (the stand-alone kernels: dirs are per sample)
template <typename DirFn>
struct LzShFromDir {
    DirFn dirfn;
    float o[16];
    __device__ __forceinline__ explicit LzShFromDir(DirFn f) : dirfn(f) {}
    __device__ __forceinline__ void prepare() {
        float dx, dy, dz;
        dirfn(dx, dy, dz);
        lz_sh_eval(dx, dy, dz, 4, o, nullptr, nullptr, nullptr);
    }
    __device__ __forceinline__ float comp_iq(int i, int q) const {    // component 4 i + q (i compile-time after unrolling)
        return q == 0 ? o[4 * i] : (q == 1 ? o[4 * i + 1] : (q == 2 ? o[4 * i + 2] : o[4 * i + 3]));
    }
    __device__ __forceinline__ float comp_qj(int q, int j) const {    // component 4 q + j
        return q == 0 ? o[j] : (q == 1 ? o[4 + j] : (q == 2 ? o[8 + j] : o[12 + j]));
    }
};
template <typename DirFn>
__device__ __forceinline__ LzShFromDir<DirFn> lz_sh_from_dir(DirFn f) { return LzShFromDir<DirFn>(f); }

// one slice: (px, py, pz) = this lane's sample position; shfn supplies SH(4) of its view direction when the colour net needs it
// (LzShFromDir: evaluated here from the direction, like the reference per sample; the fused frame kernel reads it per ray from LDS)
// FOLD (inference only): geo = Wg s2 feeds colour_net.0 with nothing but a linear map in between (network.py:304-306), so
// W_c0[:, geo] (Wg s2) = (W_c0[:, geo] Wg) s2: the host packs the 64 x 64 product into colour_net.0's geo columns (head.py: fold_geo) and
// the slice hands s2 to the colour net directly -- 64 of the 361 MFMAs per slice are never issued.  sigma (the VALU row of sigma_net.2 on
// s2) is unchanged bit for bit; rgb moves by the reassociation, a few 1e-7.
template <bool TRAIN_UNC, bool FOLD = false, bool IN_RANGE = false, bool YIELD = false, typename ShFn>
__device__ __forceinline__ void lz_head_slice(const LzHeadCtx& hc, int lane, float px, float py, float pz, ShFn shfn, LzHeadOut& out) {
    static_assert(!(TRAIN_UNC && FOLD), "the folded colour net is an inference arrangement");
    constexpr int WV = LzHeadLds<TRAIN_UNC>::WV;
    const int q = lane >> 4;
    // ---------------- gather: enc_x features f = 4i + q of sample s -> B operands (lz_head_gather.h) ----------------
    float encx[LZ_T][9];
    lz_head_gather<IN_RANGE, LZ_GATHER_PACK32, YIELD, true>(hc.emb, hc.tab, px, py, pz, q, hc.bound, hc.two_bound, encx[0]);
    __builtin_amdgcn_sched_barrier(0);  // the tile's 36 reads in flight at a time: bounds the register footprint

    // ---------------- audio channel attention: 36 -> 64 -> 32 ----------------
    float att[LZ_T][8];   // chained layout: [t*4 + r] = feature 16t + 4q + r
    {
        lz_f4 acc1[4][LZ_T];
#pragma unroll
        for (int ft = 0; ft < 4; ft++)
#pragma unroll
            for (int j = 0; j < LZ_T; j++) acc1[ft][j] = lz_f4{0, 0, 0, 0};
        lz_layer<LZ_L_A1, LZ_T>(hc.wl, lane, encx, acc1);
        float b2[LZ_T][16];
#pragma unroll
        for (int j = 0; j < LZ_T; j++)
#pragma unroll
            for (int ft = 0; ft < 4; ft++)
#pragma unroll
                for (int r = 0; r < 4; r++) b2[j][4 * ft + r] = lz_relu(acc1[ft][j][r]);
        lz_f4 acc2[2][LZ_T];
#pragma unroll
        for (int ft = 0; ft < 2; ft++)
#pragma unroll
            for (int j = 0; j < LZ_T; j++) acc2[ft][j] = lz_f4{0, 0, 0, 0};
        lz_layer<LZ_L_A2, LZ_T>(hc.wl, lane, b2, acc2);
#pragma unroll
        for (int j = 0; j < LZ_T; j++)
#pragma unroll
            for (int ft = 0; ft < 2; ft++)
#pragma unroll
                for (int r = 0; r < 4; r++) att[j][4 * ft + r] = acc2[ft][j][r];
    }
    // ambient_aud = || att ||_2 : sum of squares in the lane-partial order (att is its own weight row), then sqrt
    float ambaud[LZ_T];
#pragma unroll
    for (int j = 0; j < LZ_T; j++) {
        float acc = 0.0f;
#pragma unroll
        for (int k = 0; k < 8; k++) acc = lz_fmaf(att[j][k], att[j][k], acc);
        acc += __shfl_xor(acc, 16, 64);
        acc += __shfl_xor(acc, 32, 64);
        ambaud[j] = sqrtf(acc);
    }
    // ---------------- eye attention: 36 -> 16 -> 1, sigmoid ----------------
    float eyeatt[LZ_T];
#pragma unroll
    for (int j = 0; j < LZ_T; j++) eyeatt[j] = 0.0f;
    if (hc.has_eye) {
        lz_f4 acce[1][LZ_T];
#pragma unroll
        for (int j = 0; j < LZ_T; j++) acce[0][j] = lz_f4{0, 0, 0, 0};
        lz_layer<LZ_L_E1, LZ_T>(hc.wl, lane, encx, acce);
        float be[LZ_T][4];
#pragma unroll
        for (int j = 0; j < LZ_T; j++)
#pragma unroll
            for (int r = 0; r < 4; r++) be[j][r] = lz_relu(acce[0][j][r]);
#pragma unroll
        for (int j = 0; j < LZ_T; j++) eyeatt[j] = lz_sigmoidf(lz_lane_dot<1>(hc.wl + WV + LZ_WV_E2, q, be[j]));
    }
    // ---------------- uncertainty ----------------
    float uncv[LZ_T];
    if constexpr (TRAIN_UNC) {
        lz_f4 accu[2][LZ_T];
#pragma unroll
        for (int ft = 0; ft < 2; ft++)
#pragma unroll
            for (int j = 0; j < LZ_T; j++) accu[ft][j] = lz_f4{0, 0, 0, 0};
        lz_layer<LZ_L_U1, LZ_T>(hc.wl, lane, encx, accu);
        float bu[LZ_T][8];
#pragma unroll
        for (int j = 0; j < LZ_T; j++)
#pragma unroll
            for (int ft = 0; ft < 2; ft++)
#pragma unroll
                for (int r = 0; r < 4; r++) bu[j][4 * ft + r] = lz_relu(accu[ft][j][r]);
#pragma unroll
        for (int j = 0; j < LZ_T; j++) uncv[j] = lz_softplusf(lz_lane_dot<2>(hc.wl + WV + LZ_WV_U2, q, bu[j]));
    } else {
#pragma unroll
        for (int j = 0; j < LZ_T; j++) uncv[j] = lz_softplusf(0.0f);   // network.py:243-249, 278
    }
    // ---------------- sigma net: [enc_x 36 | enc_a * att 32 | eye * eye_att 1] -> 64 -> 64 -> 65 ----------------
    float geo[LZ_T][16];
    float sig_pre[LZ_T];      // row 0 of sigma_net.2, before the exp (evaluated with the colours' sigmoids at the end)
    {
        float b1[LZ_T][18];
#pragma unroll
        for (int j = 0; j < LZ_T; j++) {
#pragma unroll
            for (int i = 0; i < 9; i++) b1[j][i] = encx[j][i];
#pragma unroll
            for (int t = 0; t < 2; t++)
#pragma unroll
                for (int r = 0; r < 4; r++) b1[j][9 + 4 * t + r] = hc.lenca[16 * t + 4 * q + r] * att[j][4 * t + r];
            b1[j][17] = (hc.has_eye && q == 0) ? hc.eye_v * eyeatt[j] : 0.0f;
        }
        lz_f4 acc1[4][LZ_T];
#pragma unroll
        for (int ft = 0; ft < 4; ft++)
#pragma unroll
            for (int j = 0; j < LZ_T; j++) acc1[ft][j] = lz_f4{0, 0, 0, 0};
        lz_layer<LZ_L_S1, LZ_T>(hc.wl, lane, b1, acc1);
        float b2[LZ_T][16];
#pragma unroll
        for (int j = 0; j < LZ_T; j++)
#pragma unroll
            for (int ft = 0; ft < 4; ft++)
#pragma unroll
                for (int r = 0; r < 4; r++) b2[j][4 * ft + r] = lz_relu(acc1[ft][j][r]);
        lz_f4 acc2[4][LZ_T];
#pragma unroll
        for (int ft = 0; ft < 4; ft++)
#pragma unroll
            for (int j = 0; j < LZ_T; j++) acc2[ft][j] = lz_f4{0, 0, 0, 0};
        lz_layer<LZ_L_S2, LZ_T>(hc.wl, lane, b2, acc2);
        float b3[LZ_T][16];
#pragma unroll
        for (int j = 0; j < LZ_T; j++)
#pragma unroll
            for (int ft = 0; ft < 4; ft++)
#pragma unroll
                for (int r = 0; r < 4; r++) b3[j][4 * ft + r] = lz_relu(acc2[ft][j][r]);
        if constexpr (FOLD) {
#pragma unroll
            for (int j = 0; j < LZ_T; j++)
#pragma unroll
                for (int k = 0; k < 16; k++) geo[j][k] = b3[j][k];   // s2 itself: the geo projection lives in the packed colour_net.0
        } else {
            lz_f4 acc3[4][LZ_T];
#pragma unroll
            for (int ft = 0; ft < 4; ft++)
#pragma unroll
                for (int j = 0; j < LZ_T; j++) acc3[ft][j] = lz_f4{0, 0, 0, 0};
            lz_layer<LZ_L_S3, LZ_T>(hc.wl, lane, b3, acc3);
#pragma unroll
            for (int j = 0; j < LZ_T; j++)
#pragma unroll
                for (int ft = 0; ft < 4; ft++)
#pragma unroll
                    for (int r = 0; r < 4; r++) geo[j][4 * ft + r] = acc3[ft][j][r];   // geo_feat, no activation (network.py:304)
        }
#pragma unroll
        for (int j = 0; j < LZ_T; j++) sig_pre[j] = lz_lane_dot<4>(hc.wl + WV + LZ_WV_SIG, q, b3[j]);     // row 0 of sigma_net.2 on the VALU
    }
    // ---------------- colour net: [SH 16 | geo 64 | ind 4] -> 64 -> 3 ----------------
    float rgb[LZ_T][3];
    {
        float b1[LZ_T][21];
#pragma unroll
        for (int j = 0; j < LZ_T; j++) {
            // SH(4) of the view direction (constant per ray, recomputed per sample like the reference): this lane
            // keeps components 4i + q
            shfn.prepare();
#pragma unroll
            for (int i = 0; i < 4; i++) b1[j][i] = shfn.comp_iq(i, q);     // SH component 4 i + q
#pragma unroll
            for (int k = 0; k < 16; k++) b1[j][4 + k] = geo[j][k];
            b1[j][20] = hc.indq;
        }
        lz_f4 acc1[4][LZ_T];
#pragma unroll
        for (int ft = 0; ft < 4; ft++)
#pragma unroll
            for (int j = 0; j < LZ_T; j++) acc1[ft][j] = lz_f4{0, 0, 0, 0};
        lz_layer<LZ_L_C1, LZ_T>(hc.wl, lane, b1, acc1);
        float b2[LZ_T][16];
#pragma unroll
        for (int j = 0; j < LZ_T; j++)
#pragma unroll
            for (int ft = 0; ft < 4; ft++)
#pragma unroll
                for (int r = 0; r < 4; r++) b2[j][4 * ft + r] = lz_relu(acc1[ft][j][r]);
        // colour_net.1 (64 -> 3) on the VALU (network.py:275), then the four transcendentals of a sample -- sigma = exp(h0) (network.py:302)
        // and three sigmoids (:310) -- ONE per lane instead of four per lane: after the lane-partial sums every lane of a sample holds all
        // four pre-activations, so lane q evaluates the q-th (q < 3: sigmoid = 1 / (1 + exp(-x)), q == 3: exp) with the same lz_expf on
        // the same argument -- the same bits -- and the sample's lanes fetch each other's result (~120 vector instructions per slice less;
        // the exp is a 30-instruction polynomial because its bits are part of the parity contract)
#pragma unroll
        for (int j = 0; j < LZ_T; j++) {
            float d[3];
#pragma unroll
            for (int c = 0; c < 3; c++) d[c] = lz_lane_dot<4>(hc.wl + WV + LZ_WV_C2 + 64 * c, q, b2[j]);
            const float arg = q == 0 ? -d[0] : (q == 1 ? -d[1] : (q == 2 ? -d[2] : sig_pre[j]));
            const float e = lz_expf(arg);
            const float v = q == 3 ? e : (1.0f / (1.0f + e)) * 1.002f - 0.001f;
            const int s0 = lane & 15;
#pragma unroll
            for (int c = 0; c < 3; c++) rgb[j][c] = __shfl(v, s0 + 16 * c, 64);
            sig_pre[j] = __shfl(v, s0 + 48, 64);      // now sigma itself
        }
    }
    out.sigma = sig_pre[0];
    out.rgb[0] = rgb[0][0]; out.rgb[1] = rgb[0][1]; out.rgb[2] = rgb[0][2];
    out.ambaud = ambaud[0];
    out.eyeatt = eyeatt[0];
    out.unc = uncv[0];
}
#endif
