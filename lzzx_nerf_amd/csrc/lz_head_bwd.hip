// lz_head_bwd.hip -- backward of the fused triplane head for TRAINING (BASELINE cfg3): the data-gradient chain of
// NeRFNetwork.forward (nerf_triplane/network.py:252-311, training mode) in ONE kernel, activations recomputed from (xyz, dirs).
//
// torch autograd walks the reference's graph layer by layer: every Linear costs three library GEMMs over M ~ 10^7 rows and every
// activation round-trips through HBM (77 of 103 ms of a cfg3 step; 35 ms with this repo's per-layer lz_linear kernels).  Here a
// wave recomputes the forward of its 16 samples exactly as lz_k_triplane_head<true> does, then runs the chain backwards with the
// SAME packed weights: the A operand of dX = W^T dY is read from the forward fragments with a transposing address
// (4-way LDS bank conflict, hidden under the 32-cycle MFMA).  Lane (s, q) register r' of a backward D tile kt is the gradient of
// the value that lane supplied to k-step 4 kt + r' of the forward layer, so gradients stay in the layout of the activations they
// belong to: ReLU masks (kept as one bit per value) apply elementwise and a tile is directly the B operand of the next layer back.
//
// Outputs: d(enc_x) as three level-major [12, M] tensors (-> lz_grid_encode_backward per plane, grad_layout 3: coalesced reads), d(enc_a) [32] and d(ind_code) [4] (reduced over
// the samples in registers, one atomic per lane and workgroup at the end), and, for the weight gradients, the layer inputs X_l and
// the (ReLU-masked) output gradients G_l in row-major [M, *] buffers that lz_linear_grad_w reduces over M.  dW inside this kernel
// would need 379 accumulator tiles per wave or a second 96 KB LDS image next to the weights; the dump costs 3.6 KB per sample of
// HBM writes instead.
#include "lz_head_bwd_common.h"
#include "lz_head_gather.h"
#include "lzzx_sh_eval.h"

__global__ void __launch_bounds__(LZ_BWD_WG, LZ_BWD_WG / 256)
lz_k_triplane_head_backward(LzHeadBwdArgs A, const float* __restrict__ xyzs, const float* __restrict__ dirs, uint32_t M) {
    constexpr int NFRAG = LZ_FRAGS_ALL;
    constexpr int WV = NFRAG * 64, TAB = WV + LZ_WV_FLOATS;
    __shared__ float wl[TAB + LZ_LVTAB_WORDS];
    const LzHeadArgs& P = A.fwd;
    const lz_head_bwd_out& O = A.o;
    const uint32_t n_slices = (M + 15) / 16;
    const uint32_t slice_lo = (uint32_t)(((uint64_t)n_slices * blockIdx.x) / gridDim.x);
    const uint32_t slice_hi = (uint32_t)(((uint64_t)n_slices * (blockIdx.x + 1)) / gridDim.x);
    if (slice_lo >= slice_hi) return;
    {
        const float4* src = reinterpret_cast<const float4*>(P.packed);
        float4* dst = reinterpret_cast<float4*>(wl);
        for (int i = threadIdx.x; i < NFRAG * 16; i += LZ_BWD_WG) dst[i] = src[i];
        if (threadIdx.x < LZ_WV_FLOATS) wl[WV + threadIdx.x] = P.packed[LZ_FRAGS_ALL * 64 + threadIdx.x];
        lz_level_table_fill(reinterpret_cast<int*>(wl + TAB), P.offsets, P.scale, P.res);   // + the slice queue head
        if (threadIdx.x < 32) wl[TAB + LZ_LVTAB_ENCA + threadIdx.x] = P.enc_a[threadIdx.x];
    }
    __syncthreads();
    const int* tab = reinterpret_cast<const int*>(wl + TAB);
    const float* lenca = wl + TAB + LZ_LVTAB_ENCA;
    const float* wv = wl + WV;

    const int lane = threadIdx.x & 63;
    const int s = lane & 15, q = lane >> 4;
    const float two_bound = 2.0f * P.bound;
    const bool has_eye = P.eye != nullptr;
    const float eye_v = has_eye ? P.eye[0] : 0.0f;
    const float indq = P.ind_code ? P.ind_code[q] : 0.0f;
    int* queue = reinterpret_cast<int*>(wl + TAB) + LZ_LVTAB_QUEUE;
    // The two waves that share a SIMD would otherwise run in lockstep (same code, same start): both in their matrix phases, then both in
    // their VALU / store phases.  Delaying the second one by about half a slice lets one wave's MFMAs overlap the other's VALU work.
    if ((threadIdx.x >> 6) >= 4) {
        __builtin_amdgcn_s_sleep(127);
        __builtin_amdgcn_s_sleep(127);
        __builtin_amdgcn_s_sleep(127);
    }
    float acc_enca[8], acc_ind = 0.0f;   // d(enc_a)[16 t + 4 q + r], d(ind_code)[q], summed over this lane's samples
#pragma unroll
    for (int k = 0; k < 8; k++) acc_enca[k] = 0.0f;
    // weight gradients of the skinny output layers, per lane: feature 16 t + 4 q + r of this lane's samples
    float acc_e2[4], acc_u2[8], acc_c2[3][16];
#pragma unroll
    for (int k = 0; k < 4; k++) acc_e2[k] = 0.0f;
#pragma unroll
    for (int k = 0; k < 8; k++) acc_u2[k] = 0.0f;
#pragma unroll
    for (int c = 0; c < 3; c++)
#pragma unroll
        for (int k = 0; k < 16; k++) acc_c2[c][k] = 0.0f;

    for (;;) {
        int slice = 0;
        if (lane == 0) slice = atomicAdd(queue, 1);
        slice = __builtin_amdgcn_readfirstlane(slice);
        if (slice_lo + (uint32_t)slice >= slice_hi) break;
        const uint32_t base = (slice_lo + (uint32_t)slice) * 16;
        const bool valid = base + s < M;
        const uint32_t m = valid ? base + s : M - 1;   // clamped rows are computed, never stored or accumulated
        const size_t row = m;
        float* rb = lz_blk(O.rec, slice_lo + (uint32_t)slice, LZ_BWD_REC, s);   // this sample's slot in the slice block of the records
        float* dencq = O.denc + (size_t)q * M + row;

        // =============================== forward (as lz_k_triplane_head<true>) ===============================
        float encx[9];
        lz_head_gather(P.emb, tab, xyzs[(size_t)m * 3], xyzs[(size_t)m * 3 + 1], xyzs[(size_t)m * 3 + 2], q, P.bound, two_bound, encx);
        // every per-sample input is loaded here, before the first dump store of the slice: a load issued after stores can only be waited
        // for once those stores have been acknowledged (one counter, in order), which under this kernel's write stream takes microseconds
        const float g_sig = A.g_sigma[row], g_aa = A.g_amb_aud[row], g_ae = A.g_amb_eye ? A.g_amb_eye[row] : 0.0f, g_un = A.g_unc[row];
        const float g_r0 = A.g_rgb[row * 3], g_r1 = A.g_rgb[row * 3 + 1], g_r2 = A.g_rgb[row * 3 + 2];
        const float dir0 = dirs[row * 3], dir1 = dirs[row * 3 + 1], dir2 = dirs[row * 3 + 2];
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
            if (valid) lz_dump_chained<4>(rb, q, LZ_BWD_X_A1, a1[0]);
            lz_f4 acc2[2][1] = {{lz_f4{0, 0, 0, 0}}, {lz_f4{0, 0, 0, 0}}};
            lz_layer<LZ_L_A2, 1>(wl, lane, a1, acc2);
#pragma unroll
            for (int ft = 0; ft < 2; ft++)
#pragma unroll
                for (int r = 0; r < 4; r++) att[4 * ft + r] = acc2[ft][0][r];
        }
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
        float e1[4] = {0.0f, 0.0f, 0.0f, 0.0f};
        if (has_eye) {
            lz_f4 acce[1][1] = {{lz_f4{0, 0, 0, 0}}};
            lz_layer<LZ_L_E1, 1>(wl, lane, bx, acce);
#pragma unroll
            for (int r = 0; r < 4; r++) e1[r] = lz_relu(acce[0][0][r]);
            mk_e1 = lz_mask_pos(e1);
            eyeatt = lz_sigmoidf(lz_lane_dot<1>(wv + LZ_WV_E2, q, e1));
        }
        // uncertainty
        float du;   // unc = softplus(u): d loss / d u
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
            du = g_un * lz_sigmoidf(lz_lane_dot<2>(wv + LZ_WV_U2, q, u1));
            if (valid) {
#pragma unroll
                for (int k = 0; k < 8; k++) acc_u2[k] = lz_fmaf(du, u1[k], acc_u2[k]);
            }
        }
        // sigma net
        float sigma;
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
                for (int r = 0; r < 4; r++) encw[4 * t + r] = lenca[16 * t + 4 * q + r] * att[4 * t + r];
#pragma unroll
            for (int k = 0; k < 8; k++) b1[0][9 + k] = encw[k];
            b1[0][17] = (has_eye && q == 0) ? eye_v * eyeatt : 0.0f;
            if (valid) {   // sigma_net.0 input [enc_x 36 | enc_a * att 32 | eye * eye_att 1], leading dimension 72
#pragma unroll
                for (int i = 0; i < 9; i++) rb[lz_tcol(LZ_BWD_X_SIG0 + 4 * i + q)] = encx[i];
                lz_dump_chained<2>(rb, q, LZ_BWD_X_SIG0 + 36, encw);
                if (q == 0) rb[lz_tcol(LZ_BWD_X_SIG0 + 68 + q)] = b1[0][17];
            }
            lz_f4 acc1[4][1] = {{lz_f4{0, 0, 0, 0}}, {lz_f4{0, 0, 0, 0}}, {lz_f4{0, 0, 0, 0}}, {lz_f4{0, 0, 0, 0}}};
            lz_layer<LZ_L_S1, 1>(wl, lane, b1, acc1);
            float s1[1][16];
#pragma unroll
            for (int ft = 0; ft < 4; ft++)
#pragma unroll
                for (int r = 0; r < 4; r++) s1[0][4 * ft + r] = lz_relu(acc1[ft][0][r]);
            mk_s1 = lz_mask_pos(s1[0]);
            if (valid) lz_dump_chained<4>(rb, q, LZ_BWD_X_S1, s1[0]);
            lz_f4 acc2[4][1] = {{lz_f4{0, 0, 0, 0}}, {lz_f4{0, 0, 0, 0}}, {lz_f4{0, 0, 0, 0}}, {lz_f4{0, 0, 0, 0}}};
            lz_layer<LZ_L_S2, 1>(wl, lane, s1, acc2);
            float s2[1][16];
#pragma unroll
            for (int ft = 0; ft < 4; ft++)
#pragma unroll
                for (int r = 0; r < 4; r++) s2[0][4 * ft + r] = lz_relu(acc2[ft][0][r]);
            mk_s2 = lz_mask_pos(s2[0]);
            if (valid) lz_dump_chained<4>(rb, q, LZ_BWD_X_S2C, s2[0]);
            lz_f4 acc3[4][1] = {{lz_f4{0, 0, 0, 0}}, {lz_f4{0, 0, 0, 0}}, {lz_f4{0, 0, 0, 0}}, {lz_f4{0, 0, 0, 0}}};
            lz_layer<LZ_L_S3, 1>(wl, lane, s2, acc3);
#pragma unroll
            for (int ft = 0; ft < 4; ft++)
#pragma unroll
                for (int r = 0; r < 4; r++) geo[0][4 * ft + r] = acc3[ft][0][r];
            sigma = lz_expf(lz_lane_dot<4>(wv + LZ_WV_SIG, q, s2[0]));
        }
        // colour net
        float dc[3];   // colour head: rgb = sigmoid(c) * 1.002 - 0.001
        uint32_t mk_c1;
        {
            float o[16];
            lz_sh_eval(dir0, dir1, dir2, 4, o, nullptr, nullptr, nullptr);
            float b1[1][21];
#pragma unroll
            for (int i = 0; i < 4; i++) b1[0][i] = q == 0 ? o[4 * i] : (q == 1 ? o[4 * i + 1] : (q == 2 ? o[4 * i + 2] : o[4 * i + 3]));
#pragma unroll
            for (int k = 0; k < 16; k++) b1[0][4 + k] = geo[0][k];
            b1[0][20] = indq;
            if (valid) {   // colour_net.0 input [SH 16 | geo 64 | ind 4]
#pragma unroll
                for (int i = 0; i < 4; i++) rb[lz_tcol(LZ_BWD_X_S2C + 64 + 4 * i + q)] = b1[0][i];
                rb[lz_tcol(LZ_BWD_X_S2C + 80 + q)] = indq;   // geo = s2 . Wg^T is not stored: its weight gradient is finished from sum G_c1^T s2
            }
            lz_f4 acc1[4][1] = {{lz_f4{0, 0, 0, 0}}, {lz_f4{0, 0, 0, 0}}, {lz_f4{0, 0, 0, 0}}, {lz_f4{0, 0, 0, 0}}};
            lz_layer<LZ_L_C1, 1>(wl, lane, b1, acc1);
            float c1[16];
#pragma unroll
            for (int ft = 0; ft < 4; ft++)
#pragma unroll
                for (int r = 0; r < 4; r++) c1[4 * ft + r] = lz_relu(acc1[ft][0][r]);
            mk_c1 = lz_mask_pos(c1);
#pragma unroll
            for (int c = 0; c < 3; c++) {
                const float sg = lz_sigmoidf(lz_lane_dot<4>(wv + LZ_WV_C2 + 64 * c, q, c1));
                dc[c] = (c == 0 ? g_r0 : (c == 1 ? g_r1 : g_r2)) * 1.002f * sg * (1.0f - sg);
                if (valid) {
#pragma unroll
                    for (int k = 0; k < 16; k++) acc_c2[c][k] = lz_fmaf(dc[c], c1[k], acc_c2[c][k]);
                }
            }
        }

        // =============================== backward ===============================
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
                    dc1[k] = lz_mask_keep(mk_c1, k, v);
                }
            if (valid) lz_dump_chained<4>(rb, q, LZ_BWD_G_C1H, dc1);
            float dxc[21];
            lz_layer_bwd<LZ_L_C1>(wl, lane, dc1, dxc);
#pragma unroll
            for (int k = 0; k < 16; k++) dgeo[k] = dxc[4 + k];
            dind = dxc[20];
        }
        if (valid) acc_ind += dind;
        // sigma net
        const float dh0 = g_sig * sigma;
        if (valid) {   // sigma_net.2 output gradient [d geo 64 | d h0], leading dimension 68 (16-byte rows: dwordx4 stores)
            if (q == 0) rb[lz_tcol(LZ_BWD_G_C1H + 64 + q)] = dh0;   // d geo = G_c1 . W_c0[:, geo] is not stored either
        }
        float dencx[9], dencw[8], determ;
        {
            float ds2[16];
            lz_layer_bwd<LZ_L_S3>(wl, lane, dgeo, ds2);
#pragma unroll
            for (int t = 0; t < 4; t++)
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    const int k = 4 * t + r;
                    const float v = lz_fmaf(wv[LZ_WV_SIG + 16 * t + 4 * q + r], dh0, ds2[k]);
                    ds2[k] = lz_mask_keep(mk_s2, k, v);
                }
            if (valid) lz_dump_chained<4>(rb, q, LZ_BWD_G_S2, ds2);
            float ds1[16];
            lz_layer_bwd<LZ_L_S2>(wl, lane, ds2, ds1);
#pragma unroll
            for (int k = 0; k < 16; k++) ds1[k] = lz_mask_keep(mk_s1, k, ds1[k]);
            if (valid) lz_dump_chained<4>(rb, q, LZ_BWD_G_S1, ds1);
            float dxs[18];
            lz_layer_bwd<LZ_L_S1>(wl, lane, ds1, dxs);
#pragma unroll
            for (int i = 0; i < 9; i++) dencx[i] = dxs[i];
#pragma unroll
            for (int k = 0; k < 8; k++) dencw[k] = dxs[9 + k];
            determ = dxs[17];   // meaningful on lanes q == 0
        }
        // enc_w = enc_a * att ; ambient_aud = ||att||
        float datt[8];
        {
            const float inv = norm > 0.0f ? g_aa / norm : 0.0f;
#pragma unroll
            for (int t = 0; t < 2; t++)
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    const int k = 4 * t + r;
                    datt[k] = lz_fmaf(lenca[16 * t + 4 * q + r], dencw[k], inv * att[k]);
                    if (valid) acc_enca[k] = lz_fmaf(att[k], dencw[k], acc_enca[k]);
                }
            if (valid) lz_dump_chained<2>(rb, q, LZ_BWD_G_ATT, datt);
        }
        // eye attention: eterm = eye * eye_att (lane q == 0 holds its gradient), ambient_eye = |eye_att| = eye_att
        if (has_eye) {
            const float det0 = __shfl(determ, s, 64);   // from lane (s, q = 0)
            const float deye = lz_fmaf(eye_v, det0, g_ae);
            const float de2 = deye * eyeatt * (1.0f - eyeatt);
            if (valid) {
#pragma unroll
                for (int r = 0; r < 4; r++) acc_e2[r] = lz_fmaf(de2, e1[r], acc_e2[r]);
            }
            float de1[4];
#pragma unroll
            for (int r = 0; r < 4; r++) de1[r] = lz_mask_keep(mk_e1, r, wv[LZ_WV_E2 + 4 * q + r] * de2);
            if (valid) lz_dump_chained<1>(rb, q, LZ_BWD_G_X + 64, de1);
            float dxe[9];
            lz_layer_bwd<LZ_L_E1>(wl, lane, de1, dxe);
#pragma unroll
            for (int i = 0; i < 9; i++) dencx[i] += dxe[i];
        } else if (valid) {   // the stacked reduction over G_x reads these columns: no eye input, no gradient
            const float zero[4] = {0.0f, 0.0f, 0.0f, 0.0f};
            lz_dump_chained<1>(rb, q, LZ_BWD_G_X + 64, zero);
        }
        // audio channel attention
        {
            float da1[16];
            lz_layer_bwd<LZ_L_A2>(wl, lane, datt, da1);
#pragma unroll
            for (int k = 0; k < 16; k++) da1[k] = lz_mask_keep(mk_a1, k, da1[k]);
            if (valid) lz_dump_chained<4>(rb, q, LZ_BWD_G_X, da1);
            float dxa[9];
            lz_layer_bwd<LZ_L_A1>(wl, lane, da1, dxa);
#pragma unroll
            for (int i = 0; i < 9; i++) dencx[i] += dxa[i];
        }
        // uncertainty: unc = softplus(u); its input is detached (network.py:241-249): weight gradients only
        {
            float du1[8];
#pragma unroll
            for (int t = 0; t < 2; t++)
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    const int k = 4 * t + r;
                    du1[k] = lz_mask_keep(mk_u1, k, wv[LZ_WV_U2 + 16 * t + 4 * q + r] * du);
                }
            if (valid) lz_dump_chained<2>(rb, q, LZ_BWD_G_X + 80, du1);
        }
        // d(enc_x): feature 4 i + q = plane i / 3, level 4 (i % 3) + q
        if (valid) {
#pragma unroll
            for (int i = 0; i < 9; i++) dencq[(size_t)(4 * i) * M] = dencx[i];   // [3 planes][12 levels][M], level-major; feature 4 i + q
        }
    }
    // Per-lane sums -> sum over the 16 sample lanes of every q group -> sum over the workgroup's waves in LDS (the weight image is no
    // longer needed) -> one atomic per value and workgroup.  Slots: d(enc_a) 32 | d(ind) 4 | dW_e2 16 | dW_u2 32 | dW_c2 192.
    constexpr int NRED = 32 + 4 + 16 + 32 + 192;
    __syncthreads();
    float* red = wl;                       // [waves][NRED]
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
#pragma unroll
    for (int c = 0; c < 3; c++)
#pragma unroll
        for (int k = 0; k < 16; k++) put(acc_c2[c][k], 84 + 64 * c + 16 * (k >> 2) + 4 * q + (k & 3));
    __syncthreads();
    if (threadIdx.x < NRED) {
        float v = 0.0f;
        for (int w = 0; w < LZ_BWD_WG / 64; w++) v += red[w * NRED + threadIdx.x];
        const int i = threadIdx.x;
        if (v != 0.0f) atomicAdd(O.small + threadIdx.x, v);   // slot order = the layout of lz_head_bwd_out.small
    }
}

extern "C" int lz_triplane_head_backward(const lz_head_params* p, const float* xyzs, const float* dirs, uint32_t M, const float* g_sigma,
                                         const float* g_rgb, const float* g_amb_aud, const float* g_amb_eye, const float* g_unc,
                                         const lz_head_bwd_out* out, lz_stream_t stream) {
    LZ_REQUIRE(p && xyzs && dirs && g_sigma && g_rgb && g_amb_aud && g_unc && out, LZ_ERR_BAD_ARGUMENT, "triplane_head_backward: null tensor");
    LZ_REQUIRE(p->emb_xy && p->emb_yz && p->emb_xz && p->offsets && p->packed && p->enc_a, LZ_ERR_BAD_ARGUMENT,
               "triplane_head_backward: incomplete lz_head_params");
    LZ_REQUIRE(p->precision == 0 && !p->testing, LZ_ERR_UNSUPPORTED, "triplane_head_backward: f32 training mode only");
    const lz_head_bwd_out& o = *out;
    LZ_REQUIRE(o.denc && o.small && o.rec, LZ_ERR_BAD_ARGUMENT, "triplane_head_backward: incomplete lz_head_bwd_out");
    LZ_REQUIRE(((uintptr_t)o.rec & 15u) == 0, LZ_ERR_BAD_ARGUMENT, "triplane_head_backward: rec must be 16-byte aligned");
    if (M == 0) return LZ_OK;
    LzHeadBwdArgs a;
    a.fwd.emb[0] = p->emb_xy; a.fwd.emb[1] = p->emb_yz; a.fwd.emb[2] = p->emb_xz;
    a.fwd.offsets = p->offsets; a.fwd.packed = reinterpret_cast<const float*>(p->packed); a.fwd.enc_a = p->enc_a;
    a.fwd.ind_code = p->ind_code; a.fwd.eye = p->eye; a.fwd.bound = p->bound; a.fwd.testing = 0;
    for (int l = 0; l < 12; l++) {
        const float sc = exp2f((float)l * p->S) * (float)p->H - 1.0f;
        a.fwd.scale[l] = sc;
        a.fwd.res[l] = (uint32_t)ceilf(sc) + 1u;
    }
    a.g_sigma = g_sigma; a.g_rgb = g_rgb; a.g_amb_aud = g_amb_aud; a.g_amb_eye = g_amb_eye; a.g_unc = g_unc;
    a.o = o;
    a.wb16 = nullptr;
    const int n_cu = lz_cu_count();   // of the current device, per call (cached per device)
    const uint32_t slices = lz_div_up(M, 16);
    const uint32_t want = lz_div_up(slices, LZ_BWD_WG / 64);
    const uint32_t grid = want < (uint32_t)n_cu ? want : (uint32_t)n_cu;
    hipLaunchKernelGGL(lz_k_triplane_head_backward, dim3(grid), dim3(LZ_BWD_WG), 0, lz_st(stream), a, xyzs, dirs, M);
    LZ_CHECK_LAUNCH("triplane_head_backward");
    return LZ_OK;
}
