// lz_head_f16.hip -- the fused triplane head with the MLP on the f16 matrix cores (v_mfma_f32_16x16x32_f16).
//
// What it mirrors: the reference renders under torch.cuda.amp.autocast when opt.fp16 is set (TrainerUtil / renderer
// call sites; SURVEY 8a' note 17).  Under autocast every bias-free nn.Linear of NeRFNetwork.forward (network.py:252-311)
// casts its input and weight to half, accumulates in f32 and returns HALF; ReLU, sigmoid and the products enc_a * att run
// in half (computed in f32, rounded to half); the triplane tables stay f32 (C = 1 is odd, grid.py:38); SH is f32
// (custom_fwd cast_inputs=float32); exp (trunc_exp) and norm run in f32.  This kernel reproduces that rounding sequence:
// f32 gathers (bit-identical to the f32 kernel) -> half B operands -> MFMA with f32 accumulate -> half between layers.
// The GEMM's internal summation order is the matrix core's, so parity with the CPU checker (oracle/head.py,
// head_forward_fp16) is to half rounding, not bit-exact -- as between two runs of cuBLAS with different algorithms.
//
// MI355X design
//   * 16x16x32 f16 MFMA (16 cycles, K = 32): A = weights (16 features x 32 k), B = activations (32 k x 16 samples); lane
//     (s = l & 15, q = l >> 4) of B holds k slots 8q..8q+7, of a D tile holds features 16t + 4q + r.  Two D tiles
//     (2t', 2t'+1) of a layer are therefore exactly one B operand of the next layer (slot j <-> tile j >> 2, reg j & 3):
//     activations never leave registers or cross lanes; only the weights are permuted, once, at pack time.
//   * 59 MFMAs per 16-sample slice instead of 405 f32 ones: the kernel is bound by the 144 table gathers per sample
//     (texture-address rate) and the VALU work around them, not by the matrix pipe.
//   * Same workgroup geometry and work distribution as lz_k_triplane_head (1024 threads, contiguous slice shares, LDS
//     slice queue); the packed weights are 59 KB of LDS.
// Round 5: the INFERENCE kernels (this file's lz_k_triplane_head_f16w, the fused frame) run the same rounding sequence on
// v_mfma_f32_32x32x16_f16 over 32-sample slices (lz_head_f16w_slice.h: 60 MFMAs per 32 samples instead of 118, half the issue slots);
// the 16x16x32 arrangement below stays as the packer of the recording training forward (lz_head_rec16.hip).
#include "lz_head_f16_slice.h"
#include "lz_head_f16w_slice.h"

extern "C" uint32_t lz_head_packed_size_f16(void) { return (uint32_t)H_FRAGS * 64u * 16u; }

// ---- weight packing: fragment (layer, k-step, feature tile), lane (m = l & 15, kg = l >> 4) holds W[16 ft + m][k(ks, kg, j)], j < 8

struct LzPack16Args {
    const float* w[H_COUNT];
    int nout[H_COUNT];
    int ld[H_COUNT];
    int has_eye, has_ind;
};

__global__ void __launch_bounds__(256) lz_k_head_pack_f16(LzPack16Args a, _Float16* __restrict__ packed) {
    const int gid = blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= H_FRAGS * 64) return;
    const int frag = gid >> 6, lane = gid & 63;
    int layer = 0, fb = 0;
    for (int i = 0; i < H_COUNT; i++) {
        const int n = H_KS[i] * H_NT[i];
        if (frag < fb + n) { layer = i; break; }
        fb += n;
    }
    const int local = frag - fb;
    const int ks = local / H_NT[layer], ft = local - ks * H_NT[layer];
    const int row = 16 * ft + (lane & 15), kg = lane >> 4;
    // output rows.  sigma_net.2: geo rows first, the sigma row ALONE in tile 4 -- at row 12, i.e. register 0 of lane group q = 3; colour_net.1: channel
    // c at row 4 c, i.e. register 0 of lane group q = c.  The four transcendentals of a sample (sigma = exp, three colour sigmoids) then sit one
    // per lane group and are evaluated by ONE instruction sequence (lz_head16_slice_rows), not four.
    int srow;
    if (layer == H_S3) srow = row < 64 ? row + 1 : (row == 76 ? 0 : -1);
    else if (layer == H_C2) srow = ((row & 3) == 0 && row < 12) ? (row >> 2) : -1;
    else srow = row < a.nout[layer] ? row : -1;
    for (int j = 0; j < 8; j++) {
        int kf;
        switch (layer) {
            case H_A1: case H_E1: kf = h_encx(ks, kg, j); break;
            case H_A2: case H_S2: case H_S3: case H_C2: kf = h_chain(ks, kg, j, 64); break;
            case H_E2: kf = h_chain(ks, kg, j, 16); break;
            case H_S1:
                if (ks < 2) {
                    kf = h_encx(ks, kg, j);
                    if (ks == 1 && kg == 0 && j == 1) kf = a.has_eye ? 68 : -1;      // eye * eye_att rides in a spare slot
                } else {
                    const int f = h_chain(0, kg, j, 32);
                    kf = f >= 0 ? 36 + f : -1;
                }
                break;
            default:  // H_C1: k-step 0 = [SH 4 kg + j | ind on lane group 0], k-steps 1, 2 = geo
                if (ks == 0) kf = j < 4 ? 4 * kg + j : ((kg == 0 && a.has_ind) ? 80 + (j - 4) : -1);
                else { const int f = h_chain(ks - 1, kg, j, 64); kf = f >= 0 ? 16 + f : -1; }
                break;
        }
        float v = 0.0f;
        if (kf >= 0 && srow >= 0 && a.w[layer]) v = a.w[layer][(size_t)srow * a.ld[layer] + kf];
        packed[(size_t)gid * 8 + j] = (_Float16)v;   // autocast: weight.half(), round to nearest even
    }
}

extern "C" int lz_head_pack_weights_f16(const float* aud0, const float* aud1, const float* eye0, const float* eye1, const float* sig0,
                                        const float* sig1, const float* sig2, const float* col0, const float* col1, int has_eye,
                                        int has_ind, void* packed, lz_stream_t stream) {
    LZ_REQUIRE(aud0 && aud1 && sig0 && sig1 && sig2 && col0 && col1 && packed, LZ_ERR_BAD_ARGUMENT, "head_pack_weights_f16: null weight");
    LZ_REQUIRE(!has_eye || (eye0 && eye1), LZ_ERR_BAD_ARGUMENT, "head_pack_weights_f16: eye weights required when has_eye");
    LzPack16Args a;
    const float* w[H_COUNT] = {aud0, aud1, eye0, eye1, sig0, sig1, sig2, col0, col1};
    const int nout[H_COUNT] = {64, 32, 16, 1, 64, 64, 65, 64, 3};
    const int ld[H_COUNT] = {36, 64, 36, 16, 68 + (has_eye ? 1 : 0), 64, 64, 80 + (has_ind ? 4 : 0), 64};
    for (int i = 0; i < H_COUNT; i++) { a.w[i] = w[i]; a.nout[i] = nout[i]; a.ld[i] = ld[i]; }
    a.has_eye = has_eye; a.has_ind = has_ind;
    hipLaunchKernelGGL(lz_k_head_pack_f16, dim3(lz_div_up((uint64_t)H_FRAGS * 64, 256)), dim3(256), 0, lz_st(stream), a,
                       reinterpret_cast<_Float16*>(packed));
    LZ_CHECK_LAUNCH("head_pack_weights_f16");
    return LZ_OK;
}
// ---- weight packing for the 32-sample slice: fragment (layer, k-step, 32-row tile), lane (r = l & 31, h = l >> 5) holds
// W[row(32 ft + r)][k(ks, h, j)], j < 8 (lz_head_f16w_slice.h: w_chain / w_encx)
extern "C" uint32_t lz_head_packed_size_f16w(void) { return (uint32_t)W_FRAGS * 64u * 16u; }

__global__ void __launch_bounds__(256) lz_k_head_pack_f16w(LzPack16Args a, _Float16* __restrict__ packed) {
    const int gid = blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= W_FRAGS * 64) return;
    const int frag = gid >> 6, lane = gid & 63;
    int layer = 0, fb = 0;
    for (int i = 0; i < W_COUNT; i++) {
        const int n = W_KS[i] * W_NT[i];
        if (frag < fb + n) { layer = i; break; }
        fb += n;
    }
    const int local = frag - fb;
    const int ks = local / W_NT[layer], ft = local - ks * W_NT[layer];
    const int row = 32 * ft + (lane & 31), h = lane >> 5;
    // output rows.  sigma_net.2: the 64 geo rows fill tiles 0 and 1, the sigma row sits ALONE in tile 2 at row 4 = register 0 of lane half 1;
    // colour_net.1: channels 0, 1, 2 at rows 0, 4, 1 = (half 0, register 0), (half 1, register 0), (half 0, register 1): the sample's four
    // transcendentals are two instruction sequences on its two lanes (lz_head16w_slice)
    int srow;
    if (layer == W_S3) srow = row < 64 ? row + 1 : (row == 68 ? 0 : -1);
    else if (layer == W_C2) srow = row == 0 ? 0 : (row == 4 ? 1 : (row == 1 ? 2 : -1));
    else srow = row < a.nout[layer] ? row : -1;
    for (int j = 0; j < 8; j++) {
        int kf;
        switch (layer) {
            case W_A1: case W_E1: kf = w_encx(ks, h, j); break;
            case W_A2: case W_S2: case W_S3: case W_C2: kf = w_chain(ks, h, j, 64); break;
            case W_E2: kf = w_chain(ks, h, j, 16); break;
            case W_S1:
                if (ks < 3) {
                    kf = w_encx(ks, h, j);
                    if (8 * ks + j == W_EYE_SLOT && h == 0) kf = a.has_eye ? 68 : -1;      // eye * eye_att rides in a spare slot
                } else {
                    const int f = w_chain(ks - 3, h, j, 32);
                    kf = f >= 0 ? 36 + f : -1;
                }
                break;
            default:  // W_C1: k-step 0 = SH 8 h + j, k-steps 1 .. 4 = geo, k-step 5 = ind_code on lane half 0
                if (ks == 0) kf = 8 * h + j;
                else if (ks < 5) { const int f = w_chain(ks - 1, h, j, 64); kf = f >= 0 ? 16 + f : -1; }
                else kf = (h == 0 && j < 4 && a.has_ind) ? 80 + j : -1;
                break;
        }
        float v = 0.0f;
        if (kf >= 0 && srow >= 0 && a.w[layer]) v = a.w[layer][(size_t)srow * a.ld[layer] + kf];
        packed[(size_t)gid * 8 + j] = (_Float16)v;   // autocast: weight.half(), round to nearest even
    }
}

extern "C" int lz_head_pack_weights_f16w(const float* aud0, const float* aud1, const float* eye0, const float* eye1, const float* sig0,
                                         const float* sig1, const float* sig2, const float* col0, const float* col1, int has_eye,
                                         int has_ind, void* packed, lz_stream_t stream) {
    LZ_REQUIRE(aud0 && aud1 && sig0 && sig1 && sig2 && col0 && col1 && packed, LZ_ERR_BAD_ARGUMENT, "head_pack_weights_f16w: null weight");
    LZ_REQUIRE(!has_eye || (eye0 && eye1), LZ_ERR_BAD_ARGUMENT, "head_pack_weights_f16w: eye weights required when has_eye");
    LzPack16Args a;
    const float* w[W_COUNT] = {aud0, aud1, eye0, eye1, sig0, sig1, sig2, col0, col1};
    const int nout[W_COUNT] = {64, 32, 16, 1, 64, 64, 65, 64, 3};
    const int ld[W_COUNT] = {36, 64, 36, 16, 68 + (has_eye ? 1 : 0), 64, 64, 80 + (has_ind ? 4 : 0), 64};
    for (int i = 0; i < W_COUNT; i++) { a.w[i] = w[i]; a.nout[i] = nout[i]; a.ld[i] = ld[i]; }
    a.has_eye = has_eye; a.has_ind = has_ind;
    hipLaunchKernelGGL(lz_k_head_pack_f16w, dim3(lz_div_up((uint64_t)W_FRAGS * 64, 256)), dim3(256), 0, lz_st(stream), a,
                       reinterpret_cast<_Float16*>(packed));
    LZ_CHECK_LAUNCH("head_pack_weights_f16w");
    return LZ_OK;
}
// ---- the kernel ---------------------------------------------------------------------------------------
#define H_WG 1024

__global__ void __launch_bounds__(H_WG, H_WG / 256)
lz_k_triplane_head_f16w(LzHead16Args P, const float* __restrict__ xyzs, const float* __restrict__ dirs, uint32_t M,
                        const int* __restrict__ count, float* __restrict__ sigmas, float* __restrict__ rgbs,
                        float* __restrict__ amb_aud, float* __restrict__ amb_eye, float* __restrict__ unc_out) {
    __shared__ lz_h8 wl[LZ_HEAD16W_LDS_H8];   // packed A fragments, then the level table (enc_a inside)
    uint32_t Meff = M;
    if (count) {
        const int c = *count;
        Meff = c < 0 ? 0u : ((uint32_t)c < M ? (uint32_t)c : M);
    }
    const uint32_t n_slices = (Meff + 31) / 32;
    const uint32_t slice_lo = (uint32_t)(((uint64_t)n_slices * blockIdx.x) / gridDim.x);
    const uint32_t slice_hi = (uint32_t)(((uint64_t)n_slices * (blockIdx.x + 1)) / gridDim.x);
    if (slice_lo >= slice_hi) return;
    LzHead16Ctx ctx;
    lz_head16w_stage(P, wl, H_WG, ctx);
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const int s = lane & 31, h = lane >> 5;
    int* queue = reinterpret_cast<int*>(wl + W_FRAGS * 64) + LZ_LVTAB_QUEUE;
    for (;;) {
        int slice = 0;
        if (lane == 0) slice = atomicAdd(queue, 1);
        slice = __builtin_amdgcn_readfirstlane(slice);
        if (slice_lo + (uint32_t)slice >= slice_hi) break;
        const uint32_t base = (slice_lo + (uint32_t)slice) * 32;
        uint32_t m = base + s;
        if (m >= Meff) m = Meff - 1;  // clamp: computed, never stored
        const float px = xyzs[(size_t)m * 3], py = xyzs[(size_t)m * 3 + 1], pz = xyzs[(size_t)m * 3 + 2];
        LzHead16wOut o;
        lz_head16w_slice(ctx, lane, px, py, pz,
                         lz_sh_from_dir([&](float& dx, float& dy, float& dz) { dx = dirs[(size_t)m * 3]; dy = dirs[(size_t)m * 3 + 1]; dz = dirs[(size_t)m * 3 + 2]; }), o);
        // ---------------- store: lane half 0 owns rgb[0], rgb[2] and the per-sample scalars, lane half 1 rgb[1] and sigma ----------------
        if (base + s < Meff) {
            if (h == 0) {
                rgbs[(size_t)m * 3] = o.a; rgbs[(size_t)m * 3 + 2] = o.b;
                amb_aud[m] = o.ambaud;
                if (amb_eye) amb_eye[m] = o.eyeatt;
                unc_out[m] = o.unc;
            } else {
                rgbs[(size_t)m * 3 + 1] = o.a;
                sigmas[m] = o.b;
            }
        }
    }
}

// called by lz_triplane_head_forward when p->precision == 1
int lz_head_forward_f16_impl(const lz_head_params* p, const float* xyzs, const float* dirs, uint32_t M, const int32_t* count, float* sigmas,
                             float* rgbs, float* amb_aud, float* amb_eye, float* unc, hipStream_t st) {
    LZ_REQUIRE(p->testing, LZ_ERR_UNSUPPORTED, "triplane_head_forward: the f16 head is inference-only (testing must be 1)");
    LzHead16Args a;
    a.emb[0] = p->emb_xy; a.emb[1] = p->emb_yz; a.emb[2] = p->emb_xz;
    a.offsets = p->offsets; a.packed = reinterpret_cast<const lz_h8*>(p->packed); a.enc_a = p->enc_a; a.ind_code = p->ind_code;
    a.eye = p->eye; a.bound = p->bound;
    for (int l = 0; l < 12; l++) {
        const float sc = exp2f((float)l * p->S) * (float)p->H - 1.0f;
        a.scale[l] = sc;
        a.res[l] = (uint32_t)ceilf(sc) + 1u;
    }
    const int n_cu = lz_cu_count();   // of the current device, per call (cached per device)
    const uint32_t tiles = lz_div_up(M, 512);
    const uint32_t grid = tiles < (uint32_t)n_cu ? tiles : (uint32_t)n_cu;
    hipLaunchKernelGGL(lz_k_triplane_head_f16w, dim3(grid), dim3(H_WG), 0, st, a, xyzs, dirs, M, count, sigmas, rgbs, amb_aud, amb_eye, unc);
    return LZ_OK;
}
