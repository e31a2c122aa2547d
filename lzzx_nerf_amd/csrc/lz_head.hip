// lz_head.hip -- fused per-sample triplane head for gfx950: three hash-grid planes -> audio/eye attention ->
// sigma net -> SH(4) -> colour net (+ uncertainty net in training), one kernel, f32 end to end.
//
// Replaces, for the inference/training hot loop, the torch-level graph of NeRFNetwork.forward
// (nerf_triplane/network.py:252-311): 3 grid_encode launches + permutes + cat, ~10 bias-free Linear GEMMs with
// K in {36,69,84,64,16,32}, repeat/cat copies materialising [M,69] and [M,84], SH launch, exp/sigmoid kernels
// -- 52 % of the reference's loop time.  Here activations never leave registers.
//
// MI355X design
//   * GEMMs run on the matrix cores with v_mfma_f32_16x16x4_f32 (f32 in, f32 accumulate: bit-for-bit a
//     k-ordered fma chain, so the CPU checker reproduces every value exactly).  Orientation: A = weights
//     (16 output features x 4 k), B = activations (4 k x 16 samples), D = 16 features x 16 samples with
//     feature = 4*(lane>>4) + reg, sample = lane & 15.
//   * Layer chaining without data movement: lane (s, q) of a D tile holds features 16t+4q+r (r = reg) of
//     sample s, which is exactly the B operand of k-step (t, r) of the next layer if that layer's weights
//     are consumed in the permuted k order  f(t, r, q) = 16t + 4q + r.  The packed weight buffer
//     (lz_head_pack_weights) stores every A fragment pre-permuted, lane-linear: one conflict-free
//     ds_read_b32 per fragment, shared by the wave's 4 sample tiles.
//   * First-layer operands come straight from the gathers: the 4 lanes that share a sample split its 36
//     (plane, level) features (feature f = 4i + q), so the bilinear lookups are spread over all 64 lanes
//     and land directly in B-operand position.
//   * One 1024-thread workgroup per CU (16 waves = 4 per SIMD: some gather while another feeds the matrix pipe), persistent
//     over a contiguous share of the launch's 16-sample slices, which its waves pull from a queue in LDS; the packed
//     inference weights (361 fragments + the VALU-layer rows = 94 KB) stay in LDS for the life of the workgroup.
//   * The body of a slice lives in lz_head_slice.h, shared with the fused frame kernel (lz_frame.hip).
//   * `count` (device) bounds the work, so the render loop needs no host round trip.
#include "lz_head_slice.h"

extern "C" uint32_t lz_head_packed_size(void) { return (uint32_t)LZ_HEAD_PACKED_FLOATS; }

// ---- weight packing -------------------------------------------------------------------------------
__device__ __forceinline__ int lz_chained(int slot, int K) {  // slot -> feature in the chained order
    const int t = slot >> 4, r = (slot & 15) >> 2, q = slot & 3;
    const int f = 16 * t + 4 * q + r;
    return f < K ? f : -1;
}

struct LzPackArgs {
    const float* w[LZ_L_COUNT];  // source weight [nout, ld] per MFMA layer
    int nout[LZ_L_COUNT];
    int ld[LZ_L_COUNT];
    const float *eye1, *col1, *unc1;   // VALU layers: [1,16], [3,64], [1,32]
    int has_eye, has_ind;
};

// input feature of k slot `slot` of a layer's B operand (-1 = padding) / source row of output row `row` (-1 = padding row)
__device__ __forceinline__ int lz_pack_kf(int layer, int slot, const LzPackArgs& a) {
    switch (layer) {
        case LZ_L_A1: case LZ_L_E1: case LZ_L_U1: return slot < 36 ? slot : -1;
        case LZ_L_A2: case LZ_L_S2: case LZ_L_S3: return lz_chained(slot, 64);
        case LZ_L_S1:
            if (slot < 36) return slot;
            if (slot < 68) return 36 + lz_chained(slot - 36, 32);
            return (slot == 68 && a.has_eye) ? 68 : -1;
        default:  // LZ_L_C1
            if (slot < 16) return slot;
            if (slot < 80) return 16 + lz_chained(slot - 16, 64);
            return (slot < 84 && a.has_ind) ? slot : -1;
    }
}
__device__ __forceinline__ int lz_pack_srow(int layer, int row, const LzPackArgs& a) {
    if (layer == LZ_L_S3) return row < 64 ? row + 1 : -1;  // the 64 geo rows; the sigma row (0) is a VALU layer
    return row < a.nout[layer] ? row : -1;
}

__global__ void __launch_bounds__(256) lz_k_head_pack(LzPackArgs a, float* __restrict__ packed) {
    const int gid = blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= LZ_HEAD_PACKED_FLOATS) return;
    if (gid >= LZ_FRAGS_ALL * 64) {   // VALU-layer rows
        const int i = gid - LZ_FRAGS_ALL * 64;
        float v = 0.0f;
        if (i < LZ_WV_SIG) v = a.col1[i];                                                      // colour.1 [3][64]
        else if (i < LZ_WV_E2) v = a.w[LZ_L_S3][i - LZ_WV_SIG];                                // row 0 of sigma_net.2
        else if (i < LZ_WV_U2) v = (a.has_eye && a.eye1) ? a.eye1[i - LZ_WV_E2] : 0.0f;        // eye.1 [16]
        else if (i < LZ_WV_U2 + 32) v = a.unc1 ? a.unc1[i - LZ_WV_U2] : 0.0f;                  // unc.1 [32]
        packed[gid] = v;
        return;
    }
    const int frag = gid >> 6, lane = gid & 63;
    int layer = 0, fb = 0;
    for (int i = 0; i < LZ_L_COUNT; i++) {
        const int n = LZ_KS[i] * LZ_NT[i];
        if (frag < fb + n) { layer = i; break; }
        fb += n;
    }
    const int local = frag - fb;
    const int ks = local / LZ_NT[layer], ft = local - ks * LZ_NT[layer];
    const int row = 16 * ft + (lane & 15), slot = 4 * ks + (lane >> 4);
    const int kf = lz_pack_kf(layer, slot, a), srow = lz_pack_srow(layer, row, a);
    float v = 0.0f;
    if (kf >= 0 && srow >= 0 && a.w[layer]) v = a.w[layer][(size_t)srow * a.ld[layer] + kf];
    packed[gid] = v;
}

extern "C" int lz_head_pack_weights(const float* aud0, const float* aud1, const float* eye0, const float* eye1, const float* sig0,
                                    const float* sig1, const float* sig2, const float* col0, const float* col1, const float* unc0,
                                    const float* unc1, int has_eye, int has_ind, float* packed, lz_stream_t stream) {
    LZ_REQUIRE(aud0 && aud1 && sig0 && sig1 && sig2 && col0 && col1 && packed, LZ_ERR_BAD_ARGUMENT, "head_pack_weights: null weight");
    LZ_REQUIRE(!has_eye || (eye0 && eye1), LZ_ERR_BAD_ARGUMENT, "head_pack_weights: eye weights required when has_eye");
    LzPackArgs a;
    const float* w[LZ_L_COUNT] = {aud0, aud1, eye0, sig0, sig1, sig2, col0, unc0};
    const int nout[LZ_L_COUNT] = {64, 32, 16, 64, 64, 65, 64, 32};
    const int ld[LZ_L_COUNT] = {36, 64, 36, 68 + (has_eye ? 1 : 0), 64, 64, 80 + (has_ind ? 4 : 0), 36};
    for (int i = 0; i < LZ_L_COUNT; i++) { a.w[i] = w[i]; a.nout[i] = nout[i]; a.ld[i] = ld[i]; }
    a.eye1 = eye1; a.col1 = col1; a.unc1 = unc1;
    a.has_eye = has_eye; a.has_ind = has_ind;
    hipLaunchKernelGGL(lz_k_head_pack, dim3(lz_div_up((uint64_t)LZ_HEAD_PACKED_FLOATS, 256)), dim3(256), 0, lz_st(stream), a, packed);
    LZ_CHECK_LAUNCH("head_pack_weights");
    return LZ_OK;
}

// ---- transposed weights for the backward on the f16 matrix cores (lz_head_rec.hip, lz_layer_bwd16) -----------------------------
// dX = W^T dY per layer with v_mfma_f32_16x16x16_f16: fragment (layer, kt, ft), lane (rho = l & 15, kg = l >> 4) holds the four halves
// W[out row 16 ft + 4 kg + j][in slot 16 kt + 4 (rho & 3) + (rho >> 2)], j < 4 -- row rho of the backward D tile kt is the gradient of the
// value lane q = rho >> 2 supplied to the forward k-step 4 kt + (rho & 3) (lz_head_bwd_common.h), and the K index of the instruction
// runs over the output rows 4 kg + j of tile ft, which is how a D tile of dY sits in a lane's four registers.
extern "C" uint32_t lz_head_packed_bwd_size_f16(void) { return (uint32_t)LZ_BFRAGS * 64u * 8u; }

__global__ void __launch_bounds__(256) lz_k_head_pack_bwd_f16(LzPackArgs a, _Float16* __restrict__ packed) {
    const int gid = blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= LZ_BFRAGS * 64) return;
    const int frag = gid >> 6, lane = gid & 63;
    int layer = 0, fb = 0;
    for (int i = 0; i < LZ_L_COUNT; i++) {
        const int n = i == LZ_L_U1 ? 0 : lz_kt(i) * LZ_NT[i];
        if (frag < fb + n) { layer = i; break; }
        fb += n;
    }
    const int local = frag - fb;
    const int kt = local / LZ_NT[layer], ft = local - kt * LZ_NT[layer];
    const int rho = lane & 15, kg = lane >> 4;
    const int slot = 16 * kt + 4 * (rho & 3) + (rho >> 2);
    const int kf = slot < 4 * LZ_KS[layer] ? lz_pack_kf(layer, slot, a) : -1;
    for (int j = 0; j < 4; j++) {
        const int srow = lz_pack_srow(layer, 16 * ft + 4 * kg + j, a);
        float v = 0.0f;
        if (kf >= 0 && srow >= 0 && a.w[layer]) v = a.w[layer][(size_t)srow * a.ld[layer] + kf];
        packed[(size_t)gid * 4 + j] = (_Float16)v;
    }
}

extern "C" int lz_head_pack_weights_bwd_f16(const float* aud0, const float* aud1, const float* eye0, const float* sig0, const float* sig1,
                                            const float* sig2, const float* col0, int has_eye, int has_ind, void* packed_bwd16,
                                            lz_stream_t stream) {
    LZ_REQUIRE(aud0 && aud1 && sig0 && sig1 && sig2 && col0 && packed_bwd16, LZ_ERR_BAD_ARGUMENT, "head_pack_weights_bwd_f16: null weight");
    LZ_REQUIRE(!has_eye || eye0, LZ_ERR_BAD_ARGUMENT, "head_pack_weights_bwd_f16: eye weights required when has_eye");
    LzPackArgs a;
    const float* w[LZ_L_COUNT] = {aud0, aud1, eye0, sig0, sig1, sig2, col0, nullptr};
    const int nout[LZ_L_COUNT] = {64, 32, 16, 64, 64, 65, 64, 32};
    const int ld[LZ_L_COUNT] = {36, 64, 36, 68 + (has_eye ? 1 : 0), 64, 64, 80 + (has_ind ? 4 : 0), 36};
    for (int i = 0; i < LZ_L_COUNT; i++) { a.w[i] = w[i]; a.nout[i] = nout[i]; a.ld[i] = ld[i]; }
    a.eye1 = a.col1 = a.unc1 = nullptr;
    a.has_eye = has_eye; a.has_ind = has_ind;
    hipLaunchKernelGGL(lz_k_head_pack_bwd_f16, dim3(lz_div_up((uint64_t)LZ_BFRAGS * 64, 256)), dim3(256), 0, lz_st(stream), a,
                       reinterpret_cast<_Float16*>(packed_bwd16));
    LZ_CHECK_LAUNCH("head_pack_weights_bwd_f16");
    return LZ_OK;
}

// ---- the fused head ---------------------------------------------------------------------------------
#define LZ_WG 1024         // threads per workgroup (16 waves = 4 per SIMD)
#define LZ_WG_SAMPLES (LZ_WG / 64 * LZ_T * 16)

// diagnostic: shader-clock cycles (s_memtime) and 100 MHz wall ticks (s_memrealtime) that wave 0 of workgroup 0 of the most
// recent launch spent in the kernel -> the sustained shader clock under this kernel's load (lz_debug_head_clocks)
__device__ unsigned long long lz_head_probe[2];

template <bool TRAIN_UNC, bool FOLD = false>
__global__ void __launch_bounds__(LZ_WG, LZ_WG / 256)
lz_k_triplane_head(LzHeadArgs P, const float* __restrict__ xyzs, const float* __restrict__ dirs, uint32_t M,
                   const int* __restrict__ count, float* __restrict__ sigmas, float* __restrict__ rgbs,
                   float* __restrict__ amb_aud, float* __restrict__ amb_eye, float* __restrict__ unc_out) {
    __shared__ float wl[LzHeadLds<TRAIN_UNC>::FLOATS];  // packed A fragments, VALU-layer rows, level table (64 words), enc_a (32)
    uint32_t Meff = M;
    if (count) {
        const int c = *count;
        Meff = c < 0 ? 0u : ((uint32_t)c < M ? (uint32_t)c : M);
    }
    // the workgroup owns a contiguous, near-equal share of the 16-sample slices (shares differ by at most one slice, so a
    // launch whose row count is not a multiple of gridDim.x * LZ_WG_SAMPLES loses at most one slice time, not one tile time)
    const uint32_t n_slices = (Meff + LZ_T * 16 - 1) / (LZ_T * 16);
    const uint32_t slice_lo = (uint32_t)(((uint64_t)n_slices * blockIdx.x) / gridDim.x);
    const uint32_t slice_hi = (uint32_t)(((uint64_t)n_slices * (blockIdx.x + 1)) / gridDim.x);
    if (slice_lo >= slice_hi) return;
    const bool probe = blockIdx.x == 0 && threadIdx.x == 0;
    unsigned long long probe_c = 0, probe_w = 0;
    if (probe) { probe_c = clock64(); probe_w = wall_clock64(); }

    const int lane = threadIdx.x & 63;
    const int s = lane & 15, q = lane >> 4;
    LzHeadCtx ctx;
    lz_head_stage<TRAIN_UNC>(P, wl, LZ_WG, q, ctx);
    __syncthreads();

    // Work distribution: the workgroup owns slices [slice_lo, slice_hi); its 16 waves pull 16-sample slices from a queue
    // in LDS (one ds_add_rtn per slice).  The waves that share a SIMD drift apart: some gather (texture-address bound)
    // while another feeds the matrix pipe, instead of all gathering and then all multiplying in lockstep.
    int* queue = reinterpret_cast<int*>(wl + LzHeadLds<TRAIN_UNC>::TAB) + LZ_LVTAB_QUEUE;
    for (;;) {
        int slice = 0;
        if (lane == 0) slice = atomicAdd(queue, 1);
        slice = __builtin_amdgcn_readfirstlane(slice);
        if (slice_lo + (uint32_t)slice >= slice_hi) {
            if (probe) { lz_head_probe[0] = clock64() - probe_c; lz_head_probe[1] = wall_clock64() - probe_w; }
            break;
        }
        const uint32_t base = (slice_lo + (uint32_t)slice) * (LZ_T * 16);
        uint32_t m = base + s;
        if (m >= Meff) m = Meff - 1;  // clamp: computed, never stored
        const float px = xyzs[(size_t)m * 3], py = xyzs[(size_t)m * 3 + 1], pz = xyzs[(size_t)m * 3 + 2];
        LzHeadOut o;
        lz_head_slice<TRAIN_UNC, FOLD>(ctx, lane, px, py, pz,
                                 lz_sh_from_dir([&](float& dx, float& dy, float& dz) { dx = dirs[(size_t)m * 3]; dy = dirs[(size_t)m * 3 + 1]; dz = dirs[(size_t)m * 3 + 2]; }), o);
        // ---------------- store (lanes q == 0 own sample s) ----------------
        if (q == 0 && base + s < Meff) {
            sigmas[m] = o.sigma;
            rgbs[(size_t)m * 3] = o.rgb[0]; rgbs[(size_t)m * 3 + 1] = o.rgb[1]; rgbs[(size_t)m * 3 + 2] = o.rgb[2];
            amb_aud[m] = o.ambaud;
            if (amb_eye) amb_eye[m] = o.eyeatt;
            unc_out[m] = o.unc;
        }
    }
}

int lz_head_forward_f16_impl(const lz_head_params* p, const float* xyzs, const float* dirs, uint32_t M, const int32_t* count, float* sigmas,
                             float* rgbs, float* amb_aud, float* amb_eye, float* unc, hipStream_t st);   // lz_head_f16.hip

extern "C" int lz_debug_head_clocks(uint64_t* out2) {
    LZ_REQUIRE(out2, LZ_ERR_BAD_ARGUMENT, "debug_head_clocks: null");
    unsigned long long v[2] = {0, 0};
    hipError_t rc = hipMemcpyFromSymbol(v, HIP_SYMBOL(lz_head_probe), sizeof(v));
    if (rc != hipSuccess) { lz_set_error("debug_head_clocks: %s", hipGetErrorString(rc)); return (int)rc; }
    out2[0] = v[0]; out2[1] = v[1];
    return LZ_OK;
}

extern "C" int lz_triplane_head_forward(const lz_head_params* p, const float* xyzs, const float* dirs, uint32_t M,
                                        const int32_t* count, float* sigmas, float* rgbs, float* amb_aud, float* amb_eye,
                                        float* unc, lz_stream_t stream) {
    LZ_REQUIRE(p && xyzs && dirs && sigmas && rgbs && amb_aud && unc, LZ_ERR_BAD_ARGUMENT, "triplane_head_forward: null tensor");
    LZ_REQUIRE(p->emb_xy && p->emb_yz && p->emb_xz && p->offsets && p->packed && p->enc_a, LZ_ERR_BAD_ARGUMENT,
               "triplane_head_forward: incomplete lz_head_params");
    if (M == 0) return LZ_OK;
    if (p->precision == 1) {
        const int rc = lz_head_forward_f16_impl(p, xyzs, dirs, M, count, sigmas, rgbs, amb_aud, amb_eye, unc, lz_st(stream));
        if (rc != LZ_OK) return rc;
        LZ_CHECK_LAUNCH("triplane_head_forward(f16)");
        return LZ_OK;
    }
    LZ_REQUIRE(p->precision == 0 || p->precision == 2, LZ_ERR_BAD_ARGUMENT, "triplane_head_forward: precision must be 0 (f32), 1 (f16) or 2 (f32, folded geo)");
    LZ_REQUIRE(p->precision == 0 || p->testing, LZ_ERR_UNSUPPORTED, "triplane_head_forward: the folded colour net is inference-only");
    LzHeadArgs a;
    a.emb[0] = p->emb_xy; a.emb[1] = p->emb_yz; a.emb[2] = p->emb_xz;
    a.offsets = p->offsets; a.packed = reinterpret_cast<const float*>(p->packed); a.enc_a = p->enc_a; a.ind_code = p->ind_code; a.eye = p->eye;
    a.bound = p->bound; a.testing = p->testing;
    for (int l = 0; l < 12; l++) {  // gridencoder.cu:125-126 on the host, same libm call as the CPU checker
        const float sc = exp2f((float)l * p->S) * (float)p->H - 1.0f;
        a.scale[l] = sc;
        a.res[l] = (uint32_t)ceilf(sc) + 1u;
    }
    const int n_cu = lz_cu_count();   // of the current device, per call (cached per device)
    const uint32_t tiles = lz_div_up(M, LZ_WG_SAMPLES);
    const uint32_t grid = tiles < (uint32_t)n_cu ? tiles : (uint32_t)n_cu;
    if (p->precision == 2)
        hipLaunchKernelGGL((lz_k_triplane_head<false, true>), dim3(grid), dim3(LZ_WG), 0, lz_st(stream), a, xyzs, dirs, M, count, sigmas, rgbs, amb_aud, amb_eye, unc);
    else if (p->testing)
        hipLaunchKernelGGL((lz_k_triplane_head<false>), dim3(grid), dim3(LZ_WG), 0, lz_st(stream), a, xyzs, dirs, M, count, sigmas, rgbs, amb_aud, amb_eye, unc);
    else
        hipLaunchKernelGGL((lz_k_triplane_head<true>), dim3(grid), dim3(LZ_WG), 0, lz_st(stream), a, xyzs, dirs, M, count, sigmas, rgbs, amb_aud, amb_eye, unc);
    LZ_CHECK_LAUNCH("triplane_head_forward");
    return LZ_OK;
}
