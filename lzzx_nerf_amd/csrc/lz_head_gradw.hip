// lz_head_gradw.hip -- weight gradients of the wide Linear layers of the triplane head from the per-sample records that
// lz_triplane_head_backward writes (include/lzzx_nerf_hip.h: LZ_BWD_*), in ONE pass over the records.
//
// dW_layer[n, k] = sum over samples of G[s, n] * X[s, k] (network.py:73-94: bias-free nn.Linear; torch derives the same sums through
// addmm).  Five products over the same M ~ 6e6 records (2 624 B each), all HBM-bound.  A workgroup is five waves, each owning the
// accumulator tiles (16 x 16) of one product or half of one, and all five walk the same records at the same time:
//     wave 0  {aud_ch_att_net.0 | eye_att_net.0 | unc_net.0} stacked  G_X [112] x X_SIG0[:36]                      21 tiles
//     wave 1  aud_ch_att_net.1  G_ATT [32] x X_A1 [64], then sigma_net.1  G_S2 [64] x X_S1 [64]                     8 + 16
//     wave 2  sigma_net.0       G_S1 [64] x X_SIG0 [68 | 69]                                                        20
//     wave 3, 4  {color_net.0 | sigma row of sigma_net.2}  G_C1H [65] x X_S2C [84] = [s2 | SH | ind]                 18 + 12
// The last product is the shared factor of two layers: geo = s2 . Wg^T is linear in s2 and d geo = G_c1 . W_c0[:, geo] is linear in
// G_c1, so with R = sum G_c1^T s2 (its rows 0..63, columns 0..63) the host finishes dW_color0[:, geo] = R . Wg^T and
// dW_sigma2[geo rows] = W_c0[:, geo]^T . R with two 64^3 products; neither geo nor d geo is ever written to the records.
// v_mfma_f32_16x16x4_f32 with D[n, k] = A (16 n x 4 samples) . B (4 samples x 16 k): lane (i = l & 15, kk = l >> 4) holds
// G[4 j + kk][n-tile + i] resp. X[4 j + kk][k-tile + i], loaded straight from the records (64-byte row segments), with the loads
// of the next group of samples issued before the MFMAs of the current one.  No atomics: every workgroup writes its partial tiles
// (95 x 256 floats, fragment order) to the workspace and a second small kernel sums the partials and scatters them into the
// row-major outputs -- deterministic for a given grid.
#include "lz_common.h"

typedef float lz_f4 __attribute__((ext_vector_type(4)));

#define LZ_GW_TILES 95
#define LZ_GW_WAVES 5
#define LZ_GW_MAX_PARTS 768

namespace {
// first tile of each of the five products in the partial image
constexpr int T_X3 = 0, T_AUD1 = 21, T_SIG1 = 29, T_SIG0 = 45, T_C1H = 65;

template <int NBT, int KBT>
__device__ __forceinline__ void lz_gw_product(const float* __restrict__ rec, uint32_t M, int gcol, int xcol, float* __restrict__ part) {
    const uint32_t lane = threadIdx.x & 63, i = lane & 15, kk = lane >> 4;
    lz_f4 acc[NBT][KBT];
#pragma unroll
    for (int t = 0; t < NBT; t++)
#pragma unroll
        for (int u = 0; u < KBT; u++) acc[t][u] = lz_f4{0, 0, 0, 0};
    constexpr int GJ = 2;   // MFMA steps (of 4 samples) per group
    const uint32_t n_groups = (M + 4 * GJ - 1) / (4 * GJ);
    // The operands of group g + stride are requested BEFORE the MFMAs of group g are issued (and the scheduling barrier keeps them
    // there): a wave always has one group of loads (GJ x (NBT + KBT) row segments) in flight, which is what hides the HBM latency.
    // Rows past the end (and whole groups past the end) read record 0 and get a zero factor on the G side.  Columns are NOT masked:
    // a tile may run past its slot's width into padding or the next slot, but column n of G only reaches row n of D and column k of
    // X only column k of D, and rows >= N / columns >= K of D are never written out.
    float a[2][GJ][NBT], b[2][GJ][KBT], keep[2][GJ];
    auto load = [&](uint32_t g, float (&aa)[GJ][NBT], float (&bb)[GJ][KBT], float (&kp)[GJ]) {
#pragma unroll
        for (uint32_t j = 0; j < GJ; j++) {
            const uint32_t row = g * (4 * GJ) + 4 * j + kk;
            const bool row_ok = g < n_groups && row < M;
            const float* r = rec + (size_t)(row_ok ? row : 0) * LZ_BWD_REC + i;
            kp[j] = row_ok ? 1.0f : 0.0f;
#pragma unroll
            for (int t = 0; t < NBT; t++) aa[j][t] = r[gcol + 16 * t];
#pragma unroll
            for (int u = 0; u < KBT; u++) bb[j][u] = r[xcol + 16 * u];
        }
    };
    auto mma = [&](const float (&aa)[GJ][NBT], const float (&bb)[GJ][KBT], const float (&kp)[GJ]) {
#pragma unroll
        for (uint32_t j = 0; j < GJ; j++)
#pragma unroll
            for (int t = 0; t < NBT; t++) {
                const float at = aa[j][t] * kp[j];
#pragma unroll
                for (int u = 0; u < KBT; u++) acc[t][u] = __builtin_amdgcn_mfma_f32_16x16x4f32(at, bb[j][u], acc[t][u], 0, 0, 0);
            }
    };
    load(blockIdx.x, a[0], b[0], keep[0]);
    for (uint32_t g = blockIdx.x; g < n_groups; g += 2 * gridDim.x) {   // two groups per trip: the buffers are addressed statically
        load(g + gridDim.x, a[1], b[1], keep[1]);
        __builtin_amdgcn_sched_barrier(0);
        mma(a[0], b[0], keep[0]);
        __builtin_amdgcn_sched_barrier(0);
        load(g + 2 * gridDim.x, a[0], b[0], keep[0]);
        __builtin_amdgcn_sched_barrier(0);
        mma(a[1], b[1], keep[1]);   // a group past the end multiplies record 0 by zero
        __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int t = 0; t < NBT; t++)
#pragma unroll
        for (int u = 0; u < KBT; u++)
#pragma unroll
            for (int r = 0; r < 4; r++) part[((t * KBT + u) * 4 + r) * 64 + lane] = acc[t][u][r];
}
}  // namespace

__global__ void __launch_bounds__(64 * LZ_GW_WAVES)
lz_k_head_grad_w(const float* __restrict__ rec, uint32_t M, float* __restrict__ parts) {
    float* part = parts + (size_t)blockIdx.x * (LZ_GW_TILES * 256);
    // job = <output-row tiles, input-column tiles>(first G column, first X column, first tile of the partial image); a product's tiles
    // are numbered t * KBT + u, so a block of output rows is a contiguous run of tiles
#define LZ_GW_JOB(NB, KB, GCOL, XCOL, TILE) lz_gw_product<NB, KB>(rec, M, GCOL, XCOL, part + (TILE) * 256)
    switch (threadIdx.x >> 6) {   // wave-uniform
        case 0: LZ_GW_JOB(7, 3, LZ_BWD_G_X, LZ_BWD_X_SIG0, T_X3); break;
        case 1:
            LZ_GW_JOB(2, 4, LZ_BWD_G_ATT, LZ_BWD_X_A1, T_AUD1);
            LZ_GW_JOB(4, 4, LZ_BWD_G_S2, LZ_BWD_X_S1, T_SIG1);
            break;
        case 2: LZ_GW_JOB(4, 5, LZ_BWD_G_S1, LZ_BWD_X_SIG0, T_SIG0); break;
        case 3: LZ_GW_JOB(3, 6, LZ_BWD_G_C1H, LZ_BWD_X_S2C, T_C1H); break;
        default: LZ_GW_JOB(2, 6, LZ_BWD_G_C1H + 48, LZ_BWD_X_S2C, T_C1H + 18); break;
    }
#undef LZ_GW_JOB
}

struct LzGwOut {
    float* dw[5];   // x3, aud1, sig1, sig0, c1h (tile order)
    int N[5], K[5], KBT[5], tile0[6];
};

// element e = (tile, r, lane) of the partial image: D row 4 (lane >> 4) + r -> n = 16 t + row, column lane & 15 -> k = 16 u + column
// One workgroup per tile: 256 elements x 4 interleaved slices of the partials (thread = element + 256 * slice), four independent
// sums per thread so that 16 loads per element are in flight, slices combined through LDS in a fixed order.
__global__ void __launch_bounds__(1024)
lz_k_head_grad_w_reduce(const float* __restrict__ parts, uint32_t n_parts, LzGwOut o) {
    __shared__ float red[4][256];
    const uint32_t el = threadIdx.x & 255, slice = threadIdx.x >> 8;
    const uint32_t e = blockIdx.x * 256 + el;
    const float* src = parts + e;
    float s0 = 0.0f, s1 = 0.0f, s2 = 0.0f, s3 = 0.0f;
    uint32_t p = slice;
    for (; p + 12 < n_parts; p += 16) {
        s0 += src[(size_t)p * (LZ_GW_TILES * 256)];
        s1 += src[(size_t)(p + 4) * (LZ_GW_TILES * 256)];
        s2 += src[(size_t)(p + 8) * (LZ_GW_TILES * 256)];
        s3 += src[(size_t)(p + 12) * (LZ_GW_TILES * 256)];
    }
    for (; p < n_parts; p += 4) s0 += src[(size_t)p * (LZ_GW_TILES * 256)];
    red[slice][el] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (slice != 0) return;
    const float v = (red[0][el] + red[1][el]) + (red[2][el] + red[3][el]);
    const int tile = e >> 8, r = (e >> 6) & 3, lane = e & 63;
    int job = 0;
#pragma unroll
    for (int j = 1; j < 5; j++)
        if (tile >= o.tile0[j]) job = j;
    const int tl = tile - o.tile0[job];
    const int t = tl / o.KBT[job], u = tl - t * o.KBT[job];
    const int n = 16 * t + 4 * (lane >> 4) + r, k = 16 * u + (lane & 15);
    if (n < o.N[job] && k < o.K[job]) o.dw[job][(size_t)n * o.K[job] + k] = v;
}

extern "C" size_t lz_triplane_head_grad_w_workspace(void) { return (size_t)LZ_GW_MAX_PARTS * LZ_GW_TILES * 256 * sizeof(float); }

extern "C" int lz_triplane_head_grad_w(const float* rec, uint32_t M, uint32_t k_sig0, float* dW_x3, float* dW_aud1, float* dW_sig0,
                                       float* dW_sig1, float* dW_c1h, void* workspace, lz_stream_t stream) {
    LZ_REQUIRE(dW_x3 && dW_aud1 && dW_sig0 && dW_sig1 && dW_c1h && workspace, LZ_ERR_BAD_ARGUMENT, "head_grad_w: null tensor");
    LZ_REQUIRE(M == 0 || rec, LZ_ERR_BAD_ARGUMENT, "head_grad_w: null records");
    LZ_REQUIRE(k_sig0 == 68 || k_sig0 == 69, LZ_ERR_BAD_ARGUMENT, "head_grad_w: sigma_net.0 takes 68 or 69 inputs");
    static int n_cu = 0;
    if (n_cu == 0) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) n_cu = prop.multiProcessorCount;
        if (n_cu <= 0) n_cu = 256;
    }
    const uint32_t groups = lz_div_up(M, 8);
    uint32_t grid = 3u * (uint32_t)n_cu;   // the registers of a CU hold two workgroups; a third queued one evens out the tail
    if (grid > LZ_GW_MAX_PARTS) grid = LZ_GW_MAX_PARTS;
    if (grid > groups) grid = groups;
    hipStream_t st = lz_st(stream);
    float* parts = static_cast<float*>(workspace);
    if (grid > 0) {
        hipLaunchKernelGGL(lz_k_head_grad_w, dim3(grid), dim3(64 * LZ_GW_WAVES), 0, st, rec, M, parts);
        LZ_CHECK_LAUNCH("head_grad_w");
    }
    LzGwOut o;
    float* dws[5] = {dW_x3, dW_aud1, dW_sig1, dW_sig0, dW_c1h};
    const int Ns[5] = {112, 32, 64, 64, 65}, Ks[5] = {36, 64, 64, (int)k_sig0, 84}, KB[5] = {3, 4, 4, 5, 6};
    const int t0[6] = {T_X3, T_AUD1, T_SIG1, T_SIG0, T_C1H, LZ_GW_TILES};
    for (int j = 0; j < 5; j++) { o.dw[j] = dws[j]; o.N[j] = Ns[j]; o.K[j] = Ks[j]; o.KBT[j] = KB[j]; }
    for (int j = 0; j < 6; j++) o.tile0[j] = t0[j];
    hipLaunchKernelGGL(lz_k_head_grad_w_reduce, dim3(LZ_GW_TILES), dim3(1024), 0, st, parts, grid, o);
    LZ_CHECK_LAUNCH("head_grad_w_reduce");
    return LZ_OK;
}
