// lz_head_gradw.hip -- weight gradients of the wide Linear layers of the triplane head from the per-sample records that
// lz_triplane_head_backward writes (include/lzzx_nerf_hip.h: LZ_BWD_*), in ONE pass over the records.
//
// dW_layer[n, k] = sum over samples of G[s, n] * X[s, k] (network.py:73-94: bias-free nn.Linear; torch derives the same sums through
// addmm).  Five products over the same M ~ 6e6 records (2 624 B each, or 1 408 B in half), all HBM-bound.  A workgroup is six waves, each
// owning the accumulator tiles (16 x 16) of one product or part of one, and all six walk the same records at the same time, one pass each:
//     wave 0  {aud_ch_att_net.0 | eye_att_net.0 | unc_net.0} stacked  G_X [112] x X_SIG0[:36]                      21 tiles
//     wave 1  aud_ch_att_net.1  G_ATT [32] x X_A1 [64]                                                              8
//     wave 2  sigma_net.0       G_S1 [64] x X_SIG0 [68 | 69]                                                        20
//     wave 3, 4  {color_net.0 | sigma row of sigma_net.2}  G_C1H [65] x X_S2C [84] = [s2 | SH | ind]                 18 + 12
//     wave 5  sigma_net.1       G_S2 [64] x X_S1 [64]                                                               16
// The last product is the shared factor of two layers: geo = s2 . Wg^T is linear in s2 and d geo = G_c1 . W_c0[:, geo] is linear in
// G_c1, so with R = sum G_c1^T s2 (its rows 0..63, columns 0..63) the host finishes dW_color0[:, geo] = R . Wg^T and
// dW_sigma2[geo rows] = W_c0[:, geo]^T . R with two 64^3 products; neither geo nor d geo is ever written to the records.
// v_mfma_f32_16x16x4_f32 with D[n, k] = A (16 n x 4 samples) . B (4 samples x 16 k): lane (i = l & 15, kk = l >> 4) holds
// G[4 j + kk][n-tile + i] resp. X[4 j + kk][k-tile + i], loaded straight from the records (64-byte row segments), with the loads
// of the next group of samples issued before the MFMAs of the current one.  No atomics: every workgroup writes its partial tiles
// (95 x 256 floats, fragment order) to the workspace and a second small kernel sums the partials and scatters them into the
// row-major outputs -- deterministic for a given grid.
#include "lz_common.h"
#include "lz_head_bwd_common.h"

typedef float lz_f4 __attribute__((ext_vector_type(4)));

#define LZ_GW_TILES LZ_DW_TILES
#define LZ_GW_WAVES 6
#define LZ_GW_MAX_PARTS LZ_DW_MAX_PARTS

namespace {
// first tile of each of the five products in the partial image
constexpr int T_X3 = LZ_DW_T_X3, T_AUD1 = LZ_DW_T_AUD1, T_SIG1 = LZ_DW_T_SIG1, T_SIG0 = LZ_DW_T_SIG0, T_C1H = LZ_DW_T_C1H;

// f32 records: lane (i, kk) of v_mfma_f32_16x16x4_f32 reads one float per (row, tile): row 4 j + kk of MFMA step j.  Records are
// blocked by 16-sample slice, [slice][tile][sample][16 dwords]: the four rows of a load instruction are four consecutive 64-byte pieces.
template <int T0, int NT>
struct LzGwOperand32 {
    float raw[NT];
    __device__ __forceinline__ void load(const float* __restrict__ rec, size_t row, uint32_t i) {
        const float* r = rec + (row >> 4) * (size_t)(16 * LZ_BWD_REC) + (row & 15) * 16 + i;   // blocked by 16-sample slice (lz_head_bwd_common.h)
#pragma unroll
        for (int t = 0; t < NT; t++) raw[t] = r[256 * (T0 + t)];
    }
};

// G tiles [GT0, GT0 + NBT) x X tiles [XT0, XT0 + KBT), tile = 16 record columns
template <int NBT, int KBT, int GT0, int XT0>
__device__ __forceinline__ void lz_gw_product(const float* __restrict__ rec, uint32_t M, float* __restrict__ part) {
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
    typedef LzGwOperand32<GT0, NBT> OpG;
    typedef LzGwOperand32<XT0, KBT> OpX;
    OpG a[2][GJ];
    OpX b[2][GJ];
    float keep[2][GJ];
    auto load = [&](uint32_t g, OpG (&aa)[GJ], OpX (&bb)[GJ], float (&kp)[GJ]) {
#pragma unroll
        for (uint32_t j = 0; j < GJ; j++) {
            const uint32_t row = g * (4 * GJ) + 4 * j + kk;
            const bool row_ok = g < n_groups && row < M;
            kp[j] = row_ok ? 1.0f : 0.0f;
            aa[j].load(rec, row_ok ? row : 0, i);
            bb[j].load(rec, row_ok ? row : 0, i);
        }
    };
    auto mma = [&](const OpG (&aa)[GJ], const OpX (&bb)[GJ], const float (&kp)[GJ]) {
#pragma unroll
        for (uint32_t j = 0; j < GJ; j++)
#pragma unroll
            for (int t = 0; t < NBT; t++) {
                const float at = aa[j].raw[t] * kp[j];
#pragma unroll
                for (int u = 0; u < KBT; u++) acc[t][u] = __builtin_amdgcn_mfma_f32_16x16x4f32(at, bb[j].raw[u], acc[t][u], 0, 0, 0);
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

// f16 records (LZ_BWD_REC16 halves per sample, tiles interleaved in pairs: dword i of pair g = {tile 2 g column i, tile 2 g + 1
// column i}).  The operands are halves already, so the product runs on v_mfma_f32_16x16x16_f16: one instruction per tile and 16
// samples instead of four.  Its A / B operand wants, in lane (i, kk), four samples of column i as its four k slots: the lane reads the
// pair's dword of four rows of the block (as many loads as the f32 path issues for 16 samples) and two byte-permutes per tile gather
// the low (even tile) or high (odd tile) halves.  Products of halves are exact in f32 and the accumulation is f32, so the
// result equals converting to f32 first up to summation order.
typedef _Float16 lz_h4 __attribute__((ext_vector_type(4)));
template <int T0, int NT>
struct LzGwOperand16 {
    static constexpr int P0 = T0 >> 1, NP = ((T0 + NT - 1) >> 1) - P0 + 1;
    uint32_t raw[4][NP];
    // k slot m of lane (i, kk) = sample 4 m + kk of the 16-sample block (any bijection works as long as both operands use it): the
    // four lane groups of one load instruction then read four CONSECUTIVE 64-byte slots of the block, two whole cache lines.  Samples
    // past M are padding slots of the last block (allocated, never written with anything that matters): read and zeroed.
    __device__ __forceinline__ void load(const uint32_t* __restrict__ blk, uint32_t first_row, uint32_t M, bool block_ok, uint32_t i, uint32_t kk) {
#pragma unroll
        for (int m = 0; m < 4; m++) {
            const uint32_t* r = blk + (4 * m + kk) * 16 + i;
            const bool ok = block_ok && first_row + 4 * m + kk < M;
#pragma unroll
            for (int p = 0; p < NP; p++) {
                const uint32_t v = r[256 * (P0 + p)];
                raw[m][p] = ok ? v : 0u;
            }
        }
    }
    __device__ __forceinline__ lz_h4 get(int t) const {   // t compile-time after unrolling
        const int p = ((T0 + t) >> 1) - P0;
        const uint32_t sel = ((T0 + t) & 1) ? 0x07060302u : 0x05040100u;   // {b.half, a.half}: high halves for odd tiles
        typedef uint32_t lz_u2 __attribute__((ext_vector_type(2)));
        const lz_u2 w = {__builtin_amdgcn_perm(raw[1][p], raw[0][p], sel), __builtin_amdgcn_perm(raw[3][p], raw[2][p], sel)};
        return __builtin_bit_cast(lz_h4, w);
    }
};

template <int NBT, int KBT, int GT0, int XT0>
__device__ __forceinline__ void lz_gw_product16(const uint32_t* __restrict__ rec, uint32_t M, float* __restrict__ part) {
    const uint32_t lane = threadIdx.x & 63, i = lane & 15, kk = lane >> 4;
    lz_f4 acc[NBT][KBT];
#pragma unroll
    for (int t = 0; t < NBT; t++)
#pragma unroll
        for (int u = 0; u < KBT; u++) acc[t][u] = lz_f4{0, 0, 0, 0};
    const uint32_t n_groups = (M + 15) / 16;   // one group = one 16-sample block = one MFMA per tile
    typedef LzGwOperand16<GT0, NBT> OpG;
    typedef LzGwOperand16<XT0, KBT> OpX;
    OpG a[2];
    OpX b[2];
    // Unwritten padding columns may hold anything, NaN included: column n of G only reaches row n of D and column k of X only column
    // k, and padding rows / columns of D are never written out.  A group past the end reads block 0 and is zeroed.
    auto load = [&](uint32_t g, OpG& aa, OpX& bb) {
        const bool ok = g < n_groups;
        const uint32_t* blk = rec + (size_t)(ok ? g : 0) * (16 * (LZ_BWD_REC16 / 2));
        aa.load(blk, g * 16, M, ok, i, kk);
        bb.load(blk, g * 16, M, ok, i, kk);
    };
    auto mma = [&](const OpG& aa, const OpX& bb) {
        lz_h4 bv[KBT];
#pragma unroll
        for (int u = 0; u < KBT; u++) bv[u] = bb.get(u);
#pragma unroll
        for (int t = 0; t < NBT; t++) {
            const lz_h4 at = aa.get(t);
#pragma unroll
            for (int u = 0; u < KBT; u++) acc[t][u] = __builtin_amdgcn_mfma_f32_16x16x16f16(at, bv[u], acc[t][u], 0, 0, 0);
        }
    };
    load(blockIdx.x, a[0], b[0]);
    for (uint32_t g = blockIdx.x; g < n_groups; g += 2 * gridDim.x) {
        // every wave of the workgroup makes the same trips: keeping them on the same blocks lets the columns two waves share (the
        // sigma_net.0 / color_net.0 inputs) come from HBM once and from the cache the second time (FETCH_SIZE 9.8 -> 8.x GB; 1.81 ->
        // 1.72 ms).  The f32 product is left free-running: there the barrier costs more than the 18 % of re-read it saves (3.8 -> 4.1 ms)
        __syncthreads();
        load(g + gridDim.x, a[1], b[1]);
        __builtin_amdgcn_sched_barrier(0);
        mma(a[0], b[0]);
        __builtin_amdgcn_sched_barrier(0);
        load(g + 2 * gridDim.x, a[0], b[0]);
        __builtin_amdgcn_sched_barrier(0);
        mma(a[1], b[1]);
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

// tile numbers: f32 = record column / 16 (LZ_BWD_*); f16 = LZ_R16_* (include/lzzx_nerf_hip.h).  A product's tiles are numbered
// t * KBT + u in the partial image, so a block of output rows is a contiguous run of tiles.
__global__ void __launch_bounds__(64 * LZ_GW_WAVES)
lz_k_head_grad_w(const float* __restrict__ rec, uint32_t M, float* __restrict__ parts) {
    float* part = parts + (size_t)blockIdx.x * (LZ_GW_TILES * 256);
    constexpr int X_A1 = LZ_BWD_X_A1 / 16, X_SIG0 = LZ_BWD_X_SIG0 / 16, X_S1 = LZ_BWD_X_S1 / 16, X_S2C = LZ_BWD_X_S2C / 16, G_X = LZ_BWD_G_X / 16,
                  G_ATT = LZ_BWD_G_ATT / 16, G_S1 = LZ_BWD_G_S1 / 16, G_S2 = LZ_BWD_G_S2 / 16, G_C1H = LZ_BWD_G_C1H / 16;
    switch (threadIdx.x >> 6) {   // wave-uniform
        case 0: lz_gw_product<7, 3, G_X, X_SIG0>(rec, M, part + T_X3 * 256); break;
        case 1: lz_gw_product<2, 4, G_ATT, X_A1>(rec, M, part + T_AUD1 * 256); break;
        case 2: lz_gw_product<4, 5, G_S1, X_SIG0>(rec, M, part + T_SIG0 * 256); break;
        case 3: lz_gw_product<3, 6, G_C1H, X_S2C>(rec, M, part + T_C1H * 256); break;
        case 4: lz_gw_product<2, 6, G_C1H + 3, X_S2C>(rec, M, part + (T_C1H + 18) * 256); break;
        default: lz_gw_product<4, 4, G_S2, X_S1>(rec, M, part + T_SIG1 * 256); break;
    }
}

__global__ void __launch_bounds__(64 * LZ_GW_WAVES)
lz_k_head_grad_w16(const uint32_t* __restrict__ rec, uint32_t M, float* __restrict__ parts) {
    float* part = parts + (size_t)blockIdx.x * (LZ_GW_TILES * 256);
    switch (threadIdx.x >> 6) {
        case 0: lz_gw_product16<7, 3, LZ_R16_G_X, LZ_R16_X_SIG0>(rec, M, part + T_X3 * 256); break;
        case 1: lz_gw_product16<2, 4, LZ_R16_G_ATT, LZ_R16_X_A1>(rec, M, part + T_AUD1 * 256); break;
        case 2: lz_gw_product16<4, 5, LZ_R16_G_S1, LZ_R16_X_SIG0>(rec, M, part + T_SIG0 * 256); break;
        case 3: lz_gw_product16<3, 6, LZ_R16_G_C1H, LZ_R16_X_S2C>(rec, M, part + T_C1H * 256); break;
        case 4: lz_gw_product16<2, 6, LZ_R16_G_C1H + 3, LZ_R16_X_S2C>(rec, M, part + (T_C1H + 18) * 256); break;
        default: lz_gw_product16<4, 4, LZ_R16_G_S2, LZ_R16_X_S1>(rec, M, part + T_SIG1 * 256); break;
    }
}

struct LzGwOut {
    float* dw[5];   // x3, aud1, sig1, sig0, c1h (tile order)
    int N[5], K[5], KBT[5], tile0[6];
    int16_t kmap[5][96];   // X tile u, column j of a product -> input feature k (column of dW), -1 = padding
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
    const int n = 16 * t + 4 * (lane >> 4) + r, k = o.kmap[job][16 * u + (lane & 15)];
    if (n < o.N[job] && k >= 0) o.dw[job][(size_t)n * o.K[job] + k] = v;
}

extern "C" size_t lz_triplane_head_grad_w_workspace(void) { return (size_t)LZ_GW_MAX_PARTS * LZ_GW_TILES * 256 * sizeof(float); }

static int lz_grad_w_launch(const void* rec, bool h16, uint32_t M, uint32_t k_sig0, float* dW_x3, float* dW_aud1, float* dW_sig0,
                            float* dW_sig1, float* dW_c1h, void* workspace, lz_stream_t stream) {
    LZ_REQUIRE(dW_x3 && dW_aud1 && dW_sig0 && dW_sig1 && dW_c1h && workspace, LZ_ERR_BAD_ARGUMENT, "head_grad_w: null tensor");
    LZ_REQUIRE(M == 0 || rec, LZ_ERR_BAD_ARGUMENT, "head_grad_w: null records");
    LZ_REQUIRE(k_sig0 == 68 || k_sig0 == 69, LZ_ERR_BAD_ARGUMENT, "head_grad_w: sigma_net.0 takes 68 or 69 inputs");
    const int n_cu = lz_cu_count();   // of the current device, per call (cached per device)
    const uint32_t groups = lz_div_up(M, h16 ? 16 : 8);
    uint32_t grid = 3u * (uint32_t)n_cu;   // the registers of a CU hold two workgroups; a third queued one evens out the tail
    if (grid > LZ_GW_MAX_PARTS) grid = LZ_GW_MAX_PARTS;
    if (grid > groups) grid = groups;
    hipStream_t st = lz_st(stream);
    float* parts = static_cast<float*>(workspace);
    if (grid > 0) {
        if (h16) hipLaunchKernelGGL(lz_k_head_grad_w16, dim3(grid), dim3(64 * LZ_GW_WAVES), 0, st, static_cast<const uint32_t*>(rec), M, parts);
        else hipLaunchKernelGGL(lz_k_head_grad_w, dim3(grid), dim3(64 * LZ_GW_WAVES), 0, st, static_cast<const float*>(rec), M, parts);
        LZ_CHECK_LAUNCH("head_grad_w");
    }
    return lz_head_grad_w_reduce_launch(parts, grid, h16, k_sig0, dW_x3, dW_aud1, dW_sig0, dW_sig1, dW_c1h, stream);
}

int lz_head_grad_w_reduce_launch(const float* parts, uint32_t grid, bool h16, uint32_t k_sig0, float* dW_x3, float* dW_aud1, float* dW_sig0,
                                 float* dW_sig1, float* dW_c1h, lz_stream_t stream) {
    hipStream_t st = lz_st(stream);
    LzGwOut o;
    float* dws[5] = {dW_x3, dW_aud1, dW_sig1, dW_sig0, dW_c1h};
    const int Ns[5] = {112, 32, 64, 64, 65}, Ks[5] = {36, 64, 64, (int)k_sig0, 84}, KB[5] = {3, 4, 4, 5, 6};
    const int t0[6] = {T_X3, T_AUD1, T_SIG1, T_SIG0, T_C1H, LZ_GW_TILES};
    for (int j = 0; j < 5; j++) { o.dw[j] = dws[j]; o.N[j] = Ns[j]; o.K[j] = Ks[j]; o.KBT[j] = KB[j]; }
    for (int j = 0; j < 6; j++) o.tile0[j] = t0[j];
    // column maps: f32 records keep the features of a slot in order; the f16 layout (LZ_R16_*) interleaves and regroups some
    for (int j = 0; j < 5; j++)
        for (int c = 0; c < 96; c++) o.kmap[j][c] = (int16_t)((c < KB[j] * 16 && c < Ks[j]) ? c : -1);
    if (h16) {
        auto encx = [](int c) { const int p = c >> 4, jj = c & 15; return 8 * (jj & 3) + 4 * p + (jj >> 2); };   // tiles 0, 1 of X_SIG0
        for (int job : {0, 3}) {   // x3 and sig0 read the sigma_net.0 input slot
            for (int c = 0; c < 96; c++) o.kmap[job][c] = -1;
            for (int c = 0; c < 32; c++) o.kmap[job][c] = (int16_t)encx(c);
            for (int q = 0; q < 4; q++) o.kmap[job][32 + 4 * q] = (int16_t)(32 + q);
        }
        if (k_sig0 == 69) o.kmap[3][32 + 1] = 68;
        for (int c = 0; c < 32; c++) o.kmap[3][48 + c] = (int16_t)(36 + c);
        for (int c = 64; c < 96; c++) o.kmap[4][c] = -1;   // c1h: [s2 64 | SH 16 as (4 r + q at column 4 q + r) | ind q at column 4 q of the last tile]
        for (int jj = 0; jj < 16; jj++) o.kmap[4][64 + jj] = (int16_t)(64 + 4 * (jj & 3) + (jj >> 2));
        for (int q = 0; q < 4; q++) o.kmap[4][80 + 4 * q] = (int16_t)(80 + q);
    }
    hipLaunchKernelGGL(lz_k_head_grad_w_reduce, dim3(LZ_GW_TILES), dim3(1024), 0, st, parts, grid, o);
    LZ_CHECK_LAUNCH("head_grad_w_reduce");
    return LZ_OK;
}

extern "C" int lz_triplane_head_grad_w(const float* rec, uint32_t M, uint32_t k_sig0, float* dW_x3, float* dW_aud1, float* dW_sig0,
                                       float* dW_sig1, float* dW_c1h, void* workspace, lz_stream_t stream) {
    return lz_grad_w_launch(rec, false, M, k_sig0, dW_x3, dW_aud1, dW_sig0, dW_sig1, dW_c1h, workspace, stream);
}

extern "C" int lz_triplane_head_grad_w_f16(const void* rec16, uint32_t M, uint32_t k_sig0, float* dW_x3, float* dW_aud1, float* dW_sig0,
                                           float* dW_sig1, float* dW_c1h, void* workspace, lz_stream_t stream) {
    LZ_REQUIRE(((uintptr_t)rec16 & 3u) == 0, LZ_ERR_BAD_ARGUMENT, "head_grad_w_f16: records must be 4-byte aligned");
    return lz_grad_w_launch(rec16, true, M, k_sig0, dW_x3, dW_aud1, dW_sig0, dW_sig1, dW_c1h, workspace, stream);
}
