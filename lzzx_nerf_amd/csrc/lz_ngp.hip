// lz_ngp.hip -- BASELINE cfg2: a generic hash-grid NeRF ("256 x 256, 128 samples / ray, hash-grid L = 16 F = 2, forward render") on the
// operators of encoding.get_encoder (encoding.py:6-37): hashgrid (D 3, L 16, C 2, T 2^19; gridencoder.cu:75-223) -> sigma MLP 32-64-16,
// SH(4) of the view direction + 15 geometry features -> colour MLP 31-64-3, bias-free Linear + ReLU (network.py:73-94), sigma = exp,
// rgb = sigmoid -- and the reference's inference loop around it (renderer.py:495-548).
//
// MI355X design.  The hash-grid gather wants to run LEVEL-major: with level the slow launch dimension the whole chip reads one or two
// <= 4 MB levels at a time and they stay resident in the 4 MB L2 of every XCD (69 % of the HBM roofline in march order), whereas a kernel
// in which every wave touches all 16 levels at once (a persistent frame kernel like lz_frame.hip) has the whole 49 MB table as its L2
// working set and was measured at 16 %.  So the gather stays the level-major pass of lz_grid.hip and everything else is fused around it:
//     lz_loop_march            state advance + compaction + march of the survivors                    (lz_raymarch.hip)
//     lz_k_grid_forward_lmp    level-major gather, TILED output, rows bounded by the device-side sample count, no untile pass
//     lz_k_ngp_head            per 16-sample slice: B operands straight from the tiles -> sigma MLP -> SH(4) -> colour MLP on
//                              v_mfma_f32_16x16x4_f32 (96 MFMAs), sigma / rgb out -- replaces 4 Linear launches + cat + activations + copies
//     lz_loop_composite_plain  accumulate, kill, count survivors
// four launches per iteration of the reference's loop, enqueued back to back by lz_ngp_loop_run with the loop state in device memory.
//
// Arithmetic: every Linear is an fma chain in MFMA k order (oracle/ngp.py spells the order per layer); the first layer's k order follows
// the tiled layout (lane q of a sample reads levels q, q + 4, q + 8, q + 12, both channels: one 8-byte load each), the hidden layers
// consume the previous accumulator tile in place ("chained" order, as lz_head.hip).
#include "lz_common.h"
#include "lzzx_detmath.h"
#include "lzzx_sh_eval.h"
#include "lz_head_layers.h"
#include <hip/hip_fp16.h>

// fragment (ks, ft) of a layer: 64 floats, lane l = W[16 ft + (l & 15)][k(ks, l >> 4)]
#define LZN_S1 0       // 32 -> 64: 8 k-steps x 4 tiles
#define LZN_S2 32      // 64 -> 16: 16 x 1
#define LZN_C1 48      // 32 slots (SH 16 | sigma_net output 16, slot of its row 0 weighted 0) -> 64: 8 x 4
#define LZN_C2 80      // 64 -> 3 (one tile, rows 3..15 zero): 16 x 1
static_assert(LZN_C2 + 16 == LZ_NGP_FRAGS, "fragment count mismatch with the header");
#ifndef LZN_WG_PER_CU
#define LZN_WG_PER_CU 3u
#endif
#ifndef LZN_T
#define LZN_T 1     // 16-sample slices a wave takes through the head together (2: every fragment read feeds two MFMAs, but 179 registers = two waves per SIMD instead of three: the frame takes the same 2.15 ms)
#endif

struct LzNgpK {
    const float* packed;
    const void* feats;
    const float* dirs;
    const int* count;
    float* sigmas;
    float* rgbs;
    uint32_t rows;        // rows of feats / dirs (the encoder's B: it fixes the packing of the last, partial tile)
};

// FEAT: 0 = row-major f32 [rows, 32]; 1 = tiled f32 (lz_grid_encode_forward_tiled); 2 = tiled f16
#ifndef LZN_WG
#define LZN_WG 512     /* threads per workgroup: the 24 KB weight image is copied once per workgroup (256 -> 512 and 4 -> 8 passes per wave: cfg2 frame 2.07 -> 2.03 ms) */
#endif
#ifndef LZN_PASSES
#define LZN_PASSES 8   /* passes per wave the grid is sized for when there are rows enough for two workgroups per CU; halved until there are */
#endif
template <int FEAT>
__global__ void __launch_bounds__(LZN_WG) lz_k_ngp_head(LzNgpK P) {
    __shared__ __align__(16) float wl[LZ_NGP_FRAGS * 64];
    {
        const float4* src = reinterpret_cast<const float4*>(P.packed);
        float4* dst = reinterpret_cast<float4*>(wl);
        for (uint32_t i = threadIdx.x; i < LZ_NGP_FRAGS * 16; i += blockDim.x) dst[i] = src[i];
    }
    __syncthreads();
    uint32_t rows = P.rows;
    if (P.count) {
        const int c = *P.count;
        rows = c < 0 ? 0u : ((uint32_t)c < rows ? (uint32_t)c : rows);
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, s = lane & 15, q = lane >> 4;
    const uint32_t n_slices = (rows + 15u) / 16u, Tn = LZ_GRID_TILE_ROWS;
    // T slices per pass: every A fragment is read from LDS once and feeds T MFMAs, and the T accumulation chains interleave (the 16-deep
    // chains of the two 64 -> N layers are dependent MFMAs otherwise)
    constexpr int T = LZN_T;
    // the inputs of a pass (features as the B operands of sigma_net.0: levels q, q + 4, q + 8, q + 12 of sample s, both channels; direction)
    struct In { uint32_t row[T]; bool valid[T]; float b1[T][8], dx[T], dy[T], dz[T]; };
    auto load = [&](uint32_t slice0, In& I) {
#pragma unroll
        for (int u = 0; u < T; u++) {
            I.row[u] = (slice0 + u) * 16u + (uint32_t)s;
            I.valid[u] = I.row[u] < rows;
            const uint32_t r = I.valid[u] ? I.row[u] : rows - 1u;
            if constexpr (FEAT == 0) {
                const float2* f = reinterpret_cast<const float2*>(reinterpret_cast<const float*>(P.feats) + (size_t)r * 32);
#pragma unroll
                for (int i = 0; i < 4; i++) { const float2 v = f[q + 4 * i]; I.b1[u][2 * i] = v.x; I.b1[u][2 * i + 1] = v.y; }
            } else {
                const uint32_t tile = r / Tn, t = r - tile * Tn, b0 = tile * Tn, n = (P.rows - b0 < Tn) ? P.rows - b0 : Tn;
                if constexpr (FEAT == 1) {
                    const float2* f = reinterpret_cast<const float2*>(reinterpret_cast<const float*>(P.feats) + (size_t)b0 * 32);
#pragma unroll
                    for (int i = 0; i < 4; i++) { const float2 v = f[(size_t)(q + 4 * i) * n + t]; I.b1[u][2 * i] = v.x; I.b1[u][2 * i + 1] = v.y; }
                } else {
                    const __half2* f = reinterpret_cast<const __half2*>(reinterpret_cast<const __half*>(P.feats) + (size_t)b0 * 32);
#pragma unroll
                    for (int i = 0; i < 4; i++) { const float2 v = __half22float2(f[(size_t)(q + 4 * i) * n + t]); I.b1[u][2 * i] = v.x; I.b1[u][2 * i + 1] = v.y; }
                }
            }
            I.dx[u] = P.dirs[(size_t)r * 3]; I.dy[u] = P.dirs[(size_t)r * 3 + 1]; I.dz[u] = P.dirs[(size_t)r * 3 + 2];
        }
    };
    const uint32_t stride = gridDim.x * (LZN_WG / 64u) * T;
    uint32_t slice0 = (blockIdx.x * (LZN_WG / 64u) + (uint32_t)wave) * T;
    if (rows == 0 || slice0 >= n_slices) return;
    In nxt;
    load(slice0, nxt);
    for (; slice0 < n_slices; slice0 += stride) {
        // the next pass's inputs are requested before this pass's 96 MFMAs (a wave's pass otherwise starts with a round trip to the tiles)
        const In cur = nxt;
        if (slice0 + stride < n_slices) load(slice0 + stride, nxt);
        const uint32_t (&row)[T] = cur.row;
        const bool (&valid)[T] = cur.valid;
        const float (&b1)[T][8] = cur.b1;
        const float (&dx)[T] = cur.dx, (&dy)[T] = cur.dy, (&dz)[T] = cur.dz;
        // ---------------- sigma_net: 32 -> 64 (ReLU) -> 16 ----------------
        float h1[T][16];
        {
            lz_f4 acc[T][4];
#pragma unroll
            for (int u = 0; u < T; u++)
#pragma unroll
                for (int ft = 0; ft < 4; ft++) acc[u][ft] = lz_f4{0, 0, 0, 0};
#pragma unroll
            for (int ks = 0; ks < 8; ks++)
#pragma unroll
                for (int ft = 0; ft < 4; ft++) {
                    const float a = wl[(LZN_S1 + ks * 4 + ft) * 64 + lane];
#pragma unroll
                    for (int u = 0; u < T; u++) acc[u][ft] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b1[u][ks], acc[u][ft], 0, 0, 0);
                }
#pragma unroll
            for (int u = 0; u < T; u++)
#pragma unroll
                for (int ft = 0; ft < 4; ft++)
#pragma unroll
                    for (int rr = 0; rr < 4; rr++) h1[u][4 * ft + rr] = lz_relu(acc[u][ft][rr]);
        }
        lz_f4 h[T];          // h[rr] = output 4 q + rr of sigma_net: row 0 -> sigma, rows 1..15 -> geometry features
#pragma unroll
        for (int u = 0; u < T; u++) h[u] = lz_f4{0, 0, 0, 0};
#pragma unroll
        for (int ks = 0; ks < 16; ks++) {
            const float a = wl[(LZN_S2 + ks) * 64 + lane];
#pragma unroll
            for (int u = 0; u < T; u++) h[u] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, h1[u][ks], h[u], 0, 0, 0);
        }
        // ---------------- colour_net: [SH(4) of the direction | geometry] -> 64 (ReLU) -> 3 ----------------
        float c1[T][16];
        {
            float shq[T][4];     // SH components 4 ks + q of this lane's sample
#pragma unroll
            for (int u = 0; u < T; u++) {
                float sh[16];
                lz_sh_eval(dx[u], dy[u], dz[u], 4, sh, nullptr, nullptr, nullptr);
#pragma unroll
                for (int ks = 0; ks < 4; ks++) shq[u][ks] = q == 0 ? sh[4 * ks] : (q == 1 ? sh[4 * ks + 1] : (q == 2 ? sh[4 * ks + 2] : sh[4 * ks + 3]));
            }
            lz_f4 acc[T][4];
#pragma unroll
            for (int u = 0; u < T; u++)
#pragma unroll
                for (int ft = 0; ft < 4; ft++) acc[u][ft] = lz_f4{0, 0, 0, 0};
#pragma unroll
            for (int ks = 0; ks < 8; ks++)
#pragma unroll
                for (int ft = 0; ft < 4; ft++) {
                    const float a = wl[(LZN_C1 + ks * 4 + ft) * 64 + lane];
#pragma unroll
                    for (int u = 0; u < T; u++)      // SH component 4 ks + q, then sigma_net output 4 q + (ks - 4)
                        acc[u][ft] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, ks < 4 ? shq[u][ks] : h[u][ks - 4], acc[u][ft], 0, 0, 0);
                }
#pragma unroll
            for (int u = 0; u < T; u++)
#pragma unroll
                for (int ft = 0; ft < 4; ft++)
#pragma unroll
                    for (int rr = 0; rr < 4; rr++) c1[u][4 * ft + rr] = lz_relu(acc[u][ft][rr]);
        }
        lz_f4 c[T];
#pragma unroll
        for (int u = 0; u < T; u++) c[u] = lz_f4{0, 0, 0, 0};
#pragma unroll
        for (int ks = 0; ks < 16; ks++) {
            const float a = wl[(LZN_C2 + ks) * 64 + lane];
#pragma unroll
            for (int u = 0; u < T; u++) c[u] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, c1[u][ks], c[u], 0, 0, 0);
        }
#pragma unroll
        for (int u = 0; u < T; u++)
            if (q == 0 && valid[u]) {
                P.sigmas[row[u]] = lz_expf(h[u][0]);
                P.rgbs[(size_t)row[u] * 3] = lz_sigmoidf(c[u][0]);
                P.rgbs[(size_t)row[u] * 3 + 1] = lz_sigmoidf(c[u][1]);
                P.rgbs[(size_t)row[u] * 3 + 2] = lz_sigmoidf(c[u][2]);
            }
    }
}

extern "C" int lz_ngp_head_forward(const float* packed, const void* feats, int feat_layout, const float* dirs, uint32_t rows, const int32_t* count,
                                   float* sigmas, float* rgbs, lz_stream_t stream) {
    if (rows == 0) return LZ_OK;
    LZ_REQUIRE(packed && feats && dirs && sigmas && rgbs, LZ_ERR_BAD_ARGUMENT, "ngp_head_forward: null tensor");
    LZ_REQUIRE(feat_layout >= 0 && feat_layout <= 2, LZ_ERR_BAD_ARGUMENT, "ngp_head_forward: feat_layout 0 (row-major f32), 1 (tiled f32) or 2 (tiled f16)");
    LzNgpK K{packed, feats, dirs, count, sigmas, rgbs, rows};
    uint32_t passes = LZN_PASSES;      // the reference schedule's iterations hold one sample per ray: few rows, keep every CU busy
    while (passes > 1 && lz_div_up(rows, 16 * (LZN_WG / 64) * passes * LZN_T) < 512) passes >>= 1;
    uint32_t grid = lz_div_up(rows, 16 * (LZN_WG / 64) * passes * LZN_T);
    // every workgroup stages the 24 KB of weights once: as many workgroups as the chip holds at a time (3 waves per SIMD = 3 of these 4-wave
    // workgroups per CU), each looping over its share of the slices, not one per 16 slices
    const uint32_t cap = (uint32_t)lz_cu_count() * LZN_WG_PER_CU;
    grid = grid < 1 ? 1 : (grid > cap ? cap : grid);
    hipStream_t st = lz_st(stream);
    if (feat_layout == 0) hipLaunchKernelGGL((lz_k_ngp_head<0>), dim3(grid), dim3(LZN_WG), 0, st, K);
    else if (feat_layout == 1) hipLaunchKernelGGL((lz_k_ngp_head<1>), dim3(grid), dim3(LZN_WG), 0, st, K);
    else hipLaunchKernelGGL((lz_k_ngp_head<2>), dim3(grid), dim3(LZN_WG), 0, st, K);
    LZ_CHECK_LAUNCH("ngp_head_forward");
    return LZ_OK;
}

// enqueue `n_iterations` iterations of the reference's inference loop (renderer.py:503-548) around the hash-grid NeRF: march -> gather ->
// head -> composite, four launches each, no host round trip; iterations past the end of the frame are no-ops on the device
extern "C" int lz_ngp_loop_run(const lz_frame_ngp* f, uint32_t parity, uint32_t n_iterations, lz_stream_t stream) {
    LZ_REQUIRE(f, LZ_ERR_BAD_ARGUMENT, "ngp_loop_run: null");
    LZ_REQUIRE(f->state && f->workspace && f->packed && f->embeddings && f->offsets, LZ_ERR_BAD_ARGUMENT, "ngp_loop_run: incomplete lz_frame_ngp");
    if (f->N == 0) return LZ_OK;                 // no ray: nothing to enqueue
    LZ_REQUIRE(f->rays_alive[0] && f->rays_alive[1] && f->feats, LZ_ERR_BAD_ARGUMENT, "ngp_loop_run: incomplete lz_frame_ngp");
    uint32_t cur = parity & 1u;
    const int32_t* count = reinterpret_cast<const int32_t*>(f->state) + LZ_LOOP_NEXT + 2;  // n_samples of the iteration in flight
    const uint32_t rows = f->sample_budget > f->N ? f->sample_budget : f->N;  // capacity of the sample buffers
    for (uint32_t it = 0; it < n_iterations; it++) {
        const uint32_t nxt = cur ^ 1u;
        int rc = lz_loop_march(f->state, f->N, f->sample_budget, f->n_step_cap, f->rays_alive[cur], f->rays_alive[nxt], f->workspace, f->rays_t,
                               f->rays_o, f->rays_d, f->bound, f->dt_gamma, f->max_steps, f->C, f->H, f->grid, f->nears, f->fars, f->xyzs,
                               f->dirs, f->deltas, f->ray_counts, stream);
        if (rc != LZ_OK) return rc;
        rc = lz_grid_encode_forward_tiled(f->xyzs, f->embeddings, f->offsets, f->feats, rows, count, f->bound, 3, 2, f->enc_L, f->enc_S, f->enc_H, 0,
                                          0, f->emb_f16, stream);
        if (rc != LZ_OK) return rc;
        rc = lz_ngp_head_forward(f->packed, f->feats, f->emb_f16 ? 2 : 1, f->dirs, rows, count, f->sigmas, f->rgbs, stream);
        if (rc != LZ_OK) return rc;
        rc = lz_loop_composite_plain(f->state, f->N, f->T_thresh, f->rays_alive[nxt], f->rays_t, f->sigmas, f->rgbs, f->deltas, f->weights_sum,
                                     f->depth, f->image, f->workspace, stream);
        if (rc != LZ_OK) return rc;
        cur = nxt;
    }
    return LZ_OK;
}
