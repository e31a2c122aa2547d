// lz_linear.hip -- tall-skinny bias-free Linear layers for the TRAINING path of the per-sample heads.
//
// The reference's MLP (nerf_triplane/network.py:73-94) is a stack of bias-free nn.Linear with K, N in {1..84} applied to
// M ~ 10^6..10^7 samples.  Library GEMMs are tuned for the opposite shape: on MI355X the rocBLAS f32 kernels torch picks
// spend 77 ms of a 103 ms training step (BASELINE cfg3) on these layers.  Here:
//   lz_linear_forward   Y[M,N] = act(maskX(X)[M,K] . W[N,K]^T)          (also the data gradient: dX = (dY . mask) . (W^T)^T)
//   lz_linear_grad_w    dW[N,K] += (dY . mask)[M,N]^T . X[M,K]           (reduction over the M samples inside the kernel)
// both on v_mfma_f32_16x16x4_f32 with operands loaded straight from the row-major activations (every element is read once,
// as part of a 64-byte row segment), f32 accumulate, no LDS staging of activations.  `mask` (optional, same shape as the
// masked operand) implements the ReLU backward: the operand is zeroed where mask <= 0.  Leading dimensions are explicit so
// callers can read and write column slices of wider buffers.
//
// Forward orientation: D[sample, n] = A (16 samples x 4 k) . B (4 k x 16 n).  A lane (s = l & 15, kk = l >> 4) holds
// X[s][16 jj + 4 kk + c] for the MFMA step (jj, c); B holds W[n][same k]: the k order inside a block of 16 is permuted the
// same way on both operands, which a sum over k does not care about (up to rounding order).
// grad_w orientation: D[n, k] = A (16 n x 4 samples) . B (4 samples x 16 k); lane (i = l & 15, kk) holds dY[4 j + kk][n-tile + i]
// resp. X[4 j + kk][k-tile + i] for step j.  Each wave accumulates its share of the samples in registers, the workgroup reduces
// through LDS, one contiguous float atomic per element and workgroup lands in dW (order of the M-reduction is free).
#include "lz_common.h"

typedef float lz_f4 __attribute__((ext_vector_type(4)));

#define LZ_LIN_MAXB 8   // up to 128 features on either side

__device__ __forceinline__ float lz_lin_ld(const float* __restrict__ p, const float* __restrict__ mask, size_t off, bool ok) {
    if (!ok) return 0.0f;
    const float v = p[off];
    if (mask) return mask[off] > 0.0f ? v : 0.0f;
    return v;
}

// one wave = 16 samples x all N outputs per pass; W (N x K, zero-padded to 16-blocks) lives in LDS in B-fragment order
__global__ void __launch_bounds__(256)
lz_k_linear_forward(const float* __restrict__ X, uint32_t ldx, const float* __restrict__ mask, const float* __restrict__ W, uint32_t ldw,
                    float* __restrict__ Y, uint32_t ldy, uint32_t M, uint32_t K, uint32_t N, int relu_out) {
    extern __shared__ float wl[];   // [NB][KB][4 c][64 lanes]
    const uint32_t KB = (K + 15) / 16, NB = (N + 15) / 16;
    for (uint32_t i = threadIdx.x; i < NB * KB * 256; i += blockDim.x) {
        const uint32_t lane = i & 63, c = (i >> 6) & 3, blk = i >> 8;
        const uint32_t t = blk / KB, jj = blk - t * KB;
        const uint32_t n = 16 * t + (lane & 15), k = 16 * jj + 4 * (lane >> 4) + c;
        wl[i] = (n < N && k < K) ? W[(size_t)n * ldw + k] : 0.0f;
    }
    __syncthreads();
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6, s = lane & 15, kk = lane >> 4;
    const uint32_t n_groups = (M + 15) / 16;
    const uint32_t waves_total = gridDim.x * (blockDim.x >> 6);
    for (uint32_t g = blockIdx.x * (blockDim.x >> 6) + wave; g < n_groups; g += waves_total) {
        const uint32_t row = g * 16 + s;
        const bool row_ok = row < M;
        lz_f4 acc[LZ_LIN_MAXB];
#pragma unroll
        for (int t = 0; t < LZ_LIN_MAXB; t++) acc[t] = lz_f4{0, 0, 0, 0};
        for (uint32_t jj = 0; jj < KB; jj++) {
            float a[4];
#pragma unroll
            for (uint32_t c = 0; c < 4; c++) {
                const uint32_t k = 16 * jj + 4 * kk + c;
                a[c] = lz_lin_ld(X, mask, (size_t)row * ldx + k, row_ok && k < K);
            }
#pragma unroll
            for (int t = 0; t < LZ_LIN_MAXB; t++) {
                if ((uint32_t)t < NB) {
                    const float* frag = wl + ((size_t)(t * KB + jj) * 4) * 64 + lane;
#pragma unroll
                    for (uint32_t c = 0; c < 4; c++) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[c], frag[c * 64], acc[t], 0, 0, 0);
                }
            }
        }
        // D tile t: lane (n = l & 15, q = l >> 4), reg r -> sample 4 q + r, output 16 t + n
#pragma unroll
        for (int t = 0; t < LZ_LIN_MAXB; t++) {
            if ((uint32_t)t < NB) {
                const uint32_t n = 16 * t + s;
#pragma unroll
                for (uint32_t r = 0; r < 4; r++) {
                    const uint32_t orow = g * 16 + 4 * kk + r;
                    if (orow < M && n < N) {
                        float v = acc[t][r];
                        if (relu_out) v = v > 0.0f ? v : 0.0f;
                        Y[(size_t)orow * ldy + n] = v;
                    }
                }
            }
        }
    }
}

extern "C" int lz_linear_forward(const float* X, uint32_t ldx, const float* mask, const float* W, uint32_t ldw, float* Y, uint32_t ldy,
                                 uint32_t M, uint32_t K, uint32_t N, int relu_out, lz_stream_t stream) {
    if (M == 0 || N == 0) return LZ_OK;
    LZ_REQUIRE(X && W && Y, LZ_ERR_BAD_ARGUMENT, "linear_forward: null tensor");
    LZ_REQUIRE(K >= 1 && K <= 16 * LZ_LIN_MAXB && N <= 16 * LZ_LIN_MAXB, LZ_ERR_UNSUPPORTED, "linear_forward: K and N must be <= %d", 16 * LZ_LIN_MAXB);
    LZ_REQUIRE(ldx >= K && ldw >= K && ldy >= N, LZ_ERR_BAD_ARGUMENT, "linear_forward: leading dimension smaller than the row");
    const uint32_t KB = (K + 15) / 16, NB = (N + 15) / 16;
    const size_t smem = (size_t)NB * KB * 256 * sizeof(float);
    const uint32_t groups = lz_div_up(M, 16);
    uint32_t grid = lz_div_up(groups, 4);
    if (grid > 2048) grid = 2048;   // 8 workgroups per CU, grid-stride over the sample groups
    hipLaunchKernelGGL(lz_k_linear_forward, dim3(grid), dim3(256), smem, lz_st(stream), X, ldx, mask, W, ldw, Y, ldy, M, K, N, relu_out);
    LZ_CHECK_LAUNCH("linear_forward");
    return LZ_OK;
}

// dW[N,K] += (dY . mask)^T . X ; NB * KB <= LZ_LIN_MAXT accumulator tiles per wave
#define LZ_LIN_MAXT 24
template <int NBT, int KBT>
__global__ void __launch_bounds__(256)
lz_k_linear_grad_w(const float* __restrict__ dY, uint32_t ldd, const float* __restrict__ mask, const float* __restrict__ X, uint32_t ldx,
                   float* __restrict__ dW, uint32_t ldw, uint32_t M, uint32_t K, uint32_t N) {
    __shared__ float red[NBT * KBT * 256];
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6, i = lane & 15, kk = lane >> 4;
    lz_f4 acc[NBT][KBT];
#pragma unroll
    for (int t = 0; t < NBT; t++)
#pragma unroll
        for (int u = 0; u < KBT; u++) acc[t][u] = lz_f4{0, 0, 0, 0};
    const uint32_t n_groups = (M + 15) / 16;
    const uint32_t waves_total = gridDim.x * (blockDim.x >> 6);
    for (uint32_t g = blockIdx.x * (blockDim.x >> 6) + wave; g < n_groups; g += waves_total) {
#pragma unroll
        for (uint32_t j = 0; j < 4; j++) {
            const uint32_t row = g * 16 + 4 * j + kk;
            const bool row_ok = row < M;
            float a[NBT], b[KBT];
#pragma unroll
            for (int t = 0; t < NBT; t++) a[t] = lz_lin_ld(dY, mask, (size_t)row * ldd + 16 * t + i, row_ok && 16 * t + i < N);
#pragma unroll
            for (int u = 0; u < KBT; u++) b[u] = lz_lin_ld(X, nullptr, (size_t)row * ldx + 16 * u + i, row_ok && 16 * u + i < K);
#pragma unroll
            for (int t = 0; t < NBT; t++)
#pragma unroll
                for (int u = 0; u < KBT; u++) acc[t][u] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[t], b[u], acc[t][u], 0, 0, 0);
        }
    }
    // workgroup reduction through LDS (waves 1..3 add into wave 0's image), then one atomic per element
    for (uint32_t w = 0; w < (blockDim.x >> 6); w++) {
        if (wave == w) {
#pragma unroll
            for (int t = 0; t < NBT; t++)
#pragma unroll
                for (int u = 0; u < KBT; u++)
#pragma unroll
                    for (int r = 0; r < 4; r++) {
                        float* p = red + ((t * KBT + u) * 4 + r) * 64 + lane;
                        *p = (w == 0) ? acc[t][u][r] : *p + acc[t][u][r];
                    }
        }
        __syncthreads();
    }
    // element (tile t,u; reg r; lane): D row = 4 (lane >> 4) + r -> n = 16 t + row, col = lane & 15 -> k = 16 u + col
    for (uint32_t e = threadIdx.x; e < NBT * KBT * 256; e += blockDim.x) {
        const uint32_t l = e & 63, r = (e >> 6) & 3, tu = e >> 8;
        const uint32_t t = tu / KBT, u = tu - t * KBT;
        const uint32_t n = 16 * t + 4 * (l >> 4) + r, k = 16 * u + (l & 15);
        const float v = red[e];
        if (n < N && k < K && v != 0.0f) atomicAdd(dW + (size_t)n * ldw + k, v);
    }
}

template <int NBT>
static int lz_linear_grad_w_k(uint32_t KB, dim3 grid, hipStream_t st, const float* dY, uint32_t ldd, const float* mask, const float* X,
                              uint32_t ldx, float* dW, uint32_t ldw, uint32_t M, uint32_t K, uint32_t N) {
#define LZ_GW(KBT)                                                                                                            \
    case KBT:                                                                                                                 \
        if constexpr (NBT * KBT <= LZ_LIN_MAXT) {                                                                             \
            hipLaunchKernelGGL((lz_k_linear_grad_w<NBT, KBT>), grid, dim3(256), 0, st, dY, ldd, mask, X, ldx, dW, ldw, M, K, N); \
            return LZ_OK;                                                                                                     \
        }                                                                                                                     \
        break;
    switch (KB) { LZ_GW(1) LZ_GW(2) LZ_GW(3) LZ_GW(4) LZ_GW(5) LZ_GW(6) default: break; }
#undef LZ_GW
    lz_set_error("linear_grad_w: ceil(N/16) * ceil(K/16) must be <= %d, K <= 96, N <= 128", LZ_LIN_MAXT);
    return LZ_ERR_UNSUPPORTED;
}

extern "C" int lz_linear_grad_w(const float* dY, uint32_t ldd, const float* mask, const float* X, uint32_t ldx, float* dW, uint32_t ldw,
                                uint32_t M, uint32_t K, uint32_t N, lz_stream_t stream) {
    if (M == 0 || N == 0 || K == 0) return LZ_OK;
    LZ_REQUIRE(dY && X && dW, LZ_ERR_BAD_ARGUMENT, "linear_grad_w: null tensor");
    LZ_REQUIRE(ldd >= N && ldx >= K && ldw >= K, LZ_ERR_BAD_ARGUMENT, "linear_grad_w: leading dimension smaller than the row");
    const uint32_t KB = (K + 15) / 16, NB = (N + 15) / 16;
    const uint32_t groups = lz_div_up(M, 16);
    uint32_t g = lz_div_up(groups, 4 * 8);   // >= 8 sample groups per wave before paying for the reduction
    if (g > 1536) g = 1536;   // ~6 workgroups per CU: the loads feed the MFMAs directly, occupancy is what hides their latency
    if (g < 1) g = 1;
    const dim3 grid(g);
    hipStream_t st = lz_st(stream);
    int rc;
    switch (NB) {
        case 1: rc = lz_linear_grad_w_k<1>(KB, grid, st, dY, ldd, mask, X, ldx, dW, ldw, M, K, N); break;
        case 2: rc = lz_linear_grad_w_k<2>(KB, grid, st, dY, ldd, mask, X, ldx, dW, ldw, M, K, N); break;
        case 3: rc = lz_linear_grad_w_k<3>(KB, grid, st, dY, ldd, mask, X, ldx, dW, ldw, M, K, N); break;
        case 4: rc = lz_linear_grad_w_k<4>(KB, grid, st, dY, ldd, mask, X, ldx, dW, ldw, M, K, N); break;
        case 5: rc = lz_linear_grad_w_k<5>(KB, grid, st, dY, ldd, mask, X, ldx, dW, ldw, M, K, N); break;
        case 6: rc = lz_linear_grad_w_k<6>(KB, grid, st, dY, ldd, mask, X, ldx, dW, ldw, M, K, N); break;
        case 7: rc = lz_linear_grad_w_k<7>(KB, grid, st, dY, ldd, mask, X, ldx, dW, ldw, M, K, N); break;
        case 8: rc = lz_linear_grad_w_k<8>(KB, grid, st, dY, ldd, mask, X, ldx, dW, ldw, M, K, N); break;
        default: lz_set_error("linear_grad_w: N must be <= 128"); return LZ_ERR_UNSUPPORTED;
    }
    if (rc != LZ_OK) return rc;
    LZ_CHECK_LAUNCH("linear_grad_w");
    return LZ_OK;
}
