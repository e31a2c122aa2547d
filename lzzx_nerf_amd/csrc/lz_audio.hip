// lz_audio.hip -- audio conditioning front-end: NeRFNetwork.encode_audio (nerf_triplane/network.py:226-240)
// = AudioNet (network.py:40-70: four stride-2 Conv1d + LeakyReLU over a 16-frame window, two Linear) on each of the seq_len
// windows, then AudioAttNet (network.py:9-37: five Conv1d over the window axis, Linear + softmax, attention-weighted sum).
// The reference spends 10-16 % of its frame time here on ~25 tiny cuDNN / cuBLAS launches.  The whole thing is a few
// MMAC with strictly sequential layers, so it is ONE workgroup: every layer is a loop over its outputs, activations ping-pong
// between two LDS buffers, weights stream from L2.  Arithmetic = explicit f32 fma chains (input channel outer, tap inner; bias
// added after the chain), restated by oracle/audio.py.
#include "lz_common.h"
#include "lzzx_detmath.h"

#define LZ_AUDIO_THREADS 1024
#define LZ_AUDIO_BUF 2048   // floats: largest activation is [8, 32, 8]

__device__ __forceinline__ float lz_lrelu(float v) { return v > 0.0f ? v : 0.02f * v; }   // nn.LeakyReLU(0.02)

// y[n][Cout][Lout] = lrelu(conv1d(x[n][Cin][Lin], w[Cout][Cin][3], stride, padding 1) + b)
__device__ __forceinline__ void lz_conv1d_k3(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ b,
                                             float* __restrict__ y, uint32_t n, uint32_t Cin, uint32_t Cout, uint32_t Lin, uint32_t stride) {
    const uint32_t Lout = (Lin - 1) / stride + 1;   // (Lin + 2 - 3) / stride + 1
    for (uint32_t idx = threadIdx.x; idx < n * Cout * Lout; idx += blockDim.x) {
        const uint32_t t = idx % Lout, o = (idx / Lout) % Cout, win = idx / (Lout * Cout);
        const float* xr = x + (size_t)win * Cin * Lin;
        const float* wr = w + (size_t)o * Cin * 3;
        float acc = 0.0f;
        for (uint32_t ci = 0; ci < Cin; ci++)
#pragma unroll
            for (int k = 0; k < 3; k++) {
                const int pos = (int)(t * stride) + k - 1;
                if (pos >= 0 && pos < (int)Lin) acc = lz_fmaf(wr[ci * 3 + k], xr[(size_t)ci * Lin + pos], acc);
            }
        y[idx] = lz_lrelu(acc + b[o]);
    }
    __syncthreads();
}

// y[n][N] = act(x[n][K] . w[N][K]^T + b)
__device__ __forceinline__ void lz_fc(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ b, float* __restrict__ y,
                                      uint32_t n, uint32_t K, uint32_t N, bool lrelu) {
    for (uint32_t idx = threadIdx.x; idx < n * N; idx += blockDim.x) {
        const uint32_t o = idx % N, r = idx / N;
        float acc = 0.0f;
        for (uint32_t k = 0; k < K; k++) acc = lz_fmaf(w[(size_t)o * K + k], x[(size_t)r * K + k], acc);
        acc += b[o];
        y[idx] = lrelu ? lz_lrelu(acc) : acc;
    }
    __syncthreads();
}

// First AudioNet layer for wide inputs (HuBERT: dim_in = 1024, a 3072-term dot product per output): one WAVE per output, lane l
// owns input channels l, l + 64, ... (fma chain, channel outer, tap inner), the 64 partial sums are combined by an xor-shuffle
// tree (32, 16, ..., 1), bias added last.  2048 outputs spread over the whole chip instead of 3072 dependent loads per lane of
// a single workgroup (0.88 ms -> a few microseconds).  y [n, 32, 8] goes to a small scratch buffer.
#define LZ_AUDIO_WIDE 128   // dim_in from which the wide first layer is used; part of the arithmetic contract (oracle/audio.py)
__global__ void __launch_bounds__(256)
lz_k_audio_conv1_wide(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ b, float* __restrict__ y, uint32_t n,
                      uint32_t Cin) {
    const uint32_t lane = threadIdx.x & 63, idx = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (idx >= n * 32 * 8) return;
    const uint32_t t = idx % 8, o = (idx / 8) % 32, win = idx / 256;
    const float* xr = x + (size_t)win * Cin * 16;
    const float* wr = w + (size_t)o * Cin * 3;
    float acc = 0.0f;
    for (uint32_t ci = lane; ci < Cin; ci += 64)
#pragma unroll
        for (int k = 0; k < 3; k++) {
            const int pos = (int)(t * 2) + k - 1;
            if (pos >= 0 && pos < 16) acc = lz_fmaf(wr[ci * 3 + k], xr[(size_t)ci * 16 + pos], acc);
        }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off, 64);
    if (lane == 0) y[idx] = lz_lrelu(acc + b[o]);
}

__global__ void __launch_bounds__(LZ_AUDIO_THREADS)
lz_k_audio_encode(lz_audio_params P, const float* __restrict__ a, float* __restrict__ enc_a, const float* __restrict__ conv1) {
    __shared__ float A[LZ_AUDIO_BUF], B[LZ_AUDIO_BUF], feat[8 * 64];
    const uint32_t n = P.n_win, da = P.dim_aud;
    // ---- AudioNet on every window: [n, dim_in, 16] -> [n, dim_aud]  (win_size 16: the slice x[:, :, 0:16] is the whole window)
    if (conv1) {   // first layer already done by lz_k_audio_conv1_wide
        for (uint32_t i = threadIdx.x; i < n * 32 * 8; i += blockDim.x) A[i] = conv1[i];
        __syncthreads();
    } else {
        lz_conv1d_k3(a, P.c_w[0], P.c_b[0], A, n, P.dim_in, 32, 16, 2);   // [n, 32, 8]
    }
    lz_conv1d_k3(A, P.c_w[1], P.c_b[1], B, n, 32, 32, 8, 2);          // [n, 32, 4]
    lz_conv1d_k3(B, P.c_w[2], P.c_b[2], A, n, 32, 64, 4, 2);          // [n, 64, 2]
    lz_conv1d_k3(A, P.c_w[3], P.c_b[3], B, n, 64, 64, 2, 2);          // [n, 64, 1]
    lz_fc(B, P.fc_w[0], P.fc_b[0], A, n, 64, 64, true);
    lz_fc(A, P.fc_w[1], P.fc_b[1], feat, n, 64, da, false);           // feat [n, dim_aud]
    if (!P.use_att) {
        for (uint32_t i = threadIdx.x; i < n * da; i += blockDim.x) enc_a[i] = feat[i];
        return;
    }
    // ---- AudioAttNet: y = feat^T [1, dim_aud, n] -> convs over the window axis -> [1, 1, n] -> Linear(n, n) -> softmax -> weighted sum
    for (uint32_t i = threadIdx.x; i < n * da; i += blockDim.x) A[(i % da) * n + i / da] = feat[i];   // permute(0, 2, 1)
    __syncthreads();
    lz_conv1d_k3(A, P.ac_w[0], P.ac_b[0], B, 1, da, 16, n, 1);
    lz_conv1d_k3(B, P.ac_w[1], P.ac_b[1], A, 1, 16, 8, n, 1);
    lz_conv1d_k3(A, P.ac_w[2], P.ac_b[2], B, 1, 8, 4, n, 1);
    lz_conv1d_k3(B, P.ac_w[3], P.ac_b[3], A, 1, 4, 2, n, 1);
    lz_conv1d_k3(A, P.ac_w[4], P.ac_b[4], B, 1, 2, 1, n, 1);          // B[0..n)
    lz_fc(B, P.al_w, P.al_b, A, 1, n, n, false);                      // logits A[0..n)
    if (threadIdx.x == 0) {   // softmax over n <= 8 values: max, exp, sum in index order, divide
        float m = A[0];
        for (uint32_t i = 1; i < n; i++) m = lz_fmaxf(m, A[i]);
        float s = 0.0f;
        for (uint32_t i = 0; i < n; i++) { B[i] = lz_expf(A[i] - m); s += B[i]; }
        for (uint32_t i = 0; i < n; i++) B[i] = B[i] / s;
    }
    __syncthreads();
    for (uint32_t c = threadIdx.x; c < da; c += blockDim.x) {         // torch.sum(y * x, dim=1)
        float acc = 0.0f;
        for (uint32_t t = 0; t < n; t++) acc = lz_fmaf(B[t], feat[t * da + c], acc);
        enc_a[c] = acc;
    }
}

extern "C" int lz_audio_encode(const lz_audio_params* p, const float* a, float* enc_a, void* workspace, lz_stream_t stream) {
    LZ_REQUIRE(p && a && enc_a, LZ_ERR_BAD_ARGUMENT, "audio_encode: null tensor");
    for (int i = 0; i < 4; i++) LZ_REQUIRE(p->c_w[i] && p->c_b[i], LZ_ERR_BAD_ARGUMENT, "audio_encode: missing encoder_conv weights");
    LZ_REQUIRE(p->fc_w[0] && p->fc_b[0] && p->fc_w[1] && p->fc_b[1], LZ_ERR_BAD_ARGUMENT, "audio_encode: missing encoder_fc1 weights");
    LZ_REQUIRE(p->n_win >= 1 && p->n_win <= 8 && p->dim_aud >= 1 && p->dim_aud <= 64 && p->dim_in >= 1, LZ_ERR_UNSUPPORTED,
               "audio_encode: 1..8 windows, dim_aud <= 64");
    if (p->use_att) {
        for (int i = 0; i < 5; i++) LZ_REQUIRE(p->ac_w[i] && p->ac_b[i], LZ_ERR_BAD_ARGUMENT, "audio_encode: missing attentionConvNet weights");
        LZ_REQUIRE(p->al_w && p->al_b, LZ_ERR_BAD_ARGUMENT, "audio_encode: missing attentionNet weights");
    }
    const float* conv1 = nullptr;
    if (p->dim_in >= LZ_AUDIO_WIDE) {
        LZ_REQUIRE(workspace, LZ_ERR_BAD_ARGUMENT, "audio_encode: workspace (n_win * 256 floats) required for dim_in >= %d", LZ_AUDIO_WIDE);
        const uint32_t outs = p->n_win * 32 * 8;
        hipLaunchKernelGGL(lz_k_audio_conv1_wide, dim3(lz_div_up(outs, 4)), dim3(256), 0, lz_st(stream), a, p->c_w[0], p->c_b[0],
                           reinterpret_cast<float*>(workspace), p->n_win, p->dim_in);
        conv1 = reinterpret_cast<const float*>(workspace);
    }
    hipLaunchKernelGGL(lz_k_audio_encode, dim3(1), dim3(LZ_AUDIO_THREADS), 0, lz_st(stream), *p, a, enc_a, conv1);
    LZ_CHECK_LAUNCH("audio_encode");
    return LZ_OK;
}
