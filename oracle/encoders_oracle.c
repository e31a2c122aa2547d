/*
 * oracle/encoders_oracle.c -- TEST INFRASTRUCTURE ONLY (CPU checker).
 *
 * CPU restatement of the reference's two closed-form encoders and of the pure-torch arithmetic of
 * the per-sample heads:
 *   spherical harmonics  kernel_sh / kernel_sh_backward   /root/reference/shencoder/src/shencoder.cu:27-355, :358-382
 *   frequency encoding   kernel_freq / kernel_freq_backward  freqencoder/src/freqencoder.cu:30-58, :63-94
 *   bias-free Linear(+ReLU) of MLP.forward                 nerf_triplane/network.py:73-94
 *
 * Parity status: SH is pinned by closed forms (scipy.special spherical harmonics on unit vectors,
 * finite differences for the Jacobian); freq by numpy sin/cos; Linear by the reference's own
 * torch modules imported on CPU (tests/golden/).  Against the CUDA binaries: "parity unpinned".
 *
 * Linear: a torch Linear fixes no summation order (cuBLAS / MKL choose their own), so the order is
 * an explicit argument: y[n] = fma-chain over j of x[korder[j]] * W[n][korder[j]], starting from 0.
 * `korder == NULL` is the natural order 0..K-1.  An entry of -1 is a padding slot (contributes
 * fma(0, 0, acc)).  The f32 MFMA used by the product is exactly such a chain.
 */
#include <math.h>
#include <stdint.h>
#include <stddef.h>
#include "lzzx_detmath.h"
#include "lzzx_sh_eval.h"

#define PI_F 3.141592653589793f

void lzo_sh_encode_forward(const float* inputs, float* outputs, uint32_t B, uint32_t degree, float* dy_dx) {
    const uint32_t C2 = degree * degree;
#pragma omp parallel for schedule(static)
    for (uint32_t b = 0; b < B; b++) {
        float* o = outputs + (size_t)b * C2;
        if (dy_dx) {
            float* dx = dy_dx + (size_t)b * 3 * C2; /* [B, 3, C2], shencoder.cu:127-129 */
            lz_sh_eval(inputs[b * 3], inputs[b * 3 + 1], inputs[b * 3 + 2], (int)degree, o, dx, dx + C2, dx + 2 * C2);
        } else {
            lz_sh_eval(inputs[b * 3], inputs[b * 3 + 1], inputs[b * 3 + 2], (int)degree, o, 0, 0, 0);
        }
    }
}

/* grad_inputs must be zero-filled by the caller (sphere_harmonics.py:49) */
void lzo_sh_encode_backward(const float* grad, const float* dy_dx, uint32_t B, uint32_t degree, float* grad_inputs) {
    const uint32_t C2 = degree * degree;
#pragma omp parallel for schedule(static)
    for (uint32_t t = 0; t < B * 3; t++) {
        const uint32_t b = t / 3, d = t - b * 3;
        const float* g = grad + (size_t)b * C2;
        const float* j = dy_dx + (size_t)b * 3 * C2 + (size_t)d * C2;
        float r = grad_inputs[t];
        for (uint32_t ch = 0; ch < C2; ch++) r = lz_fmaf(g[ch], j[ch], r);
        grad_inputs[t] = r;
    }
}

/* outputs: [B, C], C = D + 2*D*deg; layout [x, sin(2^0 x), cos(2^0 x), sin(2^1 x), ...] (freqencoder.cu:46-57) */
void lzo_freq_encode_forward(const float* inputs, uint32_t B, uint32_t D, uint32_t deg, uint32_t C, float* outputs) {
    (void)deg;
#pragma omp parallel for schedule(static)
    for (uint32_t b = 0; b < B; b++) {
        for (uint32_t c = 0; c < C; c++) {
            float v;
            if (c < D) v = inputs[(size_t)b * D + c];
            else {
                const uint32_t col = c / D - 1, d = c % D, freq = col / 2;
                const float phase = (float)(col % 2) * (PI_F / 2);
                v = lz_sinf(lz_scalbnf(inputs[(size_t)b * D + d], (int)freq) + phase);
            }
            outputs[(size_t)b * C + c] = v;
        }
    }
}

void lzo_freq_encode_backward(const float* grad, const float* outputs, uint32_t B, uint32_t D, uint32_t deg, uint32_t C, float* grad_inputs) {
#pragma omp parallel for schedule(static)
    for (uint32_t t = 0; t < B * D; t++) {
        const uint32_t b = t / D, d = t - b * D;
        const float* g = grad + (size_t)b * C;
        const float* o = outputs + (size_t)b * C;
        float r = g[d];
        g += D; o += D;
        for (uint32_t f = 0; f < deg; f++) {
            const float inner = lz_fmaf(g[d], o[D + d], -(g[D + d] * o[d]));
            r = lz_fmaf(lz_scalbnf(1.0f, (int)f), inner, r);
            g += 2 * D; o += 2 * D;
        }
        grad_inputs[t] = r;
    }
}

/* y[B,N] = x[B,K(ld ldx)] @ W[N,K(ld ldw)]^T in the given summation order, optional ReLU */
void lzo_linear(const float* x, uint32_t ldx, const float* W, uint32_t ldw, const int32_t* korder, uint32_t nk,
                uint32_t B, uint32_t N, int relu, float* y, uint32_t ldy) {
#pragma omp parallel for schedule(static)
    for (uint32_t b = 0; b < B; b++) {
        const float* xb = x + (size_t)b * ldx;
        for (uint32_t n = 0; n < N; n++) {
            const float* w = W + (size_t)n * ldw;
            float acc = 0.0f;
            for (uint32_t j = 0; j < nk; j++) {
                const int32_t k = korder ? korder[j] : (int32_t)j;
                acc = (k < 0) ? lz_fmaf(0.0f, 0.0f, acc) : lz_fmaf(w[k], xb[k], acc);
            }
            if (relu && !(acc > 0.0f)) acc = 0.0f;
            y[(size_t)b * ldy + n] = acc;
        }
    }
}

/* "lane-partial" order of the VALU layers of the fused head (csrc/lz_head.hip, lz_lane_dot): the four lanes q that share a
 * sample each run an fma chain, in the order t = 0.., r = 0..3, over the features f = 16 t + 4 q + r they hold; the partials
 * are combined as (p0 + p1) + (p2 + p3).  W == NULL: the weight row is x itself (sum of squares behind ||att||). */
void lzo_linear_lanes(const float* x, uint32_t ldx, const float* W, uint32_t ldw, uint32_t K, uint32_t B, uint32_t N, float* y,
                      uint32_t ldy) {
#pragma omp parallel for schedule(static)
    for (uint32_t b = 0; b < B; b++) {
        const float* xb = x + (size_t)b * ldx;
        for (uint32_t n = 0; n < N; n++) {
            const float* w = W ? W + (size_t)n * ldw : xb;
            float p[4];
            for (uint32_t q = 0; q < 4; q++) {
                float acc = 0.0f;
                for (uint32_t t = 0; 16 * t < K; t++)
                    for (uint32_t r = 0; r < 4; r++) {
                        const uint32_t f = 16 * t + 4 * q + r;
                        if (f < K) acc = lz_fmaf(w[f], xb[f], acc);
                    }
                p[q] = acc;
            }
            y[(size_t)b * ldy + n] = (p[0] + p[1]) + (p[2] + p[3]);
        }
    }
}

/* elementwise y = fma(a, b, c) in f32 (numpy has no fused multiply-add) */
void lzo_vec_fma(const float* a, const float* b, const float* c, float* y, size_t n) {
    for (size_t i = 0; i < n; i++) y[i] = lz_fmaf(a[i], b[i], c[i]);
}

/* elementwise deterministic transcendentals over a vector (op: 0 exp, 1 sigmoid, 2 softplus, 3 sin, 4 log) */
void lzo_vec_unary(int op, const float* x, float* y, size_t n) {
#pragma omp parallel for schedule(static)
    for (size_t i = 0; i < n; i++) {
        const float v = x[i];
        float r;
        switch (op) {
            case 0: r = lz_expf(v); break;
            case 1: r = lz_sigmoidf(v); break;
            case 2: r = lz_softplusf(v); break;
            case 3: r = lz_sinf(v); break;
            default: r = lz_logf(v); break;
        }
        y[i] = r;
    }
}

/* test-infrastructure knob: size of the OpenMP teams of this library (bench.py's cpu_baseline sets it to the host's CPU quota) */
#include <omp.h>
void lzo_set_threads(int n) { if (n > 0) omp_set_num_threads(n); }
int lzo_get_max_threads(void) { return omp_get_max_threads(); }
