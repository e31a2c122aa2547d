// lz_encoders.hip -- spherical-harmonics and frequency encoders for gfx950, plus the library's
// error/bookkeeping entry points.
//
// Replaces shencoder/src/shencoder.cu:27-382 and freqencoder/src/freqencoder.cu:30-94.
//   * SH: the polynomials come from include/lzzx_sh_eval.h (generated from the Legendre recurrences,
//     factored as T_m(x,y) * Z_l^m(z)); degree is a template parameter so the unused orders vanish and
//     every output index is a compile-time constant (registers, no scratch).  One lane per direction;
//     the degree^2 outputs of a lane are contiguous, so stores go out as dwordx4.
//   * freq: one lane per OUTPUT element (coalesced 4 B stores), sin via lz_sinf (deterministic; the
//     reference's __sinf is an NVIDIA fast intrinsic).  Memory-bound: 8 B in, 136 B out per sample at D=2,deg=8.
#include <stdarg.h>

#include "lz_common.h"
#include "lzzx_detmath.h"
#include "lzzx_sh_eval.h"

// ------------------------------------------------------------------------------------------------
// library bookkeeping
// ------------------------------------------------------------------------------------------------
static thread_local char g_lz_err[512] = "";

void lz_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_lz_err, sizeof(g_lz_err), fmt, ap);
    va_end(ap);
}

extern "C" const char* lz_last_error(void) { return g_lz_err; }
extern "C" int lz_abi_version(void) { return 11; }

extern "C" int lz_device_ok(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) {
        (void)hipGetLastError();
        return 0;
    }
    hipDeviceProp_t prop;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return 0;
    return strncmp(prop.gcnArchName, "gfx950", 6) == 0 ? 1 : 0;
}

// ------------------------------------------------------------------------------------------------
// spherical harmonics
// ------------------------------------------------------------------------------------------------
template <int DEG, bool GRAD>
__global__ void __launch_bounds__(256)
lz_k_sh_forward(const float* __restrict__ inputs, float* __restrict__ outputs, uint32_t B, float* __restrict__ dy_dx) {
    const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    constexpr int C2 = DEG * DEG;
    const float x = inputs[(size_t)b * 3], y = inputs[(size_t)b * 3 + 1], z = inputs[(size_t)b * 3 + 2];
    float o[C2];
    if constexpr (GRAD) {
        float dx[C2], dy[C2], dz[C2];
        lz_sh_eval(x, y, z, DEG, o, dx, dy, dz);
        float* d = dy_dx + (size_t)b * 3 * C2;  // [B, 3, C2], shencoder.cu:127-129
#pragma unroll
        for (int i = 0; i < C2; i++) { d[i] = dx[i]; d[C2 + i] = dy[i]; d[2 * C2 + i] = dz[i]; }
    } else {
        lz_sh_eval(x, y, z, DEG, o, nullptr, nullptr, nullptr);
    }
    float* out = outputs + (size_t)b * C2;
    if constexpr (C2 % 4 == 0) {
#pragma unroll
        for (int i = 0; i < C2; i += 4) *reinterpret_cast<float4*>(out + i) = make_float4(o[i], o[i + 1], o[i + 2], o[i + 3]);
    } else {
#pragma unroll
        for (int i = 0; i < C2; i++) out[i] = o[i];
    }
}

__global__ void __launch_bounds__(256)
lz_k_sh_backward(const float* __restrict__ grad, uint32_t B, uint32_t C2, const float* __restrict__ dy_dx, float* __restrict__ grad_inputs) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= B * 3) return;
    const uint32_t b = t / 3, d = t - b * 3;
    const float* g = grad + (size_t)b * C2;
    const float* j = dy_dx + (size_t)b * 3 * C2 + (size_t)d * C2;
    float r = grad_inputs[t];
    for (uint32_t ch = 0; ch < C2; ch++) r = lz_fmaf(g[ch], j[ch], r);
    grad_inputs[t] = r;
}

template <bool GRAD>
static int lz_sh_dispatch(const float* in, float* out, uint32_t B, uint32_t degree, float* dy_dx, hipStream_t st) {
    dim3 grid(lz_div_up(B, 256)), block(256);
    switch (degree) {
        case 1: hipLaunchKernelGGL((lz_k_sh_forward<1, GRAD>), grid, block, 0, st, in, out, B, dy_dx); break;
        case 2: hipLaunchKernelGGL((lz_k_sh_forward<2, GRAD>), grid, block, 0, st, in, out, B, dy_dx); break;
        case 3: hipLaunchKernelGGL((lz_k_sh_forward<3, GRAD>), grid, block, 0, st, in, out, B, dy_dx); break;
        case 4: hipLaunchKernelGGL((lz_k_sh_forward<4, GRAD>), grid, block, 0, st, in, out, B, dy_dx); break;
        case 5: hipLaunchKernelGGL((lz_k_sh_forward<5, GRAD>), grid, block, 0, st, in, out, B, dy_dx); break;
        case 6: hipLaunchKernelGGL((lz_k_sh_forward<6, GRAD>), grid, block, 0, st, in, out, B, dy_dx); break;
        case 7: hipLaunchKernelGGL((lz_k_sh_forward<7, GRAD>), grid, block, 0, st, in, out, B, dy_dx); break;
        case 8: hipLaunchKernelGGL((lz_k_sh_forward<8, GRAD>), grid, block, 0, st, in, out, B, dy_dx); break;
        default: lz_set_error("SH encoder only supports degree in [1, 8]"); return LZ_ERR_UNSUPPORTED;
    }
    return LZ_OK;
}

extern "C" int lz_sh_encode_forward(const float* inputs, float* outputs, uint32_t B, uint32_t D, uint32_t degree, float* dy_dx,
                                    lz_stream_t stream) {
    LZ_REQUIRE(D == 3, LZ_ERR_UNSUPPORTED, "SH encoder only support input dim == 3");
    if (B == 0) return LZ_OK;
    LZ_REQUIRE(inputs && outputs, LZ_ERR_BAD_ARGUMENT, "sh_encode_forward: null tensor");
    if (B == 0) return LZ_OK;
    int rc = dy_dx ? lz_sh_dispatch<true>(inputs, outputs, B, degree, dy_dx, lz_st(stream))
                   : lz_sh_dispatch<false>(inputs, outputs, B, degree, nullptr, lz_st(stream));
    if (rc != LZ_OK) return rc;
    LZ_CHECK_LAUNCH("sh_encode_forward");
    return LZ_OK;
}

extern "C" int lz_sh_encode_backward(const float* grad, const float* inputs, uint32_t B, uint32_t D, uint32_t degree,
                                     const float* dy_dx, float* grad_inputs, lz_stream_t stream) {
    (void)inputs;
    LZ_REQUIRE(D == 3 && degree >= 1 && degree <= 8, LZ_ERR_UNSUPPORTED, "sh_encode_backward: D must be 3, degree in [1, 8]");
    if (B == 0) return LZ_OK;
    LZ_REQUIRE(grad && dy_dx && grad_inputs, LZ_ERR_BAD_ARGUMENT, "sh_encode_backward: null tensor");
    if (B == 0) return LZ_OK;
    hipLaunchKernelGGL(lz_k_sh_backward, dim3(lz_div_up((uint64_t)B * 3, 256)), dim3(256), 0, lz_st(stream), grad, B, degree * degree, dy_dx, grad_inputs);
    LZ_CHECK_LAUNCH("sh_encode_backward");
    return LZ_OK;
}

// ------------------------------------------------------------------------------------------------
// frequency encoding
// ------------------------------------------------------------------------------------------------
#define LZ_PI_F 3.141592653589793f

__global__ void __launch_bounds__(256)
lz_k_freq_forward(const float* __restrict__ inputs, uint32_t B, uint32_t D, uint32_t C, float* __restrict__ outputs) {
    const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (uint64_t)B * C) return;
    const uint32_t b = (uint32_t)(t / C), c = (uint32_t)(t - (uint64_t)b * C);
    const float* x = inputs + (size_t)b * D;
    float v;
    if (c < D) v = x[c];
    else {
        const uint32_t col = c / D - 1, d = c % D, freq = col / 2;
        const float phase = (float)(col % 2) * (LZ_PI_F / 2);
        v = lz_sinf(lz_scalbnf(x[d], (int)freq) + phase);
    }
    outputs[t] = v;
}

__global__ void __launch_bounds__(256)
lz_k_freq_backward(const float* __restrict__ grad, const float* __restrict__ outputs, uint32_t B, uint32_t D, uint32_t deg,
                   uint32_t C, float* __restrict__ grad_inputs) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= B * D) return;
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

extern "C" int lz_freq_encode_forward(const float* inputs, uint32_t B, uint32_t D, uint32_t deg, uint32_t C, float* outputs,
                                      lz_stream_t stream) {
    if (B == 0) return LZ_OK;
    LZ_REQUIRE(inputs && outputs, LZ_ERR_BAD_ARGUMENT, "freq_encode_forward: null tensor");
    LZ_REQUIRE(C == D + 2 * D * deg, LZ_ERR_BAD_ARGUMENT, "freq_encode_forward: C must equal D + 2*D*deg");
    if (B == 0) return LZ_OK;
    hipLaunchKernelGGL(lz_k_freq_forward, dim3(lz_div_up((uint64_t)B * C, 256)), dim3(256), 0, lz_st(stream), inputs, B, D, C, outputs);
    LZ_CHECK_LAUNCH("freq_encode_forward");
    return LZ_OK;
}

extern "C" int lz_freq_encode_backward(const float* grad, const float* outputs, uint32_t B, uint32_t D, uint32_t deg, uint32_t C,
                                       float* grad_inputs, lz_stream_t stream) {
    if (B == 0) return LZ_OK;
    LZ_REQUIRE(grad && outputs && grad_inputs, LZ_ERR_BAD_ARGUMENT, "freq_encode_backward: null tensor");
    LZ_REQUIRE(C == D + 2 * D * deg, LZ_ERR_BAD_ARGUMENT, "freq_encode_backward: C must equal D + 2*D*deg");
    if (B == 0) return LZ_OK;
    hipLaunchKernelGGL(lz_k_freq_backward, dim3(lz_div_up((uint64_t)B * D, 256)), dim3(256), 0, lz_st(stream), grad, outputs, B, D, deg, C, grad_inputs);
    LZ_CHECK_LAUNCH("freq_encode_backward");
    return LZ_OK;
}
