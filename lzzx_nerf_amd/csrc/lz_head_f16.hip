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
#include <hip/hip_fp16.h>

#include "lz_common.h"
#include "lzzx_detmath.h"
#include "lzzx_sh_eval.h"
#include "lz_head_gather.h"

typedef float lz_f4 __attribute__((ext_vector_type(4)));
typedef _Float16 lz_h8 __attribute__((ext_vector_type(8)));

enum { H_A1 = 0, H_A2, H_E1, H_E2, H_S1, H_S2, H_S3, H_C1, H_C2, H_COUNT };
//                               A1 A2 E1 E2 S1 S2 S3 C1 C2
constexpr int H_KS[H_COUNT] = {  2, 2, 2, 1, 3, 2, 2, 3, 2 };
constexpr int H_NT[H_COUNT] = {  4, 2, 1, 1, 4, 4, 5, 4, 1 };
constexpr int h_frag_base(int layer) {
    int b = 0;
    for (int i = 0; i < layer; i++) b += H_KS[i] * H_NT[i];
    return b;
}
constexpr int H_FRAGS = h_frag_base(H_COUNT);  // 59
static_assert(H_FRAGS * 64 * 16 == LZ_HEAD_PACKED_F16_BYTES, "packed size mismatch with the header");

extern "C" uint32_t lz_head_packed_size_f16(void) { return (uint32_t)H_FRAGS * 64u * 16u; }

// ---- weight packing: fragment (layer, k-step, feature tile), lane (m = l & 15, kg = l >> 4) holds W[16 ft + m][k(ks, kg, j)], j < 8
__device__ __forceinline__ int h_chain(int ks, int kg, int j, int K) {   // two D tiles -> one B operand
    const int f = 16 * (2 * ks + (j >> 2)) + 4 * kg + (j & 3);
    return f < K ? f : -1;
}
__device__ __forceinline__ int h_encx(int ks, int kg, int j) {           // lane kg gathers features 4 i + kg, i = 8 ks + j < 9
    const int i = 8 * ks + j;
    return i < 9 ? 4 * i + kg : -1;
}

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
    int srow;
    if (layer == H_S3) srow = row < 64 ? row + 1 : (row == 64 ? 0 : -1);  // geo rows first, sigma row in tile 4
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

// ---- the kernel ---------------------------------------------------------------------------------------
#define H_WG 1024

struct LzHead16Args {
    const float* emb[3];
    const int* offsets;
    const lz_h8* packed;
    const float* enc_a;
    const float* ind_code;
    const float* eye;
    float bound;
    float scale[12];
    uint32_t res[12];
};

template <int LAYER>
__device__ __forceinline__ void h_layer(const lz_h8* __restrict__ wl, int lane, const lz_h8 (&b)[H_KS[LAYER]], lz_f4 (&acc)[H_NT[LAYER]]) {
    constexpr int KS = H_KS[LAYER], NT = H_NT[LAYER];
    const lz_h8* frag = wl + h_frag_base(LAYER) * 64 + lane;
#pragma unroll
    for (int ks = 0; ks < KS; ks++)
#pragma unroll
        for (int ft = 0; ft < NT; ft++) acc[ft] = __builtin_amdgcn_mfma_f32_16x16x32_f16(frag[(ks * NT + ft) * 64], b[ks], acc[ft], 0, 0, 0);
}

__device__ __forceinline__ _Float16 h_relu16(float v) { return (_Float16)(v > 0.0f ? v : 0.0f); }   // relu(half(v)) == half(relu(v))

// two D tiles of a layer -> one B operand of the next (ReLU + round to half = the half output of an autocast Linear + relu)
__device__ __forceinline__ lz_h8 h_pair(const lz_f4& lo, const lz_f4& hi, bool relu) {
    lz_h8 b;
#pragma unroll
    for (int r = 0; r < 4; r++) {
        b[r] = relu ? h_relu16(lo[r]) : (_Float16)lo[r];
        b[4 + r] = relu ? h_relu16(hi[r]) : (_Float16)hi[r];
    }
    return b;
}

__global__ void __launch_bounds__(H_WG, H_WG / 256)
lz_k_triplane_head_f16(LzHead16Args P, const float* __restrict__ xyzs, const float* __restrict__ dirs, uint32_t M,
                       const int* __restrict__ count, float* __restrict__ sigmas, float* __restrict__ rgbs,
                       float* __restrict__ amb_aud, float* __restrict__ amb_eye, float* __restrict__ unc_out) {
    __shared__ lz_h8 wl[H_FRAGS * 64 + 24];   // packed A fragments, then 96 words: level table (64), enc_a (32)
    uint32_t Meff = M;
    if (count) {
        const int c = *count;
        Meff = c < 0 ? 0u : ((uint32_t)c < M ? (uint32_t)c : M);
    }
    const uint32_t n_slices = (Meff + 15) / 16;
    const uint32_t slice_lo = (uint32_t)(((uint64_t)n_slices * blockIdx.x) / gridDim.x);
    const uint32_t slice_hi = (uint32_t)(((uint64_t)n_slices * (blockIdx.x + 1)) / gridDim.x);
    if (slice_lo >= slice_hi) return;

    float* tabf = reinterpret_cast<float*>(wl + H_FRAGS * 64);
    int* tab = reinterpret_cast<int*>(tabf);
    {
        for (int i = threadIdx.x; i < H_FRAGS * 64; i += H_WG) wl[i] = P.packed[i];
        if (threadIdx.x < 13) tab[threadIdx.x] = P.offsets[threadIdx.x];
        if (threadIdx.x < 12) {
            tabf[16 + threadIdx.x] = P.scale[threadIdx.x];
            tab[32 + threadIdx.x] = (int)P.res[threadIdx.x];
        }
        if (threadIdx.x < 32) tabf[64 + threadIdx.x] = (float)(_Float16)P.enc_a[threadIdx.x];   // enc_a is half under autocast
        if (threadIdx.x == 0) tab[48] = 0;   // slice queue head
    }
    __syncthreads();
    const int* offs = tab;
    const float* lscale = tabf + 16;
    const int* lres = tab + 32;
    const float* lenca = tabf + 64;

    const int lane = threadIdx.x & 63;
    const int s = lane & 15, q = lane >> 4;
    const float two_bound = 2.0f * P.bound;
    const bool has_eye = P.eye != nullptr;
    const float eye_v = has_eye ? P.eye[0] : 0.0f;
    const float unc_const = lz_softplusf(0.0f);   // test mode (network.py:243-249, 278)
    int* queue = tab + 48;
    for (;;) {
        int slice = 0;
        if (lane == 0) slice = atomicAdd(queue, 1);
        slice = __builtin_amdgcn_readfirstlane(slice);
        if (slice_lo + (uint32_t)slice >= slice_hi) break;
        const uint32_t base = (slice_lo + (uint32_t)slice) * 16;
        uint32_t m = base + s;
        if (m >= Meff) m = Meff - 1;  // clamp: computed, never stored

        // ---------------- gather (f32, the same code as lz_k_triplane_head: lz_head_gather.h): lane q holds enc_x features 4 i + q
        float encx[9];
        lz_head_gather(P.emb, offs, lscale, lres, xyzs, m, q, P.bound, two_bound, encx);
        // enc_x as two half B operands (slot j of k-step ks <-> i = 8 ks + j); slot (1, q = 0, 1) is filled in for the sigma net
        lz_h8 bx[2];
#pragma unroll
        for (int j = 0; j < 8; j++) { bx[0][j] = (_Float16)encx[j]; bx[1][j] = (_Float16)0.0f; }
        bx[1][0] = (_Float16)encx[8];

        // ---------------- audio channel attention: 36 -> 64 -> 32 ----------------
        _Float16 att16[8];   // [4 t + r] = feature 16 t + 4 q + r
        {
            lz_f4 a1[4] = {lz_f4{0, 0, 0, 0}, lz_f4{0, 0, 0, 0}, lz_f4{0, 0, 0, 0}, lz_f4{0, 0, 0, 0}};
            h_layer<H_A1>(wl, lane, bx, a1);
            const lz_h8 b2[2] = {h_pair(a1[0], a1[1], true), h_pair(a1[2], a1[3], true)};
            lz_f4 a2[2] = {lz_f4{0, 0, 0, 0}, lz_f4{0, 0, 0, 0}};
            h_layer<H_A2>(wl, lane, b2, a2);
#pragma unroll
            for (int r = 0; r < 4; r++) { att16[r] = (_Float16)a2[0][r]; att16[4 + r] = (_Float16)a2[1][r]; }
        }
        // ambient_aud = || att ||_2 in f32 (norm is an autocast-to-f32 op): lane partial over its 8 features, then over q
        float ss = 0.0f;
#pragma unroll
        for (int k = 0; k < 8; k++) ss = lz_fmaf((float)att16[k], (float)att16[k], ss);
        ss += __shfl_xor(ss, 16, 64);
        ss += __shfl_xor(ss, 32, 64);
        const float ambaud = sqrtf(ss);
        // ---------------- eye attention: 36 -> 16 -> 1, sigmoid (half) ----------------
        float eyeatt = 0.0f;
        if (has_eye) {
            lz_f4 e1[1] = {lz_f4{0, 0, 0, 0}};
            h_layer<H_E1>(wl, lane, bx, e1);
            const lz_f4 z = lz_f4{0, 0, 0, 0};
            const lz_h8 be[1] = {h_pair(e1[0], z, true)};
            lz_f4 e2[1] = {lz_f4{0, 0, 0, 0}};
            h_layer<H_E2>(wl, lane, be, e2);
            eyeatt = (float)(_Float16)lz_sigmoidf((float)(_Float16)e2[0][0]);   // valid on lanes q == 0
        }
        // ---------------- sigma net: [enc_x 36 | enc_a * att 32 | eye * eye_att 1] -> 64 -> 64 -> 65 ----------------
        lz_h8 geo16[2];
        float sigma;
        {
            lz_h8 b1[3];
            b1[0] = bx[0];
            b1[1] = bx[1];
            b1[1][1] = (has_eye && q == 0) ? (_Float16)(eye_v * eyeatt) : (_Float16)0.0f;
#pragma unroll
            for (int j = 0; j < 8; j++) b1[2][j] = (_Float16)(lenca[16 * (j >> 2) + 4 * q + (j & 3)] * (float)att16[j]);
            lz_f4 s1[4] = {lz_f4{0, 0, 0, 0}, lz_f4{0, 0, 0, 0}, lz_f4{0, 0, 0, 0}, lz_f4{0, 0, 0, 0}};
            h_layer<H_S1>(wl, lane, b1, s1);
            const lz_h8 b2[2] = {h_pair(s1[0], s1[1], true), h_pair(s1[2], s1[3], true)};
            lz_f4 s2[4] = {lz_f4{0, 0, 0, 0}, lz_f4{0, 0, 0, 0}, lz_f4{0, 0, 0, 0}, lz_f4{0, 0, 0, 0}};
            h_layer<H_S2>(wl, lane, b2, s2);
            const lz_h8 b3[2] = {h_pair(s2[0], s2[1], true), h_pair(s2[2], s2[3], true)};
            lz_f4 s3[5] = {lz_f4{0, 0, 0, 0}, lz_f4{0, 0, 0, 0}, lz_f4{0, 0, 0, 0}, lz_f4{0, 0, 0, 0}, lz_f4{0, 0, 0, 0}};
            h_layer<H_S3>(wl, lane, b3, s3);
            geo16[0] = h_pair(s3[0], s3[1], false);   // geo_feat, no activation (network.py:304)
            geo16[1] = h_pair(s3[2], s3[3], false);
            sigma = lz_expf((float)(_Float16)s3[4][0]);   // trunc_exp casts its half input to f32; lanes q == 0
        }
        // ---------------- colour net: [SH 16 | geo 64 | ind 4] -> 64 -> 3 ----------------
        float rgb[3];
        {
            float o[16];
            lz_sh_eval(dirs[(size_t)m * 3], dirs[(size_t)m * 3 + 1], dirs[(size_t)m * 3 + 2], 4, o, nullptr, nullptr, nullptr);
            lz_h8 b1[3];
#pragma unroll
            for (int j = 0; j < 4; j++) {
                b1[0][j] = (_Float16)(q == 0 ? o[j] : (q == 1 ? o[4 + j] : (q == 2 ? o[8 + j] : o[12 + j])));   // SH 4 q + j
                b1[0][4 + j] = (q == 0 && P.ind_code) ? (_Float16)P.ind_code[j] : (_Float16)0.0f;
            }
            b1[1] = geo16[0];
            b1[2] = geo16[1];
            lz_f4 c1[4] = {lz_f4{0, 0, 0, 0}, lz_f4{0, 0, 0, 0}, lz_f4{0, 0, 0, 0}, lz_f4{0, 0, 0, 0}};
            h_layer<H_C1>(wl, lane, b1, c1);
            const lz_h8 b2[2] = {h_pair(c1[0], c1[1], true), h_pair(c1[2], c1[3], true)};
            lz_f4 c2[1] = {lz_f4{0, 0, 0, 0}};
            h_layer<H_C2>(wl, lane, b2, c2);
#pragma unroll
            for (int c = 0; c < 3; c++) {   // network.py:275 in half: sigmoid, * 1.002, - 0.001, each rounded to half
                const _Float16 sg = (_Float16)lz_sigmoidf((float)(_Float16)c2[0][c]);
                const _Float16 t1 = (_Float16)((float)sg * 1.002f);
                rgb[c] = (float)(_Float16)((float)t1 - 0.001f);
            }
        }
        // ---------------- store (lanes q == 0 own sample s) ----------------
        if (q == 0 && base + s < Meff) {
            sigmas[m] = sigma;
            rgbs[(size_t)m * 3] = rgb[0]; rgbs[(size_t)m * 3 + 1] = rgb[1]; rgbs[(size_t)m * 3 + 2] = rgb[2];
            amb_aud[m] = ambaud;
            if (amb_eye) amb_eye[m] = eyeatt;
            unc_out[m] = unc_const;
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
    static int n_cu = 0;
    if (n_cu == 0) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) n_cu = prop.multiProcessorCount;
        if (n_cu <= 0) n_cu = 256;
    }
    const uint32_t tiles = lz_div_up(M, 256);
    const uint32_t grid = tiles < (uint32_t)n_cu ? tiles : (uint32_t)n_cu;
    hipLaunchKernelGGL(lz_k_triplane_head_f16, dim3(grid), dim3(H_WG), 0, st, a, xyzs, dirs, M, count, sigmas, rgbs, amb_aud, amb_eye, unc);
    return LZ_OK;
}
