// lz_head_gather.h -- the triplane feature gather shared by the f32 and f16 fused heads (lz_head.hip, lz_head_f16.hip).
//
// Lane (s, q) of a wave owns sample s and the 9 enc_x features f = 4 i + q (i < 9): plane = i / 3 (xy, yz, xz;
// nerf_triplane/network.py:208-223), level = 4 (i % 3) + q of the D = 2, L = 12, C = 1 hash grid (gridencoder.cu:75-177).
// Arithmetic is the grid encoder's, bit for bit: pos = fma(x, scale, 0.5), corner weights in corner order, fma accumulation.
#ifndef LZ_HEAD_GATHER_H
#define LZ_HEAD_GATHER_H
#include "lz_common.h"
#include "lzzx_detmath.h"

// offs / lscale / lres: the per-level table in LDS ([0,13) offsets, scale, resolution); emb: the three planes' tables; (px, py, pz): the sample
// IN_RANGE drops the range clamps / out-of-range selects for a caller that guarantees |x|, |y|, |z| <= bound -- the fused f16 frame kernel,
// whose march clamps every sample (raymarching.cu:866-889) and which is bound by VALU issue; the host checks that the march's bound does
// not exceed the head's.  An identity on the values -- but mind what it exposes: with the select gone, `(_Float16)encx[i]` sat directly
// behind the last fma of the interpolation and the compiler folded the two into v_fma_mixlo_f16 (one rounding instead of f32-then-half):
// 28 pixels of a 96 x 96 frame moved by 2e-7 against the loop until the conversions went through h_round (lz_head_f16_slice.h).
template <bool IN_RANGE = false>
__device__ __forceinline__ void lz_head_gather(const float* const (&emb)[3], const int* __restrict__ offs, const float* __restrict__ lscale,
                                               const int* __restrict__ lres, float px, float py, float pz, int q, float bound,
                                               float two_bound, float (&encx)[9]) {
    // the three grid levels this lane touches (level = 4 m + q), gridencoder.cu:124-126; rebuilt per slice from LDS so that they
    // do not occupy registers during the matrix phase
    uint32_t lv_off[3], lv_hs[3], lv_stride[3];
    float lv_scale[3];
    bool lv_dense[3];
#pragma unroll
    for (int mrec = 0; mrec < 3; mrec++) {
        const int level = 4 * mrec + q;
        lv_off[mrec] = (uint32_t)offs[level];
        lv_hs[mrec] = (uint32_t)offs[level + 1] - lv_off[mrec];
        lv_scale[mrec] = lscale[level];
        lv_stride[mrec] = (uint32_t)lres[level] + 1u;
        lv_dense[mrec] = lv_stride[mrec] <= lv_hs[mrec] && lv_stride[mrec] * lv_stride[mrec] <= lv_hs[mrec];
    }
    // (x + bound) / (2 bound), grid.py:143.  When 2 bound is a power of two (bound 1, 2, 4 ...: every scene of the reference) the
    // division equals the multiplication by its exact reciprocal bit for bit, and an IEEE division is ~11 VALU instructions
    const uint32_t tb_bits = __float_as_uint(two_bound);
    const bool pow2 = (tb_bits & 0x007fffffu) == 0u && tb_bits > 0x00800000u && tb_bits < 0x7f000000u;   // wave-uniform
    float x01, y01, z01;
    if (pow2) {
        const float inv = __uint_as_float(0x7f000000u - tb_bits);    // 2^-k for two_bound = 2^k
        x01 = (px + bound) * inv; y01 = (py + bound) * inv; z01 = (pz + bound) * inv;
    } else {
        x01 = (px + bound) / two_bound; y01 = (py + bound) / two_bound; z01 = (pz + bound) / two_bound;
    }
    // Branch-free: out-of-range coordinates are clamped for ADDRESSING only and the feature is zeroed by a select
    // (gridencoder.cu:98-122), so all 36 gathers of a sample are independent loads.  Two passes so that the 36 table reads are IN
    // FLIGHT TOGETHER (one L2 round trip per slice instead of one per read): pass 1 computes fractions + table indices and issues
    // every load, the empty asm pins all 36 results as live at one point (so the compiler cannot sink a load next to its use),
    // pass 2 forms the bilinear weights and accumulates in corner order.
    float fr0[9], fr1[9], gv[9][4];
    bool oobf[9];
#pragma unroll
    for (int i = 0; i < 9; i++) {
        constexpr int kPlaneOf[9] = {0, 0, 0, 1, 1, 1, 2, 2, 2};
        const int plane = kPlaneOf[i], mrec = i % 3;
        const float u = plane == 1 ? y01 : x01;          // xy: (x,y)  yz: (y,z)  xz: (x,z)   network.py:211
        const float v = plane == 0 ? y01 : z01;
        oobf[i] = IN_RANGE ? false : (u < 0 || u > 1 || v < 0 || v > 1);
        const float uc = IN_RANGE ? u : lz_fminf(lz_fmaxf(u, 0.0f), 1.0f), vc = IN_RANGE ? v : lz_fminf(lz_fmaxf(v, 0.0f), 1.0f);
        // byte offset in 32 bits off the plane's (wave-uniform) base: one VALU op and the scalar-base addressing mode per gather, where
        // a per-lane 64-bit pointer costs two or three (the tables are 650 KB each)
        const char* gb = reinterpret_cast<const char*>(emb[plane]);
        const float p0 = lz_fmaf(uc, lv_scale[mrec], 0.5f), p1 = lz_fmaf(vc, lv_scale[mrec], 0.5f);
        const uint32_t g0 = (uint32_t)floorf(p0), g1 = (uint32_t)floorf(p1);
        fr0[i] = p0 - (float)g0;
        fr1[i] = p1 - (float)g1;
        // gridencoder.cu:54-72 for D = 2: dense while (res+1)^2 fits the level's table (then index < size and the modulo is the
        // identity), else fast_hash (primes 1, 2654435761) modulo the table size.  A hashed level's size is min(2^T, .) = 2^T
        // (grid.py:116), a power of two: the modulo is a mask (precondition, checked by the Python wrapper).  The row term of the upper
        // corners is the lower one plus a constant (mod 2^32), and the dense product fits 24 bits (full-rate multiplier): one
        // quarter-rate multiply per feature instead of four.
        const uint32_t hrow0 = g1 * 2654435761u, hrow1 = hrow0 + 2654435761u;
        const uint32_t drow0 = __umul24(g1, lv_stride[mrec]), drow1 = drow0 + lv_stride[mrec];
        const uint32_t hmask = lv_hs[mrec] - 1u;
#pragma unroll
        for (int c = 0; c < 4; c++) {
            const uint32_t c0 = g0 + (c & 1);
            const uint32_t index = lv_dense[mrec] ? c0 + ((c >> 1) ? drow1 : drow0) : ((c0 ^ ((c >> 1) ? hrow1 : hrow0)) & hmask);
            gv[i][c] = *reinterpret_cast<const float*>(gb + ((lv_off[mrec] + index) << 2));
        }
    }
    asm volatile("" ::"v"(gv[0][0]), "v"(gv[0][1]), "v"(gv[0][2]), "v"(gv[0][3]), "v"(gv[1][0]), "v"(gv[1][1]), "v"(gv[1][2]),
                 "v"(gv[1][3]), "v"(gv[2][0]), "v"(gv[2][1]), "v"(gv[2][2]), "v"(gv[2][3]), "v"(gv[3][0]), "v"(gv[3][1]),
                 "v"(gv[3][2]), "v"(gv[3][3]), "v"(gv[4][0]), "v"(gv[4][1]), "v"(gv[4][2]), "v"(gv[4][3]), "v"(gv[8][0]),
                 "v"(gv[8][1]), "v"(gv[8][2]), "v"(gv[8][3]), "v"(gv[7][0]), "v"(gv[7][1]), "v"(gv[7][2]), "v"(gv[7][3]));
    asm volatile("" ::"v"(gv[5][0]), "v"(gv[5][1]), "v"(gv[5][2]), "v"(gv[5][3]), "v"(gv[6][0]), "v"(gv[6][1]), "v"(gv[6][2]),
                 "v"(gv[6][3]));
#pragma unroll
    for (int i = 0; i < 9; i++) {
        float acc = 0.0f;
#pragma unroll
        for (int c = 0; c < 4; c++) {
            const float w = ((c & 1) ? fr0[i] : 1 - fr0[i]) * ((c >> 1) ? fr1[i] : 1 - fr1[i]);
            acc = lz_fmaf(w, gv[i][c], acc);
        }
        encx[i] = oobf[i] ? 0.0f : acc;
    }
}
#endif
