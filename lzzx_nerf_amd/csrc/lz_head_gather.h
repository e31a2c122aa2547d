// lz_head_gather.h -- the triplane feature gather shared by the f32 and f16 fused heads (lz_head.hip, lz_head_f16.hip).
//
// Lane (s, q) of a wave owns sample s and the 9 enc_x features f = 4 i + q (i < 9): plane = i / 3 (xy, yz, xz;
// nerf_triplane/network.py:208-223), level = 4 (i % 3) + q of the D = 2, L = 12, C = 1 hash grid (gridencoder.cu:75-177).
// Arithmetic is the grid encoder's, bit for bit: pos = fma(x, scale, 0.5), corner weights in corner order, fma accumulation.
//
// Instruction budget (round 3; the f16 kernels are bound by vector-instruction issue, 4 cycles per wave64 instruction): the census of
// round 2's version was 59 (per-level setup) + 213 (positions, indices) + ~60 (interpolation) VALU instructions per 16-sample slice.
//   * the per-level constants are records in LDS written once per workgroup (lz_level_table_fill), not rebuilt per slice;
//   * ONE index formula for dense and hashed levels, no select: element = ((col ^ rowH) & mask) + rowD with
//         hashed: rowH = row * P (mod 2^24), mask = size - 1, rowD = off          (gridencoder.cu:35-51, primes 1 and 2654435761)
//         dense:  rowH = 0,                  mask = ~0,       rowD = off + row * (res + 1)
//     -- two instructions per corner (v_bitop3_b32, v_add_lshl_u32).  The hash product only matters modulo the table size (a power of
//     two <= 2^24), so it is a full-rate 24-bit multiply by P mod 2^24 instead of a quarter-rate 32-bit one;
//   * what two planes share is computed once: positions / fractions / 1 - fraction per (coordinate, level) -- 9 sets, not 18 -- and the
//     row terms of the yz and xz planes (both rows are z);
//   * PACK: the interpolation can run two features per instruction on v_pk_mul_f32 / v_pk_fma_f32 (IEEE per half, same bits) -- see below.
#ifndef LZ_HEAD_GATHER_H
#define LZ_HEAD_GATHER_H
// PACK (the interpolation on v_pk_mul_f32 / v_pk_fma_f32, two features per instruction) is OFF in every head since round 4: fewer instructions
// but slower ones -- packed f32 vector ops cost more issue time than the two scalar ones they replace next to MFMA work
// (MI355X_MICROARCH.md, "packed f32 VALU ... an anti-lever beside MFMAs").  Same-box A/B on the fused frame: f16 1.916 -> 1.894 ms, f32
// 8.92 -> 8.88.  (-fno-slp-vectorize, which removes the compiler's own two dozen packed ops as well: no further gain.)
#ifndef LZ_GATHER_PACK16
#define LZ_GATHER_PACK16 false
#endif
#ifndef LZ_GATHER_PACK32
#define LZ_GATHER_PACK32 false
#endif
#include "lz_common.h"
#include "lzzx_detmath.h"

// The small per-workgroup table in LDS, LZ_LVTAB_WORDS 32-bit words:
//   [0,13) level offsets | [16,28) scale (f32) | [32,44) dense row stride res + 1 (0 on hashed levels) | [48] slice queue head of the
//   stand-alone kernels | [49,61) hash multiplier P mod 2^24 (0 on dense levels) | [64,96) enc_a (f32) | [96,108) index mask (size - 1 on
//   hashed levels, ~0 on dense ones) | f16 heads: [108,124) enc_a as 32 packed halves | [124,126) ind_code as 4 packed halves
#define LZ_LVTAB_WORDS 128
#define LZ_LVTAB_SCALE 16
#define LZ_LVTAB_STRIDE 32
#define LZ_LVTAB_QUEUE 48
#define LZ_LVTAB_HMUL 49
#define LZ_LVTAB_ENCA 64
#define LZ_LVTAB_MASK 96
#define LZ_LVTAB_ENCA16 108
#define LZ_LVTAB_IND16 124
#define LZ_LVTAB_DENSE0 62   // 1 when levels 0..3 (the first level record of every lane) are all dense: their x / x + 1 corners are adjacent entries

// threads 0..12 of a workgroup write the level part of the table (caller synchronises afterwards).  gridencoder.cu:54-72 for D = 2:
// a level is dense while (res + 1)^2 fits its table, else hashed with size = 2^T (grid.py:116; the Python wrapper checks the power of two).
__device__ __forceinline__ void lz_level_table_fill(int* tab, const int* __restrict__ offsets, const float* scale, const uint32_t* res) {
    const uint32_t t = threadIdx.x;
    if (t < 13) tab[t] = offsets[t];
    if (t < 12) {
        const uint32_t hs = (uint32_t)(offsets[t + 1] - offsets[t]), stride = res[t] + 1u;
        const bool dense = stride <= hs && (uint64_t)stride * stride <= hs;
        reinterpret_cast<float*>(tab)[LZ_LVTAB_SCALE + t] = scale[t];
        tab[LZ_LVTAB_STRIDE + t] = dense ? (int)stride : 0;
        tab[LZ_LVTAB_HMUL + t] = dense ? 0 : (int)(2654435761u & 0x00ffffffu);
        tab[LZ_LVTAB_MASK + t] = dense ? -1 : (int)(hs - 1u);
    }
    if (t == 0) {
        tab[LZ_LVTAB_QUEUE] = 0;
        int d0 = 1;
        for (int l = 0; l < 4; l++) {
            const uint32_t hs = (uint32_t)(offsets[l + 1] - offsets[l]), stride = res[l] + 1u;
            if (!(stride <= hs && (uint64_t)stride * stride <= hs)) d0 = 0;
        }
        tab[LZ_LVTAB_DENSE0] = d0;
    }
}

typedef float lz_gf2 __attribute__((ext_vector_type(2)));

// tab: the table above; emb: the three planes' tables; (px, py, pz): the sample
// IN_RANGE drops the range clamps / out-of-range selects for a caller that guarantees |x|, |y|, |z| <= bound -- the fused f16 frame kernel,
// whose march clamps every sample (raymarching.cu:866-889) and which is bound by VALU issue; the host checks that the march's bound does
// not exceed the head's.  An identity on the values -- but mind what it exposes: with the select gone, `(_Float16)encx[i]` sat directly
// behind the last fma of the interpolation and the compiler folded the two into v_fma_mixlo_f16 (one rounding instead of f32-then-half):
// 28 pixels of a 96 x 96 frame moved by 2e-7 against the loop until the conversions went through h_round (lz_head_f16_slice.h).
// YIELD: the caller runs this wave at a raised priority (s_setprio) through its march / address work so that the 36 loads leave early; the
// priority drops once they are issued and the matrix phase behind them yields to the other waves' address work (fused f32 frame kernel:
// 9.17 -> 8.99 ms; dropping it before the address arithmetic instead: 9.06; no effect on the issue-bound f16 kernel)
// PAIR0: when levels 0..3 are dense (LZ_LVTAB_DENSE0; every lane's first level record), their x / x + 1 corners are adjacent table entries
// and come with one 8-byte load per row: 30 load instructions per lane instead of 36 and 18 index instructions less.  Same-box A/B on
// the fused frame: f32 9.05 -> 8.97 ms, f16 2.025 -> 2.04 ms (no gain: the f16 slices keep the one-load-per-corner form)
// LSTRIDE: the lane's three level records are levels q + LSTRIDE mrec (mrec < 3).  4: the 16-sample heads (lane group q of four owns features
// 4 i + q).  2: the 32-sample f16 head (lz_head_f16w_slice.h), whose lane half h owns 18 features in two calls, q = h and q = 6 + h.
// (x + bound) / (2 bound) of a sample, grid.py:143 -- what lz_head_gather starts with; a caller that gathers one sample in several calls
// (the 32-sample f16 slice) maps it once and passes MAPPED = true
__device__ __forceinline__ void lz_head_map01(float px, float py, float pz, float bound, float two_bound, float (&c01)[3]) {
    // When 2 bound is a power of two (bound 1, 2, 4 ...: every scene of the reference) the division equals the multiplication by its exact
    // reciprocal bit for bit, and an IEEE division is ~11 VALU instructions (readfirstlane: `two_bound` reaches this point in a vector
    // register, and a condition computed from it counts as divergent -- the compiler then evaluates BOTH sides below, three IEEE division
    // sequences of ~10 instructions each per slice, and selects; as a scalar it is a branch)
    const uint32_t tb_bits = (uint32_t)__builtin_amdgcn_readfirstlane((int)__float_as_uint(two_bound));
    const bool pow2 = (tb_bits & 0x007fffffu) == 0u && tb_bits > 0x00800000u && tb_bits < 0x7f000000u;   // wave-uniform
    if (pow2) {
        const float inv = __uint_as_float(0x7f000000u - tb_bits);    // 2^-k for two_bound = 2^k
        c01[0] = (px + bound) * inv; c01[1] = (py + bound) * inv; c01[2] = (pz + bound) * inv;
    } else {
        c01[0] = (px + bound) / two_bound; c01[1] = (py + bound) / two_bound; c01[2] = (pz + bound) / two_bound;
    }
}

template <bool IN_RANGE = false, bool PACK = false, bool YIELD = false, bool PAIR0 = false, int LSTRIDE = 4, bool MAPPED = false>
__device__ __forceinline__ void lz_head_gather(const float* const (&emb)[3], const int* __restrict__ tab, float px, float py, float pz, int q,
                                               float bound, float two_bound, float (&encx)[9]) {
    // the three grid levels this lane touches (level = 4 m + q); read per slice from LDS so that they do not occupy registers during the
    // matrix phase
    uint32_t lv_off[3], lv_strd[3], lv_hmul[3], lv_mask[3];
    float lv_scale[3];
#pragma unroll
    for (int mrec = 0; mrec < 3; mrec++) {
        const int level = LSTRIDE * mrec + q;
        lv_off[mrec] = (uint32_t)tab[level];
        lv_scale[mrec] = reinterpret_cast<const float*>(tab)[LZ_LVTAB_SCALE + level];
        lv_strd[mrec] = (uint32_t)tab[LZ_LVTAB_STRIDE + level];
        lv_hmul[mrec] = (uint32_t)tab[LZ_LVTAB_HMUL + level];
        lv_mask[mrec] = (uint32_t)tab[LZ_LVTAB_MASK + level];
    }
    float c01[3];
    if constexpr (MAPPED) { c01[0] = px; c01[1] = py; c01[2] = pz; }      // the caller ran lz_head_map01
    else lz_head_map01(px, py, pz, bound, two_bound, c01);
    // Branch-free: out-of-range coordinates are clamped for ADDRESSING only and the feature is zeroed by a select
    // (gridencoder.cu:98-122), so all 36 gathers of a sample are independent loads.  Two passes so that the 36 table reads are IN
    // FLIGHT TOGETHER (one L2 round trip per slice instead of one per read): pass 1 computes fractions + table indices and issues
    // every load, the empty asm pins all 36 results as live at one point (so the compiler cannot sink a load next to its use),
    // pass 2 forms the bilinear weights and accumulates in corner order.
    bool oobc[3];
    float cc[3];
#pragma unroll
    for (int d = 0; d < 3; d++) {
        oobc[d] = IN_RANGE ? false : (c01[d] < 0 || c01[d] > 1);
        cc[d] = IN_RANGE ? c01[d] : lz_fminf(lz_fmaxf(c01[d], 0.0f), 1.0f);
    }
    // per (coordinate, level): cell, fraction, 1 - fraction (gridencoder.cu:128-133).  p >= 0.5, so the reference's
    // p - (float)(uint32_t)floor(p) is p - floor(p), an exact subtraction -- which is what v_fract_f32 returns for p >= 0 -- and the
    // cell is the truncating conversion of p itself.
    uint32_t cell[3][3];
    float fr[3][3], om[3][3];
#pragma unroll
    for (int mrec = 0; mrec < 3; mrec++)
#pragma unroll
        for (int d = 0; d < 3; d++) {
            const float p = lz_fmaf(cc[d], lv_scale[mrec], 0.5f);
            cell[mrec][d] = (uint32_t)p;
            fr[mrec][d] = __builtin_amdgcn_fractf(p);
            om[mrec][d] = 1 - fr[mrec][d];
        }
    // row terms per (row coordinate, level): y for the xy plane, z for yz and xz (network.py:211: xy = (x, y), yz = (y, z), xz = (x, z))
    uint32_t rowH[2][3][2], rowD[2][3][2];     // [row coordinate: 0 = y, 1 = z][level record][lower / upper corner]
#pragma unroll
    for (int rc = 0; rc < 2; rc++)
#pragma unroll
        for (int mrec = 0; mrec < 3; mrec++) {
            const uint32_t g1 = cell[mrec][1 + rc];
            rowH[rc][mrec][0] = __umul24(g1, lv_hmul[mrec]);                       // exact modulo 2^24 >= the table size
            rowH[rc][mrec][1] = rowH[rc][mrec][0] + lv_hmul[mrec];
            rowD[rc][mrec][0] = __umul24(g1, lv_strd[mrec]) + lv_off[mrec];        // v_mad_u32_u24
            rowD[rc][mrec][1] = rowD[rc][mrec][0] + lv_strd[mrec];
        }
    float gv[9][4];
    static_assert(!PAIR0 || LSTRIDE == 4, "PAIR0 pairs the corners of levels 0..3");
    const bool dense0 = PAIR0 && __builtin_amdgcn_readfirstlane(tab[LZ_LVTAB_DENSE0]) != 0;     // workgroup-uniform, a scalar branch
#pragma unroll
    for (int i = 0; i < 9; i++) {
        constexpr int kPlaneOf[9] = {0, 0, 0, 1, 1, 1, 2, 2, 2};
        const int plane = kPlaneOf[i], mrec = i % 3;
        const int cd = plane == 1 ? 1 : 0, rc = plane == 0 ? 0 : 1;          // column coordinate (x, y, x), row coordinate (y, z, z)
        // byte offset in 32 bits off the plane's (wave-uniform) base: the scalar-base addressing mode, no per-lane 64-bit pointer
        const char* gb = reinterpret_cast<const char*>(emb[plane]);
        const uint32_t g0 = cell[mrec][cd];
        if (mrec == 0 && dense0) {
            // dense level: entry = column + row term, and the x + 1 corner is the NEXT entry -- one 8-byte load per row (4-byte aligned:
            // global_load_dwordx2 takes it) instead of two loads and two index computations: 30 load instructions per lane instead of 36
#pragma unroll
            for (int r = 0; r < 2; r++) {
                struct __attribute__((packed, aligned(4))) Pair { float a, b; } pr;
                __builtin_memcpy(&pr, __builtin_assume_aligned(gb + ((g0 + rowD[rc][0][r]) << 2), 4), 8);
                gv[i][2 * r] = pr.a;
                gv[i][2 * r + 1] = pr.b;
            }
            continue;
        }
#pragma unroll
        for (int c = 0; c < 4; c++) {
            const uint32_t c0 = g0 + (c & 1);
            const uint32_t index = ((c0 ^ rowH[rc][mrec][c >> 1]) & lv_mask[mrec]) + rowD[rc][mrec][c >> 1];
            gv[i][c] = *reinterpret_cast<const float*>(gb + (index << 2));
        }
    }
    if constexpr (YIELD) __builtin_amdgcn_s_setprio(0);
    asm volatile("" ::"v"(gv[0][0]), "v"(gv[0][1]), "v"(gv[0][2]), "v"(gv[0][3]), "v"(gv[1][0]), "v"(gv[1][1]), "v"(gv[1][2]),
                 "v"(gv[1][3]), "v"(gv[2][0]), "v"(gv[2][1]), "v"(gv[2][2]), "v"(gv[2][3]), "v"(gv[3][0]), "v"(gv[3][1]),
                 "v"(gv[3][2]), "v"(gv[3][3]), "v"(gv[4][0]), "v"(gv[4][1]), "v"(gv[4][2]), "v"(gv[4][3]), "v"(gv[8][0]),
                 "v"(gv[8][1]), "v"(gv[8][2]), "v"(gv[8][3]), "v"(gv[7][0]), "v"(gv[7][1]), "v"(gv[7][2]), "v"(gv[7][3]));
    asm volatile("" ::"v"(gv[5][0]), "v"(gv[5][1]), "v"(gv[5][2]), "v"(gv[5][3]), "v"(gv[6][0]), "v"(gv[6][1]), "v"(gv[6][2]),
                 "v"(gv[6][3]));
    // bilinear weights in the reference's corner order, w = (1 * w0) * w1 with w_d = 1 - f_d or f_d (gridencoder.cu:141-152), fma chain
    if constexpr (PACK) {
        // features (i, i + 1) of a plane share the coordinates and differ in the level: two of them per packed instruction; the ninth
        // (and each plane's third) pairs up across planes -- any pairing gives the same bits, the lanes of a packed op are independent
        constexpr int kPair[5][2] = {{0, 1}, {2, 3}, {4, 5}, {6, 7}, {8, 8}};
#pragma unroll
        for (int pr = 0; pr < 5; pr++) {
            const int ia = kPair[pr][0], ib = kPair[pr][1];
            constexpr int kPl[9] = {0, 0, 0, 1, 1, 1, 2, 2, 2};
            const int ca = kPl[ia] == 1 ? 1 : 0, ra = kPl[ia] == 0 ? 1 : 2, cb = kPl[ib] == 1 ? 1 : 0, rb = kPl[ib] == 0 ? 1 : 2;
            const int ma = ia % 3, mb = ib % 3;
            const lz_gf2 f0 = {fr[ma][ca], fr[mb][cb]}, f1 = {fr[ma][ra], fr[mb][rb]}, o0 = {om[ma][ca], om[mb][cb]}, o1 = {om[ma][ra], om[mb][rb]};
            lz_gf2 acc = {0.0f, 0.0f};
#pragma unroll
            for (int c = 0; c < 4; c++) {
                const lz_gf2 w = ((c & 1) ? f0 : o0) * ((c >> 1) ? f1 : o1);
                const lz_gf2 g = {gv[ia][c], gv[ib][c]};
                acc = __builtin_elementwise_fma(w, g, acc);
            }
            encx[ia] = acc[0];
            if (ib != ia) encx[ib] = acc[1];
        }
    } else {
#pragma unroll
        for (int i = 0; i < 9; i++) {
            constexpr int kPl[9] = {0, 0, 0, 1, 1, 1, 2, 2, 2};
            const int cd = kPl[i] == 1 ? 1 : 0, rd = kPl[i] == 0 ? 1 : 2, mrec = i % 3;
            float acc = 0.0f;
#pragma unroll
            for (int c = 0; c < 4; c++) {
                const float w = ((c & 1) ? fr[mrec][cd] : om[mrec][cd]) * ((c >> 1) ? fr[mrec][rd] : om[mrec][rd]);
                acc = lz_fmaf(w, gv[i][c], acc);
            }
            encx[i] = acc;
        }
    }
    if constexpr (!IN_RANGE) {
#pragma unroll
        for (int i = 0; i < 9; i++) {
            constexpr int kPl[9] = {0, 0, 0, 1, 1, 1, 2, 2, 2};
            const bool oob = kPl[i] == 0 ? (oobc[0] || oobc[1]) : (kPl[i] == 1 ? (oobc[1] || oobc[2]) : (oobc[0] || oobc[2]));
            encx[i] = oob ? 0.0f : encx[i];
        }
    }
}
#endif
