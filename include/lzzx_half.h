/*
 * lzzx_half.h -- IEEE binary16 <-> binary32 conversion (round-to-nearest-even), written out in
 * integer arithmetic so the CPU checker rounds exactly like v_cvt_f16_f32 on gfx950.
 */
#ifndef LZZX_HALF_H
#define LZZX_HALF_H
#include <stdint.h>
#include "lzzx_detmath.h"

LZ_HD float lz_half_to_float(uint16_t h) {
    const uint32_t sign = (uint32_t)(h & 0x8000u) << 16;
    uint32_t exp = (h >> 10) & 0x1fu;
    uint32_t man = h & 0x3ffu;
    if (exp == 0) {
        if (man == 0) return lz_u2f(sign);
        /* subnormal half: value = man * 2^-24 */
        const float v = (float)man * lz_pow2i(-24);
        return lz_u2f(lz_f2u(v) | sign);
    }
    if (exp == 31) return lz_u2f(sign | 0x7f800000u | (man << 13));
    return lz_u2f(sign | ((exp + 112u) << 23) | (man << 13));
}

LZ_HD uint16_t lz_float_to_half(float f) {
    const uint32_t u = lz_f2u(f);
    const uint16_t sign = (uint16_t)((u >> 16) & 0x8000u);
    const uint32_t a = u & 0x7fffffffu;
    if (a >= 0x7f800000u) return (uint16_t)(sign | (a > 0x7f800000u ? 0x7e00u : 0x7c00u));
    if (a >= 0x477ff000u) return (uint16_t)(sign | 0x7c00u);            /* rounds to >= 65520 -> inf */
    if (a < 0x33000001u) return sign;                                    /* <= 2^-25 -> 0 (ties to even) */
    if (a < 0x38800000u) {                                               /* subnormal half */
        const int e = (int)(a >> 23);                                    /* biased f32 exponent, 102..112 */
        const uint32_t m = (a & 0x7fffffu) | 0x800000u;                  /* 24-bit significand */
        const int shift = 126 - e;                                       /* 14..24 */
        uint32_t r = m >> shift;
        const uint32_t rem = m & ((1u << shift) - 1u);
        const uint32_t half = 1u << (shift - 1);
        if (rem > half || (rem == half && (r & 1u))) r++;
        return (uint16_t)(sign | r);
    }
    uint32_t r = ((a >> 23) - 112u) << 10 | ((a >> 13) & 0x3ffu);
    const uint32_t rem = a & 0x1fffu;
    if (rem > 0x1000u || (rem == 0x1000u && (r & 1u))) r++;
    return (uint16_t)(sign | r);
}

LZ_HD float lz_round_to_half(float f) { return lz_half_to_float(lz_float_to_half(f)); }

#endif
