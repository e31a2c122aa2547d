/*
 * lzzx_detmath.h -- deterministic single-precision math shared by the gfx950 kernels
 * (lzzx_nerf_amd/csrc) and the CPU checker (oracle/).
 *
 * Why this exists: the parity contract asks for bit-exact ray / grid indices and per-ray sample
 * counts. Per-ray sample counts depend on `T < T_thresh` in the compositing step, which depends on
 * exp() of MLP outputs. Vendor libm (glibc on the host, ocml on the device) differ in the last ulp,
 * so every transcendental the path needs is written here once out of IEEE-754 basic operations
 * (add, mul, fma, div, rint) that round identically on x86-64 SSE and on gfx950 VALU.  Both sides
 * are compiled with -ffp-contract=off; every fused multiply-add is an explicit lz_fmaf().
 *
 * The reference evaluates these with CUDA fast intrinsics (__expf, __sinf: raymarching.cu:649,
 * freqencoder.cu:56) whose bit patterns cannot be reproduced off NVIDIA hardware; accuracy here is
 * <= 2 ulp, i.e. tighter than those intrinsics.
 */
#ifndef LZZX_DETMATH_H
#define LZZX_DETMATH_H

#include <stdint.h>

#if defined(__HIPCC__)
#define LZ_HD __host__ __device__ static inline
#else
#define LZ_HD static inline
#endif

LZ_HD uint32_t lz_f2u(float f) { uint32_t u; __builtin_memcpy(&u, &f, 4); return u; }
LZ_HD float lz_u2f(uint32_t u) { float f; __builtin_memcpy(&f, &u, 4); return f; }

LZ_HD float lz_fmaf(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
LZ_HD float lz_rintf(float x) { return __builtin_rintf(x); }
LZ_HD float lz_fminf(float a, float b) { return __builtin_fminf(a, b); }
LZ_HD float lz_fmaxf(float a, float b) { return __builtin_fmaxf(a, b); }
LZ_HD float lz_fabsf(float a) { return __builtin_fabsf(a); }
LZ_HD float lz_clampf(float x, float lo, float hi) { return lz_fminf(hi, lz_fmaxf(lo, x)); }
LZ_HD float lz_signf(float x) { return __builtin_copysignf(1.0f, x); }

/* 2^k for k in [-126, 127] */
LZ_HD float lz_pow2i(int k) { return lz_u2f((uint32_t)(k + 127) << 23); }

/* exponent e such that |x| = m * 2^e with m in [0.5, 1): what frexpf() stores (0 for x == 0). */
LZ_HD int lz_frexp_exp(float x) {
    uint32_t u = lz_f2u(x) & 0x7fffffffu;
    if (u == 0) return 0;
    int e = (int)(u >> 23);
    if (e == 0) { /* subnormal: renormalise */
        u = lz_f2u(lz_u2f(u) * 8388608.0f);
        return (int)(u >> 23) - 126 - 23;
    }
    return e - 126;
}

/* x * 2^n, exact unless the result leaves the normal range (then one rounding / overflow) */
LZ_HD float lz_scalbnf(float x, int n) {
    if (n > 254) n = 254;
    if (n < -252) n = -252;
    while (n > 127) { x *= lz_pow2i(127); n -= 127; }
    while (n < -126) { x *= lz_pow2i(-126); n += 126; }
    return x * lz_pow2i(n);
}

LZ_HD float lz_expf(float x) {
    if (x != x) return x;
    if (x > 88.72283935546875f) return lz_u2f(0x7f800000u);
    if (x < -103.97208404541016f) return 0.0f;
    const float n = lz_rintf(x * 1.44269502162933349609375f);
    float r = lz_fmaf(n, -0.693145751953125f, x);          /* ln2 high part (12 significant bits) */
    r = lz_fmaf(n, -1.428606765330187045037746429443359375e-06f, r); /* ln2 low part */
    float p = 1.9841270113829523324966430664062e-04f;      /* 1/5040 */
    p = lz_fmaf(p, r, 1.3888889225199818611145019531250e-03f); /* 1/720 */
    p = lz_fmaf(p, r, 8.3333337679505348205566406250000e-03f); /* 1/120 */
    p = lz_fmaf(p, r, 4.1666667908430099487304687500000e-02f); /* 1/24 */
    p = lz_fmaf(p, r, 1.6666667163372039794921875000000e-01f); /* 1/6 */
    p = lz_fmaf(p, r, 0.5f);
    p = lz_fmaf(p, r, 1.0f);
    p = lz_fmaf(p, r, 1.0f);
    const int ni = (int)n;
    const int h = ni / 2;
    return (p * lz_pow2i(h)) * lz_pow2i(ni - h);
}

/* natural log, x > 0 (x == 0 -> -inf, x < 0 -> nan) */
LZ_HD float lz_logf(float x) {
    uint32_t u = lz_f2u(x);
    if (x != x) return x;
    if ((u & 0x7fffffffu) == 0) return lz_u2f(0xff800000u);
    if (u >> 31) return lz_u2f(0x7fc00000u);
    if (u == 0x7f800000u) return x;
    int e = 0;
    if ((u >> 23) == 0) { u = lz_f2u(x * 8388608.0f); e = -23; }
    e += (int)(u >> 23) - 127;
    float m = lz_u2f((u & 0x007fffffu) | 0x3f800000u);      /* [1, 2) */
    if (m > 1.41421353816986083984375f) { m *= 0.5f; e += 1; }
    const float f = m - 1.0f;
    const float s = f / (2.0f + f);
    const float z = s * s;
    float q = 9.0909093618392944335937500e-02f;             /* 1/11 */
    q = lz_fmaf(q, z, 1.1111111193895339965820312e-01f);    /* 1/9 */
    q = lz_fmaf(q, z, 1.4285714924335479736328125e-01f);    /* 1/7 */
    q = lz_fmaf(q, z, 2.0000000298023223876953125e-01f);    /* 1/5 */
    q = lz_fmaf(q, z, 3.3333334326744079589843750e-01f);    /* 1/3 */
    q = q * z;                                              /* atanh(s)/s - 1 */
    const float two_s = s + s;
    const float lm = lz_fmaf(two_s, q, two_s);              /* log(m) */
    const float fe = (float)e;
    const float lo = lz_fmaf(fe, 1.428606765330187045037746429443359375e-06f, lm);
    return lz_fmaf(fe, 0.693145751953125f, lo);
}

LZ_HD float lz_sigmoidf(float x) { return 1.0f / (1.0f + lz_expf(-x)); }

/* log(1 + exp(x)) exactly as network.py:278 spells it (no large-x shortcut) */
LZ_HD float lz_softplusf(float x) { return lz_logf(1.0f + lz_expf(x)); }

/* sin(x), |x| up to ~1e5 keeps <= 2 ulp; deterministic beyond */
LZ_HD float lz_sinf(float x) {
    if (x != x) return x;
    const float n = lz_rintf(x * 0.636619746685028076171875f);   /* 2/pi */
    float r = lz_fmaf(n, -1.57079637050628662109375f, x);        /* pi/2 split in three floats */
    r = lz_fmaf(n, 4.371138828673792886547744274139404296875e-08f, r);
    r = lz_fmaf(n, 1.7151245100058819e-15f, r);
    const int q = (int)n & 3;
    const float z = r * r;
    float res;
    if (q & 1) {
        float c = 2.4433157514e-05f;
        c = lz_fmaf(c, z, -1.3887316255e-03f);
        c = lz_fmaf(c, z, 4.1666645683e-02f);
        c = lz_fmaf(c, z, -0.5f);
        res = lz_fmaf(c, z, 1.0f);
    } else {
        float s = 2.7183114939e-06f;
        s = lz_fmaf(s, z, -1.9839334836e-04f);
        s = lz_fmaf(s, z, 8.3333337680e-03f);
        s = lz_fmaf(s, z, -1.6666667163e-01f);
        s = s * z;
        res = lz_fmaf(s, r, r);
    }
    return (q & 2) ? -res : res;
}

#endif /* LZZX_DETMATH_H */
