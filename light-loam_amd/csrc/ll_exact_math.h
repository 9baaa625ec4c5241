/*
 * ll_exact_math.h -- bit-exact f32 atan / atan2 for the ring + azimuth assignment (a1).
 *
 * The reference calls the host libm (scanRegistration.cpp:114-117, :139, :177); scanID and
 * int(intensity) depend on those results at bin edges, so the device must return the same bits.
 * These are restatements of the fdlibm float algorithms that glibc 2.35 ships for atanf / atan2f
 * (sysdeps/ieee754/flt-32/s_atanf.c, e_atan2f.c; no FMA variant exists for them on x86-64), written
 * with plain IEEE f32 add/mul/div only.  tests/test_exact_math.py checks them against the host libm:
 * atanf over all 2^32 inputs (slow marker) or a 2^26 stratified subset, atan2f on 10^8 pairs.
 * Compile with -ffp-contract=off: a fused multiply-add anywhere below changes results.
 */
#pragma once
#include <stdint.h>

#if defined(__HIPCC__) || defined(__CUDACC__)
#define LL_HD __host__ __device__ __forceinline__
#else
#define LL_HD static inline
#endif

LL_HD uint32_t ll_f2u(float f) { return __builtin_bit_cast(uint32_t, f); }
LL_HD float ll_u2f(uint32_t u) { return __builtin_bit_cast(float, u); }

LL_HD float ll_atanf(float x)
{
    const float aT0 = 3.3333334327e-01f, aT1 = -2.0000000298e-01f, aT2 = 1.4285714924e-01f,
                aT3 = -1.1111110449e-01f, aT4 = 9.0908870101e-02f, aT5 = -7.6918758452e-02f,
                aT6 = 6.6610731184e-02f, aT7 = -5.8335702866e-02f, aT8 = 4.9768779427e-02f,
                aT9 = -3.6531571299e-02f, aT10 = 1.6285819933e-02f;
    const int32_t hx = (int32_t)ll_f2u(x);
    const int32_t ix = hx & 0x7fffffff;
    if (ix >= 0x4c000000) {                       /* |x| >= 2^25 */
        if (ix > 0x7f800000) return x + x;        /* NaN */
        const float r = 1.5707962513e+00f + 7.5497894159e-08f;
        return hx > 0 ? r : -r;
    }
    int id;
    float hi = 0.0f, lo = 0.0f;
    if (ix < 0x3ee00000) {                        /* |x| < 0.4375 */
        if (ix < 0x31000000) return x;            /* |x| < 2^-29 */
        id = -1;
    } else {
        x = ll_u2f((uint32_t)ix);                 /* fabsf */
        if (ix < 0x3f980000) {                    /* |x| < 1.1875 */
            if (ix < 0x3f300000) { id = 0; hi = 4.6364760399e-01f; lo = 5.0121582440e-09f; x = (2.0f * x - 1.0f) / (2.0f + x); }
            else                 { id = 1; hi = 7.8539812565e-01f; lo = 3.7748947079e-08f; x = (x - 1.0f) / (x + 1.0f); }
        } else {
            if (ix < 0x401c0000) { id = 2; hi = 9.8279368877e-01f; lo = 3.4473217170e-08f; x = (x - 1.5f) / (1.0f + 1.5f * x); }
            else                 { id = 3; hi = 1.5707962513e+00f; lo = 7.5497894159e-08f; x = -1.0f / x; }
        }
    }
    const float z = x * x;
    const float w = z * z;
    const float s1 = z * (aT0 + w * (aT2 + w * (aT4 + w * (aT6 + w * (aT8 + w * aT10)))));
    const float s2 = w * (aT1 + w * (aT3 + w * (aT5 + w * (aT7 + w * aT9))));
    if (id < 0) return x - x * (s1 + s2);
    const float r = hi - ((x * (s1 + s2) - lo) - x);
    return hx < 0 ? -r : r;
}

LL_HD float ll_atan2f(float y, float x)
{
    const float tiny = 1.0e-30f, pi_o_4 = 7.8539818525e-01f, pi_o_2 = 1.5707963705e+00f,
                pi = 3.1415927410e+00f, pi_lo = -8.7422776573e-08f;
    const int32_t hx = (int32_t)ll_f2u(x), hy = (int32_t)ll_f2u(y);
    const int32_t ix = hx & 0x7fffffff, iy = hy & 0x7fffffff;
    if (ix > 0x7f800000 || iy > 0x7f800000) return x + y;          /* NaN */
    if (hx == 0x3f800000) return ll_atanf(y);                       /* x == 1.0 */
    const int m = ((hy >> 31) & 1) | ((hx >> 30) & 2);              /* 2*sign(x) + sign(y) */
    if (iy == 0) {
        if (m < 2) return y;
        return m == 2 ? pi + tiny : -pi - tiny;
    }
    if (ix == 0) return hy < 0 ? -pi_o_2 - tiny : pi_o_2 + tiny;
    if (ix == 0x7f800000) {
        if (iy == 0x7f800000) {
            if (m == 0) return pi_o_4 + tiny;
            if (m == 1) return -pi_o_4 - tiny;
            if (m == 2) return 3.0f * pi_o_4 + tiny;
            return -3.0f * pi_o_4 - tiny;
        }
        if (m == 0) return 0.0f;
        if (m == 1) return -0.0f;
        if (m == 2) return pi + tiny;
        return -pi - tiny;
    }
    if (iy == 0x7f800000) return hy < 0 ? -pi_o_2 - tiny : pi_o_2 + tiny;
    const int k = (iy - ix) >> 23;
    float z;
    if (k > 60) z = pi_o_2 + 0.5f * pi_lo;                          /* |y/x| > 2^60 */
    else if (hx < 0 && k < -60) z = 0.0f;                           /* |y|/x < -2^60 */
    else z = ll_atanf(ll_u2f(ll_f2u(y / x) & 0x7fffffffu));
    if (m == 0) return z;
    if (m == 1) return ll_u2f(ll_f2u(z) ^ 0x80000000u);
    if (m == 2) return pi - (z - pi_lo);
    return (z - pi_lo) - pi;
}

/* graph vote predicate: std::exp(-(gap*gap)/1.0f) < 0.96f  (laserOdometry.cpp:239-242).
 * With glibc's expf this is exactly gap2 >= 0x3d273506 for every non-negative f32 gap2 (monotone;
 * verified over all 2^31 inputs by tests/test_exact_math.py against the host libm). */
#define LL_VOTE_GAP2_BITS 0x3d273506u
LL_HD bool ll_vote_incompatible(float gap2) { return ll_f2u(gap2) >= LL_VOTE_GAP2_BITS && !(gap2 != gap2); }
