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

#if defined(__HIPCC__)
#define LL_HD __host__ __device__ __forceinline__
#else
#define LL_HD static inline
#endif

LL_HD uint32_t ll_f2u(float f) { return __builtin_bit_cast(uint32_t, f); }
LL_HD float ll_u2f(uint32_t u) { return __builtin_bit_cast(float, u); }

/* fdlibm's argument reduction picks one of four quotients; a wave of lidar points takes all of them, so the ranges
 * are evaluated branch-free: numerator / denominator of the taken range selected first, ONE division (the unreduced
 * range divides by 1, which is exact), both result formulas, selects.  Operation for operation the same f32
 * arithmetic as the branchy original, hence the same bits. */
LL_HD float ll_atanf(float x)
{
    const float aT0 = 3.3333334327e-01f, aT1 = -2.0000000298e-01f, aT2 = 1.4285714924e-01f,
                aT3 = -1.1111110449e-01f, aT4 = 9.0908870101e-02f, aT5 = -7.6918758452e-02f,
                aT6 = 6.6610731184e-02f, aT7 = -5.8335702866e-02f, aT8 = 4.9768779427e-02f,
                aT9 = -3.6531571299e-02f, aT10 = 1.6285819933e-02f;
    const int32_t hx = (int32_t)ll_f2u(x);
    const int32_t ix = hx & 0x7fffffff;
    const float ax = ll_u2f((uint32_t)ix);        /* fabsf */
    const bool r0 = ix < 0x3f300000, r1 = ix < 0x3f980000, r2 = ix < 0x401c0000, small = ix < 0x3ee00000;
    /*                     |x| < 0.6875                    < 1.1875          < 2.4375          else  */
    float num = r0 ? 2.0f * ax - 1.0f : (r1 ? ax - 1.0f : (r2 ? ax - 1.5f : -1.0f));
    float den = r0 ? 2.0f + ax        : (r1 ? ax + 1.0f : (r2 ? 1.0f + 1.5f * ax : ax));
    const float hi = r0 ? 4.6364760399e-01f : (r1 ? 7.8539812565e-01f : (r2 ? 9.8279368877e-01f : 1.5707962513e+00f));
    const float lo = r0 ? 5.0121582440e-09f : (r1 ? 3.7748947079e-08f : (r2 ? 3.4473217170e-08f : 7.5497894159e-08f));
    num = small ? x : num; den = small ? 1.0f : den;                /* |x| < 0.4375: no reduction, sign kept */
    const float t = num / den;
    const float z = t * t;
    const float w = z * z;
    const float s1 = z * (aT0 + w * (aT2 + w * (aT4 + w * (aT6 + w * (aT8 + w * aT10)))));
    const float s2 = w * (aT1 + w * (aT3 + w * (aT5 + w * (aT7 + w * aT9))));
    const float rs = t - t * (s1 + s2);
    const float rr = hi - ((t * (s1 + s2) - lo) - t);
    float r = small ? rs : (hx < 0 ? -rr : rr);
    if (ix < 0x31000000) r = x;                                     /* |x| < 2^-29 */
    if (ix >= 0x4c000000) {                                         /* |x| >= 2^25 */
        const float q = 1.5707962513e+00f + 7.5497894159e-08f;
        r = ix > 0x7f800000 ? x + x : (hx > 0 ? q : -q);           /* NaN : +-pi/2 */
    }
    return r;
}

/* (float)((double)a / M_PI) without the f64 division (scanRegistration.cpp:139: "* 180 / M_PI" on a float).
 * q = RN(a * RN(1/pi)); one FMA residual step gives the correctly rounded f64 quotient, which is then rounded to
 * f32 like the reference's store.  tests/test_exact_math.py checks all 2^32 floats against the real division. */
LL_HD float ll_div_pi_f32(float a)
{
    const double pi = 3.14159265358979323846, rpi = 0.31830988618379067154;
    const double x = (double)a;
    const double q = x * rpi;
    if (x == 0.0 || q - q != 0.0) return (float)q;                  /* +-0 keeps its sign; inf and NaN pass through */
    const double rem = __builtin_fma(-q, pi, x);
    return (float)__builtin_fma(rem, rpi, q);
}

/* x86-64 cvttsd2si semantics of the reference's int(double): NaN / out of range -> INT_MIN ("integer indefinite") */
LL_HD int ll_trunc_to_int(double v)
{
    if (!(v > -2147483649.0 && v < 2147483648.0)) return (int)0x80000000;
    return (int)v;
}

/* scanRegistration.cpp:139-168 as a function of t = z / sqrt(x*x + y*y): the UNCLAMPED ring number
 *   angle = atan(t) * 180 / M_PI (f32 atan, f32 product, f64 division, f32 store), then the 16 / 32 / 64-ring formula
 * (ring_model 1: the 64-ring linear formula for any ring count).  Every step is monotone non-decreasing in t -- ll_atanf
 * is (checked over all floats by tests/test_exact_math.py) -- so the ring of a point is found by comparing t with R + 1
 * precomputed thresholds instead of evaluating this chain per point (ll_ring_thresholds / k_classify). */
LL_HD int ll_ring_of_t(float t, int ring_model, int R, float lower_bound, float factor)
{
    const float angle = ll_div_pi_f32(ll_atanf(t) * 180.0f);
    if (ring_model == 0 && R == 16) return ll_trunc_to_int((double)((angle + 15.0f) / 2.0f) + 0.5);        /* :144 */
    if (ring_model == 0 && R == 32) return ll_trunc_to_int(((double)angle + 92.0 / 3.0) * 3.0 / 4.0);       /* :153 */
    return ll_trunc_to_int((double)((angle - lower_bound) * factor) + 0.5);                                  /* :162 */
}

/* order-preserving int key of a float (-inf .. -0 < +0 .. +inf; NaNs beyond the infinities) */
LL_HD int32_t ll_float_key(float f) { const int32_t i = (int32_t)ll_f2u(f); return i >= 0 ? i : (int32_t)(i ^ 0x7fffffff); }
LL_HD float ll_key_float(int32_t k) { return ll_u2f((uint32_t)(k >= 0 ? k : (k ^ 0x7fffffff))); }

/* thr[k], k = 0 .. R: key of the smallest float t with ll_ring_of_t(t) >= k (INT32_MAX if there is none).
 * A point's ring is  #{k : key(t) >= thr[k]} - 1  when that is in 0 .. R-1, else the point is rejected (:164-168). */
LL_HD void ll_ring_thresholds(int ring_model, int R, float lower_bound, float factor, int32_t *thr)
{
    const int32_t kmin = ll_float_key(ll_u2f(0xff800000u)), kmax = ll_float_key(ll_u2f(0x7f800000u));      /* -inf, +inf */
    for (int k = 0; k <= R; ++k) {
        if (ll_ring_of_t(ll_key_float(kmax), ring_model, R, lower_bound, factor) < k) { thr[k] = 0x7fffffff; continue; }
        int32_t lo = kmin, hi = kmax;                                  /* invariant: ring(hi) >= k; find the smallest such key */
        while (lo < hi) {
            const int32_t mid = (int32_t)(((int64_t)lo + (int64_t)hi) >> 1);
            if (ll_ring_of_t(ll_key_float(mid), ring_model, R, lower_bound, factor) >= k) hi = mid; else lo = mid + 1;
        }
        thr[k] = lo;
    }
}

/* A first guess for the threshold search: t is cut into nb equal buckets between the first and the last threshold,
 * lut[b] = the ring of the smallest t that falls into bucket b, and a point's ring is lut[b] plus the number of the next
 * two thresholds not above t.  The bucket function is a chain of monotone f32 operations, so every bucket is an interval
 * of floats and ll_ring_lut_build can check -- exactly, at context creation -- that no bucket spans more than three
 * rings; if one does (exotic ring parameters) it returns false and k_classify keeps the binary search. */
#define LL_RING_LUT_MAX 1024
LL_HD int ll_ring_bucket(float t, float t0, float scale, int nb)
{
    const float u = (t - t0) * scale;
    int b = (int)(u < 0.0f ? 0.0f : (u > (float)(nb - 1) ? (float)(nb - 1) : u));        /* NaN never gets here */
    return b;
}
LL_HD int ll_ring_count(const int32_t *thr, int R, int32_t key)                           /* #{k <= R : thr[k] <= key} - 1 */
{
    int lo = 0, hi = R + 1;
    while (lo < hi) { const int mid = (lo + hi) >> 1; if (thr[mid] <= key) lo = mid + 1; else hi = mid; }
    return lo - 1;
}
/* out: lut[nb] (rings as ints), t0, scale; returns nb, or 0 when the guess cannot be used */
LL_HD int ll_ring_lut_build(const int32_t *thr, int R, int32_t *lut, float *t0_out, float *scale_out)
{
    const int32_t kinf_lo = ll_float_key(ll_u2f(0xff800000u)), kinf_hi = ll_float_key(ll_u2f(0x7f800000u));
    if (thr[0] <= kinf_lo || thr[R] >= kinf_hi || thr[0] >= thr[R]) return 0;             /* unbounded or empty ring range */
    int nb = 8 * R; if (nb > LL_RING_LUT_MAX) nb = LL_RING_LUT_MAX;
    const float t0 = ll_key_float(thr[0]), t1 = ll_key_float(thr[R]);
    const float scale = (float)(nb - 1) / (t1 - t0);
    if (!(scale > 0.0f) || !(scale < 3.0e38f)) return 0;
    int32_t first_key = kinf_lo;                                                           /* smallest key of bucket b */
    for (int b = 0; b < nb; ++b) {
        /* the smallest key whose bucket is >= b + 1 (monotone bucket function: binary search over all finite floats) */
        int32_t lo = first_key, hi = kinf_hi;
        while (lo < hi) {
            const int32_t mid = (int32_t)(((int64_t)lo + (int64_t)hi) >> 1);
            if (ll_ring_bucket(ll_key_float(mid), t0, scale, nb) >= b + 1) hi = mid; else lo = mid + 1;
        }
        const int32_t last_key = (b == nb - 1) ? kinf_hi : lo - 1;                        /* the last bucket runs to +inf */
        if (last_key < first_key) { lut[b] = b ? lut[b - 1] : 0; first_key = lo; continue; }   /* no float lands here */
        /* rings below 0 (t under the first threshold) are rejected before the guess is used: start at the first threshold */
        const int32_t fk = first_key < thr[0] ? thr[0] : first_key;
        int g = ll_ring_count(thr, R, fk);
        if (g < 0) g = 0;
        if (g > R - 1) g = R - 1;                                                          /* thr[g + 2] stays inside the padded array */
        const int top = ll_ring_count(thr, R, last_key);
        if (last_key >= thr[0] && top > g + 2) return 0;
        lut[b] = g;
        first_key = lo;
    }
    *t0_out = t0; *scale_out = scale;
    return nb;
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

/* ll_atan2f for the common case, specials behind one rarely taken branch: both operands finite and non-zero, the
 * quotient's exponent within +-60 and |y / x| inside ll_atanf's regular range [2^-29, 2^25).  What is left is the
 * division, the reduction + polynomial of ll_atanf for a positive argument, and the quadrant fix -- the same f32
 * operations in the same order (tests/test_exact_math.py compares the two functions over edge cases and a random sweep). */
LL_HD float ll_atan2f_finite(float y, float x)
{
    const float pi = 3.1415927410e+00f, pi_lo = -8.7422776573e-08f;
    const int32_t hx = (int32_t)ll_f2u(x), hy = (int32_t)ll_f2u(y);
    const int32_t ix = hx & 0x7fffffff, iy = hy & 0x7fffffff;
    const int k = (iy - ix) >> 23;
    const float q = ll_u2f(ll_f2u(y / x) & 0x7fffffffu);
    const int32_t iq = (int32_t)ll_f2u(q);
    if (ix == 0 || iy == 0 || hx == 0x3f800000 || k > 60 || k < -60 || iq < 0x31000000 || iq >= 0x4c000000) return ll_atan2f(y, x);
    /* ll_atanf(q), q > 0 in the regular range */
    const float aT0 = 3.3333334327e-01f, aT1 = -2.0000000298e-01f, aT2 = 1.4285714924e-01f,
                aT3 = -1.1111110449e-01f, aT4 = 9.0908870101e-02f, aT5 = -7.6918758452e-02f,
                aT6 = 6.6610731184e-02f, aT7 = -5.8335702866e-02f, aT8 = 4.9768779427e-02f,
                aT9 = -3.6531571299e-02f, aT10 = 1.6285819933e-02f;
    const bool r0 = iq < 0x3f300000, r1 = iq < 0x3f980000, r2 = iq < 0x401c0000, small = iq < 0x3ee00000;
    float num = r0 ? 2.0f * q - 1.0f : (r1 ? q - 1.0f : (r2 ? q - 1.5f : -1.0f));
    float den = r0 ? 2.0f + q        : (r1 ? q + 1.0f : (r2 ? 1.0f + 1.5f * q : q));
    const float hi = r0 ? 4.6364760399e-01f : (r1 ? 7.8539812565e-01f : (r2 ? 9.8279368877e-01f : 1.5707962513e+00f));
    const float lo = r0 ? 5.0121582440e-09f : (r1 ? 3.7748947079e-08f : (r2 ? 3.4473217170e-08f : 7.5497894159e-08f));
    num = small ? q : num; den = small ? 1.0f : den;
    const float t = num / den;
    const float z2 = t * t;
    const float w = z2 * z2;
    const float s1 = z2 * (aT0 + w * (aT2 + w * (aT4 + w * (aT6 + w * (aT8 + w * aT10)))));
    const float s2 = w * (aT1 + w * (aT3 + w * (aT5 + w * (aT7 + w * aT9))));
    const float rs = t - t * (s1 + s2);
    const float rr = hi - ((t * (s1 + s2) - lo) - t);
    const float z = small ? rs : rr;
    const int m = ((hy >> 31) & 1) | ((hx >> 30) & 2);              /* 2*sign(x) + sign(y) */
    if (m == 0) return z;
    if (m == 1) return ll_u2f(ll_f2u(z) ^ 0x80000000u);
    if (m == 2) return pi - (z - pi_lo);
    return (z - pi_lo) - pi;
}

/* the smallest float >= c / the largest float <= c: for a float a,  a < c  <=>  a < ll_f32_ceil(c)  and
 * a > c  <=>  a > ll_f32_floor(c), which turns the reference's float-against-double comparisons into f32 compares */
LL_HD float ll_f32_ceil(double c)
{
    float f = (float)c;
    if ((double)f < c) { const uint32_t b = ll_f2u(f); f = (f == 0.0f) ? ll_u2f(1u) : ll_u2f(f > 0.0f ? b + 1u : b - 1u); }
    return f;
}
LL_HD float ll_f32_floor(double c)
{
    float f = (float)c;
    if ((double)f > c) { const uint32_t b = ll_f2u(f); f = (f == 0.0f) ? ll_u2f(0x80000001u) : ll_u2f(f > 0.0f ? b - 1u : b + 1u); }
    return f;
}

/* graph vote predicate: std::exp(-(gap*gap)/1.0f) < 0.96f  (laserOdometry.cpp:239-242).
 * With glibc's expf this is exactly gap2 >= 0x3d273506 for every non-negative f32 gap2 (monotone;
 * verified over all 2^31 inputs by tests/test_exact_math.py against the host libm). */
#define LL_VOTE_GAP2_BITS 0x3d273506u
LL_HD bool ll_vote_incompatible(float gap2) { return ll_f2u(gap2) >= LL_VOTE_GAP2_BITS && !(gap2 != gap2); }
/* the same predicate on gap = fabsf(s1 - s2) itself: RN(gap * gap) is monotone in gap, and 0x3e4ee4cd is the smallest float
 * whose square rounds to at least the threshold above (tests/test_exact_math.py: every non-negative float) */
#define LL_VOTE_GAP_BITS 0x3e4ee4cdu
LL_HD bool ll_vote_incompatible_gap(float gap) { return ll_f2u(gap) >= LL_VOTE_GAP_BITS && !(gap != gap); }
