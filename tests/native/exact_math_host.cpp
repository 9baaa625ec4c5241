// Host-side harness for tests/test_exact_math.py: compiles the PRODUCT header light-loam_amd/csrc/ll_exact_math.h with
// g++ (no FMA contraction) and compares it with the host libm the reference would call.
#include <cmath>
#include <cstdint>
#include <cstring>
#include "ll_exact_math.h"

static inline uint32_t pcg(uint64_t *s)
{
    const uint64_t o = *s; *s = o * 6364136223846793005ULL + 1442695040888963407ULL;
    const uint32_t x = (uint32_t)(((o >> 18) ^ o) >> 27), r = (uint32_t)(o >> 59);
    return (x >> r) | (x << ((-r) & 31));
}

extern "C" {

// atanf over bit patterns first, first+stride, ... (< 2^32): number of results that differ bitwise (NaNs compare equal)
long long em_check_atanf(unsigned long long first, unsigned long long stride, unsigned int *first_bad)
{
    long long bad = 0;
#pragma omp parallel for reduction(+ : bad) schedule(static)
    for (long long i = (long long)first; i < (1LL << 32); i += (long long)stride) {
        const float x = ll_u2f((uint32_t)i);
        const float a = atanf(x), b = ll_atanf(x);
        if (ll_f2u(a) != ll_f2u(b) && !(a != a && b != b)) {
            bad++;
#pragma omp critical
            { if (*first_bad == 0xffffffffu || (uint32_t)i < *first_bad) *first_bad = (uint32_t)i; }
        }
    }
    return bad;
}

// (float)((double)a / M_PI) for every f32 a (bit patterns first, first+stride, ...): mismatches of ll_div_pi_f32
long long em_check_div_pi(unsigned long long first, unsigned long long stride)
{
    long long bad = 0;
#pragma omp parallel for reduction(+ : bad) schedule(static)
    for (long long i = (long long)first; i < (1LL << 32); i += (long long)stride) {
        const float x = ll_u2f((uint32_t)i);
        const float a = (float)((double)x / M_PI), b = ll_div_pi_f32(x);
        if (ll_f2u(a) != ll_f2u(b) && !(a != a && b != b)) bad++;
    }
    return bad;
}

// atan2f on n pseudo-random pairs in four regimes (raw bit patterns, lidar-range values, mixed exponents, tiny y)
long long em_check_atan2f(long long n, unsigned long long seed)
{
    long long bad = 0;
#pragma omp parallel for reduction(+ : bad)
    for (int t = 0; t < 8; t++) {
        uint64_t s = seed + (uint64_t)t * 7919ULL;
        for (long long i = 0; i < n / 8; i++) {
            const uint32_t uy = pcg(&s), ux = pcg(&s);
            float y, x;
            switch (i & 3) {
                case 0: y = ll_u2f(uy); x = ll_u2f(ux); break;
                case 1: y = (int32_t)uy * (100.0f / 2147483648.0f); x = (int32_t)ux * (100.0f / 2147483648.0f); break;
                case 2: y = ll_u2f((uy & 0x807fffffu) | 0x3f800000u); x = ll_u2f((ux & 0x807fffffu) | ((120 + (ux >> 27)) << 23)); break;
                default: y = (int32_t)uy * (1.0f / 2147483648.0f); x = ll_u2f(ux); break;
            }
            const float a = atan2f(y, x), b = ll_atan2f(y, x);
            if (ll_f2u(a) != ll_f2u(b) && !(a != a && b != b)) bad++;
        }
    }
    return bad;
}

long long em_check_atan2f_specials(void)
{
    const uint32_t sp[] = {0, 0x80000000u, 0x3f800000u, 0xbf800000u, 0x7f800000u, 0xff800000u, 0x7fc00000u, 1, 0x80000001u, 0x00800000u,
                           0x7f7fffffu, 0xff7fffffu, 0x3f000000u, 0x40490fdbu, 0x4c000000u, 0x31000000u, 0x1e000000u, 0x5e000000u};
    const int ns = sizeof(sp) / 4;
    long long bad = 0;
    for (int i = 0; i < ns; i++)
        for (int j = 0; j < ns; j++) {
            const float y = ll_u2f(sp[i]), x = ll_u2f(sp[j]);
            const float a = atan2f(y, x), b = ll_atan2f(y, x);
            if (ll_f2u(a) != ll_f2u(b) && !(a != a && b != b)) bad++;
        }
    return bad;
}

// the vote predicate: for every non-negative f32 gap2 (bit patterns first, first+stride, ... <= +inf):
// (expf(-gap2 / (1*1)) < 0.96f) == ll_vote_incompatible(gap2)      (laserOdometry.cpp:239-242)
long long em_check_vote(unsigned long long first, unsigned long long stride)
{
    long long bad = 0;
#pragma omp parallel for reduction(+ : bad) schedule(static)
    for (long long u = (long long)first; u <= 0x7f800000LL; u += (long long)stride) {
        const float g2 = ll_u2f((uint32_t)u);
        const bool ref = expf(-g2 / (1.0f * 1.0f)) < 0.96f;
        if (ref != ll_vote_incompatible(g2)) bad++;
    }
    return bad;
}

}  // extern "C"
