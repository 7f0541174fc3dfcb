// Host-side harness: the product's libm restatement (ulcx_libm.h, compiled as plain
// C++ with -mfma so __builtin_fma is the hardware instruction) against the live libm.
#include <math.h>
#include <stdint.h>
#include "../../ulc-codec_amd/csrc/ulcx_libm.h"

static inline int same_f(float a, float b) { return ulcx_f2u(a) == ulcx_f2u(b) || (a != a && b != b); }
static inline int same_d(double a, double b) { return ulcx_d2u(a) == ulcx_d2u(b) || (a != a && b != b); }

extern "C" {
float  t_expf(float x) { return ulcx_expf(x); }
float  t_logf(float x) { return ulcx_logf(x); }
double t_log(double x) { return ulcx_log(x); }

// compares over bit patterns lo, lo+stride, ... < hi; returns mismatch count, first bad pattern in *bad
long long cmp_expf(uint64_t lo, uint64_t hi, uint64_t stride, uint32_t *bad) {
    long long n = 0;
    for (uint64_t u = lo; u < hi; u += stride) {
        float x = ulcx_u2f((uint32_t)u);
        if (!same_f(ulcx_expf(x), expf(x))) { if (!n && bad) *bad = (uint32_t)u; n++; }
    }
    return n;
}
long long cmp_logf(uint64_t lo, uint64_t hi, uint64_t stride, uint32_t *bad) {
    long long n = 0;
    for (uint64_t u = lo; u < hi; u += stride) {
        float x = ulcx_u2f((uint32_t)u);
        if (!same_f(ulcx_logf(x), logf(x))) { if (!n && bad) *bad = (uint32_t)u; n++; }
    }
    return n;
}
// double log on n pseudo-random positive inputs (xorshift64*), several magnitude regimes
long long cmp_log(uint64_t seed, long long n, uint64_t *bad) {
    long long m = 0;
    uint64_t s = seed ? seed : 1;
    for (long long i = 0; i < n; i++) {
        s ^= s >> 12; s ^= s << 25; s ^= s >> 27;
        uint64_t r = s * 0x2545F4914F6CDD1DULL;
        uint64_t bits;
        switch (i & 3) {
            case 0: bits = r & 0x7fefffffffffffffULL; break;                               // any positive finite
            case 1: bits = 0x3fe0000000000000ULL + (r % 0x0020000000000000ULL); break;      // [0.5, 2): near-1 branch
            case 2: bits = ulcx_d2u((double)ulcx_u2f((uint32_t)(r & 0x7f7fffff))); break;   // float-origin values
            default: bits = r & 0x000fffffffffffffULL; break;                              // subnormals
        }
        double x = ulcx_u2d(bits);
        if (!same_d(ulcx_log(x), log(x))) { if (!m && bad) *bad = bits; m++; }
    }
    return m;
}
}
