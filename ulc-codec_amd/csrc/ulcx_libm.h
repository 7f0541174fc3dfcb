// ulcx_libm.h — bit-faithful restatements of the three libm functions the reference
// calls at data-dependent points of the hot path:
//   expf : libulc/ulcEncoder_Psyopt.c:245, ulcEncoder_NoiseFill.c:30,80,81
//   logf : libulc/ulcEncoder_BlockTransform.c:321, ulcEncoder_Encode.c:83,
//          ulcEncoder_WindowControl.c:194-195
//   log  : libulc/ulcEncoder_Psyopt.c:133,220   (binary64)
// The reference gets them from the host's glibc; a GPU has no glibc, so to keep the
// packed stream bit-identical the kernels evaluate the same algorithm glibc 2.35
// runs on an x86-64 host with FMA (ifunc variants __expf_fma/__logf_fma/__log_fma):
// binary64 table + polynomial evaluation with exactly these fused multiply-adds
// (operation order read off the shipped objects; tables lifted by
// tools/gen_libm_tables.py).  Compiles as plain C++ on the host too
// (tests/test_libm_restatement.py compares against the live libm).
#pragma once
#include <stdint.h>
#include <string.h>

#if defined(__HIPCC__)
#define ULCX_DEV __device__ __forceinline__
#define ULCX_TAB static __device__ const
#else
#define ULCX_DEV static inline
#define ULCX_TAB static const
#endif
#include "ulcx_libm_tables.h"

ULCX_DEV uint32_t ulcx_f2u(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
ULCX_DEV float ulcx_u2f(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }
ULCX_DEV uint64_t ulcx_d2u(double d) { uint64_t u; memcpy(&u, &d, 8); return u; }
ULCX_DEV double ulcx_u2d(uint64_t u) { double d; memcpy(&d, &u, 8); return d; }
ULCX_DEV double ulcx_fma(double a, double b, double c) { return __builtin_fma(a, b, c); }

// glibc 2.35 sysdeps/ieee754/flt-32/e_expf.c (EXP2F_TABLE_BITS = 5); TAB: where the 32-entry 2^(i/32) table is read from (a kernel
// that evaluates many of these may keep a copy in LDS: the look-up sits in the middle of the dependent chain)
template <typename TAB>
ULCX_DEV float ulcx_expf_t(float x, TAB tab) {
    uint32_t ix = ulcx_f2u(x);
    uint32_t abstop = (ix >> 20) & 0x7ff;
    if (abstop >= 0x42b) {                       // |x| >= 88 or NaN
        if (ix == 0xff800000u) return 0.0f;
        if (abstop >= 0x7f8) return x + x;
        if (x > 0x1.62e42ep6f) return ulcx_u2f(0x7f800000u);      // overflow
        if (x < -0x1.9fe368p6f) return 0.0f;                      // underflow
        if (x < -0x1.9d1d9ep6f) return 0x1p-149f;                 // __math_may_uflowf
    }
    const double SHIFT = ulcx_u2d(ulcx_expf_consts[0]);
    const double InvLn2N = ulcx_u2d(ulcx_expf_consts[1]);
    const double C0 = ulcx_u2d(ulcx_expf_consts[2]);
    const double C1 = ulcx_u2d(ulcx_expf_consts[3]);
    const double C2 = ulcx_u2d(ulcx_expf_consts[4]);
    double xd = (double)x;
    double zs = ulcx_fma(InvLn2N, xd, SHIFT);
    uint64_t ki = ulcx_d2u(zs);
    double kd = zs - SHIFT;
    double r = ulcx_fma(InvLn2N, xd, -kd);
    uint64_t t = tab[ki & 31] + (ki << 47);
    double s = ulcx_u2d(t);
    double z = ulcx_fma(r, C0, C1);
    double r2 = r * r;
    double y = ulcx_fma(r, C2, 1.0);
    y = ulcx_fma(z, r2, y);
    y = y * s;
    return (float)y;
}
ULCX_DEV float ulcx_expf(float x) {
    uint32_t ix = ulcx_f2u(x);
    uint32_t abstop = (ix >> 20) & 0x7ff;
    if (abstop >= 0x42b) {                       // |x| >= 88 or NaN
        if (ix == 0xff800000u) return 0.0f;
        if (abstop >= 0x7f8) return x + x;
        if (x > 0x1.62e42ep6f) return ulcx_u2f(0x7f800000u);      // overflow
        if (x < -0x1.9fe368p6f) return 0.0f;                      // underflow
        if (x < -0x1.9d1d9ep6f) return 0x1p-149f;                 // __math_may_uflowf
    }
    const double SHIFT = ulcx_u2d(ulcx_expf_consts[0]);
    const double InvLn2N = ulcx_u2d(ulcx_expf_consts[1]);
    const double C0 = ulcx_u2d(ulcx_expf_consts[2]);
    const double C1 = ulcx_u2d(ulcx_expf_consts[3]);
    const double C2 = ulcx_u2d(ulcx_expf_consts[4]);
    double xd = (double)x;
    double zs = ulcx_fma(InvLn2N, xd, SHIFT);
    uint64_t ki = ulcx_d2u(zs);
    double kd = zs - SHIFT;
    double r = ulcx_fma(InvLn2N, xd, -kd);
    uint64_t t = ulcx_exp2f_tab[ki & 31] + (ki << 47);
    double s = ulcx_u2d(t);
    double z = ulcx_fma(r, C0, C1);
    double r2 = r * r;
    double y = ulcx_fma(r, C2, 1.0);
    y = ulcx_fma(z, r2, y);
    y = y * s;
    return (float)y;
}

// glibc 2.35 sysdeps/ieee754/flt-32/e_logf.c (LOGF_TABLE_BITS = 4)
ULCX_DEV float ulcx_logf(float x) {
    uint32_t ix = ulcx_f2u(x);
    if (ix == 0x3f800000u) return 0.0f;
    if (ix - 0x00800000u >= 0x7f800000u - 0x00800000u) {
        if (ix * 2 == 0) return ulcx_u2f(0xff800000u);            // log(0) = -inf
        if (ix == 0x7f800000u) return x;                          // log(inf) = inf
        if ((ix & 0x80000000u) || ix * 2 >= 0xff000000u) return ulcx_u2f(0x7fc00000u) ;  // invalid -> NaN
        ix = ulcx_f2u(x * 0x1p23f);                               // subnormal: normalise
        ix -= 23u << 23;
    }
    uint32_t tmp = ix - 0x3f330000u;
    int i = (tmp >> 19) & 15;
    int k = (int32_t)tmp >> 23;
    uint32_t iz = ix - (tmp & 0xff800000u);
    double invc = ulcx_u2d(ulcx_logf_tab[2 * i]);
    double logc = ulcx_u2d(ulcx_logf_tab[2 * i + 1]);
    const double Ln2 = ulcx_u2d(ulcx_logf_consts[0]);
    const double A0 = ulcx_u2d(ulcx_logf_consts[1]);
    const double A1 = ulcx_u2d(ulcx_logf_consts[2]);
    const double A2 = ulcx_u2d(ulcx_logf_consts[3]);
    double z = (double)ulcx_u2f(iz);
    double y0 = ulcx_fma((double)k, Ln2, logc);
    double r = ulcx_fma(z, invc, -1.0);
    double y = ulcx_fma(r, A1, A2);
    double r2 = r * r;
    double hi = r + y0;
    y = ulcx_fma(A0, r2, y);
    y = ulcx_fma(r2, y, hi);
    return (float)y;
}

// glibc 2.35 sysdeps/ieee754/dbl-64/e_log.c (LOG_TABLE_BITS = 7), __FP_FAST_FMA path
ULCX_DEV double ulcx_log(double x) {
    uint64_t ix = ulcx_d2u(x);
    uint32_t top = (uint32_t)(ix >> 48);
    const double *K = (const double *)0; (void)K;
#define ULCX_LC(i) ulcx_u2d(ulcx_log_consts[i])
    if (ix - 0x3fee000000000000ULL < 0x0003090000000000ULL) {     // near 1.0
        if (ix == 0x3ff0000000000000ULL) return 0.0;
        double r = x - 1.0;
        double r2 = r * r;
        double r3 = r * r2;
        // B[i] = ULCX_LC(7 + i)
        double p1 = ulcx_fma(ULCX_LC(10), r2, ulcx_fma(r, ULCX_LC(9), ULCX_LC(8)));      // B1 + r B2 + r2 B3
        double p2 = ulcx_fma(ULCX_LC(13), r2, ulcx_fma(r, ULCX_LC(12), ULCX_LC(11)));    // B4 + r B5 + r2 B6
        double p3 = ulcx_fma(r2, ULCX_LC(16), ulcx_fma(r, ULCX_LC(15), ULCX_LC(14)));    // B7 + r B8 + r2 B9
        p3 = ulcx_fma(ULCX_LC(17), r3, p3);                                              //  + r3 B10
        double q = ulcx_fma(p3, r3, p2);
        q = ulcx_fma(q, r3, p1);
        double w = ulcx_fma(r, 0x1p27, r);            // r + r*2^27
        double rhi = ulcx_fma(-0x1p27, r, w);         // ... - r*2^27
        double B0 = ULCX_LC(7);
        double rhi2 = rhi * rhi;
        double rlo = r - rhi;
        double hi = ulcx_fma(rhi2, B0, r);
        double lo = ulcx_fma(rhi2, B0, r - hi);
        double t = B0 * rlo;
        lo = ulcx_fma(t, r + rhi, lo);
        double y = ulcx_fma(q, r3, lo);
        return hi + y;
    }
    if (top - 0x0010u >= 0x7ff0u - 0x0010u) {
        if (ix * 2 == 0) return ulcx_u2d(0xfff0000000000000ULL);
        if (ix == 0x7ff0000000000000ULL) return x;
        if ((top & 0x8000u) || (top & 0x7ff0u) == 0x7ff0u) return ulcx_u2d(0x7ff8000000000000ULL);
        ix = ulcx_d2u(x * 0x1p52);
        ix -= 52ULL << 52;
    }
    uint64_t tmp = ix - 0x3fe6000000000000ULL;
    int i = (int)((tmp >> 45) & 127);
    int k = (int)((int64_t)tmp >> 52);
    uint64_t iz = ix - (tmp & (0xfffULL << 52));
    double invc = ulcx_u2d(ulcx_log_tab[2 * i]);
    double logc = ulcx_u2d(ulcx_log_tab[2 * i + 1]);
    double z = ulcx_u2d(iz);
    double kd = (double)k;
    double r = ulcx_fma(z, invc, -1.0);
    double w = ulcx_fma(kd, ULCX_LC(0), logc);        // kd*Ln2hi + logc
    double pa = ulcx_fma(r, ULCX_LC(4), ULCX_LC(3));  // A1 + r A2
    double hi = r + w;
    double r2 = r * r;
    double lo = (w - hi) + r;
    lo = ulcx_fma(ULCX_LC(1), kd, lo);                // + kd*Ln2lo
    double r3 = r * r2;
    double pb = ulcx_fma(r, ULCX_LC(6), ULCX_LC(5));  // A3 + r A4
    lo = ulcx_fma(ULCX_LC(2), r2, lo);                // + r2 A0
    double p = ulcx_fma(pb, r2, pa);
    double y = ulcx_fma(r3, p, lo);
    return y + hi;
#undef ULCX_LC
}
