/*
 * ORACLE — TEST INFRASTRUCTURE ONLY.  Nothing in the product path may include,
 * link or call this file (only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg do).
 *
 * orc_fourier.c — CPU statement of the two transforms ulc-codec takes from
 * libfourier:
 *     Fourier_MDCT_MDST   call site /root/reference/libulc/ulcEncoder_BlockTransform.c:229-237
 *     Fourier_IMDCT       call sites /root/reference/libulc/ulcDecoder.c:243,249
 *
 * PARITY UNPINNED: libfourier (https://github.com/Aikku93/libfourier, see
 * /root/reference/.gitmodules:1-3) is an empty, un-vendored submodule whose pinned
 * commit is not recoverable from the mounted tree, and the reference holds no
 * test vectors at this boundary.  What is restated here is therefore the
 * *published contract* of the two functions:
 *   - IMDCT formula and sine window: /root/reference/FormatSpecs.md:150-157
 *   - argument meaning / buffer roles: the call sites above
 * and the float operation order is *defined by this project* ("fourier spec v2",
 * DESIGN.md §3).  The HIP kernels implement the same operation order; the
 * double-precision O(N^2) evaluators at the bottom of this file are the referee
 * for the 1e-5 accuracy bound.
 *
 * fourier spec v2  (round 3; v1 was the same structure with 4 products + 2 sums per
 * complex multiply, no fused operations and every twiddle multiplied)
 * ---------------
 *   DCT-IV of length N (unnormalised)   X[k] = sum_n u[n] cos(pi/N (n+1/2)(k+1/2))
 *   computed through one complex FFT of M = N/2 points:
 *     t[n]  = (u[2n] + i u[N-1-2n]) * conj(P[n]),  P[n] = exp(i pi (8n+1)/(8N))
 *     T     = FFT_M(t)    radix-2 decimation-in-frequency, in place, natural-order
 *                         input, bit-reversed output, twiddle W[j] = exp(-2 pi i j/M)
 *     y[k]  = T[k] * conj(P[k]);   X[2k] = Re y[k] = fma(T.im, s, T.re*c);
 *                                  X[N-1-2k] = -Im y[k] = fma(T.re, s, -(T.im*c))   (formed directly)
 *   complex multiply of d by conj(c + i s), two products and two fused multiply-adds
 *   (the real libfourier is an FMA build too, Makefile:125-145 of the reference):
 *       re = fma(d.im, s, d.re*c)         im = fma(-d.re, s, d.im*c)
 *   (on the device: v_pk_mul_f32 + v_pk_fma_f32).
 *   Butterflies of the last three stages (half-size h <= 4) whose twiddle is exactly
 *   1 (index 0) or -i (index M/4) are not multiplied: the difference passes through
 *   as it is, or as (d.im, -d.re).  Every other butterfly multiplies, also by a
 *   twiddle that happens to be trivial (stages h >= 8: one lane of a wave would
 *   otherwise take another path than the other 63).
 *   The inverse transform's overlap rotation is fused the same way:
 *       Out[p] = fma(c, A, -(s*B))        Out[N-1-p] = fma(s, A, c*B)
 *   All tables are evaluated in binary64 with the host libm and rounded once to
 *   binary32.  Built with -mfma: fmaf() is one instruction (and correctly rounded
 *   either way).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include "ulc_oracle.h"

#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif

/* ---- table cache (per transform length) ---------------------------------- */
typedef struct {
    int    N;          /* DCT-IV length */
    float *pre_c, *pre_s;   /* P[n], n < N/2 */
    float *tw_c,  *tw_s;    /* W[j] = (cos, sin)(2 pi j / M), j < M/2 (sin stored positive) */
    uint16_t *brev;    /* bit reversal over log2(M) bits */
} dct4_tab_t;

#define ORC_MAX_LOG2N 16
static dct4_tab_t g_tab[ORC_MAX_LOG2N + 1];

static int ilog2(int x) { int r = 0; while ((1 << r) < x) r++; return r; }

static const dct4_tab_t *get_tab(int N) {
    int lg = ilog2(N);
    dct4_tab_t *t = &g_tab[lg];
    if (t->N == N) return t;
    int M = N / 2;
    t->pre_c = (float *)malloc(sizeof(float) * M);
    t->pre_s = (float *)malloc(sizeof(float) * M);
    t->tw_c  = (float *)malloc(sizeof(float) * (M / 2 + 1));
    t->tw_s  = (float *)malloc(sizeof(float) * (M / 2 + 1));
    t->brev  = (uint16_t *)malloc(sizeof(uint16_t) * M);
    for (int n = 0; n < M; n++) {
        double th = M_PI * (double)(8 * n + 1) / (double)(8 * N);
        t->pre_c[n] = (float)cos(th);
        t->pre_s[n] = (float)sin(th);
    }
    for (int j = 0; j < M / 2; j++) {
        double th = 2.0 * M_PI * (double)j / (double)M;
        t->tw_c[j] = (float)cos(th);
        t->tw_s[j] = (float)sin(th);
    }
    int bits = ilog2(M);
    for (int i = 0; i < M; i++) {
        int r = 0;
        for (int b = 0; b < bits; b++) if (i & (1 << b)) r |= 1 << (bits - 1 - b);
        t->brev[i] = (uint16_t)r;
    }
    t->N = N;
    return t;
}

/* Sine-window ramps for an overlap of Ov samples (FormatSpecs.md:155):
 * fall[i] = cos(pi/2 (i+1/2)/Ov), rise[i] = sin(pi/2 (i+1/2)/Ov), i < Ov. */
void orc_window_tables(int Ov, float *fall, float *rise) {
    for (int i = 0; i < Ov; i++) {
        double th = M_PI * (double)(2 * i + 1) / (double)(4 * Ov);
        fall[i] = (float)cos(th);
        rise[i] = (float)sin(th);
    }
}

typedef struct { int Ov; float *fall, *rise; } win_tab_t;
static win_tab_t g_win[ORC_MAX_LOG2N + 1];
static const win_tab_t *get_win(int Ov) {
    int lg = ilog2(Ov);
    win_tab_t *w = &g_win[lg];
    if (w->Ov == Ov) return w;
    w->fall = (float *)malloc(sizeof(float) * Ov);
    w->rise = (float *)malloc(sizeof(float) * Ov);
    orc_window_tables(Ov, w->fall, w->rise);
    w->Ov = Ov;
    return w;
}

/* per-thread scratch, grown on demand (the oracle is also timed as the CPU
 * baseline, so no malloc per call) */
static __thread float *g_scratch = NULL;
static __thread int    g_scratch_n = 0;
static float *scratch(int nfloats) {
    if (nfloats > g_scratch_n) {
        free(g_scratch);
        g_scratch = (float *)malloc(sizeof(float) * (size_t)nfloats);
        g_scratch_n = nfloats;
    }
    return g_scratch;
}

/* ---- spec v2 DCT-IV ------------------------------------------------------- */
/* d * conj(c + i s): two products, two fused multiply-adds */
#define CMULC_FMA(OR, OI, DR, DI, C, S) do { float m0_ = (DR) * (C), m2_ = (DI) * (C); \
                                             (OR) = fmaf((DI), (S), m0_); (OI) = fmaf(-(DR), (S), m2_); } while (0)
/* x[] (re,im interleaved, M complex points) is transformed in place; output
 * index k lives at position brev[k]. */
static void fft_dif_inplace(float *x, int M, const dct4_tab_t *t) {
    for (int h = M / 2; h >= 1; h >>= 1) {
        int step = M / (2 * h);                 /* twiddle stride in W table */
        for (int base = 0; base < M; base += 2 * h) {
            for (int j = 0; j < h; j++) {
                int p = base + j, q = p + h;
                float ar = x[2*p], ai = x[2*p+1];
                float br = x[2*q], bi = x[2*q+1];
                float dr = ar - br, di = ai - bi;
                x[2*p]   = ar + br;
                x[2*p+1] = ai + bi;
                int idx = j * step;
                if (h <= 4 && idx == 0)          { x[2*q] = dr; x[2*q+1] = di; }            /* W = 1  */
                else if (h <= 4 && idx == M / 4) { x[2*q] = di; x[2*q+1] = -dr; }           /* W = -i */
                else CMULC_FMA(x[2*q], x[2*q+1], dr, di, t->tw_c[idx], t->tw_s[idx]);
            }
        }
    }
}

/* X <- DCT-IV(u).  u and X may alias; work holds N floats. */
void orc_dct4(float *X, const float *u, float *work, int N) {
    const dct4_tab_t *t = get_tab(N);
    int M = N / 2;
    for (int n = 0; n < M; n++) {
        float a = u[2*n], b = u[N-1-2*n];
        CMULC_FMA(work[2*n], work[2*n+1], a, b, t->pre_c[n], t->pre_s[n]);
    }
    fft_dif_inplace(work, M, t);
    for (int k = 0; k < M; k++) {
        int p = t->brev[k];
        float r = work[2*p], i = work[2*p+1];
        /* y = T conj(P):  X[2k] = Re y = fma(i, s, r c);  X[N-1-2k] = -Im y = fma(r, s, -(i c)), formed directly (the
         * negative of fma(-r, s, i c) differs from it in the sign of an exact zero, and compilers fold the one into the
         * other: the spec is the direct form) */
        float c = t->pre_c[k], s = t->pre_s[k];
        float m0 = r * c, m2 = i * c;
        X[2*k]     = fmaf(i, s, m0);
        X[N-1-2*k] = fmaf(r, s, -m2);
    }
}

/* ---- Fourier_MDCT_MDST ---------------------------------------------------- */
/* Frame f[0..2N): f[n] = Lap[n] (already windowed by the previous call),
 * f[N+n] = New[n]*fall(n).   MDCT[k] = -sum f[n] cos(pi/N (n+1/2+N/2)(k+1/2)),
 * MDST[k] = +sum f[n] sin(same).  Afterwards Lap[n] <- New[n]*rise(n).
 * fall(n) = 1 (n < a), ramp (a <= n < a+Ov), 0 after; a = (N-Ov)/2; rise(n) = fall(N-1-n).
 * New may alias Tmp (as at BlockTransform.c:232-234); Tmp holds >= N floats.
 * Folding (derivation in DESIGN.md §3):
 *   v[m] = R[N/2-1-m] + R[N/2+m]            (m <  N/2)    MDCT = DCT4(v)
 *   v[m] = Lap[3N/2-1-m] - Lap[m-N/2]       (m >= N/2)
 *   w[m] = R[N/2-1-m] - R[N/2+m]            (m <  N/2)    MDST[k] = (-1)^k DCT4(reverse(w))[k]
 *   w[m] = Lap[m-N/2] + Lap[3N/2-1-m]       (m >= N/2)
 */
void orc_mdct_mdst(float *MDCT, float *MDST, const float *New, float *Lap, float *Tmp, int N, int Overlap) {
    int a  = (N - Overlap) / 2;       /* fall: 1 on [0,a), ramp on [a,a+Ov), 0 after   */
    int a2 = N - a - Overlap;         /* rise: 0 on [0,a2), ramp on [a2,a2+Ov), 1 after */
    const win_tab_t *wt = Overlap ? get_win(Overlap) : NULL;
    float *S    = scratch(5 * N);
    float *R    = S;            /* windowed right half */
    float *nlap = S + N;
    float *v    = S + 2 * N;
    float *w    = S + 3 * N;
    float *work = S + 4 * N;
    for (int n = 0; n < N; n++) {
        float x = New[n];
        float fl = (n < a)  ? 1.0f : (n < a  + Overlap) ? wt->fall[n - a]  : 0.0f;
        float rs = (n < a2) ? 0.0f : (n < a2 + Overlap) ? wt->rise[n - a2] : 1.0f;
        R[n]    = x * fl;
        nlap[n] = x * rs;
    }
    int H = N / 2;
    for (int m = 0; m < H; m++) {
        v[m]     = R[H-1-m] + R[H+m];
        w[m]     = R[H-1-m] - R[H+m];
        v[H+m]   = Lap[N-1-m] - Lap[m];
        w[H+m]   = Lap[m] + Lap[N-1-m];
    }
    /* reverse w for the DST-IV-through-DCT-IV identity */
    for (int m = 0; m < H; m++) { float t0 = w[m]; w[m] = w[N-1-m]; w[N-1-m] = t0; }
    (void)Tmp;
    orc_dct4(MDCT, v, work, N);
    orc_dct4(MDST, w, work, N);
    for (int k = 1; k < N; k += 2) MDST[k] = -MDST[k];
    memcpy(Lap, nlap, sizeof(float) * N);
}

/* ---- Fourier_IMDCT -------------------------------------------------------- */
/* y[n] = -sum_k In[k] cos(pi/N (n+1/2+N/2)(k+1/2)) (FormatSpecs.md:150-153).
 * With z = DCT4(In):  y[N-1-p] = z[N/2+p],  y[p] = -z[N/2+p],  y[N+p] = z[N/2-1-p].
 * Lap holds the previous frame's un-windowed aliased tail time-reversed
 * (Lap[N/2-1-p] = y_prev[N+p]) as forced by the decoder FIFO
 * (/root/reference/libulc/ulcDecoder.c:253-272).  For pair p < N/2:
 *   A = Lap[N/2-1-p], B = z[N/2+p];
 *   p <  a :  Out[p] = A,           Out[N-1-p] = B
 *   p >= a :  Out[p] = fma(c,A,-(s*B)),  Out[N-1-p] = fma(s,A,c*B),  (c,s) = (fall,rise)[p-a]
 * then Lap[i] <- z[i], i < N/2.
 */
void orc_imdct(float *Out, const float *In, float *Lap, float *Tmp, int N, int Overlap) {
    int a = (N - Overlap) / 2;
    int H = N / 2;
    const win_tab_t *wt = Overlap ? get_win(Overlap) : NULL;
    float *z    = scratch(2 * N);
    float *work = z + N;
    (void)Tmp;
    orc_dct4(z, In, work, N);
    for (int p = 0; p < H; p++) {
        float A = Lap[H-1-p], B = z[H+p];
        if (p < a) {
            Out[p]     = A;
            Out[N-1-p] = B;
        } else {
            float c = wt->fall[p-a], s = wt->rise[p-a];
            float m1 = s * B, m3 = c * B;
            Out[p]     = fmaf(c, A, -m1);
            Out[N-1-p] = fmaf(s, A, m3);
        }
    }
    for (int i = 0; i < H; i++) Lap[i] = z[i];
}

/* ---- binary64 O(N^2) referees (direct evaluation of the published formulas) */
void orc_ref64_mdct_mdst(double *MDCT, double *MDST, const float *New, const double *Lap, double *LapOut, int N, int Overlap) {
    int a = (N - Overlap) / 2;
    double *f = (double *)malloc(sizeof(double) * 2 * N);
    for (int n = 0; n < N; n++) {
        double fl, rs;
        if (n < a) fl = 1.0; else if (n < a + Overlap) fl = cos(M_PI / 2 * (n - a + 0.5) / Overlap); else fl = 0.0;
        int nr = N - 1 - n;
        if (nr < a) rs = 1.0; else if (nr < a + Overlap) rs = cos(M_PI / 2 * (nr - a + 0.5) / Overlap); else rs = 0.0;
        f[n] = Lap[n];
        f[N + n] = (double)New[n] * fl;
        LapOut[n] = (double)New[n] * rs;
    }
    for (int k = 0; k < N; k++) {
        double sc = 0.0, ss = 0.0;
        for (int n = 0; n < 2 * N; n++) {
            double ph = M_PI / N * (n + 0.5 + N / 2.0) * (k + 0.5);
            sc += f[n] * cos(ph);
            ss += f[n] * sin(ph);
        }
        MDCT[k] = -sc;
        MDST[k] = ss;
    }
    free(f);
}

/* y[0..2N) = plain IMDCT of In per FormatSpecs.md:152-153 */
void orc_ref64_imdct_raw(double *y, const float *In, int N) {
    for (int n = 0; n < 2 * N; n++) {
        double s = 0.0;
        for (int k = 0; k < N; k++) s += (double)In[k] * cos(M_PI / N * (n + 0.5 + N / 2.0) * (k + 0.5));
        y[n] = -s;
    }
}
