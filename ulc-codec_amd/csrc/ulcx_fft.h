// ulcx_fft.h — LDS-resident complex FFT used by the forward and inverse transforms.
// "fourier spec v2" (oracle/orc_fourier.c): radix-2 decimation in
// frequency, in place, natural-order input, bit-reversed output, twiddle table
// W[j] = (cos, sin)(2 pi j / M).  A complex multiply d * conj(w) is two products and two
// fused multiply-adds:  re = fma(d.y, w.y, d.x*w.x),  im = fma(-d.x, w.y, d.y*w.x)
// (explicit: the file is compiled with -ffp-contract=off, the compiler fuses nothing by
// itself).  In the last three stages (half-size <= 4) a butterfly whose twiddle is
// exactly 1 (index 0) or -i (index M/4) is not multiplied: the difference passes as it
// is, or as (d.y, -d.x).
// Requires WG (workgroup size) to be defined by the includer.
#pragma once
#include <hip/hip_runtime.h>

__device__ __forceinline__ float2 cmulc(float2 d, float2 w) {    // d * conj(w): 2 products + 2 fused multiply-adds (spec v2)
    const float m0 = d.x * w.x, m2 = d.y * w.x;
    return make_float2(__builtin_fmaf(d.y, w.y, m0), __builtin_fmaf(-d.x, w.y, m2));
}
// post-twiddle of the DCT-IV (orc_dct4): (Re y, -Im y) of y = d * conj(w), the second formed directly as fma(d.x, w.y, -(d.y*w.x))
__device__ __forceinline__ float2 cmulc_post(float2 d, float2 w) {
    const float m0 = d.x * w.x, m2 = d.y * w.x;
    return make_float2(__builtin_fmaf(d.y, w.y, m0), __builtin_fmaf(d.x, w.y, -m2));
}
// the butterfly's lower output for a stage of half-size hs (in points) and twiddle index idx of an M-point transform:
// exact rotations for the trivial twiddles of the last three stages, the multiply everywhere else (per-thread selects:
// the workgroup-wide transforms are not the throughput paths)
__device__ __forceinline__ float2 fft_twid(float2 d, int idx, int hs, int M, const float2 *__restrict__ tw) {
    const float2 r = cmulc(d, tw[idx]);
    if (hs > 4) return r;
    return idx == 0 ? d : (idx * 4 == M ? make_float2(d.y, -d.x) : r);
}


// In-place radix-2 DIF FFT of two M-point arrays held in LDS ("fourier spec v2",
// oracle/orc_fourier.c), executed as merged radix-2^2 passes: each thread carries four
// points through two consecutive radix-2 stages in registers, which is arithmetically
// identical to the two separate stages.
__device__ void fft2_dif(float2 *za, float2 *zb, int M, const float2 *__restrict__ tw, int tid) {
    int h = M >> 1;
    int quarter = M >> 2;
    while (h >= 2) {
        int q = h >> 1;
        int stepA = M / (2 * h), stepB = stepA * 2;
        for (int g = tid; g < 2 * quarter; g += WG) {
            float2 *z = (g < quarter) ? za : zb;
            int gg = (g < quarter) ? g : g - quarter;
            int j = gg & (q - 1);
            int p0 = (gg - j) * 4 + j;            // (gg / q) * 2h + j, 2h = 4q
            int p1 = p0 + q, p2 = p0 + h, p3 = p2 + q;
            float2 x0 = z[p0], x1 = z[p1], x2 = z[p2], x3 = z[p3];
            float2 y0 = make_float2(x0.x + x2.x, x0.y + x2.y);
            float2 y2 = fft_twid(make_float2(x0.x - x2.x, x0.y - x2.y), j * stepA, h, M, tw);
            float2 y1 = make_float2(x1.x + x3.x, x1.y + x3.y);
            float2 y3 = fft_twid(make_float2(x1.x - x3.x, x1.y - x3.y), (j + q) * stepA, h, M, tw);
            z[p0] = make_float2(y0.x + y1.x, y0.y + y1.y);
            z[p1] = fft_twid(make_float2(y0.x - y1.x, y0.y - y1.y), j * stepB, q, M, tw);
            z[p2] = make_float2(y2.x + y3.x, y2.y + y3.y);
            z[p3] = fft_twid(make_float2(y2.x - y3.x, y2.y - y3.y), j * stepB, q, M, tw);
        }
        __syncthreads();
        h >>= 2;
    }
    if (h == 1) {
        int half = M >> 1;
        for (int g = tid; g < 2 * half; g += WG) {
            float2 *z = (g < half) ? za : zb;
            int p = 2 * ((g < half) ? g : g - half);
            float2 a = z[p], b = z[p + 1];
            z[p] = make_float2(a.x + b.x, a.y + b.y);
            z[p + 1] = make_float2(a.x - b.x, a.y - b.y);                   // (twiddle 1: spec v2 does not multiply)
        }
        __syncthreads();
    }
}


// single-array variant (decoder)
__device__ void fft1_dif(float2 *z, int M, const float2 *__restrict__ tw, int tid) {
    int h = M >> 1;
    int quarter = M >> 2;
    while (h >= 2) {
        int q = h >> 1;
        int stepA = M / (2 * h), stepB = stepA * 2;
        for (int gg = tid; gg < quarter; gg += WG) {
            int j = gg & (q - 1);
            int p0 = (gg - j) * 4 + j;
            int p1 = p0 + q, p2 = p0 + h, p3 = p2 + q;
            float2 x0 = z[p0], x1 = z[p1], x2 = z[p2], x3 = z[p3];
            float2 y0 = make_float2(x0.x + x2.x, x0.y + x2.y);
            float2 y2 = fft_twid(make_float2(x0.x - x2.x, x0.y - x2.y), j * stepA, h, M, tw);
            float2 y1 = make_float2(x1.x + x3.x, x1.y + x3.y);
            float2 y3 = fft_twid(make_float2(x1.x - x3.x, x1.y - x3.y), (j + q) * stepA, h, M, tw);
            z[p0] = make_float2(y0.x + y1.x, y0.y + y1.y);
            z[p1] = fft_twid(make_float2(y0.x - y1.x, y0.y - y1.y), j * stepB, q, M, tw);
            z[p2] = make_float2(y2.x + y3.x, y2.y + y3.y);
            z[p3] = fft_twid(make_float2(y2.x - y3.x, y2.y - y3.y), j * stepB, q, M, tw);
        }
        __syncthreads();
        h >>= 2;
    }
    if (h == 1) {
        int half = M >> 1;
        for (int g = tid; g < half; g += WG) {
            int p = 2 * g;
            float2 a = z[p], b = z[p + 1];
            z[p] = make_float2(a.x + b.x, a.y + b.y);
            z[p + 1] = make_float2(a.x - b.x, a.y - b.y);                   // (twiddle 1: spec v2 does not multiply)
        }
        __syncthreads();
    }
}

// nArr arrays of M points stored back to back (array a at z + a*M)
__device__ void fftn_dif(float2 *z0, int nArr, int M, const float2 *__restrict__ tw, int tid) {
    int h = M >> 1;
    int quarter = M >> 2;
    int qshift = 31 - __clz(quarter);
    while (h >= 2) {
        int q = h >> 1;
        int stepA = M / (2 * h), stepB = stepA * 2;
        for (int g = tid; g < nArr * quarter; g += WG) {
            int a = g >> qshift;
            int gg = g & (quarter - 1);
            float2 *z = z0 + a * M;
            int j = gg & (q - 1);
            int p0 = (gg - j) * 4 + j;
            int p1 = p0 + q, p2 = p0 + h, p3 = p2 + q;
            float2 x0 = z[p0], x1 = z[p1], x2 = z[p2], x3 = z[p3];
            float2 y0 = make_float2(x0.x + x2.x, x0.y + x2.y);
            float2 y2 = fft_twid(make_float2(x0.x - x2.x, x0.y - x2.y), j * stepA, h, M, tw);
            float2 y1 = make_float2(x1.x + x3.x, x1.y + x3.y);
            float2 y3 = fft_twid(make_float2(x1.x - x3.x, x1.y - x3.y), (j + q) * stepA, h, M, tw);
            z[p0] = make_float2(y0.x + y1.x, y0.y + y1.y);
            z[p1] = fft_twid(make_float2(y0.x - y1.x, y0.y - y1.y), j * stepB, q, M, tw);
            z[p2] = make_float2(y2.x + y3.x, y2.y + y3.y);
            z[p3] = fft_twid(make_float2(y2.x - y3.x, y2.y - y3.y), j * stepB, q, M, tw);
        }
        __syncthreads();
        h >>= 2;
    }
    if (h == 1) {
        int half = M >> 1;
        int hshift = 31 - __clz(half);
        for (int g = tid; g < nArr * half; g += WG) {
            int a = g >> hshift;
            int p = 2 * (g & (half - 1));
            float2 *z = z0 + a * M;
            float2 x = z[p], y = z[p + 1];
            z[p] = make_float2(x.x + y.x, x.y + y.y);
            z[p + 1] = make_float2(x.x - y.x, x.y - y.y);                   // (twiddle 1: spec v2 does not multiply)
        }
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------
// Wave-per-array form of the same transform.  One wavefront owns one M-point array and
// carries 2^R points per lane through R consecutive radix-2 DIF stages in registers
// (R <= 4: 16 points, <= 3 LDS round trips for M <= 4096), so no workgroup barrier is
// needed between passes: LDS operations of one wave execute in order.  Stage by stage
// the butterflies, twiddle values and rounding are exactly those of fftn_dif.
// Arrays are stored padded: one complex of padding after every 2^ps.  ps = 3 makes the 8-byte accesses of all
// three pass shapes of a 1024-point transform (stride 64, stride 4 inside 32-point blocks, adjacent) hit 16
// distinct slots per 16-lane group; ps = 4 leaves the middle pass 2-way conflicted but is what ships - the
// transforms are not LDS-bandwidth-bound and the smaller arrays keep one more workgroup per CU.
// ---------------------------------------------------------------------------
// Between two passes of one wave: pass n+1 reads LDS elements that OTHER lanes of the wave stored in pass n.  The hardware
// runs a wave's LDS operations in order; this fence + wave barrier (no instruction is emitted) keeps the compiler from
// moving a load of pass n+1 above a store of pass n to an address it can prove different for the same lane.
#define FFT_WAVE_SYNC() do { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); } while (0)
#define FFT_PADS(p, ps) ((p) + ((p) >> (ps)))
#define FFT_PADDEDS(M, ps) ((M) + ((M) >> (ps)))

// Packed binary32 arithmetic (both components of a complex number per instruction, each component rounded exactly as
// the scalar instruction rounds it).  A general radix-2 butterfly is 4 instructions: sum, difference, v_pk_mul_f32
// (d * w.xx) and v_pk_fma_f32 ((d.y, d.x) * (w.y, -w.y) + that): the spec's two products and two fused multiply-adds.
typedef float fft_v2f __attribute__((ext_vector_type(2)));
__device__ __forceinline__ fft_v2f fft_cmulc_pk(fft_v2f d, fft_v2f w) {      // d * conj(w), as cmulc
    fft_v2f p, r;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel:[0,0] op_sel_hi:[1,0]" : "=v"(p) : "v"(d), "v"(w));                          // (d.x w.x, d.y w.x)
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_hi:[0,1,0]" : "=v"(r) : "v"(d), "v"(w), "v"(p));   // (d.y w.y + p.x, d.x (-w.y) + p.y)
    return r;
}
__device__ __forceinline__ fft_v2f fft_cmulc_post_pk(fft_v2f d, fft_v2f w) {      // as cmulc_post: (Re, -Im) of d * conj(w)
    fft_v2f p, r;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel:[0,0] op_sel_hi:[1,0]" : "=v"(p) : "v"(d), "v"(w));                          // (d.x w.x, d.y w.x)
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_hi:[0,0,1]" : "=v"(r) : "v"(d), "v"(w), "v"(p));   // (d.y w.y + p.x, d.x w.y - p.y)
    return r;
}
__device__ __forceinline__ fft_v2f fft_rot_mi(fft_v2f d) { const fft_v2f r = { d.y, -d.x }; return r; }      // d * (-i), exact

// LAST: the pass that ends at half-size 1 (a lane's points are adjacent, q = 1, j = 0): which of its butterflies have a
// trivial twiddle is known at compile time.  In any other pass a stage of half-size <= 4 only occurs for 32-point
// transforms (64-sample subblocks): lane selects there.
template <int R, bool LAST, bool SEL = false>       // SEL: a pass that is not the last but reaches half-size <= 4
__device__ __forceinline__ void fft_wave_pass(float2 *z, int M, int h, const float2 *__restrict__ tw, int lane, int ps) {
    constexpr int NP = 1 << R;
    const int q = h >> (R - 1);                    // spacing of one lane's points
    const int stepA = M / (2 * h);
#ifdef FFT_PACKED
    fft_v2f *zv = (fft_v2f *)z; const fft_v2f *twv = (const fft_v2f *)tw;
    for (int gg = lane; gg < (M >> R); gg += 64) {
        int j = gg & (q - 1);
        int p0 = ((gg - j) << R) + j;
        fft_v2f x[NP];
#pragma unroll
        for (int m = 0; m < NP; m++) x[m] = zv[FFT_PADS(p0 + m * q, ps)];
#pragma unroll
        for (int s = 0; s < R; s++) {
            const int half = NP >> (s + 1);
#pragma unroll
            for (int m = 0; m < NP; m++) {
                if (m & half) continue;
                int t = m & (half - 1);
                const fft_v2f a = x[m], b = x[m + half];
                x[m] = a + b;
                const fft_v2f d = a - b;
                if (LAST) {                            // half-size of the stage = half, twiddle index = t * M / (2 half)
                    if (half <= 4 && t == 0) x[m + half] = d;
                    else if (half <= 4 && 2 * t == half) x[m + half] = fft_rot_mi(d);
                    else x[m + half] = fft_cmulc_pk(d, twv[t * (stepA << s)]);
                } else {
                    const int idx = (j + t * q) * (stepA << s);
                    fft_v2f r = fft_cmulc_pk(d, twv[idx]);
                    if (SEL) { if ((h >> s) <= 4) r = idx == 0 ? d : (idx * 4 == M ? fft_rot_mi(d) : r); }
                    x[m + half] = r;
                }
            }
        }
#pragma unroll
        for (int m = 0; m < NP; m++) zv[FFT_PADS(p0 + m * q, ps)] = x[m];
    }
#else
    for (int gg = lane; gg < (M >> R); gg += 64) {
        int j = gg & (q - 1);
        int p0 = ((gg - j) << R) + j;
        float2 x[NP];
#pragma unroll
        for (int m = 0; m < NP; m++) x[m] = z[FFT_PADS(p0 + m * q, ps)];
#pragma unroll
        for (int s = 0; s < R; s++) {
            const int half = NP >> (s + 1);
#pragma unroll
            for (int m = 0; m < NP; m++) {
                if (m & half) continue;
                int t = m & (half - 1);
                float2 a = x[m], b = x[m + half];
                x[m] = make_float2(a.x + b.x, a.y + b.y);
                x[m + half] = fft_twid(make_float2(a.x - b.x, a.y - b.y), (j + t * q) * (stepA << s), h >> s, M, tw);
            }
        }
#pragma unroll
        for (int m = 0; m < NP; m++) z[FFT_PADS(p0 + m * q, ps)] = x[m];
    }
#endif
}

// The same pass with the transform size, the stage and the padding as compile-time constants: the padded indices of a
// lane's points, the twiddle indices and the trip count fold to constants and immediate offsets (a third of the generic
// pass's vector instructions is index arithmetic).  Same butterflies, same twiddles, same order.
template <int R, int M, int H, int PS>
__device__ __forceinline__ void fft_wave_pass_ct(float2 *z, const float2 *__restrict__ tw, int lane) {
    constexpr int NP = 1 << R;
    constexpr int q = H >> (R - 1);
    constexpr int stepA = M / (2 * H);
    fft_v2f *zv = (fft_v2f *)z; const fft_v2f *twv = (const fft_v2f *)tw;
#pragma unroll
    for (int g0 = 0; g0 < (M >> R); g0 += 64) {
        const int gg = g0 + lane;
        if ((M >> R) < 64 && gg >= (M >> R)) break;
        const int j = gg & (q - 1);
        const int p0 = ((gg - j) << R) + j;
        fft_v2f x[NP];
#pragma unroll
        for (int m = 0; m < NP; m++) x[m] = zv[FFT_PADS(p0 + m * q, PS)];
#pragma unroll
        for (int s = 0; s < R; s++) {
            const int half = NP >> (s + 1);
#pragma unroll
            for (int m = 0; m < NP; m++) {
                if (m & half) continue;
                const int t = m & (half - 1);
                const fft_v2f a = x[m], b = x[m + half];
                x[m] = a + b;
                const fft_v2f d = a - b;
                constexpr int hs = H >> 0;                         // (half-size of the pass's first stage; stage s: hs >> s)
                if (q == 1 && half <= 4 && t == 0) x[m + half] = d;                 // last pass: j = 0, half-size = half
                else if (q == 1 && half <= 4 && 2 * t == half) x[m + half] = fft_rot_mi(d);
                else {
                    const int idx = (j + t * q) * (stepA << s);
                    fft_v2f r = fft_cmulc_pk(d, twv[idx]);
                    if (q != 1 && (hs >> s) <= 4) r = idx == 0 ? d : (idx * 4 == M ? fft_rot_mi(d) : r);
                    x[m + half] = r;
                }
            }
        }
#pragma unroll
        for (int m = 0; m < NP; m++) zv[FFT_PADS(p0 + m * q, PS)] = x[m];
    }
}
template <int M, int PS, int REM, int H> struct FftWaveCt {
    static __device__ __forceinline__ void run(float2 *z, const float2 *__restrict__ tw, int lane) {
        constexpr int passes = (REM + 3) >> 2;
        constexpr int r = (REM + passes - 1) / passes;
        fft_wave_pass_ct<r, M, H, PS>(z, tw, lane);
        FFT_WAVE_SYNC();
        FftWaveCt<M, PS, REM - r, (H >> r)>::run(z, tw, lane);
    }
};
template <int M, int PS, int H> struct FftWaveCt<M, PS, 0, H> { static __device__ __forceinline__ void run(float2 *, const float2 *, int) {} };
template <int M, int PS> __device__ __forceinline__ void fft_wave_dif_ct(float2 *z, const float2 *__restrict__ tw, int lane) {
    constexpr int lg = M == 4096 ? 12 : M == 2048 ? 11 : M == 1024 ? 10 : M == 512 ? 9 : M == 256 ? 8 : M == 128 ? 7 : 6;
    static_assert((1 << lg) == M, "power of two between 64 and 4096");
    FftWaveCt<M, PS, lg, (M >> 1)>::run(z, tw, lane);
}

// whole M-point transform of one padded array by one wave (M = 16 .. 4096)
__device__ __forceinline__ void fft_wave_dif(float2 *z, int M, const float2 *__restrict__ tw, int lane, int ps) {
    int rem = 31 - __clz(M);
    int h = M >> 1;
    while (rem > 0) {
        int passes = (rem + 3) >> 2;
        int r = (rem + passes - 1) / passes;       // 10 -> 4,3,3   9 -> 3,3,3   8 -> 4,4   5 -> 3,2
        if (rem == r) {                            // the last pass
            switch (r) {
                case 4: fft_wave_pass<4, true>(z, M, h, tw, lane, ps); break;
                case 3: fft_wave_pass<3, true>(z, M, h, tw, lane, ps); break;
                case 2: fft_wave_pass<2, true>(z, M, h, tw, lane, ps); break;
                default: fft_wave_pass<1, true>(z, M, h, tw, lane, ps); break;
            }
        } else if ((h >> (r - 1)) <= 4) {          // (32-point transforms only: 3 + 2 stages)
            switch (r) {
                case 4: fft_wave_pass<4, false, true>(z, M, h, tw, lane, ps); break;
                case 3: fft_wave_pass<3, false, true>(z, M, h, tw, lane, ps); break;
                case 2: fft_wave_pass<2, false, true>(z, M, h, tw, lane, ps); break;
                default: fft_wave_pass<1, false, true>(z, M, h, tw, lane, ps); break;
            }
        } else {
            switch (r) {
                case 4: fft_wave_pass<4, false>(z, M, h, tw, lane, ps); break;
                case 3: fft_wave_pass<3, false>(z, M, h, tw, lane, ps); break;
                case 2: fft_wave_pass<2, false>(z, M, h, tw, lane, ps); break;
                default: fft_wave_pass<1, false>(z, M, h, tw, lane, ps); break;
            }
        }
        h >>= r; rem -= r;
        FFT_WAVE_SYNC();
    }
}
