// ulcx_fft.h — LDS-resident complex FFT used by the forward and inverse transforms.
// "fourier spec v1" (oracle/orc_fourier.c): radix-2 decimation in
// frequency, in place, natural-order input, bit-reversed output, twiddle table
// W[j] = (cos, sin)(2 pi j / M); every complex multiply is 4 binary32 products and 2
// sums, each individually rounded (file compiled with -ffp-contract=off).
// Requires WG (workgroup size) to be defined by the includer.
#pragma once
#include <hip/hip_runtime.h>

__device__ __forceinline__ float2 cmulc(float2 d, float2 w) {    // d * conj(w), 4 mul + 2 add, unfused
    float m0 = d.x * w.x, m1 = d.y * w.y, m2 = d.y * w.x, m3 = d.x * w.y;
    return make_float2(m0 + m1, m2 - m3);
}


// In-place radix-2 DIF FFT of two M-point arrays held in LDS ("fourier spec v1",
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
            float2 wA0 = tw[j * stepA], wA1 = tw[(j + q) * stepA], wB = tw[j * stepB];
            float2 y0 = make_float2(x0.x + x2.x, x0.y + x2.y);
            float2 y2 = cmulc(make_float2(x0.x - x2.x, x0.y - x2.y), wA0);
            float2 y1 = make_float2(x1.x + x3.x, x1.y + x3.y);
            float2 y3 = cmulc(make_float2(x1.x - x3.x, x1.y - x3.y), wA1);
            z[p0] = make_float2(y0.x + y1.x, y0.y + y1.y);
            z[p1] = cmulc(make_float2(y0.x - y1.x, y0.y - y1.y), wB);
            z[p2] = make_float2(y2.x + y3.x, y2.y + y3.y);
            z[p3] = cmulc(make_float2(y2.x - y3.x, y2.y - y3.y), wB);
        }
        __syncthreads();
        h >>= 2;
    }
    if (h == 1) {
        float2 w0 = tw[0];
        int half = M >> 1;
        for (int g = tid; g < 2 * half; g += WG) {
            float2 *z = (g < half) ? za : zb;
            int p = 2 * ((g < half) ? g : g - half);
            float2 a = z[p], b = z[p + 1];
            z[p] = make_float2(a.x + b.x, a.y + b.y);
            z[p + 1] = cmulc(make_float2(a.x - b.x, a.y - b.y), w0);
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
            float2 wA0 = tw[j * stepA], wA1 = tw[(j + q) * stepA], wB = tw[j * stepB];
            float2 y0 = make_float2(x0.x + x2.x, x0.y + x2.y);
            float2 y2 = cmulc(make_float2(x0.x - x2.x, x0.y - x2.y), wA0);
            float2 y1 = make_float2(x1.x + x3.x, x1.y + x3.y);
            float2 y3 = cmulc(make_float2(x1.x - x3.x, x1.y - x3.y), wA1);
            z[p0] = make_float2(y0.x + y1.x, y0.y + y1.y);
            z[p1] = cmulc(make_float2(y0.x - y1.x, y0.y - y1.y), wB);
            z[p2] = make_float2(y2.x + y3.x, y2.y + y3.y);
            z[p3] = cmulc(make_float2(y2.x - y3.x, y2.y - y3.y), wB);
        }
        __syncthreads();
        h >>= 2;
    }
    if (h == 1) {
        float2 w0 = tw[0];
        int half = M >> 1;
        for (int g = tid; g < half; g += WG) {
            int p = 2 * g;
            float2 a = z[p], b = z[p + 1];
            z[p] = make_float2(a.x + b.x, a.y + b.y);
            z[p + 1] = cmulc(make_float2(a.x - b.x, a.y - b.y), w0);
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
            float2 wA0 = tw[j * stepA], wA1 = tw[(j + q) * stepA], wB = tw[j * stepB];
            float2 y0 = make_float2(x0.x + x2.x, x0.y + x2.y);
            float2 y2 = cmulc(make_float2(x0.x - x2.x, x0.y - x2.y), wA0);
            float2 y1 = make_float2(x1.x + x3.x, x1.y + x3.y);
            float2 y3 = cmulc(make_float2(x1.x - x3.x, x1.y - x3.y), wA1);
            z[p0] = make_float2(y0.x + y1.x, y0.y + y1.y);
            z[p1] = cmulc(make_float2(y0.x - y1.x, y0.y - y1.y), wB);
            z[p2] = make_float2(y2.x + y3.x, y2.y + y3.y);
            z[p3] = cmulc(make_float2(y2.x - y3.x, y2.y - y3.y), wB);
        }
        __syncthreads();
        h >>= 2;
    }
    if (h == 1) {
        float2 w0 = tw[0];
        int half = M >> 1;
        int hshift = 31 - __clz(half);
        for (int g = tid; g < nArr * half; g += WG) {
            int a = g >> hshift;
            int p = 2 * (g & (half - 1));
            float2 *z = z0 + a * M;
            float2 x = z[p], y = z[p + 1];
            z[p] = make_float2(x.x + y.x, x.y + y.y);
            z[p + 1] = cmulc(make_float2(x.x - y.x, x.y - y.y), w0);
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

// Packed binary32 arithmetic (v_pk_add_f32 / v_pk_mul_f32: both components of a complex number per instruction, each
// component rounded exactly as the scalar instruction rounds it - no fusion, so the spec's operation order and results are
// unchanged).  A radix-2 butterfly is 5 instructions instead of 10: sum, difference, two products (d * w.xx, d.yx * w.yy)
// and one add whose high half is negated (m0 + m1, m2 - m3).
typedef float fft_v2f __attribute__((ext_vector_type(2)));
__device__ __forceinline__ fft_v2f fft_cmulc_pk(fft_v2f d, fft_v2f w) {      // d * conj(w), as cmulc
    const fft_v2f p = d * __builtin_shufflevector(w, w, 0, 0);                                 // (d.x w.x, d.y w.x)
    const fft_v2f q = __builtin_shufflevector(d, d, 1, 0) * __builtin_shufflevector(w, w, 1, 1);   // (d.y w.y, d.x w.y)
    fft_v2f r;
    asm("v_pk_add_f32 %0, %1, %2 neg_hi:[0,1]" : "=v"(r) : "v"(p), "v"(q));                     // (p.x + q.x, p.y - q.y)
    return r;
}

template <int R>
__device__ __forceinline__ void fft_wave_pass(float2 *z, int M, int h, const float2 *__restrict__ tw, int lane, int ps) {
    constexpr int NP = 1 << R;
#ifdef FFT_PACKED
    const int q = h >> (R - 1);                    // spacing of one lane's points
    const int stepA = M / (2 * h);
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
                const fft_v2f w = twv[(j + t * q) * (stepA << s)];
                const fft_v2f a = x[m], b = x[m + half];
                x[m] = a + b;
                x[m + half] = fft_cmulc_pk(a - b, w);
            }
        }
#pragma unroll
        for (int m = 0; m < NP; m++) zv[FFT_PADS(p0 + m * q, ps)] = x[m];
    }
#else
    const int q = h >> (R - 1);                    // spacing of one lane's points
    const int stepA = M / (2 * h);
    for (int gg = lane; gg < (M >> R); gg += 64) {
        int j = gg & (q - 1);
        int p0 = ((gg - j) << R) + j;
        float2 x[NP];
#pragma unroll
        for (int m = 0; m < NP; m++) x[m] = z[FFT_PADS(p0 + m * q, ps)];
#pragma unroll
        for (int s = 0; s < R; s++) {
            constexpr int dummy = 0; (void)dummy;
            const int half = NP >> (s + 1);
#pragma unroll
            for (int m = 0; m < NP; m++) {
                if (m & half) continue;
                int t = m & (half - 1);
                float2 w = tw[(j + t * q) * (stepA << s)];
                float2 a = x[m], b = x[m + half];
                x[m] = make_float2(a.x + b.x, a.y + b.y);
                x[m + half] = cmulc(make_float2(a.x - b.x, a.y - b.y), w);
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
                const fft_v2f w = twv[(j + t * q) * (stepA << s)];
                const fft_v2f a = x[m], b = x[m + half];
                x[m] = a + b;
                x[m + half] = fft_cmulc_pk(a - b, w);
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
        switch (r) {
            case 4: fft_wave_pass<4>(z, M, h, tw, lane, ps); break;
            case 3: fft_wave_pass<3>(z, M, h, tw, lane, ps); break;
            case 2: fft_wave_pass<2>(z, M, h, tw, lane, ps); break;
            default: fft_wave_pass<1>(z, M, h, tw, lane, ps); break;
        }
        h >>= r; rem -= r;
        FFT_WAVE_SYNC();
    }
}
