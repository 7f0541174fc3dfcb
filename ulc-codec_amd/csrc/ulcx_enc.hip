// ulcx_enc.hip — batched ulc-codec encoder for gfx950 (MI355X), hand-written HIP.
//
// One call encodes K consecutive blocks of B independent streams.  Pipeline
// (reference call stack SURVEY.md §3A):
//   wc_energy / wc_forward / wc_backward / wc_integrate / wc_decide
//        transient detector -> WindowCtrl per block   (ulcEncoder_WindowControl.c:41-239)
//   xf   one workgroup per block: frames from the input timeline (closed form of the
//        lapping FIFO, BlockTransform.c:175-224), sine window, MDCT+MDST through two
//        DCT-IV = complex FFTs staged entirely in LDS, normalise, keys, per-line
//        energies                                      (BlockTransform.c:229-281)
//   cplx        ordered f32 sums -> BlockComplexity, nOutCoef (BlockTransform.c:279-325, ulcEncoder.c:140-158)
//   bark_uniform/bark_levels/nbark/nline  noise log-spectrum (un-decimated blocks on the geometry-uniform
//               kernel, the rest lane per subblock)    (ulcEncoder_Psyopt.c:168-250)
//   pbark       masking Bark levels                    (ulcEncoder_Psyopt.c:60-155)
//   select      one wave per block: keys (coefficient + masking level, BlockTransform.c:337-345) in registers,
//               the nOutCoef-th largest by bisection; exact heapsort emulation only for tie groups
//               straddling the cut                     (BlockTransform.c:20-77)
//   encode/pack nybble stream                          (ulcEncoder_Encode.c:23-360, ulcEncoder_NoiseFill.c)
// All float arithmetic is written in the reference's operation order and this file is
// compiled with -ffp-contract=off: no fused multiply-add is formed anywhere except the
// explicit ones inside the glibc restatements (ulcx_libm.h).
#include <utility>
#include "ulcx_internal.h"
#include <type_traits>
#include "ulcx_libm.h"

#define WG 256
#define FFT_PACKED                // packed binary32 butterflies (ulcx_fft.h): bit-identical, half the instructions
#include "ulcx_fft.h"

// ---------------------------------------------------------------------------
// small helpers
// ---------------------------------------------------------------------------
__device__ __forceinline__ float fastlog(float x) {              // ulcHelper.h:127-136
    uint32_t b = __float_as_uint(x);
    int e = (int)(b >> 23) - 127;
    float m = __uint_as_float((127u << 23) | (b & 0x7FFFFFu));
    return -1.7417939f + (2.8212026f + (-1.4699568f + (0.44717955f - 0.056570851f * m) * m) * m) * m + 0.6931471806f * e;
}
__device__ __forceinline__ int quant_u(float v) {                // ulcHelper.h:51-72
    return (v >= 0.5f) ? (int)(0.5f + sqrtf(v - 0.25f)) : 0;
}
__device__ __forceinline__ int quant_coef_u(float v, int lim) { int q = quant_u(v); return q < lim ? q : lim; }
__device__ __forceinline__ int quant_coef(float v, int lim) { int q = quant_coef_u(fabsf(v), lim); return v < 0.0f ? -q : q; }

// Input samples.  IN = float: the C API's layout; IN = int16_t: PCM16 ingest, converted on load
// exactly as the reference's WAV reader does (tools/WavIO_Helper.c:49-55: (float)x * 2^-15, exact).  The two blocks kept
// from previous calls (c.hist) are always float.
template <typename IN> __device__ __forceinline__ const IN *pcm_base(const UlcxEncCtx &c);
template <> __device__ __forceinline__ const float *pcm_base<float>(const UlcxEncCtx &c) { return c.pcm; }
template <> __device__ __forceinline__ const int16_t *pcm_base<int16_t>(const UlcxEncCtx &c) { return c.pcm16; }
__device__ __forceinline__ float  ld1(const float *p) { return *p; }
__device__ __forceinline__ float2 ld2(const float *p) { return *(const float2 *)p; }
__device__ __forceinline__ float4 ld4(const float *p) { return *(const float4 *)p; }
__device__ __forceinline__ float  ld1(const int16_t *p) { return (float)*p * 0x1.0p-15f; }
__device__ __forceinline__ float2 ld2(const int16_t *p) { short2 v = *(const short2 *)p; return make_float2((float)v.x * 0x1.0p-15f, (float)v.y * 0x1.0p-15f); }
__device__ __forceinline__ float4 ld4(const int16_t *p) {
    short4 v = *(const short4 *)p;
    return make_float4((float)v.x * 0x1.0p-15f, (float)v.y * 0x1.0p-15f, (float)v.z * 0x1.0p-15f, (float)v.w * 0x1.0p-15f);
}
// the C interleaved samples at time trel (relative to this call's first sample; negative = the two blocks kept from
// previous calls): n = 1, 2 or 4 consecutive floats starting at element e of that time step
template <typename IN> __device__ __forceinline__ float smp_ld1(const UlcxEncCtx &c, int s, int trel, int e) {
    if (trel < 0) return ld1(c.hist + ((size_t)s * 2 * c.BS + (trel + 2 * c.BS)) * c.C + e);
    return ld1(pcm_base<IN>(c) + ((size_t)s * c.K * c.BS + trel) * c.C + e);
}
template <typename IN> __device__ __forceinline__ float2 smp_ld2(const UlcxEncCtx &c, int s, int trel) {
    if (trel < 0) return ld2(c.hist + ((size_t)s * 2 * c.BS + (trel + 2 * c.BS)) * c.C);
    return ld2(pcm_base<IN>(c) + ((size_t)s * c.K * c.BS + trel) * c.C);
}
template <typename IN> __device__ __forceinline__ float4 smp_ld4(const UlcxEncCtx &c, int s, int trel) {   // C == 2: two time steps
    if (trel < 0) return ld4(c.hist + ((size_t)s * 2 * c.BS + (trel + 2 * c.BS)) * c.C);
    return ld4(pcm_base<IN>(c) + ((size_t)s * c.K * c.BS + trel) * c.C);
}
// sample after the encoder's M/S step (BlockTransform.c:102-110)
template <typename IN> __device__ __forceinline__ float ms_sample(const UlcxEncCtx &c, int s, int trel, int ch) {
    if (ch & 1) { float a = smp_ld1<IN>(c, s, trel, ch - 1), b = smp_ld1<IN>(c, s, trel, ch); return (a - b) * 0.5f; }
    if (ch + 1 < c.C) { float a = smp_ld1<IN>(c, s, trel, ch), b = smp_ld1<IN>(c, s, trel, ch + 1); return (a + b) * 0.5f; }
    return smp_ld1<IN>(c, s, trel, ch);
}
// Arrays that one kernel streams out and a later kernel streams in once (envelope scratch, transform outputs, noise
// pairs, masking levels): their loads/stores carry the non-temporal hint so they do not evict what is re-read.
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float4 ldnt(const float4 *p) { f32x4 v = __builtin_nontemporal_load((const f32x4 *)p); return make_float4(v.x, v.y, v.z, v.w); }
__device__ __forceinline__ void stnt(float4 *p, float4 v) { f32x4 w = { v.x, v.y, v.z, v.w }; __builtin_nontemporal_store(w, (f32x4 *)p); }
__device__ __forceinline__ void stnt(float2 *p, float2 v) { f32x2 w = { v.x, v.y }; __builtin_nontemporal_store(w, (f32x2 *)p); }
__device__ __forceinline__ void stnt(float *p, float v) { __builtin_nontemporal_store(v, p); }
__device__ __forceinline__ float ldnt(const float *p) { return __builtin_nontemporal_load(p); }
// Line energies (nsum: rows = (block, channel); amp2: rows = block; a row = the BS/2 lines of the pseudo-DFT) are stored in
// TILES of 64 rows x 32 lines, 8 KB contiguous each (round 4): the Bark kernel that takes 64 rows at a time (k_bark_uniform)
// reads whole tiles - rows side by side 4 KB apart gave it 128-byte pieces, 2.5 TB/s.  Index of line l of row r:
__device__ __forceinline__ size_t tile_idx(int half, int row, int l) {
    return (((size_t)(row >> 6) * (half >> 5) + (l >> 5)) * 64 + (row & 63)) * 32 + (l & 31);
}
// ... and the offset of line l relative to a line at a multiple of 32 of the same row (lane-per-unit kernels walk a row)
__device__ __forceinline__ int tile_off(int l) { return ((l >> 5) << 11) + (l & 31); }
// the window-control scratch (envelope planes): hinted like the rest unless built with -DWC_NO_NT (experiment: does the
// hand-over between the chain kernels stay in the Infinity Cache when the steps are small?)
#ifdef WC_NO_NT
__device__ __forceinline__ float4 wc_ld(const float4 *p) { return *p; }
__device__ __forceinline__ void wc_st(float4 *p, float4 v) { *p = v; }
#else
__device__ __forceinline__ float4 wc_ld(const float4 *p) { return ldnt(p); }
__device__ __forceinline__ void wc_st(float4 *p, float4 v) { stnt(p, v); }
#endif
// ---------------------------------------------------------------------------
// Window control
// ---------------------------------------------------------------------------
// Scratch layout for the transient detector, in floats: per (group of 64 streams sg, time quad q) two planes of 64x4,
//   env[((sg*T/4 + q)*2 + f)*256 + sl*4 + j],  stream s = sg*64+sl, time t = 4q+j, f = 0 (HP) | 1 (BP), T = maxK*BS:
// time-major inside groups of 64 streams, FOUR consecutive steps of one filter adjacent.  The kernels that walk time
// with one lane per stream[, filter] take four steps per 16-byte load/store (they are bound by instructions per step),
// a wave touches whole contiguous 1 KB planes, and the kernels that need one filter only (k_wc_backward's output,
// k_wc_integrate's input: the HP plane) move no bytes of the other.
__device__ __forceinline__ size_t envq_idx(const UlcxEncCtx &c, int s, int q) {      // HP quad of stream s; the BP quad is 256 floats on
    return ((size_t)(s >> 6) * (c.maxK * c.BS / 4) + q) * 512 + (size_t)(s & 63) * 4;
}

// WindowControl.c:31-70: E[n] = sum_ch (hp^2, bp^2) of the 3-tap FIRs centred on the
// Old/New boundary; then the sqrt of :80-81 (parallel part of the recurrence).
// One workgroup = 64 streams x 64 time steps; input rows are read along time
// (coalesced), transposed through LDS, written stream-minor.
// All window-control kernels (and k_xf) take a block range [k0, k1) of the call so the host can
// pipeline chunks of blocks: the stream-sequential recurrences of later chunks run beside the
// transform of earlier ones.
// {sqrt(E_hp), sqrt(E_bp)} of stream s at centre sample t (relative to the call's first sample)
template <typename IN> __device__ __forceinline__ float2 wc_energy_at(const UlcxEncCtx &c, int s, int t) {
    float ehp = 0.0f, ebp = 0.0f;
    if (c.C == 2) {                                // stereo fast path: three 8-byte (PCM16: 4-byte) loads
        float2 a = smp_ld2<IN>(c, s, t - 1), b = smp_ld2<IN>(c, s, t), d = smp_ld2<IN>(c, s, t + 1);
        float m0 = (a.x + a.y) * 0.5f, m1 = (b.x + b.y) * 0.5f, m2 = (d.x + d.y) * 0.5f;
        float s0 = (a.x - a.y) * 0.5f, s1 = (b.x - b.y) * 0.5f, s2 = (d.x - d.y) * 0.5f;
        float hp = -m0 + 2 * m1 - m2, bp = -m0 + m2;
        ehp += hp * hp; ebp += bp * bp;
        hp = -s0 + 2 * s1 - s2; bp = -s0 + s2;
        ehp += hp * hp; ebp += bp * bp;
    } else {
        for (int ch = 0; ch < c.C; ch++) {
            float t0 = ms_sample<IN>(c, s, t - 1, ch), t1 = ms_sample<IN>(c, s, t, ch), t2 = ms_sample<IN>(c, s, t + 1, ch);
            float hp = -t0 + 2 * t1 - t2;
            float bp = -t0 + t2;
            ehp += hp * hp;
            ebp += bp * bp;
        }
    }
    return make_float2(sqrtf(ehp), sqrtf(ebp));
}
template <typename IN>
__global__ __launch_bounds__(WG) void k_wc_energy(UlcxEncCtx c, int k0, int k1) {
    __shared__ float2 tile[64][65];
    int tiles_t = ((k1 - k0) * c.BS) / 64;
    int sg = blockIdx.x / tiles_t, tt = blockIdx.x % tiles_t + (k0 * c.BS) / 64;
    int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    int r = tt * 64 + lane;                              // k*BS + n
    int t = r - c.BS / 2;                                // centre sample
#pragma unroll 4
    for (int sl = wv; sl < 64; sl += 4) {
        int s = sg * 64 + sl;
        tile[lane][sl] = (s < c.B) ? wc_energy_at<IN>(c, s, t) : make_float2(0.0f, 0.0f);
    }
    __syncthreads();
    float *dst = (float *)c.env + envq_idx(c, sg * 64, tt * 16);      // 16 quads x 2 planes x 64 streams x 4 steps
    for (int ql = wv; ql < 16; ql += 4) {
        const float2 a0 = tile[4 * ql][lane], a1 = tile[4 * ql + 1][lane], a2 = tile[4 * ql + 2][lane], a3 = tile[4 * ql + 3][lane];
        float4 *o = (float4 *)(dst + (size_t)ql * 512 + lane * 4);
        wc_st(o, make_float4(a0.x, a1.x, a2.x, a3.x));
        wc_st(o + 64, make_float4(a0.y, a1.y, a2.y, a3.y));
    }
}

// WindowControl.c:72-88: forward one-pole smear, the only sample-rate recurrence that
// crosses blocks.  One lane per stream (both filters), strictly sequential in time.
__global__ __launch_bounds__(64) void k_wc_forward(UlcxEncCtx c, int k0, int k1) {
    // lane = (stream, filter): 32 streams x {HP, BP} per wave; the two one-pole chains are independent
    __builtin_amdgcn_s_setprio(3);                       // a serial chain: let it issue ahead of co-resident throughput kernels
    int gl = blockIdx.x * 64 + threadIdx.x;
    int s = gl >> 1, f = gl & 1;
    bool live = s < c.B;
    float4 *v = (float4 *)((float *)c.env + envq_idx(c, live ? s : 0, k0 * c.BS / 4) + f * 256);   // this lane's four steps of each quad
    float env = live ? c.wcs[s].tf[f] : 0.0f;
    float cc = f ? c.cBP : c.cHP;
    const int nq = (k1 - k0) * c.BS / 4;
    // Groups of U quads (4 steps each) addressed from one pointer with immediate offsets (quads of one stream are
    // 2 KB apart), loads D-1 groups ahead of the arithmetic: the chain is bound by instructions per step (3 dependent
    // VALU + a quarter of a load and of a store), so address arithmetic and loop control are kept out of it.
    constexpr int U = 2, D = 8;           // (K*BS/4 is a multiple of U*D)
    constexpr int QS = 512 / 4;           // float4s between consecutive quads of a stream
    const float4 *rp = v;
    float4 *wp = v;
    float4 x[D][U];
#pragma unroll
    for (int g = 0; g < D - 1; g++) {
#pragma unroll
        for (int j = 0; j < U; j++) x[g][j] = wc_ld(rp + (size_t)j * QS);
        rp += U * QS;
    }
    for (int i = 0; i < nq; i += D * U) {
#pragma unroll
        for (int g = 0; g < D; g++) {
            const bool more = (i + (g + D - 1) * U) < nq;
            const float4 *lp = more ? rp : v;               // past the end: re-read quad 0 (unused)
#pragma unroll
            for (int j = 0; j < U; j++) x[(g + D - 1) % D][j] = wc_ld(lp + (size_t)j * QS);
            rp += U * QS;
#pragma unroll
            for (int j = 0; j < U; j++) {
                float4 q = x[g][j];
                float d;
                d = q.x - env; env += d * cc; q.x = env;
                d = q.y - env; env += d * cc; q.y = env;
                d = q.z - env; env += d * cc; q.z = env;
                d = q.w - env; env += d * cc; q.w = env;
                x[g][j] = q;
            }
            if (live) {
#pragma unroll
                for (int j = 0; j < U; j++) wc_st(wp + (size_t)j * QS, x[g][j]);
            }
            wp += U * QS;
        }
    }
    if (live) c.wcs[s].tf[f] = env;                                   // state for the next call
}

// k_wc_energy + k_wc_forward in one kernel (stereo): the envelope never goes through HBM on its way into the recurrence.
// One workgroup = EF_SPW streams: wave 0 runs the two one-pole chains of each (lane = stream, filter), every other wave
// produces one stream's energies a few tiles of 64 steps ahead (lane = time step: coalesced input rows) into an LDS
// ring, transposed.  Producers and chain are decoupled by counters in LDS (tiles finished per producer wave, tiles
// taken by the chain), not by barriers: the chain never waits as long as the producers are ahead, and they keep three
// tiles of loads in flight.  Measured alone on the bench batch: 0.52 ms against 0.58 + 0.66 ms for the two kernels -
// fed from LDS the chain has 3.5 instead of 4.1 instructions per step - and 1.07 GB written + 1.07 GB read less;
// 32 streams per workgroup with 3 producer waves: 1.16 ms (the producers are the bottleneck), 32/15: 1.06, 16/15: 0.78,
// 4/4: 0.95.
#define EF_TS 68                                          // floats per (stream, filter) row of a tile: 64 steps + pad (rows stay 16-byte aligned, b128 reads conflict-free)
#ifndef EF_RT
#define EF_RT 4                                           // tiles in the ring
#endif
#define EF_SPW 8                                          // streams per workgroup
#define EF_NW 9                                           // waves per workgroup: the chain + one producer per stream
#define EF_TILE_FLOATS (EF_SPW * 2 * EF_TS)
#define EF_LDS_BYTES (EF_RT * EF_TILE_FLOATS * 4 + 4 * EF_NW)
// (stereo only: the producers split the envelope computation into its three 8-byte loads, issued tiles ahead,
//  and the arithmetic)
__device__ __forceinline__ float2 wc_energy_stereo(float2 a, float2 b, float2 d) {        // as wc_energy_at, C == 2
    float m0 = (a.x + a.y) * 0.5f, m1 = (b.x + b.y) * 0.5f, m2 = (d.x + d.y) * 0.5f;
    float s0 = (a.x - a.y) * 0.5f, s1 = (b.x - b.y) * 0.5f, s2 = (d.x - d.y) * 0.5f;
    float ehp = 0.0f, ebp = 0.0f;
    float hp = -m0 + 2 * m1 - m2, bp = -m0 + m2;
    ehp += hp * hp; ebp += bp * bp;
    hp = -s0 + 2 * s1 - s2; bp = -s0 + s2;
    ehp += hp * hp; ebp += bp * bp;
    return make_float2(sqrtf(ehp), sqrtf(ebp));
}
template <int NW, typename IN>
__global__ __launch_bounds__(NW * 64) void k_wc_ef(UlcxEncCtx c, int k0, int k1) {
    extern __shared__ float efs[];
    float *ring = efs;
    int *flags = (int *)(ring + EF_RT * EF_TILE_FLOATS);  // [0..NW-2] tiles finished by producer wave p, [NW-1] tiles taken by the chain
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int s0 = blockIdx.x * EF_SPW;
    const int nT = (k1 - k0) * c.BS / 64;
    constexpr int NP = NW - 1, NS = (EF_SPW + NP - 1) / NP;   // producer waves, streams per producer wave
    if (threadIdx.x < NW) flags[threadIdx.x] = 0;
    __syncthreads();
    if (wv > 0) {
        const int p = wv - 1;
        if (p >= EF_SPW) {                                // (more producer waves than streams: nothing to produce, but the chain counts every wave)
            if (lane == 0) __atomic_store_n(&flags[p], 0x7ffffff0, __ATOMIC_RELEASE);
            return;
        }
        // THREE tiles of loads in flight per producer wave (a tile period is shorter than the latency of a load when the
        // transform runs beside this kernel): register sets A0/A1/A2 rotate by unrolling the tile loop three times
#ifndef EF_AHEAD
#define EF_AHEAD 3
#endif
        float2 A[EF_AHEAD][NS][3];
        // this lane's sample of step 0 of the call, per stream of the wave (the 64-bit stream offset once, not per tile:
        // three quarter-rate multiplies a tile)
        const IN *sbase[NS];
#pragma unroll
        for (int i = 0; i < NS; i++) {
            const int s = s0 + p + NP * i;
            const int sc = ((p + NP * i < EF_SPW) && (s < c.B)) ? s : 0;
            sbase[i] = pcm_base<IN>(c) + (size_t)sc * c.K * c.BS * 2 + 2 * lane;
        }
        auto issue = [&](float2 (&A)[NS][3], int j) {     // the three samples of this lane's time step of tile j, every stream of this wave
            const int t0 = (k0 * c.BS + j * 64) - c.BS / 2;                                                     // lane 0's centre sample (wave-uniform)
            const int t = t0 + lane;
#pragma unroll
            for (int i = 0; i < NS; i++) {
                const int s = s0 + p + NP * i;
                const bool on = (p + NP * i < EF_SPW) && (s < c.B);
                const int sc = on ? s : 0;                // (a stream that exists: the values are not used)
                if (t0 >= 1) {
                    // the tile and its two neighbours lie in this call's input (all but the first BS/2 + 1 steps of a call): one
                    // wave-uniform base, three loads at constant offsets (round 3: the general form below - a history / input
                    // select and a 64-bit stream offset per load - was 54 of the 113 vector instructions a step costs here)
                    const IN *q = sbase[i] + 2 * (ptrdiff_t)t0;
                    A[i][0] = ld2(q - 2); A[i][1] = ld2(q); A[i][2] = ld2(q + 2);
                } else { A[i][0] = smp_ld2<IN>(c, sc, t - 1); A[i][1] = smp_ld2<IN>(c, sc, t); A[i][2] = smp_ld2<IN>(c, sc, t + 1); }
            }
        };
        auto step = [&](float2 (&A)[NS][3], int j) {
            float2 v[NS];
#pragma unroll
            for (int i = 0; i < NS; i++) v[i] = wc_energy_stereo(A[i][0], A[i][1], A[i][2]);
            if (j + EF_AHEAD < nT) issue(A, j + EF_AHEAD);
            while (j >= __atomic_load_n(&flags[NP], __ATOMIC_ACQUIRE) + EF_RT) __builtin_amdgcn_s_sleep(4);     // ring full
            float *tile = ring + (j % EF_RT) * EF_TILE_FLOATS;
#pragma unroll
            for (int i = 0; i < NS; i++) {
                const int sl = p + NP * i;
                if (sl < EF_SPW) {
                    const bool on = s0 + sl < c.B;
                    tile[(sl * 2 + 0) * EF_TS + lane] = on ? v[i].x : 0.0f;
                    tile[(sl * 2 + 1) * EF_TS + lane] = on ? v[i].y : 0.0f;
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            if (lane == 0) __atomic_store_n(&flags[p], j + 1, __ATOMIC_RELEASE);
        };
#pragma unroll
        for (int a = 0; a < EF_AHEAD; a++) if (a < nT) issue(A[a], a);
        for (int j = 0; j < nT; j += EF_AHEAD) {
#pragma unroll
            for (int a = 0; a < EF_AHEAD; a++) if (j + a < nT) step(A[a], j + a);
        }
        return;
    }
    if (lane >= 2 * EF_SPW) return;                       // the chain: lane = (stream, filter)
    __builtin_amdgcn_s_setprio(3);
    const int s = s0 + (lane >> 1), f = lane & 1;
    const bool live = s < c.B;
    float env = live ? c.wcs[s].tf[f] : 0.0f;
    const float cc = f ? c.cBP : c.cHP;
    constexpr int QS = 512 / 4;                           // float4s between consecutive quads of a stream
    // Where the chain's results go.  A lane without a stream (s >= B, the batch's last group) stores like the others, into
    // its own padded rows of the scratch: the recurrence is three dependent instructions a step (25 cycles: tools/ubench/
    // dep_chain.hip) and every other instruction between them costs 4 more - a predicated store with a 64-bit vector
    // address add was five of them per quad.
    // (four pointers 8 KB apart, each the middle of four quads: every store's offset fits the instruction's immediate)
    char *wq[4];
#pragma unroll
    for (int k = 0; k < 4; k++) wq[k] = (char *)((float *)c.env + envq_idx(c, s, k0 * c.BS / 4) + f * 256) + k * 8192 + 4096;
    for (int j = 0; j < nT; j++) {
        for (;;) {                                        // every producer wave has finished tile j
            int m = (lane < NP) ? __atomic_load_n(&flags[lane], __ATOMIC_ACQUIRE) : 0x7fffffff;
            if (__all(m > j)) break;
            __builtin_amdgcn_s_sleep(1);
        }
        const float4 *row = (const float4 *)(ring + (j % EF_RT) * EF_TILE_FLOATS + lane * EF_TS);
        float4 x[16];
#pragma unroll
        for (int q = 0; q < 16; q++) x[q] = row[q];
        // (LDS operations of a wave complete in order: this store lands behind the 16 reads, so the slot is free for the producers)
        if (lane == 0) __atomic_store_n(&flags[NP], j + 1, __ATOMIC_RELEASE);
#pragma unroll
        for (int q = 0; q < 16; q++) {
            float4 v = x[q];
            float d;
            d = v.x - env; env += d * cc; v.x = env;
            d = v.y - env; env += d * cc; v.y = env;
            d = v.z - env; env += d * cc; v.z = env;
            d = v.w - env; env += d * cc; v.w = env;
            wc_st((float4 *)(wq[q >> 2] + ((q & 3) * 2048 - 4096)), v);
        }
#pragma unroll
        for (int k = 0; k < 4; k++) wq[k] += 16 * QS * sizeof(float4);
    }
    if (live) c.wcs[s].tf[f] = env;                       // state for the next call
}

// WindowControl.c:90-104: backward sweep from each block's forward end state.
__global__ __launch_bounds__(64) void k_wc_backward(UlcxEncCtx c, int k0, int k1) {
    __builtin_amdgcn_s_setprio(3);                       // a serial chain: let it issue ahead of co-resident throughput kernels
    int sl = threadIdx.x;
    int k = k0 + blockIdx.x % (k1 - k0), sg = blockIdx.x / (k1 - k0);
    float *e = (float *)c.env + envq_idx(c, sg * 64 + sl, k * c.BS / 4);      // HP quad of the block's first four steps
    const int nq = c.BS / 4;
    constexpr int QS = 512 / 4;                          // float4s between consecutive quads of a stream
    float pHP = e[(size_t)(nq - 1) * 512 + 3], pBP = e[(size_t)(nq - 1) * 512 + 256 + 3];   // the forward end state = the block's last step
    const float qHP = c.qHP, qBP = c.qBP;
    // Walk the block backwards in groups of U quads; a group's quads are addressed from one pointer with
    // immediate offsets, and the loads run D-1 groups ahead of the arithmetic: the chain is bound by instructions
    // per step, so address arithmetic is kept out of it.  The result overwrites the HP plane.
#ifndef WC_BACK_D
#define WC_BACK_D 4
#endif
    constexpr int U = 2, D = WC_BACK_D;                  // BS/4 is a multiple of U*D
    const float4 *rp = (const float4 *)(e + (size_t)(nq - 1) * 512);   // quad being loaded (group head)
    float4 *wp = (float4 *)(e + (size_t)(nq - 1) * 512);               // quad being computed (group head)
    float4 xh[D][U], xb[D][U];
#pragma unroll
    for (int g = 0; g < D - 1; g++) {
#pragma unroll
        for (int j = 0; j < U; j++) { xh[g][j] = wc_ld(rp - (ptrdiff_t)j * QS); xb[g][j] = wc_ld(rp - (ptrdiff_t)j * QS + 64); }
        rp -= U * QS;
    }
    for (int i = 0; i < nq; i += U * D) {
#pragma unroll
        for (int g = 0; g < D; g++) {
            // prefetch the group D-1 ahead into the slot freed last (past the block start: re-read a quad of the block, unused)
            const bool more = (i + (g + D - 1) * U) < nq;
            const float4 *lp = more ? rp : (const float4 *)(e + (size_t)(U - 1) * 512);
#pragma unroll
            for (int j = 0; j < U; j++) { xh[(g + D - 1) % D][j] = wc_ld(lp - (ptrdiff_t)j * QS); xb[(g + D - 1) % D][j] = wc_ld(lp - (ptrdiff_t)j * QS + 64); }
            rp -= U * QS;
            float4 o[U];
#pragma unroll
            for (int j = 0; j < U; j++) {
                const float4 h = xh[g][j], b4 = xb[g][j];
#define WC_BACK_STEP(H, B, O) { float dHP = (H) - pHP, dBP = (B) - pBP; pHP += dHP * qHP; pBP += dBP * qBP; \
                                float a = dHP * pBP, bb = dBP * pHP; (O) = a * a + bb * bb; }
                WC_BACK_STEP(h.w, b4.w, o[j].w) WC_BACK_STEP(h.z, b4.z, o[j].z) WC_BACK_STEP(h.y, b4.y, o[j].y) WC_BACK_STEP(h.x, b4.x, o[j].x)
#undef WC_BACK_STEP
            }
#pragma unroll
            for (int j = 0; j < U; j++) wc_st(wp - (ptrdiff_t)j * QS, o[j]);
            wp -= U * QS;
        }
    }
}

// WindowControl.c:106-134: 8 bins per block, smoothing state carried across blocks.
__global__ __launch_bounds__(64) void k_wc_integrate(UlcxEncCtx c, int k0, int k1) {
    __builtin_amdgcn_s_setprio(3);
    int s = blockIdx.x * 64 + threadIdx.x;
    bool live = s < c.B;
    int sc = live ? s : 0;
    float env = c.wcs[sc].tf[2];
    float *bins = c.bins + (size_t)sc * (c.maxK + 1) * 16;
    if (live && k0 == 0) for (int i = 0; i < 8; i++) { bins[i] = c.wcs[s].binSum[i]; bins[8 + i] = c.wcs[s].binW[i]; }
    const float4 *v = (const float4 *)((const float *)c.env + envq_idx(c, s, k0 * c.BS / 4));   // the HP plane: k_wc_backward's output
    const int bin = c.BS / 8;             // >= 32, a multiple of 4*U
    const int nq = (k1 - k0) * c.BS / 4;
    // same structure as k_wc_forward: groups of U quads off one pointer, loads D-1 groups ahead
#ifndef WC_INT_D
#define WC_INT_D 8
#endif
    constexpr int U = 2, D = WC_INT_D;
    constexpr int QS = 512 / 4;
    const float4 *rp = v;
    float4 x[D][U];
#pragma unroll
    for (int g = 0; g < D - 1; g++) {
#pragma unroll
        for (int j = 0; j < U; j++) x[g][j] = wc_ld(rp + (size_t)j * QS);
        rp += U * QS;
    }
    const float cBlk = c.cBlk;
    float sum = 0.0f;
    int inBin = 0, gbin = k0 * 8;         // steps accumulated in the current bin; global bin index = k*8 + i
    // A trip = D groups of U quads.  The recurrence is three dependent instructions a step (25 cycles: tools/ubench/
    // dep_chain.hip) and every other instruction the wave issues between them costs its 4 cycles on top: all but the last
    // trip fetch ahead without asking whether there is more, and when a bin is a whole number of trips (BlockSize >= 512)
    // the bin boundary is looked for once per trip, not once per group (38 -> 34 cycles a step).
    auto trip = [&](int i, auto tailT, auto fineT) {
        constexpr bool TAIL = decltype(tailT)::value, FINE = decltype(fineT)::value;
#pragma unroll
        for (int g = 0; g < D; g++) {
            const float4 *lp = rp;
            if (TAIL) { const bool more = (i + (g + D - 1) * U) < nq; lp = more ? rp : v; }
#pragma unroll
            for (int j = 0; j < U; j++) x[(g + D - 1) % D][j] = wc_ld(lp + (size_t)j * QS);
            rp += U * QS;
#pragma unroll
            for (int j = 0; j < U; j++) {
                const float4 q = x[g][j];
                float d;
                d = q.x - env; env += d * cBlk; sum += env;
                d = q.y - env; env += d * cBlk; sum += env;
                d = q.z - env; env += d * cBlk; sum += env;
                d = q.w - env; env += d * cBlk; sum += env;
            }
            if (FINE || g == D - 1) {
                inBin += FINE ? 4 * U : 4 * U * D;
                if (inBin == bin) {       // bin boundary (bins never straddle a group); the weight is the step count
                    if (live) { float *o = bins + (size_t)(gbin / 8 + 1) * 16; o[gbin & 7] = sum; o[8 + (gbin & 7)] = (float)bin; }
                    sum = 0.0f; inBin = 0; gbin++;
                }
            }
        }
    };
    const bool coarse = (bin % (4 * U * D)) == 0;
    int i = 0;
    if (coarse) { for (; i + 2 * D * U <= nq; i += D * U) trip(i, std::false_type{}, std::false_type{}); for (; i < nq; i += D * U) trip(i, std::true_type{}, std::false_type{}); }
    else { for (; i + 2 * D * U <= nq; i += D * U) trip(i, std::false_type{}, std::true_type{}); for (; i < nq; i += D * U) trip(i, std::true_type{}, std::true_type{}); }
    if (live) c.wcs[s].tf[2] = env;       // only this kernel reads tf[2]
}

// WindowControl.c:156-238: decision from the bins of block k (R) and k-1 (L).
__global__ __launch_bounds__(64) void k_wc_decide(UlcxEncCtx c, int k0, int k1) {
    int gid = blockIdx.x * 64 + threadIdx.x;
    if (gid >= c.B * (k1 - k0)) return;
    int s = gid / (k1 - k0), k = k0 + gid % (k1 - k0);
    const float *L = c.bins + ((size_t)s * (c.maxK + 1) + k) * 16;
    const float *R = L + 16;
    int log2sub = c.lgBS - 3;
    int decimation = 1;
    float ratio = 0.0f;
    int nSeg = 8, segSize = 1;
    if (log2sub < 6) { int sh = 6 - log2sub; nSeg >>= sh; segSize <<= sh; log2sub = 6; }
    for (;;) {
        log2sub++;
        int maxSeg = 0;
        float maxRatio = -1000.0f;
        for (int seg = 0; seg < nSeg; seg++) {
            float Ls = 0.0f, Lw = 0.0f, Rs = 0.0f, Rw = 0.0f;
            for (int n = 0; n < segSize; n++) {
                // Src[n - SegmentSize] walks back from R's segment start into L (WindowControl.c:187-191)
                int ri = seg * segSize + n;
                int li = ri - segSize;
                float lS = (li >= 0) ? R[li] : L[8 + li];
                float lW = (li >= 0) ? R[8 + li] : L[16 + li];
                Ls += lS; Lw += lW;
                Rs += R[ri]; Rw += R[8 + ri];
            }
            Ls = (Ls != 0.0f) ? ulcx_logf(Ls / Lw) : -100.0f;
            Rs = (Rs != 0.0f) ? ulcx_logf(Rs / Rw) : -100.0f;
            float r = fabsf(Rs - Ls);
            if (r > maxRatio) { maxSeg = seg; maxRatio = r; }
        }
        if (maxRatio - ratio < 0x1.62E430p-1f) break;
        decimation = nSeg + maxSeg;
        ratio = maxRatio;
        if (nSeg > 1 && ratio < 0x1.62E430p-1f) { nSeg /= 2; segSize *= 2; }
        else break;
    }
    int wc;
    if (ratio < 0x1.62E430p-2f) wc = 0x10;
    else {
        ratio *= 0x1.715476p0f;
        int scale = (ratio < 0.5f) ? 0 : (ratio >= 6.5f) ? 7 : (int)rintf(ratio);   // lrintf: round-to-nearest-even
        if (log2sub - scale < 6) scale = log2sub - 6;
        wc = scale + 0x8 * (decimation != 1) + 0x10 * decimation;
    }
    int *row = c.wcArr + (size_t)s * (c.maxK + 2);
    if (k == 0) { row[0] = c.wcs[s].wcPrev; row[1] = c.wcs[s].wcCur; }
    row[k + 2] = wc;
}

// ---------------------------------------------------------------------------
// Transform + per-coefficient analysis
// ---------------------------------------------------------------------------
__device__ __forceinline__ int first_overlap(int wc, int BS) {   // BlockTransform.c:124-128
    unsigned p = ulcx_pattern(wc);
    int ov = BS >> (p & 7);
    if (p & 8) ov >>= (wc & 7);
    return ov;
}

// window value of frame sample i (0 <= i < 2S) of a subblock with left overlap ovL (ramp
// centred on the span start) and right overlap ov: closed form of the lapping FIFO
// (BlockTransform.c:175-224) + sine window of the transform (oracle/orc_fourier.c)
__device__ __forceinline__ float win_apply(float x, int i, int S, int aL, int ovL, int aR, int ov,
                                           const float *__restrict__ rise, const float *__restrict__ fall) {
    if (i < S) return (i < aL) ? 0.0f : (i < aL + ovL) ? x * rise[i - aL] : x;
    int n = i - S;
    return (n < aR) ? x : (n < aR + ov) ? x * fall[n - aR] : 0.0f;
}

// The steady state of the headline geometry, every size a compile-time constant: stereo, BlockSize 2048, an un-decimated
// block between two full-overlap neighbours (PCM16 ingest: from the call's third block on).  Same arithmetic, same
// order as the general body of k_xf below (which documents it) - only the index arithmetic, the loop bounds and the window
// selects fold away, and the four transforms run the compile-time passes (fft_wave_dif_ct).
// (BSC: 2048, the headline geometry; 4096 since round 4 - the window-switching configuration's un-decimated blocks)
template <typename IN, int BSC>
__device__ __forceinline__ int xf_fast(const UlcxEncCtx &c, float *lds, int s, int k, int blk, int tid) {
    constexpr int BS = BSC, S = BSC, M = BSC / 2, PS = 4, Mp = FFT_PADDEDS(M, PS);
    constexpr int LGM = BSC == 4096 ? 11 : 10;
    static_assert(BSC == 2048 || BSC == 4096, "sizes with a compile-time transform");
    static_assert(WG == 256 && (M / 2) % (2 * WG) == 0, "whole fold / epilogue trips per thread");
    float2 *z = (float2 *)lds;
    float2 *twl = (float2 *)(lds + 4 * FFT_PADDEDS(BS, PS));
    float2 *zc0 = z, *zs0 = z + Mp, *zc1 = z + 2 * Mp, *zs1 = z + 3 * Mp;
    const float2 *pre = c.T.pre[0];
    const float *rise = c.T.winRise + S, *fall = c.T.winFall + S;
    // frame = [(k-2) BS, k BS): its first half (positions < S) is block k-2, its second half block k-1 of the stream's
    // timeline; blocks -2 and -1 are the two the encoder keeps from the previous call (c.hist, always float)
    const IN *pcmS = pcm_base<IN>(c) + (size_t)s * c.K * BS * 2;
    const IN *frameLo = pcmS + (ptrdiff_t)(k - 2) * BS * 2, *frameHi = frameLo;       // (indexed with the frame position)
    if constexpr (std::is_same<IN, float>::value) {
        const float *histS = c.hist + (size_t)s * 2 * BS * 2;
        if (k < 2) frameLo = histS + (size_t)k * BS * 2;                      // block k-2 = history block k
        if (k < 1) frameHi = histS;                                            // block k-1 = history block 1: (hist + BS*2) - S*2
    }
#pragma unroll
    for (int i = tid; i < M / 2; i += WG) twl[i] = c.T.tw[0][i];
#pragma unroll
    for (int jj0 = 0; jj0 < M / 2; jj0 += WG) {
        const int jj = jj0 + tid;
        const int iA = 2 * jj, iB = S - 2 - 2 * jj, iC = S + 2 * jj, iD = 2 * S - 2 - 2 * jj;
        const int ip[4] = { iA, iB, iC, iD };
        float2 xs[8];
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const float4 v = ld4((r < 2 ? frameLo : frameHi) + (size_t)ip[r] * 2);
            const float2 m0 = make_float2((v.x + v.y) * 0.5f, (v.x - v.y) * 0.5f);     // the two positions ip[r], ip[r] + 1 after M/S
            const float2 m1 = make_float2((v.z + v.w) * 0.5f, (v.z - v.w) * 0.5f);
            const float2 fw = (r < 2) ? *(const float2 *)(rise + ip[r]) : *(const float2 *)(fall + ip[r] - S);     // (even positions: 8-byte aligned)
            xs[2 * r]     = make_float2(m0.x * fw.x, m0.y * fw.x);
            xs[2 * r + 1] = make_float2(m1.x * fw.y, m1.y * fw.y);
        }
#pragma unroll
        for (int hsel = 0; hsel < 2; hsel++) {
            const int n = hsel ? M / 2 + jj : M / 2 - 1 - jj;
            const float2 lbv = hsel ? xs[0] : xs[1], lav = hsel ? xs[3] : xs[2];
            const float2 rav = hsel ? xs[4] : xs[5], rbv = hsel ? xs[7] : xs[6];
            const float2 P = pre[n];
#pragma unroll
            for (int q = 0; q < 2; q++) {
                const float ra = q ? rav.y : rav.x, rb = q ? rbv.y : rbv.x, la = q ? lav.y : lav.x, lb = q ? lbv.y : lbv.x;
                const float vr = ra + rb, wr = ra - rb;
                const float vl = la - lb, wl = lb + la;
                const float v1 = hsel ? vl : vr, v2 = hsel ? vr : vl;
                const float w1 = hsel ? wl : wr, w2 = hsel ? wr : wl;
                float2 *zc = q ? zc1 : zc0, *zs = q ? zs1 : zs0;
                zc[FFT_PADS(n, PS)] = cmulc(make_float2(v1, v2), P);
                zs[FFT_PADS(n, PS)] = cmulc(make_float2(w2, w1), P);
            }
        }
    }
    __syncthreads();
    fft_wave_dif_ct<M, PS>(z + __builtin_amdgcn_readfirstlane(tid >> 6) * Mp, twl, tid & 63);
    __syncthreads();
    float *coefO = c.coef + (size_t)blk * (2 * BS);
    constexpr float norm = 2.0f / S;
    int nnz = 0;
    // A thread takes TWO neighbouring post-twiddle indices (kk = 2 tid, 2 tid + 1: M/2 = 2 WG of them), so that what it
    // writes is contiguous: coefficients 4 tid .. 4 tid + 3 and BS - 4 - 4 tid .. BS - 1 - 4 tid of each channel as 16-byte
    // stores, line energies as 8-byte stores (one index per thread gave 8- and 4-byte stores: twice the store instructions).
#pragma unroll
    for (int e0 = 0; e0 < M / 2; e0 += 2 * WG) {
        const int kA = e0 + 2 * tid, kB = e0 + 2 * tid + 1;       // k1 of the two; their mirrors k2 = M-1-kA, M-1-kB = (M-1-kA) - 1
        const int kk2[2] = { kA, kB };
        float re[2][2][4];                                        // [channel][0: the k1 side, 1: the k2 side][4 consecutive coefficients]
        float ns[2][2][2];                                        // [channel][side][2 consecutive lines]
        float am[2][2] = { { 0.0f, 0.0f }, { 0.0f, 0.0f } };      // [side][line]
#pragma unroll
        for (int u = 0; u < 2; u++) {
            const int k1 = kk2[u], k2 = M - 1 - k1;
            const int r1 = (int)(__brev((unsigned)k1) >> (32 - LGM)), r2 = (int)(__brev((unsigned)k2) >> (32 - LGM));
            const float2 P1 = pre[k1], P2 = pre[k2];
            const fft_v2f Pv1 = { P1.x, P1.y }, Pv2 = { P2.x, P2.y };
#pragma unroll
            for (int q = 0; q < 2; q++) {
                const float2 *zc = q ? zc1 : zc0, *zs = q ? zs1 : zs0;
                auto cm = [](float2 d, fft_v2f w) { const fft_v2f dv = { d.x, d.y }; const fft_v2f r = fft_cmulc_post_pk(dv, w); return make_float2(r.x, r.y); };
                const float2 yc1 = cm(zc[FFT_PADS(r1, PS)], Pv1), yc2 = cm(zc[FFT_PADS(r2, PS)], Pv2);
                const float2 ys1 = cm(zs[FFT_PADS(r1, PS)], Pv1), ys2 = cm(zs[FFT_PADS(r2, PS)], Pv2);
                const float mdct[4] = { yc1.x, yc2.y, yc2.x, yc1.y };
                const float mdst[4] = { ys1.x, ys2.y, ys2.x, ys1.y };
#pragma unroll
                for (int p = 0; p < 2; p++) {                     // p = 0: pair j = k1 (coefficients 2 k1, 2 k1 + 1); p = 1: pair j = k2
                    const float re0 = mdct[2*p] * norm,   im0 = mdst[2*p] * norm;
                    const float re1 = mdct[2*p+1] * norm, im1 = mdst[2*p+1] * norm;
                    const float re0s = re0 * re0, im0s = im0 * im0, re1s = re1 * re1, im1s = im1 * im1;
                    const float a0 = re0s + im0s, a1 = re1s + im1s;
                    nnz += (fabsf(re0) < 0.5f * ULCX_COEF_EPS) ? 0 : 1;
                    nnz += (fabsf(re1) < 0.5f * ULCX_COEF_EPS) ? 0 : 1;
                    // k1 side: pairs kA, kB ascending; k2 side: pairs k2(kB) = k2(kA) - 1 then k2(kA): ascending too
                    const int slot = p ? (1 - u) : u;
                    re[q][p][2 * slot] = re0; re[q][p][2 * slot + 1] = re1;
                    ns[q][p][slot] = a0 + a1;                     // (0 + a0) + a1
                    am[p][slot] += a0; am[p][slot] += a1;         // channel order preserved (q = 0 first)
                }
            }
        }
        const int j1 = kA, j2 = M - 1 - kB;                       // first pair index of each side
#pragma unroll
        for (int q = 0; q < 2; q++) {
            stnt((float4 *)(coefO + q * BS + 2 * j1), make_float4(re[q][0][0], re[q][0][1], re[q][0][2], re[q][0][3]));
            stnt((float4 *)(coefO + q * BS + 2 * j2), make_float4(re[q][1][0], re[q][1][1], re[q][1][2], re[q][1][3]));
            stnt((float2 *)(c.nsum + tile_idx(BS / 2, blk * 2 + q, j1)), make_float2(ns[q][0][0], ns[q][0][1]));
            stnt((float2 *)(c.nsum + tile_idx(BS / 2, blk * 2 + q, j2)), make_float2(ns[q][1][0], ns[q][1][1]));
        }
        *(float2 *)(c.amp2 + tile_idx(BS / 2, blk, j1)) = make_float2(am[0][0], am[0][1]);
        *(float2 *)(c.amp2 + tile_idx(BS / 2, blk, j2)) = make_float2(am[1][0], am[1][1]);
    }
    return nnz;
}

// which blocks take the steady-state path: stereo BlockSize 2048 (the caller's business), un-decimated, full overlap on
// both sides; PCM16 ingest from the call's third block on (its history halves are float)
template <typename IN>
__device__ __forceinline__ bool xf_is_fast(const UlcxEncCtx &c, int s, int k) {
#ifdef XF_NO_FAST
    return false;
#endif
    const int BS = c.BS;
    const int *wrow = c.wcArr + (size_t)s * (c.maxK + 2) + k;
    const int wcPrev = wrow[0], wc = wrow[1], wcNext = wrow[2];
    if (!(k >= 2 || std::is_same<IN, float>::value) || (ulcx_pattern(wc) >> 4) != 0) return false;
    unsigned pp = ulcx_pattern(wcPrev);
    int lastS = BS;
    do { lastS = BS >> (pp & 7); } while (pp >>= 4);
    int ovFirst = first_overlap(wc, BS);
    if (ovFirst > lastS) ovFirst = lastS;
    return ovFirst == BS && first_overlap(wcNext, BS) >= BS;
}
// One block (s, k) of the call by one workgroup: any window, any channel count up to BlockSize 8192.
// ST: stereo instantiation (C = 2 as a compile-time constant: one channel pair, no per-pair branches)
template <bool ST, typename IN>
__device__ __forceinline__ void xf_block(const UlcxEncCtx &c, float *lds, int s, int k, const int tid) {
    const int BS = c.BS, C = ST ? 2 : c.C;
    const int blk = s * c.K + k;
    const int ps = ulcx_xf_pad_shift(BS, C);          // FFT array padding (ulcx_fft.h)
    float2 *z    = (float2 *)lds;                     // 4 arrays of up to BS/2 complex: {MDCT, MDST} x {ch, ch+1}
    float2 *twl  = (float2 *)(lds + 4 * FFT_PADDEDS(BS, ps));    // BS/4 complex: this subblock's FFT twiddles (no global-memory latency inside the FFT passes)
    int    &s_nnz = *(int *)(lds + 4 * FFT_PADDEDS(BS, ps) + BS / 2);  // (inside the dynamic region: no static LDS in front of it)
    const bool ampLds = (C > 2);                             // line energies accumulate across channel pairs: only then in LDS
    const bool twInLds = !(ampLds && (size_t)16 * (BS + (BS >> ps)) + (size_t)BS * 4 + 32 > ULCX_LDS_LIMIT);   // (BlockSize 8192 with C > 2: no room, twiddles from global memory)
    float  *amp2 = twInLds ? lds + 4 * FFT_PADDEDS(BS, ps) + BS / 2 + 4 : lds + 4 * FFT_PADDEDS(BS, ps);   // BS/2 (takes the twiddles' place when they are not resident)
    if (tid == 0) s_nnz = 0;
    if (ampLds) for (int i = tid; i < BS / 2; i += WG) amp2[i] = 0.0f;

    const int *wrow = c.wcArr + (size_t)s * (c.maxK + 2) + k;
    int wcPrev = wrow[0], wc = wrow[1], wcNext = wrow[2];
    if (c.barkRing && tid == 0 && (ulcx_pattern(wc) & ~8u) != 0) c.decList[atomicAdd(c.decCount, 1)] = blk;     // (its Bark sums take the lane-per-subblock kernels)
    int nextOv = first_overlap(wcNext, BS);
    int ovFirst;                                       // right overlap of the previous block's last subblock
    {
        unsigned pp = ulcx_pattern(wcPrev);
        int lastS = BS;
        do { lastS = BS >> (pp & 7); } while (pp >>= 4);
        ovFirst = first_overlap(wc, BS);
        if (ovFirst > lastS) ovFirst = lastS;
    }
    size_t cb = (size_t)C * BS;
    float *coefO = c.coef + (size_t)blk * cb;
    int nnz = 0;
    __syncthreads();

    // the steady state of the headline geometry (and of BlockSize 4096) takes the all-constants path (xf_fast)
    const bool fastBlk = ST && (BS == 2048 || BS == 4096) && xf_is_fast<IN>(c, s, k);
    if (fastBlk) nnz = (BS == 2048) ? xf_fast<IN, 2048>(c, lds, s, k, blk, tid) : xf_fast<IN, 4096>(c, lds, s, k, blk, tid);
    else
    for (int ch0 = 0; ch0 < C; ch0 += 2) {             // one M/S pair (or a trailing single channel) at a time
        const int nch = ST ? 2 : ((ch0 + 1 < C) ? 2 : 1);
        unsigned pat = ulcx_pattern(wc);
        int off = 0, ovL = ovFirst;
        do {
            int S = BS >> (pat & 7);
            int d = pat & 7;
            pat >>= 4;
            int ov;
            if (pat) { ov = BS >> (pat & 7); if (pat & 8) ov >>= (wc & 7); }
            else ov = nextOv;
            if (ov > S) ov = S;
            const int M = S >> 1;
            // subblock span starts at b = (k-1.5)BS + off; frame = [b - S/2, b + 3S/2)
            int t0 = (k - 1) * BS - BS / 2 + off - S / 2;
            int aL = (S - ovL) >> 1, aR = (S - ov) >> 1;
            const float *rise = c.T.winRise + ovL, *fall = c.T.winFall + ov;
            const float2 *pre = c.T.pre[d];
            const int Mp = FFT_PADDEDS(M, ps);                // arrays are stored padded (ulcx_fft.h)
            float2 *zc0 = z, *zs0 = z + Mp, *zc1 = z + 2 * Mp, *zs1 = z + 3 * Mp;

            // 1. TDAC fold + DCT-IV pre-twiddle straight from the input timeline.
            //    Fold index n uses frame positions {M-1-2n, M+2n, S+M-1-2n, S+M+2n} (n < M/2) or their
            //    mirror images (n >= M/2); n = M/2-1-j and n = M/2+j use ADJACENT positions in all four
            //    quarters of the frame, so one lane takes both: four 16-byte loads per lane, each wave
            //    reading four contiguous 1 KB runs.
            if (twInLds) for (int i = tid; i < M / 2; i += WG) twl[i] = c.T.tw[d][i];       // visible after the barrier that ends the fold
            // Two wave-uniform specialisations of the same loop: INPCM = the whole frame lies in this call's input (no
            // history pointer select per load; every block but the first two of a call), FULLOV = both overlaps span the
            // whole subblock (the steady state: every position is on a ramp, no clamps or selects in the window).
            auto fold = [&](auto inpcmT, auto fullovT) {
                constexpr bool INPCM = decltype(inpcmT)::value, FULLOV = decltype(fullovT)::value;
                const IN *frame = pcm_base<IN>(c) + ((size_t)s * c.K * BS + (INPCM ? t0 : 0)) * C;
                auto ldE = [&](int pos, int e) -> float { return INPCM ? ld1(frame + (size_t)pos * C + e) : smp_ld1<IN>(c, s, t0 + pos, e); };
                auto ldQ = [&](int pos) -> float4 { return INPCM ? ld4(frame + (size_t)pos * C) : smp_ld4<IN>(c, s, t0 + pos); };
                for (int jj = tid; jj < M / 2; jj += WG) {
                    const int iA = 2 * jj, iB = S - 2 - 2 * jj, iC = S + 2 * jj, iD = 2 * S - 2 - 2 * jj;
                    float2 xs[8];                           // (ch0, ch0+1) after M/S at iA, iA+1, iB, iB+1, iC, iC+1, iD, iD+1
                    {
                        const int ip[4] = { iA, iB, iC, iD };
#pragma unroll
                        for (int r = 0; r < 4; r++) {
                            float2 e0, e1;                  // positions ip[r] and ip[r]+1 (same side of the history boundary: ip[r] and t0 are even)
                            if (nch == 2) {
                                if (C == 2) { float4 v = ldQ(ip[r]); e0 = make_float2(v.x, v.y); e1 = make_float2(v.z, v.w); }
                                else { e0 = make_float2(ldE(ip[r], ch0), ldE(ip[r], ch0 + 1)); e1 = make_float2(ldE(ip[r] + 1, ch0), ldE(ip[r] + 1, ch0 + 1)); }
                                // M/S (BlockTransform.c:102-110)
                                xs[2 * r]     = make_float2((e0.x + e0.y) * 0.5f, (e0.x - e0.y) * 0.5f);
                                xs[2 * r + 1] = make_float2((e1.x + e1.y) * 0.5f, (e1.x - e1.y) * 0.5f);
                            } else {
                                xs[2 * r] = make_float2(ldE(ip[r], ch0), 0.0f); xs[2 * r + 1] = make_float2(ldE(ip[r] + 1, ch0), 0.0f);
                            }
                        }
                    }
                    // window, branch-free (same factor for every channel).  Rising half (iA.., iB..): 0 below the ramp,
                    // rise[] on it, x itself above (x * 1.0f is x); falling half (iC.., iD..) mirrored.  win_apply() is the
                    // readable form of the same thing.
                    const int ipos[8] = { iA, iA + 1, iB, iB + 1, iC, iC + 1, iD, iD + 1 };
#pragma unroll
                    for (int r = 0; r < 8; r++) {
                        if (FULLOV) {
                            float f = (r < 4) ? rise[ipos[r]] : fall[ipos[r] - S];
                            xs[r].x *= f; xs[r].y *= f;
                            continue;
                        }
                        float f; bool zero;
                        if (r < 4) {
                            int idx = ipos[r] - aL;
                            int ci = idx < 0 ? 0 : (idx < ovL ? idx : 0);
                            float tv = rise[ci];                     // rise[0] exists for ovL = 0 too (table row of the zero overlap)
                            f = (idx < ovL) ? tv : 1.0f; zero = idx < 0;
                        } else {
                            int idx = ipos[r] - S - aR;
                            int ci = idx < 0 ? 0 : (idx < ov ? idx : 0);
                            float tv = fall[ci];
                            f = (idx < 0) ? 1.0f : tv; zero = idx >= ov;
                        }
                        float wx = xs[r].x * f, wy = xs[r].y * f;
                        xs[r].x = zero ? 0.0f : wx;
                        xs[r].y = zero ? 0.0f : wy;
                    }
#pragma unroll
                    for (int hsel = 0; hsel < 2; hsel++) {
                        // hsel 0: n = M/2-1-jj (Lb = iA+1, La = iB, Ra = iC+1, Rb = iD);  hsel 1: n = M/2+jj (Lb = iA, La = iB+1, Ra = iC, Rb = iD+1)
                        const int n = hsel ? M / 2 + jj : M / 2 - 1 - jj;
                        const float2 lbv = hsel ? xs[0] : xs[1], lav = hsel ? xs[3] : xs[2];
                        const float2 rav = hsel ? xs[4] : xs[5], rbv = hsel ? xs[7] : xs[6];
                        float2 P = pre[n];
#pragma unroll
                        for (int q = 0; q < 2; q++) {
                            if (q >= nch) break;
                            float ra = q ? rav.y : rav.x, rb = q ? rbv.y : rbv.x, la = q ? lav.y : lav.x, lb = q ? lbv.y : lbv.x;
                            float vr = ra + rb, wr = ra - rb;            // v[mr], w[mr]
                            float vl = la - lb, wl = lb + la;            // v[ml], w[ml]
                            float v1 = hsel ? vl : vr, v2 = hsel ? vr : vl;  // v[2n], v[S-1-2n]
                            float w1 = hsel ? wl : wr, w2 = hsel ? wr : wl;  // w[2n], w[S-1-2n]
                            float2 *zc = q ? zc1 : zc0, *zs = q ? zs1 : zs0;
                            zc[FFT_PADS(n, ps)] = cmulc(make_float2(v1, v2), P);       // u = v      : (u[2n], u[S-1-2n])
                            zs[FFT_PADS(n, ps)] = cmulc(make_float2(w2, w1), P);       // u = rev(w) : (w[S-1-2n], w[2n])
                        }
                    }
                }
            };
            if (!(ULCX_DBG(c) & 2)) {
                const bool inPcm = (t0 >= 0), fullOv = (ovL == S) && (ov == S);
                if (inPcm && fullOv) fold(std::true_type{}, std::true_type{});
                else if (inPcm) fold(std::true_type{}, std::false_type{});
                else fold(std::false_type{}, std::false_type{});
            }
            __syncthreads();

            // 2. 2*nch M-point FFTs in LDS, one wave per array, no barriers in between
            if (!(ULCX_DBG(c) & 1)) for (int a = __builtin_amdgcn_readfirstlane(tid >> 6); a < 2 * nch; a += WG / 64) {
                if (twInLds) fft_wave_dif(z + a * Mp, M, twl, tid & 63, ps);
                else fft_wave_dif(z + a * Mp, M, c.T.tw[d], tid & 63, ps);
            }
            __syncthreads();

            // 3. post-twiddle + normalise + keys + per-line energies (BlockTransform.c:243-281)
            int bits = 31 - __clz(M);
            float norm = 2.0f / S;
            if (!(ULCX_DBG(c) & 4)) for (int kk = tid; kk < M / 2; kk += WG) {
                int k1 = kk, k2 = M - 1 - kk;
                int r1 = (int)(__brev((unsigned)k1) >> (32 - bits));
                int r2 = (int)(__brev((unsigned)k2) >> (32 - bits));
                float2 P1 = pre[k1], P2 = pre[k2];
                float am1 = 0.0f, am2 = 0.0f;
                if (ampLds) { am1 = amp2[off / 2 + k1]; am2 = amp2[off / 2 + k2]; }
#pragma unroll
                for (int q = 0; q < 2; q++) {
                    if (q >= nch) break;
                    int ch = ch0 + q;
                    float2 *zc = q ? zc1 : zc0, *zs = q ? zs1 : zs0;
                    // (packed-f32 complex multiplies: lane-wise IEEE, the same two products and two fused multiply-adds as cmulc)
                    const fft_v2f Pv1 = { P1.x, P1.y }, Pv2 = { P2.x, P2.y };
                    // (Re y, -Im y) of the DCT-IV post-twiddle, as cmulc_post; the MDST's sign (it alternates, (-1)^k) only ever
                    //  meets a square
                    auto cm = [](float2 d, fft_v2f w) { const fft_v2f dv = { d.x, d.y }; const fft_v2f r = fft_cmulc_post_pk(dv, w); return make_float2(r.x, r.y); };
                    float2 yc1 = cm(zc[FFT_PADS(r1, ps)], Pv1), yc2 = cm(zc[FFT_PADS(r2, ps)], Pv2);
                    float2 ys1 = cm(zs[FFT_PADS(r1, ps)], Pv1), ys2 = cm(zs[FFT_PADS(r2, ps)], Pv2);
                    // pair j = k1: coefficients 2k1, 2k1+1 ; pair j = k2: coefficients 2k2, 2k2+1
                    float mdct[4] = { yc1.x, yc2.y, yc2.x, yc1.y };
                    float mdst[4] = { ys1.x, ys2.y, ys2.x, ys1.y };
#pragma unroll
                    for (int p = 0; p < 2; p++) {
                        float re0 = mdct[2*p] * norm,   im0 = mdst[2*p] * norm;
                        float re1 = mdct[2*p+1] * norm, im1 = mdst[2*p+1] * norm;
                        float re0s = re0 * re0, im0s = im0 * im0, re1s = re1 * re1, im1s = im1 * im1;
                        float a0 = re0s + im0s, a1 = re1s + im1s;
                        // (the importance key FastLog(Re^2) is a function of the stored coefficient: the kernels that consume
                        //  keys form it from there, key0_of(), instead of this one writing 4 more bytes per coefficient)
                        nnz += (fabsf(re0) < 0.5f * ULCX_COEF_EPS) ? 0 : 1;
                        nnz += (fabsf(re1) < 0.5f * ULCX_COEF_EPS) ? 0 : 1;
                        int j = p ? k2 : k1;
                        size_t gi = (size_t)ch * BS + off + 2 * j;
                        stnt((float2 *)(coefO + gi), make_float2(re0, re1));
                        stnt(c.nsum + tile_idx(BS / 2, blk * C + ch, off / 2 + j), a0 + a1);       // (0 + a0) + a1
                        if (p) { am2 += a0; am2 += a1; } else { am1 += a0; am1 += a1; } // channel order preserved
                    }
                }
                if (ampLds) { amp2[off / 2 + k1] = am1; amp2[off / 2 + k2] = am2; }
                else { c.amp2[tile_idx(BS / 2, blk, off / 2 + k1)] = am1; c.amp2[tile_idx(BS / 2, blk, off / 2 + k2)] = am2; }
            }
            __syncthreads();
            off += S; ovL = ov;
        } while (pat);
    }
    // wave-reduce the non-zero count
    for (int o = 32; o > 0; o >>= 1) nnz += __shfl_down(nnz, o);
    if ((tid & 63) == 0) atomicAdd(&s_nnz, nnz);
    if (ampLds) for (int i = tid; i < BS / 2; i += WG) c.amp2[tile_idx(BS / 2, blk, i)] = amp2[i];
    __syncthreads();
    if (tid == 0) c.nnz[blk] = s_nnz;
}

// Blocks [k0, k1) of every stream, one workgroup each (the chunks of the window-control pipeline: every geometry but the
// headline one, and small calls).  Workgroups are dealt round-robin over the 8 XCDs (b and b+8 share one, each XCD has its
// own L2); consecutive blocks of a stream read overlapping input (a frame spans two blocks), so an XCD gets a contiguous
// run of blocks: block = (b % 8) * ceil(NB/8) + b / 8.  Speed only, never correctness.
template <bool ST, typename IN>
__global__ __launch_bounds__(WG, 4) void k_xf(UlcxEncCtx c, int k0, int k1) {
    extern __shared__ float lds[];
    const int kc = k1 - k0;
    const int NBk = c.B * kc;
    const int per = (NBk + 7) / 8;
    const int vb = (int)(blockIdx.x % 8) * per + (int)(blockIdx.x / 8);
    if (vb >= NBk) return;
    xf_block<ST, IN>(c, lds, vb / kc, k0 + vb % kc, threadIdx.x);
}

// ---------------------------------------------------------------------------
// The same transform for BlockSize > 8192 (ulcEncoder.c:32-34 accepts up to 32768): the four arrays of a channel pair do
// not fit in LDS together, so ONE array at a time - per channel the MDCT (coefficients, non-zero count), then the MDST
// (line energies; Re^2 re-formed from the stored coefficient) - folded sample by sample through ms_sample()/win_apply(),
// transformed by the whole workgroup (fftn_dif: the same butterflies, a barrier per pass).  Every arithmetic step is the
// one k_xf takes, in the same order; not tuned (such block sizes are not a throughput case).
// ---------------------------------------------------------------------------
template <typename IN>
__global__ __launch_bounds__(WG) void k_xf_big(UlcxEncCtx c, int k0, int k1) {
    extern __shared__ float lds[];
    const int BS = c.BS, C = c.C;
    const int kc = k1 - k0;
    const int NBk = c.B * kc;
    const int vb = blockIdx.x;
    if (vb >= NBk) return;
    const int s = vb / kc, k = k0 + vb % kc;
    const int blk = s * c.K + k;
    const int tid = threadIdx.x;
    float2 *z = (float2 *)lds;                         // one array of up to BS/2 complex, unpadded
    __shared__ int s_nnz;
    if (tid == 0) s_nnz = 0;
    const int *wrow = c.wcArr + (size_t)s * (c.maxK + 2) + k;
    const int wcPrev = wrow[0], wc = wrow[1], wcNext = wrow[2];
    if (c.barkRing && threadIdx.x == 0 && (ulcx_pattern(wc) & ~8u) != 0) c.decList[atomicAdd(c.decCount, 1)] = blk;
    const int nextOv = first_overlap(wcNext, BS);
    int ovFirst;                                       // right overlap of the previous block's last subblock
    {
        unsigned pp = ulcx_pattern(wcPrev);
        int lastS = BS;
        do { lastS = BS >> (pp & 7); } while (pp >>= 4);
        ovFirst = first_overlap(wc, BS);
        if (ovFirst > lastS) ovFirst = lastS;
    }
    const size_t cb = (size_t)C * BS;
    float *coefO = c.coef + (size_t)blk * cb;
    int nnz = 0;
    __syncthreads();
    for (int ch = 0; ch < C; ch++) {
        unsigned pat = ulcx_pattern(wc);
        int off = 0, ovL = ovFirst;
        do {
            const int S = BS >> (pat & 7);
            const int d = pat & 7;
            pat >>= 4;
            int ov;
            if (pat) { ov = BS >> (pat & 7); if (pat & 8) ov >>= (wc & 7); }
            else ov = nextOv;
            if (ov > S) ov = S;
            const int M = S >> 1;
            const int t0 = (k - 1) * BS - BS / 2 + off - S / 2;   // the subblock's frame = [t0, t0 + 2S) (closed form of the lapping FIFO, as k_xf)
            const int aL = (S - ovL) >> 1, aR = (S - ov) >> 1;
            const float *rise = c.T.winRise + ovL, *fall = c.T.winFall + ov;
            const float2 *pre = c.T.pre[d];
            const int bits = 31 - __clz(M);
            const float norm = 2.0f / S;
            for (int kind = 0; kind < 2; kind++) {                // 0: MDCT, 1: MDST
                // 1. TDAC fold + DCT-IV pre-twiddle
                for (int jj = tid; jj < M / 2; jj += WG) {
                    const int iA = 2 * jj, iB = S - 2 - 2 * jj, iC = S + 2 * jj, iD = 2 * S - 2 - 2 * jj;
                    const int ipos[8] = { iA, iA + 1, iB, iB + 1, iC, iC + 1, iD, iD + 1 };
                    float xs[8];
#pragma unroll
                    for (int r = 0; r < 8; r++) xs[r] = win_apply(ms_sample<IN>(c, s, t0 + ipos[r], ch), ipos[r], S, aL, ovL, aR, ov, rise, fall);
#pragma unroll
                    for (int hsel = 0; hsel < 2; hsel++) {
                        const int n = hsel ? M / 2 + jj : M / 2 - 1 - jj;
                        const float lb = hsel ? xs[0] : xs[1], la = hsel ? xs[3] : xs[2];
                        const float ra = hsel ? xs[4] : xs[5], rb = hsel ? xs[7] : xs[6];
                        const float vr = ra + rb, wr = ra - rb;
                        const float vl = la - lb, wl = lb + la;
                        const float v1 = hsel ? vl : vr, v2 = hsel ? vr : vl;
                        const float w1 = hsel ? wl : wr, w2 = hsel ? wr : wl;
                        z[n] = kind ? cmulc(make_float2(w2, w1), pre[n]) : cmulc(make_float2(v1, v2), pre[n]);
                    }
                }
                __syncthreads();
                // 2. M-point FFT by the workgroup
                fftn_dif(z, 1, M, c.T.tw[d], tid);
                // 3. post-twiddle + normalise (BlockTransform.c:243-281)
                for (int kk = tid; kk < M / 2; kk += WG) {
                    const int kA = kk, kB = M - 1 - kk;
                    const int r1 = (int)(__brev((unsigned)kA) >> (32 - bits));
                    const int r2 = (int)(__brev((unsigned)kB) >> (32 - bits));
                    const float2 y1 = cmulc_post(z[r1], pre[kA]), y2 = cmulc_post(z[r2], pre[kB]);      // (Re y, -Im y)
#pragma unroll
                    for (int p = 0; p < 2; p++) {
                        const int j = p ? kB : kA;
                        const size_t gi = (size_t)ch * BS + off + 2 * j;
                        if (kind == 0) {
                            const float m0 = p ? y2.x : y1.x, m1 = p ? y1.y : y2.y;
                            const float re0 = m0 * norm, re1 = m1 * norm;
                            nnz += (fabsf(re0) < 0.5f * ULCX_COEF_EPS) ? 0 : 1;
                            nnz += (fabsf(re1) < 0.5f * ULCX_COEF_EPS) ? 0 : 1;
                            *(float2 *)(coefO + gi) = make_float2(re0, re1);
                        } else {
                            const float m0 = p ? y2.x : y1.x, m1 = p ? y1.y : y2.y;
                            const float im0 = m0 * norm, im1 = m1 * norm;
                            const float2 re = *(const float2 *)(coefO + gi);
                            const float re0s = re.x * re.x, im0s = im0 * im0, re1s = re.y * re.y, im1s = im1 * im1;
                            const float a0 = re0s + im0s, a1 = re1s + im1s;
                            c.nsum[tile_idx(BS / 2, blk * C + ch, off / 2 + j)] = a0 + a1;       // (0 + a0) + a1
                            float *ap = c.amp2 + tile_idx(BS / 2, blk, off / 2 + j);
                            float am = (ch == 0) ? 0.0f : *ap;                           // channel order preserved
                            am += a0; am += a1;
                            *ap = am;
                        }
                    }
                }
                __syncthreads();
            }
            off += S; ovL = ov;
        } while (pat);
    }
    for (int o = 32; o > 0; o >>= 1) nnz += __shfl_down(nnz, o);
    if ((tid & 63) == 0) atomicAdd(&s_nnz, nnz);
    __syncthreads();
    if (tid == 0) c.nnz[blk] = s_nnz;
}

// ---------------------------------------------------------------------------
// Block complexity + nOutCoef (BlockTransform.c:279-325, ulcEncoder.c:93-158)
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(64) void k_cplx(UlcxEncCtx c, int k0, int k1) {
    const int gidc = blockIdx.x * 64 + threadIdx.x, kcc = k1 - k0;     // blocks [k0, k1) of every stream
    if (gidc >= c.B * kcc) return;
    const int blk = (gidc / kcc) * c.K + k0 + gidc % kcc;
    int n = c.C * c.BS;
    const float4 *p = (const float4 *)(c.coef + (size_t)blk * n);
    float cx = 0.0f, cw = 0.0f;
    int tiny = 0;
    // n is a multiple of 256: 4 x 16-byte loads in flight per step.  The count of collapsible coefficients only feeds
    // the CBR/ABR probe shortcut below: VBR calls take the loop without it (half the instructions of this
    // issue-bound kernel; the branch is uniform for the whole launch).
    auto sums = [&](auto tinyT) {
        constexpr bool TINY = decltype(tinyT)::value;
        for (int i = 0; i < n / 4; i += 4) {
            float4 q[4];
#pragma unroll
            for (int u = 0; u < 4; u++) q[u] = p[i + u];
#pragma unroll
            for (int u = 0; u < 4; u++) {
                float4 v = q[u];
                cx += v.x * v.x; cw += fabsf(v.x);
                cx += v.y * v.y; cw += fabsf(v.y);
                cx += v.z * v.z; cw += fabsf(v.z);
                cx += v.w * v.w; cw += fabsf(v.w);
                if (TINY) {
                    // non-zero coefficients so small that the coarsest quantizer (2^31) could collapse them (Encode.c:114)
                    tiny += (fabsf(v.x) >= 0.5f * ULCX_COEF_EPS && fabsf(v.x) < 0x1.0p-29f) ? 1 : 0;
                    tiny += (fabsf(v.y) >= 0.5f * ULCX_COEF_EPS && fabsf(v.y) < 0x1.0p-29f) ? 1 : 0;
                    tiny += (fabsf(v.z) >= 0.5f * ULCX_COEF_EPS && fabsf(v.z) < 0x1.0p-29f) ? 1 : 0;
                    tiny += (fabsf(v.w) >= 0.5f * ULCX_COEF_EPS && fabsf(v.w) < 0x1.0p-29f) ? 1 : 0;
                }
            }
        }
    };
    if (c.mode == ULCX_MODE_VBR) sums(std::false_type{}); else sums(std::true_type{});
    if (cx != 0.0f) {
        cx = ulcx_logf((cw * cw) / cx) / c.cplxScale;
        if (cx < 0.0f) cx = 0.0f;
        if (cx > 1.0f) cx = 1.0f;
    }
    c.cplx[blk] = cx;
    int maxCoef = c.nnz[blk];
    if (c.mode == ULCX_MODE_VBR) {
        int nT = maxCoef;
        if (c.vbrTarget > 0.0f) {
            float ft = (c.C * c.BS) * cx / c.vbrTarget;
            if (ft < maxCoef) nT = (int)ft;
        }
        c.nout[blk] = nT;
    } else {
        // CBR/ABR binary search state (ulcEncoder.c:96-101)
        float kbps = c.p0;
        if (c.mode == ULCX_MODE_ABR) kbps = c.p0 * cx / c.p1;
        int budget = (int)((c.BS * kbps) * 1000.0f / c.rateHz);
        int lo = 0, hi = maxCoef;
        int done = (0 < maxCoef) ? 0 : 1;
        int nOut = (0 < maxCoef) ? (int)((unsigned)(0 + maxCoef) / 2u) : 0;
        // Probes that are over budget for certain are taken without encoding anything.
        // A kept coefficient is coded with >= 1 nybble unless it collapses (|c|*2^q < 2.5, Encode.c:114), and inside a
        // quantizer zone max <= 4*min with max*2^q in (12, 48] unless q is clamped at 31 (Encode.c:50-87, :218-269):
        // only coefficients below 2.5*2^-31 can collapse.  So a probe at nOut writes more than nOut - tiny nybbles,
        // and 4*(nOut - tiny + 1) > budget is exactly the "Size > BitBudget" branch of ulcEncoder.c:103-110.
        while (!done && 4 * (nOut - tiny + 1) > budget) {
            hi = nOut - 1;
            if (!(lo < hi - 1)) { done = 1; nOut = lo; }
            else nOut = (int)((unsigned)(lo + hi) / 2u);
        }
        c.cbrLo[blk] = lo; c.cbrHi[blk] = hi;
        c.cbrDone[blk] = done;
        c.selWin[blk] = make_uint4(0u, 0u, (uint32_t)(c.C * c.BS), 0u);      // the key window of the block's probes: everything
        for (int u = 0; u < c.C * 4; u++) c.tailSum[((size_t)blk * c.C * 4 + u) * 8 + 6] = 0.0f;      // k_tails: no tail sums of this call yet
        // rate searches still open (the probe passes leave at once when it reaches 0): one atomic per wave, not per block -
        // half a million adds to one word are 3 ms
        {
            const unsigned long long open = __ballot(!done);
            if (open && (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(open >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)open, 0u)) == 0 && !done)
                atomicAdd(c.cbrLive, (int)__popcll(open));
        }
        c.nout[blk] = nOut;
        c.cbrBudget[blk] = budget;
    }
    // per-call flags of the block, cleared here instead of by three fill launches in front of the selection (which waits
    // for this kernel): exact-path membership, the wave writer's retry state, and once per call the two queue counters
    c.isFb[blk] = 0;
    if (c.useWave) c.slow[blk] = 0;
    if (blk == 0) { *c.fbCount = 0; if (c.useWave) { c.slow[c.B * c.K] = 0; c.slow[c.B * c.K + 1] = 0; } }
    int s = blk / c.K, k = blk % c.K;
    if (c.wcOut)   c.wcOut[blk]   = c.wcArr[(size_t)s * (c.maxK + 2) + k + 1];
    if (c.cplxOut) c.cplxOut[blk] = cx;
}

// ---------------------------------------------------------------------------
// Bark-band levels.  Both routines accumulate three binary64 running sums line by
// line, a "low" and a "high" cursor per band (Psyopt.c:23-51); kept sequential, one
// lane per (block[,channel],subblock).
// ---------------------------------------------------------------------------
struct LineSum { int end; double fl, pk, pw; };
// one line into the three ordered binary64 running sums (Psyopt.c:23-51)
__device__ __forceinline__ void linesum_add(float vf, double &fl, double &pk, double &pw) {
    double v = (double)vf;
    double vl = (double)fastlog(0x1.0p-126f + vf);
    fl += vl;
    pk += vl * v;
    pw += v;
}
// Advance the running prefix to `end`.  `src` points at the unit's line 0 inside the tiled array (tile_off: a unit starts at a
// multiple of 32 lines) and is 16-byte aligned there: the body goes in aligned groups of four
// lines per load (one lane per unit means every load instruction touches 64 different cache lines, so these kernels
// are bound by the number of load instructions: 16 bytes per lane instead of 4 cuts them fourfold).  `prev` receives
// the prefix one line before `end` (the lower edge of a later band is floor(x) where this upper edge is ceil(x)).
__device__ __forceinline__ void linesum_advance(const float *src, LineSum &ls, int end, LineSum *prev = nullptr) {
    double fl = ls.fl, pk = ls.pk, pw = ls.pw;
    int l = ls.end;
    const int stop = (prev && end > l) ? end - 1 : end;              // stop one line early to take the snapshot
    while (l < stop && (l & 3)) { linesum_add(src[tile_off(l)], fl, pk, pw); l++; }
    for (; l + 8 <= stop; l += 8) {                                   // two aligned 16-byte loads in flight; sums keep the reference's order
        float4 a = *(const float4 *)(src + tile_off(l)), b = *(const float4 *)(src + tile_off(l + 4));
        linesum_add(a.x, fl, pk, pw); linesum_add(a.y, fl, pk, pw); linesum_add(a.z, fl, pk, pw); linesum_add(a.w, fl, pk, pw);
        linesum_add(b.x, fl, pk, pw); linesum_add(b.y, fl, pk, pw); linesum_add(b.z, fl, pk, pw); linesum_add(b.w, fl, pk, pw);
    }
    for (; l + 4 <= stop; l += 4) {
        float4 a = *(const float4 *)(src + tile_off(l));
        linesum_add(a.x, fl, pk, pw); linesum_add(a.y, fl, pk, pw); linesum_add(a.z, fl, pk, pw); linesum_add(a.w, fl, pk, pw);
    }
    for (; l < stop; l++) linesum_add(src[tile_off(l)], fl, pk, pw);
    if (prev) {
        prev->end = l; prev->fl = fl; prev->pk = pk; prev->pw = pw;
        if (l < end) { linesum_add(src[tile_off(l)], fl, pk, pw); l++; }
    }
    ls.end = end; ls.fl = fl; ls.pk = pk; ls.pw = pw;
}
// lower edge of a band: the upper cursor has already been there (its stop for an earlier band, or one line before it)
__device__ __forceinline__ void linesum_seek(const float *src, LineSum &lo, int target, const LineSum &s0, const LineSum &s1) {
    if (s0.end == target) lo = s0;
    else if (s1.end == target) lo = s1;
    else linesum_advance(src, lo, target);
}

// unit geometry: subblock j of WindowCtrl wc -> size shift d, coefficient offset off
__device__ __forceinline__ bool unit_geom(int wc, int j, int BS, int &d, int &off, int &S) {
    unsigned pat = ulcx_pattern(wc);
    off = 0;
    for (int i = 0;; i++) {
        d = pat & 7; S = BS >> d;
        if (i == j) return true;
        off += S;
        pat >>= 4;
        if (!pat) return false;
    }
}

// Psyopt.c:185-225
__global__ __launch_bounds__(64) void k_nbark(UlcxEncCtx c, int useList) {
    int tid0 = blockIdx.x * 64 + threadIdx.x;
    // (with k_bark_uniform taking the un-decimated blocks: only the blocks k_xf listed as decimated)
    const int nBlk = useList ? *c.decCount : c.B * c.K;
    int nBC = nBlk * c.C;
    if (tid0 >= nBC * 4) return;
    // subblock index slowest: waves of j >= 1 are empty for un-decimated blocks and exit at once
    int j = tid0 / nBC, rem = tid0 - j * nBC, blk = rem / c.C, ch = rem - blk * c.C;
    if (useList) blk = c.decList[blk];
    int gid = (blk * c.C + ch) * 4 + j;
    int s = blk / c.K, k = blk % c.K;
    int wc = c.wcArr[(size_t)s * (c.maxK + 2) + k + 1];
    int d, off, S;
    if (!unit_geom(wc, j, c.BS, d, off, S)) return;
    int N = S / 2;
    const float *data = c.nsum + tile_idx(c.BS / 2, blk * c.C + ch, off / 2);
    float *bark = c.barkN + (size_t)gid * ULCX_NBARK;
    float level = -100.0f;
    // lower edge of band b = floor(x), upper edge of band b-2 = ceil(x) of the same x: the lower cursor takes the upper
    // cursor's value at its stop two bands ago (or one line before it) instead of summing the lines a second time
    LineSum lo = {0, 0.0, 0.0, 0.0}, hi = {0, 0.0, 0.0, 0.0};
    LineSum n0 = {-1, 0.0, 0.0, 0.0}, n1 = n0, o0 = n0, o1 = n0;     // stops (and stop-1) of bands b-1 and b-2
    for (int b = 0; b < ULCX_NBARK; b++) {
        int l0 = c.T.nBeg[d][b], l1 = c.T.nEnd[d][b];
        linesum_seek(data, lo, l0, o0, o1);
        o0 = n0; o1 = n1;
        linesum_advance(data, hi, l1, &n1);
        n0 = hi;
        double sf = hi.fl - lo.fl, sp = hi.pk - lo.pk, sw = hi.pw - lo.pw;
        if (sw > 0.0) {
            double scale = 1.0 / (double)(l1 - l0);
            sp = sp / sw;
            sf = sf * scale;
            level = 0.5f * (float)(ulcx_log(sw * scale) + sf - sp);
        }
        bark[b] = level;
    }
    (void)N;
}

// Psyopt.c:236-248: per-line interpolation + {w, w*(log+ln2)} pair of line pair jp (0 <= jp < BS/2) of one channel of a
// block: a function of that channel's [4][25] Bark levels alone.  Round 4: the pairs are no array in HBM any more (16 KB a
// block written by one kernel and read back by three: 15 % of the step's traffic) - k_nsums forms a block's pairs into LDS
// for the sums it takes, the bitstream writer's rare fall-backs form the few they need on the spot (SumSrc), and the parity
// tap materialises the array on request (k_nline).  SEXP: expf's 2^(i/32) table from an LDS copy (it sits in the middle of
// every evaluation's dependent chain).
template <bool SEXP>
__device__ __forceinline__ float2 noise_pair(const UlcxEncCtx &c, const float *bark4, int wc, int jp, const unsigned long long *sexp) {
    unsigned pat = ulcx_pattern(wc);
    int off = 0, d = 0, S = c.BS, j = 0;
    for (;; j++) { d = pat & 7; S = c.BS >> d; if (2 * jp < off + S) break; off += S; pat >>= 4; }
    const int line = jp - off / 2;
    const int bi = c.T.bandIdx[d][line];
    const float fr = c.T.bandFrac[d][line];
    const float *bark = bark4 + j * ULCX_NBARK;
    const float L = (bi < ULCX_NBARK) ? bark[bi] : bark[ULCX_NBARK - 1];
    const float R = (bi + 1 < ULCX_NBARK) ? bark[bi + 1] : L;
    const float noise = L * (1.0f - fr) + R * fr;
    const float w = SEXP ? ulcx_expf_t(0.5f * noise, sexp) : ulcx_expf(0.5f * noise);
    return make_float2(w, w * (noise + 0x1.62E430p-1f));
}
// the pair at float2 index p of the block's flattened [C][BS/2] pair array, from the Bark levels in global memory
struct SumSrc { const UlcxEncCtx *c; const float *bark; int wc; };      // bark: the block's [C][4][25] levels
__device__ __forceinline__ SumSrc sum_src(const UlcxEncCtx &c, int blk) {
    SumSrc g; g.c = &c; g.bark = c.barkN + (size_t)blk * c.C * 4 * ULCX_NBARK;
    g.wc = c.wcArr[(size_t)(blk / c.K) * (c.maxK + 2) + (blk % c.K) + 1];
    return g;
}
__device__ __forceinline__ float2 pair_demand(const SumSrc &g, int p) {
    const int half = g.c->BS >> 1, ch = p / half;
    return noise_pair<false>(*g.c, g.bark + ch * 4 * ULCX_NBARK, g.wc, p - ch * half, nullptr);
}
// parity tap only (ulcx_encoder_debug_fetch): the whole array, as the reference leaves it in TransformNoise
__global__ __launch_bounds__(WG) void k_nline(UlcxEncCtx c) {
    const int blk = blockIdx.x, half = c.BS / 2;
    const SumSrc g = sum_src(c, blk);
    float2 *dst = (float2 *)(c.npair + (size_t)blk * (c.C * c.BS));
    for (int p = threadIdx.x; p < c.C * half; p += WG) dst[p] = pair_demand(g, p);
}
void ulcx_enc_materialise_noise(const UlcxEncCtx &c, hipStream_t st) {
    hipLaunchKernelGGL(k_nline, dim3(c.B * c.K), dim3(WG), 0, st, c);
}

// Psyopt.c:86-137 on the channel-summed energies
__global__ __launch_bounds__(64) void k_pbark(UlcxEncCtx c, int useList) {
    int tid0 = blockIdx.x * 64 + threadIdx.x;
    int NBk = useList ? *c.decCount : c.B * c.K;
    if (tid0 >= NBk * 4) return;
    int j = tid0 / NBk, blk = tid0 - j * NBk;
    if (useList) blk = c.decList[blk];
    int gid = blk * 4 + j;
    int s = blk / c.K, k = blk % c.K;
    int wc = c.wcArr[(size_t)s * (c.maxK + 2) + k + 1];
    int d, off, S;
    if (!unit_geom(wc, j, c.BS, d, off, S)) return;
    const float *data = c.amp2 + tile_idx(c.BS / 2, blk, off / 2);
    float *bark = c.barkP + (size_t)gid * ULCX_NBARK;
    float unmask = 0.0f;
    LineSum lo = {0, 0.0, 0.0, 0.0}, hi = {0, 0.0, 0.0, 0.0};             // (as k_nbark; here the lower edge of band b is the upper edge of band b-1)
    LineSum n0 = {-1, 0.0, 0.0, 0.0}, n1 = n0;
    for (int b = 0; b < ULCX_NBARK; b++) {
        int l0 = c.T.pBeg[d][b], l1 = c.T.pEnd[d][b];
        linesum_seek(data, lo, l0, n0, n1);
        linesum_advance(data, hi, l1, &n1);
        n0 = hi;
        double sf = hi.fl - lo.fl, sp = hi.pk - lo.pk, sw = hi.pw - lo.pw;
        if (sw > 0.0) {
            sp = sp / sw;
            sf = sf / (double)(l1 - l0);
            unmask = (float)(sp - sf - ulcx_log(sw));
        }
        bark[b] = unmask;
    }
}

// k_nbark / k_pbark for the UN-DECIMATED blocks (about nine in ten; UlcxEncCtx::barkRing != 0): same sums in the same order,
// but every lane of a wave has the same subblock geometry, so the band edges are scalar control flow and the lines come
// through LDS.  A wave takes 64 consecutive rows (a row = the BS/2 lines of one block[,channel]) in tiles of 32 lines: a load
// instruction covers 128-byte pieces of eight rows (one lane per row reading global memory touches 64 cache lines per
// instruction), the tile goes to LDS row-padded, every lane then walks its own row.  Both cursors of the reference
// (Psyopt.c:23-51) are prefixes of one running sum from line 0: the lane keeps a single prefix and a snapshot of it at the
// lower edge of each band still open (ring in LDS); a band's three sums are prefix(upper edge) - snapshot, the very
// subtraction the reference makes.  Lanes whose block is decimated run along and store nothing (their blocks are on k_xf's
// list for the lane-per-subblock kernels).  The per-band arithmetic (binary64 log, divisions) is k_bark_levels, one lane
// per band.
// Round 3: the kernel is a workgroup of four waves per 64 rows.  A lane's 1024-line walk was bound by the instructions it
// issues per line (the FastLog polynomial, two conversions, a product, three sums: about 30), not by the three dependent
// sums - and a wave is one instruction stream.  So the work that does not depend on the running sums moves to the other
// three waves: they fetch a tile of 32 lines x 64 rows (a load instruction covers 128-byte pieces of eight rows), form
// FastLog of every value and leave {v, log v} pairs in LDS; wave 0 only walks its rows through the finished tile - two
// conversions, the product and the three ordered sums per line (7 instructions) - while the others prepare the next tile
// in the second buffer.  One barrier per tile.  Same sums, same order.
#define BK_TL 32                                           // lines per tile = the tile of the arrays (tile_idx)
#define BK_PPR (BK_TL / 4)                                 // 16-byte pieces per row of a tile
#define BK_PIECES (64 * BK_PPR)
#define BK_NPC ((BK_PIECES + 191) / 192)                   // pieces per producer lane
#ifndef BK_AHEAD
#define BK_AHEAD 4                                         // tiles of loads the producer waves keep in flight
#endif
#define BK_RS (2 * BK_TL + 4)                              // floats per row of a tile: {v, log v} pairs + pad (16-byte reads of 64 lanes conflict-free)
#define BK_TILE_FLOATS (64 * BK_RS)
template <bool NOISE>
__global__ __launch_bounds__(256) void k_bark_uniform(UlcxEncCtx c) {
    extern __shared__ double bk_lds[];
    const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6), RM = c.barkRing - 1;
    double *ring = bk_lds + lane;                                // [barkRing][3][64]
    float *tiles = (float *)(bk_lds + c.barkRing * 3 * 64);      // [2][64][BK_RS]
    const int half = c.BS / 2;
    const int nRows = NOISE ? c.B * c.K * c.C : c.B * c.K;
    const int row0 = blockIdx.x * 64;
    const int row = min(row0 + lane, nRows - 1);                 // (lanes past the end repeat the last row and store nothing)
    const int blk = NOISE ? row / c.C : row;
    const bool mine = (row0 + lane < nRows) && (ulcx_pattern(c.wcArr[(size_t)(blk / c.K) * (c.maxK + 2) + (blk % c.K) + 1]) & ~8u) == 0;
    if (!__ballot(mine)) return;                                 // (every wave of the workgroup sees the same 64 rows)
    const float *src = NOISE ? c.nsum : c.amp2;
    const int nT = half / BK_TL;
    if (wv > 0) {
        // producers: 192 lanes, a tile is 512 pieces of four lines (row = piece / 8, lines 4 (piece % 8) ..)
        const int p0 = (wv - 1) * 64 + lane;
        // BK_AHEAD tiles of loads in flight (a tile is consumed in well under a microsecond, a load from HBM takes two or
        // three beside the other kernels of the step): register sets rotate by unrolling the tile loop BK_AHEAD times
        constexpr int AH = BK_AHEAD;
        float4 nx[AH][BK_NPC];
        auto fetch = [&](float4 (&r)[BK_NPC], int t) {
#pragma unroll
            for (int i = 0; i < BK_NPC; i++) {
                const int pc = p0 + 192 * i;
                if (pc < BK_PIECES) r[i] = *(const float4 *)(src + ((size_t)(row0 >> 6) * nT + t) * (64 * BK_TL) + pc * 4);      // (a tile of the array IS a tile of this kernel)
            }
        };
        auto put = [&](const float4 (&r)[BK_NPC], int t) {
            float *tile = tiles + (t & 1) * BK_TILE_FLOATS;
#pragma unroll
            for (int i = 0; i < BK_NPC; i++) {
                const int pc = p0 + 192 * i;
                if (pc < BK_PIECES) {
                    const float4 v = r[i];
                    float4 *o = (float4 *)(tile + (pc / BK_PPR) * BK_RS + (pc % BK_PPR) * 8);
                    o[0] = make_float4(v.x, fastlog(0x1.0p-126f + v.x), v.y, fastlog(0x1.0p-126f + v.y));
                    o[1] = make_float4(v.z, fastlog(0x1.0p-126f + v.z), v.w, fastlog(0x1.0p-126f + v.w));
                }
            }
        };
        // tile u travels in register set u % AH: fetched AH tiles before it is put
#pragma unroll
        for (int a = 0; a < AH; a++) if (a < nT) fetch(nx[a], a);
        put(nx[0], 0);
        if (AH < nT) fetch(nx[0], AH);
        __syncthreads();
        for (int t0 = 0; t0 < nT; t0 += AH) {
#pragma unroll
            for (int a = 0; a < AH; a++) {
                const int t = t0 + a;                            // wave 0 walks tile t now; tile t+1 goes to the other buffer
                if (t < nT) {
                    if (t + 1 < nT) { put(nx[(a + 1) % AH], t + 1); if (t + 1 + AH < nT) fetch(nx[(a + 1) % AH], t + 1 + AH); }
                    __syncthreads();
                }
            }
        }
        return;
    }
    // wave 0: the rows' running sums
    double *raw = (NOISE ? c.barkRawN : c.barkRawP) + (size_t)row * ULCX_NBARK * 3;
    const uint32_t *sched = c.T.barkSched + (NOISE ? 0 : ULCX_MAX_SUB * ULCX_BARK_EVENTS);      // the full-size subblock's edges
    const uint32_t evLane = sched[lane < ULCX_BARK_EVENTS ? lane : ULCX_BARK_EVENTS - 1];      // edge e of the list sits in lane e
    double fl = 0.0, pk = 0.0, pw = 0.0;
    auto add_line = [&](float vf, float lf) {                    // Psyopt.c:23-51: Floor += log, Peak += log * v, PeakW += v
        const double v = (double)vf, vl = (double)lf;
        fl += vl; pk += vl * v; pw += v;
    };
    // A tile's 32 lines are straight-line code: the row's sixteen 16-byte LDS reads are issued together, then per line two
    // conversions, the product and the three ordered sums; the band edges (wave-uniform: every row has the full-size
    // geometry) are looked at in front of every line - a scalar compare when there is none.  (As a loop over "lines up to
    // the next edge" every pair of lines paid an LDS round trip and the loop's branches: 4500 cycles per tile instead of 900.)
    int e = 0;
    bool done = false;
    uint32_t ev = (uint32_t)__builtin_amdgcn_readlane((int)evLane, 0);
    auto edges_at = [&](int pos) {                               // every edge that sits in front of line `pos`
        while (!done && (int)(ev & 0xffff) == pos) {
            const int kind = (ev >> 16) & 3, b = ev >> 24;
            double *r = ring + ((b & RM) * 3) * 64;
            if (kind == 0) { r[0] = fl; r[64] = pk; r[128] = pw; }               // lower edge: snapshot
            else if (kind == 1) {                                                // upper edge: the band's three sums
                if (mine) { raw[b * 3] = fl - r[0]; raw[b * 3 + 1] = pk - r[64]; raw[b * 3 + 2] = pw - r[128]; }
            } else { done = true; break; }                                       // end of the subblock / of the list
            e++;
            if (e >= ULCX_BARK_EVENTS) { done = true; break; }
            ev = (uint32_t)__builtin_amdgcn_readlane((int)evLane, e);
        }
    };
    __syncthreads();                                             // tile 0 is in place
    for (int t = 0; t < nT; t++) {
        const float4 *mineRow = (const float4 *)(tiles + (t & 1) * BK_TILE_FLOATS + lane * BK_RS);
        constexpr int LW = BK_TL < 32 ? BK_TL : 32;              // lines per straight-line stretch
        for (int hh = 0; hh < BK_TL / LW; hh++) {                // 32 lines at a time: sixteen 16-byte reads in registers
            float4 q[LW / 2];
#pragma unroll
            for (int j = 0; j < LW / 2; j++) q[j] = mineRow[hh * (LW / 2) + j];
#pragma unroll
            for (int i = 0; i < LW; i++) {
                edges_at(t * BK_TL + hh * LW + i);
                const float4 qq = q[i >> 1];
                add_line((i & 1) ? qq.z : qq.x, (i & 1) ? qq.w : qq.y);
            }
        }
        __syncthreads();
    }
    edges_at(half);                                              // the edges behind the last line
}

// The Bark levels of the un-decimated blocks from the band sums of k_bark_uniform: one lane per (row, band), 32 lanes per
// row.  A band without energy takes the level of the last band below it that had some (Psyopt.c:118-129, :207-218: the
// level variable is simply not reassigned).
template <bool NOISE>
__global__ __launch_bounds__(WG) void k_bark_levels(UlcxEncCtx c) {
    const int nRows = NOISE ? c.B * c.K * c.C : c.B * c.K;
    const long long gid = (long long)blockIdx.x * WG + threadIdx.x;
    int row = (int)(gid >> 5);
    const int b = (int)(gid & 31), lane = threadIdx.x & 63;
    bool alive = row < nRows;
    if (!alive) row = 0;
    const int blk = NOISE ? row / c.C : row;
    alive = alive && (ulcx_pattern(c.wcArr[(size_t)(blk / c.K) * (c.maxK + 2) + (blk % c.K) + 1]) & ~8u) == 0;
    if (!__ballot(alive)) return;
    float level = 0.0f;
    bool has = false;
    if (alive && b < ULCX_NBARK) {
        const double *raw = (NOISE ? c.barkRawN : c.barkRawP) + ((size_t)row * ULCX_NBARK + b) * 3;
        double sf = raw[0], sp = raw[1], sw = raw[2];
        const int l0 = NOISE ? c.T.nBeg[0][b] : c.T.pBeg[0][b], l1 = NOISE ? c.T.nEnd[0][b] : c.T.pEnd[0][b];
        if (sw > 0.0) {
            has = true;
            if (NOISE) {                                         // Psyopt.c:207-216
                double scale = 1.0 / (double)(l1 - l0);
                sp = sp / sw;
                sf = sf * scale;
                level = 0.5f * (float)(ulcx_log(sw * scale) + sf - sp);
            } else {                                             // Psyopt.c:118-127
                sp = sp / sw;
                sf = sf / (double)(l1 - l0);
                level = (float)(sp - sf - ulcx_log(sw));
            }
        }
    }
    const unsigned hm = (unsigned)(__ballot(has) >> (lane & 32));              // this row's bands with energy
    const unsigned below = hm & (unsigned)((2ull << b) - 1);
    const int srcBand = below ? 31 - __clz(below) : b;
    const float taken = __shfl(level, (lane & 32) + srcBand);
    if (alive && b < ULCX_NBARK)
        (NOISE ? c.barkN : c.barkP)[(size_t)row * 4 * ULCX_NBARK + b] = below ? taken : (NOISE ? -100.0f : 0.0f);
}

// BlockTransform.c:337-345: key = 2*key0 + MaskingNp[n/2] + Log[0.5^2]*(Chan&1), formed where the
// keys are consumed (selection kernels) instead of being written back to HBM.
__device__ __forceinline__ float final_key(float v, float m, int ch) {
    float t = 2 * v + m;
    if (ch & 1) t = t + -0x1.62E430p0f;
    return t;
}
// BlockTransform.c:250-253: key0 = FastLog(Re^2), or -inf for a coefficient that counts as zero
__device__ __forceinline__ float key0_of(float re) {
    float k = fastlog(re * re);                            // evaluated unconditionally: a select, not a branch per coefficient
    asm volatile("" : "+v"(k));
    return (fabsf(re) < 0.5f * ULCX_COEF_EPS) ? __uint_as_float(0xff800000u) : k;
}
// The same key as key_ord(final_key(key0_of(re), m, ch)) for the wave selection (round 3: 30 -> 21 vector instructions per
// key).  2*v is exact, so fma(v, 2, m) rounds once where 2*v + m rounds once: identical.  The key is never -0.0 (a sum is -0
// only if both terms are, and ln2 * (float)e is +0 for e = 0; the channel constant is not 0), so the map needs no zero
// test: two instructions, arithmetic shift + one three-input bit operation.
__device__ __forceinline__ uint32_t sel_key(float re, float m, int ch) {
    float k = fastlog(re * re);                            // (evaluated unconditionally: a select, not a branch per coefficient)
    asm("" : "+v"(k));
    k = (fabsf(re) < 0.5f * ULCX_COEF_EPS) ? __uint_as_float(0xff800000u) : k;
    float t = __builtin_fmaf(k, 2.0f, m);
    if (ch & 1) t = t + -0x1.62E430p0f;
    const uint32_t u = __float_as_uint(t);
    return u ^ ((uint32_t)((int32_t)u >> 31) | 0x80000000u);
}
// Psyopt.c:140-150: masking level of line pair jp (0 <= jp < BS/2) of a block: interpolation between the Bark levels of
// its subblock (bark4 = the block's [4][25] levels from k_pbark).  Evaluated where the keys are formed: no array of it in HBM.
__device__ __forceinline__ float mask_level(const UlcxEncCtx &c, const float *bark4, int wc, int jp) {
    unsigned pat = ulcx_pattern(wc);
    int off = 0, d = 0, S = c.BS, j = 0;
    for (;; j++) { d = pat & 7; S = c.BS >> d; if (2 * jp < off + S) break; off += S; pat >>= 4; }
    const int line = jp - off / 2;
    const float *bark = bark4 + j * ULCX_NBARK;
    const int bi = c.T.bandIdx[d][line];
    const float fr = c.T.bandFrac[d][line];
    const float L = (bi < ULCX_NBARK) ? bark[bi] : bark[ULCX_NBARK - 1];
    const float R = (bi + 1 < ULCX_NBARK) ? bark[bi + 1] : L;
    return L * (1.0f - fr) + R * fr;
}
// key of coefficient i of block blk, from the stored coefficient and the masking level of its line; once
// k_keys_finalize has run for the call (c.keyFinal: the multi-pass selection kernel of unusual geometries, the parity
// tap) c.key holds the same values
__device__ __forceinline__ float load_final_key(const UlcxEncCtx &c, int blk, int i) {
    if (c.keyFinal) return c.key[(size_t)blk * (c.C * c.BS) + i];
    int ch = i >> c.lgBS, n = i & (c.BS - 1);
    const int wcB = c.wcArr[(size_t)(blk / c.K) * (c.maxK + 2) + (blk % c.K) + 1];
    return final_key(key0_of(c.coef[(size_t)blk * (c.C * c.BS) + i]), mask_level(c, c.barkP + (size_t)blk * 4 * ULCX_NBARK, wcB, n >> 1), ch);
}

// debug/parity tap only: materialise the final keys in c.key (ulcx_encoder_debug_fetch)
__global__ __launch_bounds__(WG) void k_keys_finalize(UlcxEncCtx c) {
    size_t gid = (size_t)blockIdx.x * WG + threadIdx.x;
    size_t N = (size_t)c.C * c.BS;
    if (gid >= (size_t)c.B * c.K * N) return;
    int blk = (int)(gid / N), i = (int)(gid % N);
    c.keyFinal = 0;                                       // (this is the kernel that forms them)
    c.key[gid] = load_final_key(c, blk, i);
}
void ulcx_enc_finalize_keys(const UlcxEncCtx &c, hipStream_t st) {
    size_t tot = (size_t)c.B * c.K * c.C * c.BS;
    hipLaunchKernelGGL(k_keys_finalize, dim3((unsigned)((tot + WG - 1) / WG)), dim3(WG), 0, st, c);
}

// Which blocks a launch of the select/encode kernels works on:
//   probe passes skip blocks whose rate search has converged; fbMode 1 skips blocks that left the lock-step
//   path because a threshold tie group straddled the cut (c.isFb, set once per call); fbMode 2 processes
//   only those, restricted to the slots [fbLo, fbHi) of the fallback list whose ranks are resident.
// lock-step probe pass with no open rate search left (the number of passes is fixed on the host; the blocks decide how many do work)
__device__ __forceinline__ bool probes_over(const UlcxEncCtx &c, int finalPass) { return !finalPass && c.fbMode != 2 && *c.cbrLive <= 0; }
__device__ __forceinline__ bool skip_block(const UlcxEncCtx &c, int blk, int finalPass) {
    if (!finalPass && c.cbrDone[blk]) return true;
    if (c.fbMode == 1) return c.isFb[blk] != 0;
    return false;                                            // fbMode 2: the launch enumerates the owned blocks itself (fb_count / fbList)
}
// fbMode 2 launches are small fixed grids that walk the exact-path list: n = resident entries of it
__device__ __forceinline__ int fb_count(const UlcxEncCtx &c) {
    int n = *c.fbCount; if (n > c.fbHi) n = c.fbHi;
    n -= c.fbLo; return n > 0 ? n : 0;
}

// ---------------------------------------------------------------------------
// Selection of the nOutCoef most important coefficients.
// The reference heapsorts all keys into ranks (BlockTransform.c:20-77) but ranks are
// only ever consumed as "rank < nOutCoef" (Encode.c:108,220), so the sort is a
// selection: find the k-th largest key T by a 4x8-bit radix select in LDS; the kept
// set is {key > T} plus the tie group {key == T} when it fits entirely.  Only when the
// tie group straddles the cut is the exact heapsort pop order needed (k_heapsel).
// ---------------------------------------------------------------------------
// Wave-wide reductions on the VALU's data-parallel primitives (row shifts inside rows of 16 lanes, then the two row
// broadcasts of gfx9): six instructions and one v_readlane, no LDS round trips.  Every lane gets the result.
#define ULCX_DPP_STEPS(OP) \
    OP(0x111, 0xf) OP(0x112, 0xf) OP(0x114, 0xf) OP(0x118, 0xf) OP(0x142, 0xa) OP(0x143, 0xc)
__device__ __forceinline__ int wave_sum_i32(int v) {
#define STEP(ctl, rmask) v += __builtin_amdgcn_update_dpp(0, v, ctl, rmask, 0xf, false);
    ULCX_DPP_STEPS(STEP)
#undef STEP
    return __builtin_amdgcn_readlane(v, 63);
}
__device__ __forceinline__ uint32_t wave_min_u32(uint32_t v) {
#define STEP(ctl, rmask) { uint32_t o = (uint32_t)__builtin_amdgcn_update_dpp((int)0xFFFFFFFFu, (int)v, ctl, rmask, 0xf, false); v = o < v ? o : v; }
    ULCX_DPP_STEPS(STEP)
#undef STEP
    return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
}
__device__ __forceinline__ uint32_t wave_max_u32(uint32_t v) {
#define STEP(ctl, rmask) { uint32_t o = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, ctl, rmask, 0xf, false); v = o > v ? o : v; }
    ULCX_DPP_STEPS(STEP)
#undef STEP
    return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
}
__device__ __forceinline__ uint32_t key_ord(float f) {          // ascending order-preserving map
    uint32_t u = __float_as_uint(f);
    if ((u << 1) == 0) u = 0;                                   // -0 and +0 compare equal in the reference
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

__global__ __launch_bounds__(WG) void k_select(UlcxEncCtx c, int finalPass) {
    if (probes_over(c, finalPass)) return;
    __shared__ int hist[256];
    __shared__ uint32_t s_prefix;
    __shared__ int s_need;
    int blk = blockIdx.x, tid = threadIdx.x;
    if (!finalPass && c.cbrDone[blk]) return;        // rate search already converged: wait for the final pass
    if (c.isFb[blk]) return;
    int N = c.C * c.BS;
    int kSel = c.nout[blk];
    uint32_t *keep = c.keep + (size_t)blk * (N / 32);
    if (kSel <= 0) {
        for (int i = tid; i < N / 32; i += WG) keep[i] = 0;
        return;
    }
    uint32_t prefix = 0, pmask = 0;
    int need = kSel;                     // how many still to take from the current candidate set
    for (int pass = 0; pass < 4; pass++) {
        int shift = 24 - 8 * pass;
        hist[tid] = 0;                   // WG == 256 bins
        __syncthreads();
        for (int i = tid; i < N; i += WG) {
            uint32_t u = key_ord(load_final_key(c, blk, i));
            if ((u & pmask) == prefix) atomicAdd(&hist[(u >> shift) & 255], 1);
        }
        __syncthreads();
        if (tid == 0) {
            int acc = 0, dgt = 255;
            for (; dgt > 0; dgt--) { if (acc + hist[dgt] >= need) break; acc += hist[dgt]; }
            s_prefix = prefix | ((uint32_t)dgt << shift);
            s_need = need - acc;
        }
        __syncthreads();
        prefix = s_prefix; need = s_need;
        pmask |= 0xFFu << shift;
        __syncthreads();
    }
    // prefix = ordered bits of threshold T; need = r (how many of the T-ties are kept); e = hist count
    int e = hist[prefix & 255];
    bool straddle = (need < e) || (c.forceFb > 0 && blk % c.forceFb == 0);
    for (int i = tid; i < N; i += WG) {
        uint32_t u = key_ord(load_final_key(c, blk, i));
        bool kp = (u >= prefix);          // tie group fully in when not straddling
        unsigned long long m = __ballot(kp);
        int lane = tid & 63;
        if (lane == 0)  keep[i >> 5] = (uint32_t)m;
        if (lane == 32) keep[i >> 5] = (uint32_t)(m >> 32);
    }
    if (straddle && tid == 0) {
        int slot = atomicAdd(c.fbCount, 1);
        c.fbList[slot] = blk;
        c.ownSlot[blk] = slot;
        c.isFb[blk] = 1;
        if (!finalPass) atomicSub(c.cbrLive, 1);              // (the exact path finishes its search on its own)
    }
}

// The 64-bit ballot of key register I into lane I of (klo, khi): v_writelane_b32 with an immediate lane.  (No builtin for it in
// this compiler; the s_nop covers the two wait states gfx940+ wants between a vector compare's scalar result and a
// vector instruction that reads it - the hazard recogniser does not look inside inline assembly.)
template <int L> __device__ __forceinline__ void writelane2_imm(uint32_t &lo, uint32_t &hi, unsigned long long m) {
    asm("s_nop 1\n\tv_writelane_b32 %0, %2, %4\n\tv_writelane_b32 %1, %3, %4" : "+v"(lo), "+v"(hi) : "s"((uint32_t)m), "s"((uint32_t)(m >> 32)), "n"(L));
}
template <int R, int... I>
__device__ __forceinline__ void sel_gather_keep(const uint32_t (&u)[R], uint32_t T, uint32_t &klo, uint32_t &khi, std::integer_sequence<int, I...>) {
    ((void)[&] { writelane2_imm<I>(klo, khi, __ballot(u[I] >= T)); }(), ...);
}
// One WAVE per block, keys held in registers (R = N/64 per lane): no workgroup barriers,
// the 256-bin histogram of each radix pass lives in a private 1 KB LDS slice.
// PASS: 0 = one-pass call (VBR); rate search: 1 = first probe (leaves the ordered keys in c.key), 2 = later probes and the
// final pass (read them back, search the window the earlier probes left)
// PAIR: TWO waves per block, one per channel of a stereo block (R = BlockSize/64 keys per lane each), one block per
// workgroup: every count, minimum and decision of the search is formed over both waves through two words of LDS and a
// workgroup barrier (the two waves take every branch together).  BlockSize 4096 stereo: 128 keys per lane in one wave are
// 200 registers, two waves per SIMD.
template <int R, int LGBS, int PASS, bool PAIR>          // LGBS: log2(BlockSize) as a compile-time constant (0: read from the context)
__device__ __forceinline__ void select_body(const UlcxEncCtx &c, int finalPass, int blk, int wv, int lane, int half, volatile uint32_t *xch, float *sel_lds) {
    // sums / minima / maxima over the pair's two waves (wave-uniform values; the exchanges alternate between two slots, so
    // one barrier per exchange is enough)
    int xt = 0;
    auto xchg = [&](uint32_t v) -> uint32_t {
        if (lane == 0) xch[(xt & 1) * 2 + half] = v;
        __syncthreads();
        const uint32_t o = xch[(xt & 1) * 2 + (1 - half)];
        xt++;
        return o;
    };
    auto pair_sum = [&](int v) -> int { if (!PAIR) return v; return v + (int)xchg((uint32_t)v); };
    auto pair_min = [&](uint32_t v) -> uint32_t { if (!PAIR) return v; const uint32_t o = xchg(v); return o < v ? o : v; };
    auto pair_max = [&](uint32_t v) -> uint32_t { if (!PAIR) return v; const uint32_t o = xchg(v); return o > v ? o : v; };
    // (the first probe of a rate search forms and stores the keys of EVERY block: one whose search is over before it starts,
    //  or that keeps nothing in this probe, still needs them in a later pass)
    const bool idle = !finalPass && c.cbrDone[blk];
    if (PASS != 1 && idle) return;
    if (c.isFb[blk]) return;                              // already handed to the exact (heapsort-rank) path this call
    constexpr int NW = R * 64, N = PAIR ? 2 * NW : NW;      // this wave's keys, the block's
    int kSel = c.nout[blk];
    const float *coef = c.coef + (size_t)blk * N + (size_t)half * NW;
    uint32_t *keep = c.keep + (size_t)blk * (N / 32) + half * (NW / 32);
    if (kSel <= 0 && !idle) {
        for (int i = lane; i < NW / 32; i += 64) keep[i] = 0;
    }
    if (PASS != 1 && kSel <= 0) return;
    // sel_lds, per wave (PAIR: per block): BS/2 masking levels + the block's 4 x 25 Bark levels; later the candidate lists
    const int selStride = ulcx_sel_lds_words(c.BS);
    uint32_t u[R];
    const int lgK = LGBS ? LGBS : c.lgBS, bsK = LGBS ? (1 << LGBS) : c.BS;   // (constants: channel and LDS offsets of a key fold per register)
    {
        // the block's masking level per line (Psyopt.c:140-150), formed by the wave into LDS (BS/2 <= 32 R values) instead of
        // being read from an array another kernel wrote
        float *msk = sel_lds + wv * selStride;
        float *sbarkw = msk + c.BS / 2;
        if constexpr (PASS != 2) {
            for (int i = lane; i < 4 * ULCX_NBARK; i += 64) sbarkw[i] = c.barkP[(size_t)blk * 4 * ULCX_NBARK + i];
            const int wcB = c.wcArr[(size_t)(blk / c.K) * (c.maxK + 2) + (blk % c.K) + 1];
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            if (PAIR) __syncthreads();                          // (both waves have stored the same Bark levels)
            for (int jp = lane + (PAIR ? 64 * half : 0); jp < c.BS / 2; jp += (PAIR ? 128 : 64)) msk[jp] = mask_level(c, sbarkw, wcB, jp);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            if (PAIR) __syncthreads();
        }
        // batches of 8: the loads of one batch are in flight together, but the compiler may not hoist all R of them
        // above the arithmetic (that doubled the register count and halved the occupancy)
        if constexpr (PASS != 2)
#pragma unroll
        for (int r0 = 0; r0 < R; r0 += 8) {
            float cv[8], mv[8];
#pragma unroll
            for (int q = 0; q < 8 && r0 + q < R; q++) { int i = (r0 + q) * 64 + lane; cv[q] = ldnt(coef + i); mv[q] = msk[(i & (bsK - 1)) >> 1]; }
#pragma unroll
            for (int q = 0; q < 8 && r0 + q < R; q++) { int i = (r0 + q) * 64 + lane; u[r0 + q] = sel_key(cv[q], mv[q], PAIR ? half : (i >> lgK)); }
            if constexpr (PASS == 1) {
#pragma unroll
                for (int q = 0; q < 8 && r0 + q < R; q++) ((uint32_t *)c.key + (size_t)blk * N + (size_t)half * NW)[(r0 + q) * 64 + lane] = u[r0 + q];
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    if (PASS == 1 && (idle || kSel <= 0)) return;         // (keys stored above)
    // Rate search (CBR / ABR): the ordered keys of a block are the same in all its probes.  The first pass leaves them in
    // c.key (unused otherwise while the keys are formed on the fly), the later ones read them back.
    uint32_t *ukeys = (uint32_t *)c.key + (size_t)blk * N + (size_t)half * NW;
    if constexpr (PASS == 2) {
#pragma unroll
        for (int r = 0; r < R; r++) u[r] = ukeys[r * 64 + lane];
    }
    // T = kSel-th largest ordered key = the largest t with count(u >= t) >= kSel, found bit by bit - but not every bit on
    // all R keys per lane (a histogram radix select serialises on LDS atomics here: log-domain keys share their top byte):
    //  1. wave minimum and maximum: T shares their common leading bits, the probes start at the first bit that differs
    //     (log-domain keys share 6-9 leading bits: that many full probes less);
    //  2. full probes (R compares per lane) only until the window [T, T + 2^(bit+1)) that still holds T has few keys in it
    //     (count above T minus count above the window's top, both known from the probes);
    //  3. those candidates go to a few registers per lane through LDS (a lane's own list; a lane with more than SEL_CAP of
    //     them, or a window that never gets small - ties, silence -, keeps the full probes) and the remaining bits are
    //     resolved on SEL_CAP compares per lane.
    // A probe that separates exactly kSel keys ends the search at once (the answer is the smallest key above it).
    constexpr int SEL_CAP = ULCX_SEL_CAP, SEL_CAND = ULCX_SEL_CAND, SEL_WIN = ULCX_SEL_CAND + ULCX_SEL_CAND / 4;     // (two candidates per lane on average: more, and some lane of 64 has more than SEL_CAP)
    constexpr bool SEL_COMPACT = R > 2 * SEL_CAP;         // (few keys per lane: the full probes are as cheap)
    uint32_t T = 0;
    int cntT = N;                                         // keys >= T (the search keeps it: no counting pass at the end)
    {
        int cntLo = N, cntHi = 0;                         // keys >= T, keys >= T + 2^(bit+1)
        bool compacted = false, tried = false;
        uint32_t cd[SEL_CAP];
        // Rate search: the earlier probes of this block have left a window [TL, TH) of keys with count(u >= TL) = cL and
        // count(u >= TH) = cH known (pack_block), and every later threshold lies in it or is TH itself (the probe that set TH,
        // once more: the final pass): when few keys are left in the window they go to the candidate registers at once and
        // the search runs on them alone, above cH.
        bool same = false;
        if constexpr (SEL_COMPACT && PASS == 2) {
            const uint4 w = c.selWin[blk];
            same = w.y != 0u && kSel == (int)w.w;
            if (!same && (int)w.z - (int)w.w <= SEL_WIN) {
                uint32_t *cl = (uint32_t *)(sel_lds + wv * selStride) + half * (SEL_CAP * 64);
                const uint32_t span = w.y - w.x;          // (TH = 0: no upper bound yet; the subtraction wraps to 2^32 - TL)
                int nL = 0;
#pragma unroll
                for (int r = 0; r < R; r++) {
                    const bool act = (u[r] - w.x) < span;
                    if (act && nL < SEL_CAP) cl[nL * 64 + lane] = u[r];
                    nL += act ? 1 : 0;
                }
                if (pair_sum(__any(nL > SEL_CAP) ? 1 : 0) == 0) {
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
                    for (int j = 0; j < SEL_CAP; j++) cd[j] = (j < nL) ? cl[j * 64 + lane] : 0u;
                    compacted = (int)w.z > (int)w.w; tried = compacted;          // (an empty window cannot happen; the full search is right whatever the window says)
                    if (compacted) { cntLo = (int)w.z; cntHi = (int)w.w; }
                }
            }
        }
        uint32_t mn = 0xFFFFFFFFu, mx = 0u;
        if (!compacted) {
#pragma unroll
            for (int r = 0; r < R; r++) { mn = u[r] < mn ? u[r] : mn; mx = u[r] > mx ? u[r] : mx; }
        } else {
#pragma unroll
            for (int j = 0; j < SEL_CAP; j++) { const uint32_t v = cd[j] ? cd[j] : 0xFFFFFFFFu; mn = v < mn ? v : mn; mx = cd[j] > mx ? cd[j] : mx; }
        }
        mn = pair_min(wave_min_u32(mn)); mx = pair_max(wave_max_u32(mx));
        const uint32_t dif = mn ^ mx;
        if (ULCX_DBG(c) & 0x1000) T = mn;                  // (ablation build only: no search, everything is kept)
        else if (same) { T = c.selWin[blk].y; cntT = kSel; }
        else if (dif == 0) { T = mn; cntT = cntLo; }
        else {
            int bit = 31 - __clz(dif);
            T = mx & ~((2u << bit) - 1u);                 // the common prefix (count(u >= T) = cntLo >= kSel)
            for (; bit >= 0; bit--) {
                const uint32_t t = T | (1u << bit);
                // (counted on the scalar side: a compare into a lane mask, s_bcnt1, s_add - one vector instruction per key
                //  instead of two and a wait state, no reduction across the wave at the end)
                int cnt = 0;
                if (!compacted) {
#pragma unroll
                    for (int r = 0; r < R; r++) cnt += __popcll(__ballot(u[r] >= t));
                    cnt = pair_sum(cnt);
                } else {
#pragma unroll
                    for (int j = 0; j < SEL_CAP; j++) cnt += __popcll(__ballot(cd[j] >= t));
                    cnt = pair_sum(cnt) + cntHi;
                }
                if (cnt == kSel) {
                    // t falls between the kSel-th and the next key: the answer is the smallest key >= t, no need to
                    // resolve the remaining bits (typically half of them)
                    uint32_t m2 = 0xFFFFFFFFu;
                    if (!compacted) {
#pragma unroll
                        for (int r = 0; r < R; r++) { uint32_t v = (u[r] >= t) ? u[r] : 0xFFFFFFFFu; m2 = v < m2 ? v : m2; }
                    } else {
#pragma unroll
                        for (int j = 0; j < SEL_CAP; j++) { uint32_t v = (cd[j] >= t) ? cd[j] : 0xFFFFFFFFu; m2 = v < m2 ? v : m2; }
                    }
                    T = pair_min(wave_min_u32(m2)); cntLo = kSel;
                    break;
                }
                if (cnt > kSel) { T = t; cntLo = cnt; } else if (!compacted) cntHi = cnt;
                if (SEL_COMPACT && !compacted && !tried && bit > 0 && cntLo - cntHi <= SEL_CAND) {
                    // candidates: T <= u < T + 2^bit (the window the next probe halves)
                    tried = true;
                    if (PAIR) __syncthreads();                                    // (the masking levels are used up - by both waves)
                    uint32_t *cl = (uint32_t *)(sel_lds + wv * selStride) + half * (SEL_CAP * 64);
                    const uint32_t W = 1u << bit;
                    int nL = 0;
#pragma unroll
                    for (int r = 0; r < R; r++) {
                        const bool act = (u[r] - T) < W;
                        if (act && nL < SEL_CAP) cl[nL * 64 + lane] = u[r];
                        nL += act ? 1 : 0;
                    }
                    if (pair_sum(__any(nL > SEL_CAP) ? 1 : 0) == 0) {
                        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
                        for (int j = 0; j < SEL_CAP; j++) cd[j] = (j < nL) ? cl[j * 64 + lane] : 0u;       // (0 is below every probe)
                        compacted = true;
                    }
                }
            }
            cntT = cntLo;
        }
    }
    if constexpr (PASS != 0) if (half == 0) c.selT[blk] = T;                       // (uniform store: the threshold of this probe, for the window update)
    // the tie group at T straddles the cut iff more than kSel keys are >= T (kSel - #(u > T) < #(u == T))
    const bool straddle = kSel < cntT || (c.forceFb > 0 && blk % c.forceFb == 0);
    // keep bitmap: the ballot of register r is the pair of words 2r, 2r+1 - gathered into lane r (R <= 64) or lanes r, r - 64
    // and stored once per lane instead of twice per register
    if constexpr (R <= 64) {
        uint32_t klo = 0, khi = 0;
        sel_gather_keep(u, T, klo, khi, std::make_integer_sequence<int, R>());
        if (lane < R) *(uint2 *)(keep + 2 * lane) = make_uint2(klo, khi);
    } else {
#pragma unroll
        for (int r = 0; r < R; r++) {
            unsigned long long m = __ballot(u[r] >= T);
            if (lane == 0)  keep[2 * r] = (uint32_t)m;
            if (lane == 32) keep[2 * r + 1] = (uint32_t)(m >> 32);
        }
    }
    if (straddle && lane == 0 && half == 0) {
        int slot = atomicAdd(c.fbCount, 1);
        c.fbList[slot] = blk;
        c.ownSlot[blk] = slot;
        c.isFb[blk] = 1;
        if (!finalPass) atomicSub(c.cbrLive, 1);              // (the exact path finishes its search on its own)
    }
}


// One WAVE per block, four blocks per workgroup
template <int R, int LGBS = 0, int PASS = 0>
__global__ __launch_bounds__(256) void k_select_wave(UlcxEncCtx c, int finalPass) {
    if (probes_over(c, finalPass)) return;
    extern __shared__ float sel_lds[];
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int blk = blockIdx.x * 4 + wv;                  // 4 waves per workgroup, one block per wave
    if (blk >= c.B * c.K) return;
    select_body<R, LGBS, PASS, false>(c, finalPass, blk, wv, lane, 0, nullptr, sel_lds);
}
// Two waves per block (stereo: a wave per channel), one block per workgroup
template <int R, int LGBS = 0, int PASS = 0>
__global__ __launch_bounds__(128) void k_select_pair(UlcxEncCtx c, int finalPass) {
    extern __shared__ float sel_lds[];
    __shared__ uint32_t xch[4];
    __shared__ int over;
    // (the count of open searches is read ONCE per workgroup: other blocks' waves count it down while this kernel runs, and
    //  the two waves of a block must not disagree on whether to go on)
    if (threadIdx.x == 0) over = probes_over(c, finalPass) ? 1 : 0;
    __syncthreads();
    if (over) return;
    const int half = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    select_body<R, LGBS, PASS, true>(c, finalPass, blockIdx.x, 0, lane, half, xch, sel_lds);
}

// Exact emulation of the reference's min-heap heapsort for the (rare) blocks whose
// threshold tie group straddles the cut: ranks are assigned N-1 downwards in pop
// order, so the kept set is everything still in the heap after N-k pops.
// One lane per block; heap of {key, index} in LDS when it fits, else in HBM scratch.
struct HeapEnt { float v; int i; };
template <typename P>
__device__ void heap_sift(P h, int root, int n) {
    int child = 2 * root + 1;
    if (child >= n) return;
    HeapEnt r = h[root];
    for (;;) {
        HeapEnt cN = h[child];
        if (child + 1 < n) { HeapEnt c2 = h[child + 1]; if (c2.v < cN.v) { cN = c2; child++; } }
        if (cN.v > r.v) break;
        h[root] = cN;
        root = child; child = 2 * root + 1;
        if (child >= n) break;
    }
    h[root] = r;
}
__global__ __launch_bounds__(64) void k_heapsel(UlcxEncCtx c, int ldsEntries) {
    extern __shared__ HeapEnt hl[];
    int count = *c.fbCount; if (count > c.fbHi) count = c.fbHi;
    int N = c.C * c.BS;
    bool useLds = (N <= ldsEntries);
    HeapEnt *h = useLds ? hl : (HeapEnt *)c.heapScratch + (size_t)blockIdx.x * N;
    for (int idx = c.fbLo + blockIdx.x; idx < count; idx += gridDim.x) {
        int blk = c.fbList[idx];
        int *rank = c.rankBuf + (size_t)(idx - c.fbLo) * N;
        for (int i = threadIdx.x; i < N; i += 64) { h[i].v = load_final_key(c, blk, i); h[i].i = i; }
        __syncthreads();
        if (threadIdx.x == 0) {
            for (int n = N / 2 - 1; n >= 0; n--) heap_sift(h, n, N);
            for (int n = N - 1; n > 0; n--) {             // BlockTransform.c:66-76: ranks N-1 .. 1 in pop order
                rank[h[0].i] = n;
                h[0] = h[n]; heap_sift(h, 0, n);
            }
            rank[h[0].i] = 0;
        }
        __syncthreads();
    }
}

// Pipelined replay of the same heapsort, one wave per block, heap in LDS.
//  * heapify: the reference sifts nodes N/2-1 .. 0; nodes of one tree level have disjoint
//    subtrees, so a level is sifted in parallel (one lane per node), levels bottom-up.
//  * pops: pop p moves the last heap element to the root and sifts it down.  A sift at level
//    l only touches levels >= l, so pop p+1 may start once pop p is two levels down: up to
//    ~6 pops are in flight, one lane each, every step advancing each by one level.  The only
//    cross-pop hazard is the element pop p+1 lifts from the end of the heap: if an in-flight
//    sift is still on the path to that leaf it could yet replace it, so the start waits.
// Comparisons and tie behaviour are exactly those of heap_sift / BlockTransform.c:20-51.
__global__ __launch_bounds__(64) void k_heapsel_pipe(UlcxEncCtx c, int fullRanking) {
    extern __shared__ __align__(16) int4 hraw4[];
    HeapEnt *hp = (HeapEnt *)hraw4 + 1;                       // node n at slot n+1: a node's two children share one aligned 16 B pair
    int2 *slots = (int2 *)hraw4;
    const int4 *pairs = hraw4;                                // pairs[pos+1] = {child 2pos+1, child 2pos+2}
    int count = *c.fbCount; if (count > c.fbHi) count = c.fbHi;
    const int N = c.C * c.BS;
    int lane = threadIdx.x;
    for (int idx = c.fbLo + blockIdx.x; idx < count; idx += gridDim.x) {
        int blk = c.fbList[idx];
        int *rank = c.rankBuf + (size_t)(idx - c.fbLo) * N;      // full ranking, so every later nOutCoef of this block is a lookup
        for (int i = lane; i < N; i += 64) { hp[i].v = load_final_key(c, blk, i); hp[i].i = i; rank[i] = 0; }
        if (lane == 0) { hp[N].v = 0.0f; hp[N].i = 0; }
        __syncthreads();
        // ---- heapify, level by level
        int top = 31 - __clz(N / 2);                      // level of node N/2-1 (root = level 0) for power-of-two N
        for (int L = top; L >= 0; L--) {
            int first = (1 << L) - 1, last = (2 << L) - 2;
            if (last > N / 2 - 1) last = N / 2 - 1;
            for (int n = first + lane; n <= last; n += 64) heap_sift(hp, n, N);
            __syncthreads();
        }
        // ---- pipelined pops.  A step = one LDS round trip: every in-flight sift reads its child pair
        //      and moves one level.  Pop P starts (takes the root's rank, lifts the last leaf into a
        //      register) in the first half of an iteration and does its level-l compare l+1 steps later;
        //      one iteration = two steps, so the next pop finds the root already rewritten.
        //      An idle lane has pos = N+1: no children, and its store lands in a dummy slot.
        //      A one-pass (VBR) call only needs the kept set: stop after N-nOutCoef pops; rank 0 (also
        //      the last pop's rank) is what rankBuf was initialised to.
        int kSel = c.nout[blk];
        int pops = fullRanking ? N : N - (kSel > 0 ? kSel : 0);
        if (pops > N - 1) pops = N - 1;
        const int IDLE = N + 1;
        int pos = IDLE, lev = 0, size = 0; float ev = 0.0f; int ei = 0;
#define HEAP_SIFT_STEP()                                                                          \
        {                                                                                         \
            int c1 = 2 * pos + 1;                                                                 \
            int pi = pos + 1 < N / 2 ? pos + 1 : N / 2;                                           \
            int4 ch = pairs[pi];                                                                  \
            float vL = __int_as_float(ch.x), vR = __int_as_float(ch.z);                           \
            bool pickR = (c1 + 1 < size) && (vR < vL);                                            \
            float vN = pickR ? vR : vL; int iN = pickR ? ch.w : ch.y;                             \
            bool stop = !(c1 < size) || (vN > ev);                                                \
            slots[pos + 1] = make_int2(__float_as_int(stop ? ev : vN), stop ? ei : iN);           \
            pos = stop ? IDLE : c1 + (pickR ? 1 : 0);                                             \
            lev = stop ? 0 : lev + 1;                                                             \
        }
        for (int P = 0; P < pops; ) {
            int nl = N - 1 - P;                           // index of the element to lift = heap size after this pop (>= 1)
            int b1 = nl + 1, db = 31 - __clz(b1);
            // an in-flight sift still above the leaf on its root path could yet replace it: wait
            int sh = db - lev; sh = sh > 0 ? sh : 0;
            bool start = !__ballot((b1 >> sh) == pos + 1);
            int2 g = slots[1], el = slots[nl + 1];
            HEAP_SIFT_STEP();
            if (start) {
                if (lane == (P & 15)) {                   // <= 7 sifts in flight, each <= 14 steps: slot P&15 is idle again
                    rank[g.y] = nl;                       // pop p gets rank N-1-p (BlockTransform.c:66-76)
                    ev = __int_as_float(el.x); ei = el.y; size = nl; pos = 0; lev = 0;
                }
                P++;
            }
            HEAP_SIFT_STEP();
        }
        while (__ballot(pos != IDLE)) HEAP_SIFT_STEP();
#undef HEAP_SIFT_STEP
        __syncthreads();
        if (!fullRanking) {
            // one-pass call: the kept set straight from here (what k_keep_ranks would do in a launch of its own behind this
            // kernel - the end of the call waits for this chain)
            __threadfence_block();
            uint32_t *keep = c.keep + (size_t)blk * (N / 32);
            if (lane == 0) c.slow[blk] = 0;
            for (int i = lane; i < N; i += 64) {
                unsigned long long m = __ballot(rank[i] < kSel);
                if (lane == 0)  keep[i >> 5] = (uint32_t)m;
                if (lane == 32) keep[i >> 5] = (uint32_t)(m >> 32);
            }
        }
    }
}

// kept set of the exact-path blocks from their ranking: rank < nOutCoef (Encode.c:108,220)
__global__ __launch_bounds__(WG) void k_keep_ranks(UlcxEncCtx c, int finalPass) {
    int count = *c.fbCount; if (count > c.fbHi) count = c.fbHi;
    const int N = c.C * c.BS;
    for (int idx = c.fbLo + blockIdx.x; idx < count; idx += gridDim.x) {
        int blk = c.fbList[idx];
        if (!finalPass && c.cbrDone[blk]) continue;
        const int *rank = c.rankBuf + (size_t)(idx - c.fbLo) * N;
        uint32_t *keep = c.keep + (size_t)blk * (N / 32);
        if (threadIdx.x == 0) c.slow[blk] = 0;             // wave-encoder give-up bits of this pass (the main path clears its own)
        int kSel = c.nout[blk];
        for (int i = threadIdx.x; i < N; i += WG) {
            unsigned long long m = __ballot(rank[i] < kSel);
            int lane = threadIdx.x & 63;
            if (lane == 0)  keep[i >> 5] = (uint32_t)m;
            if (lane == 32) keep[i >> 5] = (uint32_t)(m >> 32);
        }
    }
}

// ---------------------------------------------------------------------------
// Speculative, massively parallel evaluation of the ordered f32 sums the bitstream state
// machine needs (NoiseFill.c:15-36, :41-62): for every kept coefficient the noise-run sums
// of the gap in front of it, assuming the gap starts right after the previous kept
// coefficient; for every unit the five HF-extension sums of the tail after its last kept
// coefficient.  Same loops, same order as get_noise_q/get_hfext, so the values are the ones
// the serial kernel would compute; it checks the assumption and recomputes if it is off
// (a kept coefficient collapsed, a noise run fell back to a zero run, ...).
// Round 4: no 16 KB-per-block pair array in HBM between them any more (was k_nline -> k_gapsums, k_tailsums).  k_nsums: a
// workgroup forms its block's {w, w*log} pairs from the 100 Bark levels per channel straight into LDS (noise_pair), lists
// the gaps and sums them.  k_tails: the units' tail chains, 64 units per workgroup (a workgroup of k_nsums that also ran
// its block's two 700-step chains lived 18 us for them).  A gap
// longer than one noise run (16 + 511 coefficients) gets EVERY further run speculated too: where run r starts follows from the
// gap's length alone as long as all runs before it are coded as noise, and its sums go to component r & 1 of
// gapSum[i - (r >> 1)] (i = the kept coefficient behind the gap; those positions lie inside the gap).  On the bench batch: 0.30
// second runs and 0.02 third runs per block.  The writer chains through them (gap_codes, write_zone) exactly as far as the
// lister listed them - a next run exists iff >= 16 zeros are left behind an all-noise prefix - and sums a run that sits
// elsewhere (a run before it fell back to zeros: ~0.0007 per block) itself, forming the pairs it needs (pair_demand).
// (tests/test_gpu_parity.py::test_long_zero_gaps_with_several_noise_runs: gaps of thousands of zeros, both writers.)
// ---------------------------------------------------------------------------
#define E_GAPCAP(N) ((N) / 16)      // gaps >= 16 per block: at most N/17 of them
#define WAVE_SYNC_E() do { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); } while (0)
#ifndef NSUMS_LB
#define NSUMS_LB 8                                         // 64 registers: 63 used, no spill; 7 workgroups per CU by LDS
#endif
__host__ __device__ static inline size_t nsums_lds_bytes(int N, int C) {
    return (size_t)N * 4 + N / 8 + 8 * (size_t)E_GAPCAP(N) + 16 + 8 * (size_t)((N / 32 + 63) / 64) + (size_t)C * 4 * ULCX_NBARK * 4 + 32 * 8 + N / 16;
}
// ordered {Sum, SumW} of the pairs that cover coefficients [start, start + n) (NoiseFill.c:24-28), finished to the
// amplitude the writer quantises (:29-30; -1: "Sum == 0")
__device__ __forceinline__ float run_amplitude(const float *pairs, int start, int n) {
    const float2 *d = (const float2 *)(pairs + (start / 2) * 2);
    const int np = (n + (start & 1) + 1) / 2;
    float sum = 0.0f, sumw = 0.0f;
    int q = 0;
    for (; q + 8 <= np; q += 8) {
        float2 p[8];
#pragma unroll
        for (int u = 0; u < 8; u++) p[u] = d[q + u];
#pragma unroll
        for (int u = 0; u < 8; u++) { sum += p[u].y; sumw += p[u].x; }
    }
    for (; q < np; q++) { float2 p = d[q]; sum += p.y; sumw += p.x; }
    return (sum == 0.0f) ? -1.0f : ulcx_expf(sum / sumw);
}
// what a workgroup fetches for its NEXT block while it works on the current one (a block's first instructions used to be
// three dependent trips to HBM - window code, keep words, Bark levels - with 6 workgroups a CU to hide them behind: the
// list phase alone took 0.86 ms of the kernel's 1.35)
#define NS_KW 2                                            // keep words per thread: N / 32 <= 512
#define NS_BK 7                                            // Bark levels per thread: C * 100 <= 1600
struct NsPre { uint32_t kw[NS_KW]; float bk[NS_BK]; int wc; };
__device__ __forceinline__ void nsums_fetch(const UlcxEncCtx &c, int blk, NsPre &p) {
    const int tid = threadIdx.x, nW = c.C * c.BS / 32, nLev = c.C * 4 * ULCX_NBARK;
    const uint32_t *keepB = c.keep + (size_t)blk * nW;
    const float *bg = c.barkN + (size_t)blk * nLev;
#pragma unroll
    for (int i = 0; i < NS_KW; i++) p.kw[i] = (tid + i * WG < nW) ? keepB[tid + i * WG] : 0u;
#pragma unroll
    for (int i = 0; i < NS_BK; i++) p.bk[i] = (tid + i * WG < nLev) ? bg[tid + i * WG] : 0.0f;
    p.wc = c.wcArr[(size_t)(blk / c.K) * (c.maxK + 2) + (blk % c.K) + 1];
}
__device__ void nsums_block(const UlcxEncCtx &c, int blk, const NsPre &pre) {
    extern __shared__ uint32_t gsm[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int N = c.C * c.BS, nW = N / 32, half = c.BS / 2;
    float *pairs = (float *)gsm;                           // N floats: the block's {w, w*log} pairs (the ones a run covers)
    uint32_t *kw = gsm + N;                                // keep words of the block
    uint32_t *wl = kw + nW;                                // work list: one entry per noise run: (kept coefficient | run << 16, start of the run)
    const int gapCap = E_GAPCAP(N);
    int *wcount = (int *)(wl + 2 * gapCap);
    unsigned long long *nzw = (unsigned long long *)(wcount + 4);      // bit w % 64 of nzw[w / 64]: keep word w is not 0
    float *sbark = (float *)(nzw + (nW + 63) / 64);        // the block's [C][4][25] Bark levels
    unsigned long long *sexp = (unsigned long long *)(sbark + c.C * 4 * ULCX_NBARK);
    uint32_t *need = (uint32_t *)(sexp + 32);              // bit p: pair p of the block is inside a listed run
    const int wc = pre.wc;
    const int nLev = c.C * 4 * ULCX_NBARK;
#pragma unroll
    for (int i = 0; i < NS_KW; i++) if (tid + i * WG < nW) kw[tid + i * WG] = pre.kw[i];
#pragma unroll
    for (int i = 0; i < NS_BK; i++) if (tid + i * WG < nLev) sbark[tid + i * WG] = pre.bk[i];
    if (tid < 32) sexp[tid] = ulcx_exp2f_tab[tid];
    for (int i = tid; i < N / 64; i += WG) need[i] = 0u;
    if (tid == 0) *wcount = 0;
    __syncthreads();
    // gaps in front of kept coefficients.  Pass 1 lists the noise runs of every gap that is long enough for one, pass 2
    // forms the pairs those runs cover, pass 3 takes one listed run per thread (a wave's time is its longest run once).
    // Pass 1 is per keep WORD, not per kept coefficient: a gap of >= 16 zeros either ends at the word's first set bit - the
    // previous kept coefficient is the top bit of the last non-zero word, found in a bit mask of the non-zero words - or
    // lies inside the word between two set bits, and only one such run fits in 32 bits.
    for (int w0 = 0; w0 < nW; w0 += WG) {
        const int w = w0 + tid;
        const unsigned long long bm = __ballot(w < nW && kw[w] != 0u);
        if (lane == 0 && w < nW) nzw[w >> 6] = bm;
    }
    __syncthreads();
    // a gap of zr zeros in front of kept coefficient `it`, starting at `st`: its noise runs - 16 + min(rest - 16, 511)
    // coefficients each while the rest is >= 16 (Encode.c:149-160: where a run starts follows from the gap's length alone
    // as long as every run before it was coded as noise) - one work item each, and their pairs marked as needed
    auto list_gap = [&](int it, int st) {
        int rem = it - st, nr = 0;
        for (int r2 = rem; r2 >= 16; nr++) { int v = r2 - 16; if (v > 0x1FF) v = 0x1FF; r2 -= v + 16; }
        int slot = atomicAdd(wcount, nr);
        int start = st;
        for (int r = 0; r < nr; r++, slot++) {
            int v = rem - 16; if (v > 0x1FF) v = 0x1FF;
            const int n = v + 16;
            if (slot < gapCap) { wl[2 * slot] = (uint32_t)it | ((uint32_t)r << 16); wl[2 * slot + 1] = (uint32_t)start; }
            const int p0 = start >> 1, p1 = p0 + (n + (start & 1) + 1) / 2 - 1;       // the pairs run_amplitude reads
            for (int pw = p0 >> 5; pw <= (p1 >> 5); pw++) {
                uint32_t m = 0xFFFFFFFFu;
                if (pw == (p0 >> 5)) m &= 0xFFFFFFFFu << (p0 & 31);
                if (pw == (p1 >> 5)) m &= 0xFFFFFFFFu >> (31 - (p1 & 31));
                atomicOr(&need[pw], m);
            }
            start += n; rem -= n;
        }
    };
    for (int w = tid; w < nW; w += WG) {
        const uint32_t m = kw[w];
        if (m == 0u) continue;
        // start of the unit that holds this word (unit bounds are multiples of 32)
        const int i0 = w * 32, ch = i0 / c.BS, r0 = i0 - ch * c.BS;
        unsigned pat = ulcx_pattern(wc);
        int off = 0;
        for (;;) { int S = c.BS >> (pat & 7); if (r0 < off + S) break; off += S; pat >>= 4; }
        const int us = ch * c.BS + off, usw = us >> 5;
        // previous kept coefficient in front of this word, inside the unit
        int prev = us - 1;
        for (int q = w >> 6; q >= (usw >> 6); q--) {
            unsigned long long mk = nzw[q];
            if (q == (w >> 6)) mk &= (1ull << (w & 63)) - 1ull;
            if (mk) { const int wp = q * 64 + 63 - __clzll(mk); if (wp >= usw) prev = wp * 32 + 31 - __clz(kw[wp]); break; }
        }
        const int f = __ffs(m) - 1;
        if (i0 + f - (prev + 1) >= 16) list_gap(i0 + f, prev + 1);
        uint32_t z = ~m, rr = z & (z >> 1); rr &= rr >> 2; rr &= rr >> 4; rr &= rr >> 8;      // bit k: bits k..k+15 of the word are 0
        rr &= ~((2u << f) - 1u);                           // runs above the first set bit only
        if (rr) {
            const int kk = __ffs(rr) - 1;                  // the run starts behind a set bit
            const uint32_t up = m >> kk;
            if (up) list_gap(i0 + kk + __ffs(up) - 1, i0 + kk);
        }
    }
    __syncthreads();
    if (ULCX_DBG(c) & 0x10000) return;                     // (ablation build: the list alone)
    // the pairs the listed runs cover (on the bench batch a third of the block's: the dense low end has no gap of 16, the
    // tails behind the last kept coefficients are k_tails' business): two neighbouring line pairs per thread and trip,
    // geometry and table entries once for every channel
    for (int jp = 2 * tid; jp < half; jp += 2 * WG) {
        uint32_t want = 0;
        for (int ch = 0; ch < c.C; ch++) want |= ((need[(ch * half + jp) >> 5] >> (jp & 31)) & 3u) << (2 * ch);
        if (!want) continue;
        unsigned pat = ulcx_pattern(wc);
        int off = 0, dd = 0, S = c.BS, j = 0;
        for (;; j++) { dd = pat & 7; S = c.BS >> dd; if (2 * jp < off + S) break; off += S; pat >>= 4; }
        const int line = jp - off / 2;                     // (subblocks are multiples of 32 lines: both pairs lie in the same one)
        const int2 bi2 = *(const int2 *)(c.T.bandIdx[dd] + line);
        const float2 fr2 = *(const float2 *)(c.T.bandFrac[dd] + line);
        for (int ch = 0; ch < c.C; ch++) {
            if (!((want >> (2 * ch)) & 3u)) continue;
            const float *bark = sbark + (ch * 4 + j) * ULCX_NBARK;
            float o[4];
#pragma unroll
            for (int q = 0; q < 2; q++) {
                const int bi = q ? bi2.y : bi2.x;
                const float fr = q ? fr2.y : fr2.x;
                const float L = (bi < ULCX_NBARK) ? bark[bi] : bark[ULCX_NBARK - 1];
                const float R = (bi + 1 < ULCX_NBARK) ? bark[bi + 1] : L;
                const float noise = L * (1.0f - fr) + R * fr;
                const float w = ulcx_expf_t(0.5f * noise, sexp);
                o[2 * q] = w; o[2 * q + 1] = w * (noise + 0x1.62E430p-1f);
            }
            *(float4 *)(pairs + (size_t)ch * c.BS + 2 * jp) = make_float4(o[0], o[1], o[2], o[3]);
        }
    }
    __syncthreads();
    if (ULCX_DBG(c) & 0x20000) return;                     // (ablation build: list + pairs)
    // ---- one listed run per thread: run r of the gap in front of kept coefficient i leaves its amplitude in component
    //      r & 1 of gapSum[i - (r >> 1)] (positions i - 1, i - 2 .. lie inside a gap that has a third, fifth .. run: not kept)
    float *gs = (float *)(c.gapSum + (size_t)blk * N);
    int nw = *wcount; if (nw > gapCap) nw = gapCap;
    for (int t = tid; t < nw; t += WG) {
        const uint32_t it = wl[2 * t];
        const int i = (int)(it & 0xFFFFu), r = (int)(it >> 16), start = (int)wl[2 * t + 1];
        int v = i - start - 16; if (v > 0x1FF) v = 0x1FF;
        gs[2 * (i - (r >> 1)) + (r & 1)] = run_amplitude(pairs, start, v + 16);
    }
}
// Persistent workgroups: block v, v + grid, .. of the launch's blocks (all of them, or the exact path's resident list);
// the next block's keep words, Bark levels and window code travel in registers while the current one is worked on.
__global__ __launch_bounds__(WG, NSUMS_LB) void k_nsums(UlcxEncCtx c, int finalPass) {
    if (probes_over(c, finalPass)) return;
    const int n = (c.fbMode == 2) ? fb_count(c) : c.B * c.K;
    auto blk_of = [&](int v) { return (c.fbMode == 2) ? c.fbList[c.fbLo + v] : v; };
    auto next_live = [&](int v) { while (v < n && skip_block(c, blk_of(v), finalPass)) v += gridDim.x; return v; };
    int v = next_live(blockIdx.x);
    if (v >= n) return;
    NsPre cur, nxt;
    nsums_fetch(c, blk_of(v), cur);
    while (v < n) {
        const int vn = next_live(v + gridDim.x);
        if (vn < n) nsums_fetch(c, blk_of(vn), nxt);
        nsums_block(c, blk_of(v), cur);
        __syncthreads();
        cur = nxt;
        v = vn;
    }
}

// Tail HF-extension sums (NoiseFill.c:41-62) for the tail after each unit's last kept coefficient: five ordered f32 chains
// per unit over the {w, wy} pairs from there to the unit's end (two thirds of a block's pairs on the bench batch).
// A workgroup = 64 units.  All four waves form pairs - thread t: unit t & 63, pairs (t >> 6) and (t >> 6) + 4 of the
// tile's eight - into a double-buffered LDS tile, stored by component ([pair][unit]: conflict-free); then a chain wave's
// lane IS a unit and every lane of a wave runs the SAME chains - wave 0: SumX = sum w x and SumX2 = sum (w x) x, wave 1:
// SumXY = sum x wy and SumY = sum wy, wave 2: SumW = sum w - so no lane selects a factor (lanes of one wave carrying
// different chains cost two selects and two multiplications by 1.0 per pair: 1.9 wave-instructions per unit and pair, now
// 1.0).  Forming tile T + 1 and summing tile T are one instruction stream between two barriers.  A pair behind a unit's
// end is formed as {0, 0}: its terms add +0.
#define TAILS_U 64                                          // units per workgroup
#define TAILS_TP 8                                          // pairs per unit and tile
__global__ __launch_bounds__(WG) void k_tails(UlcxEncCtx c, int finalPass) {
    if (probes_over(c, finalPass)) return;
    __shared__ float tileW[2][TAILS_TP][TAILS_U], tileY[2][TAILS_TP][TAILS_U];
    __shared__ float sbark[TAILS_U][ULCX_NBARK];
    __shared__ unsigned long long sexp[32];
    __shared__ int uNp[TAILS_U], uLine0[TAILS_U], uD[TAILS_U], s_npMax;
    const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nBlk = (c.fbMode == 2) ? fb_count(c) : c.B * c.K;
    const int nBC = nBlk * c.C, nUnits = nBC * 4, N = c.C * c.BS;
    if (tid < 32) sexp[tid] = ulcx_exp2f_tab[tid];
    for (int u0 = blockIdx.x * TAILS_U; u0 < nUnits; u0 += gridDim.x * TAILS_U) {
        __syncthreads();                                     // (the previous trip's tiles and unit tables are done with)
        if (tid == 0) s_npMax = 0;
        __syncthreads();
        // ---- the units: thread u < 64 finds unit u0 + u's tail (subblock index slowest: un-decimated blocks leave the
        //      workgroups of subblocks 1..3 empty at once)
        int blk = 0, ch = 0, j = 0, start = 0, np = 0;
        bool on = false;
        if (tid < TAILS_U) {
            const int ui = u0 + tid;
            on = ui < nUnits;
            if (on) { j = ui / nBC; const int rem = ui - j * nBC; blk = rem / c.C; ch = rem - blk * c.C; }
            if (on && c.fbMode == 2) blk = c.fbList[c.fbLo + blk];
            if (on && skip_block(c, blk, finalPass)) on = false;
            int dd = 0, off = 0, S = c.BS;
            if (on) { const int wc = c.wcArr[(size_t)(blk / c.K) * (c.maxK + 2) + (blk % c.K) + 1]; on = unit_geom(wc, j, c.BS, dd, off, S); }
            if (on) {
                const uint32_t *kw = c.keep + (size_t)blk * (N / 32);
                const int ub = ch * c.BS + off, ue = ub + S;
                int last = ub - 1;                               // last kept index in [ub, ue) (unit bounds are multiples of 32)
                for (int w = (ue - 1) >> 5; (w << 5) >= ub; w--) {
                    const uint32_t m = kw[w];
                    if (m) { last = (w << 5) + 31 - __clz(m); break; }
                    if (w == 0) break;
                }
                start = last + 1;
                const int n = ue - start;
                float *tsu = c.tailSum + ((size_t)(blk * c.C + ch) * 4 + j) * 8;
                // Rate search: the sums are a function of where the tail starts (and of the block's levels), and a later
                // probe of the block often ends on the same last kept coefficient: the sums an earlier probe of THIS call
                // left for the same start are taken as they are (slot 6: set here, cleared by k_cplx at the start of every
                // rate-search call - not a call counter in the context: the drop-in replays ONE captured call).
                const bool again = c.selPass == 2 && __float_as_int(tsu[6]) == 1 && __float_as_int(tsu[5]) == start;
                tsu[5] = __int_as_float(start); tsu[6] = __int_as_float(1);
                np = (n >= 16 && !again) ? (n + (start & 1) + 1) / 2 : 0;
                uLine0[tid] = (start - ub) >> 1;
                const float *bg = c.barkN + ((size_t)(blk * c.C + ch) * 4 + j) * ULCX_NBARK;
                for (int i = 0; i < ULCX_NBARK; i++) sbark[tid][i] = bg[i];
            }
            uNp[tid] = np; uD[tid] = dd;
            if (np > 0) atomicMax(&s_npMax, np);
        }
        __syncthreads();
        const int npMax = s_npMax;
        if (npMax == 0) continue;
        // ---- forming: this thread's unit and its two pairs of every tile
        const int fu = lane, fe = wv;                        // unit, first pair of the tile (the second: fe + 4)
        const int fnp = uNp[fu], fline0 = uLine0[fu];
        const int *bandIdx = c.T.bandIdx[uD[fu]];
        const float *bandFrac = c.T.bandFrac[uD[fu]];
        const float *bark = sbark[fu];
        auto form = [&](int T) {
#pragma unroll
            for (int h = 0; h < 2; h++) {
                const int e = fe + 4 * h, q = TAILS_TP * T + e;
                float w = 0.0f, wy = 0.0f;
                if (q < fnp) {
                    const int line = fline0 + q;
                    const int bi = bandIdx[line];
                    const float fr = bandFrac[line];
                    const float L = (bi < ULCX_NBARK) ? bark[bi] : bark[ULCX_NBARK - 1];
                    const float R = (bi + 1 < ULCX_NBARK) ? bark[bi + 1] : L;
                    const float noise = L * (1.0f - fr) + R * fr;
                    w = ulcx_expf_t(0.5f * noise, sexp);
                    wy = w * (noise + 0x1.62E430p-1f);
                }
                tileW[T & 1][e][fu] = w; tileY[T & 1][e][fu] = wy;
            }
        };
        float acc0 = 0.0f, acc1 = 0.0f;                      // wave 0: SumX, SumX2; wave 1: SumXY, SumY; wave 2: SumW
        const int nT = (npMax + TAILS_TP - 1) / TAILS_TP;
        form(0);
        __syncthreads();
        for (int T = 0; T < nT; T++) {
            if (T + 1 < nT) form(T + 1);
            const int b = T & 1;
            if (wv == 0) {
#pragma unroll
                for (int i = 0; i < TAILS_TP; i++) { const float x = (TAILS_TP * T + i) * 2.0f, wx = tileW[b][i][lane] * x; acc0 += wx; acc1 += wx * x; }
            } else if (wv == 1) {
#pragma unroll
                for (int i = 0; i < TAILS_TP; i++) { const float x = (TAILS_TP * T + i) * 2.0f, y = tileY[b][i][lane]; acc0 += x * y; acc1 += y; }
            } else if (wv == 2) {
#pragma unroll
                for (int i = 0; i < TAILS_TP; i++) acc0 += tileW[b][i][lane];
            }
            __syncthreads();
        }
        // ---- the sums of unit `lane`: thread lane < 64 of wave 0 knows where they go; the other chain waves look it up the same way
        if (wv < 3 && uNp[lane] > 0) {
            const int ui = u0 + lane;
            int uj = ui / nBC; const int rem = ui - uj * nBC; int ublk = rem / c.C; const int uch = rem - ublk * c.C;
            if (c.fbMode == 2) ublk = c.fbList[c.fbLo + ublk];
            float *ts = c.tailSum + ((size_t)(ublk * c.C + uch) * 4 + uj) * 8;
            if (wv == 0) { ts[0] = acc0; ts[1] = acc1; }
            else if (wv == 1) { ts[2] = acc0; ts[3] = acc1; }
            else ts[4] = acc0;
        }
    }
}

// ---------------------------------------------------------------------------
// Encode pass: one lane per (block, channel, subblock) unit writes that unit's
// nybbles into a staging row; k_pack concatenates (Encode.c:319-360).
// ---------------------------------------------------------------------------
struct NybWriter {
    uint8_t *dst; int n; unsigned long long acc; int cap;
    __device__ __forceinline__ void put(unsigned x) {
        acc |= (unsigned long long)(x & 0xF) << ((n & 15) * 4);
        n++;
        if ((n & 15) == 0) { if (n / 2 <= cap) *(unsigned long long *)(dst + n / 2 - 8) = acc; acc = 0; }
    }
    __device__ __forceinline__ void flush() {
        if (n & 15) { int base = (n & ~15) / 2; if (base + 8 <= cap) *(unsigned long long *)(dst + base) = acc; }
    }
};
__device__ __forceinline__ void put_quantizer(NybWriter &w, int qi, bool lead) {      // Encode.c:32-45
    int s = qi - 5;
    if (lead) w.put(0xF);
    if (s < 0xE) w.put((unsigned)s);
    else { w.put(0xE); w.put((unsigned)(s - 0xE)); }
}
__device__ __forceinline__ int build_quantizer(float maxv) {                         // Encode.c:50-87
    int q = (int)(0x1.657006p2f + -0x1.715476p0f * ulcx_logf(maxv));
    if (q < 5) q = 5;
    if (q > 31) q = 31;
    return q;
}
__device__ __forceinline__ bool kept(const uint32_t *keep, int i) { return (keep[i >> 5] >> (i & 31)) & 1; }
// first kept index in [i, end), or end
__device__ __forceinline__ int next_kept(const uint32_t *keep, int i, int end) {
    while (i < end) {
        uint32_t w = keep[i >> 5] >> (i & 31);
        if (w) { i += __ffs(w) - 1; return i < end ? i : end; }
        i = (i | 31) + 1;
    }
    return end;
}
// NoiseFill.c:15-36 (band = a block-level coefficient index; the pairs are formed as they are summed: SumSrc).  What the
// speculative sums of k_nsums did not cover: about 0.02 runs per block on the bench batch.
__device__ __forceinline__ int get_noise_q(const SumSrc &g, int band, int n, float q) {
    const int p0 = band / 2;
    n = (n + (band & 1) + 1) / 2;
    float sum = 0.0f, sumw = 0.0f;
    for (int i = 0; i < n; i++) { const float2 p = pair_demand(g, p0 + i); sum += p.y; sumw += p.x; }
    if (sum == 0.0f) return 0;
    float amp = ulcx_expf(sum / sumw);
    return quant_coef_u(amp * q, 8);
}
// get_noise_q with the sums already evaluated (k_nsums)
__device__ __forceinline__ int noise_q_from_sums(float sum, float sumw, float q) {
    if (sum == 0.0f) return 0;
    float amp = ulcx_expf(sum / sumw);
    return quant_coef_u(amp * q, 8);
}
// get_hfext with the five sums already evaluated (k_nsums)
__device__ __forceinline__ void hfext_from_sums(float sx, float sx2, float sxy, float sy, float sw, float q, int &noiseQ, int &noiseDecay) {
    float det = sw * sx2 - sx * sx;
    if (det == 0.0f) { noiseQ = noiseDecay = 0; return; }
    float amp = (sx2 * sy - sx * sxy) / det;
    float dec = (sw * sxy - sx * sy) / det;
    amp = ulcx_expf(amp);
    dec = (dec < 0.0f) ? ulcx_expf(dec) : 1.0f;
    int nq = quant_coef_u(amp * q * 4.0f, 16);
    int nd = quant_u((dec - 1.0f) * -0x1.0p19f);
    if (!nd) return;
    if (nd > 0xFF) nd = 0xFF;
    noiseQ = nq; noiseDecay = nd;
}
// NoiseFill.c:41-94
__device__ __forceinline__ void get_hfext(const SumSrc &g, int band, int n, float q, int &noiseQ, int &noiseDecay) {
    const int p0 = band / 2;
    n = (n + (band & 1) + 1) / 2;
    float sx = 0.0f, sx2 = 0.0f, sxy = 0.0f, sy = 0.0f, sw = 0.0f;
    for (int i = 0; i < n; i++) {
        const float x = i * 2.0f;
        const float2 p = pair_demand(g, p0 + i);
        const float wx = p.x * x;
        sx += wx;
        sx2 += wx * x;
        sxy += x * p.y;
        sy += p.y;
        sw += p.x;
    }
    hfext_from_sums(sx, sx2, sxy, sy, sw, q, noiseQ, noiseDecay);
}

// Encode.c:92-197
__device__ __forceinline__ int write_zone(NybWriter &w, int cur, int end, float quant, const float *coef, const SumSrc &pairs,
                          const uint32_t *keep, int nextCoded, const float2 *gapSum = nullptr, int *lastKept = nullptr) {
    for (;;) {
        cur = next_kept(keep, cur, end);
        if (cur >= end) break;
        int prevKept = lastKept ? *lastKept : -2;
        if (lastKept) *lastKept = cur;
        if (fabsf(coef[cur] * quant) < 2.5f) { cur++; continue; }
        int n = 0, v = 0;
        int zr = cur - nextCoded;
        bool specOk = gapSum && (nextCoded == prevKept + 1);      // k_nsums assumed exactly this gap
        int run = 0;                                               // noise runs attempted in this gap
        while (zr) {
            if (zr <= 2) {
                int q1 = quant_coef(coef[nextCoded] * quant, 7);
                int q2 = 0;
                if (zr >= 2) q2 = quant_coef(coef[nextCoded + 1] * quant, 7);
                if (abs(q1) > 1 && (zr < 2 || abs(q2) > 1)) {
                    w.put((unsigned)q1);
                    if (zr >= 2) w.put((unsigned)q2);
                    nextCoded += zr;
                    break;
                }
            }
            int nq = 0;
            if (zr >= 16) {
                v = zr - 16; if (v > 0x1FF) v = 0x1FF;
                n = v + 16;
                float amp = -2.0f;
                if (specOk) { const float2 a = gapSum[cur - (run >> 1)]; amp = (run & 1) ? a.y : a.x; }
                if (amp > -1.5f) nq = (amp < 0.0f) ? 0 : quant_coef_u(amp * quant, 8);
                else nq = get_noise_q(pairs, nextCoded, n, quant);
                run++;
            }
            specOk = specOk && nq != 0;                            // (a zero run instead moves the start of whatever follows)
            if (nq) {
                w.put(0x8); w.put((unsigned)(v >> 5)); w.put((unsigned)(v >> 1)); w.put((unsigned)((v & 1) | ((nq - 1) << 1)));
            } else if (zr < 33) {
                v = zr - 1; if (v > 0xF) v = 0xF;
                n = v + 1;
                w.put(0x0); w.put((unsigned)v);
            } else {
                v = zr - 33; if (v > 0xFF) v = 0xFF;
                n = v + 33;
                w.put(0x1); w.put((unsigned)(v >> 4)); w.put((unsigned)v);
            }
            nextCoded += n;
            zr -= n;
        }
        w.put((unsigned)quant_coef(coef[cur] * quant, 7));
        nextCoded++;
        cur++;
    }
    return nextCoded;
}

// Encode.c:200-313
__device__ void encode_units_lane(const UlcxEncCtx &c, int finalPass, int gid, int nBlk) {
    int nUnits = nBlk * c.C * 4;
    if (gid >= nUnits) return;
    // (subblock index fastest on purpose: the state machine diverges per lane, so sparse waves -
    //  4x more of them in flight - hide its latency better than dense ones; measured 7.1 vs 9.6 ms)
    int j = gid & 3, ch = (gid >> 2) % c.C, blk = gid / (4 * c.C);
    if (c.fbMode == 2) { blk = c.fbList[c.fbLo + blk]; gid = (blk * c.C + ch) * 4 + j; }
    if (skip_block(c, blk, finalPass)) return;
    if (c.useWave && !(c.slow[blk] & 2)) return;      // only what both wave-kernel attempts could not hold
    int s = blk / c.K, k = blk % c.K;
    int wc = c.wcArr[(size_t)s * (c.maxK + 2) + k + 1];
    int d, off, S;
    if (!unit_geom(wc, j, c.BS, d, off, S)) { c.unitNyb[gid] = 0; return; }
    int N = c.C * c.BS;
    const float *coef = c.coef + (size_t)blk * N;
    const SumSrc pairs = sum_src(c, blk);
    const uint32_t *keep = c.keep + (size_t)blk * (N / 32);
    NybWriter w;
    w.cap = 2 * S + 8;
    w.dst = c.unitBuf + (size_t)blk * c.C * c.unitCap + (size_t)ch * c.unitCap + 2 * off + 8 * j;
    w.n = 0; w.acc = 0;

    int idx = ch * c.BS + off;
    int end = idx + S;
    int nextCoded = idx;
    int prevQ = -1, zoneStart = -1;
    float qmin = 1000.0f, qmax = -1000.0f;
    const float2 *gapSum = c.useGapSums ? c.gapSum + (size_t)blk * N : nullptr;
    int lastKept = idx - 1;                                  // "previous kept coefficient" before the unit = unit start - 1
    do {
        idx = next_kept(keep, idx, end);
        float nmin = 0.0f, nmax = qmax, lvl = 0.0f;
        if (idx < end) {
            lvl = fabsf(coef[idx]);
            nmin = (lvl < qmin) ? lvl : qmin;
            nmax = (lvl > qmax) ? lvl : qmax;
            if (zoneStart == -1) zoneStart = idx;
        }
        if (nmax > nmin * 4.0f) {
            int qi = build_quantizer(qmax);
            if (qi != prevQ) { put_quantizer(w, qi, prevQ != -1); prevQ = qi; }
            nextCoded = write_zone(w, zoneStart, idx, (float)(1u << qi), coef, pairs, keep, nextCoded, gapSum, &lastKept);
            zoneStart = idx;
            qmin = qmax = lvl;
        } else { qmin = nmin; qmax = nmax; }
    } while (++idx <= end);

    int n = end - nextCoded;
    if (n > 4) {
        if (prevQ != -1) w.put(0xF);
        int nq = 0, nd = 0;
        if (prevQ != -1 && n >= 16) {
            const float *ts = c.tailSum + (size_t)gid * 8;
            if (c.useGapSums && __float_as_int(ts[5]) == nextCoded) hfext_from_sums(ts[0], ts[1], ts[2], ts[3], ts[4], (float)(1u << prevQ), nq, nd);
            else get_hfext(pairs, nextCoded, n, (float)(1u << prevQ), nq, nd);
        }
        if (nq) { w.put(0xF); w.put((unsigned)(nq - 1)); w.put((unsigned)(nd >> 4)); w.put((unsigned)nd); }
        else { w.put(0xE); w.put(0xF); }
    } else if (n > 0) {
        w.put(0x0); w.put((unsigned)(n - 1));
    }
    w.flush();
    c.unitNyb[gid] = w.n;
}
__global__ __launch_bounds__(64) void k_encode_units(UlcxEncCtx c, int finalPass) {
    if (probes_over(c, finalPass)) return;
    if (c.fbMode != 2) { encode_units_lane(c, finalPass, blockIdx.x * 64 + threadIdx.x, c.B * c.K); return; }
    int n = fb_count(c), total = n * c.C * 4;
    for (int t = blockIdx.x * 64; t < total; t += gridDim.x * 64) encode_units_lane(c, finalPass, t + threadIdx.x, n);
}

// ---------------------------------------------------------------------------
// Fast encode pass: ONE WAVE per (block, channel, subblock) unit.
// The reference's WriteSubBlock (Encode.c:200-313) is a serial state machine, but its
// pieces separate cleanly once the kept coefficients are compacted:
//   1. zone segmentation  = greedy min/max scan over the kept list (sequential, ~100 items,
//      run uniformly by the wave on values passed through readlane);
//   2. quantizer per zone, quantised value + "collapses" test per kept item: independent;
//   3. the run codes of each gap between consecutive coded coefficients depend only on
//      that gap (its zero-run length, its own noise sums, the zone's quantizer): one lane
//      per gap, each doing its ordered f32 sums over LDS-resident {w, w*log} pairs;
//   4. tail HF-extension fit: five ordered f32 chains -> five lanes;
//   5. nybble positions by prefix sum, parallel emission.
// Every float operation keeps the reference's order, so the nybbles are identical.
// Units that exceed the LDS capacities below fall back to k_encode_units (c.slow).
// ---------------------------------------------------------------------------
// LDS capacities of one wave (kept coefficients / quantizer zones / nybbles per unit) are launch
// parameters: a first launch with small caps (high occupancy) covers ordinary blocks, units that
// overflow are retried by a second launch with caps that hold any unit of this block size, and only
// what still does not fit goes to the serial kernel.
struct WaveCaps { int k, z, nyb; };
#define WAVE_SK 512
#define WAVE_SZ 128
#define WAVE_SN 2048
__host__ __device__ static inline int wavecaps_lds(const WaveCaps &w) { return w.k * 4 + w.z * 8 + w.k * 4 + w.z + w.nyb + 64; }

__device__ __forceinline__ int wave_excl_scan(int v, int lane, int &total) {
    // inclusive prefix by row shifts / row broadcasts (round 3: six ds_bpermute round trips before)
    int x = v;
#define STEP(ctl, rmask) x += __builtin_amdgcn_update_dpp(0, x, ctl, rmask, 0xf, false);
    ULCX_DPP_STEPS(STEP)
#undef STEP
    total = __builtin_amdgcn_readlane(x, 63);
    (void)lane;
    return x - v;
}

// run codes of one gap (Encode.c:118-188); nybbles appended LSB-first to (lo,hi), count in cnt.
// amp0 / amp1 >= -1: noise amplitude of the gap's first / second run already evaluated by k_nsums (-1 = "Sum == 0");
// anything else is summed here, the pairs formed on the spot (get_noise_q).
__device__ __forceinline__ void gap_codes(int nc, int zr, float quant, const float *coefU, const SumSrc &src, int ubase, float amp0, float amp1,
                                          const float2 *gapI /* gapSum entry of the coefficient behind the gap, or null */,
                                          unsigned long long &lo, unsigned long long &hi, int &cnt, bool dbgNoSum = false) {
    int run = 0;                                               // noise runs attempted in this gap
    // a trip's nybbles (at most four) are gathered in a 16-bit word and appended once: one 64-bit shift per trip
    // (nybbles past the 32nd are dropped but counted: the caller treats cnt > 32 as an overflow)
    // (selects between VALUES, both words updated every time: written as if / else on lo and hi the compiler indexes
    //  the pair at run time and keeps it in scratch memory - a load, an OR and a store per code)
    auto append = [&](unsigned code, int len) {
        const int sh = 4 * cnt;
        const unsigned long long c64 = code;
        const unsigned long long toLo = (cnt < 16) ? (c64 << (sh & 63)) : 0ull;
        const unsigned long long spill = (cnt > 0 && cnt < 16) ? (c64 >> ((64 - sh) & 63)) : 0ull;      // the part of a code that crosses nybble 16
        const unsigned long long toHi = (cnt >= 16 && cnt < 32) ? (c64 << ((sh - 64) & 63)) : spill;
        lo |= toLo; hi |= toHi;
        cnt += len;
    };
    while (zr) {
        int n = 0, v = 0;
        if (zr <= 2) {
            int q1 = quant_coef(coefU[nc] * quant, 7);
            int q2 = 0;
            if (zr >= 2) q2 = quant_coef(coefU[nc + 1] * quant, 7);
            if (abs(q1) > 1 && (zr < 2 || abs(q2) > 1)) {
                if (zr >= 2) append(((unsigned)q1 & 0xF) | (((unsigned)q2 & 0xF) << 4), 2);
                else append((unsigned)q1 & 0xF, 1);
                break;
            }
        }
        int nq = 0;
        if (zr >= 16) {
            v = zr - 16; if (v > 0x1FF) v = 0x1FF;
            n = v + 16;
            if (amp0 > -1.5f) nq = (amp0 < 0.0f) ? 0 : quant_coef_u(amp0 * quant, 8);
            else nq = dbgNoSum ? 0 : get_noise_q(src, ubase + nc, n, quant);
            // the next run was speculated behind runs that were all coded as noise: the second comes with the first, the
            // third .. from the entries in front of the coefficient's (k_nsums)
            run++;
            const bool chain = nq != 0 && amp0 > -1.5f && gapI != nullptr;
            float nxt = amp1;
            if (chain && run >= 2 && zr - n >= 16) { const float2 a = gapI[-(run >> 1)]; nxt = (run & 1) ? a.y : a.x; }
            amp0 = chain ? nxt : -2.0f; amp1 = -2.0f;
        } else amp0 = -2.0f;
        if (nq) append(0x8u | (((unsigned)(v >> 5) & 0xF) << 4) | (((unsigned)(v >> 1) & 0xF) << 8) | ((((unsigned)(v & 1) | ((unsigned)(nq - 1) << 1)) & 0xF) << 12), 4);
        else if (zr < 33) { v = zr - 1; if (v > 0xF) v = 0xF; n = v + 1; append((unsigned)v << 4, 2); }
        else { v = zr - 33; if (v > 0xFF) v = 0xFF; n = v + 33; append(0x1u | (((unsigned)(v >> 4) & 0xF) << 4) | (((unsigned)v & 0xF) << 8), 3); }
        nc += n;
        zr -= n;
    }
}

// wave-local ordering of LDS traffic (all 64 lanes run in lockstep; LDS ops of one wave complete in order)
#define WAVE_SYNC() do { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); } while (0)

// Direct packing (round 3): in the final pass of a stereo, un-decimated block - one unit per channel - the two waves of
// the block write their bytes straight into the output slot instead of staging rows that k_pack shifts into place
// (Encode.c:329-359: header nybble, channel 0, channel 1, byte aligned).  xch: the pair's LDS word, through which the
// channel-0 wave tells its partner {nybbles of channel 0, its last nybble, failed, trip number}.
#define XCH_WORD(total, last, fail, seq) ((unsigned long long)((uint32_t)(total) | ((uint32_t)(last) << 16) | ((uint32_t)(fail) << 20)) | ((unsigned long long)(uint32_t)(seq) << 32))
template <bool SMALL>
__device__ void encode_unit_wave(const UlcxEncCtx &c, int finalPass, int blk, int ch, int j, int wc, int lane, float *e2, const WaveCaps caps, int failBit,
                                 unsigned long long *xch = nullptr, int seq = 0) {
    // SMALL: the ordinary-block capacities as compile-time constants (constant LDS offsets); else the launch's
    const int E2_KCAP = SMALL ? WAVE_SK : caps.k, E2_ZCAP = SMALL ? WAVE_SZ : caps.z, E2_NYBCAP = SMALL ? WAVE_SN : caps.nyb;
    int gid = (blk * c.C + ch) * 4 + j;
    int d, off, S;
    if (!unit_geom(wc, j, c.BS, d, off, S)) { if (lane == 0) c.unitNyb[gid] = 0; return; }
    const int N = c.C * c.BS;
    const int ubase = ch * c.BS + off;                       // unit offset inside the block arrays
    const float *coefU = c.coef + (size_t)blk * N + ubase;
    const SumSrc src = sum_src(c, blk);
    const float2 *gapU = c.gapSum + (size_t)blk * N + ubase;
    const uint32_t *keepU = c.keep + (size_t)blk * (N / 32) + (ubase >> 5);

    float    *kval  = e2;                                    // E2_KCAP  (later: quantised value bits)
    float    *zmax  = kval + E2_KCAP;                        // E2_ZCAP
    int      *zpre  = (int *)(zmax + E2_ZCAP);               // E2_ZCAP  inclusive prefix of quantizer-code nybbles
    uint16_t *kidx  = (uint16_t *)(zpre + E2_ZCAP);          // E2_KCAP  (bit 15 later: "previous kept item was coded")
    uint16_t *kz    = kidx + E2_KCAP;                        // E2_KCAP
    int8_t   *zqi   = (int8_t *)(kz + E2_KCAP);              // E2_ZCAP
    uint8_t  *nyb   = (uint8_t *)(zqi + E2_ZCAP);            // E2_NYBCAP (one nybble per byte)

    // B. compact the kept coefficients (rank < nOutCoef): lane L holds keep word L of the unit; a prefix sum of the words'
    //    bit counts ranks every kept coefficient, and lane j of a pass takes the j-th one - its word by a six-step search of
    //    the prefix (ds_bpermute), its bit by a five-step rank select inside the word - so that 64 kept coefficients are
    //    found and loaded per pass whatever their spread.  (Round 3: a pass used to be a round of 64 coefficient SLOTS, 32
    //    rounds a unit for ~100 kept coefficients: 0.50 of the kernel's 1.48 ms.)
    int nK = 0;
    const int nWords = S >> 5;
    // (a pass's coefficients travel while the next pass finds its own: the store of pass p sits behind the search of pass
    //  p + 1 - and in front of its loads, so that no copy of a register in flight is needed)
    float pendV = 0.0f; int pendAt = -1;
    // the unit's tail sums (k_tails), asked for now and used at the very end
    const float4 tsA = *(const float4 *)(c.tailSum + (size_t)gid * 8);
    const float2 tsB = *(const float2 *)(c.tailSum + (size_t)gid * 8 + 4);
    for (int wb = 0; wb < nWords; wb += 64) {
        const uint32_t kw = (wb + lane < nWords) ? keepU[wb + lane] : 0u;      // words past the unit read as "nothing kept"
        const int pc = __popc(kw);
        int incl = pc;
#define STEP(ctl, rmask) incl += __builtin_amdgcn_update_dpp(0, incl, ctl, rmask, 0xf, false);
        ULCX_DPP_STEPS(STEP)
#undef STEP
        const int tot = __builtin_amdgcn_readlane(incl, 63);
        for (int jb = 0; jb < tot; jb += 64) {
            const int j = jb + lane;
            const int jj = j < tot ? j : tot - 1;             // (idle lanes search for the last one: in range, unused)
            int lo = 0;                                      // the smallest word w with incl[w] > jj
#pragma unroll
            for (int step = 32; step >= 1; step >>= 1) {
                const int cand = lo + step - 1;
                const int v = __builtin_amdgcn_ds_bpermute(cand << 2, incl);
                lo = (v <= jj) ? cand + 1 : lo;
            }
            const uint32_t ww = (uint32_t)__builtin_amdgcn_ds_bpermute(lo << 2, (int)kw);
            const int iw = __builtin_amdgcn_ds_bpermute(lo << 2, incl);
            int r = jj - (iw - __popc(ww));                  // rank inside the word
            uint32_t t = ww; int pos = 0;
#pragma unroll
            for (int sh = 16; sh >= 1; sh >>= 1) {
                const int cl = __popc(t & ((1u << sh) - 1u));
                const bool up = r >= cl;
                r = up ? r - cl : r; t = up ? t >> sh : t; pos = up ? pos + sh : pos;
            }
            const int idx = (wb + lo) * 32 + pos;
            if (pendAt >= 0) kval[pendAt] = pendV;
            const bool mine = j < tot && nK + j < E2_KCAP;
            pendAt = mine ? nK + j : -1;
            if (mine) { kidx[nK + j] = (uint16_t)idx; pendV = coefU[idx]; }
        }
        nK += tot;
    }
    if (pendAt >= 0) kval[pendAt] = pendV;
    bool overflow = nK > E2_KCAP;
    WAVE_SYNC();
    if ((ULCX_DBG(c) >> 8) == 1) { if (lane == 0) c.unitNyb[gid] = 0; return; }

    // C. zone segmentation: the greedy scan of Encode.c:218-269 (a zone breaks at the first coefficient whose level puts
    //    max > 4*min over the zone so far), one ZONE per trip instead of one coefficient: a window of 64 kept levels sits in
    //    the lanes, the running minimum / maximum from the zone's start are two inclusive prefix scans (DPP; levels are
    //    non-negative floats, they order as their bit patterns), the break is the first lane whose prefixes fail the test,
    //    the zone's maximum the prefix maximum of the lane in front of it.  A zone that reaches the window's end carries
    //    its minimum / maximum into the next window.  (Round 3: every kept coefficient used to scan ahead for the break of
    //    a zone started at it - the longest of 64 such scans per round, 49 steps on the bench's blocks against 10 zones a
    //    unit - followed by a chain walk through LDS.)
    int nZ = 0;
    if (!overflow && nK > 0) {
        const uint32_t INFB = 0x7F800000u;
        uint32_t cmn = 0u, cmx = 0u;                         // the open zone's minimum / maximum so far (bit patterns)
        bool open = false;                                   // a zone continues from the previous window (wave-uniform)
        for (int base = 0; base < nK; base += 64) {
            const int i = base + lane;
            const uint32_t lv = (i < nK) ? (__float_as_uint(kval[i]) & 0x7FFFFFFFu) : INFB;     // +inf behind the list: it breaks any zone
            int myz = 0;
            int s = 0;                                       // lane the current zone starts at (0 when it is carried in)
            for (;;) {
                uint32_t a = lv;
                if (base == 0 && s == 0 && lane == 0) a = (__uint_as_float(lv) < 1000.0f) ? lv : __float_as_uint(1000.0f);   // the reference's initial QuantMin (Encode.c:219)
                a = (lane >= s) ? a : 0xFFFFFFFFu;
                uint32_t b = (lane >= s) ? lv : 0u;
#define STEP(ctl, rmask) { uint32_t o = (uint32_t)__builtin_amdgcn_update_dpp((int)0xFFFFFFFFu, (int)a, ctl, rmask, 0xf, false); a = o < a ? o : a; \
                           uint32_t q = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)b, ctl, rmask, 0xf, false); b = q > b ? q : b; }
                ULCX_DPP_STEPS(STEP)
#undef STEP
                if (open) { a = cmn < a ? cmn : a; b = cmx > b ? cmx : b; }
                const bool brk = (__uint_as_float(b) > __uint_as_float(a) * 4.0f) && (open || lane > s);
                const unsigned long long m = __ballot(brk);
                if (m == 0ull) {                             // no break in this window: the zone goes on in the next one
                    if (lane >= s) myz = nZ;
                    cmn = (uint32_t)__builtin_amdgcn_readlane((int)a, 63); cmx = (uint32_t)__builtin_amdgcn_readlane((int)b, 63);
                    open = true;
                    break;
                }
                const int t = __builtin_ctzll(m);            // first coefficient of the next zone
                if (lane >= s && lane < t) myz = nZ;
                const uint32_t zm = t > 0 ? (uint32_t)__builtin_amdgcn_readlane((int)b, t - 1) : cmx;
                if (lane == 0 && nZ < E2_ZCAP) ((uint32_t *)zmax)[nZ] = zm;
                nZ++;
                s = t; open = false;
                if (base + t >= nK) break;                   // that was the sentinel behind the list
            }
            if (i < nK) kz[i] = (uint16_t)myz;
        }
        if (open) { if (lane == 0 && nZ < E2_ZCAP) ((uint32_t *)zmax)[nZ] = cmx; nZ++; }
        WAVE_SYNC();
        if (nZ <= E2_ZCAP && __builtin_amdgcn_readfirstlane((int)((uint32_t *)zmax)[nZ - 1]) == 0) nZ--;   // end sentinel: a last zone of zero levels is not closed (Encode.c:226-238)
        if (nZ > E2_ZCAP) overflow = true;
    }
    WAVE_SYNC();
    if ((ULCX_DBG(c) >> 8) == 2) { if (lane == 0) c.unitNyb[gid] = 0; return; }

    // D. quantizer per zone + nybbles of its change code (Encode.c:240-244, 32-45)
    if (!overflow) {
        int run = 0;
        for (int base = 0; base < nZ; base += 64) {
            int z = base + lane;
            int qn = 0;
            if (z < nZ) { int qi = build_quantizer(zmax[z]); zqi[z] = (int8_t)qi; }
            WAVE_SYNC();
            if (z < nZ) {
                int qi = zqi[z];
                int prev = (z > 0) ? zqi[z - 1] : -1;
                if (qi != prev) qn = ((z > 0) ? 1 : 0) + ((qi - 5 < 0xE) ? 1 : 2);
            }
            int tot, ex = wave_excl_scan(qn, lane, tot);
            if (z < nZ) zpre[z] = run + ex + qn;
            run += tot;
        }
    }
    WAVE_SYNC();
    if ((ULCX_DBG(c) >> 8) == 3) { if (lane == 0) c.unitNyb[gid] = 0; return; }

    // E. quantise kept items, drop the ones that collapse (Encode.c:114), compact in place.
    //    Bit 15 of the compacted index records "the kept item right before me was coded too",
    //    i.e. the gap in front of me is exactly the one k_nsums speculated on.
    int nC = 0;
    if (!overflow) {
        bool prevCodedCarry = true;                          // before the first kept item: gap starts at the unit start, as speculated
        for (int base = 0; base < nK; base += 64) {
            int kk = base + lane;
            bool coded = false; int qn = 0, idx = 0, z = 0;
            if (kk < nK) {
                z = kz[kk]; idx = kidx[kk];
                float cq = kval[kk] * (float)(1u << zqi[z]);
                coded = !(fabsf(cq) < 2.5f);
                qn = quant_coef(cq, 7);
            }
            unsigned long long m = __ballot(coded);
            unsigned long long valid = __ballot(kk < nK);
            // was the previous kept item (kk-1) coded?
            bool prevCoded = (lane == 0) ? prevCodedCarry : (((m >> (lane - 1)) & 1) != 0);
            int pos = nC + __popcll(m & ((1ull << lane) - 1));
            if (coded) { kidx[pos] = (uint16_t)(idx | (prevCoded ? 0x8000 : 0)); kz[pos] = (uint16_t)z; ((int *)kval)[pos] = qn; }
            int lastValid = 63 - __clzll(valid);
            prevCodedCarry = ((m >> lastValid) & 1) != 0;
            nC += __popcll(m);
        }
    }
    WAVE_SYNC();
    if ((ULCX_DBG(c) >> 8) == 4) { if (lane == 0) c.unitNyb[gid] = 0; return; }

    // F+H. gaps -> run codes; positions by prefix sum; emission
    int total = 0;
    if (!overflow) {
        // the speculated amplitudes of a round's gaps (k_nsums) are asked for one round ahead: they arrive behind the run
        // codes of the round in front
        auto gap_amps = [&](int m) {
            float2 a = make_float2(-2.0f, -2.0f);
            if (m < nC) {
                const int raw = kidx[m], idx = raw & 0x7FFF, start = (m > 0) ? (kidx[m - 1] & 0x7FFF) + 1 : 0;
                if (idx - start >= 16 && (raw & 0x8000) && c.useGapSums) a = gapU[idx];
            }
            return a;
        };
        float2 ampR = gap_amps(lane);
        for (int base = 0; base < nC; base += 64) {
            int m = base + lane;
            unsigned long long lo = 0, hi = 0; int cnt = 0, pre = 0, z = 0, zp = -1, qn = 0;
            const float2 ampN = gap_amps(m + 64);
            if (m < nC) {
                int raw = kidx[m];
                int idx = raw & 0x7FFF;
                int start = (m > 0) ? (kidx[m - 1] & 0x7FFF) + 1 : 0;
                z = kz[m]; zp = (m > 0) ? kz[m - 1] : -1;
                qn = ((int *)kval)[m];
                pre = zpre[z] - ((zp >= 0) ? zpre[zp] : 0);
                int zr = idx - start;
                const float amp0 = ampR.x, amp1 = ampR.y;
                gap_codes(start, zr, (float)(1u << zqi[z]), coefU, src, ubase, amp0, amp1, amp0 > -1.5f ? gapU + idx : nullptr, lo, hi, cnt, (ULCX_DBG(c) & 0x40) != 0);
            }
            int mine = (m < nC) ? pre + cnt + 1 : 0;
            int tot, ex = wave_excl_scan(mine, lane, tot);
            if (cnt > 32) overflow = true;
            int p = total + ex;
            if (m < nC && finalPass && p + mine <= E2_NYBCAP && cnt <= 32) {
                for (int zz = zp + 1; zz <= z; zz++) {           // quantizer codes of the zones opened since the last coded item
                    int qi = zqi[zz], prev = (zz > 0) ? zqi[zz - 1] : -1;
                    if (qi != prev) {
                        if (zz > 0) nyb[p++] = 0xF;
                        int sft = qi - 5;
                        if (sft < 0xE) nyb[p++] = (uint8_t)sft; else { nyb[p++] = 0xE; nyb[p++] = (uint8_t)(sft - 0xE); }
                    }
                }
                for (int q = 0; q < cnt; q++) nyb[p++] = (uint8_t)(((q < 16) ? (lo >> (4 * q)) : (hi >> (4 * (q - 16)))) & 0xF);
                nyb[p++] = (uint8_t)(qn & 0xF);
            }
            total += tot;
            ampR = ampN;
        }
        overflow = __any(overflow);
    }
    if ((ULCX_DBG(c) >> 8) == 5) { if (lane == 0) c.unitNyb[gid] = 0; return; }

    // G. tail (Encode.c:271-312)
    if (!overflow) {
        int zlast = (nC > 0) ? kz[nC - 1] : -1;
        int nextCoded = (nC > 0) ? (kidx[nC - 1] & 0x7FFF) + 1 : 0;
        int n = S - nextCoded;
        int prevQ = (nZ > 0) ? zqi[nZ - 1] : -1;
        // quantizer codes of zones that closed after the last coded coefficient
        int qtail = ((nZ > 0) ? zpre[nZ - 1] : 0) - ((zlast >= 0) ? zpre[zlast] : 0);
        int nq = 0, nd = 0;
        if (n > 4 && prevQ != -1 && n >= 16) {
            float sx, sx2, sxy, sy, sw;
            if (c.useGapSums && __float_as_int(tsB.y) == ubase + nextCoded) {
                sx = tsA.x; sx2 = tsA.y; sxy = tsA.z; sy = tsA.w; sw = tsB.x;
            } else {
                // NoiseFill.c:41-62: five ordered f32 sums, one per lane 0..4 (rare: the speculated tail start was off)
                const int p0 = (ubase + nextCoded) / 2;
                int np = (n + (nextCoded & 1) + 1) / 2;
                float acc = 0.0f;
                if (lane < 5) {
                    for (int i = 0; i < np; i++) {
                        float2 pv = pair_demand(src, p0 + i);
                        float x = i * 2.0f;
                        float wx = pv.x * x;
                        float term = (lane == 0) ? wx : (lane == 1) ? wx * x : (lane == 2) ? x * pv.y : (lane == 3) ? pv.y : pv.x;
                        acc += term;
                    }
                }
                sx = __shfl(acc, 0); sx2 = __shfl(acc, 1); sxy = __shfl(acc, 2); sy = __shfl(acc, 3); sw = __shfl(acc, 4);
            }
            hfext_from_sums(sx, sx2, sxy, sy, sw, (float)(1u << prevQ), nq, nd);
        }
        int tailN = 0;
        if (n > 4) tailN = ((prevQ != -1) ? 1 : 0) + (nq ? 4 : 2);
        else if (n > 0) tailN = 2;
        int p = total + qtail;
        if (finalPass && lane == 0 && p + tailN <= E2_NYBCAP) {
            int q0 = total;
            for (int zz = zlast + 1; zz < nZ; zz++) {
                int qi = zqi[zz], prev = (zz > 0) ? zqi[zz - 1] : -1;
                if (qi != prev) {
                    if (zz > 0) nyb[q0++] = 0xF;
                    int sft = qi - 5;
                    if (sft < 0xE) nyb[q0++] = (uint8_t)sft; else { nyb[q0++] = 0xE; nyb[q0++] = (uint8_t)(sft - 0xE); }
                }
            }
            if (n > 4) {
                if (prevQ != -1) nyb[p++] = 0xF;
                if (nq) { nyb[p++] = 0xF; nyb[p++] = (uint8_t)(nq - 1); nyb[p++] = (uint8_t)((nd >> 4) & 0xF); nyb[p++] = (uint8_t)(nd & 0xF); }
                else { nyb[p++] = 0xE; nyb[p++] = 0xF; }
            } else if (n > 0) { nyb[p++] = 0x0; nyb[p++] = (uint8_t)(n - 1); }
        }
        total += qtail + tailN;
        if (total > E2_NYBCAP) overflow = true;
    }
    const bool direct = xch != nullptr;                      // (wave-uniform: final pass, stereo, un-decimated block, unit 0)
    if (overflow) {                                          // hand the whole block to the serial kernel
        if (direct && ch == 0 && lane == 0) __hip_atomic_store(xch, XCH_WORD(0, 0, 1, seq), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (lane == 0) {
            int old = atomicOr(&c.slow[blk], failBit);
            if (failBit == 1 && !(old & 1)) {              // first failure of this block: queue it for the full-capacity retry launch
                const int NBq = c.B * c.K, which = (c.fbMode == 2) ? 1 : 0;
                int q = atomicAdd(&c.slow[NBq + which], 1);
                c.slow[NBq + 2 + which * NBq + q] = blk;
            }
        }
        return;
    }
    WAVE_SYNC();
    if (lane == 0) c.unitNyb[gid] = total;
    if (!finalPass) return;
    if (direct) {
        int o = 1; uint32_t prevNyb = (uint32_t)wc & 0xFu;   // block nybble this unit starts at, the nybble in front of it
        bool ok = true;
        if (ch == 0) {
            if (lane == 0) __hip_atomic_store(xch, XCH_WORD(total, total ? nyb[total - 1] : prevNyb, 0, seq), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        } else {
            unsigned long long v64;
            for (;;) {
                v64 = __hip_atomic_load(xch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                if ((uint32_t)(v64 >> 32) == (uint32_t)seq) break;
                __builtin_amdgcn_s_sleep(2);
            }
            const uint32_t v = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)v64);
            ok = !((v >> 20) & 1u);
            o = 1 + (int)(v & 0xFFFFu); prevNyb = (v >> 16) & 0xFu;
        }
        if (ok) {
            // bytes [o/2, bEnd) of the block: channel 0 leaves a last half-filled byte to channel 1, channel 1 pads its own
            const int end = o + total;
            const int bEnd = ch == 0 ? end >> 1 : (end + 1) >> 1;
            uint8_t *outB = c.out + (size_t)blk * c.slot;
            for (int b = (o >> 1) + lane; b < bEnd; b += 64) {
                const int q0 = 2 * b - o;
                const unsigned lo4 = q0 >= 0 ? nyb[q0] : prevNyb;
                const unsigned hi4 = q0 + 1 < total ? nyb[q0 + 1] : 0u;
                if (b < c.slot) outB[b] = (uint8_t)(lo4 | (hi4 << 4));
            }
            if (ch == 1 && lane == 0) { c.bits[blk] = (end * 4 + 7) & ~7; atomicOr(&c.slow[blk], 4); }      // bit 2: packed, k_pack passes
            return;
        }
    }
    // I. nybbles -> bytes in the unit's staging row (same layout k_encode_units writes)
    uint8_t *dst = c.unitBuf + (size_t)blk * c.C * c.unitCap + (size_t)ch * c.unitCap + 2 * off + 8 * j;
    int nb = (total + 1) / 2;
    for (int b = lane; b < nb; b += 64) {
        unsigned lo4 = nyb[2 * b];
        unsigned hi4 = (2 * b + 1 < total) ? nyb[2 * b + 1] : 0;
        dst[b] = (uint8_t)(lo4 | (hi4 << 4));
    }
}


// 4 waves per workgroup (single-wave workgroups are dispatch-rate bound: ~12 ns each on MI355X),
// one wave per (block, channel), looping over that channel's subblocks.
template <bool SMALL>
__global__ __launch_bounds__(256) void k_encode_wave(UlcxEncCtx c, int finalPass, WaveCaps caps, int phase) {
    if (probes_over(c, finalPass)) return;
    extern __shared__ float e2all[];
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;   // wv: wave-uniform, so is all unit geometry
    const WaveCaps capS = { WAVE_SK, WAVE_SZ, WAVE_SN };
    const int ldsPerWave = SMALL ? wavecaps_lds(capS) : wavecaps_lds(caps);
    float *e2 = (float *)((char *)e2all + (size_t)wv * ldsPerWave);
    // phase 0: small caps, a retry follows (a failing block is queued); 1: the retry, walks that queue;
    // 2: single launch.  Failures of 1 and 2 set bit 1 = left to k_encode_units.
    const int NBq = c.B * c.K, which = (c.fbMode == 2) ? 1 : 0;
    const int *queue = c.slow + NBq + 2 + which * NBq;
    int nBlk = (phase == 1) ? c.slow[NBq + which] : (c.fbMode == 2) ? fb_count(c) : NBq;
    // direct packing: waves 2p, 2p+1 of the workgroup are the two channels of one block
    unsigned long long *xchAll = (unsigned long long *)((char *)e2all + 4 * (size_t)ldsPerWave);
    // (only when every wave of the launch makes ONE trip: the pair's LDS word carries one hand-over, a channel-0 wave a trip
    //  ahead of its partner would overwrite it - exact-path launches of more blocks than the grid covers go through k_pack)
    const bool directOK = finalPass && c.C == 2 && phase != 1 && c.directPack && (long long)gridDim.x * 4 >= (long long)nBlk * c.C;
    if (directOK) { if (threadIdx.x < 2) xchAll[threadIdx.x] = 0; __syncthreads(); }
    int seq = 0;
    for (int u = blockIdx.x * 4 + wv; u < nBlk * c.C; u += gridDim.x * 4) {      // (block, channel) index; one trip for the full-batch launch
        seq++;
        int blk = u / c.C, ch = u - blk * c.C;
        if (phase == 1) blk = queue[blk];
        else {
            if (c.fbMode == 2) blk = c.fbList[c.fbLo + blk];
            if (skip_block(c, blk, finalPass)) continue;
        }
        int s = blk / c.K, k = blk % c.K;
        int wc = c.wcArr[(size_t)s * (c.maxK + 2) + k + 1];
        const bool whole = (c.BS >> (ulcx_pattern(wc) & 7)) == c.BS;          // one unit per channel
        for (int j = 0; j < 4; j++) {
            encode_unit_wave<SMALL>(c, finalPass, blk, ch, j, wc, lane, e2, caps, phase ? 2 : 1, (directOK && whole && j == 0) ? xchAll + (wv >> 1) : nullptr, seq);
            WAVE_SYNC();
        }
    }
}

// Encode.c:329-359: header nybble(s) + units in (channel, subblock) order, byte aligned.
// One wave per block.
__device__ void pack_block(const UlcxEncCtx &c, int finalPass, int blk) {
    int lane = threadIdx.x & 63;
    if (skip_block(c, blk, finalPass)) return;
    if (finalPass && c.useWave && (c.slow[blk] & 4)) return;        // the wave writer packed this block itself
    int s = blk / c.K, k = blk % c.K;
    int wc = c.wcArr[(size_t)s * (c.maxK + 2) + k + 1];
    int nU = c.C * 4;
    const int *un = c.unitNyb + (size_t)blk * nU;
    int hdr = (wc & 8) ? 2 : 1;
    // total size (serial prefix over <= 4*C units, tiny)
    int total = hdr;
    for (int u = 0; u < nU; u++) total += un[u];
    int bitsTot = ((total * 4) + 7) & ~7;
    if (!finalPass) {
        // rate-control probe: only the size matters (ulcEncoder.c:100-110)
        if (lane == 0) {
            int budget = c.cbrBudget[blk];
            int lo = c.cbrLo[blk], hi = c.cbrHi[blk], nOut = c.nout[blk];
            bool stop = false;
            if (bitsTot < budget) lo = nOut;
            else if (bitsTot > budget) hi = nOut - 1;
            else { lo = nOut; stop = true; }
            if (stop || !(lo < hi - 1)) { c.cbrDone[blk] = 1; c.nout[blk] = lo; if (c.fbMode != 2) atomicSub(c.cbrLive, 1); }   // final pass encodes at Lo (ulcEncoder.c:113-114)
            else c.nout[blk] = (int)((unsigned)(lo + hi) / 2u);
            c.cbrLo[blk] = lo; c.cbrHi[blk] = hi;
            // The key window the later probes of this block search (k_select_wave): this probe kept the nOut keys >= T, exactly
            // those (a block whose tie group straddles the cut has left for the exact path).  Fewer coefficients from here on:
            // the thresholds are >= T, inside the nOut keys from T up.  More (or the same once more, in the final pass): they
            // are T itself or lie below it, under the nOut keys from T up.  (T may be one of several equal keys: the bounds are
            // counts at T, never at its neighbour.)
            if (c.selPass && c.fbMode != 2 && nOut > 0) {
                const uint32_t T = c.selT[blk];
                uint4 w = c.selWin[blk];
                if (bitsTot > budget) { w.x = T; w.z = (uint32_t)nOut; }
                else { w.y = T; w.w = (uint32_t)nOut; }
                c.selWin[blk] = w;
            }
        }
        return;
    }
    uint8_t *out = c.out + (size_t)blk * c.slot;
    int nBytes = bitsTot / 8;
    const uint8_t *ub = c.unitBuf + (size_t)blk * c.C * c.unitCap;
    for (int b = lane; b < nBytes; b += 64) {
        unsigned byte = 0;
        for (int h = 0; h < 2; h++) {
            int q = 2 * b + h;                 // nybble index in the block
            unsigned nyb = 0;
            if (q < hdr) nyb = (q == 0) ? (wc & 0xF) : ((wc >> 4) & 0xF);
            else if (q < total) {
                int r = q - hdr;
                int u = 0;
                while (r >= un[u]) { r -= un[u]; u++; }
                int ch = u >> 2, j = u & 3;
                int d, off, S;
                unit_geom(wc, j, c.BS, d, off, S);
                const uint8_t *src = ub + (size_t)ch * c.unitCap + 2 * off + 8 * j;
                nyb = (src[r >> 1] >> ((r & 1) * 4)) & 0xF;
            }
            byte |= nyb << (4 * h);
        }
        if (b < c.slot) out[b] = (uint8_t)byte;
    }
    if (lane == 0) c.bits[blk] = bitsTot;
}
// A probe pass of the rate search on the lock-step path: only the size matters (ulcEncoder.c:100-110).  One LANE per block
// (pack_block's probe branch is one wave per block, one lane of it working, and one atomic per finished search: at half a
// million blocks the pass in which most searches end spent 3 ms on that counter).
__global__ __launch_bounds__(256) void k_rate_step(UlcxEncCtx c) {
    if (probes_over(c, 0)) return;
    const int blk = blockIdx.x * 256 + threadIdx.x;
    bool ended = false;
    if (blk < c.B * c.K && !skip_block(c, blk, 0)) {
        const int s = blk / c.K, k = blk % c.K;
        const int wc = c.wcArr[(size_t)s * (c.maxK + 2) + k + 1];
        const int nU = c.C * 4;
        const int *un = c.unitNyb + (size_t)blk * nU;
        int total = (wc & 8) ? 2 : 1;
        for (int u = 0; u < nU; u++) total += un[u];
        const int bitsTot = ((total * 4) + 7) & ~7;
        const int budget = c.cbrBudget[blk];
        int lo = c.cbrLo[blk], hi = c.cbrHi[blk];
        const int nOut = c.nout[blk];
        bool stop = false;
        if (bitsTot < budget) lo = nOut;
        else if (bitsTot > budget) hi = nOut - 1;
        else { lo = nOut; stop = true; }
        if (stop || !(lo < hi - 1)) { c.cbrDone[blk] = 1; c.nout[blk] = lo; ended = true; }      // final pass encodes at Lo (ulcEncoder.c:113-114)
        else c.nout[blk] = (int)((unsigned)(lo + hi) / 2u);
        c.cbrLo[blk] = lo; c.cbrHi[blk] = hi;
        if (c.selPass && nOut > 0) {                              // the key window of the later probes: see pack_block
            const uint32_t T = c.selT[blk];
            uint4 w = c.selWin[blk];
            if (bitsTot > budget) { w.x = T; w.z = (uint32_t)nOut; }
            else { w.y = T; w.w = (uint32_t)nOut; }
            c.selWin[blk] = w;
        }
    }
    const unsigned long long e = __ballot(ended);
    if (e && (threadIdx.x & 63) == 0) atomicSub(c.cbrLive, (int)__popcll(e));
}

// (four blocks per workgroup: since the wave writer packs most blocks itself this kernel is mostly waves that leave at
//  once, and single-wave workgroups are bound by the dispatch rate)
__global__ __launch_bounds__(256) void k_pack(UlcxEncCtx c, int finalPass) {
    if (probes_over(c, finalPass)) return;
    const int wv = threadIdx.x >> 6;
    if (c.fbMode != 2) { const int blk = blockIdx.x * 4 + wv; if (blk < c.B * c.K) pack_block(c, finalPass, blk); return; }
    int n = fb_count(c);
    for (int v = blockIdx.x * 4 + wv; v < n; v += gridDim.x * 4) pack_block(c, finalPass, c.fbList[c.fbLo + v]);
}

// ---------------------------------------------------------------------------
// Persistent state for the next call (ulcEncoder_BlockTransform.c:93, :114)
// ---------------------------------------------------------------------------
template <typename IN>
__global__ __launch_bounds__(WG) void k_state_update(UlcxEncCtx c) {
    int s = blockIdx.x, tid = threadIdx.x;
    int n = 2 * c.BS * c.C;                     // floats of history
    float *h = c.hist + (size_t)s * n;
    int newF = c.K * c.BS * c.C;
    const IN *p = pcm_base<IN>(c) + (size_t)s * newF;
    if (c.K >= 2) {
        for (int i = tid; i < n; i += WG) h[i] = ld1(p + newF - n + i);
    } else {
        int half = n / 2;                       // disjoint per-thread index sets: no hazard
        for (int i = tid; i < half; i += WG) { h[i] = h[half + i]; h[half + i] = ld1(p + i); }
    }
    if (tid == 0) {
        UlcxWcState &w = c.wcs[s];
        const int *row = c.wcArr + (size_t)s * (c.maxK + 2);
        w.wcPrev = row[c.K];
        w.wcCur = row[c.K + 1];
        const float *bins = c.bins + ((size_t)s * (c.maxK + 1) + c.K) * 16;
        for (int i = 0; i < 8; i++) { w.binSum[i] = bins[i]; w.binW[i] = bins[8 + i]; }
    }
}
// ---------------------------------------------------------------------------
// launcher
// ---------------------------------------------------------------------------
// 4 padded arrays of BS/2 complex + BS/4 twiddles + counter (+ BS/2 floats of line energies for C > 2)
size_t ulcx_enc_xf_lds_bytes(int BS, int C) {
    if (BS > 8192) return (size_t)BS * 4;                       // k_xf_big: one unpadded array of BS/2 complex
    int ps = ulcx_xf_pad_shift(BS, C);
    size_t z = (size_t)4 * (BS + (BS >> ps)) * 4;               // four padded arrays of BS/2 complex
    size_t full = z + (size_t)BS * 2 + 32 + (C > 2 ? (size_t)BS * 2 : 0);
    return full <= ULCX_LDS_LIMIT ? full : z + (size_t)BS * 2 + 32;   // (C > 2 at BlockSize 8192: twiddles stay in global memory)
}

int ulcx_enc_nsums_slots(int BS, int C) {
    const size_t lds = nsums_lds_bytes(C * BS, C);
    if (lds > 48 * 1024 && hipFuncSetAttribute((const void *)k_nsums, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) { (void)hipGetLastError(); return 0; }
    int dev = 0, cus = 0, per = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) { (void)hipGetLastError(); return 0; }
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per, (const void *)k_nsums, WG, lds) != hipSuccess) { (void)hipGetLastError(); return 0; }
    return cus * per;
}
// launch the input-reading kernels for the call's sample type (float | PCM16)
static void launch_wc_energy(const UlcxEncCtx &c, unsigned grid, hipStream_t st, int k0, int k1) {
    if (c.pcm16) hipLaunchKernelGGL(k_wc_energy<int16_t>, dim3(grid), dim3(WG), 0, st, c, k0, k1);
    else hipLaunchKernelGGL(k_wc_energy<float>, dim3(grid), dim3(WG), 0, st, c, k0, k1);
}
static void launch_wc_ef(const UlcxEncCtx &c, hipStream_t st, int k0, int k1) {
    static_assert(EF_LDS_BYTES <= 48 * 1024, "k_wc_ef: raise the dynamic LDS limit with hipFuncSetAttribute");
    if (c.pcm16) hipLaunchKernelGGL((k_wc_ef<EF_NW, int16_t>), dim3((c.B + EF_SPW - 1) / EF_SPW), dim3(EF_NW * 64), EF_LDS_BYTES, st, c, k0, k1);
    else hipLaunchKernelGGL((k_wc_ef<EF_NW, float>), dim3((c.B + EF_SPW - 1) / EF_SPW), dim3(EF_NW * 64), EF_LDS_BYTES, st, c, k0, k1);
}
static void launch_xf(const UlcxEncCtx &c, unsigned grid, size_t lds, hipStream_t st, int k0, int k1) {
    if (c.BS > 8192) {                                      // one array at a time (k_xf_big)
        if (c.pcm16) hipLaunchKernelGGL(k_xf_big<int16_t>, dim3(grid), dim3(WG), lds, st, c, k0, k1);
        else hipLaunchKernelGGL(k_xf_big<float>, dim3(grid), dim3(WG), lds, st, c, k0, k1);
        return;
    }
    if (c.pcm16) {
        if (c.C == 2) hipLaunchKernelGGL((k_xf<true, int16_t>), dim3(grid), dim3(WG), lds, st, c, k0, k1);
        else hipLaunchKernelGGL((k_xf<false, int16_t>), dim3(grid), dim3(WG), lds, st, c, k0, k1);
    } else {
        if (c.C == 2) hipLaunchKernelGGL((k_xf<true, float>), dim3(grid), dim3(WG), lds, st, c, k0, k1);
        else hipLaunchKernelGGL((k_xf<false, float>), dim3(grid), dim3(WG), lds, st, c, k0, k1);
    }
}
static void launch_state_update(const UlcxEncCtx &c, hipStream_t st) {
    if (c.pcm16) hipLaunchKernelGGL(k_state_update<int16_t>, dim3(c.B), dim3(WG), 0, st, c);
    else hipLaunchKernelGGL(k_state_update<float>, dim3(c.B), dim3(WG), 0, st, c);
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { ulcx_set_error("%s: %s", #x, hipGetErrorString(e_)); return ULCX_ERR_HIP; } } while (0)

// Per-kernel hipEvents (on the launch stream) bracket every kernel of the first pass so
// bench.py can price each one against the roofline live; ev holds ULCX_ENC_STAGES+1 events.
const char *const ulcx_enc_stage_names[ULCX_ENC_STAGES_REPORTED] = {
    "k_wc_energy", "k_wc_forward", "k_wc_backward", "k_wc_integrate", "k_wc_decide",
    "k_xf", "k_cplx", "k_pbark", "k_mask",
    "k_select", "k_nbark", "(k_nline: gone)", "k_heapsel", "k_nsums", "k_tails", "k_encode_wave", "k_encode_units", "k_pack", "cbr_probe_passes", "k_state_update", "wc_pipeline_exposed",
};

int ulcx_enc_launch(const UlcxEncCtx &cIn, hipStream_t st, hipEvent_t *ev, const UlcxEncAux &aux) {
    UlcxEncCtx c = cIn;                                        // (keyFinal is set below for geometries without a wave selection kernel)
    hipStream_t side = aux.side, side2 = aux.side2, side3 = aux.side3;
    hipEvent_t evFork = aux.evFork, evJoin = aux.evJoin, evFork2 = aux.evFork2, *evWC = aux.evWC;
    const int wcPipe = (side && side2 && side3) ? aux.wcPipe : 1;
    if (aux.nXf) *aux.nXf = 0;
    if (c.mode != ULCX_MODE_VBR) CK(hipMemsetAsync(c.cbrLive, 0, sizeof(int), st));
    if (c.barkRing) CK(hipMemsetAsync(c.decCount, 0, sizeof(int), st));           // k_xf lists this call's decimated blocks
    int NB = c.B * c.K;
    int stage = 0;
#define MARK() do { if (ev) CK(hipEventRecord(ev[stage++], st)); } while (0)
    MARK();
    // --- window control + transform
    hipEvent_t *evX = evWC + 7 + 3 * ULCX_WC_MAXCH;            // [ULCX_XF_MAXCH] transform chunk done, [ULCX_XF_MAXCH]: all early k_cplx launches done
    const bool cplxEarly = wcPipe > 1;                        // the ordered complexity sums per transform chunk, beside the next chunk
    {
        int SG = (c.B + 63) / 64;
        // Chunks of blocks: the window-control kernels of chunk j+1.. (two stream-long serial recurrences, a few
        // hundred waves: latency-bound, nearly no machine resources) run on the side stream beside the
        // transform of chunk j on the main stream.  wcPipe = 1 keeps everything on the main stream.
        const int nCh = wcPipe;
        size_t lds = ulcx_enc_xf_lds_bytes(c.BS, c.C);
        if (lds > 48 * 1024) {
            CK(hipFuncSetAttribute((const void *)k_xf<true, float>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            CK(hipFuncSetAttribute((const void *)k_xf<false, float>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            CK(hipFuncSetAttribute((const void *)k_xf<true, int16_t>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            CK(hipFuncSetAttribute((const void *)k_xf<false, int16_t>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            CK(hipFuncSetAttribute((const void *)k_xf_big<float>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            CK(hipFuncSetAttribute((const void *)k_xf_big<int16_t>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        }
        const bool wcFuse = c.C == 2 && aux.wcFuse;   // stereo: k_wc_energy + k_wc_forward in one kernel (k_wc_ef)
        auto launch_wc = [&](hipStream_t s2, int k0, int k1, bool marks) -> int {
            int kc = k1 - k0;
            if (wcFuse) { if (marks) MARK(); launch_wc_ef(c, s2, k0, k1);                                             if (marks) MARK(); }
            else {
            launch_wc_energy(c, (unsigned)(SG * ((kc * c.BS) / 64)), s2, k0, k1);   if (marks) MARK();
            hipLaunchKernelGGL(k_wc_forward, dim3((c.B * 2 + 63) / 64), dim3(64), 0, s2, c, k0, k1);                  if (marks) MARK();
            }
            hipLaunchKernelGGL(k_wc_backward, dim3(SG * kc), dim3(64), 0, s2, c, k0, k1);                             if (marks) MARK();
            hipLaunchKernelGGL(k_wc_integrate, dim3((c.B + 63) / 64), dim3(64), 0, s2, c, k0, k1);                    if (marks) MARK();
            hipLaunchKernelGGL(k_wc_decide, dim3((c.B * kc + 63) / 64), dim3(64), 0, s2, c, k0, k1);                  if (marks) MARK();
            return ULCX_OK;
        };
        if (nCh <= 1) {
            int rc = launch_wc(st, 0, c.K, true); if (rc) return rc;
            launch_xf(c, ((NB + 7) / 8) * 8, lds, st, 0, c.K);
            MARK();
        } else {
            for (int i = 0; i < 5; i++) MARK();                    // (window-control stages: hidden in the k_xf interval in this mode)
            // side: envelope + forward recurrence of step w (the sample-rate chain, back to back over the steps);
            // side2: backward_w behind forward_w;  side3: integrate_w, decide_w behind backward_w;
            // main: transform chunk j behind the step that decides its last block.  The first chunk is a single block so the transform starts early.
            hipEvent_t ev0 = evWC[0], *evF = evWC + 1, *evB = evWC + 1 + ULCX_WC_MAXCH, *evD = evWC + 1 + 2 * ULCX_WC_MAXCH;
            // The window-control kernels advance in uniform steps of a few blocks (ULCX_WC_STEPS; 0 = in the same chunks as
            // the transform: first chunk one block, then thirds).  Measured with the chunked transform: 4 steps 8.65-8.70 ms
            // per step of 65536 blocks, 8 steps 8.73-8.81, 16 steps 8.9-9.0 (every launch of a chain kernel costs its fixed
            // latency, and a step that has to be dispatched beside a transform chunk waits for its workgroup slots).
            int nW = aux.wcSteps;
            if (nW < 0) nW = (c.K >= 8) ? 4 : 0;                               // default: 4 uniform steps (of >= 2 blocks)
            if (nW > ULCX_WC_MAXCH) nW = ULCX_WC_MAXCH;
            if (nW > c.K) nW = c.K;
            const bool sameCuts = nW < 1;
            if (sameCuts) nW = nCh;
            int cut[ULCX_XF_MAXCH + 1];
            // (transform chunks cut where the window-control steps end - eight blocks behind the first step instead of one - :
            //  the transform's interval +0.37 ms, the exposed window control -0.37 ms, the phase the same 4.0 ms)
            cut[0] = 0; cut[1] = 1;
            for (int j = 2; j <= nCh; j++) cut[j] = 1 + (c.K - 1) * (j - 1) / (nCh - 1);
            int wcs[ULCX_WC_MAXCH + 1];
            for (int w = 0; w <= nW; w++) wcs[w] = sameCuts ? cut[w] : (int)((long long)c.K * w / nW);
            CK(hipEventRecord(ev0, st));
            CK(hipStreamWaitEvent(side, ev0, 0));
            int jx = 0;                                        // next transform chunk to enqueue
            for (int w = 0; w < nW; w++) {
                const int k0 = wcs[w], k1 = wcs[w + 1], kc = k1 - k0;
                if (wcFuse) launch_wc_ef(c, side, k0, k1);
                else {
                    launch_wc_energy(c, (unsigned)(SG * ((kc * c.BS) / 64)), side, k0, k1);
                    hipLaunchKernelGGL(k_wc_forward, dim3((c.B * 2 + 63) / 64), dim3(64), 0, side, c, k0, k1);
                }
                CK(hipEventRecord(evF[w], side));
                CK(hipStreamWaitEvent(side2, evF[w], 0));
                hipLaunchKernelGGL(k_wc_backward, dim3(SG * kc), dim3(64), 0, side2, c, k0, k1);
                CK(hipEventRecord(evB[w], side2));
                CK(hipStreamWaitEvent(side3, evB[w], 0));
                hipLaunchKernelGGL(k_wc_integrate, dim3((c.B + 63) / 64), dim3(64), 0, side3, c, k0, k1);
                hipLaunchKernelGGL(k_wc_decide, dim3((c.B * kc + 63) / 64), dim3(64), 0, side3, c, k0, k1);
                CK(hipEventRecord(evD[w], side3));
                // transform chunks whose last block is now decided
                while (jx < nCh && cut[jx + 1] <= k1) {
                    int x0 = cut[jx], x1 = cut[jx + 1];
                    int nbk = c.B * (x1 - x0);
                    CK(hipStreamWaitEvent(st, evD[w], 0));
                    if (ev) CK(hipEventRecord(aux.evXf[2 * jx], st));
                    if (!(ULCX_DBG(c) & 0x2000))               // (ablation build: window control alone)
                    launch_xf(c, ((nbk + 7) / 8) * 8, lds, st, x0, x1);
                    if (ev) CK(hipEventRecord(aux.evXf[2 * jx + 1], st));
                    if (cplxEarly) CK(hipEventRecord(evX[jx], st));
                    jx++;
                }
            }
            if (aux.nXf) *aux.nXf = nCh;
            MARK();
            // the ordered complexity sums (k_cplx: lane-serial, HBM-bound) per transform chunk, on the envelope kernels' stream
            // (all of those are enqueued by now): only the last chunk's are left beside the masking sums
            if (cplxEarly) {
                for (int j = 0; j < nCh; j++) {
                    CK(hipStreamWaitEvent(side, evX[j], 0));
                    const int kc2 = cut[j + 1] - cut[j];
                    hipLaunchKernelGGL(k_cplx, dim3((c.B * kc2 + 63) / 64), dim3(64), 0, side, c, cut[j], cut[j + 1]);
                }
                CK(hipEventRecord(evX[ULCX_XF_MAXCH], side));
            }
        }
    }
    if (ULCX_DBG(c) & 0x6000) { MARK(); return ULCX_OK; }     // (ablation build: stop behind window control / transform)
    int nUnits = NB * c.C * 4;
    // (the noise log-spectrum does not feed the keys: it is launched after the selection so that the
    //  main stream has work to run beside the side-stream heapsort of tie-straddle blocks)
    const size_t barkLds = (size_t)c.barkRing * 3 * 64 * 8 + (size_t)2 * BK_TILE_FLOATS * 4;
    if (c.barkRing && barkLds > 48 * 1024) {
        CK(hipFuncSetAttribute((const void *)k_bark_uniform<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)barkLds));
        CK(hipFuncSetAttribute((const void *)k_bark_uniform<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)barkLds));
    }
    auto launch_noise = [&](hipStream_t s2, bool ev0) -> int {
        if (c.barkRing) {
            hipLaunchKernelGGL(k_bark_uniform<true>, dim3((NB * c.C + 63) / 64), dim3(256), barkLds, s2, c);
            hipLaunchKernelGGL(k_bark_levels<true>, dim3((unsigned)(((size_t)NB * c.C * 32 + WG - 1) / WG)), dim3(WG), 0, s2, c);
        }
        hipLaunchKernelGGL(k_nbark, dim3((nUnits + 63) / 64), dim3(64), 0, s2, c, c.barkRing ? 1 : 0);    if (ev0) MARK();
        if (ev0) MARK();
        return ULCX_OK;
    };
    // The noise log-spectrum (k_nbark: lane-serial ordered sums, latency-bound; k_nline) depends on the
    // transform only: it runs on a side stream beside the psychoacoustics / selection chain.
    const bool noiseAside = (side2 != nullptr && side3 != nullptr);
    hipEvent_t evN0 = evWC[1 + 3 * ULCX_WC_MAXCH], evNoise = evWC[2 + 3 * ULCX_WC_MAXCH];
    // Three lane-serial latency-bound kernels (k_pbark, k_cplx, k_nbark) fill the machine's wave slots by themselves:
    // running all three at once only makes each slower.  k_cplx (~1000 waves) runs beside k_pbark; the noise chain
    // starts behind k_pbark and runs beside the throughput-bound k_mask / k_select.
    hipEvent_t evCplx = evWC[3 + 3 * ULCX_WC_MAXCH], evTail0 = evWC[4 + 3 * ULCX_WC_MAXCH], evTail1 = evWC[5 + 3 * ULCX_WC_MAXCH], evState = evWC[6 + 3 * ULCX_WC_MAXCH];
    if (noiseAside) {
        CK(hipEventRecord(evN0, st));
        CK(hipStreamWaitEvent(side3, evN0, 0));
        if (cplxEarly) CK(hipStreamWaitEvent(side3, evX[ULCX_XF_MAXCH], 0));          // (launched per transform chunk: below)
        else hipLaunchKernelGGL(k_cplx, dim3((NB + 63) / 64), dim3(64), 0, side3, c, 0, c.K);
        CK(hipEventRecord(evCplx, side3));
        MARK();
        // the state for the next call only needs the transform to be done with the history: off the main stream
        launch_state_update(c, side3);
        CK(hipEventRecord(evState, side3));
    } else { hipLaunchKernelGGL(k_cplx, dim3((NB + 63) / 64), dim3(64), 0, st, c, 0, c.K);                 MARK(); }
    const bool noiseEarly = noiseAside;                        // the noise chain right behind the transform (it only needs nsum), beside the masking sums
    if (noiseEarly) {
        CK(hipStreamWaitEvent(side2, evN0, 0));
        int rcn = launch_noise(side2, false); if (rcn) return rcn;
        CK(hipEventRecord(evNoise, side2));
    }
    {
        const bool uniP = c.barkRing != 0;                  // masking sums of the un-decimated blocks on the geometry-uniform kernel too
        if (uniP) {
            hipLaunchKernelGGL(k_bark_uniform<false>, dim3((NB + 63) / 64), dim3(256), barkLds, st, c);
            hipLaunchKernelGGL(k_bark_levels<false>, dim3((unsigned)(((size_t)NB * 32 + WG - 1) / WG)), dim3(WG), 0, st, c);
        }
        hipLaunchKernelGGL(k_pbark, dim3((NB * 4 + 63) / 64), dim3(64), 0, st, c, uniP ? 1 : 0);          MARK();
        if (noiseAside && !noiseEarly) {
            CK(hipEventRecord(evTail0, st));                       // (reused: behind k_pbark)
            CK(hipStreamWaitEvent(side2, evTail0, 0));
            int rcn = launch_noise(side2, false); if (rcn) return rcn;
            CK(hipEventRecord(evNoise, side2));
        }
        // ("k_mask": gone - the masking level per line is formed where the keys are, mask_level(); geometries on the generic
        //  selection kernel evaluate it per key)
        MARK();
    }
    if (noiseAside) CK(hipStreamWaitEvent(st, evCplx, 0));
    // --- selection + encode pass(es)
    {
        // geometries the one-wave-per-block selection does not cover go through the multi-pass kernel, which reads every
        // key several times: form the final keys once for it (and for the exact path's heapsort)
        const int Nk = c.C * c.BS, R = Nk / 64;
        const bool selWave = (Nk % 64 == 0) && (R == 4 || R == 8 || R == 16 || R == 32 || R == 64 || R == 128);
        if (!selWave) { ulcx_enc_finalize_keys(c, st); c.keyFinal = 1; }
    }
    int N = c.C * c.BS;
    int ldsEntries = ((size_t)N * 8 <= ULCX_HEAP_LDS_BYTES) ? N : 0;
    size_t heapLds = ldsEntries ? (size_t)N * 8 + (size_t)N / 8 : 0;
    if (heapLds > 48 * 1024) { CK(hipFuncSetAttribute((const void *)k_heapsel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)heapLds)); CK(hipFuncSetAttribute((const void *)k_heapsel_pipe, hipFuncAttributeMaxDynamicSharedMemorySize, (int)heapLds)); }
    int fbGrid = NB < ULCX_HEAP_GRID ? NB : ULCX_HEAP_GRID;
    // VBR: one pass.  CBR/ABR: the reference's binary search (ulcEncoder.c:98-110) needs at most
    // ceil(log2(MaxCoef))+1 probes; every block runs its own search in lock step, then one final pass.
    int probes = 0;
    if (c.mode != ULCX_MODE_VBR) {
        probes = 2; int m = N; while (m > 1) { probes++; m >>= 1; }
        // No read-back: the host always enqueues the full count and a pass whose blocks have all converged (c.cbrLive,
        // counted down on the device) returns at the top of every kernel - nothing inside the call waits for the device.
    }
    const size_t selLds = (size_t)4 * ulcx_sel_lds_words(c.BS) * sizeof(float);
    if (selLds > 48 * 1024 && selLds <= ULCX_LDS_LIMIT && (N / 64 == 128 || N / 64 == 64)) {   // (mono BlockSize 8192: 67 KB)
#define SELA(...) do { CK(hipFuncSetAttribute((const void *)k_select_wave<__VA_ARGS__, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)selLds)); \
                       CK(hipFuncSetAttribute((const void *)k_select_wave<__VA_ARGS__, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)selLds)); \
                       CK(hipFuncSetAttribute((const void *)k_select_wave<__VA_ARGS__, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)selLds)); } while (0)
        SELA(128, 0); SELA(64, 0); SELA(64, 11);
#undef SELA
    }
    const size_t selLdsPair = (size_t)ulcx_sel_lds_words(c.BS) * sizeof(float);      // one block per workgroup
    auto launch_select = [&](int fin) {
        int R = N / 64;
        const dim3 g((NB + 3) / 4), b(256);
#define SELW(...) do { if (c.selPass == 1) hipLaunchKernelGGL((k_select_wave<__VA_ARGS__, 1>), g, b, selLds, st, c, fin); \
                       else if (c.selPass == 2) hipLaunchKernelGGL((k_select_wave<__VA_ARGS__, 2>), g, b, selLds, st, c, fin); \
                       else hipLaunchKernelGGL((k_select_wave<__VA_ARGS__, 0>), g, b, selLds, st, c, fin); } while (0)
        switch (R) {
            case 128:                                        // (one wave: ~200 VGPRs, two waves per SIMD)
                if (c.C == 2 && c.selPair) {                 // stereo BlockSize 4096: a wave per channel
                    const dim3 gp(NB), bp(128);
                    if (c.selPass == 1) hipLaunchKernelGGL((k_select_pair<64, 12, 1>), gp, bp, selLdsPair, st, c, fin);
                    else if (c.selPass == 2) hipLaunchKernelGGL((k_select_pair<64, 12, 2>), gp, bp, selLdsPair, st, c, fin);
                    else hipLaunchKernelGGL((k_select_pair<64, 12, 0>), gp, bp, selLdsPair, st, c, fin);
                } else SELW(128, 0);
                return true;
            case 64: if (c.lgBS == 11) SELW(64, 11); else SELW(64, 0); return true;      // (11: stereo BlockSize 2048; with a wave per
                                                                                         //  channel its selection is 1.08 -> 1.23 ms: barriers)
            case 32: SELW(32, 0); return true;
            case 16: SELW(16, 0); return true;
            case 8:  SELW(8, 0); return true;
            case 4:  SELW(4, 0); return true;
            default: return false;
        }
#undef SELW
    };
    // wave-kernel capacities: small (ordinary blocks, high occupancy) and full (any unit of this block size)
    WaveCaps capS = { WAVE_SK, WAVE_SZ, WAVE_SN };
    WaveCaps capF = { (c.BS + 63) & ~63, ((c.BS / 2) + 63) & ~63, 4 * c.BS + 64 };
    while ((size_t)wavecaps_lds(capF) * 4 > 150 * 1024) {       // largest that 4 waves fit in LDS; beyond it k_encode_units
        capF.k = (capF.k / 2 + 63) & ~63; capF.z = (capF.z / 2 + 63) & ~63; capF.nyb = capF.nyb / 2 + 32;
    }
    bool haveFull = capF.k > capS.k;
    // rate-control probes: k_cplx has already taken every probe that is over budget for certain, so a probe keeps at most
    // ~BitBudget/4 coefficients per block: the retry of the small launch runs with medium capacities (2 workgroups per CU
    // instead of 1), and what even they cannot hold goes to k_encode_units
    WaveCaps capM = { 1024, 512, 4096 };
    const bool haveMid = capM.k < capF.k;
    if (haveFull && (size_t)wavecaps_lds(capF) * 4 > 48 * 1024) CK(hipFuncSetAttribute((const void *)k_encode_wave<false>, hipFuncAttributeMaxDynamicSharedMemorySize, wavecaps_lds(capF) * 4 + 16));
    auto launch_encode = [&](UlcxEncCtx cc, hipStream_t s2, int fin, bool ev0, bool bigFirst) -> int {
        const bool fb2 = (cc.fbMode == 2);                 // exact path: small grids that walk the list of owned blocks
        const int fbW = NB < 128 ? NB : 128;
        if (cc.useGapSums) {
            const size_t glds = nsums_lds_bytes(N, cc.C);
            if (glds > 48 * 1024) CK(hipFuncSetAttribute((const void *)k_nsums, hipFuncAttributeMaxDynamicSharedMemorySize, (int)glds));
            // the two speculative-sum kernels are independent: on the main path the tail chains run on a side stream beside the gaps
            const bool tailAside = !fb2 && side2 != nullptr && s2 == st;
            const unsigned tg = (unsigned)((nUnits + TAILS_U - 1) / TAILS_U);
            if (tailAside) {
                CK(hipEventRecord(evTail0, s2));
                CK(hipStreamWaitEvent(side2, evTail0, 0));
                hipLaunchKernelGGL(k_tails, dim3(tg), dim3(WG), 0, side2, cc, fin);
                CK(hipEventRecord(evTail1, side2));
            }
            const int nsGrid = NB < aux.nsSlots ? NB : aux.nsSlots;             // persistent: what the device holds at once
            hipLaunchKernelGGL(k_nsums, dim3(fb2 ? fbW : nsGrid), dim3(WG), glds, s2, cc, fin);
            if (ev0 && ev) CK(hipEventRecord(ev[stage++], s2));
            if (tailAside) CK(hipStreamWaitEvent(s2, evTail1, 0));
            else hipLaunchKernelGGL(k_tails, dim3(fb2 ? fbW : tg), dim3(WG), 0, s2, cc, fin);
            if (ev0 && ev) CK(hipEventRecord(ev[stage++], s2));
        } else if (ev0 && ev) { CK(hipEventRecord(ev[stage++], s2)); CK(hipEventRecord(ev[stage++], s2)); }
        if (cc.useWave) {
            int nBC = NB * cc.C;
            // early CBR probes keep ~N/2 coefficients per block: go straight to the full-size caps there
            WaveCaps first = (bigFirst && haveFull) ? capF : capS;
            bool twoPhase = haveFull && !bigFirst;
            if (bigFirst && haveFull) hipLaunchKernelGGL(k_encode_wave<false>, dim3(fb2 ? fbW : (nBC + 3) / 4), dim3(256), (size_t)wavecaps_lds(first) * 4 + 16, s2, cc, fin, first, 2);
            else hipLaunchKernelGGL(k_encode_wave<true>, dim3(fb2 ? fbW : (nBC + 3) / 4), dim3(256), (size_t)wavecaps_lds(first) * 4 + 16, s2, cc, fin, first, twoPhase ? 0 : 2);
            if (twoPhase)
            {
                // (the exact path's few blocks also retry with the medium capacities: a full-capacity workgroup needs a whole
                //  CU's LDS and would wait for the main path's kernel to drain)
                const WaveCaps capR = (haveMid && (fb2 || (probes > 0 && !fin))) ? capM : capF;
                hipLaunchKernelGGL(k_encode_wave<false>, dim3(fb2 ? fbW : ((nBC + 3) / 4 < 512 ? (nBC + 3) / 4 : 512)), dim3(256), (size_t)wavecaps_lds(capR) * 4 + 16, s2, cc, fin, capR, 1);
            }
        }
        if (ev0 && ev) CK(hipEventRecord(ev[stage++], s2));
        hipLaunchKernelGGL(k_encode_units, dim3(fb2 ? fbW : (nUnits + 63) / 64), dim3(64), 0, s2, cc, fin);
        if (ev0 && ev) CK(hipEventRecord(ev[stage++], s2));
        if (!fin && !fb2) hipLaunchKernelGGL(k_rate_step, dim3((NB + 255) / 256), dim3(256), 0, s2, cc);
        else hipLaunchKernelGGL(k_pack, dim3(fb2 ? (fbW + 3) / 4 : (NB + 3) / 4), dim3(256), 0, s2, cc, fin);
        if (ev0 && ev) CK(hipEventRecord(ev[stage++], s2));
        return ULCX_OK;
    };
    // Exact path for tie-straddle blocks (~4e-4 of all): ONE heapsort per block and call gives the full
    // ranking, from which the block finishes its own rate search / final pass by lookup.
    auto exact_sort = [&](hipStream_t s2, int lo) -> int {
        UlcxEncCtx cf = c; cf.fbMode = 2; cf.fbLo = lo; cf.fbHi = lo + c.rankSlots;
        if (ldsEntries) hipLaunchKernelGGL(k_heapsel_pipe, dim3(fbGrid), dim3(64), heapLds, s2, cf, probes > 0 ? 1 : 0);
        else hipLaunchKernelGGL(k_heapsel, dim3(fbGrid), dim3(64), heapLds, s2, cf, ldsEntries);
        return ULCX_OK;
    };
    auto exact_passes = [&](hipStream_t s2, int lo) -> int {
        UlcxEncCtx cf = c; cf.fbMode = 2; cf.fbLo = lo; cf.fbHi = lo + c.rankSlots;
        for (int p = 0; p <= probes; p++) {
            int fin = (p == probes) ? 1 : 0;
            // (one-pass calls: k_heapsel_pipe has written the kept set, and for the first group of rank slots k_cplx cleared the counter)
            const bool fromSort = (probes == 0 && ldsEntries);
            if (!fromSort) hipLaunchKernelGGL(k_keep_ranks, dim3(fbGrid), dim3(WG), 0, s2, cf, fin);
            if (cf.useWave && !(fromSort && lo == 0)) CK(hipMemsetAsync(cf.slow + NB + 1, 0, sizeof(int), s2));      // its own retry-queue counter
            int rc = launch_encode(cf, s2, fin, false, false); if (rc) return rc;
        }
        return ULCX_OK;
    };
    // (c.fbCount, c.isFb and the first pass's c.slow are cleared by k_cplx)
    // VBR (one pass): the exact path runs on a side stream next to the encode pass of all other blocks.
    // CBR/ABR: it runs after the lock-step passes (a block joins it at whatever pass it first straddles).
    // The exact path forks at the FINAL pass (a block can first straddle there) and runs beside the main path's
    // final encode: VBR has only that pass; CBR/ABR blocks replay their whole search from the ranking there.
    const bool canFork = (side != nullptr);
    for (int p = 0; p <= probes; p++) {
        int fin = (p == probes) ? 1 : 0;
        bool ev0 = (p == 0);
        const bool async_fb = canFork && fin;
        if (c.useWave && p > 0) CK(hipMemsetAsync(c.slow, 0, sizeof(int) * ((size_t)NB + 2), st));
        c.selPass = (probes > 0 && !c.keyFinal) ? (p == 0 ? 1 : 2) : 0;
        if (!launch_select(fin)) {
            hipLaunchKernelGGL(k_select, dim3(NB), dim3(WG), 0, st, c, fin);
        }
        if (ev0) MARK();
        if (async_fb) {
            CK(hipEventRecord(evFork, st));
            CK(hipStreamWaitEvent(side, evFork, 0));
            int rc = exact_sort(side, 0); if (rc) return rc;   // needs only the keys: starts right behind the select
        }
        if (p == 0) {
            if (noiseAside) { CK(hipStreamWaitEvent(st, evNoise, 0)); MARK(); MARK(); }     // (k_nbark / k_nline intervals: hidden)
            else { int rcn = launch_noise(st, ev0); if (rcn) return rcn; }
        }
        if (async_fb) {
            CK(hipEventRecord(evFork2, st));                    // the exact path's encode pass needs the noise pairs too
            CK(hipStreamWaitEvent(side, evFork2, 0));
            int rc = exact_passes(side, 0); if (rc) return rc;
            for (int lo = c.rankSlots; lo < NB; lo += c.rankSlots) {
                rc = exact_sort(side, lo); if (rc) return rc;
                rc = exact_passes(side, lo); if (rc) return rc;
            }
            CK(hipEventRecord(evJoin, side));
        }
        if (ev0) MARK();                                       // ("k_heapsel": empty interval on the main stream)
        UlcxEncCtx cm = c; cm.fbMode = 1;
        int rc = launch_encode(cm, st, fin, ev0, false); if (rc) return rc;
        if (async_fb) CK(hipStreamWaitEvent(st, evJoin, 0));
    }
    if (!canFork) {
        for (int lo = 0; lo < NB; lo += c.rankSlots) {
            int rc = exact_sort(st, lo); if (rc) return rc;
            rc = exact_passes(st, lo); if (rc) return rc;
        }
    }
    MARK();   // cbr_probe_passes (empty interval for VBR)
    if (noiseAside) { CK(hipStreamWaitEvent(st, evState, 0));                                               MARK(); }
    else { launch_state_update(c, st);                              MARK(); }
    CK(hipGetLastError());
    return ULCX_OK;
}
