// ulcx_enc_wc.hip - window control (libulc/ulcEncoder_WindowControl.c:31-239): envelope energies, forward / backward recurrences, integration, decision
// (one of the encoder's translation units; shared device code and every kernel's declaration: ulcx_enc_dev.h; the launch
// sequence: ulcx_enc.hip.)  Compiled with -ffp-contract=off like every file of the library.
#include "ulcx_enc_dev.h"

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

// ---- explicit instantiations (declared extern in ulcx_enc_dev.h)
template __global__ void k_wc_energy<float>(UlcxEncCtx, int, int);
template __global__ void k_wc_energy<int16_t>(UlcxEncCtx, int, int);
template __global__ void k_wc_ef<EF_NW, float>(UlcxEncCtx, int, int);
template __global__ void k_wc_ef<EF_NW, int16_t>(UlcxEncCtx, int, int);
