// ulcx_enc_xf.hip - transform and analysis (libulc/ulcEncoder_BlockTransform.c:95-356): TDAC fold, MDCT + MDST, line energies, block complexity; next-call state
// (one of the encoder's translation units; shared device code and every kernel's declaration: ulcx_enc_dev.h; the launch
// sequence: ulcx_enc.hip.)  Compiled with -ffp-contract=off like every file of the library.
#include "ulcx_enc_dev.h"

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
    // (the test xf_is_fast() makes, on the window codes and overlaps formed above - not a second round of loads and pattern look-ups)
#ifdef XF_NO_FAST
    const bool fastBlk = false;
#else
    const bool fastBlk = ST && (BS == 2048 || BS == 4096) && (k >= 2 || std::is_same<IN, float>::value) && (ulcx_pattern(wc) >> 4) == 0
                         && ovFirst == BS && nextOv >= BS;
#endif
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

// ---- explicit instantiations (declared extern in ulcx_enc_dev.h)
template __global__ void k_xf<false, float>(UlcxEncCtx, int, int);
template __global__ void k_xf<false, int16_t>(UlcxEncCtx, int, int);
template __global__ void k_xf<true, float>(UlcxEncCtx, int, int);
template __global__ void k_xf<true, int16_t>(UlcxEncCtx, int, int);
template __global__ void k_xf_big<float>(UlcxEncCtx, int, int);
template __global__ void k_xf_big<int16_t>(UlcxEncCtx, int, int);
template __global__ void k_state_update<float>(UlcxEncCtx);
template __global__ void k_state_update<int16_t>(UlcxEncCtx);
