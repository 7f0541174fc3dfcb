// ulcx_enc_wr.hip - speculative noise sums and the stream writer (libulc/ulcEncoder_NoiseFill.c, ulcEncoder_Encode.c:23-360), rate-search step, packing
// (one of the encoder's translation units; shared device code and every kernel's declaration: ulcx_enc_dev.h; the launch
// sequence: ulcx_enc.hip.)  Compiled with -ffp-contract=off like every file of the library.
#include "ulcx_enc_dev.h"

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
        const float4 t0 = c.T.bandW[dd][line], t1 = c.T.bandW[dd][line + 1];      // per line {left level index, right one (clamped), 1 - frac, frac}
        for (int ch = 0; ch < c.C; ch++) {
            if (!((want >> (2 * ch)) & 3u)) continue;
            const float *bark = sbark + (ch * 4 + j) * ULCX_NBARK;
            float o[4];
#pragma unroll
            for (int q = 0; q < 2; q++) {
                const float4 t = q ? t1 : t0;
                const float noise = bark[__float_as_int(t.x)] * t.z + bark[__float_as_int(t.y)] * t.w;
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
        const float4 *bandW = c.T.bandW[uD[fu]];             // per line {left level index, right one (clamped), 1 - frac, frac}
        const float *bark = sbark[fu];
        auto form = [&](int T) {
#pragma unroll
            for (int h = 0; h < 2; h++) {
                const int e = fe + 4 * h, q = TAILS_TP * T + e;
                float w = 0.0f, wy = 0.0f;
                if (q < fnp) {
                    const int line = fline0 + q;
                    const float4 t = bandW[line];
                    const float noise = bark[__float_as_int(t.x)] * t.z + bark[__float_as_int(t.y)] * t.w;
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

template <bool SMALL>
__device__ void encode_unit_wave(const UlcxEncCtx &c, int finalPass, int blk, int ch, int j, int wc, int lane, float *e2, const WaveCaps caps, int failBit,
                                 unsigned long long *xch = nullptr, int seq = 0, uint32_t kwPre = 0u, bool havePre = false) {
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
        // (round 6: a channel's first unit starts at the channel's first keep word whatever the window code says, so its first 64 words
        //  were asked for beside the window code - k_encode_wave - instead of behind it: one trip to memory less in front of the search)
        const uint32_t kw = (wb + lane < nWords) ? ((havePre && wb == 0) ? kwPre : keepU[wb + lane]) : 0u;      // words past the unit read as "nothing kept"
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
__global__ __launch_bounds__(256, SMALL ? EW_LB : 1) void k_encode_wave(UlcxEncCtx c, int finalPass, WaveCaps caps, int phase) {
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
        // (the first keep words of the channel: where they are does not depend on the window code)
        const int nW0 = c.BS >> 5;
        const uint32_t kwPre = (lane < nW0) ? c.keep[(size_t)blk * (c.C * c.BS / 32) + (size_t)ch * nW0 + lane] : 0u;
        int wc = c.wcArr[(size_t)s * (c.maxK + 2) + k + 1];
        const bool whole = (c.BS >> (ulcx_pattern(wc) & 7)) == c.BS;          // one unit per channel
        for (int j = 0; j < 4; j++) {
            encode_unit_wave<SMALL>(c, finalPass, blk, ch, j, wc, lane, e2, caps, phase ? 2 : 1, (directOK && whole && j == 0) ? xchAll + (wv >> 1) : nullptr, seq, kwPre, j == 0);
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

int ulcx_enc_nsums_slots(int BS, int C) {
    const size_t lds = nsums_lds_bytes(C * BS, C);
    if (lds > 48 * 1024 && hipFuncSetAttribute((const void *)k_nsums, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) { (void)hipGetLastError(); return 0; }
    int dev = 0, cus = 0, per = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) { (void)hipGetLastError(); return 0; }
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per, (const void *)k_nsums, WG, lds) != hipSuccess) { (void)hipGetLastError(); return 0; }
    return cus * per;
}

// ---- explicit instantiations (declared extern in ulcx_enc_dev.h)
template __global__ void k_encode_wave<false>(UlcxEncCtx, int, WaveCaps, int);
template __global__ void k_encode_wave<true>(UlcxEncCtx, int, WaveCaps, int);
