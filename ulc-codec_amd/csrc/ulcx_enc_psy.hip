// ulcx_enc_psy.hip - psychoacoustics, importance keys, coefficient selection and the exact heapsort path (libulc/ulcEncoder_Psyopt.c, ulcEncoder_BlockTransform.c:20-77)
// (one of the encoder's translation units; shared device code and every kernel's declaration: ulcx_enc_dev.h; the launch
// sequence: ulcx_enc.hip.)  Compiled with -ffp-contract=off like every file of the library.
#include "ulcx_enc_dev.h"

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
    const float4 t = c.T.bandW[d][line];                   // {left index, right index (clamped), 1 - frac, frac}
    return bark[__float_as_int(t.x)] * t.z + bark[__float_as_int(t.y)] * t.w;
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
#ifndef SEL_BRK_DLO
#define SEL_BRK_DLO 2.5f          // the sample bracket: ranks q - DLO and q + DHI of the 128 sample keys (q = kSel * 128 / N)
#define SEL_BRK_DHI 3.5f
#endif
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
        // (round 6: the first SEL_PRE coefficients of a lane are asked for here, in front of the masking levels - the registers are
        //  free until the keys pile up, and the block's first trip to HBM runs beside the levels instead of behind them)
        constexpr int SEL_PRE = (PASS != 2 && R >= 16) ? 16 : 0;
        float cpre[SEL_PRE ? SEL_PRE : 1];
        if constexpr (SEL_PRE > 0) {
#pragma unroll
            for (int q = 0; q < SEL_PRE; q++) cpre[q] = ldnt(coef + q * 64 + lane);
        }
        if constexpr (PASS != 2) {
            for (int i = lane; i < 4 * ULCX_NBARK; i += 64) sbarkw[i] = c.barkP[(size_t)blk * 4 * ULCX_NBARK + i];
            const int wcB = c.wcArr[(size_t)(blk / c.K) * (c.maxK + 2) + (blk % c.K) + 1];
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            if (PAIR) __syncthreads();                          // (both waves have stored the same Bark levels)
            if ((ulcx_pattern(wcB) & ~8u) == 0) {
                // un-decimated block (nine in ten; wave-uniform): one geometry, so a line's two level indices - clamped as mask_level()
                // clamps them - and its weights come ready from one 16-byte table entry (round 6: 9 instead of ~30 instructions a line)
#pragma unroll 4
                for (int jp = lane + (PAIR ? 64 * half : 0); jp < c.BS / 2; jp += (PAIR ? 128 : 64)) {
                    const float4 t = c.T.bandW[0][jp];
                    const float L = sbarkw[__float_as_int(t.x)], Rv = sbarkw[__float_as_int(t.y)];
                    msk[jp] = L * t.z + Rv * t.w;
                }
            } else
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
            for (int q = 0; q < 8 && r0 + q < R; q++) { int i = (r0 + q) * 64 + lane; cv[q] = (r0 + q < SEL_PRE) ? cpre[r0 + q < SEL_PRE ? r0 + q : 0] : ldnt(coef + i); mv[q] = msk[(i & (bsK - 1)) >> 1]; }
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
        constexpr bool BRK = SEL_COMPACT && PASS != 2 && R % 4 == 0;       // the sample bracket below: needs no minimum / maximum of all the keys
        uint32_t mn = 0xFFFFFFFFu, mx = 0u;
        if (!compacted) {
            if (!BRK || (ULCX_DBG(c) & 0x1000)) {
#pragma unroll
                for (int r = 0; r < R; r++) { mn = u[r] < mn ? u[r] : mn; mx = u[r] > mx ? u[r] : mx; }
            }
        } else {
#pragma unroll
            for (int j = 0; j < SEL_CAP; j++) { const uint32_t v = cd[j] ? cd[j] : 0xFFFFFFFFu; mn = v < mn ? v : mn; mx = cd[j] > mx ? cd[j] : mx; }
        }
        if (!BRK || compacted || (ULCX_DBG(c) & 0x1000)) { mn = pair_min(wave_min_u32(mn)); mx = pair_max(wave_max_u32(mx)); }
        // Round 6 - a sample bracket in front of the search (one-pass calls, one wave per block of 4096 keys).  The bit-by-bit
        // search below needs ~14 full probes (64 compares, 128 scalar instructions each) before the window that holds T is down
        // to SEL_CAND keys: log-domain keys spend their first probes on empty value space between an outlier and the bulk.
        // Instead: (i) 128 sample keys - lane group g = lane / 4 takes registers 4 g + 1 and 4 g + 3: sixteen runs of four
        // lines spread over both channels' spectra; (ii) ONE bit descent on the sample (2 compares a probe) to the value of
        // sample rank q + 3.5, q = kSel * 128 / N, keeping the smallest value it met whose sample count is below rank q - 2.5;
        // (iii) those two values are the first full probes: they bracket T in nine blocks of ten, and whatever they say the
        // window [lo, hiX] stays valid (count(u >= lo) >= kSel > count(u > hiX)); (iv) further full probes at the value where a
        // linear count between the ends crosses kSel (Illinois weights: an end that stays put counts half) until the window holds
        // <= SEL_CAP * 64 keys; (v) those go DENSELY to the candidate registers and the search below finishes on them, exactly
        // as it does for a rate search's window.  2.2 full probes + 16 sample probes on the bench's blocks instead of 13.6
        // (sized offline on the oracle's keys: tools/sel_probe_sim.py); k_select_wave<64, 11, 0> 1.07 -> 0.90 ms.  T is the same
        // number: any probe value keeps the invariant.
        bool solved = false;
        // Every geometry with more than two candidate registers' worth of keys per lane, one-pass calls and the first probe of a
        // rate search (its later probes already search a window: selWin).  R keys per lane: lane group g of 256 / R lanes takes
        // registers 4 g + 1, 4 g + 3.  PAIR (a wave per channel): the two waves swap their 128 sample keys through LDS once and
        // each runs the descent on all 256 - no exchange per sample probe; the full probes' counts are summed over the pair.
        if constexpr (BRK) {
            if (!(ULCX_DBG(c) & 0x1000) && !compacted) {
                constexpr int G = R / 4, LPG = 64 / G, STOT = PAIR ? 256 : 128;
                uint32_t s0 = 0u, s1 = 0u, s2 = 0u, s3 = 0u;
                __builtin_amdgcn_sched_barrier(0);           // (picked here, not while the keys are formed: that is where the registers are tightest)
#pragma unroll
                for (int g = 0; g < G; g++) { const bool in = (lane / LPG) == g; s0 = in ? u[4 * g + 1] : s0; s1 = in ? u[4 * g + 3] : s1; }
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (PAIR) {
                    __syncthreads();                                               // (the masking levels are used up - by both waves)
                    uint32_t *sx = (uint32_t *)(sel_lds + wv * selStride);
                    sx[half * 128 + lane] = s0; sx[half * 128 + 64 + lane] = s1;
                    __syncthreads();
                    s2 = sx[(1 - half) * 128 + lane]; s3 = sx[(1 - half) * 128 + 64 + lane];
                    __syncthreads();                                               // (the region is the candidates' next)
                }
                const float q = (float)kSel * ((float)STOT / (float)N);
                int kkLo = (int)ceilf(q + SEL_BRK_DHI); kkLo = (kkLo > STOT - 1 ? STOT - 1 : kkLo) + 1;
                int kkHi = (int)floorf(q - SEL_BRK_DLO); kkHi = (kkHi < 0 ? 0 : kkHi) + 1;
                uint32_t tLoS, tHiS = 0u; bool haveHi = false;
                // (the window starts as [0, 2^32 - 1]: count(u >= 0) = N, nothing above the top - no pass over the keys for their minimum and
                //  maximum; the descent's prefix comes from the sample's own)
                uint32_t smn = s0 < s1 ? s0 : s1, smx = s0 > s1 ? s0 : s1;
                if constexpr (PAIR) { const uint32_t a2 = s2 < s3 ? s2 : s3, b2 = s2 > s3 ? s2 : s3; smn = a2 < smn ? a2 : smn; smx = b2 > smx ? b2 : smx; }
                smn = wave_min_u32(smn); smx = wave_max_u32(smx);
                if (smn == smx) tLoS = smn;                  // (a sample of equal keys: that value is the one probe it can offer)
                else {
                    int b = 31 - __clz(smn ^ smx);
                    uint32_t Ts = smx & ~((2u << b) - 1u);
                    for (; b >= 0; b--) {
                        const uint32_t t = Ts | (1u << b);
                        int cS = __popcll(__ballot(s0 >= t)) + __popcll(__ballot(s1 >= t));
                        if constexpr (PAIR) cS += __popcll(__ballot(s2 >= t)) + __popcll(__ballot(s3 >= t));
                        Ts = (cS >= kkLo) ? t : Ts;
                        if (cS < kkHi && (!haveHi || t < tHiS)) { tHiS = t; haveHi = true; }
                        if (cS == kkLo) break;
                    }
                    tLoS = Ts;
                }
                uint32_t lo = 0u, hiX = 0xFFFFFFFFu; int cLo = N, cHi = 0;
                int last = 0, shLo = 0, shHi = 0;                 // Illinois weights as shifts: an end that stayed put counts half (all wave-uniform integers: scalar registers)
                int iter = 0, stage = 0;                          // stage 0 / 1: the sample's two values, then interpolation
                bool want = false;
                for (;;) {
                    if (lo == hiX) { T = lo; cntT = cLo; solved = true; break; }
                    uint32_t t;
                    if (stage == 0) { stage = 1; t = tLoS; if (!(t > lo && t <= hiX)) continue; }
                    else if (stage == 1) { stage = 2; t = tHiS; last = 0; shLo = shHi = 0; if (!(haveHi && t > lo && t <= hiX)) continue; }
                    else {
                        if (cLo - cHi <= SEL_CAP * 64) { want = true; break; }  // few enough for the candidate registers
                        if (cLo == kSel) {                    // exactly kSel keys from lo up: T is the smallest of them
                            uint32_t m2 = 0xFFFFFFFFu;
#pragma unroll
                            for (int r = 0; r < R; r++) { const uint32_t v = (u[r] >= lo) ? u[r] : 0xFFFFFFFFu; m2 = v < m2 ? v : m2; }
                            T = pair_min(wave_min_u32(m2)); cntT = kSel; solved = true;
                            break;
                        }
                        const int aI = (cLo - kSel) >> shLo, bI = (kSel - cHi) >> shHi;
                        const float frac = (++iter > 40 || aI + bI == 0) ? 0.5f : (float)aI / (float)(aI + bI);      // (a window that will not close by interpolation - heavy ties - is halved)
                        uint32_t off = (uint32_t)(frac * (float)(hiX - lo));
                        off = off < 1u ? 1u : off; off = off > hiX - lo ? hiX - lo : off;
                        t = (uint32_t)__builtin_amdgcn_readfirstlane((int)(lo + off));
                    }
                    // one full probe at t (lo < t <= hiX)
                    int cnt = 0;
#pragma unroll
                    for (int r = 0; r < R; r++) cnt += __popcll(__ballot(u[r] >= t));
                    cnt = pair_sum(cnt);
                    if (cnt >= kSel) { lo = t; cLo = cnt; shHi = (last == 1) ? (shHi < 20 ? shHi + 1 : shHi) : 0; shLo = 0; last = 1; }
                    else { hiX = t - 1u; cHi = cnt; shLo = (last == -1) ? (shLo < 20 ? shLo + 1 : shLo) : 0; shHi = 0; last = -1; }
                }
                if (want) {
                    // the keys of [lo, hiX] to the candidate registers, DENSELY: key number n of the window (register-major, lane-minor)
                    // goes to slot n of a list in LDS (lane mask of the compare -> mbcnt), lane l then takes slots l, 64 + l, ...: up to
                    // SEL_CAP * 64 = 512 candidates whatever their spread over the lanes (a lane's own list overflows at SEL_CAP:
                    // the search below stops at SEL_CAND = 128 for that reason and needs two more full probes to get there)
                    uint32_t *cl = (uint32_t *)(sel_lds + wv * selStride) + half * (SEL_CAP * 64);
                    const uint32_t span = hiX - lo;
                    int base = 0;
#pragma unroll
                    for (int r = 0; r < R; r++) {
                        const bool act = (u[r] - lo) <= span;
                        const unsigned long long m = __ballot(act);
                        const int idx = base + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
                        if (act) cl[idx] = u[r];
                        base += (int)__popcll(m);
                    }
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
                    for (int j = 0; j < SEL_CAP; j++) cd[j] = (j * 64 + lane < base) ? cl[j * 64 + lane] : 0u;       // (base = this wave's share of the cLo - cHi)
                    compacted = true; tried = true; cntLo = cLo; cntHi = cHi;
                }
                if (compacted) {
                    mn = 0xFFFFFFFFu; mx = 0u;
#pragma unroll
                    for (int j = 0; j < SEL_CAP; j++) { const uint32_t v = cd[j] ? cd[j] : 0xFFFFFFFFu; mn = v < mn ? v : mn; mx = cd[j] > mx ? cd[j] : mx; }
                    mn = pair_min(wave_min_u32(mn)); mx = pair_max(wave_max_u32(mx));
                }
            }
        }
        const uint32_t dif = mn ^ mx;
        if (solved) { }
        else if (ULCX_DBG(c) & 0x1000) T = mn;             // (ablation build only: no search, everything is kept)
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
template <int R, int LGBS, int PASS>
__global__ __launch_bounds__(256, SEL_MINW(R, LGBS, PASS)) void k_select_wave(UlcxEncCtx c, int finalPass) {
    if (probes_over(c, finalPass)) return;
    extern __shared__ float sel_lds[];
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int blk = blockIdx.x * 4 + wv;                  // 4 waves per workgroup, one block per wave
    if (blk >= c.B * c.K) return;
    select_body<R, LGBS, PASS, false>(c, finalPass, blk, wv, lane, 0, nullptr, sel_lds);
}

// Two waves per block (stereo: a wave per channel), one block per workgroup
template <int R, int LGBS, int PASS>
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

// ---- explicit instantiations (declared extern in ulcx_enc_dev.h)
template __global__ void k_bark_uniform<false>(UlcxEncCtx);
template __global__ void k_bark_uniform<true>(UlcxEncCtx);
template __global__ void k_bark_levels<false>(UlcxEncCtx);
template __global__ void k_bark_levels<true>(UlcxEncCtx);
template __global__ void k_select_wave<128, 0, 0>(UlcxEncCtx, int);
template __global__ void k_select_wave<128, 0, 1>(UlcxEncCtx, int);
template __global__ void k_select_wave<128, 0, 2>(UlcxEncCtx, int);
template __global__ void k_select_wave<16, 0, 0>(UlcxEncCtx, int);
template __global__ void k_select_wave<16, 0, 1>(UlcxEncCtx, int);
template __global__ void k_select_wave<16, 0, 2>(UlcxEncCtx, int);
template __global__ void k_select_wave<32, 0, 0>(UlcxEncCtx, int);
template __global__ void k_select_wave<32, 0, 1>(UlcxEncCtx, int);
template __global__ void k_select_wave<32, 0, 2>(UlcxEncCtx, int);
template __global__ void k_select_wave<4, 0, 0>(UlcxEncCtx, int);
template __global__ void k_select_wave<4, 0, 1>(UlcxEncCtx, int);
template __global__ void k_select_wave<4, 0, 2>(UlcxEncCtx, int);
template __global__ void k_select_wave<64, 0, 0>(UlcxEncCtx, int);
template __global__ void k_select_wave<64, 0, 1>(UlcxEncCtx, int);
template __global__ void k_select_wave<64, 0, 2>(UlcxEncCtx, int);
template __global__ void k_select_wave<64, 11, 0>(UlcxEncCtx, int);
template __global__ void k_select_wave<64, 11, 1>(UlcxEncCtx, int);
template __global__ void k_select_wave<64, 11, 2>(UlcxEncCtx, int);
template __global__ void k_select_wave<8, 0, 0>(UlcxEncCtx, int);
template __global__ void k_select_wave<8, 0, 1>(UlcxEncCtx, int);
template __global__ void k_select_wave<8, 0, 2>(UlcxEncCtx, int);
template __global__ void k_select_pair<64, 12, 0>(UlcxEncCtx, int);
template __global__ void k_select_pair<64, 12, 1>(UlcxEncCtx, int);
template __global__ void k_select_pair<64, 12, 2>(UlcxEncCtx, int);
