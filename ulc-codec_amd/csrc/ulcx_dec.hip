// ulcx_dec.hip — batched ulc-codec decoder for gfx950 (MI355X), hand-written HIP.
//
//   k_dscan / k_dseed / k_dgen
//             nybble parser + dequantiser + noise synthesis
//             (libulc/ulcDecoder.c:75-197; syntax FormatSpecs.md:57-141).  The syntax is a
//             sequential state machine per (channel, subblock) unit and the noise RNG is one
//             chain per stream; the chain is cut by counting draws per unit (scan) and
//             jumping the xorshift state ahead (GF(2) matrix powers), so generation runs
//             one lane per unit over the whole batch.
//   k_dimdct  one workgroup per stream, blocks in order, lapping state resident in LDS:
//             IMDCT (one DCT-IV = complex FFT in LDS), sine-window overlap-add, the
//             reversed-time centring FIFO, inverse M/S, interleave
//             (libulc/ulcDecoder.c:198-302; IMDCT per FormatSpecs.md:150-157).
// Compiled with -ffp-contract=off (see ulcx_enc.hip).
#include "ulcx_internal.h"

#define WG 256
#include "ulcx_fft.h"

// ---------------------------------------------------------------------------
struct NybReader {                                                // ulcDecoder.c:82-88, low nybble first
    const uint8_t *p; int size;                                   // size in bits, like the reference's counter
    int limit;                                                    // a valid block never reaches the last 4 bytes of its slot;
                                                                  // past that the block is corrupt (the reference would run off its buffer)
    __device__ __forceinline__ void init(const uint8_t *base, int bitpos, int slotBytes) { p = base; size = bitpos; limit = slotBytes * 8 - 32; }
    __device__ __forceinline__ bool overrun() const { return size > limit; }
    __device__ __forceinline__ unsigned get() {
        unsigned x = p[size >> 3];
        unsigned n = (size & 4) ? (x >> 4) : (x & 0xF);
        size += 4;
        return n;
    }
};
#define ESC_STOP (-1)
#define ESC_STOP_NOISE (-2)
__device__ __forceinline__ int get_quantizer(NybReader &r) {      // ulcDecoder.c:89-95
    int q = (int)r.get();
    if (q == 0xF) return ESC_STOP_NOISE;
    if (q == 0xE) q += (int)r.get();
    if (q == 0xE + 0xF) return ESC_STOP;
    return q;
}
__device__ __forceinline__ float expand_quantizer(int q) {        // ulcDecoder.c:96-98
    return 0x1.0p-31f * (float)((1u << (31 - 5)) >> q);
}
__device__ __forceinline__ uint32_t xorshift32(uint32_t s) {      // ulcDecoder.c:75-81
    s ^= s << 13; s ^= s >> 17; s ^= s << 5;
    return s;
}
struct CoefWriter {                                               // the destination is pre-zeroed: zero runs just skip
    float *dst; int n;
    __device__ __forceinline__ void put(float v) { dst[n++] = v; }
    __device__ __forceinline__ void skip(int k) { n += k; }
};

// ulcDecoder.c:99-197.  Returns 0 on a run that overruns the subblock (corrupt).
// GEN = false: syntax walk only, counting the RNG draws the subblock consumes (one per
// noise coefficient, ulcDecoder.c:156-160,181-184); GEN = true: also emits coefficients.
template <bool GEN>
__device__ int decode_subblock(CoefWriter &w, int N, NybReader &r, uint32_t &seed, int &draws) {
    int n, v;
    bool bad = false;
    v = get_quantizer(r);
    if (v == ESC_STOP) { if (GEN) w.skip(N); return 1; }
    float quant = expand_quantizer(v);
    for (;;) {
        v = (int)r.get();
        if (v != 0x0 && v != 0x1 && v != 0x8 && v != 0xF) {
            if (GEN) {
                v = (v ^ 0x8) - 0x8;
                v = (v < 0) ? (-v * v) : (+v * v);
                w.put((float)v * quant);
            }
            if (--N == 0) break;
            continue;
        }
        if (v == 0x0) {
            n = (int)r.get() + 1;
            if (n > N) return 0;
            N -= n;
            if (GEN) w.skip(n);
            if (N == 0) break;
            continue;
        }
        if (v == 0x1) {
            n = (int)r.get();
            n = (int)r.get() | (n << 4);
            n += 33;
            if (n > N) return 0;
            N -= n;
            if (GEN) w.skip(n);
            if (N == 0) break;
            continue;
        }
        if (v == 0x8) {
            n = (int)r.get();
            n = (int)r.get() | (n << 4);
            v = (int)r.get();
            n = (v & 1) | (n << 1);
            v = (v >> 1) + 1;
            n += 16;
            if (n > N) return 0;
            N -= n;
            draws += n;
            if (GEN) {
                float p = (float)(v * v) * quant * (1.0f / 4);
                do {
                    seed = xorshift32(seed);
                    if (seed & 0x80000000u) p = -p;
                    w.put(p);
                } while (--n);
            }
            if (N == 0) break;
            continue;
        }
        v = get_quantizer(r);
        // quantizer changes are the only codes that consume no coefficient: bound them by the slot so a
        // corrupt stream cannot walk off the buffer (every other code shrinks N).  Branch-free on purpose:
        // an extra early return here cost +80 % kernel time (control-flow restructuring).
        bad |= (r.size > r.limit);
        v = bad ? ESC_STOP : v;
        if (v >= 0) { quant = expand_quantizer(v); continue; }
        if (v == ESC_STOP_NOISE) {
            v = (int)r.get() + 1;
            n = (int)r.get();
            n = (int)r.get() | (n << 4);
            draws += N;
            if (GEN) {
                float p = (float)(v * v) * quant * (1.0f / 16);
                float rr = 1.0f + (float)(n * n) * -0x1.0p-19f;
                do {
                    seed = xorshift32(seed);
                    if (seed & 0x80000000u) p = -p;
                    w.put(p); p *= rr;
                } while (--N);
            }
            break;
        }
        if (v == ESC_STOP) { if (GEN) w.skip(N); break; }
    }
    return bad ? 0 : 1;
}

// ---------------------------------------------------------------------------
// Branch-free syntax walk for the scan pass.  The block syntax (FormatSpecs.md:57-141,
// ulcDecoder.c:99-197) is a small finite-state machine over nybbles; written with selects
// instead of a per-lane branch cascade, every lane of a wave executes the same instruction
// stream per nybble whatever its own state.
// ---------------------------------------------------------------------------
enum { S_CODE = 0, S_Z0, S_Z1a, S_Z1b, S_N8a, S_N8b, S_N8c, S_QF, S_QFE, S_TN1, S_TN2, S_TN3, S_Q0, S_Q0E };
struct ScanFsm {
    int state, acc, N, draws;          // N = coefficients still to come in the current subblock
    bool done, bad;
    __device__ __forceinline__ void start(int n) { state = S_Q0; acc = 0; N = n; done = false; }
    __device__ __forceinline__ void step(int v) {
        const int st = state;
        const bool plain = (v != 0x0) & (v != 0x1) & (v != 0x8) & (v != 0xF);      // +-2..+-7: one coefficient
        // next state
        int nxCode = plain ? S_CODE : (v == 0x0) ? S_Z0 : (v == 0x1) ? S_Z1a : (v == 0x8) ? S_N8a : S_QF;
        int nxQF = (v == 0xF) ? S_TN1 : (v == 0xE) ? S_QFE : S_CODE;               // Fh,Fh.. / Fh,Eh.. / quantizer change
        int nx = S_CODE;
        nx = (st == S_CODE) ? nxCode : nx;
        nx = (st == S_Z1a) ? S_Z1b : nx;
        nx = (st == S_N8a) ? S_N8b : nx;
        nx = (st == S_N8b) ? S_N8c : nx;
        nx = (st == S_QF) ? nxQF : nx;
        nx = (st == S_TN1) ? S_TN2 : nx;
        nx = (st == S_TN2) ? S_TN3 : nx;
        nx = (st == S_Q0 && v == 0xE) ? S_Q0E : nx;                                 // first quantizer, extended form
        // effects of the code completed by this nybble
        const int a2 = (acc << 4) | v;
        int n = (st == S_CODE && plain) ? 1 : 0;
        n = (st == S_Z0) ? v + 1 : n;                                               // 0h,X      : 1..16 zeros
        n = (st == S_Z1b) ? a2 + 33 : n;                                            // 1h,Y,X    : 33..288 zeros
        const int nn = ((acc << 1) | (v & 1)) + 16;                                 // 8h,Z,Y,X  : 16..527 noise coefficients
        n = (st == S_N8c) ? nn : n;
        int dr = (st == S_N8c) ? nn : 0;
        const bool chk = (st == S_Z0) | (st == S_Z1b) | (st == S_N8c);
        const bool stopZ = ((st == S_QFE) | (st == S_Q0E)) & (v == 0xF);            // [Fh,]Eh,Fh : zeros to the end
        const bool stopN = (st == S_TN3);                                            // Fh,Fh,Z,Y,X: noise to the end
        dr = stopN ? N : dr;
        n = (stopZ | stopN) ? N : n;
        const bool over = chk & (n > N);                                            // ulcDecoder.c:127,139,154
        bad = bad | over;
        N -= over ? 0 : n;
        draws += over ? 0 : dr;
        acc = ((st == S_Z1a) | (st == S_N8a)) ? v : (st == S_N8b) ? a2 : acc;
        state = nx;
        done = (N == 0) | over;
    }
};

// base pointer of block blk's bytes: its slot, or (packed mode) its parsed offset inside the stream payload
__device__ __forceinline__ const uint8_t *block_ptr(const UlcxDecCtx &c, int blk) {
    if (!c.packed) return c.in + (size_t)blk * c.slot;
    return c.in + (size_t)(blk / c.K) * c.payStride + c.blkOff[blk];
}

// Syntax walk of one block starting at p (limitBits readable): records unit starts / draw counts.
// Returns bits consumed (0 = corrupt).
__device__ __forceinline__ int scan_block(const UlcxDecCtx &c, int blk, const uint8_t *p, int limit) {
    int pos = 0;
    // never reads at or past `limit` bits (a clamped read of byte 0 instead): running past it marks the block corrupt below
    auto get = [&]() { unsigned x = p[(pos < limit ? pos : 0) >> 3]; int v = (pos & 4) ? (x >> 4) : (x & 0xF); pos += 4; return v; };
    int wc = get();                                                 // ulcDecoder.c:211-216
    { int v2 = (int)((p[(pos < limit ? pos : 0) >> 3] >> (pos & 4)) & 0xF); bool dec = (wc & 0x8) != 0; wc |= dec ? (v2 << 4) : (1 << 4); pos += dec ? 4 : 0; }
    unsigned pat = ulcx_pattern(wc);                                // (code 0000 behaves as one plain N/1 block, as in the reference)
    int nsub = 0; { unsigned q = pat; do nsub++; while (q >>= 4); }
    if ((c.BS >> (pat & 7)) == c.BS) nsub = 1;                      // ulcDecoder.c:242-245
    int total = c.C * nsub;
    int *ustart = c.unitStart + (size_t)blk * c.C * 4;
    int *udraw  = c.unitDraws + (size_t)blk * c.C * 4;
    ScanFsm f; f.draws = 0; f.bad = false;
    int u = 0;
    ustart[0] = pos; udraw[0] = 0;
    f.start(c.BS >> (pat & 7));
    bool fin = (limit < 16);
    f.bad = fin;
    while (!fin) {
        int v = get();
        f.step(v);
        if (f.done) {
            u++;
            fin = f.bad | (u >= total);
            if (!fin) {
                int ch = u / nsub, j = u - ch * nsub;
                ustart[ch * 4 + j] = pos; udraw[ch * 4 + j] = f.draws;
                f.start(c.BS >> ((pat >> (4 * j)) & 7));
            }
        }
        if (pos > limit) { f.bad = true; fin = true; }              // ran off the readable bytes: corrupt
    }
    bool ok = !f.bad;
    c.bits[blk] = ok ? pos : 0;
    c.wc[blk] = ok ? wc : 0;
    c.draws[blk] = f.draws;
    return ok ? pos : 0;
}

// Pass 1 — one lane per block: walk the syntax, record where each (channel, subblock)
// unit starts (nybble offset) and how many RNG draws precede it inside the block.
__global__ __launch_bounds__(64) void k_dscan(UlcxDecCtx c) {
    int blk = blockIdx.x * 64 + threadIdx.x;
    if (blk >= c.B * c.K) return;
    scan_block(c, blk, c.in + (size_t)blk * c.slot, c.slot * 8 - 32);
}

// Pass 1, packed payloads — one lane per stream: a block's start is only known once the previous
// block has been parsed (the container stores no block lengths, tools/ulcDecodeTool.c:153-165).
__global__ __launch_bounds__(64) void k_dscan_packed(UlcxDecCtx c) {
    int s = blockIdx.x * 64 + threadIdx.x;
    if (s >= c.B) return;
    int off = c.packOff[s];
    int avail = c.payBytes[s];
    const uint8_t *base = c.in + (size_t)s * c.payStride;
    bool dead = false;
    for (int k = 0; k < c.K; k++) {
        int blk = s * c.K + k;
        c.blkOff[blk] = off;
        int bits = 0;
        if (!dead && off < avail) bits = scan_block(c, blk, base + off, (avail - off) * 8);
        else { c.bits[blk] = 0; c.wc[blk] = 0; c.draws[blk] = 0; }
        if (!bits) dead = true;
        off += (bits + 7) >> 3;                                     // the tool rounds every block up to a byte
    }
    c.packOff[s] = off;
}

// xorshift32 is linear over GF(2): state after n draws = T^n * state.  jump[i] holds the
// 32 columns of T^(2^i) (host-built, ulcx_api.cpp), so a jump costs popcount(n) mat-vecs.
__device__ __forceinline__ uint32_t rng_jump(const uint32_t *__restrict__ jump, uint32_t s, uint32_t n) {
    for (int i = 0; n; i++, n >>= 1) {
        if (n & 1) {
            const uint32_t *J = jump + i * 32;
            uint32_t r = 0, t = s;
            while (t) { int b = __ffs(t) - 1; r ^= J[b]; t &= t - 1; }
            s = r;
        }
    }
    return s;
}

// Pass 2 — one lane per stream: the RNG chain across blocks (ulcDecoder.c:75-81 keeps one
// seed for the life of the stream) and "a corrupt block ends the stream" (ulcDecodeTool.c:154-157).
__global__ __launch_bounds__(64) void k_dseed(UlcxDecCtx c) {
    int s = blockIdx.x * 64 + threadIdx.x;
    if (s >= c.B) return;
    uint32_t seed = c.seed[s];
    int dead = c.dead[s];
    for (int k = 0; k < c.K; k++) {
        int blk = s * c.K + k;
        if (!dead && c.wc[blk] == 0) dead = 1;
        if (dead) { c.bits[blk] = 0; c.wc[blk] = 0; continue; }
        c.blockSeed[blk] = seed;
        seed = rng_jump(c.jump, seed, (uint32_t)c.draws[blk]);
    }
    c.seed[s] = seed;
    c.dead[s] = dead;
}

// Pass 3 — one lane per (block, channel, subblock): dequantise + noise synthesis.
__global__ __launch_bounds__(64) void k_dgen(UlcxDecCtx c) {
    int tid0 = blockIdx.x * 64 + threadIdx.x;
    int nBC = c.B * c.K * c.C;
    if (tid0 >= nBC * 4) return;
    // subblock index slowest: waves of j >= 1 are empty for un-decimated blocks
    int j = tid0 / nBC, rem = tid0 - j * nBC, blk = rem / c.C, ch = rem - blk * c.C;
    int wc = c.wc[blk];
    if (wc == 0) return;
    unsigned pat = ulcx_pattern(wc);
    int off = 0, S = c.BS;
    for (int i = 0;; i++) {
        S = c.BS >> (pat & 7);
        if (i == j) break;
        if (S == c.BS) return;
        off += S;
        pat >>= 4;
        if (!pat) return;
    }
    // Only blocks the scan pass found well-formed get here (wc != 0), so the walk below follows a valid syntax.
    // ONE flat loop, no inner loops: an iteration either decodes one whole code (1-5 nybbles, read as one
    // 32-bit window) or emits up to four noise coefficients.  With one unit per lane an inner loop makes
    // every lane of the wave wait for the longest run in flight (measured: 460 k wave instructions per
    // wave of 64 units, 10x the work of any single lane); here a noise run is a state of the lane.
    typedef uint32_t u32_any_align __attribute__((aligned(1)));
    const uint8_t *src = block_ptr(c, blk);
    int bitpos = c.unitStart[(size_t)blk * c.C * 4 + ch * 4 + j];
    uint32_t seed = rng_jump(c.jump, c.blockSeed[blk], (uint32_t)c.unitDraws[(size_t)blk * c.C * 4 + ch * 4 + j]);
    float *dst = c.coef + (size_t)blk * c.C * c.BS + (size_t)ch * c.BS + off;       // pre-zeroed: zero runs just skip
    int N = S, pos = 0, pend = 0;
    float quant = 0.0f, lev = 0.0f, rr = 1.0f;
    bool first = true;                                     // the unit opens with a quantizer code without its Fh prefix
    int guard = 2 * c.slot + S + 64;                       // codes of a slot + noise coefficients: cannot be exceeded
    bool done = false;
    while (!done && guard-- > 0) {
        if (pend > 0) {
            // ulcDecoder.c:156-160 / :181-184: draw, flip on the MSB (cumulative), store, decay (rr = 1 for runs)
#pragma unroll
            for (int k = 0; k < 4; k++) {
                if (pend > 0) {
                    seed = xorshift32(seed);
                    if (seed & 0x80000000u) lev = -lev;
                    dst[pos] = lev;
                    lev *= rr;
                    pos++; pend--; N--;
                }
            }
            done = (N == 0);
        } else {
            uint32_t w = *(const u32_any_align *)(src + (bitpos >> 3));
            w >>= (bitpos & 4);                            // >= 7 valid nybbles, low nybble first (ulcDecoder.c:82-88)
            // unit start: same grammar as after an Fh, except that a leading Fh there is quantizer 15 (ScanFsm S_Q0)
            const bool q15 = first & ((w & 0xF) == 0xF);
            w = first ? ((w << 4) | 0xF) : w;
            const int v0 = w & 0xF, v1 = (w >> 4) & 0xF, v2 = (w >> 8) & 0xF, v3 = (w >> 12) & 0xF, v4 = (w >> 16) & 0xF;
            const bool plain = (v0 != 0x0) & (v0 != 0x1) & (v0 != 0x8) & (v0 != 0xF);
            const bool z0 = (v0 == 0x0), z1 = (v0 == 0x1), n8 = (v0 == 0x8), esc = (v0 == 0xF);
            const bool tail = esc & (v1 == 0xF) & !q15;                     // Fh,Fh,Z,Y,X : noise to the end
            const bool qext = esc & (v1 == 0xE);                            // Fh,Eh,X     : extended quantizer / stop
            const bool stop = qext & (v2 == 0xF);
            const bool q1 = esc & !tail & !qext;                            // Fh,X
            // coefficient (ulcDecoder.c:69-73)
            int sv = (v0 ^ 0x8) - 0x8;
            sv = (sv < 0) ? (-sv * sv) : (+sv * sv);
            if (plain) dst[pos] = (float)sv * quant;
            // code length in nybbles and coefficients consumed now
            int len = plain ? 1 : z0 ? 2 : z1 ? 3 : n8 ? 4 : tail ? 5 : qext ? 3 : 2;
            len -= first ? 1 : 0;
            int n = plain ? 1 : z0 ? v1 + 1 : z1 ? ((v1 << 4) | v2) + 33 : 0;
            n = (n > N) ? N : n;                                            // (cannot happen in a block the scan accepted)
            // noise run / tail parameters (ulcDecoder.c:95-115, :123-137)
            int np = n8 ? ((((v1 << 4) | v2) << 1) | (v3 & 1)) + 16 : tail ? N : 0;
            np = (np > N) ? N : np;
            const int l = n8 ? (v3 >> 1) + 1 : v2 + 1;
            const float lvl = (float)(l * l) * quant * (n8 ? (1.0f / 4) : (1.0f / 16));
            const int dn = (v3 << 4) | v4;
            lev = (n8 | tail) ? lvl : lev;
            rr = tail ? 1.0f + (float)(dn * dn) * -0x1.0p-19f : (n8 ? 1.0f : rr);
            pend = np;
            // quantizer change (ulcDecoder.c:89-98)
            quant = q1 ? expand_quantizer(v1) : (qext & !stop) ? expand_quantizer(0xE + v2) : quant;
            bitpos += 4 * len;
            pos += n; N -= n;
            first = false;
            done = (N == 0) | stop;
        }
    }
}

// ---------------------------------------------------------------------------
// IMDCT + overlap-add.  LDS carve (floats): lap [C][BS/2] | z [2*BS] (two arrays of BS/2 complex) |
// dec [BS] | tmpq [BS/2] | stage [2][BS]
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(WG) void k_dimdct(UlcxDecCtx c) {
    extern __shared__ float lds[];
    const int BS = c.BS, C = c.C, H2 = BS / 2;
    int s = blockIdx.x, tid = threadIdx.x;
    float  *lap   = lds;
    float2 *z     = (float2 *)(lap + (size_t)C * H2);
    float  *dec   = (float *)(z + 2 * H2);
    float  *tmpq  = dec + BS;
    float  *stage = tmpq + H2;
    float *glap = c.lap + (size_t)s * C * H2;
    for (int i = tid; i < C * H2; i += WG) lap[i] = glap[i];
    int lastSub = c.lastSub[s];
    __syncthreads();

    for (int k = 0; k < c.K; k++) {
        int blk = s * c.K + k;
        int wc = c.wc[blk];
        float *outp = c.pcm + (size_t)blk * C * BS;
        if (wc == 0) {                                              // corrupt block / dead stream
            for (int i = tid; i < C * BS; i += WG) outp[i] = 0.0f;
            continue;
        }
        const float *coefB = c.coef + (size_t)blk * C * BS;
        int newLast = lastSub;
        // ---- stereo, un-decimated block (the common case): both channels at once, inverse M/S in
        //      registers, interleaved stores; no staging through LDS
        if (C == 2 && (BS >> (ulcx_pattern(wc) & 7)) == BS) {
            const int S = BS, M = BS >> 1;
            unsigned pat0 = ulcx_pattern(wc);
            int ov = S;                                             // ulcDecoder.c:234-239
            if (pat0 & 8) ov >>= (wc & 7);
            if (ov > lastSub) ov = lastSub;
            const float2 *pre = c.T.pre[0];
            float2 *z0 = z, *z1 = z + M;
            const float *X0 = coefB, *X1 = coefB + BS;
            for (int n = tid; n < M; n += WG) {
                float2 P = pre[n];
                z0[n] = cmulc(make_float2(X0[2 * n], X0[S - 1 - 2 * n]), P);
                z1[n] = cmulc(make_float2(X1[2 * n], X1[S - 1 - 2 * n]), P);
            }
            __syncthreads();
            fftn_dif(z, 2, M, c.T.tw[0], tid);
            int a = (S - ov) >> 1;
            const float *fall = c.T.winFall + ov, *rise = c.T.winRise + ov;
            int bits = 31 - __clz(M);
            float *L0 = lap, *L1 = lap + H2;
            for (int kk = tid; kk < M / 2; kk += WG) {
                int k1 = kk, k2 = M - 1 - kk;
                int r1 = (int)(__brev((unsigned)k1) >> (32 - bits));
                int r2 = (int)(__brev((unsigned)k2) >> (32 - bits));
                float2 P1 = pre[k1], P2 = pre[k2];
                float2 ya1 = cmulc(z0[r1], P1), ya2 = cmulc(z0[r2], P2);     // channel 0 (M)
                float2 yb1 = cmulc(z1[r1], P1), yb2 = cmulc(z1[r2], P2);     // channel 1 (S)
                float A0m = L0[2 * k1], A1m = L0[2 * k1 + 1], A0s = L1[2 * k1], A1s = L1[2 * k1 + 1];
                float Bm[2] = { -ya1.y, ya2.x }, Bs[2] = { -yb1.y, yb2.x };
                float Am[2] = { A0m, A1m }, As[2] = { A0s, A1s };
                int   pv[2] = { M - 1 - 2 * k1, M - 2 - 2 * k1 };
#pragma unroll
                for (int q = 0; q < 2; q++) {
                    int p = pv[q];
                    float mLo, mHi, sLo, sHi;                                  // outputs at positions p and S-1-p
                    if (p < a) { mLo = Am[q]; mHi = Bm[q]; sLo = As[q]; sHi = Bs[q]; }
                    else {
                        float cw = fall[p - a], sw = rise[p - a];
                        float m0 = cw * Am[q], m1 = sw * Bm[q], m2 = sw * Am[q], m3 = cw * Bm[q];
                        mLo = m0 - m1; mHi = m2 + m3;
                        float s0 = cw * As[q], s1 = sw * Bs[q], s2 = sw * As[q], s3 = cw * Bs[q];
                        sLo = s0 - s1; sHi = s2 + s3;
                    }
                    // inverse M/S (ulcDecoder.c:281-289) + interleave (:292-297)
                    *(float2 *)(outp + 2 * p) = make_float2(mLo + sLo, mLo - sLo);
                    *(float2 *)(outp + 2 * (S - 1 - p)) = make_float2(mHi + sHi, mHi - sHi);
                }
                L0[2 * k1] = ya1.x; L0[2 * k1 + 1] = -ya2.y;
                L1[2 * k1] = yb1.x; L1[2 * k1 + 1] = -yb2.y;
            }
            __syncthreads();
            lastSub = S;
            continue;
        }
        for (int ch = 0; ch < C; ch++) {
            int last = lastSub;                                     // ulcDecoder.c:219
            float *dst = stage + (size_t)(ch & 1) * BS;
            float *L = lap + (size_t)ch * H2;
            unsigned pat = ulcx_pattern(wc);
            int off = 0, dpos = 0;
            do {
                int d = pat & 7, S = BS >> d, M = S >> 1;
                int ov = S;                                         // ulcDecoder.c:234-239
                if (pat & 8) ov >>= (wc & 7);
                if (ov > last) ov = last;
                last = S;
                const float *X = coefB + (size_t)ch * BS + off;
                const float2 *pre = c.T.pre[d];
                // DCT-IV pre-twiddle
                for (int n = tid; n < M; n += WG) z[n] = cmulc(make_float2(X[2 * n], X[S - 1 - 2 * n]), pre[n]);
                __syncthreads();
                fft1_dif(z, M, c.T.tw[d], tid);
                // post-twiddle fused with the windowed overlap (oracle/orc_fourier.c orc_imdct):
                //   zz[2k] = Re y[k], zz[S-1-2k] = -Im y[k];  pair p: A = lap[M-1-p], B = zz[M+p]
                float *out = (S == BS) ? dst : dec;
                int a = (S - ov) >> 1;
                const float *fall = c.T.winFall + ov, *rise = c.T.winRise + ov;
                int bits = 31 - __clz(M);
                for (int kk = tid; kk < M / 2; kk += WG) {
                    int k1 = kk, k2 = M - 1 - kk;
                    int r1 = (int)(__brev((unsigned)k1) >> (32 - bits));
                    int r2 = (int)(__brev((unsigned)k2) >> (32 - bits));
                    float2 y1 = cmulc(z[r1], pre[k1]), y2 = cmulc(z[r2], pre[k2]);
                    // zz[2k1] = y1.x, zz[2k1+1] = -y2.y (new lap);  zz[S-1-2k1] = -y1.y, zz[S-2-2k1] = y2.x (B values)
                    float A0 = L[2 * k1], A1 = L[2 * k1 + 1];
                    float Bv[2] = { -y1.y, y2.x };
                    float Av[2] = { A0, A1 };
                    int   pv[2] = { M - 1 - 2 * k1, M - 2 - 2 * k1 };
#pragma unroll
                    for (int q = 0; q < 2; q++) {
                        int p = pv[q];
                        float A = Av[q], B = Bv[q];
                        if (p < a) { out[p] = A; out[S - 1 - p] = B; }
                        else {
                            float cw = fall[p - a], sw = rise[p - a];
                            float m0 = cw * A, m1 = sw * B, m2 = sw * A, m3 = cw * B;
                            out[p] = m0 - m1;
                            out[S - 1 - p] = m2 + m3;
                        }
                    }
                    L[2 * k1] = y1.x;
                    L[2 * k1 + 1] = -y2.y;
                }
                __syncthreads();
                if (S == BS) break;                                 // ulcDecoder.c:242-245
                // reversed-time centring FIFO in lap[M .. BS/2) (ulcDecoder.c:253-272)
                int avail = (BS - S) >> 1;
                for (int q = tid; q < avail; q += WG) tmpq[q] = L[H2 - 1 - q];      // queue[q], q = 0 is the oldest
                __syncthreads();
                for (int n = tid; n < S; n += WG)
                    dst[dpos + n] = (n < avail) ? tmpq[n] : dec[n - avail];
                if (S <= avail) {
                    for (int q = tid; q < avail; q += WG)
                        L[H2 - 1 - q] = (q < avail - S) ? tmpq[q + S] : dec[q - (avail - S)];
                } else {
                    for (int q = tid; q < avail; q += WG) L[H2 - 1 - q] = dec[S - avail + q];
                }
                __syncthreads();
                dpos += S; off += S;
            } while (pat >>= 4);
            newLast = last;
            // inverse M/S + interleave once both members of a pair (or a trailing single) are staged
            bool pairDone = (ch & 1) || (ch == C - 1);
            if (pairDone) {
                __syncthreads();
                if (ch & 1) {
                    for (int n = tid; n < BS; n += WG) {
                        float m = stage[n], sd = stage[BS + n];                 // ulcDecoder.c:281-289
                        float l = m + sd, r = m - sd;
                        if (C == 2) *(float2 *)(outp + 2 * n) = make_float2(l, r);
                        else { outp[(size_t)n * C + ch - 1] = l; outp[(size_t)n * C + ch] = r; }
                    }
                } else {
                    for (int n = tid; n < BS; n += WG) outp[(size_t)n * C + ch] = stage[n];
                }
                __syncthreads();
            }
        }
        lastSub = newLast;
    }
    for (int i = tid; i < C * H2; i += WG) glap[i] = lap[i];
    if (tid == 0) c.lastSub[s] = lastSub;
}

size_t ulcx_dec_lds_bytes(int BS, int C) {
    return sizeof(float) * ((size_t)C * (BS / 2) + 2 * (size_t)BS + BS + BS / 2 + 2 * (size_t)BS);
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { ulcx_set_error("%s: %s", #x, hipGetErrorString(e_)); return ULCX_ERR_HIP; } } while (0)

int ulcx_dec_launch(const UlcxDecCtx &c, hipStream_t st, hipEvent_t *ev) {
    int stage = 0;
    if (ev) CK(hipEventRecord(ev[stage++], st));
    int NB = c.B * c.K;
    if (c.packed) hipLaunchKernelGGL(k_dscan_packed, dim3((c.B + 63) / 64), dim3(64), 0, st, c);
    else hipLaunchKernelGGL(k_dscan, dim3((NB + 63) / 64), dim3(64), 0, st, c);
    if (ev) CK(hipEventRecord(ev[stage++], st));
    hipLaunchKernelGGL(k_dseed, dim3((c.B + 63) / 64), dim3(64), 0, st, c);
    if (ev) CK(hipEventRecord(ev[stage++], st));
    CK(hipMemsetAsync(c.coef, 0, sizeof(float) * (size_t)NB * c.C * c.BS, st));     // zero runs are not written by k_dgen
    hipLaunchKernelGGL(k_dgen, dim3((NB * c.C * 4 + 63) / 64), dim3(64), 0, st, c);
    if (ev) CK(hipEventRecord(ev[stage++], st));
    size_t lds = ulcx_dec_lds_bytes(c.BS, c.C);
    if (lds > 48 * 1024) CK(hipFuncSetAttribute((const void *)k_dimdct, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(k_dimdct, dim3(c.B), dim3(WG), lds, st, c);
    if (ev) CK(hipEventRecord(ev[stage++], st));
    CK(hipGetLastError());
    return ULCX_OK;
}

// ---------------------------------------------------------------------------
// Slots -> contiguous per-stream payloads (tools/ulcEncodeTool.c:160-169)
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_pack_streams(int nBlocks, int slotBytes, const uint8_t *slots, const int32_t *bits,
                                                       uint8_t *payload, long long stride, int32_t *payloadBytes, int32_t *maxBlock) {
    __shared__ int s_off, s_len;
    int s = blockIdx.x, tid = threadIdx.x;
    uint8_t *dst = payload + (size_t)s * stride;
    int off = 0, mx = 0;
    for (int k = 0; k < nBlocks; k++) {
        int n = (bits[(size_t)s * nBlocks + k] + 7) >> 3;
        const uint8_t *src = slots + ((size_t)s * nBlocks + k) * slotBytes;
        if ((size_t)off + n <= (size_t)stride) for (int i = tid; i < n; i += 256) dst[off + i] = src[i];
        off += n;
        mx = n > mx ? n : mx;
    }
    (void)s_off; (void)s_len;
    if (tid == 0) { payloadBytes[s] = off; if (maxBlock) maxBlock[s] = mx; }
}
int ulcx_pack_launch(int nStreams, int nBlocks, int slotBytes, const uint8_t *d_slots, const int32_t *d_bits, uint8_t *d_payload,
                     long long stride, int32_t *d_payloadBytes, int32_t *d_maxBlock, hipStream_t st) {
    hipLaunchKernelGGL(k_pack_streams, dim3(nStreams), dim3(256), 0, st, nBlocks, slotBytes, d_slots, d_bits, d_payload, stride, d_payloadBytes, d_maxBlock);
    CK(hipGetLastError());
    return ULCX_OK;
}
