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
#define DPS 4        // FFT array padding of k_dimdct (ulcx_fft.h; 3 measured no faster)
// Two blocks per trip with one wave per transform (four padded arrays) needs 4.25*BS floats of LDS for z instead of
// 2.5*BS (two arrays | dec | tmpq) and makes BlockSize 8192 stereo unsupported: measured 1.11 ms vs 1.13 ms for one block per
// trip at 5 workgroups per CU - kept off.
#define DIMDCT_PAIRS 0
#define DIMDCT_ZFLOATS(BS) (DIMDCT_PAIRS ? 4 * FFT_PADDEDS(BS, DPS) : (5 * (BS)) / 2)

// ---------------------------------------------------------------------------
__device__ __forceinline__ float expand_quantizer(int q) {        // ulcDecoder.c:96-98
    return 0x1.0p-31f * (float)((1u << (31 - 5)) >> q);
}
__device__ __forceinline__ uint32_t xorshift32(uint32_t s) {      // ulcDecoder.c:75-81
    s ^= s << 13; s ^= s >> 17; s ^= s << 5;
    return s;
}

// ---------------------------------------------------------------------------
// One whole code of the block syntax (FormatSpecs.md:57-141, ulcDecoder.c:99-197) decoded from a
// 32-bit window (>= 7 nybbles, low nybble first), with selects instead of a branch cascade:
// every lane of a wave executes the same instruction stream whatever its own code is.
//   plain  +-2..+-7          1 nybble   one coefficient
//   0h,X                     2          X+1 zeros
//   1h,Y,X                   3          YX+33 zeros
//   8h,Z,Y,X                 4          noise run: n = (ZY<<1 | X&1) + 16, level (X>>1)+1
//   Fh,X (X < Eh)            2          quantizer X
//   Fh,Eh,X (X < Fh)         3          quantizer Eh+X;  Fh,Eh,Fh = stop (zeros to the end)
//   Fh,Fh,Z,Y,X              5          noise to the end: level Z+1, decay YX
// A unit opens with a quantizer code without its Fh prefix (`first`); a leading Fh there (only a corrupt
// stream has one) gives the quantizer 0.0 the reference computes for it.
// ---------------------------------------------------------------------------
struct Code {
    int len;            // nybbles
    int n;              // coefficients consumed at once (1, or a zero run)
    int np;             // noise coefficients of a run (tail: the caller uses N)
    int l, dn, sv;      // noise level / tail decay / signed square of a plain coefficient
    int qnew;           // new quantizer index or -1
    bool plain, zrun, n8, tail, stop;
};
__device__ __forceinline__ Code decode_code(uint32_t w, bool first) {
    Code k;
    const bool q15 = first & ((w & 0xF) == 0xF);
    w = first ? ((w << 4) | 0xF) : w;
    const int v0 = w & 0xF, v1 = (w >> 4) & 0xF, v2 = (w >> 8) & 0xF, v3 = (w >> 12) & 0xF, v4 = (w >> 16) & 0xF;
    k.plain = (v0 != 0x0) & (v0 != 0x1) & (v0 != 0x8) & (v0 != 0xF);
    const bool z0 = (v0 == 0x0), z1 = (v0 == 0x1), esc = (v0 == 0xF);
    k.n8 = (v0 == 0x8);
    k.zrun = z0 | z1;
    k.tail = esc & (v1 == 0xF) & !q15;
    const bool qext = esc & (v1 == 0xE);
    k.stop = qext & (v2 == 0xF);
    const bool q1 = esc & !k.tail & !qext;
    int sv = (v0 ^ 0x8) - 0x8;
    k.sv = (sv < 0) ? (-sv * sv) : (+sv * sv);
    int len = k.plain ? 1 : z0 ? 2 : z1 ? 3 : k.n8 ? 4 : k.tail ? 5 : qext ? 3 : 2;
    k.len = len - (first ? 1 : 0);
    k.n = k.plain ? 1 : z0 ? v1 + 1 : z1 ? ((v1 << 4) | v2) + 33 : 0;
    k.np = k.n8 ? ((((v1 << 4) | v2) << 1) | (v3 & 1)) + 16 : 0;
    k.l = k.n8 ? (v3 >> 1) + 1 : v2 + 1;
    k.dn = (v3 << 4) | v4;
    // (opening Fh: the reference expands quantizer -2, ulcDecoder.c:89-98,107 - a shift by -2, i.e. by 30 on x86-64:
    //  the unit's quantizer is exactly 0 until a change code; index 30 expands to the same 0)
    k.qnew = q15 ? 30 : q1 ? v1 : (qext & !k.stop) ? 0xE + v2 : -1;
    return k;
}
// number of leading nybbles of w (low first, at most 7) that are plain coefficients, i.e. none of 0h 1h 8h Fh
__device__ __forceinline__ int plain_prefix(uint32_t w) {
    // a nybble is special iff its low three bits are all equal and ... : {0,1,8,F} = {0000,0001,1000,1111}
    // zero-nybble detector on w ^ pattern for each of the four values
    auto zn = [](uint32_t x) { return (x - 0x11111111u) & ~x & 0x88888888u; };       // bit 3 of every nybble that is 0 (exact for the lowest such nybble)
    uint32_t sp = zn(w) | zn(w ^ 0x11111111u) | zn(w ^ 0x88888888u) | zn(w ^ 0xFFFFFFFFu);
    sp |= 0x80000000u;                                   // the 8th nybble is not part of the window
    return (__ffs((int)sp) - 1) >> 2;                    // index of the first special nybble
}
typedef uint32_t u32_any_align __attribute__((aligned(1)));
// 32-bit window at bit position pos; bytes at or past readBytes read as 0
__device__ __forceinline__ uint32_t code_window(const uint8_t *p, int pos, int readBytes) {
    int b = pos >> 3;
    uint32_t w;
    if (b + 4 <= readBytes) w = *(const u32_any_align *)(p + b);
    else {
        w = 0;
        for (int i = 0; i < 4; i++) if (b + i < readBytes) w |= (uint32_t)p[b + i] << (8 * i);
    }
    return w >> (pos & 4);
}

// base pointer of block blk's bytes: its slot, or (packed mode) its parsed offset inside the stream payload
__device__ __forceinline__ const uint8_t *block_ptr(const UlcxDecCtx &c, int blk) {
    if (!c.packed) return c.in + (size_t)blk * c.slot;
    return c.in + (size_t)(blk / c.K) * c.payStride + c.blkOff[blk];
}

// Checkpoints: every unit is cut at the first code that starts at or after coefficient q*S/8
// (q = 0..7), so pass 3 can decode eight pieces of a unit on eight lanes.  {bit position, coefficients
// still to come, draws so far in the block, quantizer index (-1: the unit's opening code)}; N = 0 = no piece.
#define DCP_PER_UNIT 8

// Syntax walk of one block starting at p (limit = bits that may be consumed, readBytes = bytes that may be
// touched): records unit starts / draw counts / checkpoints.  Returns bits consumed (0 = corrupt).
// One flat loop, one code (or one checkpoint) per trip.
__device__ __forceinline__ int scan_block(const UlcxDecCtx &c, int blk, const uint8_t *p, int limit, int readBytes) {
    int pos = 0;
    int wc;
    {
        uint32_t w0 = (limit >= 8) ? code_window(p, 0, readBytes) : 0;     // ulcDecoder.c:211-216
        wc = w0 & 0xF;
        bool dec = (wc & 0x8) != 0;
        wc |= dec ? (int)(w0 & 0xF0) : (1 << 4);
        pos = dec ? 8 : 4;
    }
    unsigned pat = ulcx_pattern(wc);                                // (code 0000 behaves as one plain N/1 block, as in the reference)
    int nsub = 0; { unsigned q = pat; do nsub++; while (q >>= 4); }
    if ((c.BS >> (pat & 7)) == c.BS) nsub = 1;                      // ulcDecoder.c:242-245
    int total = c.C * nsub;
    if (nsub > 1) c.decList[atomicAdd(c.decCount, 1)] = blk;        // its units j >= 1 get their own (small) pass-3 launch
    int *ustart = c.unitStart + (size_t)blk * c.C * 4;
    int *udraw  = c.unitDraws + (size_t)blk * c.C * 4;
    int4 *cp = c.cp + (size_t)blk * c.C * 4 * DCP_PER_UNIT;
    int u = 0, draws = 0, qidx = 0, nextQ = 0, uslot = 0;
    ustart[0] = pos; udraw[0] = 0;
    int S = c.BS >> (pat & 7), N = S;
    bool first = true;
    bool fin = (limit < 16), bad = fin;
    while (!fin) {
        if (nextQ < DCP_PER_UNIT && (S - N) >= nextQ * (S >> 3)) {
            cp[uslot * DCP_PER_UNIT + nextQ] = make_int4(pos, N, draws, first ? -1 : qidx);
            nextQ++;
            continue;
        }
        const uint32_t w = code_window(p, pos, readBytes);
        if (!first) {
            // a run of plain coefficient nybbles (+-2..+-7) is consumed in one trip: the scan needs nothing
            // from them but their count.  Never across the next checkpoint position, the unit end or the
            // 7 nybbles the window holds.
            int m = plain_prefix(w);
            int room = (nextQ < DCP_PER_UNIT) ? nextQ * (S >> 3) - (S - N) : N;
            m = m < room ? m : room;
            m = m < N ? m : N;
            if (m > 0 && pos + 4 * m <= limit) {
                pos += 4 * m; N -= m;
                if (N > 0) continue;
                // (unit complete: fall through the common end-of-unit code below with a zero-length code)
                u++;
                fin = bad | (u >= total);
                if (!fin) {
                    int ch = u / nsub, j = u - ch * nsub;
                    uslot = ch * 4 + j;
                    ustart[uslot] = pos; udraw[uslot] = draws;
                    S = c.BS >> ((pat >> (4 * j)) & 7); N = S;
                    first = true; nextQ = 0;
                }
                continue;
            }
        }
        Code k = decode_code(w, first);
        const bool over = (k.zrun & (k.n > N)) | (k.n8 & (k.np > N));     // ulcDecoder.c:127,139,154
        const bool toEnd = k.stop | k.tail;
        const int used = over ? 0 : (toEnd ? N : k.n + k.np);
        draws += over ? 0 : (k.tail ? N : k.np);
        qidx = (k.qnew >= 0) ? k.qnew : qidx;
        pos += 4 * k.len;
        N -= used;
        bad |= over;
        first = false;
        if ((N == 0) | over) {
            u++;
            fin = bad | (u >= total);
            if (!fin) {
                int ch = u / nsub, j = u - ch * nsub;
                uslot = ch * 4 + j;
                ustart[uslot] = pos; udraw[uslot] = draws;
                S = c.BS >> ((pat >> (4 * j)) & 7); N = S;
                first = true; nextQ = 0;
            }
        }
        if (pos > limit) { bad = true; fin = true; }               // ran off the readable bytes: corrupt
    }
    bool ok = !bad;
    c.bits[blk] = ok ? pos : 0;
    c.wcScan[blk] = ok ? wc : 0;
    c.draws[blk] = draws;
    return ok ? pos : 0;
}

// Pass 1 — one lane per block: walk the syntax, record where each (channel, subblock)
// unit starts (nybble offset) and how many RNG draws precede it inside the block.
// The walk is one long dependent chain per lane, so a full wave per SIMD would sit at ~10 cycles per
// instruction: only DSCAN_LANES lanes of each wave are used, which puts 64/DSCAN_LANES times the waves
// (more chains) on every SIMD.
#define DSCAN_LANES 64
__global__ __launch_bounds__(64) void k_dscan(UlcxDecCtx c) {
    if (threadIdx.x >= DSCAN_LANES) return;
    int blk = blockIdx.x * DSCAN_LANES + threadIdx.x;
    if (blk >= c.B * c.K) return;
    scan_block(c, blk, c.in + (size_t)blk * c.slot, c.slot * 8 - 32, c.slot);
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
        if (!dead && off < avail) bits = scan_block(c, blk, base + off, (avail - off) * 8, avail - off);
        else { c.bits[blk] = 0; c.wcScan[blk] = 0; c.draws[blk] = 0; }
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

// Pass 2 — the RNG chain across blocks (ulcDecoder.c:75-81 keeps one seed for the life of the stream)
// and "a corrupt block ends the stream" (ulcDecodeTool.c:154-157).  The chain is linear, so block k's
// seed is the stream's seed jumped by the draws of blocks 0..k-1: one lane per BLOCK, each summing its
// predecessors' counts itself (K is small) - no stream-long serial walk.  The stream's state for the
// next call is staged (seedNext/deadNext) and committed by a second tiny kernel, because every block
// of the stream reads the old value here.
__global__ __launch_bounds__(64) void k_dseed(UlcxDecCtx c) {
    int blk = blockIdx.x * 64 + threadIdx.x;
    if (blk >= c.B * c.K) return;
    int s = blk / c.K, k = blk - s * c.K;
    int dead = c.dead[s];
    unsigned before = 0;
    for (int i = 0; i < k; i++) {
        if (c.wcScan[s * c.K + i] == 0) dead = 1;
        before += dead ? 0u : (unsigned)c.draws[s * c.K + i];
    }
    int wcMine = c.wcScan[blk];
    if (wcMine == 0) dead = 1;
    c.wc[blk] = dead ? 0 : wcMine;
    if (dead) c.bits[blk] = 0;
    uint32_t seed0 = c.seed[s];
    if (!dead) c.blockSeed[blk] = rng_jump(c.jump, seed0, before);
    if (k == c.K - 1) {
        unsigned total = before + (dead ? 0u : (unsigned)c.draws[blk]);
        c.seedNext[s] = rng_jump(c.jump, seed0, total);
        c.deadNext[s] = dead;
    }
}
__global__ __launch_bounds__(64) void k_dseed_commit(UlcxDecCtx c) {
    int s = blockIdx.x * 64 + threadIdx.x;
    if (s >= c.B) return;
    c.seed[s] = c.seedNext[s];
    c.dead[s] = c.deadNext[s];
}

// Pass 3 — dequantise + noise synthesis, one lane per PIECE of a (block, channel, subblock) unit:
// the scan cut every unit at eight checkpoints, so a lane's serial walk is ~1/8 of a unit and there are
// eight times the lanes (a wave runs as long as its longest lane, and whole units differ several-fold).
// ONE flat loop, no inner loops: a trip either decodes one whole code or emits up to four noise
// coefficients - a noise run is a state of the lane (with inner loops every lane of the wave waited for
// the longest run in flight: measured 460 k wave instructions per wave of 64 units).
__device__ __forceinline__ void dgen_piece(const UlcxDecCtx &c, int blk, int ch, int j, int q) {
    int wc = c.wc[blk];
    if (wc == 0) return;
    const int4 *cpU = c.cp + ((size_t)(blk * c.C + ch) * 4 + j) * DCP_PER_UNIT;
    int4 cp = cpU[q];
    int N = cp.y;
    if (N == 0) return;                                    // no such piece (unit ended before this checkpoint / no such unit)
    int Nstop = (q + 1 < DCP_PER_UNIT) ? cpU[q + 1].y : 0;
    if (N <= Nstop) return;                                // a long run jumped over this piece
    unsigned pat = ulcx_pattern(wc);
    int off = 0;
    for (int i = 0; i < j; i++) { off += c.BS >> (pat & 7); pat >>= 4; }
    const int S = c.BS >> (pat & 7);
    const uint8_t *src = block_ptr(c, blk);
    const int readBytes = c.packed ? (1 << 30) : c.slot;   // (a block the scan accepted is never read past its end)
    int bitpos = cp.x;
    uint32_t seed = rng_jump(c.jump, c.blockSeed[blk], (uint32_t)cp.z);
    float *dst = c.coef + (size_t)blk * c.C * c.BS + (size_t)ch * c.BS + off;       // pre-zeroed: zero runs just skip
    bool first = cp.w < 0;
    float quant = first ? 0.0f : expand_quantizer(cp.w);
    int pos = S - N, pend = 0;
    float lev = 0.0f, rr = 1.0f;
    int guard = 2 * c.slot + S + 64;                       // codes of a slot + noise coefficients: cannot be exceeded
    while (N > Nstop && guard-- > 0) {
        if (pend > 0) {
            // ulcDecoder.c:156-160 / :181-184: draw, flip on the MSB (cumulative), store, decay (rr = 1 for runs).
            // Up to the next multiple of four positions per trip, so the body of a long run goes out as
            // aligned 16-byte stores (this kernel is bound by the number of store requests, not by arithmetic).
            const int room = 4 - (pos & 3);
            const int cnt = pend < room ? pend : room;
            float v[4];
            if (cnt == 4) {                                // the body of a run: a full aligned group, nothing predicated
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    seed = xorshift32(seed);
                    lev = __uint_as_float(__float_as_uint(lev) ^ (seed & 0x80000000u));      // flip on the MSB
                    v[k] = lev;
                    lev *= rr;
                }
                *(float4 *)(dst + pos) = make_float4(v[0], v[1], v[2], v[3]);
            } else {
#pragma unroll
                for (int k = 0; k < 3; k++) {
                    if (k < cnt) {
                        seed = xorshift32(seed);
                        lev = __uint_as_float(__float_as_uint(lev) ^ (seed & 0x80000000u));
                        dst[pos + k] = lev;
                        lev *= rr;
                    }
                }
            }
            pos += cnt; pend -= cnt; N -= cnt;
        } else {
            const uint32_t w = code_window(src, bitpos, readBytes);
            if (!first) {
                // a run of plain coefficients (ulcDecoder.c:69-73): up to the next multiple of four positions in
                // one trip, an aligned 16-byte store when that is a full group
                int m = plain_prefix(w);
                const int room = 4 - (pos & 3);
                m = m < room ? m : room;
                m = m < N - Nstop ? m : N - Nstop;
                if (m > 0) {
                    float v[4];
#pragma unroll
                    for (int k = 0; k < 4; k++) {
                        int sv = (int)((w >> (4 * k)) & 0xF);
                        sv = (sv ^ 0x8) - 0x8;
                        sv = (sv < 0) ? (-sv * sv) : (+sv * sv);
                        v[k] = (float)sv * quant;
                    }
                    if (m == 4) *(float4 *)(dst + pos) = make_float4(v[0], v[1], v[2], v[3]);
                    else {
#pragma unroll
                        for (int k = 0; k < 3; k++) if (k < m) dst[pos + k] = v[k];
                    }
                    bitpos += 4 * m; pos += m; N -= m;
                    continue;
                }
            }
            Code k = decode_code(w, first);
            if (k.plain) dst[pos] = (float)k.sv * quant;                    // ulcDecoder.c:69-73
            int n = k.stop ? N : k.n;
            n = (n > N) ? N : n;                                            // (cannot happen in a block the scan accepted)
            int np = k.tail ? N : k.np;
            np = (np > N) ? N : np;
            // noise run / tail parameters (ulcDecoder.c:95-115, :123-137)
            const float lvl = (float)(k.l * k.l) * quant * (k.n8 ? (1.0f / 4) : (1.0f / 16));
            lev = (k.n8 | k.tail) ? lvl : lev;
            rr = k.tail ? 1.0f + (float)(k.dn * k.dn) * -0x1.0p-19f : (k.n8 ? 1.0f : rr);
            pend = np;
            quant = (k.qnew >= 0) ? expand_quantizer(k.qnew) : quant;      // ulcDecoder.c:89-98
            bitpos += 4 * k.len;
            pos += n; N -= n;
            first = false;
        }
    }
}
// Units j = 0 of every block: lane = (piece q, channel, block), block fastest.  Units j >= 1 exist only
// in decimated blocks: the first DGEN_DEC_WGS workgroups of the same launch walk the list the scan made
// (first, so they start early and run beside the rest instead of after it).
#define DGEN_DEC_WGS 1024
__global__ __launch_bounds__(256) void k_dgen(UlcxDecCtx c) {
    if (blockIdx.x < DGEN_DEC_WGS) {
        int n = *c.decCount;
        int per = 3 * c.C * DCP_PER_UNIT;
        for (long long t = (long long)blockIdx.x * 256 + threadIdx.x; t < (long long)n * per; t += (long long)DGEN_DEC_WGS * 256) {
            int e = (int)(t / per), r = (int)(t % per);
            int q = r % DCP_PER_UNIT, ch = (r / DCP_PER_UNIT) % c.C, j = 1 + r / (DCP_PER_UNIT * c.C);
            dgen_piece(c, c.decList[e], ch, j, q);
        }
        return;
    }
    int tid0 = (blockIdx.x - DGEN_DEC_WGS) * 256 + threadIdx.x;
    int NBd = c.B * c.K;
    if (tid0 >= NBd * c.C * DCP_PER_UNIT) return;
    int blk = tid0 % NBd, r = tid0 / NBd, ch = r % c.C, q = r / c.C;
    dgen_piece(c, blk, ch, 0, q);
}

// ---------------------------------------------------------------------------
// IMDCT + overlap-add.  LDS carve (floats): lap [C][BS/2] | z [4.25*BS] (four padded arrays of BS/2 complex;
// the general path uses the first BS floats as one array and keeps dec [BS] | tmpq [BS/2] behind it) | twl [BS/2]
// ---------------------------------------------------------------------------
// Output samples.  OUT = float: the C API's layout; OUT = int16_t: PCM16 output, converted on store
// exactly as the reference's WAV writer does (tools/WavIO_Helper.c:9-13,56-63: lrintf(clamp(x * 2^15, -32768, 32767))).
__device__ __forceinline__ int16_t to_pcm16(float x) {
    float v = x * 0x1.0p+15f;
    v = (v < -32768.0f) ? -32768.0f : (v > 32767.0f) ? 32767.0f : v;
    return (int16_t)__float2int_rn(v);
}
// (the decoded samples are written once and not read again by the decoder, the coefficients are read once: non-temporal)
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void st1(float *p, float a) { __builtin_nontemporal_store(a, p); }
__device__ __forceinline__ void st2(float *p, float a, float b) { f32x2 w = { a, b }; __builtin_nontemporal_store(w, (f32x2 *)p); }
__device__ __forceinline__ void st4(float *p, float a, float b, float d, float e) { f32x4 w = { a, b, d, e }; __builtin_nontemporal_store(w, (f32x4 *)p); }
__device__ __forceinline__ float2 ld2nt(const float *p) { f32x2 v = __builtin_nontemporal_load((const f32x2 *)p); return make_float2(v.x, v.y); }
__device__ __forceinline__ void st1(int16_t *p, float a) { *p = to_pcm16(a); }
__device__ __forceinline__ void st2(int16_t *p, float a, float b) { *(short2 *)p = make_short2(to_pcm16(a), to_pcm16(b)); }
__device__ __forceinline__ void st4(int16_t *p, float a, float b, float d, float e) { *(short4 *)p = make_short4(to_pcm16(a), to_pcm16(b), to_pcm16(d), to_pcm16(e)); }
template <typename OUT> __device__ __forceinline__ OUT *out_base(const UlcxDecCtx &c);
template <> __device__ __forceinline__ float *out_base<float>(const UlcxDecCtx &c) { return c.pcm; }
template <> __device__ __forceinline__ int16_t *out_base<int16_t>(const UlcxDecCtx &c) { return c.pcm16; }

template <typename OUT>
__global__ __launch_bounds__(WG) void k_dimdct(UlcxDecCtx c) {
    extern __shared__ float lds[];
    const int BS = c.BS, C = c.C, H2 = BS / 2;
    int s = blockIdx.x, tid = threadIdx.x;
    float  *lap   = lds;
    float2 *z     = (float2 *)(lap + (size_t)C * H2);
    float  *dec   = (float *)z + BS;
    float  *tmpq  = dec + BS;
    float2 *twl   = (float2 *)((float *)z + DIMDCT_ZFLOATS(BS));   // BS/4 complex: FFT twiddles of the full-size transform
    float *glap = c.lap + (size_t)s * C * H2;
    for (int i = tid; i < C * H2; i += WG) lap[i] = glap[i];
    for (int i = tid; i < BS / 4; i += WG) twl[i] = c.T.tw[0][i];
    int lastSub = c.lastSub[s];
    __syncthreads();

    // stereo, un-decimated block (the common case): both channels at once, inverse M/S in registers,
    // interleaved stores.  Post-twiddle fused with the windowed overlap (oracle/orc_fourier.c orc_imdct).
    auto fast_block = [&](int wcv) { return C == 2 && wcv != 0 && (BS >> (ulcx_pattern(wcv) & 7)) == BS; };
    auto fast_overlap = [&](int wcv, int last) {
        int ov = BS;                                                // ulcDecoder.c:234-239
        if (ulcx_pattern(wcv) & 8) ov >>= (wcv & 7);
        return ov > last ? last : ov;
    };
    auto fast_pre = [&](const float *coefB, float2 *za, float2 *zb, bool padded) {
        const int S = BS, M = BS >> 1;
        const float2 *pre = c.T.pre[0];
        const float *X0 = coefB, *X1 = coefB + BS;
        // n and M-1-n together: their four inputs are the two aligned pairs (X[2n], X[2n+1]) and (X[S-2-2n], X[S-1-2n])
        for (int n = tid; n < M / 2; n += WG) {
            const int n2 = M - 1 - n;
            float2 P = pre[n], P2 = pre[n2];
            int pn = padded ? FFT_PADS(n, DPS) : n, pn2 = padded ? FFT_PADS(n2, DPS) : n2;
            float2 a0 = ld2nt(X0 + 2 * n), b0 = ld2nt(X0 + S - 2 - 2 * n);
            float2 a1 = ld2nt(X1 + 2 * n), b1 = ld2nt(X1 + S - 2 - 2 * n);
            za[pn]  = cmulc(make_float2(a0.x, b0.y), P);
            za[pn2] = cmulc(make_float2(b0.x, a0.y), P2);
            zb[pn]  = cmulc(make_float2(a1.x, b1.y), P);
            zb[pn2] = cmulc(make_float2(b1.x, a1.y), P2);
        }
    };
    auto fast_post = [&](const float2 *z0, const float2 *z1, OUT *outp, int ov, bool padded) {
        const int S = BS, M = BS >> 1;
        const float2 *pre = c.T.pre[0];
        int a = (S - ov) >> 1;
        const float *fall = c.T.winFall + ov, *rise = c.T.winRise + ov;
        int bits = 31 - __clz(M);
        float *L0 = lap, *L1 = lap + H2;
        for (int kk = tid; kk < M / 2; kk += WG) {
            int k1 = kk, k2 = M - 1 - kk;
            int r1 = (int)(__brev((unsigned)k1) >> (32 - bits));
            int r2 = (int)(__brev((unsigned)k2) >> (32 - bits));
            if (padded) { r1 = FFT_PADS(r1, DPS); r2 = FFT_PADS(r2, DPS); }
            float2 P1 = pre[k1], P2 = pre[k2];
            float2 ya1 = cmulc(z0[r1], P1), ya2 = cmulc(z0[r2], P2);     // channel 0 (M)
            float2 yb1 = cmulc(z1[r1], P1), yb2 = cmulc(z1[r2], P2);     // channel 1 (S)
            float A0m = L0[2 * k1], A1m = L0[2 * k1 + 1], A0s = L1[2 * k1], A1s = L1[2 * k1 + 1];
            float Bm[2] = { -ya1.y, ya2.x }, Bs[2] = { -yb1.y, yb2.x };
            float Am[2] = { A0m, A1m }, As[2] = { A0s, A1s };
            int   pv[2] = { M - 1 - 2 * k1, M - 2 - 2 * k1 };
            float2 lo2[2], hi2[2];                                         // interleaved L/R at positions p and S-1-p
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
                lo2[q] = make_float2(mLo + sLo, mLo - sLo);
                hi2[q] = make_float2(mHi + sHi, mHi - sHi);
            }
            // positions pv[1] = pv[0]-1 and S-1-pv[0], S-pv[0] are neighbours: two aligned 16-byte stores
            st4(outp + 2 * pv[1], lo2[1].x, lo2[1].y, lo2[0].x, lo2[0].y);
            st4(outp + 2 * (S - 1 - pv[0]), hi2[0].x, hi2[0].y, hi2[1].x, hi2[1].y);
            L0[2 * k1] = ya1.x; L0[2 * k1 + 1] = -ya2.y;
            L1[2 * k1] = yb1.x; L1[2 * k1 + 1] = -yb2.y;
        }
    };

    for (int k = 0; k < c.K; k++) {
        int blk = s * c.K + k;
        int wc = c.wc[blk];
        OUT *outp = out_base<OUT>(c) + (size_t)blk * C * BS;
        if (wc == 0) {                                              // corrupt block / dead stream
            for (int i = tid; i < C * BS; i += WG) st1(outp + i, 0.0f);
            continue;
        }
        const float *coefB = c.coef + (size_t)blk * C * BS;
        int newLast = lastSub;
        if (fast_block(wc)) {
            const int M = BS >> 1, Mp = FFT_PADDEDS(M, DPS);
            int wc2 = (k + 1 < c.K) ? c.wc[blk + 1] : 0;
            if (DIMDCT_PAIRS && fast_block(wc2)) {
                // ---- two consecutive blocks: four transforms, ONE WAVE PER ARRAY (16 points per lane through
                //      four radix-2 stages in registers, no barrier between passes), then the two overlap-adds in order
                if (!(c.dbgSkip & 1)) { fast_pre(coefB, z, z + Mp, true);
                fast_pre(coefB + (size_t)C * BS, z + 2 * Mp, z + 3 * Mp, true); }
                __syncthreads();
                if (!(c.dbgSkip & 2)) fft_wave_dif(z + __builtin_amdgcn_readfirstlane(tid >> 6) * Mp, M, twl, tid & 63, DPS);
                __syncthreads();
                if (!(c.dbgSkip & 4)) fast_post(z, z + Mp, outp, fast_overlap(wc, lastSub), true);
                __syncthreads();
                if (!(c.dbgSkip & 4)) fast_post(z + 2 * Mp, z + 3 * Mp, outp + (size_t)C * BS, fast_overlap(wc2, BS), true);
                __syncthreads();
                lastSub = BS;
                k++;
                continue;
            }
            fast_pre(coefB, z, z + M, false);
            __syncthreads();
            fftn_dif(z, 2, M, twl, tid);
            fast_post(z, z + M, outp, fast_overlap(wc, lastSub), false);
            __syncthreads();
            lastSub = BS;
            continue;
        }
        for (int ch = 0; ch < C; ch++) {
            int last = lastSub;                                     // ulcDecoder.c:219
            // the channel's time samples are staged over its own (already consumed) coefficients in global memory:
            // keeps 2*BS floats out of LDS (occupancy of the common path) - only mono/odd/decimated blocks come here
            float *dst = c.coef + (size_t)blk * C * BS + (size_t)ch * BS;
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
                        float m = dst[(ptrdiff_t)n - BS], sd = dst[n];                 // ulcDecoder.c:281-289
                        float l = m + sd, r = m - sd;
                        if (C == 2) st2(outp + 2 * n, l, r);
                        else { st1(outp + (size_t)n * C + ch - 1, l); st1(outp + (size_t)n * C + ch, r); }
                    }
                } else {
                    for (int n = tid; n < BS; n += WG) st1(outp + (size_t)n * C + ch, dst[n]);
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
    return sizeof(float) * ((size_t)C * (BS / 2) + (size_t)DIMDCT_ZFLOATS(BS) + BS / 2);
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { ulcx_set_error("%s: %s", #x, hipGetErrorString(e_)); return ULCX_ERR_HIP; } } while (0)

int ulcx_dec_launch(const UlcxDecCtx &c, hipStream_t st, hipEvent_t *ev, hipStream_t side, hipEvent_t evFork, hipEvent_t evSide) {
    int stage = 0;
    if (ev) CK(hipEventRecord(ev[stage++], st));
    int NB = c.B * c.K;
    // zero runs are not written by pass 3: the coefficient buffer is cleared beside the (latency-bound) scan
    if (side) {
        CK(hipEventRecord(evFork, st));
        CK(hipStreamWaitEvent(side, evFork, 0));
        CK(hipMemsetAsync(c.coef, 0, sizeof(float) * (size_t)NB * c.C * c.BS, side));
        CK(hipEventRecord(evSide, side));
    }
    CK(hipMemsetAsync(c.cp, 0, sizeof(int4) * (size_t)NB * c.C * 4 * DCP_PER_UNIT, st));   // N = 0: no piece
    CK(hipMemsetAsync(c.decCount, 0, sizeof(int), st));
    if (c.packed) hipLaunchKernelGGL(k_dscan_packed, dim3((c.B + 63) / 64), dim3(64), 0, st, c);
    else hipLaunchKernelGGL(k_dscan, dim3((NB + DSCAN_LANES - 1) / DSCAN_LANES), dim3(64), 0, st, c);
    if (ev) CK(hipEventRecord(ev[stage++], st));
    hipLaunchKernelGGL(k_dseed, dim3((NB + 63) / 64), dim3(64), 0, st, c);
    hipLaunchKernelGGL(k_dseed_commit, dim3((c.B + 63) / 64), dim3(64), 0, st, c);
    if (ev) CK(hipEventRecord(ev[stage++], st));
    if (!side) CK(hipMemsetAsync(c.coef, 0, sizeof(float) * (size_t)NB * c.C * c.BS, st));
    if (side) CK(hipStreamWaitEvent(st, evSide, 0));                                   // the coefficient buffer is zeroed
    hipLaunchKernelGGL(k_dgen, dim3(DGEN_DEC_WGS + (NB * c.C * DCP_PER_UNIT + 255) / 256), dim3(256), 0, st, c);
    if (ev) CK(hipEventRecord(ev[stage++], st));
    size_t lds = ulcx_dec_lds_bytes(c.BS, c.C);
    if (lds > 48 * 1024) {
        CK(hipFuncSetAttribute((const void *)k_dimdct<float>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        CK(hipFuncSetAttribute((const void *)k_dimdct<int16_t>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    }
    if (c.pcm16) hipLaunchKernelGGL(k_dimdct<int16_t>, dim3(c.B), dim3(WG), lds, st, c);
    else hipLaunchKernelGGL(k_dimdct<float>, dim3(c.B), dim3(WG), lds, st, c);
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
