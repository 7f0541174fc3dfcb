// ulcx_dec.hip — batched ulc-codec decoder for gfx950 (MI355X), hand-written HIP.
//
// Two kernels per call; dequantised coefficients never exist in HBM:
//   k_dscan   one lane per block: the serial syntax walk (libulc/ulcDecoder.c:99-197, syntax
//             FormatSpecs.md:57-141).  It is the one thing that cannot be done in parallel, so it does
//             nothing but walk and take notes: 8-byte records of the runs of coded coefficients and of
//             the noise runs (about 1 KB per block instead of 16 KB of coefficients), RNG draw counts,
//             and the decaying-noise level chain of each unit's tail.
//   k_dsyn    stereo streams: one workgroup (2 waves) per stream, blocks in order, one wave per channel:
//             the records are scattered STRAIGHT INTO the FFT's LDS arrays, noise runs are synthesised
//             32 coefficients per lane from a jumped-ahead xorshift state (top bits by parity masks), then
//             IMDCT (one DCT-IV = complex FFT per channel, one wave per array, no barrier between the
//             passes), sine-window overlap-add, the centring of decimated subblocks on one timeline
//             (the reference's reversed-time FIFO without the shifting: dec_time_wave), inverse M/S,
//             interleave (libulc/ulcDecoder.c:198-302; IMDCT per FormatSpecs.md:150-157).
//   k_dgen    every other geometry (mono, multichannel, BlockSize > 4096): same pieces, one array.
// Compiled with -ffp-contract=off (see ulcx_enc.hip).
#include "ulcx_internal.h"

#define WG 128
#define FFT_PACKED
#include "ulcx_fft.h"
#ifndef DSYN_C2048
#define DSYN_C2048 1
#endif
#ifndef DSYN_C4096
#define DSYN_C4096 1   // BlockSize 4096 with the size as a compile-time constant too (the pipelined epilogue; round 5)
#endif
#define DSYN_TWL ULCX_DSYN_TWL
#ifndef DSYN_EPI2
#define DSYN_EPI2 1  // headline geometry: the epilogue's global operands fetched one trip ahead (0: the generic loop)
#endif
#ifndef DPS
#define DPS 4        // FFT array padding (ulcx_fft.h): one complex after every 16 (3: after every 8 - conflict-free passes, 1 KB more LDS)
#endif

// ---------------------------------------------------------------------------
__device__ __forceinline__ float expand_quantizer(int q) {        // ulcDecoder.c:96-98
    return 0x1.0p-31f * (float)((1u << (31 - 5)) >> q);
}
__host__ __device__ constexpr uint32_t xorshift32(uint32_t s) {    // ulcDecoder.c:75-81
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
    int plain, zrun, n8, tail, stop;     // 0 / 1
};
// Classification by bit tests on constants indexed with the nybble, and arithmetic on the 0/1 results: written with
// comparisons (v0 == 0, == 1, == 8, == Fh ...) the compiler recognises a switch and lowers it to a tree of branches with
// EXEC-mask bookkeeping - in a kernel whose every instruction costs a wave ~9 cycles.
__device__ __forceinline__ Code decode_code(uint32_t w, bool first) {
    Code k;
    const int f = first ? 1 : 0;
    const int q15 = f & (int)(((w & 0xF) + 1) >> 4);
    w = first ? ((w << 4) | 0xF) : w;
    const int v0 = w & 0xF, v1 = (w >> 4) & 0xF, v2 = (w >> 8) & 0xF, v3 = (w >> 12) & 0xF, v4 = (w >> 16) & 0xF;
    const int z0 = (0x0001 >> v0) & 1, z1 = (0x0002 >> v0) & 1, esc = (0x8000 >> v0) & 1;
    k.n8 = (0x0100 >> v0) & 1;
    k.plain = (0x7EFC >> v0) & 1;
    k.zrun = (0x0003 >> v0) & 1;
    k.tail = esc & ((v1 + 1) >> 4) & (q15 ^ 1);
    const int qext = esc & ((0x4000 >> v1) & 1);
    k.stop = qext & ((v2 + 1) >> 4);
    const int q1 = esc & (k.tail ^ 1) & (qext ^ 1);
    const int sgn = (v0 ^ 0x8) - 0x8;
    const int sq = sgn * sgn;
    k.sv = (sgn < 0) ? -sq : sq;
    // nybbles: 1 plain, 2 short zero run, 3 long zero run, 4 noise run; Fh: 2, +1 quantizer extension / stop, +3 tail
    const int len0 = (int)((0x2111111411111132ull >> (4 * v0)) & 0xF);
    k.len = len0 + 3 * k.tail + qext - f;
    const int v12 = (v1 << 4) | v2;
    k.n = k.plain + z0 * (v1 + 1) + z1 * (v12 + 33);
    k.np = k.n8 * (((v12 << 1) | (v3 & 1)) + 16);
    k.l = (v2 + 1) + k.n8 * ((v3 >> 1) - v2);
    k.dn = (v3 << 4) | v4;
    // (opening Fh: the reference expands quantizer -2, ulcDecoder.c:89-98,107 - a shift by -2, i.e. by 30 on x86-64:
    //  the unit's quantizer is exactly 0 until a change code; index 30 expands to the same 0)
    const int qn = -1 + q1 * (v1 + 1) + (qext & (k.stop ^ 1)) * (0xE + v2 + 1);
    k.qnew = q15 ? 30 : qn;
    return k;
}
// number of leading nybbles of w (low first, at most 7) that are plain coefficients, i.e. none of 0h 1h 8h Fh
__device__ __forceinline__ int plain_prefix(uint32_t w) {
    auto zn = [](uint32_t x) { return (x - 0x11111111u) & ~x & 0x88888888u; };       // bit 3 of every nybble that is 0 (exact for the lowest such nybble)
    uint32_t sp = zn(w) | zn(w ^ 0x11111111u) | zn(w ^ 0x88888888u) | zn(w ^ 0xFFFFFFFFu);
    sp |= 0x80000000u;                                   // the 8th nybble is not part of the window
    return (__ffs((int)sp) - 1) >> 2;                    // index of the first special nybble
}
// ---------------------------------------------------------------------------
// The scan's view of the stream: 16-byte aligned chunks kept in registers, the next one always in flight, so a
// trip of the walk waits for memory once per 32 nybbles instead of once per code.
// ---------------------------------------------------------------------------
struct NybWin {
    const uint8_t *p16;          // 16-byte aligned address at or below the block's first byte
    int a;                       // bytes between p16 and the block's first byte
    int readBytes;               // block bytes that may be used (later ones read as 0)
    const uint8_t *bufBeg, *bufEnd;   // the caller's whole input: nothing outside it is touched
    uint32_t c0, c1, c2, c3, n0, n1, n2, n3; int chunk;       // current and next 16-byte chunk (scalars: a struct member picked
                                                              // by a run-time index makes the compiler move the window to LDS)

    __device__ __forceinline__ void load_chunk(int ci, uint32_t &o0, uint32_t &o1, uint32_t &o2, uint32_t &o3) const {
        const uint8_t *q = p16 + (ptrdiff_t)ci * 16;
        const int rel = ci * 16 - a;                       // block-relative offset of the chunk's first byte
        uint32_t d0, d1, d2, d3;
        if (q >= bufBeg && q + 16 <= bufEnd) { const uint4 v = *(const uint4 *)q; d0 = v.x; d1 = v.y; d2 = v.z; d3 = v.w; }
        else {
            auto word = [&](int j) { uint32_t w = 0; for (int t = 0; t < 4; t++) { const uint8_t *b = q + 4 * j + t; if (b >= bufBeg && b < bufEnd) w |= (uint32_t)*b << (8 * t); } return w; };
            d0 = word(0); d1 = word(1); d2 = word(2); d3 = word(3);
        }
        if (rel + 16 > readBytes) {
            auto mask = [&](int j) { const int keep = readBytes - (rel + 4 * j); return keep >= 4 ? 0xFFFFFFFFu : keep <= 0 ? 0u : ((1u << (8 * keep)) - 1u); };
            d0 &= mask(0); d1 &= mask(1); d2 &= mask(2); d3 &= mask(3);
        }
        o0 = d0; o1 = d1; o2 = d2; o3 = d3;
    }
    __device__ __forceinline__ void init(const uint8_t *p, int rb, const uint8_t *b0, const uint8_t *b1) {
        a = (int)((uintptr_t)p & 15); p16 = p - a; readBytes = rb; bufBeg = b0; bufEnd = b1;
        chunk = 0; load_chunk(0, c0, c1, c2, c3); load_chunk(1, n0, n1, n2, n3);
    }
    __device__ __forceinline__ void advance(int q) {
        if ((q >> 5) > chunk) { c0 = n0; c1 = n1; c2 = n2; c3 = n3; chunk++; load_chunk(chunk + 1, n0, n1, n2, n3); }
    }
    // window at bit position pos of the block (positions only ever advance, by at most 12 nybbles per call)
    __device__ __forceinline__ uint32_t at(int pos) {
        const int q = (pos >> 2) + 2 * a;
        advance(q);
        const int di = (q >> 3) & 3, sh = (q & 7) * 4;
        uint32_t x0 = c0, x1 = c1, x2 = c2, x3 = c3, x4 = n0;
        asm volatile("" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4));
        const uint32_t lo = di == 0 ? x0 : di == 1 ? x1 : di == 2 ? x2 : x3;
        const uint32_t hi = di == 0 ? x1 : di == 1 ? x2 : di == 2 ? x3 : x4;
        return __builtin_amdgcn_alignbit(hi, lo, sh);
    }
    // the same, 64 bits wide: a trip's run of plain nybbles (<= 28 bits) and the code behind it (<= 20 bits) from one look
    __device__ __forceinline__ uint64_t at64(int pos) {
        const int q = (pos >> 2) + 2 * a;
        advance(q);
        const int di = (q >> 3) & 3, sh = (q & 7) * 4;
        // (values first, opaque to the optimiser, then selects: a select between loads of struct members becomes a load from
        //  a selected address, and a window that is indexed at run time is moved to LDS - one LDS round trip per trip)
        uint32_t x0 = c0, x1 = c1, x2 = c2, x3 = c3, x4 = n0, x5 = n1;
        asm volatile("" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5));
        const uint32_t d0 = di == 0 ? x0 : di == 1 ? x1 : di == 2 ? x2 : x3;
        const uint32_t d1 = di == 0 ? x1 : di == 1 ? x2 : di == 2 ? x3 : x4;
        const uint32_t d2 = di == 0 ? x2 : di == 1 ? x3 : di == 2 ? x4 : x5;
        return (uint64_t)__builtin_amdgcn_alignbit(d1, d0, sh) | ((uint64_t)__builtin_amdgcn_alignbit(d2, d1, sh) << 32);
    }
};

// The same view for a lane whose every look stays inside the caller's buffer (k_dscan checks that per wave: all but the
// waves at the buffer's ends), without NybWin's stall: there the load sits in a branch and its result is copied into the
// loop-carried registers right behind it, so the compiler waits for it on the spot (`s_waitcnt vmcnt(0)`, records stored
// since included) - a memory round trip in almost every trip of a wave, some lane of 64 always being at a chunk's end.
// Here every look issues ONE load into registers nothing touches before the next look (a trip of the walk is ~1500
// cycles of dependent arithmetic: a chunk's latency): a lane that moved on to its next chunk fetches the chunk two
// ahead, every other lane reads one common address (one line for all of them), and the next look takes the value over
// with a select.  The shift to the next chunk is a select too.
// Bytes behind readBytes are not masked: limit = 8 readBytes in every caller, a code that uses a nybble behind it ends
// behind limit, and the walk calls that block corrupt whatever the nybble was.
struct NybWinFast {
    const uint4 *p16, *common; int a; int chunk; bool gNew;
    uint32_t c0, c1, c2, c3, n0, n1, n2, n3, f0, f1, f2, f3, g0, g1, g2, g3;
    __device__ __forceinline__ void init(const uint8_t *p, int, const uint8_t *, const uint8_t *) {
        a = (int)((uintptr_t)p & 15); p16 = (const uint4 *)(p - a); chunk = 0; gNew = false;
        common = (const uint4 *)(((uintptr_t)__builtin_amdgcn_readfirstlane((int)((uintptr_t)p16 >> 32)) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane((int)(uintptr_t)p16));
        const uint4 v0 = p16[0], v1 = p16[1], v2 = p16[2];
        c0 = v0.x; c1 = v0.y; c2 = v0.z; c3 = v0.w; n0 = v1.x; n1 = v1.y; n2 = v1.z; n3 = v1.w; f0 = v2.x; f1 = v2.y; f2 = v2.z; f3 = v2.w;
        g0 = g1 = g2 = g3 = 0;
    }
    __device__ __forceinline__ uint64_t at64(int pos) {
        const int q = (pos >> 2) + 2 * a;
        f0 = gNew ? g0 : f0; f1 = gNew ? g1 : f1; f2 = gNew ? g2 : f2; f3 = gNew ? g3 : f3;      // the last look's load
        const bool adv = (q >> 5) > chunk;                   // (a trip moves at most 12 nybbles: one chunk at a time)
        c0 = adv ? n0 : c0; c1 = adv ? n1 : c1; c2 = adv ? n2 : c2; c3 = adv ? n3 : c3;
        n0 = adv ? f0 : n0; n1 = adv ? f1 : n1; n2 = adv ? f2 : n2; n3 = adv ? f3 : n3;
        chunk += adv ? 1 : 0;
        const uint4 v = *(adv ? p16 + (chunk + 2) : common);
        g0 = v.x; g1 = v.y; g2 = v.z; g3 = v.w; gNew = adv;
        const int di = (q >> 3) & 3, sh = (q & 7) * 4;
        uint32_t x0 = c0, x1 = c1, x2 = c2, x3 = c3, x4 = n0, x5 = n1;
        asm volatile("" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5));
        const uint32_t d0 = di == 0 ? x0 : di == 1 ? x1 : di == 2 ? x2 : x3;
        const uint32_t d1 = di == 0 ? x1 : di == 1 ? x2 : di == 2 ? x3 : x4;
        const uint32_t d2 = di == 0 ? x2 : di == 1 ? x3 : di == 2 ? x4 : x5;
        return (uint64_t)__builtin_amdgcn_alignbit(d1, d0, sh) | ((uint64_t)__builtin_amdgcn_alignbit(d2, d1, sh) << 32);
    }
    __device__ __forceinline__ uint32_t at(int pos) { return (uint32_t)at64(pos); }
};

// Where the walk's records go (k_dscan).  Every lane appends 8-byte records to the row of ITS block: stored one by one
// that is up to 64 partial lines per store instruction and two such instructions per trip - more than half of the
// kernel's time, and three times the records' bytes in HBM traffic (partly written lines evicted and fetched again).
// Instead a lane parks its records in an LDS ring (row r of the ring holds the records number r mod R of every lane, rotated
// by r so that one lane's consecutive records lie in different banks), and when some lane's ring is nearly full the
// wave writes out all of them: two owners per trip, a half-wave per owner, the owner's pending records to consecutive
// addresses of its row.
#define DSCAN_RP 24              // plain-run records a lane may have pending (a trip adds at most two)
#define DSCAN_RN 12              // noise records (at most one per trip)
#define DSCAN_WAVE_SYNC() do { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); } while (0)
template <int R>
__device__ __forceinline__ void ring_put(uint2 *ring, int lane, int slotR, uint2 rec) { ring[slotR * 64 + ((lane + slotR) & 63)] = rec; }
// f .. n-1: this lane's pending records; rowIdx: its block; base + rowIdx * stride: the block's row (stride records)
template <int R>
__device__ __forceinline__ void ring_flush(uint2 *ring, int lane, int &f, int n, uint2 *base, int rowIdx, int stride) {
    DSCAN_WAVE_SYNC();
    const int cnt = n - f, half = lane >> 5, i = lane & 31;
    const unsigned long long pend = __ballot(cnt > 0);       // (a wave with few live lanes - the drop-in's one block - skips the rest)
#pragma unroll 4
    for (int it = 0; it < 32; it++) {
        if (!((pend >> (2 * it)) & 3ull)) continue;
        const int j = 2 * it + half;
        const int cj = __builtin_amdgcn_ds_bpermute(j << 2, cnt);
        const int fj = __builtin_amdgcn_ds_bpermute(j << 2, f);
        const int rj = __builtin_amdgcn_ds_bpermute(j << 2, rowIdx);
        const int idx = fj + i;
        const int r = idx % R;
        if (i < cj && idx < stride) base[(size_t)rj * stride + idx] = ring[r * 64 + ((j + r) & 63)];
    }
    f = n;
    DSCAN_WAVE_SYNC();
}

// Syntax walk of one block starting at p (limit = bits that may be consumed, readBytes = bytes that may be
// used).  The walk is the one thing that cannot be done in parallel; everything it learns on the way is left for the
// synthesis in a form that can: per (channel, subblock) unit
//   * PLAIN-RUN records, 8 bytes: up to seven consecutive coded coefficients {first coefficient | quantizer index << 15 |
//     count << 20, their nybbles};
//   * NOISE records, 8 bytes (the layout synth_noise reads): {first coefficient | count << 16 | tail << 31,
//     draws made in the unit before the run | level << 16 | quantizer index << 21};
//   * the unit's draws-so-far, its decaying-noise tail's parameters and (after the walk) the tail's level chain;
// and bits / WindowCtrl / draws of the block.  Returns bits consumed (0 = corrupt).  One flat loop, one code (or one
// run of plain coefficients) per trip.
// RING: the records through the wave's LDS rings (every lane of the wave is here, `live` or not, and stays to the end);
// else stored one by one (lanes come and go as they please: k_dscan_packed).
template <typename WIN = NybWin, bool RING = false>
__device__ __forceinline__ int scan_block(const UlcxDecCtx &c, int blk, const uint8_t *p, int limit, int readBytes,
                                          const uint8_t *bufBeg, const uint8_t *bufEnd, bool live = true, uint2 *ringP = nullptr, uint2 *ringN = nullptr) {
    const int rlane = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
    int fP = 0, fN = 0, wP = 0, wN = 0;                             // RING: records written out so far, ring rows of the next records
    WIN win; win.init(p, readBytes, bufBeg, bufEnd);
    int pos = 0;
    int wc;
    {
        uint32_t w0 = (limit >= 8) ? win.at(0) : 0;                 // ulcDecoder.c:211-216
        wc = w0 & 0xF;
        bool dec = (wc & 0x8) != 0;
        wc |= dec ? (int)(w0 & 0xF0) : (1 << 4);
        pos = dec ? 8 : 4;
    }
    unsigned pat = ulcx_pattern(wc);                                // (code 0000 behaves as one plain N/1 block, as in the reference)
    int nsub = 0; { unsigned q = pat; do nsub++; while (q >>= 4); }
    if ((c.BS >> (pat & 7)) == c.BS) nsub = 1;                      // ulcDecoder.c:242-245
    int total = c.C * nsub;
    int *udraw  = c.unitDraws + (size_t)blk * c.C * 4;
    int4 *urec = c.unitRec + (size_t)blk * c.C * 4;
    float4 *utail = c.unitTail + (size_t)blk * c.C * 4;
    uint2 *prec = c.prec + (size_t)blk * c.precStride;
    uint2 *nrec = c.nrec + (size_t)blk * c.nrecStride;
    int nP = 0, nN = 0, uP0 = 0, uN0 = 0;                           // records written so far in the block / at the current unit's start
    auto put_prec = [&](uint2 rec) {
        if (RING) { ring_put<DSCAN_RP>(ringP, rlane, wP, rec); wP = (wP + 1 == DSCAN_RP) ? 0 : wP + 1; }
        else if (nP < c.precStride) prec[nP] = rec;
        nP++;
    };
    auto put_nrec = [&](uint2 rec) {
        if (RING) { ring_put<DSCAN_RN>(ringN, rlane, wN, rec); wN = (wN + 1 == DSCAN_RN) ? 0 : wN + 1; }
        else if (nN < c.nrecStride) nrec[nN] = rec;
        nN++;
    };
    int u = 0, draws = 0, uslot = 0, uDraw0 = 0, uj = 0, uch4 = 0;
    if (live) { udraw[0] = 0; utail[0] = make_float4(0.0f, 0.0f, 0.0f, 0.0f); }
    int S = c.BS >> (pat & 7), N = S;
    bool first = true;
    bool fin = (limit < 16) | !live, bad = fin;
    auto next_unit = [&]() {
        urec[uslot] = make_int4(uP0, nP - uP0, uN0, nN - uN0);
        u++;
        fin = bad | (u >= total);
        if (!fin) {
            uj++;                                                  // (subblock within the channel, channel * 4: counters, not a division by nsub)
            if (uj == nsub) { uj = 0; uch4 += 4; }
            const int j = uj;
            uslot = uch4 + j;
            udraw[uslot] = draws; uDraw0 = draws;
            utail[uslot] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
            uP0 = nP; uN0 = nN;
            S = c.BS >> ((pat >> (4 * j)) & 7); N = S;
            first = true;
        }
    };
    int qidx = 30;                                                  // (index 30 expands to 0.0; every unit opens with a quantizer code)
    // One trip = a run of plain coefficient nybbles (possibly empty), then ONE other code: every lane does both parts every
    // trip, so lanes that alternate between the two kinds (the usual stream) do not wait for each other's other half.
    for (;;) {
        if (RING) { if (!__any(!fin)) break; } else if (fin) break;
        if (!fin) {
        const uint64_t w64 = win.at64(pos);
        uint32_t w = (uint32_t)w64;
        // 1. plain coefficients (+-2..+-7), up to the seven the window holds.  Never across the unit end or the block's bits;
        //    not at a unit's opening code.
        int m = first ? 0 : plain_prefix(w);
        m = m < N ? m : N;
        m = (pos + 4 * m <= limit) ? m : 0;
        if (m > 0) {
            put_prec(make_uint2((uint32_t)(S - N) | ((uint32_t)qidx << 15) | ((uint32_t)m << 20), w & (0xFFFFFFFFu >> (32 - 4 * m))));
            pos += 4 * m; N -= m;
            if (N == 0) next_unit();
            w = (uint32_t)(w64 >> (4 * m));
        }
        if (!fin) {
        // 2. one code
        Code k = decode_code(w, first);
        const bool over = (k.zrun & (k.n > N)) | (k.n8 & (k.np > N));     // ulcDecoder.c:127,139,154
        const bool toEnd = k.stop | k.tail;
        const int used = over ? 0 : (toEnd ? N : k.n + k.np);
        qidx = (k.qnew >= 0) ? k.qnew : qidx;                              // ulcDecoder.c:89-98
        if (k.plain) put_prec(make_uint2((uint32_t)(S - N) | ((uint32_t)qidx << 15) | (1u << 20), w & 0xFu));   // (a plain coefficient part 1 left: the block's bits end inside the run)
        if ((k.n8 | k.tail) & !over) {
            const int np = k.tail ? N : k.np;
            put_nrec(make_uint2((uint32_t)(S - N) | ((uint32_t)(np - (k.tail ? 1 : 0)) << 16) | (k.tail ? 0x80000000u : 0u),
                                (uint32_t)(draws - uDraw0) | ((uint32_t)k.l << 16) | ((uint32_t)qidx << 21)));
        }
        if (k.tail & !over) {
            // ulcDecoder.c:163-186: start amplitude, decay, first coefficient, count; the chain itself runs after the walk
            const float lev0 = (float)(k.l * k.l) * expand_quantizer(qidx) * (1.0f / 16);
            const float rr = 1.0f + (float)(k.dn * k.dn) * -0x1.0p-19f;
            utail[uslot] = make_float4(lev0, rr, __int_as_float(S - N), __int_as_float(N));
        }
        draws += over ? 0 : (k.tail ? N : k.np);
        pos += 4 * k.len;
        N -= used;
        bad |= over;
        first = false;
        if ((N == 0) | over) next_unit();
        if (pos > limit) { bad = true; fin = true; }               // ran off the readable bytes: corrupt
        }
        }
        if (RING && __any((nP - fP > DSCAN_RP - 2) | (nN - fN > DSCAN_RN - 1))) {
            ring_flush<DSCAN_RP>(ringP, rlane, fP, nP, c.prec, blk, c.precStride);
            ring_flush<DSCAN_RN>(ringN, rlane, fN, nN, c.nrec, blk, c.nrecStride);
        }
    }
    if (RING) {
        ring_flush<DSCAN_RP>(ringP, rlane, fP, nP, c.prec, blk, c.precStride);
        ring_flush<DSCAN_RN>(ringN, rlane, fN, nN, c.nrec, blk, c.nrecStride);
        if (!live) return 0;
    }
    bool ok = !bad;
    c.bits[blk] = ok ? pos : 0;
    c.wcScan[blk] = ok ? wc : 0;
    c.draws[blk] = draws;
    // Decaying-noise tails (ulcDecoder.c:176-186: v = p; p *= r per coefficient): the magnitude chain is one serial
    // float recurrence per unit.  It runs here, 64 units abreast, and leaves the magnitude at every 32nd coefficient
    // position so that the synthesis can start anywhere.
    if (ok) {
        for (int t = 0; t < total; t++) {
            int ch = t / nsub, j = t - ch * nsub, us = ch * 4 + j;
            float4 tp = utail[us];
            int n = __float_as_int(tp.w);
            if (n <= 0) continue;
            int p0 = __float_as_int(tp.z);
            float lev = tp.x; const float rr = tp.y;
            float *tm = c.tailMag + ((size_t)blk * c.C * 4 + us) * c.tailStride;
            int i = p0, end = p0 + n;
            while (i < end && (i & 31)) { lev *= rr; i++; }        // up to the first chunk boundary
            for (; i < end; i += 32) {
                tm[i >> 5] = lev;
#pragma unroll
                for (int q = 0; q < 32; q++) lev *= rr;
            }
        }
    }
    return ok ? pos : 0;
}

// Pass 1 - one lane per block.
__global__ __launch_bounds__(64) void k_dscan(UlcxDecCtx c) {
    __shared__ uint2 ringP[DSCAN_RP * 64], ringN[DSCAN_RN * 64];
    const int id0 = blockIdx.x * 64 + threadIdx.x, Kc = c.k1 - c.k0;      // streams [s0, s1), blocks [k0, k1) of each
    const bool live = id0 < (c.s1 - c.s0) * Kc;
    const int id = live ? id0 : (c.s1 - c.s0) * Kc - 1;                   // (a lane without a block shadows the last one, and writes nothing)
    const int blk = (c.s0 + id / Kc) * c.K + c.k0 + id % Kc;
    const uint8_t *p = c.in + (size_t)blk * c.slot, *bufEnd = c.in + c.inBytes;
    // every look of every lane of the wave inside the buffer (a look reads up to three 16-byte chunks behind the one
    // the position is in, and the position stops at the slot's end): the window without bounds checks
    const bool inside = (size_t)(p - c.in) >= ((uintptr_t)p & 15) && (size_t)(bufEnd - p) >= (size_t)c.slot + 80;
    if (__ballot(!inside) == 0ull) scan_block<NybWinFast, true>(c, blk, p, c.slot * 8, c.slot, c.in, bufEnd, live, ringP, ringN);
    else scan_block<NybWin, true>(c, blk, p, c.slot * 8, c.slot, c.in, bufEnd, live, ringP, ringN);
}

// Pass 1, packed payloads - one lane per stream: a block's start is only known once the previous
// block has been parsed (the container stores no block lengths, tools/ulcDecodeTool.c:153-165).
__global__ __launch_bounds__(64) void k_dscan_packed(UlcxDecCtx c) {
    int s = c.s0 + blockIdx.x * 64 + threadIdx.x;
    if (s >= c.s1) return;
    int off = c.packOff[s];
    int avail = c.payBytes[s];
    const uint8_t *base = c.in + (size_t)s * c.payStride;
    bool dead = false;
    for (int k = 0; k < c.K; k++) {
        int blk = s * c.K + k;
        c.blkOff[blk] = off;
        int bits = 0;
        if (!dead && off < avail) bits = scan_block(c, blk, base + off, (avail - off) * 8, avail - off, c.in, c.in + c.inBytes);
        else { c.bits[blk] = 0; c.wcScan[blk] = 0; c.draws[blk] = 0; }
        if (!bits) dead = true;
        off += (bits + 7) >> 3;                                     // the tool rounds every block up to a byte
    }
    c.packOff[s] = off;
}

// ---------------------------------------------------------------------------
// xorshift32 is linear over GF(2): the state after n draws is T^n * state.  jumpT holds T^(d*16^i) for every
// hexadecimal digit d of n at every position i, each as four 256-entry byte tables (host-built, ulcx_api.cpp):
// a jump costs one table-driven mat-vec (4 lookups) per non-zero digit.
// ---------------------------------------------------------------------------
__device__ __forceinline__ uint32_t rng_jump(const uint32_t *__restrict__ jt, uint32_t s, uint32_t n) {
    for (int i = 0; n; i++, n >>= 4) {
        const uint32_t dgt = n & 15u;
        if (dgt) {
            const uint32_t *J = jt + ((size_t)(i * 16 + dgt) << 10);
            s = J[s & 255u] ^ J[256 + ((s >> 8) & 255u)] ^ J[512 + ((s >> 16) & 255u)] ^ J[768 + (s >> 24)];
        }
    }
    return s;
}

// wave-wide inclusive prefix sum / prefix maximum of one 32-bit value per lane (row shifts + row broadcasts)
__device__ __forceinline__ uint32_t wave_scan_add(uint32_t v) {
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xF, 0xF, false);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xF, 0xF, false);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xF, 0xF, false);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xF, 0xF, false);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xA, 0xF, false);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xC, 0xF, false);
    return v;
}
__device__ __forceinline__ uint32_t umax32(uint32_t a, uint32_t b) { return a > b ? a : b; }
__device__ __forceinline__ uint32_t wave_scan_max(uint32_t v) {
    v = umax32(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xF, 0xF, false));
    v = umax32(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xF, 0xF, false));
    v = umax32(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xF, 0xF, false));
    v = umax32(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xF, 0xF, false));
    v = umax32(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xA, 0xF, false));
    v = umax32(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xC, 0xF, false));
    return v;
}
#define WAVE_SYNC() do { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); } while (0)

// float index inside a padded FFT array (two floats of padding after every 32): complex n sits at FFT_PADS(n, DPS)
__device__ __forceinline__ int padf(int f) { return f + ((f >> (DPS + 1)) << 1); }      // float index into a padded array of complex

// Noise runs found while decoding a unit, 8 bytes each:
//   x = first coefficient | count << 16 (count - 1 for a tail, which may span the whole unit) | tail << 31
//   y = draws made in the unit before the run | level << 16 | quantizer index << 21
#ifdef ULCX_DSYN_STAMPS
#define DSYN_NSTAMP 20
struct DsynStamps { unsigned long long t[DSYN_NSTAMP], t0; };
#define SWAITALL() asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory")      // diagnostic: the stamp behind it sees the wait for everything in flight
#define SSTAMP(sw, i) do { if ((sw).stp) { unsigned long long t_ = __builtin_amdgcn_s_memtime(); (sw).stp->t[i] += t_ - (sw).stp->t0; (sw).stp->t0 = t_; } } while (0)
#else
#define SSTAMP(sw, i) do {} while (0)
#define SWAITALL() do {} while (0)
#endif
struct SynWave {                 // one wave's working set while it synthesises one (channel, subblock) unit
    float *A;                    // the unit's coefficients = the FFT's input array (LDS, padded)
    int   *pre;                  // 64 prefix counts (LDS)
    uint32_t *seedTab;           // the unit's sign-parity stream P (LDS, synth_noise)
    int lane;
#ifdef ULCX_DSYN_STAMPS
    DsynStamps *stp;
#endif
};
#define SEEDTAB_DRAWS 2048       // draws of a unit the table covers; pieces beyond it jump on their own

// Noise synthesis (ulcDecoder.c:146-160, :166-186).  The decoder draws one xorshift value per noise coefficient and
// only ever looks at its top bit: the sign of the run's level flips, cumulatively, on every draw with the top bit set.
//   1. the unit's draws, densely: lane l owns draws 32l+1 .. 32l+32; its word of P (bit n of the stream = parity of the top
//      bits of draws 1 .. n+1) is the parity prefix of those 32 top bits - a linear function of the unit's start state, read
//      from tables (par_word) - completed by a parity carry across the lanes.  P goes to LDS (2048 bits per pass).
//   2. one lane per (run, 32-coefficient chunk) piece: 32 sign bits are a window of P (xor the parity at the run's
//      start), the level is the run's constant or the tail's chain value at the chunk (k_dscan) decaying from there.
// No sequential generator in step 2, no jumps, and every coefficient is written exactly once.
// A lane's word of a unit's sign-parity stream P, from the state the (pass of the) unit starts from: bit i = parity of the top
// bits of draws 32 lane + 1 .. 32 lane + i + 1.  Linear in the state: four byte tables per lane (host: build_rng_tables), and
// since the state is the same in all lanes, each look-up is ONE coalesced row of 64 words (c.parT [4][256][lane]).
__device__ __forceinline__ uint32_t par_word(const uint32_t *__restrict__ parT, uint32_t st, int lane) {
    const uint32_t *J = parT + lane;
    return J[(st & 255u) << 6] ^ J[(256u + ((st >> 8) & 255u)) << 6] ^ J[(512u + ((st >> 16) & 255u)) << 6] ^ J[(768u + (st >> 24)) << 6];
}
__device__ __forceinline__ void synth_noise(const UlcxDecCtx &c, const SynWave &sw, const uint2 *__restrict__ list, int nE, const uint2 ent0, uint32_t x0, uint32_t unitSeed, int unitDraws,
                                            float tailRR, const float *tailMag) {
    const int lane = sw.lane;
    uint32_t *P = sw.seedTab;                                 // the unit's whole stream: max(64, BS/32) words + a zero word
    // (ent0: round 0 of the run list, x0: the lane's word of the first pass - both fetched by the caller with the unit's first records)
    // ---- 1. sign-parity stream, 2048 draws per pass (one pass unless the unit has more than 2048 noise coefficients)
    uint32_t passPar = 0;                                     // parity of the top bits of all earlier passes
    for (int dbase = 0; dbase < unitDraws; dbase += 2048) {
        // bit i of x = parity of the top bits of draws dbase+32*lane+1 .. +i+1 (bit 31: of all 32)
        const uint32_t x = (dbase == 0) ? x0 : par_word(c.parT, rng_jump(c.jumpT, unitSeed, (uint32_t)dbase), lane);
        const unsigned long long odd = __ballot((int)x < 0);
        const uint32_t carry = ((uint32_t)__builtin_amdgcn_mbcnt_hi((uint32_t)(odd >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)odd, 0)) + passPar) & 1u;
        P[(dbase >> 5) + lane] = carry ? ~x : x;
        passPar ^= (uint32_t)__popcll(odd) & 1u;
    }
    if (lane == 0) P[((unitDraws + 2047) >> 11) << 6] = 0u;   // the word behind the last pass: windows at the very end read it
    WAVE_SYNC();
    {
        // ---- 2. pieces
        for (int e0 = 0; e0 < nE; e0 += 64) {
            const int e = e0 + lane;
            const bool have = e < nE;
            const uint2 ent = (e0 == 0) ? ent0 : (have ? list[e] : make_uint2(0u, 0u));
            const int pos = ent.x & 0xFFFF, isTail = (int)(ent.x >> 31);
            const int np = (int)((ent.x >> 16) & 0x7FFF) + isTail;
            const int nseg = have ? ((pos + np - 1) >> 5) - (pos >> 5) + 1 : 0;
            const uint32_t incl = wave_scan_add((uint32_t)nseg);
            const int total = __builtin_amdgcn_readlane((int)incl, 63);
            sw.pre[lane] = have ? (int)(incl - nseg) : 0x7FFFFFFF;
            WAVE_SYNC();
            for (int sb = 0; sb < total; sb += 64) {
                const int si = sb + lane;
                const bool act = si < total;
                // the run this piece belongs to: the last one whose first piece is at or before si
                int lo = 0, hi = 63;
#pragma unroll
                for (int it = 0; it < 6; it++) { int mid = (lo + hi + 1) >> 1; bool le = sw.pre[mid] <= si; lo = le ? mid : lo; hi = le ? hi : mid - 1; }
                const int j = act ? lo : 0;
                const uint32_t ex = (uint32_t)__builtin_amdgcn_ds_bpermute(j << 2, (int)ent.x);
                const uint32_t ey = (uint32_t)__builtin_amdgcn_ds_bpermute(j << 2, (int)ent.y);
                const int first = __builtin_amdgcn_ds_bpermute(j << 2, (int)(incl - nseg));
                const int rpos = ex & 0xFFFF, rtail = (int)(ex >> 31), rnp = (int)((ex >> 16) & 0x7FFF) + rtail;
                const int d0 = ey & 0xFFFF, lvl = (ey >> 16) & 31, qi = (ey >> 21) & 63;
                const int chunk = (rpos >> 5) + (act ? si - first : 0);          // (idle lanes shadow the first piece of run 0: every address they form is valid)
                const int plo = rpos > (chunk << 5) ? rpos : (chunk << 5);
                const int phi = (rpos + rnp) < ((chunk << 5) + 32) ? (rpos + rnp) : ((chunk << 5) + 32);
                const int off = plo - rpos;
                const int dc = d0 + off;                                         // draws of the unit in front of the piece's first coefficient
                const int n = act ? phi - plo : 0;
                // sign bits of the piece: P bits dc .. dc+31 (parity after its i-th draw), relative to the parity where the run began
                uint32_t win = __builtin_amdgcn_alignbit(P[(dc >> 5) + 1], P[dc >> 5], dc & 31);
                uint32_t s0 = 0;                                                 // parity where the run began
                if (d0 > 0) s0 = (P[(d0 - 1) >> 5] >> ((d0 - 1) & 31)) & 1u;
                win ^= 0u - s0;
                const float quant = expand_quantizer(qi);
                float mag = (float)(lvl * lvl) * quant * (rtail ? (1.0f / 16) : (1.0f / 4));      // ulcDecoder.c:146-150, :166-170
                if (rtail && off > 0) mag = tailMag[chunk];                                         // the tail's chain at coefficient 32*chunk (k_dscan)
                const float rr = rtail ? tailRR : 1.0f;
                float *dst = sw.A + padf(plo);                                                      // (DPS 4: a piece never crosses a padding gap; DPS 3: up to two)
                constexpr int GAPF = 2 << DPS;                                                      // floats between two padding gaps
                const int g1 = GAPF - (plo & (GAPF - 1));                                           // piece elements in front of the first gap
                SSTAMP(sw, 9);
#pragma unroll
                for (int i = 0; i < 32; i++) {
                    // ulcDecoder.c:156-160 / :181-184: flip on the draw's top bit (cumulative), store, decay (r = 1 for runs)
                    const float v = __uint_as_float(__float_as_uint(mag) | ((win << (31 - i)) & 0x80000000u));
                    if (i < n) dst[GAPF >= 32 ? i : i + (i >= g1 ? 2 : 0) + (i >= g1 + GAPF ? 2 : 0)] = v;
                    mag *= rr;
                }
            }
            WAVE_SYNC();
        }
    }
}

// Dequantise one (channel, subblock) unit into A[0 .. S) (zeroed by the caller) from what the scan left: one lane per
// plain-run record (up to seven coefficients each, ulcDecoder.c:69-73), then the noise runs (synth_noise).
// ur = {first plain-run record, their count, first noise record, their count} of the unit.
__device__ __forceinline__ void synth_unit(const UlcxDecCtx &c, const SynWave &sw, int S, const uint2 *__restrict__ prec, const uint2 *__restrict__ nrec,
                                           int4 ur, uint32_t unitSeed, int unitDraws, float tailRR, const float *tailMag, int zeroFloat2 = 0) {
    const int lane = sw.lane;
    // the lane's word of the unit's sign-parity stream (synth_noise): four coalesced table rows from the unit's start state, in
    // flight behind the zero fill and the scatter (round 5; a two-step jump to the lane's own state - eight 64-address gathers -
    // and 32 draws before: the noise synthesis waited 2-3 k cycles for them)
    const uint32_t x0 = par_word(c.parT, unitSeed, lane);
    const uint2 *pr = prec + ur.x;
    // rounds 0..2 of the plain-run records: in flight behind the zero fill (one round trip to memory, not one per round)
    const uint2 recFirst = (lane < ur.y) ? pr[lane] : make_uint2(0u, 0u);
    const uint2 rec1 = (lane + 64 < ur.y) ? pr[lane + 64] : make_uint2(0u, 0u), rec2 = (lane + 128 < ur.y) ? pr[lane + 128] : make_uint2(0u, 0u);
    const uint2 ent0 = (lane < ur.w) ? nrec[ur.z + lane] : make_uint2(0u, 0u);       // round 0 of the noise runs: one round trip for all of the unit's lists
    if (zeroFloat2) {                                            // (the caller's array: cleared here, behind the loads above)
        float2 *Az = (float2 *)sw.A;
        for (int i = lane; i < zeroFloat2; i += 64) Az[i] = make_float2(0.0f, 0.0f);
        WAVE_SYNC();
    }
    SSTAMP(sw, 12);
    SWAITALL();
    SSTAMP(sw, 13);
    for (int r0 = 0; r0 < ur.y; r0 += 64) {
        const int r = r0 + lane;
        if (r < ur.y) {
            const uint2 rec = (r0 == 0) ? recFirst : (r0 == 64) ? rec1 : (r0 == 128) ? rec2 : pr[r];
            const int pos = rec.x & 0x7FFF, qi = (rec.x >> 15) & 31, m = (rec.x >> 20) & 7;
            const float quant = expand_quantizer(qi);
            float *dst = sw.A + padf(pos);
            const int gap = (2 << DPS) - (pos & ((2 << DPS) - 1));   // coefficients before the next padding gap (a run of <= 7 crosses at most one)
#pragma unroll
            for (int i = 0; i < 7; i++) {
                int sv = (int)((rec.y >> (4 * i)) & 0xF);
                sv = (sv ^ 0x8) - 0x8;
                sv = (sv < 0) ? (-sv * sv) : (+sv * sv);
                if (i < m && pos + i < S) dst[i + (i >= gap ? 2 : 0)] = (float)sv * quant;
            }
        }
    }
    SSTAMP(sw, 8);
    if (ur.w > 0) {
        synth_noise(c, sw, nrec + ur.z, ur.w, ent0, x0, unitSeed, unitDraws, tailRR, tailMag);
    }
    SSTAMP(sw, 10);
    WAVE_SYNC();
}

// ---------------------------------------------------------------------------
// Time domain of one channel of a DECIMATED block of a stereo stream, by one wave (ulcDecoder.c:219-273).
// The reference keeps BS/2 pending samples per channel in TransformInvLap, time-reversed: the tail of the last subblock
// (to be overlapped with the next) in front, behind it a FIFO that delays decimated subblocks to the centre of the
// block, and it moves the FIFO once per subblock.  Read in time order (P[n] = Lap[BS/2-1-n]) the two parts are ONE list,
// and a subblock of size S at coefficient offset off only ever (i) replaces the M = S/2 pending samples in front of
// time off by the first half of its windowed output, (ii) appends the second half and (iii) appends its own tail:
//     T[off-M+p]     = c A - s B      A = T[off-M+p], B = z[M+p]          (pass-through below the ramp)
//     T[off+M-1-p]   = s A + c B
//     T[off+M+i]     = z[M-1-i]
// on the timeline T = [pending (times -BS/2..-1) | this block's BS new samples]; the block's output is T[-BS/2 .. BS/2)
// and T[BS/2 .. BS) is the new pending list.  Nothing is shifted.  T's non-negative times live in the channel's FFT
// array (a subblock's spectrum sits at [off, off+S): everything it writes at times >= off lands in its own region, so
// "all lanes read, wave barrier, all lanes write" is enough), negative times in the lapping state L (L[-1-t]).
// Returns the size of the last subblock; the caller emits the output and stores the new pending list (dec_time_finish).
// DEC_MAXT = (BlockSize/2)/64 register slots per lane: a template parameter of the kernel (16: BlockSize <= 2048, 32: <= 4096)
// ---------------------------------------------------------------------------
template <int DEC_MAXT>
__device__ __forceinline__ int dec_time_wave(const UlcxDecCtx &c, float2 *zc, float *L, int wc, unsigned pat0, int nsub, int lastSub, int lane) {
    const int BS = c.BS;
    float *arr = (float *)zc;                                    // times >= 0, padded
    int last = lastSub;
    unsigned pat = pat0; int off = 0;
    for (int j = 0; j < nsub; j++, pat >>= 4) {
        const int d = pat & 7, S = BS >> d, M = S >> 1;
        int ov = S;                                              // ulcDecoder.c:234-239
        if (pat & 8) ov >>= (wc & 7);
        if (ov > last) ov = last;
        last = S;
        const float2 *zj = zc + FFT_PADS(off >> 1, DPS);
        const float2 *pre = c.T.pre[d];
        const int a = (S - ov) >> 1;
        const float *fall = c.T.winFall + ov, *rise = c.T.winRise + ov;
        const int bits = 31 - __clz(M);
        const int tA = off - M;                                  // time of A(0)
        // post-twiddle fused with the windowed overlap (oracle/orc_fourier.c orc_imdct): pair p: A = T[tA+p], B = z[M+p]
        float o[DEC_MAXT / 4][4], nl[DEC_MAXT / 4][2];
#pragma unroll
        for (int t = 0; t < DEC_MAXT / 4; t++) {
            const int kk = lane + 64 * t;
            if (kk < M / 2) {
                const int k1 = kk, k2 = M - 1 - kk;
                const int r1 = FFT_PADS((int)(__brev((unsigned)k1) >> (32 - bits)), DPS);
                const int r2 = FFT_PADS((int)(__brev((unsigned)k2) >> (32 - bits)), DPS);
                const float2 y1 = cmulc_post(zj[r1], pre[k1]), y2 = cmulc_post(zj[r2], pre[k2]);     // (Re y, -Im y)
                const int   pv[2] = { M - 1 - 2 * k1, M - 2 - 2 * k1 };
                float Av[2];
                // (L may be global memory, arr is LDS: separate branches, never a select between the two pointers)
                if (tA + pv[1] < 0) { Av[0] = L[-1 - (tA + pv[0])]; Av[1] = L[-1 - (tA + pv[1])]; }      // (pv[1] = pv[0]-1, even time: both on the same side of 0)
                else { Av[0] = arr[padf(tA + pv[0])]; Av[1] = arr[padf(tA + pv[1])]; }
                const float Bv[2] = { y1.y, y2.x };
#pragma unroll
                for (int q = 0; q < 2; q++) {
                    const int p = pv[q];
                    const float A = Av[q], B = Bv[q];
                    if (p < a) { o[t][2 * q] = A; o[t][2 * q + 1] = B; }
                    else {
                        const float cw = fall[p - a], sn = rise[p - a];
                        const float m1 = sn * B, m3 = cw * B;               // spec v2: fused (orc_imdct)
                        o[t][2 * q] = __builtin_fmaf(cw, A, -m1);
                        o[t][2 * q + 1] = __builtin_fmaf(sn, A, m3);
                    }
                }
                nl[t][0] = y1.x; nl[t][1] = y2.y;               // z[2 k1], z[2 k1 + 1]: the tail at times off+M+pv[0], off+M+pv[1]
            }
        }
        WAVE_SYNC();
#pragma unroll
        for (int t = 0; t < DEC_MAXT / 4; t++) {
            const int kk = lane + 64 * t;
            if (kk < M / 2) {
                const int p0 = M - 1 - 2 * kk, p1 = p0 - 1;
                if (tA + p1 < 0) { L[-1 - (tA + p0)] = o[t][0]; L[-1 - (tA + p1)] = o[t][2]; }
                else { arr[padf(tA + p0)] = o[t][0]; arr[padf(tA + p1)] = o[t][2]; }
                arr[padf(off + M - 1 - p0)] = o[t][1]; arr[padf(off + M - 1 - p1)] = o[t][3];
                arr[padf(off + M + p0)] = nl[t][0]; arr[padf(off + M + p1)] = nl[t][1];
            }
        }
        WAVE_SYNC();
        off += S;
    }
    return last;
}

// ---------------------------------------------------------------------------
// Output samples.  OUT = float: the C API's layout; OUT = int16_t: PCM16 output, converted on store
// exactly as the reference's WAV writer does (tools/WavIO_Helper.c:9-13,56-63: lrintf(clamp(x * 2^15, -32768, 32767))).
__device__ __forceinline__ int16_t to_pcm16(float x) {
    float v = x * 0x1.0p+15f;
    v = (v < -32768.0f) ? -32768.0f : (v > 32767.0f) ? 32767.0f : v;
    return (int16_t)__float2int_rn(v);
}
// (the decoded samples are written once and not read again by the decoder: non-temporal)
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void st1(float *p, float a) { __builtin_nontemporal_store(a, p); }
__device__ __forceinline__ void st2(float *p, float a, float b) { f32x2 w = { a, b }; __builtin_nontemporal_store(w, (f32x2 *)p); }
__device__ __forceinline__ void st4(float *p, float a, float b, float d, float e) { f32x4 w = { a, b, d, e }; __builtin_nontemporal_store(w, (f32x4 *)p); }
__device__ __forceinline__ void st1(int16_t *p, float a) { *p = to_pcm16(a); }
__device__ __forceinline__ void st2(int16_t *p, float a, float b) { *(short2 *)p = make_short2(to_pcm16(a), to_pcm16(b)); }
__device__ __forceinline__ void st4(int16_t *p, float a, float b, float d, float e) { *(short4 *)p = make_short4(to_pcm16(a), to_pcm16(b), to_pcm16(d), to_pcm16(e)); }
template <typename OUT> __device__ __forceinline__ OUT *out_base(const UlcxDecCtx &c);
template <> __device__ __forceinline__ float *out_base<float>(const UlcxDecCtx &c) { return c.pcm; }
template <> __device__ __forceinline__ int16_t *out_base<int16_t>(const UlcxDecCtx &c) { return c.pcm16; }

// LDS carve, in floats.  Stereo kernel (k_dsyn):  z [2 padded arrays of BS/2 complex] | lap [2][BS/2] | twl [BS/4 complex] |
//   per wave: noise runs, prefix counts, seed table | 128 block / unit seeds.  General kernel (k_dgen): z [1 array] | per-wave lists.
#define DSYN_PWORDS(BS) (((BS) / 32 > 64 ? (BS) / 32 : 64) + 2)
#define DSYN_CHUNK 32            // blocks of a stream whose RNG states and headers the stereo kernel stages at once (<= 64)
struct DsynLds { int zFloats, lapFloats, twFloats, listFloats; };
__host__ __device__ static inline DsynLds dsyn_lds(int BS, int C, int fast, int twInLds) {
    DsynLds l;
    l.zFloats = (fast ? 2 : 1) * 2 * FFT_PADDEDS(BS / 2, DPS);
    // twInLds: 1 = lapping state and twiddles in LDS, 0 = both in global memory, 2 = twiddles in LDS, lapping state in global memory
    l.lapFloats = (fast && BS <= 2048 && twInLds == 1) ? 2 * (BS / 2) : 0;      // (above 2048 the stereo kernel keeps the lapping state in global memory: a third workgroup per CU)
    l.twFloats = (fast && BS <= 2048 && twInLds != 0) ? BS / 2 : 0;               // (likewise the FFT twiddles: read from the tables in global memory, L1/L2-hot, a fourth workgroup per CU)
    l.listFloats = 2 * (64 + DSYN_PWORDS(BS)) + 128;             // per wave: prefix counts, sign-parity stream; per workgroup: 128 block / channel RNG states
    if (fast) l.listFloats += 2 * (DSYN_CHUNK + 4) * 8;          // stereo kernel: the headers of a chunk of blocks + of a decimated block's four units, per channel
    (void)C;
    return l;
}

// ---------------------------------------------------------------------------
// Stereo streams (BlockSize <= 4096): one workgroup = one stream, one wave per channel up to the end of the FFTs.
// ---------------------------------------------------------------------------
// The lapping state lives in global memory (L2-hot: every element is re-read by the thread that wrote it one block earlier);
// TWL: FFT twiddles in LDS (BlockSize <= 2048); SPLIT: the grid is an even cut of the (stream, block) pairs (else one
// workgroup per stream: the index arithmetic and the choice of the lapping rows below fold away)
// BSC: BlockSize as a compile-time constant (2048: the headline geometry - the loops of an un-decimated block unroll, the
// table and lapping-state loads of a thread's trips can be issued together; 0: read from the context)
template <typename OUT, int DEC_MAXT, bool TWL, bool SPLIT = false, int BSC = 0>
__global__ __launch_bounds__(WG, 3) void k_dsyn(UlcxDecCtx c) {
    constexpr bool LAPG = true;
    extern __shared__ float lds[];
    const int BS = BSC ? BSC : c.BS, H2 = BS / 2;
    constexpr int C = 2;
    const int tid = threadIdx.x;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const DsynLds L = dsyn_lds(BS, C, 1, TWL ? 2 : 0);
    float2 *z    = (float2 *)lds;
    // lapping state: in LDS for the stream's blocks of this launch, or (LAPG) in global memory: the arrays a stream's state
    // is read from / written to at its first / last block of the launch, the workgroup's own scratch rows in between.
    // Every element is read and rewritten by the same thread in un-decimated blocks and by the same wave in decimated
    // ones; the barriers between the two kinds of block order the rest.
    float  *ldsLap = lds + L.zFloats;
    // (one workgroup per stream - grid = streams -: the stream's own rows serve, nothing else touches them during the launch)
    // (a cut launch: the leading workgroups still take one whole stream each and use its rows; the others their own scratch rows)
    const bool ownStream = !SPLIT || (int)blockIdx.x < c.synFull;
    float  *scr = !LAPG ? ldsLap : ownStream ? c.lapO + (size_t)(c.s0 + blockIdx.x) * C * H2 : c.lapScratch + (size_t)((int)blockIdx.x - c.synFull) * C * H2;
    float2 *twl  = (float2 *)(lds + L.zFloats + L.lapFloats);        // FFT twiddles: the full-size table, or the three of a decimated block's sizes
    SynWave sw;
    sw.pre  = (int *)(lds + L.zFloats + L.lapFloats + L.twFloats) + wv * (64 + DSYN_PWORDS(BS));
    sw.seedTab = (uint32_t *)(sw.pre + 64);
    uint32_t *bseed = (uint32_t *)(lds + L.zFloats + L.lapFloats + L.twFloats + 2 * (64 + DSYN_PWORDS(BS)));   // [0,64): RNG state at each block's start, [64,128): at its second channel
    // [wave][block of the chunk][8]: {window code, the first unit's four record fields, its draws (un-decimated block), its tail decay, -}
    int *hdr = (int *)(bseed + 128) + wv * ((DSYN_CHUNK + 4) * 8);
    sw.lane = lane;
    if (TWL) for (int i = tid; i < BS / 4; i += WG) twl[i] = c.T.tw[0][i];
    bool twFull = true;
    const int M0 = BS >> 1, Mp0 = FFT_PADDEDS(M0, DPS);
    float2 *zc = z + wv * Mp0;                                       // this wave's channel
#ifdef ULCX_DSYN_STAMPS
    // diagnostic build only: shader cycles per phase, summed over the workgroup's blocks
    DsynStamps stq; for (int i = 0; i < DSYN_NSTAMP; i++) stq.t[i] = 0; stq.t0 = __builtin_amdgcn_s_memtime();
    sw.stp = &stq;
#define STAMP(i) do { unsigned long long t_ = __builtin_amdgcn_s_memtime(); stq.t[i] += t_ - stq.t0; stq.t0 = t_; } while (0)
#else
#define STAMP(i) do {} while (0)
#endif
    __syncthreads();

    // The (stream, block) pairs of the launch in stream-major order, cut evenly over the grid (round 3).  One workgroup
    // per stream is the special case grid = streams.  A workgroup that enters a stream behind its first block of the
    // launch runs the block in front of its range without output first: the lapping state a block leaves depends on that
    // block alone (the pending list is rewritten in full: dec_time_wave, and the un-decimated epilogue below), the RNG
    // state is reached by a jump over the draws of the blocks in front, LastSubBlockSize follows from the previous
    // block's window code.  So 4096 streams need not come in rounds of the machine's 1536 workgroup slots (2.67 rounds,
    // the last one a third full), and a few long streams fill the machine too.
    const int Kc = c.k1 - c.k0;
    const long long T = (long long)(c.s1 - c.s0) * Kc;
    long long f0 = (long long)blockIdx.x * Kc, f1 = f0 + Kc;
    if (!ownStream) {
        const long long base = (long long)c.synFull * Kc, Tr = T - base;
        const int j = (int)blockIdx.x - c.synFull, n = (int)gridDim.x - c.synFull;
        f0 = base + Tr * j / n; f1 = base + Tr * (j + 1) / n;
    }
    if (f0 >= f1) return;
    bool warm = SPLIT && (f0 % Kc) != 0;
    int nTrip = (int)(f1 - f0) + (warm ? 1 : 0);
    int s = c.s0 + (int)((f0 - (warm ? 1 : 0)) / Kc), k = c.k0 + (int)((f0 - (warm ? 1 : 0)) % Kc) - 1;
    int sCur = -1, lastSub = 0, dead = 0, chunkK = -1;
    uint32_t seed = 0;
    const int laneOuter = lane, tidOuter = tid;
    for (; nTrip > 0; nTrip--, warm = false) {
        // Inside this loop over the blocks the compiler hoists every load and index that depends on the thread only
        // (pre-twiddles, window factors, bit-reversed positions, padded FFT addresses) out of the loop - registers held
        // across all blocks.  Opaque copies of the thread's indices keep them where they are used.
        int lane = laneOuter, tid = tidOuter;
        asm volatile("" : "+v"(lane), "+v"(tid));
        sw.lane = lane;                                              // (round 3: 168 registers with spills -> 156 without, 1.05 -> 1.00 ms)
        if (++k == c.k1) { k = c.k0; s++; }
        const int blk = s * c.K + k;
        if (s != sCur) {
            // entering a stream, at its first block of the launch or behind it: the state in front of block k
            if (!LAPG && sCur >= 0) { __syncthreads(); float *go = c.lapO + (size_t)sCur * C * H2; for (int i = tid; i < 2 * H2; i += WG) go[i] = ldsLap[i]; __syncthreads(); }
            sCur = s;
            if (!LAPG) { const float *gi = c.lap + (size_t)s * C * H2; for (int i = tid; i < 2 * H2; i += WG) ldsLap[i] = gi[i]; }
            __syncthreads();
            int bad = 0; uint32_t dsum = 0;
            for (int j = c.k0 + tid; j < k; j += WG) {               // blocks of the launch in front of k (none at the stream's start)
                const int bj = s * c.K + j;
                if (c.wcScan[bj] == 0) bad = 1;
                dsum += (uint32_t)c.draws[bj];
            }
            bad = __syncthreads_or(bad);
            dead = c.dead[s] | bad;                                  // (a corrupt block in front: the stream is dead, no other state matters)
            lastSub = c.lastSub[s];
            seed = c.seed[s];
            if (k > c.k0) {
                const uint32_t ws = wave_scan_add(dsum);             // (lane 63: the wave's sum)
                if (lane == 63) bseed[wv] = ws;
                __syncthreads();
                seed = rng_jump(c.jumpT, seed, bseed[0] + bseed[1]);
                __syncthreads();
                const unsigned pp0 = ulcx_pattern(c.wcScan[blk - 1]);        // LastSubBlockSize as block k-1 left it (ulcDecoder.c:300)
                unsigned pp = pp0; int ls = BS;
                if ((BS >> (pp0 & 7)) != BS) do { ls = BS >> (pp & 7); } while (pp >>= 4);
                lastSub = ls;
            }
            chunkK = -1;
        }
        if (chunkK < 0 || k >= chunkK + DSYN_CHUNK) {
            // The stream's one RNG chain (ulcDecoder.c:75-81) for the next DSYN_CHUNK blocks at once: a block starts draws-
            // of-its-predecessors after the chunk's first state - one lane per block, prefix sum, one jump each (wave 1: the
            // state at each block's second channel).  A corrupt block ends the stream: it and its successors draw nothing.
            // The same lanes leave what a block's first unit needs from the walk's per-block arrays in LDS: a block
            // then starts from LDS instead of from a round trip to HBM.
            __syncthreads();
            const int kk = k + lane, bk = s * c.K + (kk < c.k1 && lane < DSYN_CHUNK ? kk : k);
            const bool on = kk < c.k1 && lane < DSYN_CHUNK;
            const int wcl = c.wcScan[bk];
            const int drb = c.draws[bk];
            const int ud4 = c.unitDraws[(size_t)bk * C * 4 + 4];
            const int4 url = c.unitRec[(size_t)bk * C * 4 + wv * 4];
            const float rrl = c.unitTail[(size_t)bk * C * 4 + wv * 4].y;
            int dr = on ? drb : 0;
            const unsigned long long badm = __ballot(on && wcl == 0);
            if (badm && lane >= __builtin_ctzll(badm)) dr = 0;
            const uint32_t incl = wave_scan_add((uint32_t)dr);
            const uint32_t before = incl - (uint32_t)dr + (wv ? (uint32_t)ud4 : 0u);
            bseed[wv * 64 + lane] = rng_jump(c.jumpT, seed, before);
            if (lane < DSYN_CHUNK) {
                int *h = hdr + lane * 8;
                h[0] = wcl; h[1] = url.x; h[2] = url.y; h[3] = url.z; h[4] = url.w;
                h[5] = wv ? drb - ud4 : ud4;                       // draws of the channel's one unit in an un-decimated block (unitDraws[0] = 0)
                h[6] = __float_as_int(rrl);
            }
            const uint32_t chunkDraws = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
            __syncthreads();
            seed = rng_jump(c.jumpT, seed, chunkDraws);            // state after the chunk: the next chunk's start / the stream's state behind the launch
            chunkK = k;
        }
        const int kb = k - chunkK;                                   // this block's slot in the chunk's tables
        const int *hb = hdr + kb * 8;
        const int wc = hb[0];
        if (wc == 0) dead = 1;                                       // a corrupt block ends the stream (ulcDecodeTool.c:154-157)
        const bool lastOfStream = k == c.k1 - 1;
        OUT *outp = out_base<OUT>(c) + (size_t)blk * C * BS;
        // where this block finds and leaves the lapping state
        const float *lapR = !SPLIT ? scr : (k == c.k0 ? c.lap + (size_t)s * C * H2 : scr);
        float *lapW = !SPLIT ? scr : (lastOfStream ? c.lapO + (size_t)s * C * H2 : scr);
        if (dead) {
            if (!warm) {
                for (int i = tid; i < C * BS; i += WG) st1(outp + i, 0.0f);
                if (tid == 0) c.bits[blk] = 0;
            }
            if (lastOfStream && tid == 0) { c.lastSubO[s] = lastSub; c.seedO[s] = seed; c.deadO[s] = 1; }
            if (LAPG && lastOfStream) for (int i = tid; i < 2 * H2; i += WG) lapW[i] = 0.0f;     // (never read again: the stream stays dead)
            continue;
        }
        const int *udraw = c.unitDraws + (size_t)blk * C * 4;
        const int4 *urec = c.unitRec + (size_t)blk * C * 4;
        const float4 *utail = c.unitTail + (size_t)blk * C * 4;
        const uint2 *prec = c.prec + (size_t)blk * c.precStride, *nrec = c.nrec + (size_t)blk * c.nrecStride;
        const float *tmag = c.tailMag + (size_t)blk * C * 4 * c.tailStride;
        const unsigned pat0 = ulcx_pattern(wc);
        const bool whole = (BS >> (pat0 & 7)) == BS;                 // one subblock per channel (ulcDecoder.c:242-245)
        int nsub = 1;
        if (!whole) { nsub = 0; unsigned q = pat0; do nsub++; while (q >>= 4); }
        if (TWL && whole != twFull) {
            // twiddle tables for this block's transform sizes: the full-size one, or those of N/2, N/4, N/8 back to back
            if (whole) for (int i = tid; i < BS / 4; i += WG) twl[i] = c.T.tw[0][i];
            else for (int i = tid; i < 7 * BS / 32; i += WG) {
                const int dd = i < BS / 8 ? 1 : i < 3 * BS / 16 ? 2 : 3;
                twl[i] = c.T.tw[dd][i - (dd == 1 ? 0 : dd == 2 ? BS / 8 : 3 * BS / 16)];
            }
            twFull = whole;
            __syncthreads();
        }
        STAMP(0);
        // Decimated block (round 5): what its 2..4 units need from the walk - record ranges, draws, tail decay - and their RNG start
        // states (a table jump of up to four dependent look-ups each) are fetched for ALL units at once, unit j in lane j, and
        // handed out by lane reads; a unit used to start with its own loads and its own jump, one after the other.
        int *uh = hdr + DSYN_CHUNK * 8;                              // [4 units][8] of this wave, behind its chunk table
        if (!whole) {
            if (lane < nsub) {
                const int dj = udraw[wv * 4 + lane];
                const int dn = (lane + 1 < nsub) ? udraw[wv * 4 + lane + 1] : (wv + 1 < C) ? udraw[(wv + 1) * 4] : c.draws[blk];
                const int4 urL = urec[wv * 4 + lane];
                const float rrL = utail[wv * 4 + lane].y;
                const uint32_t seedL = rng_jump(c.jumpT, bseed[kb], (uint32_t)dj);
                int *u = uh + lane * 8;
                u[0] = (int)seedL; u[1] = urL.x; u[2] = urL.y; u[3] = urL.z; u[4] = urL.w; u[5] = dn - dj; u[6] = __float_as_int(rrL);
            }
            WAVE_SYNC();
        }
        // ---- coefficients -> spectra, this wave's channel, subblock by subblock (independent of each other)
        {
            unsigned pat = pat0; int off = 0;
            for (int j = 0; j < nsub; j++, pat >>= 4) {
                const int d = pat & 7, S = BS >> d, M = S >> 1, Mp = FFT_PADDEDS(M, DPS);
                float2 *zj = zc + FFT_PADS(off >> 1, DPS);
                sw.A = (float *)zj;
                const int *uj = uh + j * 8;
                const uint32_t unitSeed = whole ? bseed[wv * 64 + kb] : (uint32_t)uj[0];
                STAMP(1);
                const int4 ur = whole ? make_int4(hb[1], hb[2], hb[3], hb[4]) : make_int4(uj[1], uj[2], uj[3], uj[4]);
                const int ud = whole ? hb[5] : uj[5];
                const float urr = __int_as_float(whole ? hb[6] : uj[6]);
                if (!(ULCX_DBG(c) & 1)) synth_unit(c, sw, S, prec, nrec, ur, unitSeed, ud, urr, tmag + (size_t)(wv * 4 + j) * c.tailStride, Mp);
                else for (int i = lane; i < Mp; i += 64) zj[i] = make_float2(0.0f, 0.0f);
                STAMP(2);

                // DCT-IV pre-twiddle in place: n and M-1-n together read and write the same two complex slots
                const float2 *pre = c.T.pre[d];
                if (!(ULCX_DBG(c) & 2)) {
                if (BSC == 2048 && whole) {
                    // the headline geometry: eight trips per lane, four at a time with every load of the four in front of the
                    // arithmetic (a trip is a chain load -> multiply -> store; in a loop the eight chains run one after the other)
                    constexpr int MC = 1024;
#pragma unroll
                    for (int h = 0; h < 2; h++) {
                        float2 va[4], vb[4], p1[4], p2[4];
#pragma unroll
                        for (int t = 0; t < 4; t++) {
                            const int n = lane + 64 * (4 * h + t), n2 = MC - 1 - n;
                            va[t] = zj[FFT_PADS(n, DPS)]; vb[t] = zj[FFT_PADS(n2, DPS)]; p1[t] = pre[n]; p2[t] = pre[n2];
                        }
#pragma unroll
                        for (int t = 0; t < 4; t++) {
                            const int n = lane + 64 * (4 * h + t), n2 = MC - 1 - n;
                            zj[FFT_PADS(n, DPS)]  = cmulc(make_float2(va[t].x, vb[t].y), p1[t]);
                            zj[FFT_PADS(n2, DPS)] = cmulc(make_float2(vb[t].x, va[t].y), p2[t]);
                        }
                    }
                } else {
#pragma unroll
                for (int n = lane; n < M / 2; n += 64) {
                    const int n2 = M - 1 - n;
                    const int pn = FFT_PADS(n, DPS), pn2 = FFT_PADS(n2, DPS);
                    const float2 a = zj[pn], b = zj[pn2];            // (X[2n], X[2n+1]), (X[S-2-2n], X[S-1-2n])
                    zj[pn]  = cmulc(make_float2(a.x, b.y), pre[n]);
                    zj[pn2] = cmulc(make_float2(b.x, a.y), pre[n2]);
                }
                }
                STAMP(11);
                if constexpr (!TWL) { if (M == 1024) fft_wave_dif_ct<1024, DPS>(zj, c.T.tw[d], lane); else fft_wave_dif(zj, M, c.T.tw[d], lane, DPS); }
                else if (M == 1024) fft_wave_dif_ct<1024, DPS>(zj, twl, lane);     // (BlockSize 2048, un-decimated: index arithmetic folded at compile time)
                else fft_wave_dif(zj, M, twl + (d <= 1 ? 0 : d == 2 ? BS / 8 : 3 * BS / 16), lane, DPS);
                }
                STAMP(3);
                off += S;
            }
        }
        if (whole) {
            // ---- un-decimated block (the common case), both channels together: post-twiddle fused with the windowed
            //      overlap-add (oracle/orc_fourier.c orc_imdct), inverse M/S in registers, interleaved stores
            int ov = BS;                                             // ulcDecoder.c:234-239
            if (pat0 & 8) ov >>= (wc & 7);
            if (ov > lastSub) ov = lastSub;
            if constexpr ((BSC == 2048 || BSC == 4096) && DSYN_EPI2) {
                // Headline geometry and BlockSize 4096 (round 5): a two-deep pipeline over the thread's four (eight) trips.  What a trip reads from global
                // memory (two post-twiddles, the pending halves of both channels, the window pair: 12 registers) is asked for one
                // trip ahead - the first trip's BEFORE the barrier, where the transform's registers are free and the other
                // wave may still be transforming.  (All four at once spill: 1.70 -> 1.89 ms.)  Window pair: p1 = p0 - 1 is even
                // and so are a and ov: aligned 8-byte loads; below the ramp the index is clamped, the value unused.
                constexpr int S = BSC, M = BSC / 2, NT = M / 2 / WG;
                const float2 *pre = c.T.pre[0];
                const float2 *z0 = z, *z1 = z + Mp0;
                const int a = (S - ov) >> 1;
                const float *fall = c.T.winFall + ov, *rise = c.T.winRise + ov;
                float *L0 = lapW, *L1 = lapW + H2;
                struct Ops { float2 P1, P2, Am, As, F, R; };
                auto fetch = [&](int t) {
                    Ops o;
                    const int k1 = tid + WG * t, k2 = M - 1 - k1;
                    o.P1 = pre[k1]; o.P2 = pre[k2];
                    o.Am = make_float2(0.0f, 0.0f); o.As = o.Am; o.F = o.Am; o.R = o.Am;
                    if (!warm) {
                        o.Am = *(const float2 *)(lapR + 2 * k1); o.As = *(const float2 *)(lapR + H2 + 2 * k1);
                        int wi = M - 2 - 2 * k1 - a; wi = wi < 0 ? 0 : wi;
                        o.F = *(const float2 *)(fall + wi); o.R = *(const float2 *)(rise + wi);
                    }
                    return o;
                };
                // The 8-byte window loads want a, ov and p1 even: ov = BS >> (0..7) capped by lastSub >= BS / 8, both multiples of 4
                // for every BlockSize this branch is compiled for (>= 2048), so a = (S - ov) / 2 is even too.
                static_assert(BSC >= 2048 && (BSC >> 7) % 4 == 0 && (BSC >> 3) % 4 == 0, "k_dsyn: the pipelined epilogue loads window pairs as float2");
                Ops nxt = fetch(0);
                __syncthreads();
                STAMP(4);
                if (!(ULCX_DBG(c) & 4))                             // (ablation builds only: the epilogue skipped, as in the generic loop below)
#pragma unroll
                for (int t = 0; t < NT; t++) {
                    const Ops cur = nxt;
                    if (t + 1 < NT) nxt = fetch(t + 1);
                    const int k1 = tid + WG * t, k2 = M - 1 - k1;
                    constexpr int LGM = BSC == 4096 ? 11 : 10;
                    int r1 = (int)(__brev((unsigned)k1) >> (32 - LGM)), r2 = (int)(__brev((unsigned)k2) >> (32 - LGM));
                    r1 = FFT_PADS(r1, DPS); r2 = FFT_PADS(r2, DPS);
                    const float2 ya1 = cmulc_post(z0[r1], cur.P1), ya2 = cmulc_post(z0[r2], cur.P2);     // channel 0 (M): (Re y, -Im y)
                    const float2 yb1 = cmulc_post(z1[r1], cur.P1), yb2 = cmulc_post(z1[r2], cur.P2);     // channel 1 (S)
                    if (!warm) {
                        const float Bm[2] = { ya1.y, ya2.x }, Bs[2] = { yb1.y, yb2.x };
                        const float Am[2] = { cur.Am.x, cur.Am.y }, As[2] = { cur.As.x, cur.As.y };
                        const float Fw[2] = { cur.F.y, cur.F.x }, Rw[2] = { cur.R.y, cur.R.x };        // (the pair is {p1, p0})
                        const int   pv[2] = { M - 1 - 2 * k1, M - 2 - 2 * k1 };
                        float2 lo2[2], hi2[2];
#pragma unroll
                        for (int q = 0; q < 2; q++) {
                            float mLo, mHi, sLo, sHi;
                            if (pv[q] < a) { mLo = Am[q]; mHi = Bm[q]; sLo = As[q]; sHi = Bs[q]; }
                            else {
                                const float cw = Fw[q], sn = Rw[q];
                                const float m1 = sn * Bm[q], m3 = cw * Bm[q];
                                mLo = __builtin_fmaf(cw, Am[q], -m1); mHi = __builtin_fmaf(sn, Am[q], m3);
                                const float s1 = sn * Bs[q], s3 = cw * Bs[q];
                                sLo = __builtin_fmaf(cw, As[q], -s1); sHi = __builtin_fmaf(sn, As[q], s3);
                            }
                            lo2[q] = make_float2(mLo + sLo, mLo - sLo);
                            hi2[q] = make_float2(mHi + sHi, mHi - sHi);
                        }
                        st4(outp + 2 * pv[1], lo2[1].x, lo2[1].y, lo2[0].x, lo2[0].y);
                        st4(outp + 2 * (S - 1 - pv[0]), hi2[0].x, hi2[0].y, hi2[1].x, hi2[1].y);
                    }
                    *(float2 *)(L0 + 2 * k1) = make_float2(ya1.x, ya2.y);
                    *(float2 *)(L1 + 2 * k1) = make_float2(yb1.x, yb2.y);
                }
            } else {
            __syncthreads();
            STAMP(4);
            if (!(ULCX_DBG(c) & 4)) {
                const int S = BS, M = M0;
                const float2 *pre = c.T.pre[0];
                const float2 *z0 = z, *z1 = z + Mp0;
                const int a = (S - ov) >> 1;
                const float *fall = c.T.winFall + ov, *rise = c.T.winRise + ov;
                const int bits = 31 - __clz(M);
                const float *R0 = lapR, *R1 = lapR + H2;
                float *L0 = lapW, *L1 = lapW + H2;
#pragma unroll
                for (int kk = tid; kk < M / 2; kk += WG) {
                    const int k1 = kk, k2 = M - 1 - kk;
                    int r1 = (int)(__brev((unsigned)k1) >> (32 - bits));
                    int r2 = (int)(__brev((unsigned)k2) >> (32 - bits));
                    r1 = FFT_PADS(r1, DPS); r2 = FFT_PADS(r2, DPS);
                    const float2 P1 = pre[k1], P2 = pre[k2];
                    const float2 ya1 = cmulc_post(z0[r1], P1), ya2 = cmulc_post(z0[r2], P2);     // channel 0 (M): (Re y, -Im y)
                    const float2 yb1 = cmulc_post(z1[r1], P1), yb2 = cmulc_post(z1[r2], P2);     // channel 1 (S)
                    if (!warm) {
                    const float A0m = R0[2 * k1], A1m = R0[2 * k1 + 1], A0s = R1[2 * k1], A1s = R1[2 * k1 + 1];
                    const float Bm[2] = { ya1.y, ya2.x }, Bs[2] = { yb1.y, yb2.x };
                    const float Am[2] = { A0m, A1m }, As[2] = { A0s, A1s };
                    const int   pv[2] = { M - 1 - 2 * k1, M - 2 - 2 * k1 };
                    float2 lo2[2], hi2[2];                                             // interleaved L/R at positions p and S-1-p
#pragma unroll
                    for (int q = 0; q < 2; q++) {
                        const int p = pv[q];
                        float mLo, mHi, sLo, sHi;                                      // outputs at positions p and S-1-p
                        if (p < a) { mLo = Am[q]; mHi = Bm[q]; sLo = As[q]; sHi = Bs[q]; }
                        else {
                            const float cw = fall[p - a], sn = rise[p - a];
                            // spec v2 (orc_imdct): Out[p] = fma(c, A, -(s B)), Out[S-1-p] = fma(s, A, c B)
                            const float m1 = sn * Bm[q], m3 = cw * Bm[q];
                            mLo = __builtin_fmaf(cw, Am[q], -m1); mHi = __builtin_fmaf(sn, Am[q], m3);
                            const float s1 = sn * Bs[q], s3 = cw * Bs[q];
                            sLo = __builtin_fmaf(cw, As[q], -s1); sHi = __builtin_fmaf(sn, As[q], s3);
                        }
                        // inverse M/S (ulcDecoder.c:281-289) + interleave (:292-297)
                        lo2[q] = make_float2(mLo + sLo, mLo - sLo);
                        hi2[q] = make_float2(mHi + sHi, mHi - sHi);
                    }
                    // positions pv[1] = pv[0]-1 and S-1-pv[0], S-pv[0] are neighbours: two aligned 16-byte stores
                    st4(outp + 2 * pv[1], lo2[1].x, lo2[1].y, lo2[0].x, lo2[0].y);
                    st4(outp + 2 * (S - 1 - pv[0]), hi2[0].x, hi2[0].y, hi2[1].x, hi2[1].y);
                    }
                    L0[2 * k1] = ya1.x; L0[2 * k1 + 1] = ya2.y;
                    L1[2 * k1] = yb1.x; L1[2 * k1 + 1] = yb2.y;
                }
            }
            }
            STAMP(5);
            __syncthreads();
            STAMP(6);
            lastSub = BS;
        } else {
            // ---- decimated block: each wave finishes its channel in LDS, then both channels together:
            //      inverse M/S (ulcDecoder.c:281-289) + interleave (:292-297)
            // (the time-domain pass works on the lapping state in place: in the workgroup's own rows)
            if (LAPG && lapR != scr && !warm) { for (int i = tid; i < 2 * H2; i += WG) scr[i] = lapR[i]; __syncthreads(); }
            float *lapD = LAPG ? scr : ldsLap;
            WAVE_SYNC();
            const int newLast = dec_time_wave<DEC_MAXT>(c, zc, lapD + wv * H2, wc, pat0, nsub, lastSub, lane);
            __syncthreads();
            // output = times [-BS/2, BS/2) of both channels: the old pending lists (time -1-i at lap[i]), then the arrays
            const float *t0 = (const float *)z, *t1 = (const float *)(z + Mp0);
            const float *L0 = lapD, *L1 = lapD + H2;
            if (!warm) {
            for (int n = 2 * tid; n < H2; n += 2 * WG) {
                const float mx = L0[H2 - 1 - n], my = L0[H2 - 2 - n], sx = L1[H2 - 1 - n], sy = L1[H2 - 2 - n];
                st4(outp + 2 * n, mx + sx, mx - sx, my + sy, my - sy);
            }
            for (int n = 2 * tid; n < H2; n += 2 * WG) {
                const float2 m = *(const float2 *)(t0 + padf(n)), sd = *(const float2 *)(t1 + padf(n));
                st4(outp + 2 * (H2 + n), m.x + sd.x, m.x - sd.x, m.y + sd.y, m.y - sd.y);
            }
            }
            __syncthreads();
            // new pending lists: times [BS/2, BS) of the arrays
            for (int i = tid; i < 2 * H2; i += WG) {
                const int ch = i >= H2 ? 1 : 0, m = i - ch * H2;
                lapW[ch * H2 + H2 - 1 - m] = ((const float *)(z + ch * Mp0))[padf(H2 + m)];
            }
            __syncthreads();
            lastSub = newLast;
            STAMP(7);
        }
        if (lastOfStream && tid == 0) { c.lastSubO[s] = lastSub; c.seedO[s] = seed; c.deadO[s] = dead; }
    }
#ifdef ULCX_DSYN_STAMPS
    if (lane == 0) for (int i = 0; i < DSYN_NSTAMP; i++) ((unsigned long long *)(c.scratch + (size_t)s * 4 * BS))[wv * DSYN_NSTAMP + i] = stq.t[i];
#endif
    if (!LAPG && sCur >= 0) { __syncthreads(); float *go = c.lapO + (size_t)sCur * C * H2; for (int i = tid; i < 2 * H2; i += WG) go[i] = ldsLap[i]; }
}

// ---------------------------------------------------------------------------
// Everything else (mono / multichannel / BlockSize > 4096): one (channel, subblock) at a time through ONE LDS array,
// the lapping state and the staging of the time samples in global memory.  Correct for every geometry the
// reference accepts (ulcDecoder.c:33-35: up to 255 channels, BlockSize up to 32768), not tuned.
// ---------------------------------------------------------------------------
template <typename OUT>
__global__ __launch_bounds__(WG) void k_dgen(UlcxDecCtx c) {
    extern __shared__ float lds[];
    const int BS = c.BS, C = c.C, H2 = BS / 2;
    const int s = c.s0 + blockIdx.x, tid = threadIdx.x;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const DsynLds L = dsyn_lds(BS, C, 0, 0);
    float2 *z = (float2 *)lds;
    SynWave sw;
    sw.pre  = (int *)(lds + L.zFloats) + wv * (64 + DSYN_PWORDS(BS));
    sw.seedTab = (uint32_t *)(sw.pre + 64);
    sw.lane = lane;
#ifdef ULCX_DSYN_STAMPS
    sw.stp = nullptr;
#endif
    float *glap = c.lap + (size_t)s * C * H2;
    int lastSub = c.lastSub[s];
    int dead = c.dead[s];
    uint32_t seed = c.seed[s];
    float *scr = c.scratch + (size_t)s * 4 * BS;                     // staging: dst[2][BS] | dec[BS] | tmpq[BS/2]
    for (int k = 0; k < c.K; k++) {
        const int blk = s * c.K + k;
        const int wc = c.wcScan[blk];
        if (wc == 0) dead = 1;                                       // a corrupt block ends the stream (ulcDecodeTool.c:154-157)
        OUT *outp = out_base<OUT>(c) + (size_t)blk * C * BS;
        if (dead) {
            for (int i = tid; i < C * BS; i += WG) st1(outp + i, 0.0f);
            if (tid == 0) c.bits[blk] = 0;
            continue;
        }
        const int *udraw = c.unitDraws + (size_t)blk * C * 4;
        const int4 *urec = c.unitRec + (size_t)blk * C * 4;
        const float4 *utail = c.unitTail + (size_t)blk * C * 4;
        const uint2 *prec = c.prec + (size_t)blk * c.precStride, *nrec = c.nrec + (size_t)blk * c.nrecStride;
        const float *tmag = c.tailMag + (size_t)blk * C * 4 * c.tailStride;
        const unsigned pat0 = ulcx_pattern(wc);
        const bool whole = (BS >> (pat0 & 7)) == BS;                 // one subblock per channel (ulcDecoder.c:242-245)
        int nsub = 1;
        if (!whole) { nsub = 0; unsigned q = pat0; do nsub++; while (q >>= 4); }
        auto unit_draws = [&](int ch, int j) { return ((j + 1 < nsub) ? udraw[ch * 4 + j + 1] : (ch + 1 < C) ? udraw[(ch + 1) * 4] : c.draws[blk]) - udraw[ch * 4 + j]; };
        {
            float *dec = scr + 2 * BS, *tmpq = scr + 3 * BS;
            int newLast = lastSub;
            for (int ch = 0; ch < C; ch++) {
                int last = lastSub;                                  // ulcDecoder.c:219
                float *dst = scr + (size_t)(ch & 1) * BS;
                float *Lp = glap + (size_t)ch * H2;
                unsigned pat = pat0;
                int dpos = 0, j = 0;
                do {
                    const int d = pat & 7, S = BS >> d, M = S >> 1, Mp = FFT_PADDEDS(M, DPS);
                    int ov = S;                                      // ulcDecoder.c:234-239
                    if (pat & 8) ov >>= (wc & 7);
                    if (ov > last) ov = last;
                    last = S;
                    for (int i = tid; i < Mp; i += WG) z[i] = make_float2(0.0f, 0.0f);
                    __syncthreads();
                    if (wv == 0) {
                        sw.A = (float *)z;
                        const uint32_t unitSeed = rng_jump(c.jumpT, seed, (uint32_t)udraw[ch * 4 + j]);
                        synth_unit(c, sw, S, prec, nrec, urec[ch * 4 + j], unitSeed, unit_draws(ch, j), utail[ch * 4 + j].y, tmag + (size_t)(ch * 4 + j) * c.tailStride);
                    }
                    __syncthreads();
                    const float2 *pre = c.T.pre[d];
                    for (int n = tid; n < M / 2; n += WG) {
                        const int n2 = M - 1 - n;
                        const int pn = FFT_PADS(n, DPS), pn2 = FFT_PADS(n2, DPS);
                        const float2 a = z[pn], b = z[pn2];
                        z[pn]  = cmulc(make_float2(a.x, b.y), pre[n]);
                        z[pn2] = cmulc(make_float2(b.x, a.y), pre[n2]);
                    }
                    __syncthreads();
                    if (wv == 0) fft_wave_dif(z, M, c.T.tw[d], lane, DPS);
                    __syncthreads();
                    // post-twiddle fused with the windowed overlap (oracle/orc_fourier.c orc_imdct):
                    //   zz[2k] = Re y[k], zz[S-1-2k] = -Im y[k];  pair p: A = lap[M-1-p], B = zz[M+p]
                    float *out = (S == BS) ? dst : dec;
                    const int a = (S - ov) >> 1;
                    const float *fall = c.T.winFall + ov, *rise = c.T.winRise + ov;
                    const int bits = 31 - __clz(M);
                    for (int kk = tid; kk < M / 2; kk += WG) {
                        const int k1 = kk, k2 = M - 1 - kk;
                        const int r1 = FFT_PADS((int)(__brev((unsigned)k1) >> (32 - bits)), DPS);
                        const int r2 = FFT_PADS((int)(__brev((unsigned)k2) >> (32 - bits)), DPS);
                        const float2 y1 = cmulc_post(z[r1], pre[k1]), y2 = cmulc_post(z[r2], pre[k2]);      // (Re y, -Im y)
                        // zz[2k1] = Re y1, zz[2k1+1] = -Im y2 (new lap);  zz[S-1-2k1] = -Im y1, zz[S-2-2k1] = Re y2 (B values)
                        const float A0 = Lp[2 * k1], A1 = Lp[2 * k1 + 1];
                        const float Bv[2] = { y1.y, y2.x };
                        const float Av[2] = { A0, A1 };
                        const int   pv[2] = { M - 1 - 2 * k1, M - 2 - 2 * k1 };
#pragma unroll
                        for (int q = 0; q < 2; q++) {
                            const int p = pv[q];
                            const float A = Av[q], B = Bv[q];
                            if (p < a) { out[p] = A; out[S - 1 - p] = B; }
                            else {
                                const float cw = fall[p - a], sn = rise[p - a];
                                const float m1 = sn * B, m3 = cw * B;       // spec v2: fused (orc_imdct)
                                out[p] = __builtin_fmaf(cw, A, -m1);
                                out[S - 1 - p] = __builtin_fmaf(sn, A, m3);
                            }
                        }
                        Lp[2 * k1] = y1.x;
                        Lp[2 * k1 + 1] = y2.y;
                    }
                    __syncthreads();
                    if (S == BS) break;                              // ulcDecoder.c:242-245
                    // reversed-time centring FIFO in lap[M .. BS/2) (ulcDecoder.c:253-272)
                    const int avail = (BS - S) >> 1;
                    for (int q = tid; q < avail; q += WG) tmpq[q] = Lp[H2 - 1 - q];      // queue[q], q = 0 is the oldest
                    __syncthreads();
                    for (int n = tid; n < S; n += WG)
                        dst[dpos + n] = (n < avail) ? tmpq[n] : dec[n - avail];
                    if (S <= avail) {
                        for (int q = tid; q < avail; q += WG)
                            Lp[H2 - 1 - q] = (q < avail - S) ? tmpq[q + S] : dec[q - (avail - S)];
                    } else {
                        for (int q = tid; q < avail; q += WG) Lp[H2 - 1 - q] = dec[S - avail + q];
                    }
                    __syncthreads();
                    dpos += S; j++;
                } while (pat >>= 4);
                newLast = last;
                // inverse M/S + interleave once both members of a pair (or a trailing single) are staged
                const bool pairDone = (ch & 1) || (ch == C - 1);
                if (pairDone) {
                    __syncthreads();
                    if (ch & 1) {
                        const float *dm = scr, *ds = scr + BS;
                        for (int n = tid; n < BS; n += WG) {
                            const float m = dm[n], sd = ds[n];                            // ulcDecoder.c:281-289
                            const float l = m + sd, r = m - sd;
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
        seed = rng_jump(c.jumpT, seed, (uint32_t)c.draws[blk]);      // the one RNG chain of the stream (ulcDecoder.c:75-81)
    }
    if (tid == 0) { c.lastSub[s] = lastSub; c.seed[s] = seed; c.dead[s] = dead; }
}

size_t ulcx_dec_lds_bytes(int BS, int C, int fast, int twInLds) {
    DsynLds l = dsyn_lds(BS, C, fast, twInLds);
    return sizeof(float) * ((size_t)l.zFloats + l.lapFloats + l.twFloats + l.listFloats);
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { ulcx_set_error("%s: %s", #x, hipGetErrorString(e_)); return ULCX_ERR_HIP; } } while (0)

// resident workgroups of the stereo synthesis kernel this context runs (float output; the PCM16 instantiation has the same
// resources): what an even cut of the batch is sized for
// the instantiation launch_syn() starts for this context (the one whose occupancy and dynamic-LDS attribute count)
template <typename OUT>
static const void *syn_fn(const UlcxDecCtx &cc, bool split) {
    const bool small = cc.BS <= 2048;
    if (!cc.fastOK) return (const void *)k_dgen<OUT>;
    if (cc.BS == 2048 && DSYN_C2048) return split ? (const void *)k_dsyn<OUT, 16, DSYN_TWL != 0, true, 2048> : (const void *)k_dsyn<OUT, 16, DSYN_TWL != 0, false, 2048>;
    if (small) return split ? (const void *)k_dsyn<OUT, 16, DSYN_TWL != 0, true> : (const void *)k_dsyn<OUT, 16, DSYN_TWL != 0, false>;
    if (cc.BS == 4096 && DSYN_C4096) return split ? (const void *)k_dsyn<OUT, 32, false, true, 4096> : (const void *)k_dsyn<OUT, 32, false, false, 4096>;
    return split ? (const void *)k_dsyn<OUT, 32, false, true> : (const void *)k_dsyn<OUT, 32, false, false>;
}
int ulcx_dec_syn_slots(const UlcxDecCtx &c) {
    if (!c.fastOK) return 0;
    // what a cut would launch: the float and the PCM16 instantiation are both asked, the smaller residency sizes the cut
    const void *fns[2] = { syn_fn<float>(c, true), syn_fn<int16_t>(c, true) };
    const size_t lds = ulcx_dec_lds_bytes(c.BS, c.C, c.fastOK, c.twInLds);
    int dev = 0, cus = 0, per = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) { (void)hipGetLastError(); return 0; }
    for (const void *fn : fns) {
        int p1 = 0;
        if (lds > 48 * 1024 && hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) { (void)hipGetLastError(); return 0; }
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&p1, fn, WG, lds) != hipSuccess) { (void)hipGetLastError(); return 0; }
        per = (fn == fns[0] || p1 < per) ? p1 : per;
    }
    return cus * per;
}

// Two kernels on the caller's stream: the syntax walk, then the synthesis.  (Measured and dropped, profiles/NOTES_r01-r04.md:
// the walk of a chunk of streams / of the second half of the blocks beside the synthesis of the previous one - the
// synthesis slows down by more than the walk it hides.)
template <typename OUT>
static void launch_syn(const UlcxDecCtx &cc, unsigned g, size_t lds, hipStream_t s2, bool split) {
    const bool small = cc.BS <= 2048;
    if (!cc.fastOK) hipLaunchKernelGGL(k_dgen<OUT>, dim3(g), dim3(WG), lds, s2, cc);
    else if (cc.BS == 2048 && DSYN_C2048) { if (split) hipLaunchKernelGGL((k_dsyn<OUT, 16, DSYN_TWL != 0, true, 2048>), dim3(g), dim3(WG), lds, s2, cc); else hipLaunchKernelGGL((k_dsyn<OUT, 16, DSYN_TWL != 0, false, 2048>), dim3(g), dim3(WG), lds, s2, cc); }
    else if (small) { if (split) hipLaunchKernelGGL((k_dsyn<OUT, 16, DSYN_TWL != 0, true>), dim3(g), dim3(WG), lds, s2, cc); else hipLaunchKernelGGL((k_dsyn<OUT, 16, DSYN_TWL != 0, false>), dim3(g), dim3(WG), lds, s2, cc); }
    else if (cc.BS == 4096 && DSYN_C4096) { if (split) hipLaunchKernelGGL((k_dsyn<OUT, 32, false, true, 4096>), dim3(g), dim3(WG), lds, s2, cc); else hipLaunchKernelGGL((k_dsyn<OUT, 32, false, false, 4096>), dim3(g), dim3(WG), lds, s2, cc); }
    else { if (split) hipLaunchKernelGGL((k_dsyn<OUT, 32, false, true>), dim3(g), dim3(WG), lds, s2, cc); else hipLaunchKernelGGL((k_dsyn<OUT, 32, false, false>), dim3(g), dim3(WG), lds, s2, cc); }
}
int ulcx_dec_launch(const UlcxDecCtx &cIn, hipStream_t st, hipEvent_t *ev, const UlcxDecAux &aux) {
    int stage = 0;
    if (ev) CK(hipEventRecord(ev[stage++], st));
    UlcxDecCtx c = cIn;
    c.s0 = 0; c.s1 = c.B; c.k0 = 0; c.k1 = c.K;
    const size_t lds = ulcx_dec_lds_bytes(c.BS, c.C, c.fastOK, c.twInLds);
    if (lds > 48 * 1024) {
        const bool split = c.fastOK && aux.synGrid > 0;
        const void *fn = c.pcm16 ? syn_fn<int16_t>(c, split) : syn_fn<float>(c, split);
        CK(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    }
    if (c.packed) hipLaunchKernelGGL(k_dscan_packed, dim3((c.B + 63) / 64), dim3(64), 0, st, c);
    else hipLaunchKernelGGL(k_dscan, dim3((c.B * c.K + 63) / 64), dim3(64), 0, st, c);
    if (ev) CK(hipEventRecord(ev[stage++], st));
    if (!(ULCX_DBG(c) & 8)) {
        const bool split = c.fastOK && aux.synGrid > 0;
        const unsigned g = split ? (unsigned)aux.synGrid : (unsigned)c.B;
        c.synFull = split ? aux.synFull : 0;
        if (c.pcm16) launch_syn<int16_t>(c, g, lds, st, split); else launch_syn<float>(c, g, lds, st, split);
    }
    if (ev) CK(hipEventRecord(ev[stage++], st));
    CK(hipGetLastError());
    return ULCX_OK;
}

// ---------------------------------------------------------------------------
// Slots -> contiguous per-stream payloads (tools/ulcEncodeTool.c:160-169)
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_pack_streams(int nBlocks, int slotBytes, const uint8_t *slots, const int32_t *bits,
                                                       uint8_t *payload, long long stride, int32_t *payloadBytes, int32_t *maxBlock) {
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
    if (tid == 0) { payloadBytes[s] = off; if (maxBlock) maxBlock[s] = mx; }
}
int ulcx_pack_launch(int nStreams, int nBlocks, int slotBytes, const uint8_t *d_slots, const int32_t *d_bits, uint8_t *d_payload,
                     long long stride, int32_t *d_payloadBytes, int32_t *d_maxBlock, hipStream_t st) {
    hipLaunchKernelGGL(k_pack_streams, dim3(nStreams), dim3(256), 0, st, nBlocks, slotBytes, d_slots, d_bits, d_payload, stride, d_payloadBytes, d_maxBlock);
    CK(hipGetLastError());
    return ULCX_OK;
}
