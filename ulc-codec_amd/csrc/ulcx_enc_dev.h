// ulcx_enc_dev.h - what the encoder's translation units share (round 5: ulcx_enc.hip split by phase): device helpers, the
// LDS geometry macros the host-side launch code needs too, and the declarations of every kernel (the launch sequence in
// ulcx_enc.hip starts kernels defined in ulcx_enc_wc.hip / ulcx_enc_xf.hip / ulcx_enc_psy.hip / ulcx_enc_wr.hip; template kernels are
// instantiated explicitly where they are defined).  Not part of the public ABI.
#pragma once
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

#ifndef EF_NW
#define EF_NW 9                                           // waves per workgroup: the chain + one producer per stream
#endif

#define EF_TILE_FLOATS (EF_SPW * 2 * EF_TS)

#define EF_LDS_BYTES (EF_RT * EF_TILE_FLOATS * 4 + 4 * EF_NW)

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
    const float4 t = c.T.bandW[d][line];                   // {left index, right index (clamped), 1 - frac, frac}
    const float *bark = bark4 + j * ULCX_NBARK;
    const float noise = bark[__float_as_int(t.x)] * t.z + bark[__float_as_int(t.y)] * t.w;
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

#ifndef WAVE_SK
#define WAVE_SK 512
#endif
#ifndef WAVE_SZ
#define WAVE_SZ 128
#endif
#ifndef WAVE_SN
#define WAVE_SN 1280                                       // (round 5: 2048 -> 1280, 7 360 -> 6 592 bytes per wave: six workgroups per CU)
#endif

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

// wave-local ordering of LDS traffic (all 64 lanes run in lockstep; LDS ops of one wave complete in order)
#define WAVE_SYNC() do { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); } while (0)

// Direct packing (round 3): in the final pass of a stereo, un-decimated block - one unit per channel - the two waves of
// the block write their bytes straight into the output slot instead of staging rows that k_pack shifts into place
// (Encode.c:329-359: header nybble, channel 0, channel 1, byte aligned).  xch: the pair's LDS word, through which the
// channel-0 wave tells its partner {nybbles of channel 0, its last nybble, failed, trip number}.
#define XCH_WORD(total, last, fail, seq) ((unsigned long long)((uint32_t)(total) | ((uint32_t)(last) << 16) | ((uint32_t)(fail) << 20)) | ((unsigned long long)(uint32_t)(seq) << 32))

// ---- kernels (defined in the unit named in ulcx_enc.hip's header comment)
template <typename IN>
__global__ __launch_bounds__(WG) void k_wc_energy(UlcxEncCtx c, int k0, int k1);
extern template __global__ void k_wc_energy<float>(UlcxEncCtx, int, int);
extern template __global__ void k_wc_energy<int16_t>(UlcxEncCtx, int, int);
__global__ void k_wc_forward(UlcxEncCtx c, int k0, int k1);
template <int NW, typename IN>
__global__ __launch_bounds__(NW * 64) void k_wc_ef(UlcxEncCtx c, int k0, int k1);
extern template __global__ void k_wc_ef<EF_NW, float>(UlcxEncCtx, int, int);
extern template __global__ void k_wc_ef<EF_NW, int16_t>(UlcxEncCtx, int, int);
__global__ void k_wc_backward(UlcxEncCtx c, int k0, int k1);
__global__ void k_wc_integrate(UlcxEncCtx c, int k0, int k1);
__global__ void k_wc_decide(UlcxEncCtx c, int k0, int k1);
template <bool ST, typename IN>
__global__ __launch_bounds__(WG, 4) void k_xf(UlcxEncCtx c, int k0, int k1);
extern template __global__ void k_xf<false, float>(UlcxEncCtx, int, int);
extern template __global__ void k_xf<false, int16_t>(UlcxEncCtx, int, int);
extern template __global__ void k_xf<true, float>(UlcxEncCtx, int, int);
extern template __global__ void k_xf<true, int16_t>(UlcxEncCtx, int, int);
template <typename IN>
__global__ __launch_bounds__(WG) void k_xf_big(UlcxEncCtx c, int k0, int k1);
extern template __global__ void k_xf_big<float>(UlcxEncCtx, int, int);
extern template __global__ void k_xf_big<int16_t>(UlcxEncCtx, int, int);
__global__ void k_cplx(UlcxEncCtx c, int k0, int k1);
__global__ void k_nbark(UlcxEncCtx c, int useList);
__global__ void k_nline(UlcxEncCtx c);
__global__ void k_pbark(UlcxEncCtx c, int useList);
template <bool NOISE>
__global__ __launch_bounds__(256) void k_bark_uniform(UlcxEncCtx c);
extern template __global__ void k_bark_uniform<false>(UlcxEncCtx);
extern template __global__ void k_bark_uniform<true>(UlcxEncCtx);
template <bool NOISE>
__global__ __launch_bounds__(WG) void k_bark_levels(UlcxEncCtx c);
extern template __global__ void k_bark_levels<false>(UlcxEncCtx);
extern template __global__ void k_bark_levels<true>(UlcxEncCtx);
__global__ void k_keys_finalize(UlcxEncCtx c);
__global__ void k_select(UlcxEncCtx c, int finalPass);
// A kernel TEMPLATE's __launch_bounds__ must stand on the declaration its explicit instantiations see: written on the
// definition alone they are dropped without a word and the kernel is compiled for 1024-thread workgroups - 128 registers at
// most (round 5 found k_select_wave<128, ...> spilling 340 bytes under that cap after the split into translation units).
// The selection keeps its keys in registers: R = 64 takes 78 (the headline instantiation: six waves per SIMD) to 96, R = 128 140-160.
#define SEL_MINW(R, LGBS, PASS) ((R) >= 128 ? 2 : ((R) == 64 && !((LGBS) == 11 && (PASS) == 0)) ? 5 : 6)
template <int R, int LGBS = 0, int PASS = 0>
__global__ __launch_bounds__(256, SEL_MINW(R, LGBS, PASS)) void k_select_wave(UlcxEncCtx c, int finalPass);
extern template __global__ void k_select_wave<128, 0, 0>(UlcxEncCtx, int);
extern template __global__ void k_select_wave<128, 0, 1>(UlcxEncCtx, int);
extern template __global__ void k_select_wave<128, 0, 2>(UlcxEncCtx, int);
extern template __global__ void k_select_wave<16, 0, 0>(UlcxEncCtx, int);
extern template __global__ void k_select_wave<16, 0, 1>(UlcxEncCtx, int);
extern template __global__ void k_select_wave<16, 0, 2>(UlcxEncCtx, int);
extern template __global__ void k_select_wave<32, 0, 0>(UlcxEncCtx, int);
extern template __global__ void k_select_wave<32, 0, 1>(UlcxEncCtx, int);
extern template __global__ void k_select_wave<32, 0, 2>(UlcxEncCtx, int);
extern template __global__ void k_select_wave<4, 0, 0>(UlcxEncCtx, int);
extern template __global__ void k_select_wave<4, 0, 1>(UlcxEncCtx, int);
extern template __global__ void k_select_wave<4, 0, 2>(UlcxEncCtx, int);
extern template __global__ void k_select_wave<64, 0, 0>(UlcxEncCtx, int);
extern template __global__ void k_select_wave<64, 0, 1>(UlcxEncCtx, int);
extern template __global__ void k_select_wave<64, 0, 2>(UlcxEncCtx, int);
extern template __global__ void k_select_wave<64, 11, 0>(UlcxEncCtx, int);
extern template __global__ void k_select_wave<64, 11, 1>(UlcxEncCtx, int);
extern template __global__ void k_select_wave<64, 11, 2>(UlcxEncCtx, int);
extern template __global__ void k_select_wave<8, 0, 0>(UlcxEncCtx, int);
extern template __global__ void k_select_wave<8, 0, 1>(UlcxEncCtx, int);
extern template __global__ void k_select_wave<8, 0, 2>(UlcxEncCtx, int);
template <int R, int LGBS = 0, int PASS = 0>
__global__ __launch_bounds__(128) void k_select_pair(UlcxEncCtx c, int finalPass);
extern template __global__ void k_select_pair<64, 12, 0>(UlcxEncCtx, int);
extern template __global__ void k_select_pair<64, 12, 1>(UlcxEncCtx, int);
extern template __global__ void k_select_pair<64, 12, 2>(UlcxEncCtx, int);
__global__ void k_heapsel(UlcxEncCtx c, int ldsEntries);
__global__ void k_heapsel_pipe(UlcxEncCtx c, int fullRanking);
__global__ void k_keep_ranks(UlcxEncCtx c, int finalPass);
__global__ void k_nsums(UlcxEncCtx c, int finalPass);
__global__ void k_tails(UlcxEncCtx c, int finalPass);
__global__ void k_encode_units(UlcxEncCtx c, int finalPass);
// (the bound has to be on THIS declaration: on the definition alone it is dropped without a word, and the kernel ran at
//  88 registers = five waves per SIMD.  Six - 80 registers, 12 bytes of spills, and a workgroup's LDS down from 29.4 to 26.4 KB
//  so that six fit a CU - take the small-capacity launch from 1.30 to 1.15 ms; seven and eight are no faster.  The
//  full-capacity retry holds a CU's whole LDS: one workgroup per CU, no bound.)
#ifndef EW_LB
#define EW_LB 6
#endif
template <bool SMALL>
__global__ __launch_bounds__(256, SMALL ? EW_LB : 1) void k_encode_wave(UlcxEncCtx c, int finalPass, WaveCaps caps, int phase);
extern template __global__ void k_encode_wave<false>(UlcxEncCtx, int, WaveCaps, int);
extern template __global__ void k_encode_wave<true>(UlcxEncCtx, int, WaveCaps, int);
__global__ void k_rate_step(UlcxEncCtx c);
__global__ void k_pack(UlcxEncCtx c, int finalPass);
template <typename IN>
__global__ __launch_bounds__(WG) void k_state_update(UlcxEncCtx c);
extern template __global__ void k_state_update<float>(UlcxEncCtx);
extern template __global__ void k_state_update<int16_t>(UlcxEncCtx);
