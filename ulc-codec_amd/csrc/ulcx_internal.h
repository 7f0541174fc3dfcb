// ulcx_internal.h — structures shared by the host API (ulcx_api.cpp) and the HIP
// kernels (ulcx_enc.hip / ulcx_dec.hip).  Not part of the public ABI.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/ulc_amd.h"

#ifndef ULCX_DSYN_TWL
#define ULCX_DSYN_TWL 1                  // stereo synthesis, BlockSize <= 2048: FFT twiddles in LDS (0: from the tables in global memory; A/B builds)
#endif
#define ULCX_NBARK 25
#define ULCX_MAX_SUB 4
#define ULCX_BARK_EVENTS 52              // 25 lower edges + 25 upper edges + end of subblock (+ pad)
#define ULCX_MAX_BS_DEVICE 32768       // the reference's own limit (ulcEncoder.c:32-34); above 8192 the transform takes one array at a time (k_xf_big)
#define ULCX_COEF_EPS (0x1.0p-31f)     // include/ulcEncoder.h:36
#define ULCX_HEAP_LDS_BYTES (128 * 1024)
#define ULCX_HEAP_GRID 256
// Ablation switches for timing experiments (they break the results): compiled in only with `make EXTRA=-DULCX_ABLATE`,
// then set by ULCX_DBG_SKIP=bits at create time.  The shipped library has no such branches.
#ifdef ULCX_ABLATE
#define ULCX_DBG(c) ((c).dbgSkip)
#else
#define ULCX_DBG(c) 0
#endif

// Host-precomputed, data-independent tables: everything
// the reference evaluates with sinhf/asinhf/cos/sin or expf on arguments that depend
// only on (BlockSize, RateHz).  Built with the HOST libm in ulcx_tables.cpp, one
// set per encoder/decoder, resident in HBM (hot in L2).  Index d = log2(BS/S).
struct UlcxTables {
    const float2 *pre[ULCX_MAX_SUB];     // DCT-IV pre/post twiddle P[n] = (cos,sin)(pi(8n+1)/(8S)), n < S/2
    const float2 *tw[ULCX_MAX_SUB];      // FFT twiddle W[j] = (cos,sin)(2 pi j / (S/2)), j < S/4
    const float  *winFall;               // ramp for overlap Ov (power of two) at [Ov + i], i < Ov
    const float  *winRise;
    const int    *bandIdx[ULCX_MAX_SUB]; // per line < S/2: (int)Bark(line)         Psyopt.c:141-143,237-239
    const float  *bandFrac[ULCX_MAX_SUB];//               Bark(line) - (int)Bark(line)
    const float4 *bandW[ULCX_MAX_SUB];   // the two above ready to use (round 6): per line {index of the left Bark level, of the right one - both
                                         // clamped as Psyopt.c:143-147 / :240-244 clamp them, as int bits -, 1 - frac, frac}
    const uint32_t *barkSched;           // [2][ULCX_MAX_SUB][ULCX_BARK_EVENTS] band edges in line order (k_bark_uniform)
    // Bark band edges per subblock size (lines of the S/2-line pseudo-DFT)
    short nBeg[ULCX_MAX_SUB][ULCX_NBARK], nEnd[ULCX_MAX_SUB][ULCX_NBARK];   // noise:  [b, b+2)        Psyopt.c:198-205
    short pBeg[ULCX_MAX_SUB][ULCX_NBARK], pEnd[ULCX_MAX_SUB][ULCX_NBARK];   // psycho: [b-.75, b+.25)  Psyopt.c:109-116
};

struct UlcxWcState {                     // per stream, persistent (ulcEncoder.h:65-77)
    float tf[3];                         // TransientFilter
    int   wcPrev, wcCur;                 // WindowCtrl of block k0-1 and k0 (k0 = next call's first block)
    float binSum[8], binW[8];            // TransientBuffer R half of the last analysed block
    int   pad;
};

struct UlcxEncCtx {
    // geometry
    int B, K, C, BS, lgBS;               // streams, blocks this call, channels, block size
    int barkRing;                        // k_bark_uniform: snapshots a lane keeps of open Bark bands (power of two; 0 = k_nbark / k_pbark for every block)
    int *decList, *decCount;             // blocks of this call with a decimated window (listed by the transform): they take k_nbark / k_pbark
    int maxK;                            // allocation stride for per-call arrays
    int slot;                            // bytes per output slot
    int unitCap;                         // bytes per (chan,subblock) nybble staging row = 2*BS+32 per channel
    int mode; float p0, p1;              // rate control
    float vbrTarget;                     // 0x1.E4EFB7p3f*logf(100/Quality) (host libm), ulcEncoder.c:144
    // window-control constants (1 - rate), host expf: WindowControl.c:75,76,94,95,120
    float cHP, cBP, qHP, qBP, cBlk;
    float cplxScale;                     // BlockTransform.c:320
    int   rateHz;
    UlcxTables T;
    // inputs / outputs of this call
    int keyFinal;                        // c.key holds this call's final keys (k_keys_finalize has run): consumers read them instead of forming them
    const float *pcm; const int16_t *pcm16;      // exactly one is set: the C API's f32 input, or PCM16 ingest
    uint8_t *out; int32_t *bits; int32_t *wcOut; float *cplxOut;
    // persistent state
    float *hist;                         // [B][2*BS][C] previous two input blocks (raw, interleaved)
    UlcxWcState *wcs;                    // [B]
    // per-call scratch
    float2 *env;                         // [ceil(B/64)][maxK*BS][64] {hp,bp} energies -> envelopes -> transient curve (.x)
    float  *bins;                        // [B][maxK+1][16] {Sum[8],SumW[8]}; row 0 = previous block
    int    *wcArr;                       // [B][maxK+2] WindowCtrl of blocks k0-1 .. k0+K
    float  *coef;                        // [NB][C*BS]   normalised MDCT (TransformBuffer)
    float  *key;                         // [NB][C*BS]   key0 = FastLog(Re^2) | -inf; final keys are formed on the fly (final_key)
    float  *nsum;                        // [NB][C*BS/2] per-line |X|^2 (noise input)
    float  *npair;                       // [NB][C*BS]   {w, w*log} pairs (TransformNoise): parity tap only, allocated on its first use
    float  *amp2;                        // [NB][BS/2]
    float  *barkN;                       // [NB][C*4][25]
    float  *barkP;                       // [NB][4][25]
    double *barkRawN, *barkRawP;         // [rows][25][3] ordered sums of each band of un-decimated blocks (k_bark_uniform -> k_bark_levels)
    int    *nnz;                         // [NB]
    float  *cplx;                        // [NB]
    int    *nout;                        // [NB]   nOutCoef of the current pass
    int    *cbrLo, *cbrHi, *cbrDone;     // [NB]
    uint32_t *keep;                      // [NB][C*BS/32]
    int    *fbList; int *fbCount;        // tie-straddle fallback list
    uint8_t *unitBuf;                    // [NB][C][unitCap]
    int    *unitNyb;                     // [NB][C*4]
    int    *cbrBudget;                   // [NB] bit budget (ulcEncoder.c:96)
    uint4  *selWin;                      // [NB] rate search: {TL, TH, count(key >= TL), count(key >= TH)} - the ordered-key window later probes search
    uint32_t *selT;                      // [NB] the threshold key of the current probe
    int     selPair;                     // stereo BlockSize 4096: the selection with a wave per channel (k_select_pair); ULCX_SEL_PAIR=0: one wave per block
    int     selPass;                     // k_select_wave in a rate search: 1 = first probe (stores the ordered keys in `key`), 2 = later ones (read them); 0 = one-pass call
    int    *cbrLive;                     // [1] rate searches of the lock-step path still open (probe passes leave at once at 0)
    int    *slow;                        // [NB] wave-encoder give-up bits (1: small caps, 2: full caps -> k_encode_units); then 2 queue counters, 2 retry queues [NB]
    float2 *gapSum;                      // [NB][C*BS] {Sum, SumW} of the noise run in front of each kept coefficient (speculative)
    float  *tailSum;                     // [NB][C*4][8] five HF-extension sums + start index of the tail they assume
    int     useGapSums;
    int    *isFb;                        // [NB] 1 = a threshold tie group straddled the cut this call: block is on the exact (rank) path
    int    *ownSlot;                     // [NB] its slot in fbList
    int    *rankBuf;                     // [rankSlots][C*BS] full heapsort ranking of exact-path blocks
    int     dbgSkip;
    int     forceFb;                     // test hook (ulcx_encoder_debug_force_exact): every forceFb-th block takes the exact path whatever its ties
    int     rankSlots, fbLo, fbHi;       // resident rank slots; slot window of the current exact-path launch
    int     fbMode;                      // 0 = all blocks, 1 = skip isFb blocks, 2 = only isFb blocks
    int     useWave;                     // wave-per-unit encode pass (k_encode_wave); serial kernel only for overflow blocks
    int     directPack;                  // the wave writer packs stereo un-decimated blocks into the output slot itself (ULCX_DIRECT_PACK=0: k_pack does all)
    void   *heapScratch;                 // [ULCX_HEAP_GRID][C*BS] {key,idx} heaps, only when C*BS*8 exceeds the LDS budget
};

struct UlcxDecCtx {
    int B, K, C, BS, lgBS, maxK;
    int s0, s1;                          // streams [s0, s1) this launch works on
    int slot;
    int dbgSkip;                         // timing experiments only (ULCX_DBG_SKIP)
    UlcxTables T;
    const uint8_t *in; float *pcm; int16_t *pcm16;   // exactly one of pcm / pcm16 is set (f32 as the C API, or PCM16 output)
    long long inBytes;                   // readable extent of `in`: nothing outside [in, in + inBytes) is touched
    int32_t *bits;
    // persistent
    float *lap;                          // [B][C][BS/2] TransformInvLap
    int   *lastSub;                      // [B] LastSubBlockSize
    uint32_t *seed;                      // [B] noise RNG state (ulcDecoder.c:75-81)
    int   *dead;                         // [B] stream hit a corrupt block
    // k_dsyn with several workgroups per stream (round 3): the state after the launch goes to a second set of arrays (a
    // workgroup that enters a stream in the middle reads the state in front of the launch while the one that finishes the
    // stream writes the new one); the host swaps the two sets afterwards.  Without a split both sets are the same arrays.
    float *lapO; int *lastSubO; uint32_t *seedO; int *deadO;
    float *lapScratch;                   // [cut workgroups][C][BS/2] a workgroup's own lapping state between its blocks
    int    synFull;                      // cut launches: the first synFull workgroups take one whole stream each, the rest an even cut of the remaining streams
    int    k0, k1;                       // blocks [k0, k1) of every stream this synthesis launch works on
    // per-call scratch: what the scan leaves for the synthesis (ulcx_dec.hip)
    int   *wcScan;                       // [NB] WindowCtrl as the scan saw it (0 = corrupt)
    int   *draws;                        // [NB] RNG draws consumed by the block
    int   *unitDraws;                    // [NB][C*4] draws made in the block before each (chan,subblock) unit
    float4 *unitTail;                    // [NB][C*4] decaying-noise tail of the unit: {start level, decay, first coefficient, count}
    int4  *unitRec;                      // [NB][C*4] {first plain-run record, count, first noise record, count} of the unit
    uint2 *prec; int precStride;         // [NB][C*BS] plain-run records of the block (ulcx_dec.hip scan_block)
    uint2 *nrec; int nrecStride;         // [NB][C*BS/16 + C*4] noise records
    float *tailMag;                      // [NB][C*4][tailStride] tail level at every 32nd coefficient of the unit
    int    tailStride;                   // BS/32
    float *scratch;                      // [B][4*BS] general-path staging of time samples (decimated / non-stereo blocks)
    const uint32_t *jumpT;               // [8][16][4][256] byte tables of T^(d*16^i), T = one xorshift32 step
    const uint32_t *parT;                // [4][256][64] byte tables, lane fastest: a lane's word of a unit's sign-parity stream from the unit's start state
    int    fastOK, twInLds;              // stereo fast path / FFT twiddles resident in LDS
    // packed-stream mode (.ulc payloads): blocks are located by parsing, not by slot
    int   packed;
    long long payStride;                 // bytes between stream payloads
    const int32_t *payBytes;             // [B] valid bytes per stream
    int  *packOff;                       // [B] persistent read position of each stream
    int  *blkOff;                        // [NB] byte offset of each block inside its stream payload
};

// ulcHelper.h:24-46
__host__ __device__ static inline unsigned ulcx_pattern(int wc) {
    // (ulcHelper.h:24-46, the sixteen patterns packed four to a 64-bit constant and picked by selects and a shift: as a switch
    //  this became a table in constant memory - a scalar load with its wait at the top of every workgroup of the transform,
    //  a 64-address gather in the lane-per-block kernels)
    const unsigned i = ((unsigned)wc >> 4) & 15u;
    const unsigned long long t0 = 0x0091001900080000ull, t1 = 0x0A2102A101A2012Aull, t2 = 0x1B3213B212B3123Bull, t3 = 0xB3213B212B3123B1ull;
    const unsigned long long t = (i & 8u) ? ((i & 4u) ? t3 : t2) : ((i & 4u) ? t1 : t0);
    return (unsigned)(t >> (16u * (i & 3u))) & 0xFFFFu;
}

// host side (ulcx_tables.cpp)
struct UlcxHostTables;
int  ulcx_tables_build(UlcxTables *devT, void **devBlob, int BS, int rateHz, bool forEncoder);
void ulcx_set_error(const char *fmt, ...);

// launchers (ulcx_enc.hip / ulcx_dec.hip)
#define ULCX_ENC_STAGES 20
#define ULCX_ENC_STAGES_REPORTED (ULCX_ENC_STAGES + 1)   // + "wc_pipeline_exposed" (computed, not an event interval)
extern const char *const ulcx_enc_stage_names[ULCX_ENC_STAGES_REPORTED];
#define ULCX_DEC_STAGES 2
#define ULCX_WC_MAXCH 32    // fine steps of the window-control pipeline per call
#define ULCX_XF_MAXCH 8     // coarse transform chunks per call
#define ULCX_LDS_LIMIT (160 * 1024)     // LDS per workgroup on gfx950
// streams and events the encoder launch uses beside the caller's stream
struct UlcxEncAux {
    hipStream_t side, side2, side3;      // NULL: everything on the caller's stream
    hipEvent_t evFork, evJoin, evFork2;  // exact-path fork/join
    hipEvent_t *evWC;                    // [7 + 3*ULCX_WC_MAXCH] window-control pipeline; then noise-spectrum fork/join, k_cplx join, k_tailsums fork/join, k_state_update join
    hipEvent_t *evXf;                    // [2*ULCX_XF_MAXCH] timing pairs around each transform launch (used when ev != NULL)
    int wcPipe;                          // chunks of blocks pipelined between window control and transform; 1 = off
    int wcSteps;                         // fine steps of the window-control kernels per call (ULCX_WC_STEPS)
    int wcFuse;                          // stereo: envelope + forward recurrence in one kernel (k_wc_ef)
    int nsSlots;                         // workgroups of k_nsums the device holds at once (its persistent grid)
    int *nXf;                            // out: transform launches this call
};
int ulcx_enc_launch(const UlcxEncCtx &c, hipStream_t st, hipEvent_t *ev /* ULCX_ENC_STAGES+1 or NULL */, const UlcxEncAux &aux);
struct UlcxDecAux {
    int synGrid;                         // > 0: workgroups of the synthesis over a cut of the (stream, block) pairs; 0: one per stream
    int synFull;                         // of those, the leading ones that take one whole stream each (0: an even cut of everything)
};
int ulcx_dec_launch(const UlcxDecCtx &c, hipStream_t st, hipEvent_t *ev, const UlcxDecAux &aux);
size_t ulcx_dec_lds_bytes(int BS, int C, int fast, int twInLds);
int ulcx_dec_syn_slots(const UlcxDecCtx &c);      // resident workgroups of the stereo synthesis kernel on the current device
int ulcx_pack_launch(int nStreams, int nBlocks, int slotBytes, const uint8_t *d_slots, const int32_t *d_bits, uint8_t *d_payload,
                     long long stride, int32_t *d_payloadBytes, int32_t *d_maxBlock, hipStream_t st);
size_t ulcx_enc_xf_lds_bytes(int BS, int C);
int ulcx_enc_nsums_slots(int BS, int C);                 // resident workgroups of k_nsums on the current device
// FFT array padding of k_xf (ulcx_fft.h).  One complex per 8 makes every pass conflict-free but costs 2 KB of LDS at
// BlockSize 2048 and with it the 4th workgroup per CU: measured 2.13 ms vs 1.88 ms with one per 16.
__host__ __device__ static inline int ulcx_xf_pad_shift(int BS, int C) { (void)BS; (void)C; return 4; }
// k_select_wave: candidate keys a lane keeps once the search window is small, and the LDS words of one wave's region
// (masking levels + Bark levels first, the lanes' candidate lists later)
#define ULCX_SEL_CAP 8
#ifndef ULCX_SEL_CAND
#define ULCX_SEL_CAND 128
#endif
__host__ __device__ static inline int ulcx_sel_lds_words(int BS) { int a = BS / 2 + 4 * ULCX_NBARK, b = ULCX_SEL_CAP * 64; return a > b ? a : b; }
void ulcx_enc_finalize_keys(const UlcxEncCtx &c, hipStream_t st);
void ulcx_enc_materialise_noise(const UlcxEncCtx &c, hipStream_t st);     // parity tap: the {w, w*log} pairs of the last call into c.npair
