// ulcx_api.cpp — C ABI of the batched layer (include/ulc_amd.h §2): object lifetime,
// HBM layout, host-side constants, launches.  Host code only; every kernel lives in
// ulcx_enc.hip / ulcx_dec.hip.  There is no CPU fallback: if HIP cannot give us a
// device, every entry point returns ULCX_ERR_NO_DEVICE.
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include "ulcx_internal.h"


#define CKR(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { ulcx_set_error("%s: %s", #x, hipGetErrorString(e_)); return ULCX_ERR_HIP; } } while (0)

struct ulcx_encoder {
    int device, B, C, BS, rate, maxK;
    UlcxEncCtx ctx;
    void *tables;
    std::vector<void *> allocs;
    hipEvent_t ev[ULCX_ENC_STAGES + 1];
    bool evOk, evRecorded;
    int lastK;
    hipStream_t side; hipEvent_t evFork, evJoin, evFork2; bool sideOk; bool timing;
    hipEvent_t evWC[7 + 3 * ULCX_WC_MAXCH + ULCX_XF_MAXCH + 1]; int wcPipe; hipStream_t side2, side3; hipEvent_t evXf[2 * ULCX_XF_MAXCH]; int nXf;      // window-control / transform pipeline (ULCX_WC_PIPE chunks, default 4)
    bool keysFinal;
    int wcSteps, wcFuse;      // environment switches, read once at create
    int nsSlots;              // resident workgroups of k_nsums (its persistent grid)
    // staging for the host-pointer API
    float *d_pcm; uint8_t *d_out; int32_t *d_bits, *d_wc; float *d_cplx;
    // single-block path (ulcx_encode_block1): own stream, pinned staging, the captured launch sequence
    struct Block1Meta { int32_t bits, wc; float cplx; int32_t pad; UlcxWcState wcs; };
    hipStream_t b1Stream; hipGraph_t b1Graph; hipGraphExec_t b1Exec; bool b1Init, b1Graphed, b1NoGraph;
    int b1Mode; float b1P0, b1P1; int b1Rekeys;
    float *pinIn; uint8_t *pinOut; Block1Meta *pinMeta;
};
struct ulcx_decoder {
    int device, B, C, BS, maxK;
    UlcxDecCtx ctx;
    void *tables;
    std::vector<void *> allocs;
    hipEvent_t ev[ULCX_DEC_STAGES + 1];
    bool evOk, evRecorded, timing;
    uint8_t *d_in; size_t d_in_bytes; float *d_pcm; int32_t *d_bits;
    uint8_t *d_pay; int32_t *d_payBytes; long long payStride;     // resident packed payloads (ulcx_decoder_upload_payload)
    // k_dsyn over an even cut of the call's (stream, block) pairs: the second set of state arrays, the resident
    // workgroups of the kernel on this device, ULCX_DSYN_SPLIT=0 switches it off
    float *lap2; int *lastSub2; uint32_t *seed2; int *dead2; int synSlots, scratchRows, lastGrid, lastFull; bool splitOK, tailCut;
    // single-block path (ulcx_decode_block1)
    hipStream_t b1Stream; hipGraph_t b1Graph; hipGraphExec_t b1Exec; bool b1Init, b1Graphed, b1NoGraph; int b1Slot;
    uint8_t *pinIn; float *pinPcm; int32_t *pinMeta;
    uint32_t b1Seed;                                              // the stream's RNG state between single-block calls
    // The device word (ctx.seed) is the authoritative state of the object's own noise chain; b1Seed is its host copy, which the
    // single-block path uploads in front of every block.  Any OTHER decode call on this object advances the device word only:
    // it marks the copy stale, and the next single-block call without a caller-owned state reads the device word back first.
    bool b1SeedStale, inBlock1;
};

#ifndef ULCX_SRC_REV
#define ULCX_SRC_REV "unknown"
#endif
extern "C" const char *ulcx_build_rev(void) { return ULCX_SRC_REV; }
extern "C" int ulcx_device_count(void) {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) { ulcx_set_error("hipGetDeviceCount: %s", hipGetErrorString(e)); return 0; }
    return n;
}

static int validate(int C, int BS) {                 // ulcEncoder.c:32-34 / ulcDecoder.c:33-35
    if (C < 1 || C > 255) return 0;
    if (BS < 256 || BS > 32768) return 0;
    if ((BS & (-BS)) != BS) return 0;
    return 1;
}
static int ilog2i(int x) { int r = 0; while ((1 << r) < x) r++; return r; }

template <typename T>
static int dalloc(std::vector<void *> &v, T **p, size_t count, bool zero) {
    void *q = nullptr;
    size_t bytes = count * sizeof(T);
    if (bytes == 0) bytes = 16;
    hipError_t e = hipMalloc(&q, bytes);
    if (e != hipSuccess) { ulcx_set_error("hipMalloc(%zu bytes): %s", bytes, hipGetErrorString(e)); return ULCX_ERR_NOMEM; }
    if (zero) { e = hipMemset(q, 0, bytes); if (e != hipSuccess) { ulcx_set_error("hipMemset: %s", hipGetErrorString(e)); hipFree(q); return ULCX_ERR_HIP; } }
    v.push_back(q);
    *p = (T *)q;
    return ULCX_OK;
}
#define DA(ptr, count, zero) do { int rc_ = dalloc(e->allocs, &(ptr), (size_t)(count), zero); if (rc_) { cleanup(e); return rc_; } } while (0)

static int select_device(int device) {
    int n = ulcx_device_count();
    if (n <= 0) { if (!ulcx_last_error()[0]) ulcx_set_error("no HIP device visible"); return ULCX_ERR_NO_DEVICE; }
    if (device < 0 || device >= n) { ulcx_set_error("device %d out of range (have %d)", device, n); return ULCX_ERR_ARG; }
    hipError_t e = hipSetDevice(device);
    if (e != hipSuccess) { ulcx_set_error("hipSetDevice: %s", hipGetErrorString(e)); return ULCX_ERR_NO_DEVICE; }
    return ULCX_OK;
}

// ---------------------------------------------------------------------------
// encoder
// ---------------------------------------------------------------------------
static void cleanup(ulcx_encoder *e) {
    if (!e) return;
    for (void *p : e->allocs) hipFree(p);
    if (e->tables) hipFree(e->tables);
    if (e->evOk) for (auto &v : e->ev) hipEventDestroy(v);
    if (e->b1Graphed) { hipGraphExecDestroy(e->b1Exec); hipGraphDestroy(e->b1Graph); }
    if (e->b1Stream) hipStreamDestroy(e->b1Stream);
    if (e->pinIn) hipHostFree(e->pinIn);
    if (e->pinOut) hipHostFree(e->pinOut);
    if (e->pinMeta) hipHostFree(e->pinMeta);
    if (e->sideOk) { hipStreamDestroy(e->side); hipEventDestroy(e->evFork); hipEventDestroy(e->evJoin); hipEventDestroy(e->evFork2); for (auto &v : e->evWC) hipEventDestroy(v); for (auto &v : e->evXf) hipEventDestroy(v); hipStreamDestroy(e->side2); hipStreamDestroy(e->side3); }
    delete e;
}

static int enc_reset_state(ulcx_encoder *e) {
    UlcxEncCtx &c = e->ctx;
    CKR(hipMemset(c.hist, 0, sizeof(float) * (size_t)e->B * 2 * e->BS * e->C));
    std::vector<UlcxWcState> w((size_t)e->B);
    for (auto &x : w) {
        memset(&x, 0, sizeof(x));
        x.wcPrev = 0x10;              // virtual block -1: full-size, full-overlap
        x.wcCur  = 0x10;              // State->NextWindowCtrl = 0x10 (ulcEncoder.c:70)
    }
    CKR(hipMemcpy(c.wcs, w.data(), sizeof(UlcxWcState) * w.size(), hipMemcpyHostToDevice));
    return ULCX_OK;
}

extern "C" int ulcx_encoder_create(ulcx_encoder **out, int device, int nStreams, int nChan, int BlockSize, int RateHz, int maxBlocksPerCall) {
    if (!out) return ULCX_ERR_ARG;
    *out = nullptr;
    if (!validate(nChan, BlockSize) || nStreams < 1 || maxBlocksPerCall < 1 || RateHz < 1) {
        ulcx_set_error("invalid encoder geometry (nStreams=%d nChan=%d BlockSize=%d RateHz=%d maxBlocks=%d)", nStreams, nChan, BlockSize, RateHz, maxBlocksPerCall);
        return ULCX_ERR_ARG;
    }
    if (BlockSize > ULCX_MAX_BS_DEVICE) { ulcx_set_error("BlockSize %d > %d not built for the device yet", BlockSize, ULCX_MAX_BS_DEVICE); return ULCX_ERR_UNSUPPORTED; }
    int rc = select_device(device);
    if (rc) return rc;
    ulcx_encoder *e = new ulcx_encoder();
    e->device = device; e->B = nStreams; e->C = nChan; e->BS = BlockSize; e->rate = RateHz; e->maxK = maxBlocksPerCall;
    e->tables = nullptr; e->evOk = false; e->evRecorded = false; e->timing = true; e->lastK = 0; e->sideOk = false; e->side = nullptr; e->keysFinal = false;
    e->d_pcm = nullptr; e->d_out = nullptr; e->d_bits = nullptr; e->d_wc = nullptr; e->d_cplx = nullptr;
    e->b1Init = e->b1Graphed = e->b1NoGraph = false; e->b1Stream = nullptr; e->b1Rekeys = 0; e->pinIn = nullptr; e->pinOut = nullptr; e->pinMeta = nullptr;
    UlcxEncCtx &c = e->ctx;
    memset(&c, 0, sizeof(c));
    c.B = nStreams; c.C = nChan; c.BS = BlockSize; c.lgBS = ilog2i(BlockSize); c.maxK = maxBlocksPerCall; c.K = 0;
    c.rateHz = RateHz;
    c.slot = 2 * nChan * BlockSize + 16;          // >= worst case 4 nybbles/coefficient + header
    c.unitCap = 2 * BlockSize + 32;
    // data-independent libm calls of the reference, evaluated on the host like the reference does
    c.cHP  = 1.0f - expf(-0x1.CC845Cp6f / RateHz);            // WindowControl.c:75,82
    c.cBP  = 1.0f - expf(-0x1.596344p8f / RateHz);            // :76,83
    c.qHP  = 1.0f - expf(-0x1.CC845Cp7f / RateHz);            // :94,101
    c.qBP  = 1.0f - expf(-0x1.596344p8f / RateHz);            // :95,102
    c.cBlk = 1.0f - expf(-0x1.1AF110p-6f * BlockSize / RateHz); // :120,126
    c.cplxScale = 0x1.62E430p-1f * (31 - __builtin_clz((unsigned)BlockSize));   // BlockTransform.c:320
    rc = ulcx_tables_build(&c.T, &e->tables, BlockSize, RateHz, true);
    if (rc) { cleanup(e); return rc; }
    {   // k_bark_uniform keeps one snapshot of the running sums per Bark band that is open (lower edge passed, upper edge
        // not yet): the most the full-size band tables ever have open at once sizes its ring
        int most = 0;
        const int N = BlockSize / 2;
        for (int t = 0; t < 2; t++) {
            const short *beg = t ? c.T.pBeg[0] : c.T.nBeg[0], *end = t ? c.T.pEnd[0] : c.T.nEnd[0];
            for (int b = 0; b < ULCX_NBARK; b++) {
                if (beg[b] > end[b] || end[b] > N) most = 1 << 20;
                if (b && (beg[b] < beg[b - 1] || end[b] < end[b - 1])) most = 1 << 20;
            }
            for (int pos = 0; pos <= N; pos++) {
                int open = 0;
                for (int b = 0; b < ULCX_NBARK; b++) open += (beg[b] <= pos && pos <= end[b]) ? 1 : 0;
                if (open > most) most = open;
            }
        }
        int ring = 4;
        while (ring < most) ring *= 2;
        c.barkRing = (ring <= 8 && N % 32 == 0) ? ring : 0;               // (0: k_nbark / k_pbark for every block - geometries whose band edges need a deeper ring)
        // (round 3: also for a few blocks per call - the drop-in's one: the four-wave kernel walks a row's 1024 lines in a
        //  quarter of the time the lane-per-subblock kernels take, which is what a single stream waits for)
    }
    size_t B = nStreams, K = maxBlocksPerCall, NB = B * K, cb = (size_t)nChan * BlockSize;
    DA(c.hist, B * 2 * BlockSize * nChan, true);
    DA(c.wcs, B, true);
    DA(c.env, ((B + 63) / 64) * 64 * K * BlockSize, true);
    DA(c.bins, B * (K + 1) * 16, true);
    DA(c.wcArr, B * (K + 2), true);
    DA(c.coef, NB * cb, false);
    DA(c.key, NB * cb, false);
    DA(c.nsum, ((NB * nChan + 63) / 64) * 64 * (size_t)(BlockSize / 2), false);      // rows in tiles of 64 (tile_idx)
    DA(c.amp2, ((NB + 63) / 64) * 64 * (size_t)(BlockSize / 2), false);
    DA(c.barkN, NB * nChan * 4 * ULCX_NBARK, true);
    DA(c.barkP, NB * 4 * ULCX_NBARK, true);
    if (c.barkRing) { DA(c.barkRawN, NB * nChan * ULCX_NBARK * 3, false); DA(c.barkRawP, NB * ULCX_NBARK * 3, false); DA(c.decList, NB, false); DA(c.decCount, 1, true); }
    DA(c.nnz, NB, true);
    DA(c.cplx, NB, true);
    DA(c.nout, NB, true);
    DA(c.slow, 3 * NB + 2, true);                      // flags [NB], retry-queue counters [2], retry queues [2][NB]
    c.useWave = 1;        // wave-per-unit encode pass fed by k_nsums / k_tails; the serial lane-per-unit kernel takes what its capacities cannot hold
    c.useGapSums = (cb <= 16384 && nChan <= 16) ? 1 : 0;          // (k_nsums: a block's pairs in LDS, two bits per channel in a word)
    DA(c.gapSum, NB * cb, false);
    DA(c.tailSum, NB * nChan * 4 * 8, true);
    c.directPack = 1;
    if (const char *ev = getenv("ULCX_DIRECT_PACK")) c.directPack = (ev[0] != '0');
    c.dbgSkip = 0;
#ifdef ULCX_ABLATE
    if (const char *ev = getenv("ULCX_DBG_SKIP")) c.dbgSkip = atoi(ev);   // timing experiments only (breaks results)
#endif
    DA(c.cbrLo, NB, true); DA(c.cbrHi, NB, true); DA(c.cbrDone, NB, true); DA(c.cbrBudget, NB, true); DA(c.cbrLive, 1, true);
    DA(c.selWin, NB, true); DA(c.selT, NB, true); c.selPass = 0;
    { const char *ev = getenv("ULCX_SEL_PAIR"); c.selPair = ev ? atoi(ev) : 1; }
    DA(c.keep, NB * cb / 32, true);
    DA(c.fbList, NB, true);
    DA(c.fbCount, 4, true);
    DA(c.unitBuf, NB * nChan * (size_t)c.unitCap, true);
    DA(c.unitNyb, NB * nChan * 4, true);
    {
        size_t heapBytes = (cb * 8 > ULCX_HEAP_LDS_BYTES) ? (size_t)ULCX_HEAP_GRID * cb * 8 : 16;
        uint8_t *hp = nullptr;
        DA(hp, heapBytes, false);
        c.heapScratch = hp;
    }
    for (auto &v : e->ev) { if (hipEventCreate(&v) != hipSuccess) { ulcx_set_error("hipEventCreate failed"); cleanup(e); return ULCX_ERR_HIP; } }
    e->evOk = true;
    {
        const char *evs = getenv("ULCX_ASYNC_FB");
        if (!(evs && evs[0] == '0')) {
            if (hipStreamCreateWithFlags(&e->side, hipStreamNonBlocking) == hipSuccess &&
                hipEventCreateWithFlags(&e->evFork, hipEventDisableTiming) == hipSuccess &&
                hipEventCreateWithFlags(&e->evJoin, hipEventDisableTiming) == hipSuccess &&
                hipEventCreateWithFlags(&e->evFork2, hipEventDisableTiming) == hipSuccess) {
                bool ok = hipStreamCreateWithFlags(&e->side2, hipStreamNonBlocking) == hipSuccess && hipStreamCreateWithFlags(&e->side3, hipStreamNonBlocking) == hipSuccess;
                for (auto &v : e->evWC) ok = ok && hipEventCreateWithFlags(&v, hipEventDisableTiming) == hipSuccess;
                for (auto &v : e->evXf) ok = ok && hipEventCreate(&v) == hipSuccess;
                e->sideOk = ok;
            }
        }
        e->wcPipe = e->sideOk ? 4 : 1;                         // transform chunks per call: 1 block, then thirds (4 vs 5 chunks: 9.50 vs 9.56 ms per bench step)
        if (const char *pv = getenv("ULCX_WC_PIPE")) { int n = atoi(pv); if (n >= 1 && n <= ULCX_XF_MAXCH && n != 2 && (n == 1 || e->sideOk)) e->wcPipe = n; }
        e->wcSteps = -1; if (const char *sv = getenv("ULCX_WC_STEPS")) e->wcSteps = atoi(sv);      // -1: default; 0: the transform's chunks
        e->wcFuse = 1;                                           // (stereo: k_wc_ef; every other channel count: k_wc_energy + k_wc_forward)
    }
    e->nsSlots = c.useGapSums ? ulcx_enc_nsums_slots(BlockSize, nChan) : 0;
    if (e->nsSlots <= 0) e->nsSlots = 1024;
    DA(c.isFb, NB, true);
    DA(c.ownSlot, NB, true);
    {
        // Rank slots of the exact path (one full ranking per tie-straddle block).  One slot per block while that stays
        // within 8 GiB (a rate-search call over 524 288 blocks with 1 GiB of slots enqueued the exact path's fifteen passes
        // for eight groups of slots, seven of them always empty) AND within a quarter of what the device has free right
        // now: several encoders may share one GPU (`ulcx-tool -devices:N` above the visible count, two host threads, eight
        // test ranks).  ULCX_RANK_SLOTS=n overrides (tests).  If the allocation fails all the same, halve down to 64 slots
        // rather than fail the create: the launch walks groups of slots whatever their number (ulcx_enc_launch).
        size_t slots = NB;
        size_t maxSlots = ((size_t)8 << 30) / (cb * 4);
        size_t freeB = 0, totalB = 0;
        if (hipMemGetInfo(&freeB, &totalB) == hipSuccess && freeB / 4 / (cb * 4) < maxSlots) maxSlots = freeB / 4 / (cb * 4);
        if (maxSlots < 64) maxSlots = 64;
        if (slots > maxSlots) slots = maxSlots;
        if (const char *ev = getenv("ULCX_RANK_SLOTS")) { long n = atol(ev); if (n >= 1 && (size_t)n < slots) slots = (size_t)n; }
        for (;;) {
            int rc_ = dalloc(e->allocs, &c.rankBuf, slots * cb, false);
            if (rc_ == ULCX_OK) break;
            if (rc_ != ULCX_ERR_NOMEM || slots <= 64) { cleanup(e); return rc_; }
            (void)hipGetLastError();
            slots = (slots + 1) / 2; if (slots < 64) slots = 64;
        }
        c.rankSlots = (int)slots;
    }
    rc = enc_reset_state(e);
    if (rc) { cleanup(e); return rc; }
    *out = e;
    return ULCX_OK;
}

extern "C" void ulcx_encoder_destroy(ulcx_encoder *e) { if (e) { hipSetDevice(e->device); cleanup(e); } }
extern "C" int ulcx_encoder_reset(ulcx_encoder *e) { if (!e) return ULCX_ERR_ARG; CKR(hipSetDevice(e->device)); return enc_reset_state(e); }
extern "C" int ulcx_encoder_slot_bytes(const ulcx_encoder *e) { return e ? e->ctx.slot : 0; }

static int encode_dev_any(ulcx_encoder *e, int mode, float p0, float p1, const float *d_pcm, const int16_t *d_pcm16, int nBlocks,
                          uint8_t *d_out, int32_t *d_bits, int32_t *d_wc, float *d_cplx, void *hipStream) {
    if (!e || (!d_pcm && !d_pcm16) || !d_out || !d_bits || nBlocks < 1 || nBlocks > e->maxK) { ulcx_set_error("ulcx_encode_dev: bad argument"); return ULCX_ERR_ARG; }
    if (mode != ULCX_MODE_VBR && mode != ULCX_MODE_CBR && mode != ULCX_MODE_ABR) { ulcx_set_error("bad mode"); return ULCX_ERR_ARG; }
    CKR(hipSetDevice(e->device));
    UlcxEncCtx c = e->ctx;
    c.K = nBlocks; c.keyFinal = 0; c.mode = mode; c.p0 = p0; c.p1 = p1;
    c.vbrTarget = (mode == ULCX_MODE_VBR) ? 0x1.E4EFB7p3f * logf(100.0f / p0) : 0.0f;     // ulcEncoder.c:144 (host libm, data independent)
    c.pcm = d_pcm; c.pcm16 = d_pcm16; c.out = d_out; c.bits = d_bits; c.wcOut = d_wc; c.cplxOut = d_cplx;
    UlcxEncAux aux;
    aux.side = e->sideOk ? e->side : nullptr; aux.side2 = e->sideOk ? e->side2 : nullptr; aux.side3 = e->sideOk ? e->side3 : nullptr;
    aux.evFork = e->evFork; aux.evJoin = e->evJoin; aux.evFork2 = e->evFork2; aux.evWC = e->evWC; aux.evXf = e->evXf;
    aux.wcPipe = (nBlocks >= 2 * e->wcPipe) ? e->wcPipe : (nBlocks >= 6 && e->wcPipe > 1 ? 3 : 1); aux.nXf = &e->nXf;
    aux.wcSteps = e->wcSteps; aux.wcFuse = e->wcFuse; aux.nsSlots = e->nsSlots;
    const int rc = ulcx_enc_launch(c, (hipStream_t)hipStream, e->timing ? e->ev : nullptr, aux);
    e->evRecorded = (rc == ULCX_OK) && e->timing;
    e->lastK = nBlocks;
    e->keysFinal = false;
    return rc;
}

extern "C" int ulcx_encode_dev(ulcx_encoder *e, int mode, float p0, float p1, const float *d_pcm, int nBlocks,
                               uint8_t *d_out, int32_t *d_bits, int32_t *d_wc, float *d_cplx, void *hipStream) {
    if (!d_pcm) { ulcx_set_error("ulcx_encode_dev: bad argument"); return ULCX_ERR_ARG; }
    return encode_dev_any(e, mode, p0, p1, d_pcm, nullptr, nBlocks, d_out, d_bits, d_wc, d_cplx, hipStream);
}
extern "C" int ulcx_encode_dev_pcm16(ulcx_encoder *e, int mode, float p0, float p1, const int16_t *d_pcm16, int nBlocks,
                                     uint8_t *d_out, int32_t *d_bits, int32_t *d_wc, float *d_cplx, void *hipStream) {
    if (!d_pcm16) { ulcx_set_error("ulcx_encode_dev_pcm16: bad argument"); return ULCX_ERR_ARG; }
    return encode_dev_any(e, mode, p0, p1, nullptr, d_pcm16, nBlocks, d_out, d_bits, d_wc, d_cplx, hipStream);
}

extern "C" int ulcx_encode_host(ulcx_encoder *e, int mode, float p0, float p1, const float *h_pcm, int nBlocks,
                                uint8_t *h_out, int32_t *h_bits, int32_t *h_wc, float *h_cplx) {
    if (!e || !h_pcm || !h_out || !h_bits) return ULCX_ERR_ARG;
    CKR(hipSetDevice(e->device));
    size_t NBmax = (size_t)e->B * e->maxK, cb = (size_t)e->C * e->BS;
    if (!e->d_pcm) {
        int rc;
        if ((rc = dalloc(e->allocs, &e->d_pcm, NBmax * cb, false))) return rc;
        if ((rc = dalloc(e->allocs, &e->d_out, NBmax * e->ctx.slot, false))) return rc;
        if ((rc = dalloc(e->allocs, &e->d_bits, NBmax, false))) return rc;
        if ((rc = dalloc(e->allocs, &e->d_wc, NBmax, false))) return rc;
        if ((rc = dalloc(e->allocs, &e->d_cplx, NBmax, false))) return rc;
    }
    if (nBlocks < 1 || nBlocks > e->maxK) { ulcx_set_error("nBlocks out of range"); return ULCX_ERR_ARG; }
    size_t NB = (size_t)e->B * nBlocks;
    CKR(hipMemcpy(e->d_pcm, h_pcm, sizeof(float) * NB * cb, hipMemcpyHostToDevice));
    int rc = ulcx_encode_dev(e, mode, p0, p1, e->d_pcm, nBlocks, e->d_out, e->d_bits, e->d_wc, e->d_cplx, nullptr);
    if (rc) return rc;
    CKR(hipDeviceSynchronize());
    CKR(hipMemcpy(h_out, e->d_out, NB * e->ctx.slot, hipMemcpyDeviceToHost));
    CKR(hipMemcpy(h_bits, e->d_bits, sizeof(int32_t) * NB, hipMemcpyDeviceToHost));
    if (h_wc) CKR(hipMemcpy(h_wc, e->d_wc, sizeof(int32_t) * NB, hipMemcpyDeviceToHost));
    if (h_cplx) CKR(hipMemcpy(h_cplx, e->d_cplx, sizeof(float) * NB, hipMemcpyDeviceToHost));
    return ULCX_OK;
}

// One block of one stream per call (the drop-in ABI): include/ulc_amd.h.  The launch sequence of ulcx_encode_dev - side
// streams and their event fork/joins included - is captured once into a graph together with the copies between the pinned
// staging buffers and the device; a call is then memcpy, one graph launch, one synchronisation, memcpy.
extern "C" int ulcx_encode_block1(ulcx_encoder *e, int mode, float p0, float p1, const float *h_pcm,
                                  uint8_t *h_out, int32_t *bits, float *cplx, int32_t stateOut[2], float transientFilter[3]) {
    if (!e || !h_pcm || !h_out || e->B != 1 || e->maxK != 1) { ulcx_set_error("ulcx_encode_block1: needs an encoder of one stream, one block per call"); return ULCX_ERR_ARG; }
    CKR(hipSetDevice(e->device));
    const size_t cb = (size_t)e->C * e->BS, slot = (size_t)e->ctx.slot;
    if (!e->b1Init) {
        int rc;
        if (!e->d_pcm) {
            if ((rc = dalloc(e->allocs, &e->d_pcm, cb, false))) return rc;
            if ((rc = dalloc(e->allocs, &e->d_out, slot, false))) return rc;
            if ((rc = dalloc(e->allocs, &e->d_bits, 1, false))) return rc;
            if ((rc = dalloc(e->allocs, &e->d_wc, 1, false))) return rc;
            if ((rc = dalloc(e->allocs, &e->d_cplx, 1, false))) return rc;
        }
        // (b1Init only once everything exists: a failed allocation leaves the call to be retried from scratch, never a
        //  later call copying into a null staging buffer)
        if (!e->b1Stream) CKR(hipStreamCreateWithFlags(&e->b1Stream, hipStreamNonBlocking));
        if (!e->pinIn) CKR(hipHostMalloc((void **)&e->pinIn, sizeof(float) * cb, hipHostMallocDefault));
        if (!e->pinOut) CKR(hipHostMalloc((void **)&e->pinOut, slot, hipHostMallocDefault));
        if (!e->pinMeta) CKR(hipHostMalloc((void **)&e->pinMeta, sizeof(*e->pinMeta), hipHostMallocDefault));
        e->b1Init = true;
        e->timing = false;                                 // (per-kernel events cannot be captured, and nobody reads them here)
    }
    auto enqueue = [&]() -> int {
        CKR(hipMemcpyAsync(e->d_pcm, e->pinIn, sizeof(float) * cb, hipMemcpyHostToDevice, e->b1Stream));
        int rc = ulcx_encode_dev(e, mode, p0, p1, e->d_pcm, 1, e->d_out, e->d_bits, e->d_wc, e->d_cplx, e->b1Stream);
        if (rc) return rc;
        CKR(hipMemcpyAsync(e->pinOut, e->d_out, slot, hipMemcpyDeviceToHost, e->b1Stream));
        CKR(hipMemcpyAsync(&e->pinMeta->bits, e->d_bits, sizeof(int32_t), hipMemcpyDeviceToHost, e->b1Stream));
        CKR(hipMemcpyAsync(&e->pinMeta->wc, e->d_wc, sizeof(int32_t), hipMemcpyDeviceToHost, e->b1Stream));
        CKR(hipMemcpyAsync(&e->pinMeta->cplx, e->d_cplx, sizeof(float), hipMemcpyDeviceToHost, e->b1Stream));
        CKR(hipMemcpyAsync(&e->pinMeta->wcs, e->ctx.wcs, sizeof(UlcxWcState), hipMemcpyDeviceToHost, e->b1Stream));
        return ULCX_OK;
    };
    memcpy(e->pinIn, h_pcm, sizeof(float) * cb);
    if (e->b1Graphed && (e->b1Mode != mode || e->b1P0 != p0 || e->b1P1 != p1)) {          // the parameters are part of the captured kernels' arguments
        hipGraphExecDestroy(e->b1Exec); hipGraphDestroy(e->b1Graph); e->b1Graphed = false;
        // a caller whose parameters change from block to block (ULC_EncodeBlock_ABR: the reference's tool updates
        // AvgComplexity every block) would pay a capture + instantiate per call: after the second change, direct launches
        if (++e->b1Rekeys >= 2) e->b1NoGraph = true;
    }
    if (!e->b1Graphed && !e->b1NoGraph) {
        bool ok = hipStreamBeginCapture(e->b1Stream, hipStreamCaptureModeThreadLocal) == hipSuccess;
        if (ok) {
            const int rc = enqueue();
            hipGraph_t g = nullptr;
            const hipError_t ee = hipStreamEndCapture(e->b1Stream, &g);
            ok = (rc == ULCX_OK) && ee == hipSuccess && g != nullptr;
            if (ok) ok = hipGraphInstantiate(&e->b1Exec, g, nullptr, nullptr, 0) == hipSuccess;
            if (ok) { e->b1Graph = g; e->b1Graphed = true; e->b1Mode = mode; e->b1P0 = p0; e->b1P1 = p1; }
            else if (g) hipGraphDestroy(g);
        }
        if (!ok) { (void)hipGetLastError(); e->b1NoGraph = true; }                         // direct launches from here on
    }
    if (e->b1Graphed) CKR(hipGraphLaunch(e->b1Exec, e->b1Stream));
    else { int rc = enqueue(); if (rc) return rc; }
    CKR(hipStreamSynchronize(e->b1Stream));
    const int32_t nb = e->pinMeta->bits;
    memcpy(h_out, e->pinOut, (nb > 0 && (size_t)(nb + 7) / 8 <= slot) ? (size_t)(nb + 7) / 8 : slot);
    if (bits) *bits = nb;
    if (cplx) *cplx = e->pinMeta->cplx;
    if (stateOut) { stateOut[0] = e->pinMeta->wcs.wcPrev; stateOut[1] = e->pinMeta->wcs.wcCur; }     // WindowCtrl of this block, NextWindowCtrl
    if (transientFilter) for (int i = 0; i < 3; i++) transientFilter[i] = e->pinMeta->wcs.tf[i];
    return ULCX_OK;
}

extern "C" int ulcx_encoder_debug_fetch(ulcx_encoder *e, int nBlocks, float *h_coef, float *h_noise, float *h_keys, uint8_t *h_keep, int32_t *h_nout) {
    if (!e || nBlocks < 1 || nBlocks > e->maxK) return ULCX_ERR_ARG;
    CKR(hipSetDevice(e->device));
    CKR(hipDeviceSynchronize());
    size_t NB = (size_t)e->B * nBlocks, cb = (size_t)e->C * e->BS;
    if (h_coef)  CKR(hipMemcpy(h_coef, e->ctx.coef, sizeof(float) * NB * cb, hipMemcpyDeviceToHost));
    if (h_noise) {
        // the {w, w*log} pairs (the reference's TransformNoise) are not an array of the pipeline any more: formed here, for the tap
        if (!e->ctx.npair) { int rc = dalloc(e->allocs, &e->ctx.npair, (size_t)e->B * e->maxK * cb, false); if (rc) return rc; }
        UlcxEncCtx c2 = e->ctx; c2.K = nBlocks;
        ulcx_enc_materialise_noise(c2, nullptr);
        CKR(hipDeviceSynchronize());
        CKR(hipMemcpy(h_noise, e->ctx.npair, sizeof(float) * NB * cb, hipMemcpyDeviceToHost));
    }
    if (h_keys) {
        if (!e->keysFinal) {                       // the pipeline never writes final keys back; materialise them for the tap
            UlcxEncCtx c2 = e->ctx; c2.K = nBlocks; c2.keyFinal = 0;
            ulcx_enc_finalize_keys(c2, nullptr);
            CKR(hipDeviceSynchronize());
            e->keysFinal = true;
        }
        CKR(hipMemcpy(h_keys, e->ctx.key, sizeof(float) * NB * cb, hipMemcpyDeviceToHost));
    }
    if (h_nout)  CKR(hipMemcpy(h_nout, e->ctx.nout, sizeof(int32_t) * NB, hipMemcpyDeviceToHost));
    if (h_keep) {
        std::vector<uint32_t> bits(NB * cb / 32);
        CKR(hipMemcpy(bits.data(), e->ctx.keep, sizeof(uint32_t) * bits.size(), hipMemcpyDeviceToHost));
        for (size_t i = 0; i < NB * cb; i++) h_keep[i] = (bits[i >> 5] >> (i & 31)) & 1;
    }
    return ULCX_OK;
}

extern "C" int ulcx_encoder_debug_force_exact(ulcx_encoder *e, int every) {
    if (!e || every < 0) return ULCX_ERR_ARG;
    e->ctx.forceFb = every;
    return ULCX_OK;
}
extern "C" int ulcx_encoder_last_fallbacks(ulcx_encoder *e) {
    if (!e) return ULCX_ERR_ARG;
    CKR(hipSetDevice(e->device));
    CKR(hipDeviceSynchronize());
    int n = 0;
    CKR(hipMemcpy(&n, e->ctx.fbCount, sizeof(int), hipMemcpyDeviceToHost));
    return n;
}
extern "C" const char *ulcx_encoder_stage_name(int i) { return (i >= 0 && i < ULCX_ENC_STAGES_REPORTED) ? ulcx_enc_stage_names[i] : ""; }
extern "C" int ulcx_encoder_stage_ms(ulcx_encoder *e, float *ms, int maxStages) {
    if (!e || !e->evRecorded) return 0;
    int n = 0;
    for (int i = 0; i < ULCX_ENC_STAGES && i < maxStages; i++) {
        float t = 0;
        if (hipEventElapsedTime(&t, e->ev[i], e->ev[i + 1]) != hipSuccess) break;
        ms[n++] = t;
    }
    // Pipelined window control: the k_xf interval spans start-up + waits + the transform chunk launches.
    // Report the launches themselves as k_xf (what a kernel trace shows) and the rest as "wc_pipeline_exposed".
    if (n == ULCX_ENC_STAGES && n < maxStages) {
        const int IX_XF = 5;
        float exposed = 0.0f;
        if (e->nXf > 0) {
            float sum = 0.0f; bool ok = true;
            for (int j = 0; j < e->nXf; j++) { float t = 0; if (hipEventElapsedTime(&t, e->evXf[2 * j], e->evXf[2 * j + 1]) != hipSuccess) { ok = false; break; } sum += t; }
            if (ok) { exposed = ms[IX_XF] - sum; ms[IX_XF] = sum; }
        }
        ms[n++] = exposed;
    }
    return n;
}

extern "C" int ulcx_encoder_last_xf_launches(ulcx_encoder *e) { return (e && e->evRecorded) ? (e->nXf > 0 ? e->nXf : 1) : 0; }

// ---------------------------------------------------------------------------
// decoder
// ---------------------------------------------------------------------------
static void cleanup(ulcx_decoder *e) {
    if (!e) return;
    for (void *p : e->allocs) hipFree(p);
    if (e->d_pay) hipFree(e->d_pay);
    if (e->d_payBytes) hipFree(e->d_payBytes);
    if (e->tables) hipFree(e->tables);
    if (e->evOk) for (auto &v : e->ev) hipEventDestroy(v);
    if (e->b1Graphed) { hipGraphExecDestroy(e->b1Exec); hipGraphDestroy(e->b1Graph); }
    if (e->b1Stream) hipStreamDestroy(e->b1Stream);
    if (e->pinIn) hipHostFree(e->pinIn);
    if (e->pinPcm) hipHostFree(e->pinPcm);
    if (e->pinMeta) hipHostFree(e->pinMeta);
    delete e;
}
static int dec_reset_state(ulcx_decoder *e) {
    UlcxDecCtx &c = e->ctx;
    CKR(hipMemset(c.lap, 0, sizeof(float) * (size_t)e->B * e->C * (e->BS / 2)));     // ulcDecoder.c:56
    CKR(hipMemset(c.lastSub, 0, sizeof(int) * (size_t)e->B));                          // ulcDecoder.c:52
    CKR(hipMemset(c.dead, 0, sizeof(int) * (size_t)e->B));
    CKR(hipMemset(c.packOff, 0, sizeof(int) * (size_t)e->B));
    std::vector<uint32_t> seed((size_t)e->B, 1234567u);                                // ulcDecoder.c:76, one RNG per stream
    CKR(hipMemcpy(c.seed, seed.data(), sizeof(uint32_t) * seed.size(), hipMemcpyHostToDevice));
    e->b1Seed = 1234567u; e->b1SeedStale = false; e->inBlock1 = false;
    return ULCX_OK;
}

// Tables of the noise RNG (ulcDecoder.c:75-81).  xorshift32 is linear over GF(2): a matrix is kept as its 32 columns.
static uint32_t gf2_matvec(const uint32_t *col, uint32_t v) { uint32_t r = 0; for (int b = 0; b < 32; b++) if (v >> b & 1) r ^= col[b]; return r; }
static void gf2_matmul(uint32_t *out, const uint32_t *A, const uint32_t *Bm) { uint32_t t[32]; for (int b = 0; b < 32; b++) t[b] = gf2_matvec(A, Bm[b]); memcpy(out, t, sizeof(t)); }
static void build_rng_tables(std::vector<uint32_t> &jumpT) {
    auto step = [](uint32_t s) { s ^= s << 13; s ^= s >> 17; s ^= s << 5; return s; };
    // jumpT[i][d][k][v]: byte k = v of the state, through T^(d * 16^i)
    jumpT.assign((size_t)8 * 16 * 4 * 256, 0u);
    uint32_t P[32];                                       // T^(16^i)
    for (int b = 0; b < 32; b++) P[b] = step(1u << b);
    for (int i = 0; i < 8; i++) {
        uint32_t Md[32];                                  // P^d
        for (int b = 0; b < 32; b++) Md[b] = 1u << b;
        for (int d = 1; d < 16; d++) {
            gf2_matmul(Md, P, Md);
            uint32_t *tab = jumpT.data() + ((size_t)(i * 16 + d) << 10);
            for (int k = 0; k < 4; k++)
                for (int v = 0; v < 256; v++) {
                    uint32_t r = 0;
                    for (int t = 0; t < 8; t++) if (v >> t & 1) r ^= Md[8 * k + t];
                    tab[k * 256 + v] = r;
                }
        }
        gf2_matmul(Md, P, Md);                            // P^16 = the next position's unit
        memcpy(P, Md, sizeof(P));
    }
    // parT[k][v][l] (round 5), appended: what lane l of the synthesis contributes to a unit's sign-parity stream, straight from
    // the unit's start state.  Bit i of the word = parity of the top bits of draws 32 l + 1 .. 32 l + i + 1 - linear in the
    // state, so it is (parity map) x T^(32 l), kept as four byte tables.  The start state is the same for all lanes of the wave:
    // with the LANE as the fastest index a look-up is one coalesced 256-byte row, not a gather.
    auto par_word = [&](uint32_t st) { uint32_t x = 0, par = 0; for (int i = 0; i < 32; i++) { st = step(st); par ^= st >> 31; x |= par << i; } return x; };
    uint32_t A[32], Ml[32];                               // A = T^32, Ml = A^l
    for (int b = 0; b < 32; b++) { uint32_t v = 1u << b; for (int i = 0; i < 32; i++) v = step(v); A[b] = v; Ml[b] = 1u << b; }
    const size_t base = jumpT.size();
    jumpT.resize(base + (size_t)4 * 256 * 64);
    for (int l = 0; l < 64; l++) {
        uint32_t col[32];
        for (int b = 0; b < 32; b++) col[b] = par_word(Ml[b]);
        for (int k = 0; k < 4; k++)
            for (int v = 0; v < 256; v++) {
                uint32_t r = 0;
                for (int t = 0; t < 8; t++) if (v >> t & 1) r ^= col[8 * k + t];
                jumpT[base + ((size_t)(k * 256 + v) << 6) + l] = r;
            }
        gf2_matmul(Ml, A, Ml);
    }
}

// One decode launch.  When the batch does not fill the machine in whole rounds of one workgroup per stream - 4096 streams on
// 1536 resident workgroups, or a few long streams - the synthesis takes an even cut of the (stream, block) pairs instead: a
// workgroup then runs one extra block (the one in front of its range, for the lapping state), so the cut must pay for that.
// The cut itself (host arithmetic, exported for the tests): workgroups of the synthesis for a call of nBlocks blocks of
// nStreams streams on a device that holds residentWG workgroups of the kernel; 0 = one workgroup per stream.  Cost in block
// times: ceil(streams / resident) rounds of nBlocks blocks against blocks-per-workgroup + 1 (the block in front of the range);
// the cut has to win by 1.5 x - an even cut of uneven streams ends with its slowest workgroup (measured on the bench batch).
extern "C" int ulcx_dec_split_plan(int nStreams, int nBlocks, int residentWG) {
    if (nStreams < 1 || nBlocks < 1 || residentWG < 1) return 0;
    const long long T = (long long)nStreams * nBlocks;
    long long per = (T + residentWG - 1) / residentWG; if (per < 8) per = 8;
    const long long grid = T / per;
    const long long costStream = (((long long)nStreams + residentWG - 1) / residentWG) * nBlocks;
    return (grid >= 1 && grid != nStreams && (per + 1) * 3 < costStream * 2) ? (int)grid : 0;
}
// Round 5: a batch runs in rounds of one workgroup per stream, and its last round is partly empty (4096 streams on 1536
// resident workgroups: 2.67 rounds).  The whole rounds stay as they are - the hardware hands a free slot the next stream,
// which evens out workgroups of different speed -; only the streams of the last round are cut, into pieces of
// ULCX_DSYN_TAIL_LEN blocks (a quarter of a longer call's) at the end of the grid, each of which runs one block in front of
// its range for the lapping state.  What the cut buys is a short end of the launch, what it costs is the extra block per
// piece.  Measured (profiles/NOTES_r05.md), synthesis of 32 blocks of 4096 / 2048 / 1024 / 5000 streams: 1.50 -> 1.44, 0.89
// -> 0.75, 0.485 -> 0.45, 1.81 -> 1.76 ms; ONE piece per slot (the even cut of the last round) 1.51, pieces of 4 blocks
// 1.47-1.49, every stream cut 1.53-1.59; calls of 16 blocks (two pieces per stream) -1 % / +1.5 %: not cut.
// Returns the number of pieces (0: no cut), *full = the leading workgroups that take one whole stream each.  The cut is
// taken when the last round is at most four fifths full and a stream has at least three pieces.
#define ULCX_DSYN_TAIL_LEN 8
extern "C" int ulcx_dec_tail_plan(int nStreams, int nBlocks, int residentWG, int *full) {
    if (full) *full = 0;
    if (nStreams < 1 || nBlocks < 3 * ULCX_DSYN_TAIL_LEN || residentWG < 1) return 0;
    const int rem = nStreams % residentWG;
    if (rem == 0 || (long long)rem * 5 > (long long)residentWG * 4) return 0;
    const int len = nBlocks / 4 > ULCX_DSYN_TAIL_LEN ? nBlocks / 4 : ULCX_DSYN_TAIL_LEN;     // (a long call: four pieces per stream)
    const long long n = (long long)rem * nBlocks / len;                                      // < 4 residentWG
    if (full) *full = nStreams - rem;
    return (int)n;
}
extern "C" int ulcx_decoder_last_cut(ulcx_decoder *e, int *workgroups, int *wholeStreams, int *residentWG) {
    if (!e) return ULCX_ERR_ARG;
    if (workgroups) *workgroups = e->lastGrid;
    if (wholeStreams) *wholeStreams = e->lastFull;
    if (residentWG) *residentWG = e->synSlots;
    return ULCX_OK;
}
static int dec_launch(ulcx_decoder *e, UlcxDecCtx &c, hipStream_t st) {
    if (!e->inBlock1) e->b1SeedStale = true;                      // (a batched / packed call on a one-stream decoder: see b1Seed)
    UlcxDecAux a;
    a.synGrid = 0; a.synFull = 0;
    c.lapO = c.lap; c.lastSubO = c.lastSub; c.seedO = c.seed; c.deadO = c.dead;
    if (e->splitOK && e->synSlots > 0) {
        a.synGrid = ulcx_dec_split_plan(e->B, c.K, e->synSlots);
        if (!a.synGrid) {
            int full = 0;
            const int tail = ulcx_dec_tail_plan(e->B, c.K, e->synSlots, &full);
            if (tail > 0 && tail <= e->scratchRows && e->tailCut) { a.synGrid = full + tail; a.synFull = full; }
        }
        if (a.synGrid) {
            if (getenv("ULCX_DEBUG_PRINT")) fprintf(stderr, "[ulcx] synthesis: %lld (stream, block) pairs over %d workgroups (%d of them one stream each; %d resident)\n", (long long)e->B * c.K, a.synGrid, a.synFull, e->synSlots);
            c.lapO = e->lap2; c.lastSubO = e->lastSub2; c.seedO = e->seed2; c.deadO = e->dead2;
        }
    }
    e->lastGrid = a.synGrid; e->lastFull = a.synFull;
    const int rc = ulcx_dec_launch(c, st, e->timing ? e->ev : nullptr, a);
    if (rc == ULCX_OK && a.synGrid) {
        std::swap(e->ctx.lap, e->lap2); std::swap(e->ctx.lastSub, e->lastSub2); std::swap(e->ctx.seed, e->seed2); std::swap(e->ctx.dead, e->dead2);
    }
    return rc;
}

extern "C" int ulcx_decoder_create(ulcx_decoder **out, int device, int nStreams, int nChan, int BlockSize, int maxBlocksPerCall) {
    if (!out) return ULCX_ERR_ARG;
    *out = nullptr;
    if (!validate(nChan, BlockSize) || nStreams < 1 || maxBlocksPerCall < 1) { ulcx_set_error("invalid decoder geometry"); return ULCX_ERR_ARG; }
    int rc = select_device(device);
    if (rc) return rc;
    ulcx_decoder *e = new ulcx_decoder();
    e->device = device; e->B = nStreams; e->C = nChan; e->BS = BlockSize; e->maxK = maxBlocksPerCall;
    e->tables = nullptr; e->evOk = false; e->evRecorded = false; e->timing = true;
    e->d_in = nullptr; e->d_in_bytes = 0; e->d_pcm = nullptr; e->d_bits = nullptr; e->d_pay = nullptr; e->d_payBytes = nullptr; e->payStride = 0;
    e->lap2 = nullptr; e->lastSub2 = nullptr; e->seed2 = nullptr; e->dead2 = nullptr; e->synSlots = 0; e->scratchRows = 0; e->lastGrid = 0; e->lastFull = 0; e->splitOK = false; e->tailCut = true;
    e->b1Init = e->b1Graphed = e->b1NoGraph = false; e->b1Stream = nullptr; e->pinIn = nullptr; e->pinPcm = nullptr; e->pinMeta = nullptr; e->b1Slot = 0;
    UlcxDecCtx &c = e->ctx;
    memset(&c, 0, sizeof(c));
    c.B = nStreams; c.C = nChan; c.BS = BlockSize; c.lgBS = ilog2i(BlockSize); c.maxK = maxBlocksPerCall;
#ifdef ULCX_ABLATE
    if (const char *ev = getenv("ULCX_DBG_SKIP")) c.dbgSkip = atoi(ev);
#endif
    // stereo streams up to BlockSize 4096 keep their lapping state, both channels' FFT arrays and the twiddles in LDS, one wave
    // per channel (k_dsyn); everything else takes the general kernel (k_dgen: one array, state in HBM)
    c.fastOK = (nChan == 2 && BlockSize <= 4096) ? 1 : 0;
    if (const char *ev = getenv("ULCX_DEC_FAST")) c.fastOK = c.fastOK && (ev[0] != '0');
    // stereo synthesis kernel: lapping state in global memory; BlockSize <= 2048: FFT twiddles in LDS (mode 2), above: from the tables
    c.twInLds = c.fastOK ? ((BlockSize <= 2048 && ULCX_DSYN_TWL) ? 2 : 0) : 0;
    rc = ulcx_tables_build(&c.T, &e->tables, BlockSize, 44100, false);
    if (rc) { cleanup(e); return rc; }
    size_t B = nStreams, NB = B * maxBlocksPerCall;
    DA(c.lap, B * nChan * (BlockSize / 2), true);
    DA(c.lastSub, B, true);
    DA(c.seed, B, true);
    DA(c.dead, B, true);
    c.lapO = c.lap; c.lastSubO = c.lastSub; c.seedO = c.seed; c.deadO = c.dead; c.lapScratch = nullptr; c.k0 = 0; c.k1 = 0;
    if (c.fastOK) {                                                       // the kernel keeps the lapping state in global memory: any grid
        bool want = true;
        if (const char *ev = getenv("ULCX_DSYN_SPLIT")) want = ev[0] != '0';
        if (const char *ev = getenv("ULCX_DSYN_TAIL")) e->tailCut = ev[0] != '0';          // (A/B: the last round uncut)
        e->synSlots = want ? ulcx_dec_syn_slots(c) : 0;
        if (e->synSlots > 0) {
            DA(e->lap2, B * nChan * (BlockSize / 2), true);
            DA(e->lastSub2, B, true);
            DA(e->seed2, B, true);
            DA(e->dead2, B, true);
            e->scratchRows = 4 * e->synSlots;                              // (an even cut: <= synSlots workgroups; a cut of the last round: < synSlots streams in <= 4 pieces each)
            DA(c.lapScratch, (size_t)e->scratchRows * nChan * (BlockSize / 2), true);
            e->splitOK = true;
        }
    }
    DA(c.wcScan, NB, true);
    DA(c.draws, NB, true);
    DA(c.packOff, B, true);
    DA(c.blkOff, NB, true);
    DA(c.unitDraws, NB * nChan * 4, true);
    DA(c.unitTail, NB * nChan * 4, true);
    DA(c.unitRec, NB * nChan * 4, true);
    // what the scan leaves for the synthesis: at most one plain-run record per coefficient, one noise record per 16 (+ a tail per unit)
    c.precStride = nChan * BlockSize;
    c.nrecStride = nChan * BlockSize / 16 + nChan * 4;
    DA(c.prec, NB * (size_t)c.precStride, false);
    DA(c.nrec, NB * (size_t)c.nrecStride, false);
    c.tailStride = BlockSize / 32;
    DA(c.tailMag, NB * nChan * 4 * (size_t)c.tailStride, false);
    DA(c.scratch, B * 4 * (size_t)BlockSize, false);
    {
        std::vector<uint32_t> jt;
        build_rng_tables(jt);
        uint32_t *dj = nullptr;
        DA(dj, jt.size(), false);
        if (hipMemcpy(dj, jt.data(), sizeof(uint32_t) * jt.size(), hipMemcpyHostToDevice) != hipSuccess) { ulcx_set_error("hipMemcpy(rng tables)"); cleanup(e); return ULCX_ERR_HIP; }
        c.jumpT = dj;
        c.parT = dj + (size_t)8 * 16 * 4 * 256;
    }
    for (auto &v : e->ev) { if (hipEventCreate(&v) != hipSuccess) { ulcx_set_error("hipEventCreate failed"); cleanup(e); return ULCX_ERR_HIP; } }
    e->evOk = true;
    rc = dec_reset_state(e);
    if (rc) { cleanup(e); return rc; }
    *out = e;
    return ULCX_OK;
}
extern "C" void ulcx_decoder_destroy(ulcx_decoder *e) { if (e) { hipSetDevice(e->device); cleanup(e); } }
extern "C" int ulcx_decoder_reset(ulcx_decoder *e) { if (!e) return ULCX_ERR_ARG; CKR(hipSetDevice(e->device)); return dec_reset_state(e); }

static int decode_dev_any(ulcx_decoder *e, const uint8_t *d_in, int slotBytes, int nBlocks, float *d_pcm, int16_t *d_pcm16, int32_t *d_bits, void *hipStream) {
    if (!e || !d_in || (!d_pcm && !d_pcm16) || !d_bits || slotBytes < 1 || nBlocks < 1 || nBlocks > e->maxK) { ulcx_set_error("ulcx_decode_dev: bad argument"); return ULCX_ERR_ARG; }
    CKR(hipSetDevice(e->device));
    UlcxDecCtx c = e->ctx;
    c.K = nBlocks; c.slot = slotBytes; c.in = d_in; c.pcm = d_pcm; c.pcm16 = d_pcm16; c.bits = d_bits;
    c.inBytes = (long long)e->B * nBlocks * slotBytes;
    int rc = dec_launch(e, c, (hipStream_t)hipStream);
    e->evRecorded = (rc == ULCX_OK) && e->timing;
    return rc;
}
extern "C" int ulcx_decode_dev(ulcx_decoder *e, const uint8_t *d_in, int slotBytes, int nBlocks, float *d_pcm, int32_t *d_bits, void *hipStream) {
    if (!d_pcm) { ulcx_set_error("ulcx_decode_dev: bad argument"); return ULCX_ERR_ARG; }
    return decode_dev_any(e, d_in, slotBytes, nBlocks, d_pcm, nullptr, d_bits, hipStream);
}
extern "C" int ulcx_decode_dev_pcm16(ulcx_decoder *e, const uint8_t *d_in, int slotBytes, int nBlocks, int16_t *d_pcm16, int32_t *d_bits, void *hipStream) {
    if (!d_pcm16) { ulcx_set_error("ulcx_decode_dev_pcm16: bad argument"); return ULCX_ERR_ARG; }
    return decode_dev_any(e, d_in, slotBytes, nBlocks, nullptr, d_pcm16, d_bits, hipStream);
}
extern "C" int ulcx_decode_host(ulcx_decoder *e, const uint8_t *h_in, int slotBytes, int nBlocks, float *h_pcm, int32_t *h_bits) {
    if (!e || !h_in || !h_pcm || !h_bits || nBlocks < 1 || nBlocks > e->maxK || slotBytes < 1) return ULCX_ERR_ARG;
    CKR(hipSetDevice(e->device));
    size_t NBmax = (size_t)e->B * e->maxK, cb = (size_t)e->C * e->BS, NB = (size_t)e->B * nBlocks;
    size_t inBytes = NBmax * (size_t)slotBytes + 16;
    if (!e->d_in || e->d_in_bytes < inBytes) {
        int rc;
        if ((rc = dalloc(e->allocs, &e->d_in, inBytes, true))) return rc;
        e->d_in_bytes = inBytes;
    }
    if (!e->d_pcm) {
        int rc;
        if ((rc = dalloc(e->allocs, &e->d_pcm, NBmax * cb, false))) return rc;
        if ((rc = dalloc(e->allocs, &e->d_bits, NBmax, false))) return rc;
    }
    CKR(hipMemcpy(e->d_in, h_in, NB * slotBytes, hipMemcpyHostToDevice));
    int rc = ulcx_decode_dev(e, e->d_in, slotBytes, nBlocks, e->d_pcm, e->d_bits, nullptr);
    if (rc) return rc;
    CKR(hipDeviceSynchronize());
    CKR(hipMemcpy(h_pcm, e->d_pcm, sizeof(float) * NB * cb, hipMemcpyDeviceToHost));
    CKR(hipMemcpy(h_bits, e->d_bits, sizeof(int32_t) * NB, hipMemcpyDeviceToHost));
    return ULCX_OK;
}
extern "C" int ulcx_decode_block1(ulcx_decoder *e, const uint8_t *h_in, int nBytes, float *h_pcm, int32_t *bits, int32_t *lastSubBlockSize) {
    return ulcx_decode_block1_rng(e, h_in, nBytes, h_pcm, bits, lastSubBlockSize, nullptr);
}
// rngState: the noise generator's state (ulcDecoder.c:75-81) before the block in, after it out.  The reference keeps it in a
// function-static word, i.e. ONE state per process that every decoder object draws from; a caller that wants that behaviour
// (the drop-in of section 1 does) owns the word and hands it through here.  NULL: the state stays with this decoder object.
extern "C" int ulcx_decode_block1_rng(ulcx_decoder *e, const uint8_t *h_in, int nBytes, float *h_pcm, int32_t *bits, int32_t *lastSubBlockSize, uint32_t *rngState) {
    if (!e || !h_in || !h_pcm || nBytes < 1 || e->B != 1 || e->maxK != 1) { ulcx_set_error("ulcx_decode_block1: needs a decoder of one stream, one block per call"); return ULCX_ERR_ARG; }
    CKR(hipSetDevice(e->device));
    const size_t cb = (size_t)e->C * e->BS;
    const int slot = 2 * e->C * e->BS + 16;                       // the largest block
    if (nBytes > slot) nBytes = slot;
    if (!e->b1Init) {
        int rc;
        if (!e->d_in || e->d_in_bytes < (size_t)slot + 16) { if ((rc = dalloc(e->allocs, &e->d_in, (size_t)slot + 16, true))) return rc; e->d_in_bytes = (size_t)slot + 16; }
        if (!e->d_pcm) { if ((rc = dalloc(e->allocs, &e->d_pcm, cb, false))) return rc; if ((rc = dalloc(e->allocs, &e->d_bits, 1, false))) return rc; }
        if (!e->b1Stream) CKR(hipStreamCreateWithFlags(&e->b1Stream, hipStreamNonBlocking));
        if (!e->pinIn) CKR(hipHostMalloc((void **)&e->pinIn, (size_t)slot, hipHostMallocDefault));
        if (!e->pinPcm) CKR(hipHostMalloc((void **)&e->pinPcm, sizeof(float) * cb, hipHostMallocDefault));
        if (!e->pinMeta) CKR(hipHostMalloc((void **)&e->pinMeta, 4 * sizeof(int32_t), hipHostMallocDefault));
        e->b1Init = true; e->b1Slot = slot;
        e->timing = false;
    }
    auto enqueue = [&]() -> int {
        CKR(hipMemcpyAsync(e->d_in, e->pinIn, (size_t)slot, hipMemcpyHostToDevice, e->b1Stream));
        CKR(hipMemcpyAsync(e->ctx.seed, &e->pinMeta[2], sizeof(uint32_t), hipMemcpyHostToDevice, e->b1Stream));
        int rc = ulcx_decode_dev(e, e->d_in, slot, 1, e->d_pcm, e->d_bits, e->b1Stream);
        if (rc) return rc;
        CKR(hipMemcpyAsync(&e->pinMeta[3], e->ctx.seed, sizeof(uint32_t), hipMemcpyDeviceToHost, e->b1Stream));
        CKR(hipMemcpyAsync(e->pinPcm, e->d_pcm, sizeof(float) * cb, hipMemcpyDeviceToHost, e->b1Stream));
        CKR(hipMemcpyAsync(&e->pinMeta[0], e->d_bits, sizeof(int32_t), hipMemcpyDeviceToHost, e->b1Stream));
        CKR(hipMemcpyAsync(&e->pinMeta[1], e->ctx.lastSub, sizeof(int32_t), hipMemcpyDeviceToHost, e->b1Stream));
        return ULCX_OK;
    };
    memcpy(e->pinIn, h_in, (size_t)nBytes);
    memset(e->pinIn + nBytes, 0, (size_t)(slot - nBytes));         // (only the block's own bytes are the caller's: the rest of the slot reads as zero)
    if (!rngState && e->b1SeedStale) {                            // mixed use: ulcx_decode_dev / _host / _packed ran on this object since
        CKR(hipDeviceSynchronize());
        CKR(hipMemcpy(&e->b1Seed, e->ctx.seed, sizeof(uint32_t), hipMemcpyDeviceToHost));
    }
    e->b1SeedStale = false;
    e->pinMeta[2] = (int32_t)(rngState ? *rngState : e->b1Seed);
    struct InB1 { ulcx_decoder *d; InB1(ulcx_decoder *x) : d(x) { d->inBlock1 = true; } ~InB1() { d->inBlock1 = false; } } inB1(e);
    if (!e->b1Graphed && !e->b1NoGraph) {
        bool ok = hipStreamBeginCapture(e->b1Stream, hipStreamCaptureModeThreadLocal) == hipSuccess;
        if (ok) {
            const int rc = enqueue();
            hipGraph_t g = nullptr;
            const hipError_t ee = hipStreamEndCapture(e->b1Stream, &g);
            ok = (rc == ULCX_OK) && ee == hipSuccess && g != nullptr;
            if (ok) ok = hipGraphInstantiate(&e->b1Exec, g, nullptr, nullptr, 0) == hipSuccess;
            if (ok) { e->b1Graph = g; e->b1Graphed = true; }
            else if (g) hipGraphDestroy(g);
        }
        if (!ok) { (void)hipGetLastError(); e->b1NoGraph = true; }
    }
    if (e->b1Graphed) CKR(hipGraphLaunch(e->b1Exec, e->b1Stream));
    else { int rc = enqueue(); if (rc) return rc; }
    CKR(hipStreamSynchronize(e->b1Stream));
    memcpy(h_pcm, e->pinPcm, sizeof(float) * cb);
    if (bits) *bits = e->pinMeta[0];
    if (lastSubBlockSize) *lastSubBlockSize = e->pinMeta[1];
    e->b1Seed = (uint32_t)e->pinMeta[3];
    if (rngState) *rngState = e->b1Seed;
    return ULCX_OK;
}

// ---------------------------------------------------------------------------
// .ulc container + packed streams (tools/ulc_Helper.h:10-20, ulcEncodeTool.c:92-100,160-195, ulcDecodeTool.c:73-80,123-166)
// ---------------------------------------------------------------------------
static void put16(uint8_t *p, uint32_t v) { p[0] = (uint8_t)v; p[1] = (uint8_t)(v >> 8); }
static void put32(uint8_t *p, uint32_t v) { put16(p, v); put16(p + 2, v >> 16); }
static uint32_t get16(const uint8_t *p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8); }
static uint32_t get32(const uint8_t *p) { return get16(p) | (get16(p + 2) << 16); }
extern "C" void ulcx_ulc_header_pack(uint8_t dst[24], const ulcx_file_header *h) {
    put32(dst + 0x00, h->Magic); put16(dst + 0x04, h->BlockSize); put16(dst + 0x06, h->MaxBlockSize);
    put32(dst + 0x08, h->nBlocks); put32(dst + 0x0C, h->RateHz); put16(dst + 0x10, h->nChan);
    put16(dst + 0x12, h->RateKbps); put32(dst + 0x14, h->StreamOffs);
}
extern "C" int ulcx_ulc_header_parse(ulcx_file_header *h, const uint8_t *src, size_t len) {
    if (!h || !src || len < 24) { ulcx_set_error("ulc header: need 24 bytes"); return ULCX_ERR_ARG; }
    h->Magic = get32(src); h->BlockSize = (uint16_t)get16(src + 4); h->MaxBlockSize = (uint16_t)get16(src + 6);
    h->nBlocks = get32(src + 8); h->RateHz = get32(src + 12); h->nChan = (uint16_t)get16(src + 16);
    h->RateKbps = (uint16_t)get16(src + 18); h->StreamOffs = get32(src + 20);
    if (h->Magic != ULCX_ULC_MAGIC) { ulcx_set_error("not a ULC2 container"); return ULCX_ERR_ARG; }   /* ulcDecodeTool.c:77-80 */
    return ULCX_OK;
}
extern "C" int ulcx_ulc_rate_kbps(uint64_t totalBytes, uint32_t RateHz, uint32_t BlockSize, uint32_t nBlocks) {
    double avg = (double)totalBytes * 8.0 * RateHz / 1000.0 / ((double)BlockSize * nBlocks);          /* ulcEncodeTool.c:173,190 */
    return (int)lrint(avg);
}
extern "C" int ulcx_pack_streams_dev(int device, int nStreams, int nBlocks, int slotBytes, const uint8_t *d_slots, const int32_t *d_bits,
                                     uint8_t *d_payload, long long payloadStride, int32_t *d_payloadBytes, int32_t *d_maxBlock, void *hipStream) {
    if (nStreams < 1 || nBlocks < 1 || slotBytes < 1 || !d_slots || !d_bits || !d_payload || !d_payloadBytes || payloadStride < 1) { ulcx_set_error("ulcx_pack_streams_dev: bad argument"); return ULCX_ERR_ARG; }
    int rc = select_device(device);
    if (rc) return rc;
    return ulcx_pack_launch(nStreams, nBlocks, slotBytes, d_slots, d_bits, d_payload, payloadStride, d_payloadBytes, d_maxBlock, (hipStream_t)hipStream);
}
extern "C" int ulcx_decode_packed_dev(ulcx_decoder *e, const uint8_t *d_payload, long long payloadStride, const int32_t *d_payloadBytes,
                                      int nBlocks, float *d_pcm, int32_t *d_bits, void *hipStream) {
    if (!e || !d_payload || !d_payloadBytes || !d_pcm || !d_bits || payloadStride < 1 || nBlocks < 1 || nBlocks > e->maxK) { ulcx_set_error("ulcx_decode_packed_dev: bad argument"); return ULCX_ERR_ARG; }
    CKR(hipSetDevice(e->device));
    UlcxDecCtx c = e->ctx;
    c.K = nBlocks; c.slot = 0; c.in = d_payload; c.pcm = d_pcm; c.pcm16 = nullptr; c.bits = d_bits;
    c.packed = 1; c.payStride = payloadStride; c.payBytes = d_payloadBytes;
    c.inBytes = (long long)e->B * payloadStride;
    int rc = dec_launch(e, c, (hipStream_t)hipStream);
    e->evRecorded = (rc == ULCX_OK) && e->timing;
    return rc;
}
extern "C" int ulcx_decode_packed_host(ulcx_decoder *e, const uint8_t *h_payload, long long payloadStride, const int32_t *h_payloadBytes,
                                       int nBlocks, float *h_pcm, int32_t *h_bits) {
    if (!e || !h_payload || !h_payloadBytes || !h_pcm || !h_bits || nBlocks < 1 || nBlocks > e->maxK || payloadStride < 1) return ULCX_ERR_ARG;
    CKR(hipSetDevice(e->device));
    size_t cb = (size_t)e->C * e->BS, NB = (size_t)e->B * nBlocks;
    uint8_t *dp = nullptr; int32_t *dn = nullptr; float *dpcm = nullptr; int32_t *dbits = nullptr;
    size_t payBytes = (size_t)e->B * (size_t)payloadStride + 16;
    CKR(hipMalloc((void **)&dp, payBytes)); CKR(hipMalloc((void **)&dn, sizeof(int32_t) * e->B));
    CKR(hipMalloc((void **)&dpcm, sizeof(float) * NB * cb)); CKR(hipMalloc((void **)&dbits, sizeof(int32_t) * NB));
    CKR(hipMemset(dp, 0, payBytes));
    CKR(hipMemcpy(dp, h_payload, (size_t)e->B * (size_t)payloadStride, hipMemcpyHostToDevice));
    CKR(hipMemcpy(dn, h_payloadBytes, sizeof(int32_t) * e->B, hipMemcpyHostToDevice));
    int rc = ulcx_decode_packed_dev(e, dp, payloadStride, dn, nBlocks, dpcm, dbits, nullptr);
    if (rc == ULCX_OK) {
        CKR(hipDeviceSynchronize());
        CKR(hipMemcpy(h_pcm, dpcm, sizeof(float) * NB * cb, hipMemcpyDeviceToHost));
        CKR(hipMemcpy(h_bits, dbits, sizeof(int32_t) * NB, hipMemcpyDeviceToHost));
    }
    hipFree(dp); hipFree(dn); hipFree(dpcm); hipFree(dbits);
    return rc;
}

// Whole files: the payloads go to the device once, every later call decodes the next nBlocks of every stream from there
// (ulcx_decode_packed_host re-uploads everything per call: fine for one call, quadratic over a long file).
extern "C" int ulcx_decoder_upload_payload(ulcx_decoder *e, const uint8_t *h_payload, long long payloadStride, const int32_t *h_payloadBytes) {
    if (!e || !h_payload || !h_payloadBytes || payloadStride < 1) return ULCX_ERR_ARG;
    CKR(hipSetDevice(e->device));
    if (e->d_pay) { hipFree(e->d_pay); e->d_pay = nullptr; }
    if (e->d_payBytes) { hipFree(e->d_payBytes); e->d_payBytes = nullptr; }
    size_t bytes = (size_t)e->B * (size_t)payloadStride;
    CKR(hipMalloc((void **)&e->d_pay, bytes + 16)); CKR(hipMalloc((void **)&e->d_payBytes, sizeof(int32_t) * e->B));
    CKR(hipMemset(e->d_pay, 0, bytes + 16));
    CKR(hipMemcpy(e->d_pay, h_payload, bytes, hipMemcpyHostToDevice));
    CKR(hipMemcpy(e->d_payBytes, h_payloadBytes, sizeof(int32_t) * e->B, hipMemcpyHostToDevice));
    e->payStride = payloadStride;
    return ulcx_decoder_reset(e);
}
extern "C" int ulcx_decode_resident_host(ulcx_decoder *e, int nBlocks, float *h_pcm, int32_t *h_bits) {
    if (!e || !h_pcm || !h_bits || nBlocks < 1 || nBlocks > e->maxK) return ULCX_ERR_ARG;
    if (!e->d_pay) { ulcx_set_error("ulcx_decode_resident_host: no payload uploaded"); return ULCX_ERR_ARG; }
    CKR(hipSetDevice(e->device));
    size_t cb = (size_t)e->C * e->BS, NB = (size_t)e->B * nBlocks;
    if (!e->d_pcm) {
        int rc0; size_t NBmax = (size_t)e->B * e->maxK;
        if ((rc0 = dalloc(e->allocs, &e->d_pcm, NBmax * cb, false))) return rc0;
        if ((rc0 = dalloc(e->allocs, &e->d_bits, NBmax, false))) return rc0;
    }
    int rc = ulcx_decode_packed_dev(e, e->d_pay, e->payStride, e->d_payBytes, nBlocks, e->d_pcm, e->d_bits, nullptr);
    if (rc != ULCX_OK) return rc;
    CKR(hipDeviceSynchronize());
    CKR(hipMemcpy(h_pcm, e->d_pcm, sizeof(float) * NB * cb, hipMemcpyDeviceToHost));
    CKR(hipMemcpy(h_bits, e->d_bits, sizeof(int32_t) * NB, hipMemcpyDeviceToHost));
    return ULCX_OK;
}

// diagnostic, only in a `make EXTRA=-DULCX_DSYN_STAMPS` build (tools/dsyn_stamps.py): first nBytes of the general-path staging
// buffer, where that build leaves per-phase cycle counts
#ifdef ULCX_DSYN_STAMPS
extern "C" int ulcx_decoder_debug_scratch(ulcx_decoder *e, void *h_out, size_t strideBytes, size_t nBytes, int nStreams);
extern "C" int ulcx_decoder_debug_scratch(ulcx_decoder *e, void *h_out, size_t strideBytes, size_t nBytes, int nStreams) {
    if (!e || !h_out) return ULCX_ERR_ARG;
    CKR(hipSetDevice(e->device));
    CKR(hipDeviceSynchronize());
    for (int s = 0; s < nStreams && s < e->B; s++)
        CKR(hipMemcpy((char *)h_out + (size_t)s * nBytes, (const char *)e->ctx.scratch + (size_t)s * strideBytes, nBytes, hipMemcpyDeviceToHost));
    return ULCX_OK;
}
#endif
// per-stage hipEvents around every kernel (ulcx_*_stage_ms): on by default; a caller that does not read them can switch
// them off - each record is a marker packet in the stream between two kernels
extern "C" int ulcx_encoder_set_timing(ulcx_encoder *e, int on) { if (!e) return ULCX_ERR_ARG; e->timing = on != 0; if (!on) e->evRecorded = false; return ULCX_OK; }
extern "C" int ulcx_decoder_set_timing(ulcx_decoder *e, int on) { if (!e) return ULCX_ERR_ARG; e->timing = on != 0; if (!on) e->evRecorded = false; return ULCX_OK; }
static const char *kDecStage[ULCX_DEC_STAGES] = { "k_dscan", "k_dsyn" };
extern "C" const char *ulcx_decoder_stage_name(int i) { return (i >= 0 && i < ULCX_DEC_STAGES) ? kDecStage[i] : ""; }
extern "C" int ulcx_decoder_stage_ms(ulcx_decoder *e, float *ms, int maxStages) {
    if (!e || !e->evRecorded) return 0;
    int n = 0;
    for (int i = 0; i < ULCX_DEC_STAGES && i < maxStages; i++) {
        float t = 0;
        if (hipEventElapsedTime(&t, e->ev[i], e->ev[i + 1]) != hipSuccess) break;
        ms[n++] = t;
    }
    return n;
}
