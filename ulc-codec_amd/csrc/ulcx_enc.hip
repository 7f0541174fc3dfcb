// ulcx_enc.hip — batched ulc-codec encoder for gfx950 (MI355X), hand-written HIP.
//
// One call encodes K consecutive blocks of B independent streams.  Pipeline
// (reference call stack SURVEY.md §3A):
//   wc_energy / wc_forward / wc_backward / wc_integrate / wc_decide
//        transient detector -> WindowCtrl per block   (ulcEncoder_WindowControl.c:41-239)
//   xf   one workgroup per block: frames from the input timeline (closed form of the
//        lapping FIFO, BlockTransform.c:175-224), sine window, MDCT+MDST through two
//        DCT-IV = complex FFTs staged entirely in LDS, normalise, keys, per-line
//        energies                                      (BlockTransform.c:229-281)
//   cplx        ordered f32 sums -> BlockComplexity, nOutCoef (BlockTransform.c:279-325, ulcEncoder.c:140-158)
//   bark_uniform/bark_levels/nbark/nline  noise log-spectrum (un-decimated blocks on the geometry-uniform
//               kernel, the rest lane per subblock)    (ulcEncoder_Psyopt.c:168-250)
//   pbark       masking Bark levels                    (ulcEncoder_Psyopt.c:60-155)
//   select      one wave per block: keys (coefficient + masking level, BlockTransform.c:337-345) in registers,
//               the nOutCoef-th largest by bisection; exact heapsort emulation only for tie groups
//               straddling the cut                     (BlockTransform.c:20-77)
//   encode/pack nybble stream                          (ulcEncoder_Encode.c:23-360, ulcEncoder_NoiseFill.c)
// All float arithmetic is written in the reference's operation order and this file is
// compiled with -ffp-contract=off: no fused multiply-add is formed anywhere except the
// explicit ones inside the glibc restatements (ulcx_libm.h).
#include "ulcx_enc_dev.h"

// ---------------------------------------------------------------------------
// launcher
// ---------------------------------------------------------------------------
// 4 padded arrays of BS/2 complex + BS/4 twiddles + counter (+ BS/2 floats of line energies for C > 2)
size_t ulcx_enc_xf_lds_bytes(int BS, int C) {
    if (BS > 8192) return (size_t)BS * 4;                       // k_xf_big: one unpadded array of BS/2 complex
    int ps = ulcx_xf_pad_shift(BS, C);
    size_t z = (size_t)4 * (BS + (BS >> ps)) * 4;               // four padded arrays of BS/2 complex
    size_t full = z + (size_t)BS * 2 + 32 + (C > 2 ? (size_t)BS * 2 : 0);
    return full <= ULCX_LDS_LIMIT ? full : z + (size_t)BS * 2 + 32;   // (C > 2 at BlockSize 8192: twiddles stay in global memory)
}

// launch the input-reading kernels for the call's sample type (float | PCM16)
static void launch_wc_energy(const UlcxEncCtx &c, unsigned grid, hipStream_t st, int k0, int k1) {
    if (c.pcm16) hipLaunchKernelGGL(k_wc_energy<int16_t>, dim3(grid), dim3(WG), 0, st, c, k0, k1);
    else hipLaunchKernelGGL(k_wc_energy<float>, dim3(grid), dim3(WG), 0, st, c, k0, k1);
}

static void launch_wc_ef(const UlcxEncCtx &c, hipStream_t st, int k0, int k1) {
    static_assert(EF_LDS_BYTES <= 48 * 1024, "k_wc_ef: raise the dynamic LDS limit with hipFuncSetAttribute");
    if (c.pcm16) hipLaunchKernelGGL((k_wc_ef<EF_NW, int16_t>), dim3((c.B + EF_SPW - 1) / EF_SPW), dim3(EF_NW * 64), EF_LDS_BYTES, st, c, k0, k1);
    else hipLaunchKernelGGL((k_wc_ef<EF_NW, float>), dim3((c.B + EF_SPW - 1) / EF_SPW), dim3(EF_NW * 64), EF_LDS_BYTES, st, c, k0, k1);
}

static void launch_xf(const UlcxEncCtx &c, unsigned grid, size_t lds, hipStream_t st, int k0, int k1) {
    if (c.BS > 8192) {                                      // one array at a time (k_xf_big)
        if (c.pcm16) hipLaunchKernelGGL(k_xf_big<int16_t>, dim3(grid), dim3(WG), lds, st, c, k0, k1);
        else hipLaunchKernelGGL(k_xf_big<float>, dim3(grid), dim3(WG), lds, st, c, k0, k1);
        return;
    }
    if (c.pcm16) {
        if (c.C == 2) hipLaunchKernelGGL((k_xf<true, int16_t>), dim3(grid), dim3(WG), lds, st, c, k0, k1);
        else hipLaunchKernelGGL((k_xf<false, int16_t>), dim3(grid), dim3(WG), lds, st, c, k0, k1);
    } else {
        if (c.C == 2) hipLaunchKernelGGL((k_xf<true, float>), dim3(grid), dim3(WG), lds, st, c, k0, k1);
        else hipLaunchKernelGGL((k_xf<false, float>), dim3(grid), dim3(WG), lds, st, c, k0, k1);
    }
}

static void launch_state_update(const UlcxEncCtx &c, hipStream_t st) {
    if (c.pcm16) hipLaunchKernelGGL(k_state_update<int16_t>, dim3(c.B), dim3(WG), 0, st, c);
    else hipLaunchKernelGGL(k_state_update<float>, dim3(c.B), dim3(WG), 0, st, c);
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { ulcx_set_error("%s: %s", #x, hipGetErrorString(e_)); return ULCX_ERR_HIP; } } while (0)

// Per-kernel hipEvents (on the launch stream) bracket every kernel of the first pass so
// bench.py can price each one against the roofline live; ev holds ULCX_ENC_STAGES+1 events.
const char *const ulcx_enc_stage_names[ULCX_ENC_STAGES_REPORTED] = {
    "k_wc_energy", "k_wc_forward", "k_wc_backward", "k_wc_integrate", "k_wc_decide",
    "k_xf", "k_cplx", "k_pbark", "k_mask",
    "k_select", "k_nbark", "(k_nline: gone)", "k_heapsel", "k_nsums", "k_tails", "k_encode_wave", "k_encode_units", "k_pack", "cbr_probe_passes", "k_state_update", "wc_pipeline_exposed",
};

int ulcx_enc_launch(const UlcxEncCtx &cIn, hipStream_t st, hipEvent_t *ev, const UlcxEncAux &aux) {
    UlcxEncCtx c = cIn;                                        // (keyFinal is set below for geometries without a wave selection kernel)
    hipStream_t side = aux.side, side2 = aux.side2, side3 = aux.side3;
    hipEvent_t evFork = aux.evFork, evJoin = aux.evJoin, evFork2 = aux.evFork2, *evWC = aux.evWC;
    const int wcPipe = (side && side2 && side3) ? aux.wcPipe : 1;
    if (aux.nXf) *aux.nXf = 0;
    if (c.mode != ULCX_MODE_VBR) CK(hipMemsetAsync(c.cbrLive, 0, sizeof(int), st));
    if (c.barkRing) CK(hipMemsetAsync(c.decCount, 0, sizeof(int), st));           // k_xf lists this call's decimated blocks
    int NB = c.B * c.K;
    int stage = 0;
#define MARK() do { if (ev) CK(hipEventRecord(ev[stage++], st)); } while (0)
    MARK();
    // --- window control + transform
    hipEvent_t *evX = evWC + 7 + 3 * ULCX_WC_MAXCH;            // [ULCX_XF_MAXCH] transform chunk done, [ULCX_XF_MAXCH]: all early k_cplx launches done
    const bool cplxEarly = wcPipe > 1;                        // the ordered complexity sums per transform chunk, beside the next chunk
    {
        int SG = (c.B + 63) / 64;
        // Chunks of blocks: the window-control kernels of chunk j+1.. (two stream-long serial recurrences, a few
        // hundred waves: latency-bound, nearly no machine resources) run on the side stream beside the
        // transform of chunk j on the main stream.  wcPipe = 1 keeps everything on the main stream.
        const int nCh = wcPipe;
        size_t lds = ulcx_enc_xf_lds_bytes(c.BS, c.C);
        if (lds > 48 * 1024) {
            CK(hipFuncSetAttribute((const void *)k_xf<true, float>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            CK(hipFuncSetAttribute((const void *)k_xf<false, float>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            CK(hipFuncSetAttribute((const void *)k_xf<true, int16_t>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            CK(hipFuncSetAttribute((const void *)k_xf<false, int16_t>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            CK(hipFuncSetAttribute((const void *)k_xf_big<float>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            CK(hipFuncSetAttribute((const void *)k_xf_big<int16_t>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        }
        const bool wcFuse = c.C == 2 && aux.wcFuse;   // stereo: k_wc_energy + k_wc_forward in one kernel (k_wc_ef)
        auto launch_wc = [&](hipStream_t s2, int k0, int k1, bool marks) -> int {
            int kc = k1 - k0;
            if (wcFuse) { if (marks) MARK(); launch_wc_ef(c, s2, k0, k1);                                             if (marks) MARK(); }
            else {
            launch_wc_energy(c, (unsigned)(SG * ((kc * c.BS) / 64)), s2, k0, k1);   if (marks) MARK();
            hipLaunchKernelGGL(k_wc_forward, dim3((c.B * 2 + 63) / 64), dim3(64), 0, s2, c, k0, k1);                  if (marks) MARK();
            }
            hipLaunchKernelGGL(k_wc_backward, dim3(SG * kc), dim3(64), 0, s2, c, k0, k1);                             if (marks) MARK();
            hipLaunchKernelGGL(k_wc_integrate, dim3((c.B + 63) / 64), dim3(64), 0, s2, c, k0, k1);                    if (marks) MARK();
            hipLaunchKernelGGL(k_wc_decide, dim3((c.B * kc + 63) / 64), dim3(64), 0, s2, c, k0, k1);                  if (marks) MARK();
            return ULCX_OK;
        };
        if (nCh <= 1) {
            int rc = launch_wc(st, 0, c.K, true); if (rc) return rc;
            launch_xf(c, ((NB + 7) / 8) * 8, lds, st, 0, c.K);
            MARK();
        } else {
            for (int i = 0; i < 5; i++) MARK();                    // (window-control stages: hidden in the k_xf interval in this mode)
            // side: envelope + forward recurrence of step w (the sample-rate chain, back to back over the steps);
            // side2: backward_w behind forward_w;  side3: integrate_w, decide_w behind backward_w;
            // main: transform chunk j behind the step that decides its last block.  The first chunk is a single block so the transform starts early.
            hipEvent_t ev0 = evWC[0], *evF = evWC + 1, *evB = evWC + 1 + ULCX_WC_MAXCH, *evD = evWC + 1 + 2 * ULCX_WC_MAXCH;
            // The window-control kernels advance in uniform steps of a few blocks (ULCX_WC_STEPS; 0 = in the same chunks as
            // the transform: first chunk one block, then thirds).  Measured with the chunked transform: 4 steps 8.65-8.70 ms
            // per step of 65536 blocks, 8 steps 8.73-8.81, 16 steps 8.9-9.0 (every launch of a chain kernel costs its fixed
            // latency, and a step that has to be dispatched beside a transform chunk waits for its workgroup slots).
            int nW = aux.wcSteps;
            if (nW < 0) nW = (c.K >= 8) ? 4 : 0;                               // default: 4 uniform steps (of >= 2 blocks)
            if (nW > ULCX_WC_MAXCH) nW = ULCX_WC_MAXCH;
            if (nW > c.K) nW = c.K;
            const bool sameCuts = nW < 1;
            if (sameCuts) nW = nCh;
            int cut[ULCX_XF_MAXCH + 1];
            // (transform chunks cut where the window-control steps end - eight blocks behind the first step instead of one - :
            //  the transform's interval +0.37 ms, the exposed window control -0.37 ms, the phase the same 4.0 ms)
            cut[0] = 0; cut[1] = 1;
            for (int j = 2; j <= nCh; j++) cut[j] = 1 + (c.K - 1) * (j - 1) / (nCh - 1);
            int wcs[ULCX_WC_MAXCH + 1];
            for (int w = 0; w <= nW; w++) wcs[w] = sameCuts ? cut[w] : (int)((long long)c.K * w / nW);
            CK(hipEventRecord(ev0, st));
            CK(hipStreamWaitEvent(side, ev0, 0));
            int jx = 0;                                        // next transform chunk to enqueue
            for (int w = 0; w < nW; w++) {
                const int k0 = wcs[w], k1 = wcs[w + 1], kc = k1 - k0;
                if (wcFuse) launch_wc_ef(c, side, k0, k1);
                else {
                    launch_wc_energy(c, (unsigned)(SG * ((kc * c.BS) / 64)), side, k0, k1);
                    hipLaunchKernelGGL(k_wc_forward, dim3((c.B * 2 + 63) / 64), dim3(64), 0, side, c, k0, k1);
                }
                CK(hipEventRecord(evF[w], side));
                CK(hipStreamWaitEvent(side2, evF[w], 0));
                hipLaunchKernelGGL(k_wc_backward, dim3(SG * kc), dim3(64), 0, side2, c, k0, k1);
                CK(hipEventRecord(evB[w], side2));
                CK(hipStreamWaitEvent(side3, evB[w], 0));
                hipLaunchKernelGGL(k_wc_integrate, dim3((c.B + 63) / 64), dim3(64), 0, side3, c, k0, k1);
                hipLaunchKernelGGL(k_wc_decide, dim3((c.B * kc + 63) / 64), dim3(64), 0, side3, c, k0, k1);
                CK(hipEventRecord(evD[w], side3));
                // transform chunks whose last block is now decided
                while (jx < nCh && cut[jx + 1] <= k1) {
                    int x0 = cut[jx], x1 = cut[jx + 1];
                    int nbk = c.B * (x1 - x0);
                    CK(hipStreamWaitEvent(st, evD[w], 0));
                    if (ev) CK(hipEventRecord(aux.evXf[2 * jx], st));
                    if (!(ULCX_DBG(c) & 0x2000))               // (ablation build: window control alone)
                    launch_xf(c, ((nbk + 7) / 8) * 8, lds, st, x0, x1);
                    if (ev) CK(hipEventRecord(aux.evXf[2 * jx + 1], st));
                    if (cplxEarly) CK(hipEventRecord(evX[jx], st));
                    jx++;
                }
            }
            if (aux.nXf) *aux.nXf = nCh;
            MARK();
            // the ordered complexity sums (k_cplx: lane-serial, HBM-bound) per transform chunk, on the envelope kernels' stream
            // (all of those are enqueued by now): only the last chunk's are left beside the masking sums
            if (cplxEarly) {
                for (int j = 0; j < nCh; j++) {
                    CK(hipStreamWaitEvent(side, evX[j], 0));
                    const int kc2 = cut[j + 1] - cut[j];
                    hipLaunchKernelGGL(k_cplx, dim3((c.B * kc2 + 63) / 64), dim3(64), 0, side, c, cut[j], cut[j + 1]);
                }
                CK(hipEventRecord(evX[ULCX_XF_MAXCH], side));
            }
        }
    }
    if (ULCX_DBG(c) & 0x6000) { MARK(); return ULCX_OK; }     // (ablation build: stop behind window control / transform)
    int nUnits = NB * c.C * 4;
    // (the noise log-spectrum does not feed the keys: it is launched after the selection so that the
    //  main stream has work to run beside the side-stream heapsort of tie-straddle blocks)
    const size_t barkLds = (size_t)c.barkRing * 3 * 64 * 8 + (size_t)2 * BK_TILE_FLOATS * 4;
    if (c.barkRing && barkLds > 48 * 1024) {
        CK(hipFuncSetAttribute((const void *)k_bark_uniform<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)barkLds));
        CK(hipFuncSetAttribute((const void *)k_bark_uniform<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)barkLds));
    }
    auto launch_noise = [&](hipStream_t s2, bool ev0) -> int {
        if (c.barkRing) {
            hipLaunchKernelGGL(k_bark_uniform<true>, dim3((NB * c.C + 63) / 64), dim3(256), barkLds, s2, c);
            hipLaunchKernelGGL(k_bark_levels<true>, dim3((unsigned)(((size_t)NB * c.C * 32 + WG - 1) / WG)), dim3(WG), 0, s2, c);
        }
        hipLaunchKernelGGL(k_nbark, dim3((nUnits + 63) / 64), dim3(64), 0, s2, c, c.barkRing ? 1 : 0);    if (ev0) MARK();
        if (ev0) MARK();
        return ULCX_OK;
    };
    // The noise log-spectrum (k_nbark: lane-serial ordered sums, latency-bound; k_nline) depends on the
    // transform only: it runs on a side stream beside the psychoacoustics / selection chain.
    const bool noiseAside = (side2 != nullptr && side3 != nullptr);
    hipEvent_t evN0 = evWC[1 + 3 * ULCX_WC_MAXCH], evNoise = evWC[2 + 3 * ULCX_WC_MAXCH];
    // Three lane-serial latency-bound kernels (k_pbark, k_cplx, k_nbark) fill the machine's wave slots by themselves:
    // running all three at once only makes each slower.  k_cplx (~1000 waves) runs beside k_pbark; the noise chain
    // starts behind k_pbark and runs beside the throughput-bound k_mask / k_select.
    hipEvent_t evCplx = evWC[3 + 3 * ULCX_WC_MAXCH], evTail0 = evWC[4 + 3 * ULCX_WC_MAXCH], evTail1 = evWC[5 + 3 * ULCX_WC_MAXCH], evState = evWC[6 + 3 * ULCX_WC_MAXCH];
    if (noiseAside) {
        CK(hipEventRecord(evN0, st));
        CK(hipStreamWaitEvent(side3, evN0, 0));
        if (cplxEarly) CK(hipStreamWaitEvent(side3, evX[ULCX_XF_MAXCH], 0));          // (launched per transform chunk: below)
        else hipLaunchKernelGGL(k_cplx, dim3((NB + 63) / 64), dim3(64), 0, side3, c, 0, c.K);
        CK(hipEventRecord(evCplx, side3));
        MARK();
        // the state for the next call only needs the transform to be done with the history: off the main stream
        launch_state_update(c, side3);
        CK(hipEventRecord(evState, side3));
    } else { hipLaunchKernelGGL(k_cplx, dim3((NB + 63) / 64), dim3(64), 0, st, c, 0, c.K);                 MARK(); }
    const bool noiseEarly = noiseAside;                        // the noise chain right behind the transform (it only needs nsum), beside the masking sums
    if (noiseEarly) {
        CK(hipStreamWaitEvent(side2, evN0, 0));
        int rcn = launch_noise(side2, false); if (rcn) return rcn;
        CK(hipEventRecord(evNoise, side2));
    }
    {
        const bool uniP = c.barkRing != 0;                  // masking sums of the un-decimated blocks on the geometry-uniform kernel too
        if (uniP) {
            hipLaunchKernelGGL(k_bark_uniform<false>, dim3((NB + 63) / 64), dim3(256), barkLds, st, c);
            hipLaunchKernelGGL(k_bark_levels<false>, dim3((unsigned)(((size_t)NB * 32 + WG - 1) / WG)), dim3(WG), 0, st, c);
        }
        hipLaunchKernelGGL(k_pbark, dim3((NB * 4 + 63) / 64), dim3(64), 0, st, c, uniP ? 1 : 0);          MARK();
        if (noiseAside && !noiseEarly) {
            CK(hipEventRecord(evTail0, st));                       // (reused: behind k_pbark)
            CK(hipStreamWaitEvent(side2, evTail0, 0));
            int rcn = launch_noise(side2, false); if (rcn) return rcn;
            CK(hipEventRecord(evNoise, side2));
        }
        // ("k_mask": gone - the masking level per line is formed where the keys are, mask_level(); geometries on the generic
        //  selection kernel evaluate it per key)
        MARK();
    }
    if (noiseAside) CK(hipStreamWaitEvent(st, evCplx, 0));
    // --- selection + encode pass(es)
    {
        // geometries the one-wave-per-block selection does not cover go through the multi-pass kernel, which reads every
        // key several times: form the final keys once for it (and for the exact path's heapsort)
        const int Nk = c.C * c.BS, R = Nk / 64;
        const bool selWave = (Nk % 64 == 0) && (R == 4 || R == 8 || R == 16 || R == 32 || R == 64 || R == 128);
        if (!selWave) { ulcx_enc_finalize_keys(c, st); c.keyFinal = 1; }
    }
    int N = c.C * c.BS;
    int ldsEntries = ((size_t)N * 8 <= ULCX_HEAP_LDS_BYTES) ? N : 0;
    size_t heapLds = ldsEntries ? (size_t)N * 8 + (size_t)N / 8 : 0;
    if (heapLds > 48 * 1024) { CK(hipFuncSetAttribute((const void *)k_heapsel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)heapLds)); CK(hipFuncSetAttribute((const void *)k_heapsel_pipe, hipFuncAttributeMaxDynamicSharedMemorySize, (int)heapLds)); }
    int fbGrid = NB < ULCX_HEAP_GRID ? NB : ULCX_HEAP_GRID;
    // VBR: one pass.  CBR/ABR: the reference's binary search (ulcEncoder.c:98-110) needs at most
    // ceil(log2(MaxCoef))+1 probes; every block runs its own search in lock step, then one final pass.
    int probes = 0;
    if (c.mode != ULCX_MODE_VBR) {
        probes = 2; int m = N; while (m > 1) { probes++; m >>= 1; }
        // No read-back: the host always enqueues the full count and a pass whose blocks have all converged (c.cbrLive,
        // counted down on the device) returns at the top of every kernel - nothing inside the call waits for the device.
    }
    const size_t selLds = (size_t)4 * ulcx_sel_lds_words(c.BS) * sizeof(float);
    if (selLds > 48 * 1024 && selLds <= ULCX_LDS_LIMIT && (N / 64 == 128 || N / 64 == 64)) {   // (mono BlockSize 8192: 67 KB)
#define SELA(...) do { CK(hipFuncSetAttribute((const void *)k_select_wave<__VA_ARGS__, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)selLds)); \
                       CK(hipFuncSetAttribute((const void *)k_select_wave<__VA_ARGS__, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)selLds)); \
                       CK(hipFuncSetAttribute((const void *)k_select_wave<__VA_ARGS__, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)selLds)); } while (0)
        SELA(128, 0); SELA(64, 0); SELA(64, 11);
#undef SELA
    }
    const size_t selLdsPair = (size_t)ulcx_sel_lds_words(c.BS) * sizeof(float);      // one block per workgroup
    auto launch_select = [&](int fin) {
        int R = N / 64;
        const dim3 g((NB + 3) / 4), b(256);
#define SELW(...) do { if (c.selPass == 1) hipLaunchKernelGGL((k_select_wave<__VA_ARGS__, 1>), g, b, selLds, st, c, fin); \
                       else if (c.selPass == 2) hipLaunchKernelGGL((k_select_wave<__VA_ARGS__, 2>), g, b, selLds, st, c, fin); \
                       else hipLaunchKernelGGL((k_select_wave<__VA_ARGS__, 0>), g, b, selLds, st, c, fin); } while (0)
        switch (R) {
            case 128:                                        // (one wave: ~200 VGPRs, two waves per SIMD)
                if (c.C == 2 && c.selPair) {                 // stereo BlockSize 4096: a wave per channel
                    const dim3 gp(NB), bp(128);
                    if (c.selPass == 1) hipLaunchKernelGGL((k_select_pair<64, 12, 1>), gp, bp, selLdsPair, st, c, fin);
                    else if (c.selPass == 2) hipLaunchKernelGGL((k_select_pair<64, 12, 2>), gp, bp, selLdsPair, st, c, fin);
                    else hipLaunchKernelGGL((k_select_pair<64, 12, 0>), gp, bp, selLdsPair, st, c, fin);
                } else SELW(128, 0);
                return true;
            case 64: if (c.lgBS == 11) SELW(64, 11); else SELW(64, 0); return true;      // (11: stereo BlockSize 2048; with a wave per
                                                                                         //  channel its selection is 1.08 -> 1.23 ms: barriers)
            case 32: SELW(32, 0); return true;
            case 16: SELW(16, 0); return true;
            case 8:  SELW(8, 0); return true;
            case 4:  SELW(4, 0); return true;
            default: return false;
        }
#undef SELW
    };
    // wave-kernel capacities: small (ordinary blocks, high occupancy) and full (any unit of this block size)
    WaveCaps capS = { WAVE_SK, WAVE_SZ, WAVE_SN };
    WaveCaps capF = { (c.BS + 63) & ~63, ((c.BS / 2) + 63) & ~63, 4 * c.BS + 64 };
    while ((size_t)wavecaps_lds(capF) * 4 > 150 * 1024) {       // largest that 4 waves fit in LDS; beyond it k_encode_units
        capF.k = (capF.k / 2 + 63) & ~63; capF.z = (capF.z / 2 + 63) & ~63; capF.nyb = capF.nyb / 2 + 32;
    }
    bool haveFull = capF.k > capS.k;
    // rate-control probes: k_cplx has already taken every probe that is over budget for certain, so a probe keeps at most
    // ~BitBudget/4 coefficients per block: the retry of the small launch runs with medium capacities (2 workgroups per CU
    // instead of 1), and what even they cannot hold goes to k_encode_units
    WaveCaps capM = { 1024, 512, 4096 };
    const bool haveMid = capM.k < capF.k;
    if (haveFull && (size_t)wavecaps_lds(capF) * 4 > 48 * 1024) CK(hipFuncSetAttribute((const void *)k_encode_wave<false>, hipFuncAttributeMaxDynamicSharedMemorySize, wavecaps_lds(capF) * 4 + 16));
    auto launch_encode = [&](UlcxEncCtx cc, hipStream_t s2, int fin, bool ev0, bool bigFirst) -> int {
        const bool fb2 = (cc.fbMode == 2);                 // exact path: small grids that walk the list of owned blocks
        const int fbW = NB < 128 ? NB : 128;
        if (cc.useGapSums) {
            const size_t glds = nsums_lds_bytes(N, cc.C);
            if (glds > 48 * 1024) CK(hipFuncSetAttribute((const void *)k_nsums, hipFuncAttributeMaxDynamicSharedMemorySize, (int)glds));
            // the two speculative-sum kernels are independent: on the main path the tail chains run on a side stream beside the gaps
            const bool tailAside = !fb2 && side2 != nullptr && s2 == st;
            const unsigned tg = (unsigned)((nUnits + TAILS_U - 1) / TAILS_U);
            if (tailAside) {
                CK(hipEventRecord(evTail0, s2));
                CK(hipStreamWaitEvent(side2, evTail0, 0));
                hipLaunchKernelGGL(k_tails, dim3(tg), dim3(WG), 0, side2, cc, fin);
                CK(hipEventRecord(evTail1, side2));
            }
            const int nsGrid = NB < aux.nsSlots ? NB : aux.nsSlots;             // persistent: what the device holds at once
            hipLaunchKernelGGL(k_nsums, dim3(fb2 ? fbW : nsGrid), dim3(WG), glds, s2, cc, fin);
            if (ev0 && ev) CK(hipEventRecord(ev[stage++], s2));
            if (tailAside) CK(hipStreamWaitEvent(s2, evTail1, 0));
            else hipLaunchKernelGGL(k_tails, dim3(fb2 ? fbW : tg), dim3(WG), 0, s2, cc, fin);
            if (ev0 && ev) CK(hipEventRecord(ev[stage++], s2));
        } else if (ev0 && ev) { CK(hipEventRecord(ev[stage++], s2)); CK(hipEventRecord(ev[stage++], s2)); }
        if (cc.useWave) {
            int nBC = NB * cc.C;
            // early CBR probes keep ~N/2 coefficients per block: go straight to the full-size caps there
            WaveCaps first = (bigFirst && haveFull) ? capF : capS;
            bool twoPhase = haveFull && !bigFirst;
            if (bigFirst && haveFull) hipLaunchKernelGGL(k_encode_wave<false>, dim3(fb2 ? fbW : (nBC + 3) / 4), dim3(256), (size_t)wavecaps_lds(first) * 4 + 16, s2, cc, fin, first, 2);
            else hipLaunchKernelGGL(k_encode_wave<true>, dim3(fb2 ? fbW : (nBC + 3) / 4), dim3(256), (size_t)wavecaps_lds(first) * 4 + 16, s2, cc, fin, first, twoPhase ? 0 : 2);
            if (twoPhase)
            {
                // (the exact path's few blocks also retry with the medium capacities: a full-capacity workgroup needs a whole
                //  CU's LDS and would wait for the main path's kernel to drain)
                const WaveCaps capR = (haveMid && (fb2 || (probes > 0 && !fin))) ? capM : capF;
                hipLaunchKernelGGL(k_encode_wave<false>, dim3(fb2 ? fbW : ((nBC + 3) / 4 < 512 ? (nBC + 3) / 4 : 512)), dim3(256), (size_t)wavecaps_lds(capR) * 4 + 16, s2, cc, fin, capR, 1);
            }
        }
        if (ev0 && ev) CK(hipEventRecord(ev[stage++], s2));
        hipLaunchKernelGGL(k_encode_units, dim3(fb2 ? fbW : (nUnits + 63) / 64), dim3(64), 0, s2, cc, fin);
        if (ev0 && ev) CK(hipEventRecord(ev[stage++], s2));
        if (!fin && !fb2) hipLaunchKernelGGL(k_rate_step, dim3((NB + 255) / 256), dim3(256), 0, s2, cc);
        else hipLaunchKernelGGL(k_pack, dim3(fb2 ? (fbW + 3) / 4 : (NB + 3) / 4), dim3(256), 0, s2, cc, fin);
        if (ev0 && ev) CK(hipEventRecord(ev[stage++], s2));
        return ULCX_OK;
    };
    // Exact path for tie-straddle blocks (~4e-4 of all): ONE heapsort per block and call gives the full
    // ranking, from which the block finishes its own rate search / final pass by lookup.
    auto exact_sort = [&](hipStream_t s2, int lo) -> int {
        UlcxEncCtx cf = c; cf.fbMode = 2; cf.fbLo = lo; cf.fbHi = lo + c.rankSlots;
        if (ldsEntries) hipLaunchKernelGGL(k_heapsel_pipe, dim3(fbGrid), dim3(64), heapLds, s2, cf, probes > 0 ? 1 : 0);
        else hipLaunchKernelGGL(k_heapsel, dim3(fbGrid), dim3(64), heapLds, s2, cf, ldsEntries);
        return ULCX_OK;
    };
    auto exact_passes = [&](hipStream_t s2, int lo) -> int {
        UlcxEncCtx cf = c; cf.fbMode = 2; cf.fbLo = lo; cf.fbHi = lo + c.rankSlots;
        for (int p = 0; p <= probes; p++) {
            int fin = (p == probes) ? 1 : 0;
            // (one-pass calls: k_heapsel_pipe has written the kept set, and for the first group of rank slots k_cplx cleared the counter)
            const bool fromSort = (probes == 0 && ldsEntries);
            if (!fromSort) hipLaunchKernelGGL(k_keep_ranks, dim3(fbGrid), dim3(WG), 0, s2, cf, fin);
            if (cf.useWave && !(fromSort && lo == 0)) CK(hipMemsetAsync(cf.slow + NB + 1, 0, sizeof(int), s2));      // its own retry-queue counter
            int rc = launch_encode(cf, s2, fin, false, false); if (rc) return rc;
        }
        return ULCX_OK;
    };
    // (c.fbCount, c.isFb and the first pass's c.slow are cleared by k_cplx)
    // VBR (one pass): the exact path runs on a side stream next to the encode pass of all other blocks.
    // CBR/ABR: it runs after the lock-step passes (a block joins it at whatever pass it first straddles).
    // The exact path forks at the FINAL pass (a block can first straddle there) and runs beside the main path's
    // final encode: VBR has only that pass; CBR/ABR blocks replay their whole search from the ranking there.
    const bool canFork = (side != nullptr);
    for (int p = 0; p <= probes; p++) {
        int fin = (p == probes) ? 1 : 0;
        bool ev0 = (p == 0);
        const bool async_fb = canFork && fin;
        if (c.useWave && p > 0) CK(hipMemsetAsync(c.slow, 0, sizeof(int) * ((size_t)NB + 2), st));
        c.selPass = (probes > 0 && !c.keyFinal) ? (p == 0 ? 1 : 2) : 0;
        if (!launch_select(fin)) {
            hipLaunchKernelGGL(k_select, dim3(NB), dim3(WG), 0, st, c, fin);
        }
        if (ev0) MARK();
        if (async_fb) {
            CK(hipEventRecord(evFork, st));
            CK(hipStreamWaitEvent(side, evFork, 0));
            int rc = exact_sort(side, 0); if (rc) return rc;   // needs only the keys: starts right behind the select
        }
        if (p == 0) {
            if (noiseAside) { CK(hipStreamWaitEvent(st, evNoise, 0)); MARK(); MARK(); }     // (k_nbark / k_nline intervals: hidden)
            else { int rcn = launch_noise(st, ev0); if (rcn) return rcn; }
        }
        if (async_fb) {
            CK(hipEventRecord(evFork2, st));                    // the exact path's encode pass needs the noise pairs too
            CK(hipStreamWaitEvent(side, evFork2, 0));
            int rc = exact_passes(side, 0); if (rc) return rc;
            for (int lo = c.rankSlots; lo < NB; lo += c.rankSlots) {
                rc = exact_sort(side, lo); if (rc) return rc;
                rc = exact_passes(side, lo); if (rc) return rc;
            }
            CK(hipEventRecord(evJoin, side));
        }
        if (ev0) MARK();                                       // ("k_heapsel": empty interval on the main stream)
        UlcxEncCtx cm = c; cm.fbMode = 1;
        int rc = launch_encode(cm, st, fin, ev0, false); if (rc) return rc;
        if (async_fb) CK(hipStreamWaitEvent(st, evJoin, 0));
    }
    if (!canFork) {
        for (int lo = 0; lo < NB; lo += c.rankSlots) {
            int rc = exact_sort(st, lo); if (rc) return rc;
            rc = exact_passes(st, lo); if (rc) return rc;
        }
    }
    MARK();   // cbr_probe_passes (empty interval for VBR)
    if (noiseAside) { CK(hipStreamWaitEvent(st, evState, 0));                                               MARK(); }
    else { launch_state_update(c, st);                              MARK(); }
    CK(hipGetLastError());
    return ULCX_OK;
}
