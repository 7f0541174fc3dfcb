// ulcx_tables.cpp — host construction of the data-independent tables the kernels read
// (SURVEY.md Appendix C.6).  Everything here is evaluated ONCE per encoder/decoder with
// the HOST libm, exactly where the reference evaluates it per block with the same
// arguments:
//   Bark band edges / per-line Bark position : libulc/ulcEncoder_Psyopt.c:109-116,141-143,198-205,237-239
//                                              via libulc/ulcHelper.h:96-120
//   DCT-IV / FFT twiddles, sine-window ramps : "fourier spec v2" (FormatSpecs.md:155)
// so that the device never needs sinhf/asinhf/cos/sin.
#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include "ulcx_internal.h"

static thread_local char g_err[512] = "";
void ulcx_set_error(const char *fmt, ...) {
    va_list ap; va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
extern "C" const char *ulcx_last_error(void) { return g_err; }

// ulcHelper.h:96-120 (binary32 expressions, same operation order)
static float freq_to_line(float hz, float nyq, uint32_t n) { return (hz * (float)n / nyq) - 0.5f; }
static float line_to_freq(uint32_t line, float nyq, uint32_t n) { return ((float)line + 0.5f) * nyq / (float)n; }
static float freq_to_bark(float hz) { return 6.0f * asinhf(hz * (1.0f / 600.0f)); }
static float bark_to_freq(float bark) { return 600.0f * sinhf(bark * (1.0f / 6.0f)); }

static void band_edges(short *beg, short *end, int nLines, float nyq, float offLo, float offHi) {
    for (int b = 0; b < ULCX_NBARK; b++) {
        float f0 = bark_to_freq((float)b + offLo);
        float f1 = bark_to_freq((float)b + offHi);
        int l0 = (int)floorf(freq_to_line(f0, nyq, (uint32_t)nLines));
        int l1 = (int)ceilf(freq_to_line(f1, nyq, (uint32_t)nLines));
        if (l0 < 0) l0 = 0;
        if (l1 < 0) l1 = 0;
        if (l0 > nLines - 1) l0 = nLines - 1;
        if (l1 > nLines) l1 = nLines;
        beg[b] = (short)l0; end[b] = (short)l1;
    }
}

#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif

int ulcx_tables_build(UlcxTables *T, void **devBlob, int BS, int rateHz, bool forEncoder) {
    memset(T, 0, sizeof(*T));
    std::vector<unsigned char> blob;
    auto reserve = [&](size_t bytes) { size_t off = (blob.size() + 255) & ~(size_t)255; blob.resize(off + bytes); return off; };
    size_t offPre[ULCX_MAX_SUB], offTw[ULCX_MAX_SUB], offIdx[ULCX_MAX_SUB], offFrac[ULCX_MAX_SUB], offW[ULCX_MAX_SUB] = {0};
    float nyq = (float)rateHz * 0.5f;
    for (int d = 0; d < ULCX_MAX_SUB; d++) {
        int S = BS >> d, M = S / 2;
        offPre[d] = reserve(sizeof(float2) * M);
        offTw[d]  = reserve(sizeof(float2) * (M / 2 + 1));
        float2 *pre = (float2 *)(blob.data() + offPre[d]);
        for (int n = 0; n < M; n++) {
            double th = M_PI * (double)(8 * n + 1) / (double)(8 * S);
            pre[n].x = (float)cos(th); pre[n].y = (float)sin(th);
        }
        float2 *tw = (float2 *)(blob.data() + offTw[d]);
        for (int j = 0; j < M / 2; j++) {
            double th = 2.0 * M_PI * (double)j / (double)M;
            tw[j].x = (float)cos(th); tw[j].y = (float)sin(th);
        }
        if (forEncoder) {
            int nLines = S / 2;
            offIdx[d]  = reserve(sizeof(int) * nLines);
            offFrac[d] = reserve(sizeof(float) * nLines);
            int *bi = (int *)(blob.data() + offIdx[d]);
            float *fr = (float *)(blob.data() + offFrac[d]);
            for (int line = 0; line < nLines; line++) {
                float bb = freq_to_bark(line_to_freq((uint32_t)line, nyq, (uint32_t)nLines));
                int i = (int)bb;
                bi[line] = i;
                fr[line] = bb - (float)i;
            }
            {
                offW[d] = reserve(sizeof(float) * 4 * nLines);
                float *w0 = (float *)(blob.data() + offW[d]);
                bi = (int *)(blob.data() + offIdx[d]); fr = (float *)(blob.data() + offFrac[d]);      // (reserve() may have moved the blob)
                for (int line = 0; line < nLines; line++) {
                    const int iL = bi[line] < ULCX_NBARK ? bi[line] : ULCX_NBARK - 1;                   // Psyopt.c:143-147
                    const int iR = bi[line] + 1 < ULCX_NBARK ? bi[line] + 1 : iL;
                    memcpy(&w0[4 * line + 0], &iL, 4); memcpy(&w0[4 * line + 1], &iR, 4);
                    w0[4 * line + 2] = 1.0f - fr[line]; w0[4 * line + 3] = fr[line];
                }
            }
            band_edges(T->nBeg[d], T->nEnd[d], nLines, nyq, 0.0f, 2.0f);
            band_edges(T->pBeg[d], T->pEnd[d], nLines, nyq, -0.75f, 0.25f);
        }
    }
    // window ramps: overlap Ov = 2^j stored at [Ov, 2 Ov)
    size_t offFall = reserve(sizeof(float) * 2 * BS), offRise = reserve(sizeof(float) * 2 * BS);
    {
        float *fall = (float *)(blob.data() + offFall), *rise = (float *)(blob.data() + offRise);
        for (int ov = 1; ov <= BS; ov <<= 1)
            for (int i = 0; i < ov; i++) {
                double th = M_PI * (double)(2 * i + 1) / (double)(4 * ov);
                fall[ov + i] = (float)cos(th);
                rise[ov + i] = (float)sin(th);
            }
    }
    // k_bark_uniform: the band edges of a subblock size as one list in line order, [noise|psycho][d][ULCX_BARK_EVENTS] words
    // line | kind << 16 | band << 24 (kind 0: lower edge, 1: upper edge, 2: end of the subblock); at equal lines lower
    // edges come first (a band of no lines opens, then closes)
    size_t offSched = reserve(sizeof(uint32_t) * 2 * ULCX_MAX_SUB * ULCX_BARK_EVENTS);
    if (forEncoder) {
        uint32_t *sch = (uint32_t *)(blob.data() + offSched);
        for (int t = 0; t < 2; t++) for (int d = 0; d < ULCX_MAX_SUB; d++) {
            const short *beg = t ? T->pBeg[d] : T->nBeg[d], *end = t ? T->pEnd[d] : T->nEnd[d];
            uint32_t *o = sch + (size_t)(t * ULCX_MAX_SUB + d) * ULCX_BARK_EVENTS;
            int n = 0, bo = 0, bc = 0, N = (BS >> d) / 2;
            bool ordered = true;
            for (int b = 0; b < ULCX_NBARK; b++) {
                if (beg[b] > end[b] || end[b] > N) ordered = false;
                if (b && (beg[b] < beg[b - 1] || end[b] < end[b - 1])) ordered = false;
            }
            if (ordered) while (bc < ULCX_NBARK) {
                int pos = (bo < ULCX_NBARK && beg[bo] <= end[bc]) ? beg[bo] : end[bc];
                while (bo < ULCX_NBARK && beg[bo] == pos) { o[n++] = (uint32_t)pos | (0u << 16) | ((uint32_t)bo << 24); bo++; }
                while (bc < bo && end[bc] == pos) { o[n++] = (uint32_t)pos | (1u << 16) | ((uint32_t)bc << 24); bc++; }
            }
            o[n++] = (uint32_t)N | (2u << 16);
            while (n < ULCX_BARK_EVENTS) o[n++] = 0xffffu | (3u << 16);
        }
    }
    void *dev = nullptr;
    hipError_t e = hipMalloc(&dev, blob.size());
    if (e != hipSuccess) { ulcx_set_error("hipMalloc(tables): %s", hipGetErrorString(e)); return ULCX_ERR_NOMEM; }
    e = hipMemcpy(dev, blob.data(), blob.size(), hipMemcpyHostToDevice);
    if (e != hipSuccess) { ulcx_set_error("hipMemcpy(tables): %s", hipGetErrorString(e)); hipFree(dev); return ULCX_ERR_HIP; }
    unsigned char *base = (unsigned char *)dev;
    for (int d = 0; d < ULCX_MAX_SUB; d++) {
        T->pre[d] = (const float2 *)(base + offPre[d]);
        T->tw[d]  = (const float2 *)(base + offTw[d]);
        if (forEncoder) {
            T->bandIdx[d]  = (const int *)(base + offIdx[d]);
            T->bandFrac[d] = (const float *)(base + offFrac[d]);
            T->bandW[d]    = (const float4 *)(base + offW[d]);
        }
    }
    T->barkSched = (const uint32_t *)(base + offSched);
    T->winFall = (const float *)(base + offFall);
    T->winRise = (const float *)(base + offRise);
    *devBlob = dev;
    return ULCX_OK;
}
