/*
 * ORACLE — TEST INFRASTRUCTURE ONLY.
 *
 * orc_bench.c — the worker loop of bench.py's `cpu_baseline` leg (and nothing else): N POSIX threads, each running the
 * oracle's whole-stream encode (ulcEncoder.c:93-158 restated in orc_encoder.c) and/or decode (ulcDecoder.c:198-302,
 * orc_decoder.c) over a small set of seeded streams until a wall-clock budget is spent.  No Python in the loop: the
 * round-4 harness drove the same calls from Python threads and 64 of them delivered 13 x one thread.
 */
#define _GNU_SOURCE
#include <pthread.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include "ulc_oracle.h"

typedef struct {
    int id, mode, legs, RateHz, nChan, BS, S, nBlocks, slot;
    float p0;
    const float *pcm;              /* [S][nBlocks * BS * nChan] interleaved                         */
    const uint8_t *enc;            /* [S][nBlocks][slot]: the streams' encoded blocks (decode-only) */
    double t0, budget, tEnd;
    long long streams;
    int rc;
} orc_bench_job;

static double now_s(void) { struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec; }

static void *orc_bench_worker(void *arg) {
    orc_bench_job *j = (orc_bench_job *)arg;
    const size_t nS = (size_t)j->nBlocks * j->BS * j->nChan, nO = (size_t)j->nBlocks * j->slot;
    uint8_t *out = (uint8_t *)malloc(nO);
    int32_t *bits = (int32_t *)malloc(sizeof(int32_t) * j->nBlocks);
    float *dec = (float *)malloc(sizeof(float) * nS);
    if (!out || !bits || !dec) { j->rc = -1; free(out); free(bits); free(dec); return NULL; }
    for (int r = 0; ; r++) {
        const int s = (j->id + r) % j->S;
        const uint8_t *src = j->enc + (size_t)s * nO;
        if (j->legs & 1) {
            int rc = j->mode ? orc_encode_stream_cbr(j->RateHz, j->nChan, j->BS, j->pcm + (size_t)s * nS, j->nBlocks, j->p0, out, j->slot, bits, NULL, NULL)
                             : orc_encode_stream_vbr(j->RateHz, j->nChan, j->BS, j->pcm + (size_t)s * nS, j->nBlocks, j->p0, out, j->slot, bits, NULL, NULL);
            if (rc) { j->rc = rc; break; }
            src = out;
        }
        if (j->legs & 2) {
            if (orc_decode_stream(j->nChan, j->BS, src, j->slot, j->nBlocks, dec, NULL)) { j->rc = -2; break; }
        }
        j->streams++;
        j->tEnd = now_s();
        if (j->tEnd - j->t0 >= j->budget) break;
    }
    free(out); free(bits); free(dec);
    return NULL;
}

/* legs: bit 0 encode, bit 1 decode.  mode: 0 VBR (p0 = quality), 1 CBR (p0 = kbps).  enc: the streams already encoded
 * (read by a decode-only run; may be NULL when legs has bit 0).  Every thread works until `seconds` have passed and then
 * finishes the stream it is on.  Returns 0 and *elapsed (start to the LAST thread's end), *streamsDone (all threads). */
int orc_bench_threads(int nThreads, int mode, int legs, int RateHz, int nChan, int BS, const float *pcm, const uint8_t *enc, int S, int nBlocks,
                      int slot, float p0, double seconds, double *elapsed, long long *streamsDone) {
    if (nThreads < 1 || S < 1 || nBlocks < 1 || !(legs & 3) || (!(legs & 1) && !enc)) return -1;
    orc_bench_job *jobs = (orc_bench_job *)calloc((size_t)nThreads, sizeof(*jobs));
    pthread_t *th = (pthread_t *)calloc((size_t)nThreads, sizeof(*th));
    if (!jobs || !th) { free(jobs); free(th); return -1; }
    const double t0 = now_s();
    int started = 0, rc = 0;
    for (int i = 0; i < nThreads; i++) {
        orc_bench_job *j = &jobs[i];
        j->id = i; j->mode = mode; j->legs = legs; j->RateHz = RateHz; j->nChan = nChan; j->BS = BS; j->S = S; j->nBlocks = nBlocks; j->slot = slot;
        j->p0 = p0; j->pcm = pcm; j->enc = enc; j->t0 = t0; j->budget = seconds; j->tEnd = t0;
        if (pthread_create(&th[i], NULL, orc_bench_worker, j)) { rc = -3; break; }
        started++;
    }
    double tEnd = t0; long long n = 0;
    for (int i = 0; i < started; i++) {
        pthread_join(th[i], NULL);
        if (jobs[i].rc && !rc) rc = jobs[i].rc;
        if (jobs[i].tEnd > tEnd) tEnd = jobs[i].tEnd;
        n += jobs[i].streams;
    }
    if (elapsed) *elapsed = tEnd - t0;
    if (streamsDone) *streamsDone = n;
    free(jobs); free(th);
    return rc;
}
