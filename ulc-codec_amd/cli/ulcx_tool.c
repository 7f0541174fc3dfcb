/*
 * ulcx_tool.c — batched front-end over libulc_amd.so.
 *
 *   ulcx-tool encode OUTDIR RATE[,AvgComplexity] [-blocksize:N] [-devices:N] IN1.wav IN2.wav ...
 *   ulcx-tool decode OUTDIR [-format:PCM16|FLOAT32] [-devices:N]             IN1.ulc IN2.ulc ...
 *
 * What tools/ulcEncodeTool.c / tools/ulcDecodeTool.c of the reference do for ONE file per
 * process, done for MANY files per call: every input is one stream of the batch, all streams
 * advance K blocks per library call.  RATE follows the reference's convention
 * (ulcEncodeTool.c:38-50): negative = VBR quality, positive = CBR kbps, "kbps,complexity" = ABR.
 * Files written are byte-identical to the reference tools' (tests/test_gpu_dropin.py):
 * container layout tools/ulc_Helper.h:10-20, block count ulcEncodeTool.c:93-98 (+2 blocks of
 * coding/MDCT delay), sample conversion WavIO_Helper.c:49-63 (x 2^-15 in, lrintf(clamp(x 2^15)) out).
 * All inputs of one call must share rate / channel count (encode) or rate / channels / block size
 * (decode); inputs may have different lengths (shorter ones are padded with silence and trimmed
 * to their own block count on output).
 *
 * -devices:N (independent streams shard by plain batch split, one host thread per device, no collective): the
 * inputs are dealt round-robin over N groups, every group gets its own encoder / decoder and its own host thread; group g
 * runs on device g % (visible devices), so N may exceed the device count (two groups then share a GPU).  The files written
 * do not depend on N.
 *
 * WAV support is deliberately minimal: RIFF/WAVE, "fmt " PCM 16-bit or IEEE float 32-bit, one
 * "data" chunk.  Host code is plain C over the C ABI of include/ulc_amd.h.
 */
#include <math.h>
#include <pthread.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "../../include/ulc_amd.h"

#define KBLOCKS 16                      /* blocks per stream per library call */

struct wav { int rate, chan, bits, isFloat; uint32_t nFrames; long dataOffs; FILE *f; };

static uint32_t rd32(const uint8_t *p) { return p[0] | p[1] << 8 | p[2] << 16 | (uint32_t)p[3] << 24; }
static uint16_t rd16(const uint8_t *p) { return (uint16_t)(p[0] | p[1] << 8); }

static int wav_open(struct wav *w, const char *path) {
    uint8_t h[12], ck[8], fmt[40];
    memset(w, 0, sizeof(*w));
    w->f = fopen(path, "rb");
    if (!w->f) return -1;
    if (fread(h, 1, 12, w->f) != 12 || memcmp(h, "RIFF", 4) || memcmp(h + 8, "WAVE", 4)) return -2;
    int haveFmt = 0;
    for (;;) {
        if (fread(ck, 1, 8, w->f) != 8) return -3;
        uint32_t sz = rd32(ck + 4);
        if (!memcmp(ck, "fmt ", 4)) {
            uint32_t n = sz < sizeof(fmt) ? sz : (uint32_t)sizeof(fmt);
            if (sz < 16 || fread(fmt, 1, n, w->f) != n) return -4;
            if (sz > n) fseek(w->f, (long)(sz - n), SEEK_CUR);
            int tag = rd16(fmt);
            if (tag == 0xFFFE && sz >= 26) tag = rd16(fmt + 24);          /* WAVE_FORMAT_EXTENSIBLE: sub-format */
            w->chan = rd16(fmt + 2); w->rate = (int)rd32(fmt + 4); w->bits = rd16(fmt + 14);
            w->isFloat = (tag == 3);
            if (!((tag == 1 && w->bits == 16) || (tag == 3 && w->bits == 32))) return -5;
            haveFmt = 1;
        } else if (!memcmp(ck, "data", 4)) {
            if (!haveFmt) return -6;
            w->dataOffs = ftell(w->f);
            w->nFrames = sz / (uint32_t)(w->chan * w->bits / 8);
            return 0;
        } else fseek(w->f, (long)(sz + (sz & 1)), SEEK_CUR);
    }
}
/* frames [pos, pos+n) as float, zero padded past the end (WavIO_Reader.c:115-150) */
static void wav_read(struct wav *w, uint32_t pos, uint32_t n, float *dst, void *tmp) {
    uint32_t have = pos < w->nFrames ? w->nFrames - pos : 0;
    if (have > n) have = n;
    size_t fb = (size_t)w->chan * w->bits / 8;
    if (have) {
        fseek(w->f, w->dataOffs + (long)(pos * fb), SEEK_SET);
        size_t got = fread(tmp, fb, have, w->f);
        if (got < have) have = (uint32_t)got;
    }
    size_t ns = (size_t)have * w->chan;
    if (w->isFloat) memcpy(dst, tmp, ns * 4);
    else { const int16_t *s = (const int16_t *)tmp; for (size_t i = 0; i < ns; i++) dst[i] = (float)s[i] * 0x1.0p-15f; }
    for (size_t i = ns; i < (size_t)n * w->chan; i++) dst[i] = 0.0f;
}
static void wav_write_header(FILE *f, int rate, int chan, int isFloat, uint32_t nFrames) {
    int bits = isFloat ? 32 : 16;
    uint32_t dataBytes = nFrames * (uint32_t)(chan * bits / 8);
    uint8_t h[44] = { 'R','I','F','F', 0,0,0,0, 'W','A','V','E', 'f','m','t',' ', 16,0,0,0 };
    uint32_t riff = 36 + dataBytes, bps = (uint32_t)(rate * chan * bits / 8);
    memcpy(h + 4, &riff, 4);
    h[20] = isFloat ? 3 : 1; h[22] = (uint8_t)chan; h[23] = (uint8_t)(chan >> 8);
    memcpy(h + 24, &rate, 4); memcpy(h + 28, &bps, 4);
    h[32] = (uint8_t)(chan * bits / 8); h[34] = (uint8_t)bits;
    memcpy(h + 36, "data", 4); memcpy(h + 40, &dataBytes, 4);
    fwrite(h, 1, 44, f);
}
static const char *base_name(const char *p) { const char *s = strrchr(p, '/'); return s ? s + 1 : p; }
static void out_path(char *dst, size_t n, const char *dir, const char *in, const char *ext) {
    char stem[512];
    snprintf(stem, sizeof(stem), "%s", base_name(in));
    char *dot = strrchr(stem, '.'); if (dot) *dot = 0;
    snprintf(dst, n, "%s/%s%s", dir, stem, ext);
}
#define DIE(...) do { fprintf(stderr, "ulcx-tool: " __VA_ARGS__); fprintf(stderr, "\n"); return 2; } while (0)

/* one group of inputs = one batch on one device (the whole command line, or a -devices:N share of it on its own thread) */
struct group { int decode, device, n; char **files; const char *outdir; float rate, avgc; int bs, isFloat; int rc; };

static int encode_group(const struct group *g) {
    const char *outdir = g->outdir;
    const float rate = g->rate, avgc = g->avgc;
    const int bs = g->bs, B = g->n, a = 0;
    char **argv = g->files;
    struct wav *w = (struct wav *)calloc((size_t)B, sizeof(*w));
    uint32_t maxBlk = 0;
    for (int s = 0; s < B; s++) {
        int e = wav_open(&w[s], argv[a + s]);
        if (e) DIE("cannot read '%s' (error %d: RIFF PCM16 / float32 only)", argv[a + s], e);
        if (w[s].rate != w[0].rate || w[s].chan != w[0].chan) DIE("'%s': all inputs of one call must share rate and channel count", argv[a + s]);
        uint32_t nb = (w[s].nFrames + (uint32_t)bs - 1) / (uint32_t)bs + 2;      /* ulcEncodeTool.c:93-98 */
        if (nb > maxBlk) maxBlk = nb;
    }
    const int C = w[0].chan, hz = w[0].rate;
    int mode = rate < 0.0f ? ULCX_MODE_VBR : (avgc > 0.0f ? ULCX_MODE_ABR : ULCX_MODE_CBR);
    float p0 = rate < 0.0f ? -rate : rate;
    ulcx_encoder *enc = NULL;
    if (ulcx_encoder_create(&enc, g->device, B, C, bs, hz, KBLOCKS) != ULCX_OK) DIE("encoder: %s", ulcx_last_error());
    const int slot = ulcx_encoder_slot_bytes(enc);
    size_t frame = (size_t)bs * C;
    float *pcm = (float *)malloc(sizeof(float) * (size_t)B * KBLOCKS * frame);
    uint8_t *out = (uint8_t *)malloc((size_t)B * KBLOCKS * slot);
    int32_t *bits = (int32_t *)malloc(sizeof(int32_t) * (size_t)B * KBLOCKS);
    float *cplx = (float *)malloc(sizeof(float) * (size_t)B * KBLOCKS);
    double *cplxSum = (double *)calloc((size_t)B, sizeof(double));   /* ulcEncodeTool.c:130,164: feeds the ABR workflow */
    void *tmp = malloc(frame * 4 * KBLOCKS);
    FILE **fo = (FILE **)calloc((size_t)B, sizeof(FILE *));
    uint64_t *total = (uint64_t *)calloc((size_t)B, sizeof(uint64_t));
    uint32_t *maxb = (uint32_t *)calloc((size_t)B, sizeof(uint32_t));
    char path[1024];
    for (int s = 0; s < B; s++) {
        out_path(path, sizeof(path), outdir, argv[a + s], ".ulc");
        fo[s] = fopen(path, "wb");
        if (!fo[s]) DIE("cannot create '%s'", path);
        fseek(fo[s], 24, SEEK_SET);
    }
    for (uint32_t k0 = 0; k0 < maxBlk; k0 += KBLOCKS) {
        int K = (maxBlk - k0 < KBLOCKS) ? (int)(maxBlk - k0) : KBLOCKS;
        for (int s = 0; s < B; s++)
            wav_read(&w[s], k0 * (uint32_t)bs, (uint32_t)(K * bs), pcm + (size_t)s * K * frame, tmp);
        if (ulcx_encode_host(enc, mode, p0, avgc, pcm, K, out, bits, NULL, cplx) != ULCX_OK) DIE("encode: %s", ulcx_last_error());
        for (int s = 0; s < B; s++) {
            uint32_t nb = (w[s].nFrames + (uint32_t)bs - 1) / (uint32_t)bs + 2;
            for (int k = 0; k < K && k0 + (uint32_t)k < nb; k++) {
                uint32_t sz = (uint32_t)(bits[s * K + k] + 7) / 8u;
                fwrite(out + ((size_t)s * K + k) * slot, 1, sz, fo[s]);               /* ulcEncodeTool.c:160-169 */
                total[s] += sz; if (sz > maxb[s]) maxb[s] = sz;
                cplxSum[s] += cplx[s * K + k];
            }
        }
    }
    for (int s = 0; s < B; s++) {
        uint32_t nb = (w[s].nFrames + (uint32_t)bs - 1) / (uint32_t)bs + 2;
        ulcx_file_header h;
        h.Magic = ULCX_ULC_MAGIC; h.BlockSize = (uint16_t)bs; h.MaxBlockSize = (uint16_t)maxb[s]; h.nBlocks = nb;
        h.RateHz = (uint32_t)hz; h.nChan = (uint16_t)C; h.StreamOffs = 24;
        h.RateKbps = (uint16_t)ulcx_ulc_rate_kbps(total[s], (uint32_t)hz, (uint32_t)bs, nb);
        uint8_t hb[24]; ulcx_ulc_header_pack(hb, &h);
        fseek(fo[s], 0, SEEK_SET); fwrite(hb, 1, 24, fo[s]); fclose(fo[s]); fclose(w[s].f);
        printf("%s: %u blocks, %.2f KiB, %u kbps, avg complexity %.5f\n", base_name(argv[a + s]), nb, total[s] / 1024.0, h.RateKbps,
               cplxSum[s] / nb);                                                      /* ulcEncodeTool.c:176,186 */
    }
    ulcx_encoder_destroy(enc);
    free(pcm); free(out); free(bits); free(cplx); free(cplxSum); free(tmp); free(fo); free(total); free(maxb); free(w);
    return 0;
}

static int decode_group(const struct group *g) {
    const char *outdir = g->outdir;
    const int isFloat = g->isFloat, B = g->n, a = 0;
    char **argv = g->files;
    ulcx_file_header *h = (ulcx_file_header *)calloc((size_t)B, sizeof(*h));
    uint8_t **pay = (uint8_t **)calloc((size_t)B, sizeof(uint8_t *));
    int32_t *payBytes = (int32_t *)calloc((size_t)B, sizeof(int32_t));
    long long stride = 0;
    uint32_t maxBlk = 0;
    for (int s = 0; s < B; s++) {
        FILE *f = fopen(argv[a + s], "rb");
        if (!f) DIE("cannot open '%s'", argv[a + s]);
        fseek(f, 0, SEEK_END); long len = ftell(f); fseek(f, 0, SEEK_SET);
        uint8_t *buf = (uint8_t *)malloc((size_t)len + 16);
        if (fread(buf, 1, (size_t)len, f) != (size_t)len) DIE("short read on '%s'", argv[a + s]);
        fclose(f);
        if (ulcx_ulc_header_parse(&h[s], buf, (size_t)len)) DIE("'%s' is not a ULC2 container", argv[a + s]);
        /* the header is untrusted input: validate it before anything is sized or indexed from it */
        if (h[s].StreamOffs < 24 || (long)h[s].StreamOffs > len) DIE("'%s': stream offset %u outside the file (%ld bytes)", argv[a + s], h[s].StreamOffs, len);
        if (h[s].nChan < 1 || h[s].nChan > 255 || h[s].BlockSize < 256 || h[s].BlockSize > 32768 || (h[s].BlockSize & (h[s].BlockSize - 1)))
            DIE("'%s': invalid geometry in the header (BlockSize %u, %u channels)", argv[a + s], h[s].BlockSize, h[s].nChan);
        if (h[s].BlockSize != h[0].BlockSize || h[s].nChan != h[0].nChan || h[s].RateHz != h[0].RateHz)
            DIE("'%s': all inputs of one call must share block size, channels and rate", argv[a + s]);
        /* block count and payload size too: a block is at least two bytes (window nybble + one code per channel), so a
         * header that counts more blocks than the payload can hold is corrupt, and so is a payload the 32-bit read
         * positions of the library cannot address; the WAV header's sample count is checked in 64 bits */
        if (len - (long)h[s].StreamOffs > 0x7fffffffL) DIE("'%s': payload of %ld bytes is more than this tool takes (2 GiB)", argv[a + s], len - (long)h[s].StreamOffs);
        if ((uint64_t)h[s].nBlocks > (uint64_t)(len - (long)h[s].StreamOffs) / 2) DIE("'%s': header counts %u blocks, the payload has %ld bytes", argv[a + s], h[s].nBlocks, len - (long)h[s].StreamOffs);
        if ((uint64_t)h[s].nBlocks * h[s].BlockSize * h[s].nChan * 4 > 0xfffff000ull) DIE("'%s': %u blocks decode to more than a WAV file holds", argv[a + s], h[s].nBlocks);
        pay[s] = buf; payBytes[s] = (int32_t)(len - (long)h[s].StreamOffs);
        if (payBytes[s] + 8 > stride) stride = payBytes[s] + 8;
        if (h[s].nBlocks > maxBlk) maxBlk = h[s].nBlocks;
    }
    const int bs = h[0].BlockSize, C = h[0].nChan;
    stride = (stride + 15) & ~15LL;
    uint8_t *payload = (uint8_t *)calloc((size_t)B, (size_t)stride);
    for (int s = 0; s < B; s++) { memcpy(payload + (size_t)s * stride, pay[s] + h[s].StreamOffs, (size_t)payBytes[s]); free(pay[s]); }
    ulcx_decoder *dec = NULL;
    if (ulcx_decoder_create(&dec, g->device, B, C, bs, KBLOCKS) != ULCX_OK) DIE("decoder: %s", ulcx_last_error());
    size_t frame = (size_t)bs * C;
    float *pcm = (float *)malloc(sizeof(float) * (size_t)B * KBLOCKS * frame);
    int32_t *bits = (int32_t *)malloc(sizeof(int32_t) * (size_t)B * KBLOCKS);
    int16_t *tmp = (int16_t *)malloc(sizeof(int16_t) * frame);
    FILE **fo = (FILE **)calloc((size_t)B, sizeof(FILE *));
    char path[1024];
    for (int s = 0; s < B; s++) {
        out_path(path, sizeof(path), outdir, argv[a + s], ".wav");
        fo[s] = fopen(path, "wb");
        if (!fo[s]) DIE("cannot create '%s'", path);
        wav_write_header(fo[s], (int)h[s].RateHz, C, isFloat, h[s].nBlocks * (uint32_t)bs);
    }
    if (ulcx_decoder_upload_payload(dec, payload, stride, payBytes) != ULCX_OK) DIE("upload: %s", ulcx_last_error());   /* once, not per call */
    int rcAll = 0;
    for (uint32_t k0 = 0; k0 < maxBlk; k0 += KBLOCKS) {
        int K = (maxBlk - k0 < KBLOCKS) ? (int)(maxBlk - k0) : KBLOCKS;
        if (ulcx_decode_resident_host(dec, K, pcm, bits) != ULCX_OK) DIE("decode: %s", ulcx_last_error());
        for (int s = 0; s < B; s++)
            for (int k = 0; k < K && k0 + (uint32_t)k < h[s].nBlocks; k++) {
                if (!bits[s * K + k]) { fprintf(stderr, "ulcx-tool: %s: corrupted stream at block %u\n", argv[a + s], k0 + (uint32_t)k); rcAll = 1; }
                const float *src = pcm + ((size_t)s * K + k) * frame;
                if (isFloat) fwrite(src, 4, frame, fo[s]);
                else {
                    for (size_t i = 0; i < frame; i++) {                                  /* WavIO_Helper.c:57-63 */
                        float v = src[i] * 0x1.0p+15f;
                        v = v < -32768.0f ? -32768.0f : (v > 32767.0f ? 32767.0f : v);
                        tmp[i] = (int16_t)lrintf(v);
                    }
                    fwrite(tmp, 2, frame, fo[s]);
                }
            }
    }
    for (int s = 0; s < B; s++) fclose(fo[s]);
    ulcx_decoder_destroy(dec);
    free(payload); free(pcm); free(bits); free(tmp); free(fo); free(h); free(pay); free(payBytes);
    return rcAll;
}

static void *group_main(void *p) {
    struct group *g = (struct group *)p;
    g->rc = g->decode ? decode_group(g) : encode_group(g);
    return NULL;
}
/* the command line's inputs as nDev groups (input i goes to group i % nDev), one host thread and one codec object each */
static int run_groups(struct group *proto, int nFiles, char **files, int nDev) {
    if (nFiles < 1) DIE("no input files");
    if (nDev > nFiles) nDev = nFiles;
    const int have = ulcx_device_count();
    if (have < 1) DIE("no HIP device: %s", ulcx_last_error());
    if (nDev <= 1) { proto->device = 0; proto->n = nFiles; proto->files = files; return proto->decode ? decode_group(proto) : encode_group(proto); }
    /* what a single group checks per batch - every input of one call shares rate / channels (encode) or block size /
     * channels / rate (decode) - is checked here ONCE over all files, before they are dealt out: a command line that a
     * one-group run rejects must not be partly accepted with -devices:N */
    {
        uint32_t ref[3] = { 0, 0, 0 };
        for (int i = 0; i < nFiles; i++) {
            uint32_t cur[3] = { 0, 0, 0 };
            if (!proto->decode) {
                struct wav w;
                const int e = wav_open(&w, files[i]);
                if (e) DIE("cannot read '%s' (error %d: RIFF PCM16 / float32 only)", files[i], e);
                cur[0] = w.rate; cur[1] = (uint32_t)w.chan; fclose(w.f);
            } else {
                uint8_t hb[24]; ulcx_file_header h;
                FILE *f = fopen(files[i], "rb");
                if (!f) DIE("cannot open '%s'", files[i]);
                const size_t got = fread(hb, 1, sizeof(hb), f); fclose(f);
                if (got != sizeof(hb) || ulcx_ulc_header_parse(&h, hb, sizeof(hb))) DIE("'%s' is not a ULC2 container", files[i]);
                cur[0] = h.RateHz; cur[1] = h.nChan; cur[2] = h.BlockSize;
            }
            if (i == 0) memcpy(ref, cur, sizeof(ref));
            else if (memcmp(ref, cur, sizeof(ref))) DIE("'%s': all inputs of one call must share %s", files[i], proto->decode ? "block size, channels and rate" : "rate and channel count");
        }
    }
    struct group *gs = (struct group *)calloc((size_t)nDev, sizeof(*gs));
    char **deal = (char **)calloc((size_t)nFiles, sizeof(char *));
    pthread_t *th = (pthread_t *)calloc((size_t)nDev, sizeof(pthread_t));
    if (!gs || !deal || !th) { free(gs); free(deal); free(th); DIE("out of memory"); }
    int at = 0, rc = 0, started = 0;
    for (int g = 0; g < nDev; g++) {
        gs[g] = *proto; gs[g].device = g % have; gs[g].files = deal + at; gs[g].n = 0; gs[g].rc = 0;
        for (int i = g; i < nFiles; i += nDev) deal[at + gs[g].n++] = files[i];
        at += gs[g].n;
    }
    for (int g = 0; g < nDev; g++) {
        if (pthread_create(&th[g], NULL, group_main, &gs[g])) { fprintf(stderr, "ulcx-tool: cannot start a host thread for group %d\n", g); rc = 2; break; }
        started++;
    }
    /* (a failed start: the groups already running work on gs / deal - they are joined before anything is freed) */
    for (int g = 0; g < started; g++) { pthread_join(th[g], NULL); if (gs[g].rc > rc) rc = gs[g].rc; }
    free(gs); free(deal); free(th);
    return rc;
}
static int do_encode(int argc, char **argv) {
    if (argc < 5) DIE("usage: ulcx-tool encode OUTDIR RATE[,AvgComplexity] [-blocksize:N] [-devices:N] IN.wav ...");
    struct group g; memset(&g, 0, sizeof(g));
    g.outdir = argv[2];
    sscanf(argv[3], "%f,%f", &g.rate, &g.avgc);
    if (g.rate == 0.0f || g.avgc < 0.0f) DIE("invalid coding rate '%s'", argv[3]);
    int a = 4, nDev = 1;
    g.bs = 2048;
    for (; a < argc && argv[a][0] == '-'; a++) {
        if (!strcmp(argv[a], "--")) { a++; break; }          /* end of options: input names may start with '-' behind it */
        if (!strncmp(argv[a], "-blocksize:", 11)) g.bs = atoi(argv[a] + 11);
        else if (!strncmp(argv[a], "-devices:", 9)) nDev = atoi(argv[a] + 9);
        else DIE("unknown option '%s'", argv[a]);
    }
    if (g.bs < 256 || g.bs > 8192 || (g.bs & -g.bs) != g.bs) DIE("unsupported block size %d", g.bs);
    if (nDev < 1 || nDev > 64) DIE("-devices:%d out of range", nDev);
    return run_groups(&g, argc - a, argv + a, nDev);
}
static int do_decode(int argc, char **argv) {
    if (argc < 4) DIE("usage: ulcx-tool decode OUTDIR [-format:PCM16|FLOAT32] [-devices:N] IN.ulc ...");
    struct group g; memset(&g, 0, sizeof(g));
    g.decode = 1; g.outdir = argv[2];
    int a = 3, nDev = 1;
    for (; a < argc && argv[a][0] == '-'; a++) {
        if (!strcmp(argv[a], "--")) { a++; break; }
        if (!strncmp(argv[a], "-format:", 8)) {
            const char *f = argv[a] + 8;
            if (!strcmp(f, "FLOAT32") || !strcmp(f, "float32")) g.isFloat = 1;
            else if (strcmp(f, "PCM16") && strcmp(f, "pcm16")) DIE("unsupported output format '%s'", f);
        } else if (!strncmp(argv[a], "-devices:", 9)) nDev = atoi(argv[a] + 9);
        else DIE("unknown option '%s'", argv[a]);
    }
    if (nDev < 1 || nDev > 64) DIE("-devices:%d out of range", nDev);
    return run_groups(&g, argc - a, argv + a, nDev);
}

int main(int argc, char **argv) {
    if (argc >= 2 && !strcmp(argv[1], "encode")) return do_encode(argc, argv);
    if (argc >= 2 && !strcmp(argv[1], "decode")) return do_decode(argc, argv);
    fprintf(stderr,
            "ulcx-tool - batched ulc-codec front-end over libulc_amd.so (MI355X)\n"
            "  ulcx-tool encode OUTDIR RATE[,AvgComplexity] [-blocksize:N] [-devices:N] IN1.wav IN2.wav ...\n"
            "      RATE < 0: VBR quality; RATE > 0: CBR kbps; RATE,AvgComplexity: ABR  (as ulcencodetool)\n"
            "  ulcx-tool decode OUTDIR [-format:PCM16|FLOAT32] [-devices:N] IN1.ulc IN2.ulc ...\n"
            "  --          end of options (input names that start with '-')\n"
            "  -devices:N  inputs dealt round-robin over N groups, one host thread + one codec object each (device g %% visible)\n");
    return 1;
}
