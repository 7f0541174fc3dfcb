/*
 * ORACLE — TEST INFRASTRUCTURE ONLY (see ulc_oracle.h for the pinning status).
 *
 * orc_decoder.c — CPU restatement of /root/reference/libulc/ulcDecoder.c, checked
 * against the normative syntax in /root/reference/FormatSpecs.md.
 *
 * One deliberate, documented difference: the reference keeps the noise RNG seed in
 * a function-static shared by every decoder in the process (ulcDecoder.c:75-81).
 * The tools run one decoder per process, so "stream == fresh seed 1234567" is the
 * observable behaviour; here (and in the batched product) the seed is per state.
 * orc_decode_stream_seeded hands the state in and out for the one case where the
 * difference shows: several files decoded by one process through the drop-in ABI.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include "ulc_oracle.h"

/* ulcDecoder.c:75-81 */
uint32_t orc_xorshift32(uint32_t s) {
    s ^= s << 13;
    s ^= s >> 17;
    s ^= s << 5;
    return s;
}

typedef struct { const uint8_t *p; int size; } nyb_reader;

/* ulcDecoder.c:82-88 — low nybble first */
static unsigned get_nybble(nyb_reader *r) {
    unsigned x = *r->p;
    r->size += 4;
    if ((unsigned)r->size % 8u == 0) { x >>= 4; r->p++; }
    return x & 0xF;
}

#define ESC_STOP       (-1)
#define ESC_STOP_NOISE (-2)
/* ulcDecoder.c:89-95 */
static int get_quantizer(nyb_reader *r) {
    int q = (int)get_nybble(r);
    if (q == 0xF) return ESC_STOP_NOISE;
    if (q == 0xE) q += (int)get_nybble(r);
    if (q == 0xE + 0xF) return ESC_STOP;
    return q;
}
/* ulcDecoder.c:96-98.  q is -2 when a unit OPENS with Fh (get_quantizer's ESC_STOP_NOISE reaches here unchecked,
 * ulcDecoder.c:103-112; no encoder writes that and FormatSpecs.md allocates no such code): the reference then shifts by a
 * negative count - undefined in C, and on x86-64 `shr` takes the count modulo 32, i.e. 30, which leaves 0: the unit's quantizer
 * is exactly 0.0 until a change code.  That is the behaviour of the reference BINARY on x86, not a rule of the format; it is
 * written out here (count & 31) instead of being inherited from the compiler, and the device decoder does the same explicitly
 * (csrc/ulcx_dec.hip: index 30).  tests/test_oracle_codec.py::test_opening_Fh_quantizer_is_x86_shift_behaviour. */
static float expand_quantizer(int q) { return 0x1.0p-31f * ((1u << (31 - 5)) >> ((unsigned)q & 31u)); }

/* ulcDecoder.c:99-197 */
static int decode_subblock(orc_decoder *st, float *dst, int N, nyb_reader *r) {
    int32_t n, v;
    v = get_quantizer(r);
    if (v == ESC_STOP) { do *dst++ = 0.0f; while (--N); return 1; }
    float quant = expand_quantizer(v);
    for (;;) {
        v = (int32_t)get_nybble(r);
        if (v != 0x0 && v != 0x1 && v != 0x8 && v != 0xF) {                /* :115-122 */
            v = (v ^ 0x8) - 0x8;
            v = (v < 0) ? (-v * v) : (+v * v);
            *dst++ = v * quant;
            if (--N == 0) break;
            continue;
        }
        if (v == 0x0) {                                                    /* :125-132 */
            n = (int32_t)get_nybble(r) + 1;
            if (n > N) return 0;
            N -= n;
            do *dst++ = 0.0f; while (--n);
            if (N == 0) break;
            continue;
        }
        if (v == 0x1) {                                                    /* :135-144 */
            n = (int32_t)get_nybble(r);
            n = (int32_t)get_nybble(r) | (n << 4);
            n += 33;
            if (n > N) return 0;
            N -= n;
            do *dst++ = 0.0f; while (--n);
            if (N == 0) break;
            continue;
        }
        if (v == 0x8) {                                                    /* :147-164 */
            n = (int32_t)get_nybble(r);
            n = (int32_t)get_nybble(r) | (n << 4);
            v = (int32_t)get_nybble(r);
            n = (v & 1) | (n << 1);
            v = (v >> 1) + 1;
            n += 16;
            if (n > N) return 0;
            N -= n;
            float p = (v * v) * quant * (1.0f / 4);
            do {
                st->Seed = orc_xorshift32(st->Seed);
                if (st->Seed & 0x80000000u) p = -p;
                *dst++ = p;
            } while (--n);
            if (N == 0) break;
            continue;
        }
        v = get_quantizer(r);                                              /* :168-172 */
        if (v >= 0) { quant = expand_quantizer(v); continue; }
        if (v == ESC_STOP_NOISE) {                                         /* :175-186 */
            v = (int32_t)get_nybble(r) + 1;
            n = (int32_t)get_nybble(r);
            n = (int32_t)get_nybble(r) | (n << 4);
            float p = (v * v) * quant * (1.0f / 16);
            float rr = 1.0f + (n * n) * -0x1.0p-19f;
            do {
                st->Seed = orc_xorshift32(st->Seed);
                if (st->Seed & 0x80000000u) p = -p;
                *dst++ = p; p *= rr;
            } while (--N);
            break;
        }
        if (v == ESC_STOP) { do *dst++ = 0.0f; while (--N); break; }       /* :191-194 */
    }
    return 1;
}

/* ulcDecoder.c:26-60 */
int orc_decoder_init(orc_decoder *st) {
    int C = st->nChan, BS = st->BlockSize;
    st->TransformBuffer = NULL;
    if (C < 1 || C > 255) return -1;
    if (BS < 256 || BS > 32768) return -1;
    if ((BS & (-BS)) != BS) return -1;
    st->LastSubBlockSize = 0;
    st->Seed = 1234567u;
    st->TransformBuffer = (float *)calloc((size_t)BS, sizeof(float));
    st->TransformTemp   = (float *)calloc((size_t)C * BS * 2, sizeof(float));
    st->TransformInvLap = (float *)calloc((size_t)C * (BS / 2), sizeof(float));
    st->CoefDbg         = (float *)calloc((size_t)C * BS, sizeof(float));
    return 1;
}
void orc_decoder_destroy(orc_decoder *st) {
    free(st->TransformBuffer); free(st->TransformTemp); free(st->TransformInvLap); free(st->CoefDbg);
    memset(st, 0, sizeof(*st));
}

/* ulcDecoder.c:198-302 */
int orc_decode_block(orc_decoder *st, float *DstData, const uint8_t *SrcBuffer) {
    int C = st->nChan, BS = st->BlockSize;
    float *tbuf = st->TransformBuffer, *ttmp = st->TransformTemp, *invlap = st->TransformInvLap;
    nyb_reader rd = { SrcBuffer, 0 };
    int last = 0;
    int wc = (int)get_nybble(&rd);                                         /* :211-216 */
    if (wc & 0x8) wc |= (int)get_nybble(&rd) << 4;
    else          wc |= 1 << 4;
    for (int ch = 0; ch < C; ch++) {
        last = st->LastSubBlockSize;                                       /* :219 */
        float *dst = DstData + ch * BS;
        float *lap = invlap;
        float *dbg = st->CoefDbg + (size_t)ch * BS;
        uint16_t pat = orc_decimation_pattern(wc);
        do {
            int S = BS >> (pat & 7);
            if (!decode_subblock(st, tbuf, S, &rd)) return 0;              /* :228-231 */
            memcpy(dbg, tbuf, sizeof(float) * S); dbg += S;
            int ov = S;                                                    /* :234-239 */
            if (pat & 8) ov >>= (wc & 7);
            if (ov > last) ov = last;
            last = S;
            if (S == BS) { orc_imdct(dst, tbuf, lap, ttmp, S, ov); break; }  /* :242-245 */
            float *dec = ttmp + S;                                         /* :248-249 */
            orc_imdct(dec, tbuf, lap, ttmp, S, ov);
            int avail = (BS - S) / 2;                                      /* :253-272 reversed-time FIFO */
            float *ld = lap + BS / 2;
            const float *ls = ld;
            int n;
            if (S <= avail) {
                for (n = 0; n < S; n++)     *dst++ = *--ls;
                for (; n < avail; n++)      *--ld = *--ls;
                for (n = 0; n < S; n++)     *--ld = *dec++;
            } else {
                for (n = 0; n < avail; n++) *dst++ = *--ls;
                for (; n < S; n++)          *dst++ = *dec++;
                for (n = 0; n < avail; n++) *--ld = *dec++;
            }
        } while (pat >>= 4);
        invlap += BS / 2;
    }
    for (int ch = 1; ch < C; ch += 2) {                                    /* :281-289 */
        float *b = DstData + ch * BS;
        for (int n = 0; n < BS; n++) {
            float m = b[n - BS], s = b[n];
            b[n - BS] = m + s;
            b[n]      = m - s;
        }
    }
    if (C != 1) {                                                          /* :292-297 */
        for (int n = 0; n < BS * C; n++) ttmp[n] = DstData[n];
        for (int ch = 0; ch < C; ch++)
            for (int n = 0; n < BS; n++) DstData[n * C + ch] = ttmp[ch * BS + n];
    }
    st->LastSubBlockSize = last;
    return rd.size;
}

int orc_decode_stream(int nChan, int BlockSize, const uint8_t *in, int slotBytes, int nBlocks, float *pcm, int32_t *bitsRead) {
    orc_decoder st; memset(&st, 0, sizeof(st));
    st.nChan = nChan; st.BlockSize = BlockSize;
    if (orc_decoder_init(&st) < 0) return -1;
    size_t blk = (size_t)nChan * BlockSize;
    int rc = 0;
    for (int k = 0; k < nBlocks; k++) {
        int b = orc_decode_block(&st, pcm + k * blk, in + (size_t)k * slotBytes);
        if (bitsRead) bitsRead[k] = b;
        if (!b) { rc = k + 1; break; }
    }
    orc_decoder_destroy(&st);
    return rc;
}

/* As orc_decode_stream with the noise generator's state handed in and out: the reference's generator is a function-static
 * word (ulcDecoder.c:75-81), so a process that decodes a second file continues the first file's chain.  *seed = 1234567 for
 * the first decoder of a "process", then whatever the previous call left. */
int orc_decode_stream_seeded(int nChan, int BlockSize, const uint8_t *in, int slotBytes, int nBlocks, float *pcm, int32_t *bitsRead, uint32_t *seed) {
    orc_decoder st; memset(&st, 0, sizeof(st));
    st.nChan = nChan; st.BlockSize = BlockSize;
    if (orc_decoder_init(&st) < 0) return -1;
    st.Seed = *seed;
    size_t blk = (size_t)nChan * BlockSize;
    int rc = 0;
    for (int k = 0; k < nBlocks; k++) {
        int b = orc_decode_block(&st, pcm + k * blk, in + (size_t)k * slotBytes);
        if (bitsRead) bitsRead[k] = b;
        if (!b) { rc = k + 1; break; }
    }
    *seed = st.Seed;
    orc_decoder_destroy(&st);
    return rc;
}

/* as orc_decode_stream, also returning every block's dequantised coefficients [nBlocks][nChan*BlockSize]
 * (tests/test_spec_decoder.py checks them against a decoder written from FormatSpecs.md alone) */
int orc_decode_stream_coefs(int nChan, int BlockSize, const uint8_t *in, int slotBytes, int nBlocks, float *pcm, int32_t *bitsRead, float *coefs) {
    orc_decoder st; memset(&st, 0, sizeof(st));
    st.nChan = nChan; st.BlockSize = BlockSize;
    if (orc_decoder_init(&st) < 0) return -1;
    size_t blk = (size_t)nChan * BlockSize;
    int rc = 0;
    for (int k = 0; k < nBlocks; k++) {
        int b = orc_decode_block(&st, pcm + k * blk, in + (size_t)k * slotBytes);
        if (bitsRead) bitsRead[k] = b;
        if (!b) { rc = k + 1; break; }
        if (coefs) memcpy(coefs + k * blk, st.CoefDbg, sizeof(float) * blk);
    }
    orc_decoder_destroy(&st);
    return rc;
}
