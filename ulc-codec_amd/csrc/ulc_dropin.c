/*
 * ulc_dropin.c — the reference's public API (include/ulcEncoder.h:85-137,
 * include/ulcDecoder.h:39-56) implemented as plain C over the batched layer: every
 * call is a batch of ONE stream x ONE block through the same HIP kernels, with the
 * codec state resident on the GPU.  This is the shim that lets
 * tools/ulcEncodeTool.c / tools/ulcDecodeTool.c link unchanged.
 * Correct but launch-latency bound; the throughput path is ulcx_encode_dev /
 * ulcx_decode_dev over many streams.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "../../include/ulc_amd.h"

struct enc_priv { ulcx_encoder *enc; unsigned char *out; int slot; };
struct dec_priv { ulcx_decoder *dec; int slot; };

/* HIP device the drop-in states live on: ULC_AMD_DEVICE (ordinal), default 0 */
static int dropin_device(void) {
    const char *v = getenv("ULC_AMD_DEVICE");
    return (v && v[0]) ? atoi(v) : 0;
}

/* ulcEncoder.c:25-80: 1 on success, -1 on failure */
int ULC_EncoderState_Init(struct ULC_EncoderState_t *State) {
    State->BufferData = NULL;
    State->SampleBuffer = State->TransformBuffer = State->TransformNoise = NULL;
    State->TransformFwdLap = State->TransformTemp = NULL;
    State->TransformIndex = NULL;
    State->TransientBuffer = NULL;
    ulcx_encoder *enc = NULL;
    int rc = ulcx_encoder_create(&enc, dropin_device(), 1, State->nChan, State->BlockSize, State->RateHz, 1);
    if (rc != ULCX_OK) {
        if (rc != ULCX_ERR_ARG) fprintf(stderr, "libulc_amd: encoder init failed: %s\n", ulcx_last_error());
        return -1;
    }
    struct enc_priv *p = (struct enc_priv *)malloc(sizeof(*p));
    if (!p) { ulcx_encoder_destroy(enc); return -1; }
    p->enc = enc;
    ulcx_encoder_set_timing(enc, 0);                          /* nobody reads per-kernel events through this ABI */
    p->slot = ulcx_encoder_slot_bytes(enc);
    p->out = (unsigned char *)malloc((size_t)p->slot);
    if (!p->out) { ulcx_encoder_destroy(enc); free(p); return -1; }
    State->BufferData = p;
    State->TransformTemp = (float *)p->out;      /* the pointer EncodeBlock_* returns (ulcEncoder.c:118) */
    State->WindowCtrl = 0;
    State->NextWindowCtrl = 0x10;                /* ulcEncoder.c:70 */
    State->BlockComplexity = 0.0f;
    State->TransientFilter[0] = State->TransientFilter[1] = State->TransientFilter[2] = 0.0f;
    return 1;
}

void ULC_EncoderState_Destroy(struct ULC_EncoderState_t *State) {
    struct enc_priv *p = (struct enc_priv *)State->BufferData;
    if (!p) return;
    ulcx_encoder_destroy(p->enc);
    free(p->out);
    free(p);
    State->BufferData = NULL;
}

static const void *encode_one(struct ULC_EncoderState_t *State, const float *Src, int *Size, int mode, float p0, float p1) {
    struct enc_priv *p = (struct enc_priv *)State->BufferData;
    int32_t bits = 0, st[2] = { 0, 0x10 }; float cplx = 0.0f, tf[3] = { 0.0f, 0.0f, 0.0f };
    int rc = ulcx_encode_block1(p->enc, mode, p0, p1, Src, p->out, &bits, &cplx, st, tf);
    if (rc != ULCX_OK) {
        /* The reference cannot fail here and its ABI has no way to say so; its tools never look at *Size
         * (tools/ulcEncodeTool.c:157-168 write (Size+7)/8 bytes and carry on), so any "empty block" convention would leave a
         * .ulc whose header counts blocks the payload lacks, with exit code 0.  A lost device mid-file is therefore fatal,
         * loudly: message, then abort(). */
        fprintf(stderr, "libulc_amd: encode failed (%s): the ULC_EncodeBlock_* ABI cannot report an error - aborting\n", ulcx_last_error());
        abort();
    }
    /* the fields the reference leaves behind in the caller's struct (ulcEncoder_BlockTransform.c:116-125, ulcEncoder_WindowControl.c:88-89,131) */
    State->WindowCtrl = st[0];
    State->NextWindowCtrl = st[1];
    State->TransientFilter[0] = tf[0]; State->TransientFilter[1] = tf[1]; State->TransientFilter[2] = tf[2];
    State->BlockComplexity = cplx;
    if (Size) *Size = bits;
    return p->out;
}
const void *ULC_EncodeBlock_CBR(struct ULC_EncoderState_t *State, const float *SrcData, int *Size, float RateKbps) {
    return encode_one(State, SrcData, Size, ULCX_MODE_CBR, RateKbps, 0.0f);
}
const void *ULC_EncodeBlock_ABR(struct ULC_EncoderState_t *State, const float *SrcData, int *Size, float RateKbps, float AvgComplexity) {
    return encode_one(State, SrcData, Size, ULCX_MODE_ABR, RateKbps, AvgComplexity);
}
const void *ULC_EncodeBlock_VBR(struct ULC_EncoderState_t *State, const float *SrcData, int *Size, float Quality) {
    return encode_one(State, SrcData, Size, ULCX_MODE_VBR, Quality, 0.0f);
}

/* ulcDecoder.c:26-60 */
int ULC_DecoderState_Init(struct ULC_DecoderState_t *State) {
    State->BufferData = NULL;
    State->TransformBuffer = State->TransformTemp = State->TransformInvLap = NULL;
    ulcx_decoder *dec = NULL;
    int rc = ulcx_decoder_create(&dec, dropin_device(), 1, State->nChan, State->BlockSize, 1);
    if (rc != ULCX_OK) {
        if (rc != ULCX_ERR_ARG) fprintf(stderr, "libulc_amd: decoder init failed: %s\n", ulcx_last_error());
        return -1;
    }
    struct dec_priv *p = (struct dec_priv *)malloc(sizeof(*p));
    if (!p) { ulcx_decoder_destroy(dec); return -1; }
    p->dec = dec;
    ulcx_decoder_set_timing(dec, 0);
    /* a block is at most 4 nybbles per coefficient + header; the caller's buffer is read
     * only up to the end of the block, so copy that bound (ulcDecoder.h:54) */
    p->slot = 2 * State->nChan * State->BlockSize + 16;
    State->BufferData = p;
    State->LastSubBlockSize = 0;
    return 1;
}
void ULC_DecoderState_Destroy(struct ULC_DecoderState_t *State) {
    struct dec_priv *p = (struct dec_priv *)State->BufferData;
    if (!p) return;
    ulcx_decoder_destroy(p->dec);
    free(p);
    State->BufferData = NULL;
}
/* Bytes of SrcBuffer the reference's decoder would touch (ulcDecoder.c:82-88: a byte is read with its first nybble): a host
 * walk of the block syntax (ulcDecoder.c:99-216, FormatSpecs.md:57-141) that only counts - header, then per channel and
 * subblock the opening quantizer and codes until the subblock's coefficients are accounted for or a run overshoots (the
 * reference returns there).  Never reads at or past maxBytes. */
int ulcx_block_extent_bytes(const void *SrcBuffer, int nChan, int BS, int maxBytes) {
    const unsigned char *src = (const unsigned char *)SrcBuffer;
    static const unsigned short pattern[16] = { 0x0000, 0x0008, 0x0019, 0x0091, 0x012A, 0x01A2, 0x02A1, 0x0A21,
                                                0x123B, 0x12B3, 0x13B2, 0x1B32, 0x23B1, 0x2B31, 0x3B21, 0xB321 };   /* ulcHelper.h:24-46 */
    const long maxNyb = 2L * maxBytes;
    long n = 0;                                              /* nybbles consumed */
#define NYB() ((n < maxNyb) ? (unsigned)((src[n >> 1] >> ((n & 1) * 4)) & 0xF) : 0xFu); if (n >= maxNyb) return maxBytes; n++
    unsigned wc, v;
    wc = NYB();
    if (wc & 8) { v = NYB(); wc |= v << 4; } else wc |= 1 << 4;
    for (int ch = 0; ch < nChan; ch++) {
        unsigned pat = pattern[(wc >> 4) & 15];
        do {
            int N = BS >> (pat & 7);
            const int whole = (N == BS);
            /* opening quantizer (ulcDecoder.c:89-107): Eh takes a second nybble; Eh,Fh = stop; a bare Fh leaves quantizer 0 */
            v = NYB();
            if (v == 0xE) { v = NYB(); if (v == 0xF) N = 0; }
            while (N > 0) {
                v = NYB();
                if (v == 0x0) { v = NYB(); int r = (int)v + 1; if (r > N) goto done; N -= r; }
                else if (v == 0x1) { unsigned y = NYB(); unsigned x = NYB(); int r = (int)((y << 4) | x) + 33; if (r > N) goto done; N -= r; }
                else if (v == 0x8) { unsigned z = NYB(); unsigned y = NYB(); unsigned x = NYB(); int r = (int)((z << 5) | (y << 1) | (x & 1)) + 16; if (r > N) goto done; N -= r; }
                else if (v == 0xF) {
                    v = NYB();
                    if (v == 0xF) { v = NYB(); v = NYB(); v = NYB(); N = 0; }        /* noise to the end */
                    else if (v == 0xE) { v = NYB(); if (v == 0xF) N = 0; }            /* extended quantizer / stop */
                }
                else N -= 1;
            }
            if (whole) break;                                /* ulcDecoder.c:242-245 */
        } while (pat >>= 4);
    }
done:
#undef NYB
    { long bytes = (n + 1) >> 1; return bytes < maxBytes ? (int)bytes : maxBytes; }
}

/* The noise generator of the reference is a function-static word (ulcDecoder.c:75-81, `static uint32_t Seed = 1234567`):
 * one chain per process, shared by all decoder states and not reset by ULC_DecoderState_Init.  Same here: the word lives in
 * this shim and travels through every ULC_DecodeBlock call.  (As in the reference, two threads decoding at once interleave
 * their draws in an unspecified order; the accesses themselves are atomic here.) */
static unsigned int g_decode_rng = 1234567u;

/* ulcDecoder.c:198-302.  The reference reads SrcBuffer only as far as the block extends (ulcDecoder.h:54), and so does this:
 * the block's extent is found on the host first, exactly those bytes are staged (the rest of the staging slot is zero). */
int ULC_DecodeBlock(struct ULC_DecoderState_t *State, float *DstData, const void *SrcBuffer) {
    struct dec_priv *p = (struct dec_priv *)State->BufferData;
    int32_t bits = 0, lastSub = State->LastSubBlockSize;
    const int ext = ulcx_block_extent_bytes(SrcBuffer, State->nChan, State->BlockSize, p->slot);
    uint32_t rng = __atomic_load_n(&g_decode_rng, __ATOMIC_RELAXED);
    int rc = ulcx_decode_block1_rng(p->dec, (const unsigned char *)SrcBuffer, ext, DstData, &bits, &lastSub, &rng);
    if (rc != ULCX_OK) { fprintf(stderr, "libulc_amd: decode failed: %s\n", ulcx_last_error()); return 0; }
    __atomic_store_n(&g_decode_rng, rng, __ATOMIC_RELAXED);
    State->LastSubBlockSize = lastSub;                       /* ulcDecoder.c:300 */
    return bits;
}
