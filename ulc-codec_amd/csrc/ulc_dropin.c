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
struct dec_priv { ulcx_decoder *dec; unsigned char *in; int slot; };

/* ulcEncoder.c:25-80: 1 on success, -1 on failure */
int ULC_EncoderState_Init(struct ULC_EncoderState_t *State) {
    State->BufferData = NULL;
    State->SampleBuffer = State->TransformBuffer = State->TransformNoise = NULL;
    State->TransformFwdLap = State->TransformTemp = NULL;
    State->TransformIndex = NULL;
    State->TransientBuffer = NULL;
    ulcx_encoder *enc = NULL;
    int rc = ulcx_encoder_create(&enc, 0, 1, State->nChan, State->BlockSize, State->RateHz, 1);
    if (rc != ULCX_OK) {
        if (rc != ULCX_ERR_ARG) fprintf(stderr, "libulc_amd: encoder init failed: %s\n", ulcx_last_error());
        return -1;
    }
    struct enc_priv *p = (struct enc_priv *)malloc(sizeof(*p));
    if (!p) { ulcx_encoder_destroy(enc); return -1; }
    p->enc = enc;
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
    int32_t bits = 0, wc = 0; float cplx = 0.0f;
    int rc = ulcx_encode_host(p->enc, mode, p0, p1, Src, 1, p->out, &bits, &wc, &cplx);
    if (rc != ULCX_OK) {
        /* the reference cannot fail here; a lost GPU is fatal for a drop-in */
        fprintf(stderr, "libulc_amd: encode failed: %s\n", ulcx_last_error());
        abort();
    }
    State->WindowCtrl = wc;
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
    int rc = ulcx_decoder_create(&dec, 0, 1, State->nChan, State->BlockSize, 1);
    if (rc != ULCX_OK) {
        if (rc != ULCX_ERR_ARG) fprintf(stderr, "libulc_amd: decoder init failed: %s\n", ulcx_last_error());
        return -1;
    }
    struct dec_priv *p = (struct dec_priv *)malloc(sizeof(*p));
    if (!p) { ulcx_decoder_destroy(dec); return -1; }
    p->dec = dec;
    /* a block is at most 4 nybbles per coefficient + header; the caller's buffer is read
     * only up to the end of the block, so copy that bound (ulcDecoder.h:54) */
    p->slot = 2 * State->nChan * State->BlockSize + 16;
    p->in = (unsigned char *)malloc((size_t)p->slot);
    if (!p->in) { ulcx_decoder_destroy(dec); free(p); return -1; }
    State->BufferData = p;
    State->LastSubBlockSize = 0;
    return 1;
}
void ULC_DecoderState_Destroy(struct ULC_DecoderState_t *State) {
    struct dec_priv *p = (struct dec_priv *)State->BufferData;
    if (!p) return;
    ulcx_decoder_destroy(p->dec);
    free(p->in);
    free(p);
    State->BufferData = NULL;
}
/* ulcDecoder.c:198-302.  NOTE: the reference reads SrcBuffer only as far as the block
 * extends; the tool hands in a sliding window with at least MaxBlockSize bytes valid
 * (tools/ulcDecodeTool.c:123-166), which is what ULC_DecodeBlockN below relies on. */
int ULC_DecodeBlock(struct ULC_DecoderState_t *State, float *DstData, const void *SrcBuffer) {
    struct dec_priv *p = (struct dec_priv *)State->BufferData;
    int32_t bits = 0;
    memcpy(p->in, SrcBuffer, (size_t)p->slot);
    int rc = ulcx_decode_host(p->dec, p->in, p->slot, 1, DstData, &bits);
    if (rc != ULCX_OK) { fprintf(stderr, "libulc_amd: decode failed: %s\n", ulcx_last_error()); return 0; }
    return bits;
}
