/*
 * ulc_amd.h — C ABI of libulc_amd.so, the MI355X (gfx950) implementation of the
 * ulc-codec per-block hot path.  Plain C, plain pointers and sizes; no torch, no
 * C++ types.  Two layers:
 *
 *  1. DROP-IN layer: the reference's own public API, same symbols, same struct
 *     layout (size/offsets are ABI: ULC_EncoderState_t = 104 bytes,
 *     ULC_DecoderState_t = 48 bytes on x86-64), same return conventions, so
 *     /root/reference/tools/ulcEncodeTool.c and ulcDecodeTool.c compile against the
 *     reference's own headers and link against this library unchanged.
 *     Replaces: /root/reference/include/ulcEncoder.h:47-78,85-88,135-137 and
 *               /root/reference/include/ulcDecoder.h:13-32,39-42,56
 *     (implemented by /root/reference/libulc/ulcEncoder.c:25-158 and ulcDecoder.c:26-302).
 *     Each call is a batch of one stream x one block through the batched path below.
 *
 *  2. BATCHED layer (ulcx_*): B independent streams x K consecutive blocks per
 *     call, state resident in HBM between calls.  This is what bench.py measures.
 *     It is what a maintainer would bind from ulcEncodeTool.c's block loop
 *     (tools/ulcEncodeTool.c:133-169) when encoding many files at once
 *     (INTEGRATION.md).
 *
 * Every entry point fails loudly (negative return + ulcx_last_error()) when no
 * gfx950 device / HIP runtime is usable — there is no CPU fallback in this library.
 */
#ifndef ULC_AMD_H
#define ULC_AMD_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ------------------------------------------------------------------------- */
/* 1. Drop-in layer (reference ABI)                                           */
/* ------------------------------------------------------------------------- */
#ifndef ULC_AMD_NO_DROPIN_TYPES
struct ULC_TransientData_t { float Sum, SumW; };           /* ulcEncoder.h:44-46 */

/* ulcEncoder.h:47-78.  Caller sets RateHz/nChan/BlockSize, then Init.  The
 * pointer fields are private to the library: BufferData owns the (host) block,
 * TransformTemp receives each encoded block (the pointer the EncodeBlock calls
 * return), BlockComplexity is readable after each call (ulcEncodeTool.c:164). */
struct ULC_EncoderState_t {
    int    RateHz;
    int    nChan;
    int    BlockSize;
    int    WindowCtrl;
    int    NextWindowCtrl;
    float  BlockComplexity;
    float  TransientFilter[3];
    void  *BufferData;
    float *SampleBuffer;
    float *TransformBuffer;
    float *TransformNoise;
    float *TransformFwdLap;
    float *TransformTemp;
    int   *TransformIndex;
    struct ULC_TransientData_t *TransientBuffer;
};

/* ulcDecoder.h:13-32 */
struct ULC_DecoderState_t {
    int    nChan;
    int    BlockSize;
    int    LastSubBlockSize;
    void  *BufferData;
    float *TransformBuffer;
    float *TransformTemp;
    float *TransformInvLap;
};
#endif

/* ulcEncoder.h:85-88 / ulcEncoder.c:25-88.  Returns 1 on success, -1 on failure
 * (bad nChan/BlockSize, out of memory, or no usable GPU). */
int  ULC_EncoderState_Init(struct ULC_EncoderState_t *State);
void ULC_EncoderState_Destroy(struct ULC_EncoderState_t *State);

/* ulcEncoder.h:135-137 / ulcEncoder.c:93-158.  SrcData: BlockSize*nChan
 * interleaved floats.  Returns a pointer valid until the next call on State;
 * *Size (may be NULL) = bits, multiple of 8. */
const void *ULC_EncodeBlock_CBR(struct ULC_EncoderState_t *State, const float *SrcData, int *Size, float RateKbps);
const void *ULC_EncodeBlock_ABR(struct ULC_EncoderState_t *State, const float *SrcData, int *Size, float RateKbps, float AvgComplexity);
const void *ULC_EncodeBlock_VBR(struct ULC_EncoderState_t *State, const float *SrcData, int *Size, float Quality);

/* ulcDecoder.h:39-42,56 / ulcDecoder.c:26-302.  DecodeBlock returns bits
 * consumed (nybble granular), 0 = corrupt block. */
int  ULC_DecoderState_Init(struct ULC_DecoderState_t *State);
void ULC_DecoderState_Destroy(struct ULC_DecoderState_t *State);
int  ULC_DecodeBlock(struct ULC_DecoderState_t *State, float *DstData, const void *SrcBuffer);

/* Bytes of SrcBuffer that ULC_DecodeBlock (the reference's, ulcDecoder.c:82-216, and this library's) touches for the
 * block that starts there: a host-side walk of the block syntax that only counts.  Reads nothing at or past maxBytes and
 * returns at most maxBytes.  Host code, no GPU involved: ULC_DecodeBlock stages exactly this many bytes, and a
 * container reader can use it to index a .ulc payload (blocks carry no length, tools/ulcDecodeTool.c:153-165). */
int  ulcx_block_extent_bytes(const void *SrcBuffer, int nChan, int BlockSize, int maxBytes);
/* Round 3: the slot-form decoder cuts a call's (stream, block) pairs evenly over the synthesis workgroups when one workgroup
 * per stream would leave the device idle (few long streams).  This is the cut's arithmetic, exported for inspection and the
 * tests: the number of workgroups for nBlocks blocks of nStreams streams on a device that holds residentWG workgroups of the
 * synthesis kernel (1536 on an MI355X for stereo BlockSize 2048), or 0 = one workgroup per stream.  No reference counterpart
 * (the reference decodes one block per call, ulcDecoder.c:200-302). */
int  ulcx_dec_split_plan(int nStreams, int nBlocks, int residentWG);
/* Round 5: a batch of more streams than the device holds workgroups runs in rounds of one workgroup per stream; when the last
 * round is partly empty (4096 streams on 1536 resident workgroups) only ITS streams are cut, over the returned number of
 * workgroups at the end of the grid (0: no cut); *full receives the number of leading workgroups that take one whole stream
 * each.  Host arithmetic, exported for inspection and the tests.  No reference counterpart. */
int  ulcx_dec_tail_plan(int nStreams, int nBlocks, int residentWG, int *full);

/* ------------------------------------------------------------------------- */
/* 2. Batched layer                                                           */
/* ------------------------------------------------------------------------- */
typedef struct ulcx_encoder ulcx_encoder;
typedef struct ulcx_decoder ulcx_decoder;

enum {
    ULCX_OK            =  0,
    ULCX_ERR_ARG       = -1,   /* same validation as ulcEncoder.c:32-34 + batch limits */
    ULCX_ERR_NO_DEVICE = -2,   /* HIP runtime / gfx950 device not usable */
    ULCX_ERR_HIP       = -3,   /* a HIP call failed (message in ulcx_last_error) */
    ULCX_ERR_NOMEM     = -4,
    ULCX_ERR_UNSUPPORTED = -5  /* valid for the reference, not built for the device (no geometry is refused on this ground any more:
                                  BlockSize up to 32768 and up to 255 channels are accepted, as ulcEncoder.c:32-34 / ulcDecoder.c:33-35 do) */
};

enum { ULCX_MODE_VBR = 0, ULCX_MODE_CBR = 1, ULCX_MODE_ABR = 2 };

const char *ulcx_last_error(void);
int  ulcx_device_count(void);                 /* <= 0 when no usable device */
const char *ulcx_build_rev(void);             /* 12 hex digits: sha1 of the sources the library was built from (Makefile) */

/* Encoder for nStreams independent streams (all same RateHz/nChan/BlockSize);
 * at most maxBlocksPerCall blocks per stream per call.  device = HIP ordinal. */
int  ulcx_encoder_create(ulcx_encoder **enc, int device, int nStreams, int nChan, int BlockSize, int RateHz, int maxBlocksPerCall);
void ulcx_encoder_destroy(ulcx_encoder *enc);
int  ulcx_encoder_reset(ulcx_encoder *enc);   /* back to the state right after create */
int  ulcx_encoder_slot_bytes(const ulcx_encoder *enc);   /* bytes reserved per encoded block */

/* Encode nBlocks consecutive blocks of every stream.  All pointers are DEVICE
 * pointers; work is enqueued on hipStream (a hipStream_t, NULL = default stream)
 * and is asynchronous with respect to the host in every mode: nothing inside the call
 * waits for the device (the CBR/ABR rate search enqueues its full number of probe passes;
 * passes that find every block converged return at once on the device).  Internally the
 * call also uses private side streams, all joined back into hipStream before it returns.
 *   d_pcm  [nStreams][nBlocks][BlockSize][nChan] f32 interleaved (the layout the
 *          reference's EncodeBlock reads: ulcEncoder_BlockTransform.c:96-98)
 *   d_out  [nStreams][nBlocks][slot_bytes] encoded blocks, byte aligned, each
 *          identical to what ULC_EncodeBlock_* returns for that call
 *   d_bits [nStreams][nBlocks] int32 block size in bits (multiple of 8)
 *   d_wc   optional [nStreams][nBlocks] int32 State->WindowCtrl of that call
 *   d_cplx optional [nStreams][nBlocks] f32 State->BlockComplexity of that call
 * mode/param0/param1: VBR(Quality) | CBR(RateKbps) | ABR(RateKbps, AvgComplexity). */
int  ulcx_encode_dev(ulcx_encoder *enc, int mode, float param0, float param1,
                     const float *d_pcm, int nBlocks,
                     uint8_t *d_out, int32_t *d_bits, int32_t *d_wc, float *d_cplx, void *hipStream);

/* PCM16 ingest: as ulcx_encode_dev, with d_pcm16 [nStreams][nBlocks][BlockSize][nChan]
 * int16 interleaved.  Samples are converted on load exactly as the reference's WAV reader feeds
 * ULC_EncodeBlock_* (tools/WavIO_Helper.c:49-55: (float)x * 2^-15), so the stream is identical to
 * converting on the host and calling ulcx_encode_dev; the input traffic is halved. */
int  ulcx_encode_dev_pcm16(ulcx_encoder *enc, int mode, float param0, float param1,
                           const int16_t *d_pcm16, int nBlocks,
                           uint8_t *d_out, int32_t *d_bits, int32_t *d_wc, float *d_cplx, void *hipStream);

/* Host-pointer convenience (H2D, encode, D2H, synchronous). */
int  ulcx_encode_host(ulcx_encoder *enc, int mode, float param0, float param1,
                      const float *h_pcm, int nBlocks,
                      uint8_t *h_out, int32_t *h_bits, int32_t *h_wc, float *h_cplx);

/* Debug/parity taps (device->host copies of the intermediates of the LAST call;
 * what the reference keeps in State->TransformBuffer / TransformNoise / the final
 * importance keys).  Each array is [nStreams][nBlocks][nChan*BlockSize] f32; pass
 * NULL to skip.  keep: [nStreams][nBlocks][nChan*BlockSize] uint8 (1 = coefficient
 * rank < nOutCoef of the final pass). nout: [nStreams][nBlocks] int32. */
int  ulcx_encoder_debug_fetch(ulcx_encoder *enc, int nBlocks, float *h_coef, float *h_noise, float *h_keys,
                              uint8_t *h_keep, int32_t *h_nout);

/* Number of blocks of the last call (final pass) whose threshold tie group straddled the
 * cut and went through the exact heapsort emulation (BlockTransform.c:20-77).  Test hook. */
int  ulcx_encoder_last_fallbacks(ulcx_encoder *enc);
/* Test hook: how the last decode call's synthesis was launched - workgroups (0: one per stream), how many of them took one
 * whole stream each (ulcx_dec_tail_plan; 0 for an even cut, ulcx_dec_split_plan), and the workgroups of the synthesis kernel
 * the device holds at once (0: this geometry runs the general kernel, never cut). */
int  ulcx_decoder_last_cut(ulcx_decoder *dec, int *workgroups, int *wholeStreams, int *residentWG);
/* Test hook: from the next call on every `every`-th block of a call (block index % every == 0) is handed to the exact
 * heapsort path whether or not its threshold tie group straddles the cut (0 = off, the default).  The results must not
 * change: the full ranking decides the same kept set.  Exercises that path at a scale natural ties never reach. */
int  ulcx_encoder_debug_force_exact(ulcx_encoder *enc, int every);

int  ulcx_decoder_create(ulcx_decoder **dec, int device, int nStreams, int nChan, int BlockSize, int maxBlocksPerCall);
void ulcx_decoder_destroy(ulcx_decoder *dec);
int  ulcx_decoder_reset(ulcx_decoder *dec);

/* Decode nBlocks consecutive blocks of every stream (device pointers).
 *   d_in   [nStreams][nBlocks][slotBytes] encoded blocks (each starts at its slot; slotBytes need not exceed the
 *          largest block: nothing outside [d_in, d_in + nStreams*nBlocks*slotBytes) is read)
 *   d_pcm  [nStreams][nBlocks][BlockSize][nChan] f32 interleaved
 *          (ulcDecoder.c:291-297)
 *   d_bits [nStreams][nBlocks] int32 bits consumed per block; 0 = corrupt: that
 *          stream stops there (later blocks also report 0), like the tool aborting
 *          (tools/ulcDecodeTool.c:154-157). */
int  ulcx_decode_dev(ulcx_decoder *dec, const uint8_t *d_in, int slotBytes, int nBlocks,
                     float *d_pcm, int32_t *d_bits, void *hipStream);
int  ulcx_decode_host(ulcx_decoder *dec, const uint8_t *h_in, int slotBytes, int nBlocks,
                      float *h_pcm, int32_t *h_bits);
/* One block of ONE stream per call, host pointers: what the drop-in ABI of section 1 does per ULC_EncodeBlock_* /
 * ULC_DecodeBlock call (encoder / decoder created with nStreams = 1, maxBlocksPerCall = 1).  The call's whole launch
 * sequence is captured into a HIP graph the first time (again when mode / parameters change) and replayed from pinned
 * staging buffers with one synchronisation; when capture is not possible the same sequence is enqueued directly.
 *   h_out     slot bytes (ulcx_encoder_slot_bytes)          stateOut  {WindowCtrl, NextWindowCtrl} and TransientFilter[3]
 *                                                                     as the reference leaves them in its state struct
 *                                                                     (ulcEncoder.h:57-64; BlockTransform.c:116-125)
 *   lastSubBlockSize  ulcDecoder.h:24 / ulcDecoder.c:300 */
int  ulcx_encode_block1(ulcx_encoder *enc, int mode, float param0, float param1, const float *h_pcm,
                        uint8_t *h_out, int32_t *bits, float *cplx, int32_t stateOut[2], float transientFilter[3]);
int  ulcx_decode_block1(ulcx_decoder *dec, const uint8_t *h_in, int nBytes, float *h_pcm, int32_t *bits, int32_t *lastSubBlockSize);
/* As ulcx_decode_block1 with the noise generator's state handed in and out.  The reference keeps that state in a
 * function-static word (libulc/ulcDecoder.c:75-81): one xorshift32 chain per PROCESS, shared by every decoder object, never
 * re-seeded by ULC_DecoderState_Init.  The drop-in of section 1 owns such a word and passes it here, so a process that
 * decodes several files one after the other draws the same noise as with the reference.  rngState NULL = the state stays
 * with the decoder object (what the batched entries do per stream). */
int  ulcx_decode_block1_rng(ulcx_decoder *dec, const uint8_t *h_in, int nBytes, float *h_pcm, int32_t *bits, int32_t *lastSubBlockSize,
                            uint32_t *rngState);

/* PCM16 output: as ulcx_decode_dev, writing d_pcm16 [nStreams][nBlocks][BlockSize][nChan]
 * int16, converted on store exactly as the reference's WAV writer does with ULC_DecodeBlock's output
 * (tools/WavIO_Helper.c:9-13,56-63: lrintf(clamp(x * 2^15, -32768, 32767))). */
int  ulcx_decode_dev_pcm16(ulcx_decoder *dec, const uint8_t *d_in, int slotBytes, int nBlocks,
                           int16_t *d_pcm16, int32_t *d_bits, void *hipStream);

/* ------------------------------------------------------------------------- */
/* 3. `.ulc` container and packed streams               */
/* ------------------------------------------------------------------------- */
/* 24-byte little-endian file header, tools/ulc_Helper.h:10-20.  Blocks follow at
 * StreamOffs, each rounded up to a whole byte, with NO per-block length
 * (tools/ulcEncodeTool.c:160-169, tools/ulcDecodeTool.c:153-165). */
#define ULCX_ULC_MAGIC 0x32434C55u            /* 'U' | 'L'<<8 | 'C'<<16 | '2'<<24 */
typedef struct ulcx_file_header {
    uint32_t Magic;
    uint16_t BlockSize;
    uint16_t MaxBlockSize;                     /* largest block in bytes (0 = unknown) */
    uint32_t nBlocks;
    uint32_t RateHz;
    uint16_t nChan;
    uint16_t RateKbps;                         /* lrint(total_bytes*8*RateHz/1000/(BlockSize*nBlocks)), ulcEncodeTool.c:173-190 */
    uint32_t StreamOffs;
} ulcx_file_header;
void ulcx_ulc_header_pack(uint8_t dst[24], const ulcx_file_header *h);
int  ulcx_ulc_header_parse(ulcx_file_header *h, const uint8_t *src, size_t len);   /* 0 ok, ULCX_ERR_ARG: short / bad magic */
int  ulcx_ulc_rate_kbps(uint64_t totalBytes, uint32_t RateHz, uint32_t BlockSize, uint32_t nBlocks);

/* Concatenate the per-block slots an encode call produced into one contiguous payload
 * per stream (what the tool's fwrite loop produces).  Device pointers.
 *   d_payload      [nStreams][payloadStride] bytes, stream s starts at s*payloadStride
 *   d_payloadBytes [nStreams] bytes written;  d_maxBlock [nStreams] largest block (optional) */
int  ulcx_pack_streams_dev(int device, int nStreams, int nBlocks, int slotBytes, const uint8_t *d_slots, const int32_t *d_bits,
                           uint8_t *d_payload, long long payloadStride, int32_t *d_payloadBytes, int32_t *d_maxBlock, void *hipStream);

/* Decode the next nBlocks blocks of every stream from packed payloads: block k+1 begins at the
 * byte after block k ends, which only parsing reveals; the per-stream read position persists
 * across calls (ulcx_decoder_reset rewinds it).  d_payloadBytes[s] = valid bytes of stream s. */
int  ulcx_decode_packed_dev(ulcx_decoder *dec, const uint8_t *d_payload, long long payloadStride, const int32_t *d_payloadBytes,
                            int nBlocks, float *d_pcm, int32_t *d_bits, void *hipStream);
int  ulcx_decode_packed_host(ulcx_decoder *dec, const uint8_t *h_payload, long long payloadStride, const int32_t *h_payloadBytes,
                             int nBlocks, float *h_pcm, int32_t *h_bits);

/* Whole files (what ulcx-tool does, tools/ulcDecodeTool.c:123-166 batched): upload every stream's payload ONCE - the
 * read positions are rewound - then each ulcx_decode_resident_host call decodes the next nBlocks blocks of every stream
 * from the device-resident copy.  (ulcx_decode_packed_host uploads all payloads on every call.) */
int  ulcx_decoder_upload_payload(ulcx_decoder *dec, const uint8_t *h_payload, long long payloadStride, const int32_t *h_payloadBytes);
int  ulcx_decode_resident_host(ulcx_decoder *dec, int nBlocks, float *h_pcm, int32_t *h_bits);

/* Timing helper for bench.py: device time (ms, hipEvent) of the kernels the last
 * ulcx_*_dev call enqueued, per pipeline stage; returns number of stages written.
 * Only valid after the stream has been synchronised. */
int  ulcx_encoder_stage_ms(ulcx_encoder *enc, float *ms, int maxStages);
/* The events behind the two calls above are recorded around every kernel of every ulcx_*_dev call (default on);
 * a caller that never reads them can switch them off. */
int  ulcx_encoder_set_timing(ulcx_encoder *enc, int on);
int  ulcx_decoder_set_timing(ulcx_decoder *dec, int on);
const char *ulcx_encoder_stage_name(int stage);
/* Launches of the transform kernel (k_xf) in the last call: the stage time above is their sum (1 when the
 * window-control pipeline is off). */
int  ulcx_encoder_last_xf_launches(ulcx_encoder *enc);
int  ulcx_decoder_stage_ms(ulcx_decoder *dec, float *ms, int maxStages);
const char *ulcx_decoder_stage_name(int stage);

#ifdef __cplusplus
}
#endif
#endif
