/*
 * ORACLE — TEST INFRASTRUCTURE ONLY (see oracle/README.md).
 *
 * CPU restatement, in plain C, of the ulc-codec per-block hot path
 * (BASELINE.json north_star).  Every function cites the
 * reference file:line it follows.  The product (ulc-codec_amd/) never includes
 * or links this.
 *
 * Pinning status:
 *   PINNED against the real reference, compiled directly from
 *   /root/reference/libulc/{ulcEncoder_WindowControl,ulcEncoder_Psyopt,
 *   ulcEncoder_NoiseFill}.c into oracle/_ref/ (those three files need nothing
 *   the image lacks):
 *       orc_get_window_ctrl, orc_calc_psychoacoustics,
 *       orc_calc_noise_log_spectrum, orc_get_noise_q, orc_get_hfext_params
 *   PARITY UNPINNED: everything that includes "Fourier.h" in the reference
 *   (ulcEncoder.c, ulcEncoder_BlockTransform.c, ulcEncoder_Encode.c,
 *   ulcDecoder.c) cannot be built here without writing a stand-in for the
 *   absent libfourier header, and the reference ships no tests or golden
 *   vectors.  Those parts are restated from the source text and checked against
 *   FormatSpecs.md (normative bitstream + IMDCT definition) and by
 *   encode->decode round trips only.
 */
#ifndef ULC_ORACLE_H
#define ULC_ORACLE_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ORC_MAX_SUBBLOCKS 4
#define ORC_MAX_DECIMATION 8
#define ORC_COEF_EPS (0x1.0p-31f)          /* include/ulcEncoder.h:36 */
#define ORC_N_BARK_BANDS 25

typedef struct { float Sum, SumW; } orc_transient_t;   /* include/ulcEncoder.h:44-46 */

typedef struct orc_encoder {
    int RateHz, nChan, BlockSize;
    int WindowCtrl, NextWindowCtrl;
    float BlockComplexity;
    float TransientFilter[3];
    float *SampleBuffer;      /* [2][nChan][BlockSize]  Old | New              */
    float *TransformBuffer;   /* [nChan][BlockSize] normalised MDCT coefs      */
    float *TransformNoise;    /* [nChan][BlockSize] {w, w*logNoise} pairs      */
    float *TransformFwdLap;   /* [nChan][BlockSize]                            */
    float *TransformTemp;     /* [max(2,nChan)][BlockSize]                     */
    int   *TransformIndex;    /* [nChan][BlockSize] keys (f32 bits) then ranks */
    float *Keys;              /* copy of the final keys (debug/parity only)    */
    float *Masking;           /* [BlockSize/2] copy of MaskingNp (debug only)  */
    float *MDSTdbg;           /* [nChan][BlockSize] raw un-normalised MDST     */
    orc_transient_t TransientBuffer[ORC_MAX_DECIMATION * 2];
    int nNzCoef;
    int lastNOutCoef;         /* nOutCoef of the final pass of the last call   */
} orc_encoder;

typedef struct orc_decoder {
    int nChan, BlockSize;
    int LastSubBlockSize;
    uint32_t Seed;            /* per-state RNG (reference: function static, ulcDecoder.c:76) */
    float *TransformBuffer;   /* [BlockSize]           */
    float *TransformTemp;     /* [2*nChan*BlockSize]   */
    float *TransformInvLap;   /* [nChan][BlockSize/2]  */
    float *CoefDbg;           /* [nChan][BlockSize] dequantised coefficients of last block */
} orc_decoder;

/* ---- fourier (orc_fourier.c) ---- */
void orc_window_tables(int Ov, float *fall, float *rise);
void orc_dct4(float *X, const float *u, float *work, int N);
void orc_mdct_mdst(float *MDCT, float *MDST, const float *New, float *Lap, float *Tmp, int N, int Overlap);
void orc_imdct(float *Out, const float *In, float *Lap, float *Tmp, int N, int Overlap);
void orc_ref64_mdct_mdst(double *MDCT, double *MDST, const float *New, const double *Lap, double *LapOut, int N, int Overlap);
void orc_ref64_imdct_raw(double *y, const float *In, int N);

/* ---- helpers (ulcHelper.h) ---- */
uint16_t orc_decimation_pattern(int WindowCtrl);
float orc_fastlog(float x);
int   orc_companded_quantize_unsigned(float v);
int   orc_build_quantizer(float MaxVal);
int   orc_companded_quantize(float v);
int   orc_quant_coef_unsigned(float v, int limit);
int   orc_quant_coef(float v, int limit);
float orc_freq_to_line(float hz, float nyq, uint32_t n);
float orc_line_to_freq(uint32_t line, float nyq, uint32_t n);
float orc_freq_to_bark(float hz);
float orc_bark_to_freq(float bark);

/* ---- pinned units (same signatures as the reference's ULCi_* functions) ---- */
int  orc_get_window_ctrl(const float *BlockData, orc_transient_t *TransientBuffer, float *TransientFilter,
                         float *TmpBuffer, int BlockSize, int nChan, int RateHz);
void orc_calc_psychoacoustics(float *MaskingNp, float *BufferAmp2, void *BufferTemp, int BlockSize, int RateHz, uint32_t WindowCtrl);
void orc_calc_noise_log_spectrum(float *Data, void *Temp, int N, int RateHz);
int  orc_get_noise_q(const float *Data, int Band, int N, float q);
void orc_get_hfext_params(const float *Data, int Band, int N, float q, int *NoiseQ, int *NoiseDecay);

/* ---- encoder ---- */
int  orc_encoder_init(orc_encoder *st);            /* 1 / -1 like ULC_EncoderState_Init */
void orc_encoder_destroy(orc_encoder *st);
void orc_sort_indices(int *SortedIndices, const float *SortValues, int *Temp, int N);
int  orc_transform_block(orc_encoder *st, const float *Data);
int  orc_encode_pass(const orc_encoder *st, void *Dst, int nOutCoef);
int  orc_encode_block_vbr(orc_encoder *st, uint8_t *Dst, const float *Src, float Quality);
int  orc_encode_block_cbr(orc_encoder *st, uint8_t *Dst, const float *Src, float RateKbps);
int  orc_encode_block_abr(orc_encoder *st, uint8_t *Dst, const float *Src, float RateKbps, float AvgComplexity);

/* ---- decoder ---- */
int  orc_decoder_init(orc_decoder *st);
void orc_decoder_destroy(orc_decoder *st);
int  orc_decode_block(orc_decoder *st, float *Dst, const uint8_t *Src);
uint32_t orc_xorshift32(uint32_t s);

/* ---- whole-stream convenience used by tests and bench cpu_baseline ---- */
/* Encodes nBlocks blocks of interleaved pcm for one stream (fresh state).
 * out: slotBytes per block; bits[]: size in bits; wc[]: WindowCtrl; cplx[]: BlockComplexity bits. */
int orc_encode_stream_vbr(int RateHz, int nChan, int BlockSize, const float *pcm, int nBlocks, float Quality,
                          uint8_t *out, int slotBytes, int32_t *bits, int32_t *wc, float *cplx);
int orc_encode_stream_cbr(int RateHz, int nChan, int BlockSize, const float *pcm, int nBlocks, float RateKbps,
                          uint8_t *out, int slotBytes, int32_t *bits, int32_t *wc, float *cplx);
int orc_encode_stream_debug(int mode, int RateHz, int nChan, int BlockSize, const float *pcm, int nBlocks, float p0, float p1,
                            uint8_t *out, int slotBytes, int32_t *bits, int32_t *wc, float *cplx,
                            float *coef, float *noise, float *keys, int32_t *ranks, int32_t *nout);
/* Decodes nBlocks from per-block slots (fresh state, fresh RNG seed). Returns 0 on success, blockIndex+1 of the first corrupt block otherwise. */
int orc_decode_stream(int nChan, int BlockSize, const uint8_t *in, int slotBytes, int nBlocks, float *pcm, int32_t *bitsRead);
int orc_decode_stream_seeded(int nChan, int BlockSize, const uint8_t *in, int slotBytes, int nBlocks, float *pcm, int32_t *bitsRead, uint32_t *seed);
int orc_decode_stream_coefs(int nChan, int BlockSize, const uint8_t *in, int slotBytes, int nBlocks, float *pcm, int32_t *bitsRead, float *coefs);

/* bench.py's cpu_baseline worker loop (orc_bench.c): nThreads POSIX threads over S seeded streams for `seconds` of wall clock */
int orc_bench_threads(int nThreads, int mode, int legs, int RateHz, int nChan, int BS, const float *pcm, const uint8_t *enc, int S, int nBlocks,
                      int slot, float p0, double seconds, double *elapsed, long long *streamsDone);

#ifdef __cplusplus
}
#endif
#endif
