/* ref_helper_harness.c — TEST INFRASTRUCTURE.  A translation unit of our own that #includes the
 * reference's libulc/ulcHelper.h WHERE IT LIES under /root/reference (never copied) and exports
 * its always-inline helpers under C names, so that tests/test_oracle_pinned.py can call the real
 * FastLog / ULCi_CompandedQuantize* / ULCi_SubBlockDecimationPattern / Bark-line helpers
 * (ulcHelper.h:24-136) and compare the oracle's restatement with them bit for bit.
 * Built only by oracle/Makefile's `ref` target into oracle/_ref/libulc_ref_partial.so. */
#include "ulcHelper.h"

float    ref_FastLog(float x)                                        { return FastLog(x); }
int      ref_CompandedQuantizeUnsigned(float v)                      { return ULCi_CompandedQuantizeUnsigned(v); }
int      ref_CompandedQuantize(float v)                              { return ULCi_CompandedQuantize(v); }
int      ref_CompandedQuantizeCoefficientUnsigned(float v, int lim)  { return ULCi_CompandedQuantizeCoefficientUnsigned(v, lim); }
int      ref_CompandedQuantizeCoefficient(float v, int lim)          { return ULCi_CompandedQuantizeCoefficient(v, lim); }
unsigned ref_SubBlockDecimationPattern(int WindowCtrl)               { return ULCi_SubBlockDecimationPattern(WindowCtrl); }
float    ref_FreqToLine(float fHz, float NyquistHz, uint32_t N)      { return ULCi_FreqToLine(fHz, NyquistHz, N); }
float    ref_LineToFreq(uint32_t Line, float NyquistHz, uint32_t N)  { return ULCi_LineToFreq(Line, NyquistHz, N); }
float    ref_FreqToBark(float fHz)                                   { return ULCi_FreqToBark(fHz); }
float    ref_BarkToFreq(float Bark)                                  { return ULCi_BarkToFreq(Bark); }
