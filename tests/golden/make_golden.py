#!/usr/bin/env python3
"""Generates tests/golden/*.npz.  Run in the build container (needs /root/reference for
the partial real-reference build oracle/_ref/libulc_ref_partial.so).

ref_units.npz    inputs + outputs of the REAL reference functions that compile on their own
                 (ULCi_GetWindowCtrl, ULCi_CalculatePsychoacoustics,
                 ULCi_CalculateNoiseLogSpectrum, ULCi_GetNoiseQ, ULCi_GetHFExtParams) on seeded
                 inputs — data only; the oracle must reproduce them bit for bit
                 (tests/test_golden.py), also where the reference tree is absent.
oracle_streams.npz  SHA-256 of whole encoded streams produced by the oracle (regression lock
                 for the unpinned parts; not reference-derived).
"""
import hashlib
import os
import sys
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from ulc_testlib import ref_partial, synth_pcm, ptr, f32p, i32p, oracle_encode_stream  # noqa: E402


def chan_major_ms(pcm, bs, k):
    n, ch = pcm.shape
    def blk(i):
        if i < 0:
            return np.zeros((ch, bs), np.float32)
        b = pcm[i * bs:(i + 1) * bs].T.copy()
        for c in range(1, ch, 2):
            l, r = b[c - 1].copy(), b[c].copy()
            b[c - 1] = (l + r) * np.float32(0.5)
            b[c] = (l - r) * np.float32(0.5)
        return b
    return np.ascontiguousarray(np.concatenate([blk(k - 1).reshape(-1), blk(k).reshape(-1)]))


def helper_inputs():
    """Seeded float32 inputs for the ulcHelper.h helpers: every binade the codec can see, exact quantiser
    decision points n^2 - n + 0.5 and their float neighbours, denormals, zeros, both signs."""
    rng = np.random.default_rng(2718)
    xs = [np.float32(0.0), np.float32(-0.0), np.float32(2.0 ** -126), np.float32(2.0 ** -149), np.float32(1.0), np.float32(0.5), np.float32(0.25)]
    for e in range(-140, 40):
        xs += list((rng.uniform(1.0, 2.0, 6) * 2.0 ** e).astype(np.float32))
    for n in range(0, 20):
        t = np.float32(n * n - n + 0.5) if n else np.float32(0.5)
        xs += [t, np.nextafter(t, np.float32(0)), np.nextafter(t, np.float32(1e9))]
    xs = np.array(xs, np.float32)
    sign = np.where(rng.integers(0, 2, xs.size) == 1, np.float32(-1), np.float32(1)).astype(np.float32)
    return (xs * sign).astype(np.float32)


def main():
    REF = ref_partial()
    assert REF is not None, "needs /root/reference (or a prebuilt oracle/_ref)"
    out = {}
    # --- window control: per-call WindowCtrl + carried state for a few stream shapes
    for tag, (bs, ch, rate) in {"wc_2048s": (2048, 2, 44100), "wc_256m": (256, 1, 44100), "wc_4096s": (4096, 2, 48000)}.items():
        nblk = 24
        pcm = synth_pcm(11, nblk * bs, ch, rate, transient=True, seed=bs)
        tb = np.zeros(32, np.float32); tf = np.zeros(3, np.float32); tmp = np.zeros(2 * bs, np.float32)
        wcs, tfs, tbs = [], [], []
        for k in range(nblk):
            sb = chan_major_ms(pcm, bs, k)
            wcs.append(REF.ULCi_GetWindowCtrl(ptr(sb, f32p), ptr(tb, f32p), ptr(tf, f32p), ptr(tmp, f32p), bs, ch, rate))
            tfs.append(tf.copy()); tbs.append(tb.copy())
        out[tag + "_wc"] = np.array(wcs, np.int32); out[tag + "_tf"] = np.array(tfs); out[tag + "_tb"] = np.array(tbs)
    # --- psychoacoustics / noise spectrum
    rng = np.random.default_rng(1)
    for bs, rate in ((2048, 44100), (512, 48000)):
        for wc in (0x10, 0x8B, 0xDA, 0x3B):
            amp = (rng.normal(0, 1, bs // 2) ** 2 * 10 ** rng.uniform(-12, 0, bs // 2)).astype(np.float32)
            a = amp.copy(); m = np.zeros(bs // 2, np.float32); t = np.zeros(64, np.float32)
            REF.ULCi_CalculatePsychoacoustics(ptr(m, f32p), ptr(a, f32p), t.ctypes.data, bs, rate, wc)
            out[f"psy_{bs}_{rate}_{wc:02x}_in"] = amp; out[f"psy_{bs}_{rate}_{wc:02x}_out"] = m
    for n, rate in ((2048, 44100), (256, 44100), (1024, 48000)):
        d = np.zeros(n, np.float32)
        d[: n // 2] = (rng.normal(0, 1, n // 2) ** 2 * 10 ** rng.uniform(-14, 0, n // 2)).astype(np.float32)
        o = d.copy(); t = np.zeros(n + 64, np.float32)
        REF.ULCi_CalculateNoiseLogSpectrum(ptr(o, f32p), t.ctypes.data, n, rate)
        out[f"nls_{n}_{rate}_in"] = d; out[f"nls_{n}_{rate}_out"] = o
    # --- noise-fill parameters on a realistic pair array
    pairs = out["nls_2048_44100_out"]
    q_in, q_out, h_out = [], [], []
    for _ in range(200):
        band = int(rng.integers(0, 2048 - 16)); cnt = int(rng.integers(16, min(527, 2048 - band) + 1)); q = float(2.0 ** rng.integers(5, 20))
        q_in.append((band, cnt, q))
        q_out.append(REF.ULCi_GetNoiseQ(ptr(pairs, f32p), band, cnt, q))
        a, b = np.zeros(1, np.int32), np.zeros(1, np.int32)
        REF.ULCi_GetHFExtParams(ptr(pairs, f32p), band, 2048 - band, q, ptr(a, i32p), ptr(b, i32p))
        h_out.append((a[0], b[0]))
    out["nf_in"] = np.array(q_in, np.float64); out["nf_q"] = np.array(q_out, np.int32); out["nf_hf"] = np.array(h_out, np.int32)
    # --- ulcHelper.h inline helpers through oracle/ref_helper_harness.c (the reference's own header, compiled in place)
    import ctypes as C
    REF.ref_FastLog.restype = C.c_float; REF.ref_FastLog.argtypes = [C.c_float]
    for f in ("ref_CompandedQuantizeUnsigned", "ref_CompandedQuantize"):
        getattr(REF, f).argtypes = [C.c_float]
    for f in ("ref_CompandedQuantizeCoefficientUnsigned", "ref_CompandedQuantizeCoefficient"):
        getattr(REF, f).argtypes = [C.c_float, C.c_int]
    REF.ref_SubBlockDecimationPattern.restype = C.c_uint
    for f, at in (("ref_FreqToLine", [C.c_float, C.c_float, C.c_uint32]), ("ref_LineToFreq", [C.c_uint32, C.c_float, C.c_uint32]),
                  ("ref_FreqToBark", [C.c_float]), ("ref_BarkToFreq", [C.c_float])):
        getattr(REF, f).restype = C.c_float; getattr(REF, f).argtypes = at
    x = helper_inputs()
    out["hlp_x"] = x
    out["hlp_fastlog"] = np.array([REF.ref_FastLog(float(v)) for v in x], np.float32)
    out["hlp_qu"] = np.array([REF.ref_CompandedQuantizeUnsigned(float(abs(v))) for v in x], np.int32)
    out["hlp_q"] = np.array([REF.ref_CompandedQuantize(float(v)) for v in x], np.int32)
    out["hlp_qc7"] = np.array([REF.ref_CompandedQuantizeCoefficient(float(v), 7) for v in x], np.int32)
    out["hlp_qcu8"] = np.array([REF.ref_CompandedQuantizeCoefficientUnsigned(float(abs(v)), 8) for v in x], np.int32)
    out["hlp_qcu16"] = np.array([REF.ref_CompandedQuantizeCoefficientUnsigned(float(abs(v)), 16) for v in x], np.int32)
    out["hlp_pattern"] = np.array([REF.ref_SubBlockDecimationPattern(w << 4) for w in range(16)], np.uint32)
    lines = []
    for n, rate in ((1024, 44100), (128, 48000), (2048, 48000), (16, 32000)):
        nyq = np.float32(rate) * np.float32(0.5)
        for line in range(n):
            f = REF.ref_LineToFreq(line, float(nyq), n)
            b = REF.ref_FreqToBark(f)
            lines.append((n, rate, line, f, b))
    out["hlp_line"] = np.array(lines, np.float64)
    barks = []
    for rate in (44100, 48000):
        nyq = np.float32(rate) * np.float32(0.5)
        for bi in np.arange(-1.0, 27.0, 0.25, dtype=np.float32):
            f = REF.ref_BarkToFreq(float(bi))
            barks.append((rate, float(bi), f, REF.ref_FreqToLine(f, float(nyq), 1024)))
    out["hlp_bark"] = np.array(barks, np.float64)
    np.savez_compressed(os.path.join(HERE, "ref_units.npz"), **out)

    # --- oracle whole-stream regression hashes
    hs = {}
    for tag, (bs, ch, rate, kw) in {"vbr50_2048s": (2048, 2, 44100, dict(quality=50.0)), "vbr50_2048m": (2048, 1, 44100, dict(quality=50.0)),
                                    "cbr64_2048s48k": (2048, 2, 48000, dict(kbps=64.0)), "vbr70_4096s": (4096, 2, 48000, dict(quality=70.0)),
                                    "vbr90_256m": (256, 1, 44100, dict(quality=90.0))}.items():
        pcm = synth_pcm(21, 20 * bs, ch, rate, transient=True, seed=9)
        o, bits, wc, cplx = oracle_encode_stream(pcm, bs, rate, **kw)
        payload = b"".join(o[k, : bits[k] // 8].tobytes() for k in range(len(bits)))
        hs[tag] = np.frombuffer(hashlib.sha256(payload).digest(), np.uint8)
        hs[tag + "_bits"] = bits; hs[tag + "_wc"] = wc; hs[tag + "_cplx"] = cplx
    np.savez_compressed(os.path.join(HERE, "oracle_streams.npz"), **hs)
    print("wrote golden fixtures")


if __name__ == "__main__":
    main()
