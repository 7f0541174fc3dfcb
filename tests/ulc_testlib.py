"""Shared test plumbing: ctypes handles for the oracle (checker only), the partial
real-reference build (oracle/_ref, when prebuilt) and a seeded synthetic PCM
generator."""
import ctypes as C
import os
import subprocess
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
f32p = C.POINTER(C.c_float)
f64p = C.POINTER(C.c_double)
i32p = C.POINTER(C.c_int32)
u8p = C.POINTER(C.c_uint8)


def ptr(a, t):
    return a.ctypes.data_as(t)


def build_oracle():
    so = os.path.join(ORACLE_DIR, "liboracle.so")
    srcs = [os.path.join(ORACLE_DIR, f) for f in ("orc_fourier.c", "orc_encoder.c", "orc_decoder.c", "ulc_oracle.h")]
    if (not os.path.exists(so)) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs):
        subprocess.check_call(["make", "-C", ORACLE_DIR, "liboracle.so"], stdout=subprocess.DEVNULL)
    return so


_oracle = None


def oracle():
    global _oracle
    if _oracle is None:
        lib = C.CDLL(build_oracle())
        lib.orc_fastlog.restype = C.c_float
        lib.orc_fastlog.argtypes = [C.c_float]
        lib.orc_companded_quantize_unsigned.argtypes = [C.c_float]
        lib.orc_build_quantizer.argtypes = [C.c_float]
        lib.orc_companded_quantize.argtypes = [C.c_float]
        lib.orc_quant_coef_unsigned.argtypes = [C.c_float, C.c_int]
        lib.orc_quant_coef.argtypes = [C.c_float, C.c_int]
        for f, at in (("orc_freq_to_line", [C.c_float, C.c_float, C.c_uint32]), ("orc_line_to_freq", [C.c_uint32, C.c_float, C.c_uint32]),
                      ("orc_freq_to_bark", [C.c_float]), ("orc_bark_to_freq", [C.c_float])):
            getattr(lib, f).restype = C.c_float; getattr(lib, f).argtypes = at
        lib.orc_get_noise_q.argtypes = [f32p, C.c_int, C.c_int, C.c_float]
        lib.orc_get_hfext_params.argtypes = [f32p, C.c_int, C.c_int, C.c_float, i32p, i32p]
        lib.orc_get_window_ctrl.argtypes = [f32p, f32p, f32p, f32p, C.c_int, C.c_int, C.c_int]
        lib.orc_calc_psychoacoustics.argtypes = [f32p, f32p, C.c_void_p, C.c_int, C.c_int, C.c_uint32]
        lib.orc_calc_noise_log_spectrum.argtypes = [f32p, C.c_void_p, C.c_int, C.c_int]
        lib.orc_mdct_mdst.argtypes = [f32p, f32p, f32p, f32p, f32p, C.c_int, C.c_int]
        lib.orc_imdct.argtypes = [f32p, f32p, f32p, f32p, C.c_int, C.c_int]
        lib.orc_dct4.argtypes = [f32p, f32p, f32p, C.c_int]
        lib.orc_ref64_mdct_mdst.argtypes = [f64p, f64p, f32p, f64p, f64p, C.c_int, C.c_int]
        lib.orc_ref64_imdct_raw.argtypes = [f64p, f32p, C.c_int]
        lib.orc_window_tables.argtypes = [C.c_int, f32p, f32p]
        lib.orc_sort_indices.argtypes = [i32p, f32p, i32p, C.c_int]
        lib.orc_xorshift32.restype = C.c_uint32
        lib.orc_xorshift32.argtypes = [C.c_uint32]
        lib.orc_decimation_pattern.restype = C.c_uint16
        lib.orc_encode_stream_vbr.argtypes = [C.c_int, C.c_int, C.c_int, f32p, C.c_int, C.c_float, u8p, C.c_int, i32p, i32p, f32p]
        lib.orc_encode_stream_cbr.argtypes = lib.orc_encode_stream_vbr.argtypes
        lib.orc_encode_stream_debug.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, f32p, C.c_int, C.c_float, C.c_float,
                                                u8p, C.c_int, i32p, i32p, f32p, f32p, f32p, f32p, i32p, i32p]
        lib.orc_decode_stream.argtypes = [C.c_int, C.c_int, u8p, C.c_int, C.c_int, f32p, i32p]
        lib.orc_decode_stream_coefs.argtypes = [C.c_int, C.c_int, u8p, C.c_int, C.c_int, f32p, i32p, f32p]
        _oracle = lib
    return _oracle


def ref_partial():
    """The real reference's WindowControl/Psyopt/NoiseFill objects (oracle/_ref), or None."""
    so = os.path.join(ORACLE_DIR, "_ref", "libulc_ref_partial.so")
    if not os.path.exists(so):
        if os.path.isdir("/root/reference/libulc"):
            subprocess.check_call(["make", "-C", ORACLE_DIR, "ref"], stdout=subprocess.DEVNULL)
        else:
            return None
    lib = C.CDLL(so)
    lib.ULCi_GetNoiseQ.argtypes = [f32p, C.c_int, C.c_int, C.c_float]
    lib.ULCi_GetHFExtParams.argtypes = [f32p, C.c_int, C.c_int, C.c_float, i32p, i32p]
    lib.ULCi_GetWindowCtrl.argtypes = [f32p, f32p, f32p, f32p, C.c_int, C.c_int, C.c_int]
    lib.ULCi_CalculatePsychoacoustics.argtypes = [f32p, f32p, C.c_void_p, C.c_int, C.c_int, C.c_uint32]
    lib.ULCi_CalculateNoiseLogSpectrum.argtypes = [f32p, C.c_void_p, C.c_int, C.c_int]
    if hasattr(lib, "ref_FastLog"):           # oracle/ref_helper_harness.c: the reference's ulcHelper.h inline helpers
        lib.ref_FastLog.restype = C.c_float; lib.ref_FastLog.argtypes = [C.c_float]
        for f in ("ref_CompandedQuantizeUnsigned", "ref_CompandedQuantize"):
            getattr(lib, f).argtypes = [C.c_float]
        for f in ("ref_CompandedQuantizeCoefficientUnsigned", "ref_CompandedQuantizeCoefficient"):
            getattr(lib, f).argtypes = [C.c_float, C.c_int]
        lib.ref_SubBlockDecimationPattern.restype = C.c_uint
        for f, at in (("ref_FreqToLine", [C.c_float, C.c_float, C.c_uint32]), ("ref_LineToFreq", [C.c_uint32, C.c_float, C.c_uint32]),
                      ("ref_FreqToBark", [C.c_float]), ("ref_BarkToFreq", [C.c_float])):
            getattr(lib, f).restype = C.c_float; getattr(lib, f).argtypes = at
    return lib


# ---------------------------------------------------------------------------
# synthetic PCM: tones + noise (+ decaying bursts), quantised to the
# PCM16 grid and scaled by 2^-15 exactly as a WAV reader would.
# ---------------------------------------------------------------------------
def synth_pcm(stream_id, n_samples, n_chan=2, rate=44100, transient=False, seed=0):
    rng = np.random.default_rng([0x9E3779B9, seed, stream_id])
    t = np.arange(n_samples, dtype=np.float64) / rate
    x = np.zeros(n_samples)
    for _ in range(3):
        f = np.exp(rng.uniform(np.log(80.0), np.log(12000.0)))
        a = rng.uniform(0.05, 0.3)
        x += a * np.sin(2 * np.pi * f * t + rng.uniform(0, 2 * np.pi))
    x += rng.normal(0, 0.02, n_samples)
    if transient:
        pos = 0
        while True:
            pos += int(rng.uniform(0.1, 0.3) * rate) if rate else 0
            if pos >= n_samples:
                break
            ln = min(n_samples - pos, 3000)
            amp = 10 ** rng.uniform(-2.5, -0.3)
            x[pos:pos + ln] += rng.normal(0, amp, ln) * np.exp(-np.arange(ln) / 300.0)
    chans = [x]
    for c in range(1, n_chan):
        d = 7 * c
        y = 0.8 * np.concatenate([np.zeros(d), x[:-d]]) + rng.normal(0, 0.01, n_samples)
        chans.append(y)
    pcm = np.stack(chans, axis=1)
    q = np.clip(np.rint(pcm * 32768.0), -32768, 32767)
    return (q * (1.0 / 32768.0)).astype(np.float32)  # [n_samples][n_chan] interleaved


def oracle_encode_stream(pcm, block_size, rate, quality=None, kbps=None):
    """pcm: [n][C] float32 (n multiple of block_size). Returns (bytes[nBlk][slot], bits, wc, cplx)."""
    lib = oracle()
    n, ch = pcm.shape
    nblk = n // block_size
    slot = 4 * ch * block_size
    out = np.zeros((nblk, slot), np.uint8)
    bits = np.zeros(nblk, np.int32)
    wc = np.zeros(nblk, np.int32)
    cplx = np.zeros(nblk, np.float32)
    flat = np.ascontiguousarray(pcm.reshape(-1))
    if quality is not None:
        rc = lib.orc_encode_stream_vbr(rate, ch, block_size, ptr(flat, f32p), nblk, quality, ptr(out, u8p), slot,
                                       ptr(bits, i32p), ptr(wc, i32p), ptr(cplx, f32p))
    else:
        rc = lib.orc_encode_stream_cbr(rate, ch, block_size, ptr(flat, f32p), nblk, kbps, ptr(out, u8p), slot,
                                       ptr(bits, i32p), ptr(wc, i32p), ptr(cplx, f32p))
    assert rc == 0
    return out, bits, wc, cplx


def oracle_decode_stream(blocks, n_chan, block_size):
    lib = oracle()
    nblk, slot = blocks.shape
    pcm = np.zeros((nblk * block_size, n_chan), np.float32)
    bits = np.zeros(nblk, np.int32)
    rc = lib.orc_decode_stream(n_chan, block_size, ptr(np.ascontiguousarray(blocks), u8p), slot, nblk,
                               ptr(pcm, f32p), ptr(bits, i32p))
    return rc, pcm, bits


def oracle_decode_stream_coefs(blocks, n_chan, block_size):
    """As oracle_decode_stream, plus the dequantised coefficients of every block [nblk][n_chan*block_size]."""
    lib = oracle()
    nblk, slot = blocks.shape
    pcm = np.zeros((nblk * block_size, n_chan), np.float32)
    bits = np.zeros(nblk, np.int32)
    coefs = np.zeros((nblk, n_chan * block_size), np.float32)
    rc = lib.orc_decode_stream_coefs(n_chan, block_size, ptr(np.ascontiguousarray(blocks), u8p), slot, nblk,
                                     ptr(pcm, f32p), ptr(bits, i32p), ptr(coefs, f32p))
    return rc, pcm, bits, coefs


def oracle_encode_debug(pcm, block_size, rate, mode=0, p0=50.0, p1=0.0, slot=None):
    """Full oracle encode of one stream with intermediates. pcm [n][C]."""
    lib = oracle()
    n, ch = pcm.shape
    nblk = n // block_size
    slot = slot or (2 * ch * block_size + 16)
    cb = ch * block_size
    r = dict(out=np.zeros((nblk, slot), np.uint8), bits=np.zeros(nblk, np.int32), wc=np.zeros(nblk, np.int32),
             cplx=np.zeros(nblk, np.float32), coef=np.zeros((nblk, cb), np.float32), noise=np.zeros((nblk, cb), np.float32),
             keys=np.zeros((nblk, cb), np.float32), ranks=np.zeros((nblk, cb), np.int32), nout=np.zeros(nblk, np.int32))
    flat = np.ascontiguousarray(pcm.reshape(-1))
    rc = lib.orc_encode_stream_debug(mode, rate, ch, block_size, ptr(flat, f32p), nblk, p0, p1, ptr(r["out"], u8p), slot,
                                     ptr(r["bits"], i32p), ptr(r["wc"], i32p), ptr(r["cplx"], f32p), ptr(r["coef"], f32p),
                                     ptr(r["noise"], f32p), ptr(r["keys"], f32p), ptr(r["ranks"], i32p), ptr(r["nout"], i32p))
    assert rc == 0, rc
    return r


# ---------------------------------------------------------------------------
# Hand-assembled block streams: every code of FormatSpecs.md:57-141, including the ones the
# encoder never emits (decimation codes 2h-7h at any block size, all overlap scales, extended quantizers,
# long zero runs, both stop codes).  Always valid syntax; the decoder under test and the oracle must agree.
# ---------------------------------------------------------------------------
_PATTERNS = [0x0000, 0x0008, 0x0019, 0x0091, 0x012A, 0x01A2, 0x02A1, 0x0A21,
             0x123B, 0x12B3, 0x13B2, 0x1B32, 0x23B1, 0x2B31, 0x3B21, 0xB321]      # ulcHelper.h:24-46


def _gen_unit(rng, S, nyb, spec_only=False):
    """spec_only: only codes FormatSpecs.md allocates (no opening Fh, extended quantizers Eh,0h..Ch)."""
    N = S
    xmax = 13 if spec_only else 15
    r = rng.random()
    if spec_only and 0.04 <= r < 0.07:
        r = 0.5
    if r < 0.04:
        nyb += [0xE, 0xF]                                  # unit opens with the stop code: all zeros
        return
    if r < 0.07:
        nyb += [0xF]                                       # opening Fh (no encoder writes it): the reference expands quantizer -2 = 0.0
    elif r < 0.3:
        nyb += [0xE, int(rng.integers(0, xmax))]           # extended quantizer Eh,X
    else:
        nyb += [int(rng.integers(0, 14))]
    dense = rng.random() < 0.5
    while N > 0:
        r = rng.random()
        if r < (0.7 if dense else 0.35):
            v = int(rng.integers(2, 8))
            nyb.append(v if rng.random() < 0.5 else 16 - v)
            N -= 1
        elif r < 0.78:
            n = int(rng.integers(1, min(16, N) + 1))
            nyb += [0x0, n - 1]
            N -= n
        elif r < 0.83 and N >= 33:
            n = int(rng.integers(33, min(288, N) + 1))
            v = n - 33
            nyb += [0x1, v >> 4, v & 15]
            N -= n
        elif r < 0.91 and N >= 16:
            n = int(rng.integers(16, min(527, N) + 1))
            v = n - 16
            lvl = int(rng.integers(1, 9))
            nyb += [0x8, (v >> 5) & 15, (v >> 1) & 15, ((lvl - 1) << 1) | (v & 1)]
            N -= n
        elif r < 0.955:
            if rng.random() < 0.7:
                nyb += [0xF, int(rng.integers(0, 14))]
            else:
                nyb += [0xF, 0xE, int(rng.integers(0, xmax))]
        elif r < 0.975:
            nyb += [0xF, 0xE, 0xF]                         # stop: zeros to the end
            return
        elif r < 1.0:
            nyb += [0xF, 0xF, int(rng.integers(0, 16)), int(rng.integers(0, 16)), int(rng.integers(0, 16))]   # noise to the end
            return


def synth_block_stream(seed, n_blocks, n_chan, block_size, slot, spec_only=False):
    """[n_blocks][slot] uint8: random valid blocks, low nybble first (ulcDecoder.c:82-88)."""
    rng = np.random.default_rng(seed)
    out = np.zeros((n_blocks, slot), np.uint8)
    nbits = np.zeros(n_blocks, np.int32)
    for b in range(n_blocks):
        nyb = []
        f = int(rng.integers(0, 16))
        if rng.random() < 0.5:
            f &= 7                                         # un-decimated, overlap scale 0..7
        nyb.append(f)
        pat = _PATTERNS[1]
        if f & 8:
            p = int(rng.integers(2, 16)) if spec_only else int(rng.integers(0, 16))
            nyb.append(p)
            pat = _PATTERNS[p]
        subs = []
        q = pat
        while True:
            subs.append(block_size >> (q & 7))
            q >>= 4
            if not q:
                break
        if subs[0] == block_size:
            subs = subs[:1]                                # ulcDecoder.c:242-245
        for ch in range(n_chan):
            for S in subs:
                _gen_unit(rng, S, nyb, spec_only)
        assert len(nyb) <= 2 * (slot - 4), "block does not fit its slot"
        nbits[b] = 4 * len(nyb)
        if len(nyb) & 1:
            nyb.append(0)
        a = np.array(nyb, np.uint8)
        out[b, :len(a) // 2] = a[0::2] | (a[1::2] << 4)
    return out, nbits
