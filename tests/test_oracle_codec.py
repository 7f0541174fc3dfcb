"""Oracle codec checks that need no reference build: known-answer vectors for the shared
inline math, FormatSpecs.md conformance of the decoder on hand-assembled nybble streams
(codes the encoder never emits included), and encode->decode round trips."""
import ctypes as C
import numpy as np
import pytest
from ulc_testlib import oracle, ptr, f32p, i32p, u8p, synth_pcm, oracle_encode_stream, oracle_decode_stream


def test_pattern_table_matches_format_spec():
    """FormatSpecs.md:35-51 — window sizes per decimation code, '*' = overlap-scaled."""
    spec = {2: "N/2*,N/2", 3: "N/2,N/2*", 4: "N/4*,N/4,N/2", 5: "N/4,N/4*,N/2", 6: "N/2,N/4*,N/4", 7: "N/2,N/4,N/4*",
            8: "N/8*,N/8,N/4,N/2", 9: "N/8,N/8*,N/4,N/2", 10: "N/4,N/8*,N/8,N/2", 11: "N/4,N/8,N/8*,N/2",
            12: "N/2,N/8*,N/8,N/4", 13: "N/2,N/8,N/8*,N/4", 14: "N/2,N/4,N/8*,N/8", 15: "N/2,N/4,N/8,N/8*"}
    lib = oracle()
    for code, txt in spec.items():
        pat = lib.orc_decimation_pattern(code << 4)
        got = []
        while pat:
            got.append(f"N/{1 << (pat & 7)}" + ("*" if pat & 8 else ""))
            pat >>= 4
        assert ",".join(got) == txt
        assert sum(1.0 / int(g.strip('*')[2:]) for g in got) == 1.0
    assert lib.orc_decimation_pattern(0x10) == 0x0008


def test_fastlog_and_quantizer_known_answers():
    lib = oracle()
    # FastLog: exponent extraction + quartic; exact at powers of two up to the polynomial's own error
    for x in [1.0, 2.0, 0.5, 1e-10, 3.7, 2.0 ** -126]:
        assert abs(lib.orc_fastlog(x) - np.log(x)) < 2e-4 * max(1.0, abs(np.log(x)))
    # companded quantiser (ulcHelper.h:51-72): rounds sqrt(v) with the rounding point at v = n(n+1)+0.25... i.e. v >= (n+0.5)^2 - ...
    cases = {0.0: 0, 0.49: 0, 0.5: 1, 2.49: 1, 2.5: 2, 6.49: 2, 6.5: 3, 12.5: 4, 49.0: 7, 56.5: 8}
    for v, q in cases.items():
        assert lib.orc_companded_quantize_unsigned(v) == q, v
    # BuildQuantizer (Encode.c:50-87): clamp range and the documented 2/3 rounding rule
    assert lib.orc_build_quantizer(10.0) == 5 and lib.orc_build_quantizer(1e-30) == 31
    for q in range(6, 31):
        hi = np.float32(1.5 * 2.0 ** (4 - q))          # change point ~ 1.5 * 2^(4-q)
        assert lib.orc_build_quantizer(float(hi * np.float32(1.001))) == q
        assert lib.orc_build_quantizer(float(hi * np.float32(0.999))) == q + 1


def test_xorshift_sequence():
    lib = oracle()
    s, seq = 1234567, []
    for _ in range(4):
        s = lib.orc_xorshift32(s); seq.append(s)
    # independent evaluation of xorshift32(13,17,5) (ulcDecoder.c:75-81)
    t, exp = 1234567, []
    for _ in range(4):
        t ^= (t << 13) & 0xFFFFFFFF; t ^= t >> 17; t ^= (t << 5) & 0xFFFFFFFF; exp.append(t)
    assert seq == exp


def test_heapsort_ranks_are_a_descending_key_permutation():
    lib = oracle()
    rng = np.random.default_rng(3)
    keys = rng.normal(0, 5, 4096).astype(np.float32)
    keys[rng.integers(0, 4096, 300)] = -np.inf
    keys[100:110] = keys[200]                              # a finite tie group
    buf = keys.copy()
    tmp = np.zeros(4096, np.int32)
    lib.orc_sort_indices(buf.view(np.int32).ctypes.data_as(i32p), ptr(buf, f32p), ptr(tmp, i32p), 4096)   # aliased like BlockTransform.c:353
    ranks = buf.view(np.int32)
    assert sorted(ranks.tolist()) == list(range(4096))
    order = np.argsort(ranks)
    ks = keys[order]
    assert (ks[:-1] >= ks[1:]).all()                       # rank 0 = largest key


# ---- decoder conformance on hand-assembled streams (FormatSpecs.md:57-141) ----------------
def _pack(nybbles, slot):
    b = np.zeros(slot, np.uint8)
    for i, n in enumerate(nybbles):
        b[i // 2] |= (n & 0xF) << (4 * (i & 1))           # low nybble first
    return b


def _decode_coefs(nybbles, bs=256, ch=1):
    """Decode one block through the oracle and return the dequantised coefficients (CoefDbg tap)."""
    lib = oracle()

    class Dec(C.Structure):
        _fields_ = [("nChan", C.c_int), ("BlockSize", C.c_int), ("LastSubBlockSize", C.c_int), ("Seed", C.c_uint32),
                    ("TransformBuffer", f32p), ("TransformTemp", f32p), ("TransformInvLap", f32p), ("CoefDbg", f32p)]
    d = Dec(); d.nChan = ch; d.BlockSize = bs
    assert lib.orc_decoder_init(C.byref(d)) == 1
    src = _pack(nybbles, 4 * bs * ch)
    out = np.zeros(bs * ch, np.float32)
    lib.orc_decode_block.argtypes = [C.c_void_p, f32p, u8p]
    bits = lib.orc_decode_block(C.byref(d), ptr(out, f32p), ptr(src, u8p))
    coef = np.ctypeslib.as_array(d.CoefDbg, shape=(bs * ch,)).copy()
    lib.orc_decoder_destroy(C.byref(d))
    return bits, coef, out


def test_decoder_syntax_codes():
    bs = 256
    # header 0h (no decimation, scale 0); quantizer nybble 3 -> 2^-(5+3); coefs +2, -7; zeros 0h,4h (5); long zeros 1h,0h,2h (35);
    # quantizer change Fh,1h -> 2^-6; coef +3; extended quantizer Fh,Eh,2h -> 2^-(19+2); coef -2; stop Fh,Eh,Fh
    nyb = [0x0, 0x3, 0x2, 0x9, 0x0, 0x4, 0x1, 0x0, 0x2, 0xF, 0x1, 0x3, 0xF, 0xE, 0x2, 0xE, 0xF, 0xE, 0xF]
    bits, coef, _ = _decode_coefs(nyb, bs)
    assert bits == 4 * len(nyb)
    q = 2.0 ** -8
    exp = np.zeros(bs, np.float32)
    exp[0] = 4 * q; exp[1] = -49 * q
    exp[42] = 9 * 2.0 ** -6
    exp[43] = -4 * 2.0 ** -21
    assert np.array_equal(coef, exp)


def test_opening_Fh_quantizer_is_x86_shift_behaviour():
    """A unit that OPENS with Fh (no encoder writes it, FormatSpecs.md allocates no such code) reaches the reference's
    quantizer expansion with index -2 (ulcDecoder.c:89-98,103-112): a shift by a negative count - undefined in C, 30 on
    x86-64's `shr`, which leaves 0.  So the unit's quantizer is exactly 0.0 until a change code: the behaviour of the
    reference binary on x86, not a format rule; the oracle (count & 31) and the device decoder (index 30) write it out."""
    bs = 256
    # header 0h; opening Fh; coefs +2, -7, +5 (all scaled by 0.0: signed zeros); change Fh,2h -> 2^-(5+2); coef +3; stop Fh,Eh,Fh
    nyb = [0x0, 0xF, 0x2, 0x9, 0x5, 0xF, 0x2, 0x3, 0xF, 0xE, 0xF]
    bits, coef, out = _decode_coefs(nyb, bs)
    assert bits == 4 * len(nyb)
    assert not coef[:3].any(), "quantizer -2 must expand to exactly 0.0"
    assert np.signbit(coef[1]) and not np.signbit(coef[0]), "the coefficients are v * 0.0: the zero keeps the sign of v"
    assert coef[3] == np.float32(9 * 2.0 ** -7) and not coef[4:].any()
    assert np.isfinite(out).all()


def test_decoder_noise_fill_and_tail():
    bs = 256
    # quantizer 0 (2^-5); noise run 8h,Z=0,Y=2,X=5 -> n = (0<<5|2<<1|1)+16 = 21, level ((5>>1)+1)^2 * Q/4 = 9Q/4
    # then tail noise Fh,Fh,Z=3,Y=1,X=0 -> amp (3+1)^2 Q/16 = Q, decay 1 - 2^-19 * 16^2
    nyb = [0x0, 0x0, 0x8, 0x0, 0x2, 0x5, 0xF, 0xF, 0x3, 0x1, 0x0]
    bits, coef, _ = _decode_coefs(nyb, bs)
    assert bits == 4 * len(nyb)
    Q = np.float32(2.0 ** -5)
    lvl = np.float32(9) * Q * np.float32(0.25)
    assert np.allclose(np.abs(coef[:21]), lvl, rtol=0, atol=0)
    # signs follow the xorshift MSB, cumulatively (ulcDecoder.c:156-160)
    s, p, exp = 1234567, float(lvl), []
    for _ in range(21):
        s ^= (s << 13) & 0xFFFFFFFF; s ^= s >> 17; s ^= (s << 5) & 0xFFFFFFFF
        if s & 0x80000000: p = -p
        exp.append(p)
    assert np.array_equal(coef[:21], np.array(exp, np.float32))
    amp = np.float32(16) * Q * np.float32(1.0 / 16)
    r = np.float32(1.0) + np.float32(256) * np.float32(-2.0 ** -19)
    mag = [amp]
    for _ in range(bs - 22): mag.append(np.float32(mag[-1] * r))
    assert np.array_equal(np.abs(coef[21:]), np.array(mag, np.float32))


def test_decoder_rejects_overrunning_runs_and_accepts_silent_block():
    bs = 256
    bits, coef, out = _decode_coefs([0x0, 0xE, 0xF], bs)                 # [Fh,]Eh,Fh: silent channel
    assert bits == 12 and not coef.any() and not out.any()
    nyb = [0x0, 0x0] + [0x1, 0xF, 0xF] + [0x0, 0x0]                      # 288 zeros > 256 remaining -> corrupt
    bits, _, _ = _decode_coefs(nyb, bs)
    assert bits == 0


@pytest.mark.parametrize("code", list(range(2, 16)))
@pytest.mark.parametrize("scale", [0, 3, 7])
def test_decoder_accepts_every_window_code(code, scale):
    """All 14 decimation codes x overlap scales decode (the encoder itself only emits 8..15)."""
    bs = 512
    lib = oracle()
    pat = lib.orc_decimation_pattern(code << 4)
    nsub = 0
    p = pat
    while p: nsub += 1; p >>= 4
    nyb = [0x8 | scale, code] + [0x2, 0x5, 0xF, 0xE, 0xF] * nsub          # each subblock: q=2, one coef, stop
    bits, coef, out = _decode_coefs(nyb, bs)
    assert bits == 4 * len(nyb)
    assert np.count_nonzero(coef) == nsub and np.isfinite(out).all()


@pytest.mark.parametrize("bs,ch,rate,q,minsnr", [(2048, 1, 44100, 50.0, 12.0), (2048, 2, 44100, 80.0, 14.0), (256, 1, 44100, 100.0, 12.0), (4096, 2, 48000, 70.0, 14.0)])
def test_roundtrip_delay_is_two_blocks(bs, ch, rate, q, minsnr):
    nblk = 40
    pcm = synth_pcm(1, nblk * bs, ch, rate, transient=True)
    out, bits, wc, cplx = oracle_encode_stream(pcm, bs, rate, quality=q)
    rc, dec, br = oracle_decode_stream(out, ch, bs)
    assert rc == 0
    assert (br <= bits).all() and (bits - br < 8).all() and (bits % 8 == 0).all()
    assert len(set(wc.tolist())) > 1                                       # window switching exercised
    d = 2 * bs
    x, y = pcm[:-d], dec[d:]
    snr = 10 * np.log10((x ** 2).sum() / ((x - y) ** 2).sum())
    assert snr > minsnr
    # neighbouring alignments are worse: the delay is exactly 2*BlockSize
    x1, y1 = pcm[:-d - 1], dec[d + 1:]
    assert 10 * np.log10((x1 ** 2).sum() / ((x1 - y1) ** 2).sum()) < snr
    x2, y2 = pcm[:-d + 1], dec[d - 1:]
    assert 10 * np.log10((x2 ** 2).sum() / ((x2 - y2) ** 2).sum()) < snr


def test_cbr_never_exceeds_budget():
    bs, ch, rate, kbps = 2048, 2, 48000, 64.0
    pcm = synth_pcm(2, 24 * bs, ch, rate, transient=True)
    out, bits, wc, cplx = oracle_encode_stream(pcm, bs, rate, kbps=kbps)
    budget = int((bs * kbps) * 1000.0 / rate)
    assert (bits[2:] <= budget + 7).all() and bits[2:].mean() > 0.9 * budget


def test_oracle_decodes_hand_assembled_streams_consuming_exactly_their_bits():
    """Streams assembled code by code from FormatSpecs.md:57-141 (every code, including decimation codes 2h-7h,
    extended quantizers, both stop codes): the oracle decoder must accept them and consume exactly the
    assembled number of bits per block."""
    from ulc_testlib import synth_block_stream, oracle_decode_stream
    for bs, ch in ((512, 2), (2048, 2), (1024, 1), (256, 3)):
        slot = 2 * ch * bs + 16
        for s in range(4):
            blk, nbits = synth_block_stream(77 + 13 * s + bs, 6, ch, bs, slot)
            rc, pcm, bits = oracle_decode_stream(blk, ch, bs)
            assert rc == 0, f"bs={bs} ch={ch} stream {s}: rejected at block {rc - 1}"
            assert np.array_equal(bits, nbits)
            assert np.isfinite(pcm).all()
