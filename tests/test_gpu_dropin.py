"""The drop-in boundary end to end (BASELINE configs[0]): the reference's OWN tools
(tools/ulcEncodeTool.c, tools/ulcDecodeTool.c, WAV reader/writer - compiled unchanged from /root/reference by
oracle/Makefile into oracle/_ref/, linked against libulc_amd.so instead of libulc + libfourier) run on the GPU box;
the .ulc file they write must be byte-identical to the container assembled from the oracle's blocks, and the WAV
they decode must be sample-identical to the oracle's decode."""
import os
import struct
import subprocess
import sys
import wave
import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ulc-codec_amd"))
from ulc_testlib import synth_pcm, oracle_encode_debug, oracle_decode_stream

pytestmark = pytest.mark.gpu
ENC = os.path.join(ROOT, "oracle", "_ref", "ulcencodetool_amd")
DEC = os.path.join(ROOT, "oracle", "_ref", "ulcdecodetool_amd")
needs_tools = pytest.mark.skipif(not (os.path.exists(ENC) and os.path.exists(DEC)),
                                 reason="oracle/_ref tools not built (needs /root/reference at build time)")


def _write_wav16(path, pcm16, rate):
    with wave.open(str(path), "wb") as w:
        w.setnchannels(pcm16.shape[1]); w.setsampwidth(2); w.setframerate(rate)
        w.writeframes(pcm16.astype("<i2").tobytes())


def _expected_ulc(pcm16, rate, bs, mode, p0):
    n, ch = pcm16.shape
    nblk = (n + bs - 1) // bs + 2                                   # ulcEncodeTool.c:93-98
    x = np.zeros((nblk * bs, ch), np.float32)
    x[:n] = pcm16.astype(np.float32) * np.float32(2.0 ** -15)        # WavIO_Helper.c:49-55
    ref = oracle_encode_debug(x, bs, rate, mode, p0, slot=2 * ch * bs + 16)
    sizes = (ref["bits"] + 7) // 8
    payload = b"".join(ref["out"][k, :sizes[k]].tobytes() for k in range(nblk))
    total = int(sizes.sum())
    kbps = int(np.rint(total * 8.0 * rate / 1000.0 / (bs * nblk)))    # ulcEncodeTool.c:173,190 (lrint)
    hdr = struct.pack("<IHHIIHHI", 0x32434C55, bs, int(sizes.max()), nblk, rate, ch, kbps, 24)   # tools/ulc_Helper.h:10-20
    return hdr + payload, ref, nblk


def _run(cmd):
    env = dict(os.environ)
    env["LD_LIBRARY_PATH"] = os.path.join(ROOT, "ulc-codec_amd") + ":/opt/rocm/lib:" + env.get("LD_LIBRARY_PATH", "")
    p = subprocess.run(cmd, capture_output=True, env=env, timeout=600)
    assert p.returncode == 0, f"{cmd[0]} failed ({p.returncode}): {p.stdout.decode()[-400:]} {p.stderr.decode()[-400:]}"
    return p


@needs_tools
@pytest.mark.parametrize("ch,rate,seconds,arg,mode,p0,bs", [
    (1, 44100, 10.0, "-50", 0, 50.0, 2048),        # BASELINE configs[0]: 10 s mono 44.1 kHz PCM16, BlockSize 2048, VBR -50
    (2, 44100, 2.0, "-50", 0, 50.0, 2048),
    (2, 48000, 1.5, "64", 1, 64.0, 2048),          # CBR 64 kbps
    (2, 44100, 1.5, "-70", 0, 70.0, 4096),
])
def test_reference_tools_over_libulc_amd_write_the_oracles_bytes(ch, rate, seconds, arg, mode, p0, bs, tmp_path):
    n = int(seconds * rate)
    pcm = synth_pcm(3, n, ch, rate, transient=True, seed=11)
    pcm16 = np.clip(np.rint(pcm * 32767.0), -32768, 32767).astype(np.int16)
    wav_in, ulc, wav_f32, wav_16 = tmp_path / "in.wav", tmp_path / "out.ulc", tmp_path / "dec_f32.wav", tmp_path / "dec_16.wav"
    _write_wav16(wav_in, pcm16, rate)
    args = [ENC, str(wav_in), str(ulc), arg] + ([f"-blocksize:{bs}"] if bs != 2048 else [])
    _run(args)
    want, ref, nblk = _expected_ulc(pcm16, rate, bs, mode, p0)
    got = open(ulc, "rb").read()
    assert got[:24] == want[:24], f"container header differs: {got[:24].hex()} vs {want[:24].hex()}"
    assert len(got) == len(want), f".ulc size {len(got)} != {len(want)}"
    assert got == want, "the .ulc bytes written by ulcencodetool over libulc_amd.so differ from the oracle's"
    # decode with the reference's decode tool (float32 and the default PCM16)
    rc, ref_pcm, _ = oracle_decode_stream(ref["out"], ch, bs)
    assert rc == 0
    _run([DEC, str(ulc), str(wav_f32), "-format:FLOAT32"])
    raw = open(wav_f32, "rb").read()
    data = np.frombuffer(raw[-nblk * bs * ch * 4:], dtype="<f4").reshape(nblk * bs, ch)
    assert np.array_equal(data.view(np.uint32), ref_pcm.view(np.uint32)), "decoded float samples differ from the oracle's"
    _run([DEC, str(ulc), str(wav_16)])
    raw = open(wav_16, "rb").read()
    d16 = np.frombuffer(raw[-nblk * bs * ch * 2:], dtype="<i2").reshape(nblk * bs, ch)
    want16 = np.rint(np.clip(ref_pcm * np.float32(32768.0), -32768.0, 32767.0)).astype(np.int16)   # WavIO_Helper.c:57-63
    assert np.array_equal(d16, want16), "decoded PCM16 differs from the oracle's"


TOOL = os.path.join(ROOT, "ulc-codec_amd", "ulcx-tool")


@needs_tools
@pytest.mark.skipif(not os.path.exists(TOOL), reason="ulc-codec_amd/ulcx-tool not built")
@pytest.mark.parametrize("arg,fmt", [("-50", "FLOAT32"), ("64", "PCM16")])
def test_batched_front_end_matches_the_reference_tools_file_for_file(arg, fmt, tmp_path):
    """ulcx-tool encodes and decodes MANY files per call; every file it writes must be
    byte-identical to what the reference's one-file-per-process tools write for the same input."""
    rate, ch = 44100, 2
    lens = [1.3, 0.7, 2.05, 0.31, 1.0]                       # different lengths in one batch
    ins = []
    for i, sec in enumerate(lens):
        pcm = synth_pcm(20 + i, int(sec * rate), ch, rate, transient=(i & 1) == 0, seed=5)
        p = tmp_path / f"in{i}.wav"
        _write_wav16(p, np.clip(np.rint(pcm * 32767.0), -32768, 32767).astype(np.int16), rate)
        ins.append(p)
    refdir, gotdir = tmp_path / "ref", tmp_path / "got"
    refdir.mkdir(); gotdir.mkdir()
    _run([TOOL, "encode", str(gotdir), arg] + [str(p) for p in ins])
    for p in ins:
        _run([ENC, str(p), str(refdir / (p.stem + ".ulc")), arg])
        assert open(gotdir / (p.stem + ".ulc"), "rb").read() == open(refdir / (p.stem + ".ulc"), "rb").read(), \
            f"{p.name}: batched encode differs from ulcencodetool"
    ulcs = [gotdir / (p.stem + ".ulc") for p in ins]
    wavdir = tmp_path / "wav"; wavdir.mkdir()
    _run([TOOL, "decode", str(wavdir), f"-format:{fmt}"] + [str(u) for u in ulcs])
    for u in ulcs:
        ref_wav = refdir / (u.stem + ".wav")
        _run([DEC, str(u), str(ref_wav), f"-format:{fmt}"])
        got = open(wavdir / (u.stem + ".wav"), "rb").read()
        ref = open(ref_wav, "rb").read()
        nbytes = struct.unpack("<I", open(u, "rb").read()[8:12])[0] * 2048 * ch * (4 if fmt == "FLOAT32" else 2)
        assert got[-nbytes:] == ref[-nbytes:], f"{u.name}: batched decode differs from ulcdecodetool"


@pytest.mark.skipif(not os.path.exists(TOOL), reason="ulc-codec_amd/ulcx-tool not built")
@pytest.mark.parametrize("ndev", [2, 3, 8])
def test_batched_front_end_devices_option_does_not_change_the_files(ndev, tmp_path):
    """SURVEY.md §8(e) below bench.py: `ulcx-tool -devices:N` deals the inputs round-robin over N groups, each with its own
    encoder / decoder and its own host thread (group g on device g % visible: on the 1-GPU box the groups share device 0).
    Every file must be the one the single-group run writes."""
    rate, ch = 44100, 2
    ins = []
    for i, sec in enumerate([0.9, 0.4, 1.3, 0.25, 0.7, 1.1, 0.5]):
        pcm = synth_pcm(60 + i, int(sec * rate), ch, rate, transient=(i % 3) != 1, seed=13)
        p = tmp_path / f"d{i}.wav"
        _write_wav16(p, np.clip(np.rint(pcm * 32767.0), -32768, 32767).astype(np.int16), rate)
        ins.append(p)
    one, many = tmp_path / "one", tmp_path / "many"
    one.mkdir(); many.mkdir()
    _run([TOOL, "encode", str(one), "-45"] + [str(p) for p in ins])
    _run([TOOL, "encode", str(many), "-45", f"-devices:{ndev}"] + [str(p) for p in ins])
    for p in ins:
        assert open(many / (p.stem + ".ulc"), "rb").read() == open(one / (p.stem + ".ulc"), "rb").read(), f"{p.name}: -devices:{ndev} changed the encoded file"
    w1, wn = tmp_path / "w1", tmp_path / "wn"
    w1.mkdir(); wn.mkdir()
    ulcs = [str(one / (p.stem + ".ulc")) for p in ins]
    _run([TOOL, "decode", str(w1)] + ulcs)
    _run([TOOL, "decode", str(wn), f"-devices:{ndev}"] + ulcs)
    for p in ins:
        assert open(wn / (p.stem + ".wav"), "rb").read() == open(w1 / (p.stem + ".wav"), "rb").read(), f"{p.name}: -devices:{ndev} changed the decoded file"


@needs_tools
@pytest.mark.skipif(not os.path.exists(TOOL), reason="ulc-codec_amd/ulcx-tool not built")
def test_abr_workflow_analyse_then_encode_matches_the_reference_tools(tmp_path):
    """SURVEY.md §8f rank 3, the ABR usage of ulcEncodeTool.c:157-188: a first run reports the stream's average
    complexity, a second run passes it back as "kbps,complexity".  Both numbers printed and both files written by the
    batched front end must equal the reference tool's (linked over libulc_amd)."""
    import re
    rate, ch = 44100, 2
    ins = []
    for i, sec in enumerate([1.1, 0.6, 1.7]):
        pcm = synth_pcm(40 + i, int(sec * rate), ch, rate, transient=(i != 1), seed=9)
        p = tmp_path / f"abr{i}.wav"
        _write_wav16(p, np.clip(np.rint(pcm * 32767.0), -32768, 32767).astype(np.int16), rate)
        ins.append(p)
    d1 = tmp_path / "pass1"; d1.mkdir()
    out = _run([TOOL, "encode", str(d1), "-50"] + [str(p) for p in ins]).stdout.decode()
    got = {m.group(1): m.group(2) for m in re.finditer(r"^(\S+?): .*avg complexity ([0-9.]+)$", out, re.M)}
    refdir, gotdir = tmp_path / "ref", tmp_path / "got"
    refdir.mkdir(); gotdir.mkdir()
    for p in ins:
        ref_out = _run([ENC, str(p), str(refdir / "a.ulc"), "-50"]).stdout.decode()
        ref_c = re.search(r"Avg complexity = ([0-9.]+)", ref_out).group(1)
        assert got[p.name] == ref_c, f"{p.name}: average complexity {got[p.name]} != reference tool's {ref_c}"
        arg = f"64,{ref_c}"
        _run([ENC, str(p), str(refdir / (p.stem + ".ulc")), arg])
        sub = gotdir / p.stem; sub.mkdir()
        _run([TOOL, "encode", str(sub), arg, str(p)])
        assert open(sub / (p.stem + ".ulc"), "rb").read() == open(refdir / (p.stem + ".ulc"), "rb").read(), \
            f"{p.name}: ABR file differs from ulcencodetool's"


def test_decode_block_reads_exactly_the_block_and_matches_the_oracle():
    """ULC_DecodeBlock through the drop-in ABI with every block flush against an unreadable page (the reference reads
    SrcBuffer only as far as the block runs, ulcDecoder.h:54): one byte too many is a fault.  Output = oracle decode."""
    import ctypes as C
    import mmap
    from ulc_testlib import oracle_encode_stream
    lib = C.CDLL(os.path.join(ROOT, "ulc-codec_amd", "libulc_amd.so"))

    class Dec(C.Structure):
        _fields_ = [("nChan", C.c_int), ("BlockSize", C.c_int), ("LastSubBlockSize", C.c_int), ("BufferData", C.c_void_p),
                    ("TransformBuffer", C.c_void_p), ("TransformTemp", C.c_void_p), ("TransformInvLap", C.c_void_p)]
    lib.ULC_DecodeBlock.argtypes = [C.POINTER(Dec), C.POINTER(C.c_float), C.c_void_p]
    libc = C.CDLL(None, use_errno=True)
    libc.mprotect.argtypes = [C.c_void_p, C.c_size_t, C.c_int]
    PAGE = mmap.PAGESIZE
    m = mmap.mmap(-1, 5 * PAGE)
    base = C.addressof(C.c_char.from_buffer(m))
    assert libc.mprotect(base + 4 * PAGE, PAGE, 0) == 0
    end = base + 4 * PAGE
    bs, ch, rate, nblk = 2048, 2, 44100, 10
    pcm = synth_pcm(9, nblk * bs, ch, rate, transient=True, seed=4)
    out, bits, wc, cplx = oracle_encode_stream(pcm, bs, rate, quality=50.0)
    rc, ref_pcm, ref_bits = oracle_decode_stream(out, ch, bs)
    assert rc == 0
    st = Dec(); st.nChan = ch; st.BlockSize = bs
    assert lib.ULC_DecoderState_Init(C.byref(st)) == 1
    got = np.zeros((nblk, bs * ch), np.float32)
    for k in range(nblk):
        nbytes = (int(ref_bits[k]) + 7) // 8
        dst = end - nbytes
        C.memmove(dst, out[k].ctypes.data, nbytes)
        r = lib.ULC_DecodeBlock(C.byref(st), got[k].ctypes.data_as(C.POINTER(C.c_float)), C.c_void_p(dst))
        assert r == ref_bits[k], f"block {k}: {r} bits consumed, oracle {ref_bits[k]}"
    lib.ULC_DecoderState_Destroy(C.byref(st))
    assert np.array_equal(got.reshape(nblk * bs, ch).view(np.uint32), ref_pcm.view(np.uint32))
    libc.mprotect(base + 4 * PAGE, PAGE, 3)


def test_state_fields_the_reference_updates_are_mirrored_back():
    """The reference leaves WindowCtrl / NextWindowCtrl / TransientFilter (ulcEncoder_BlockTransform.c:116-125,
    ulcEncoder_WindowControl.c:88-89,131) and LastSubBlockSize (ulcDecoder.c:300) in the caller's struct after every call;
    so does the drop-in (the values are the device-resident state's, fetched with the block)."""
    import ctypes as C
    from ulc_testlib import _PATTERNS
    lib = C.CDLL(os.path.join(ROOT, "ulc-codec_amd", "libulc_amd.so"))

    class Enc(C.Structure):
        _fields_ = [("RateHz", C.c_int), ("nChan", C.c_int), ("BlockSize", C.c_int), ("WindowCtrl", C.c_int), ("NextWindowCtrl", C.c_int),
                    ("BlockComplexity", C.c_float), ("TransientFilter", C.c_float * 3), ("BufferData", C.c_void_p), ("SampleBuffer", C.c_void_p),
                    ("TransformBuffer", C.c_void_p), ("TransformNoise", C.c_void_p), ("TransformFwdLap", C.c_void_p), ("TransformTemp", C.c_void_p),
                    ("TransformIndex", C.c_void_p), ("TransientBuffer", C.c_void_p)]

    class Dec(C.Structure):
        _fields_ = [("nChan", C.c_int), ("BlockSize", C.c_int), ("LastSubBlockSize", C.c_int), ("BufferData", C.c_void_p),
                    ("TransformBuffer", C.c_void_p), ("TransformTemp", C.c_void_p), ("TransformInvLap", C.c_void_p)]
    f32p = C.POINTER(C.c_float)
    lib.ULC_EncodeBlock_VBR.restype = C.c_void_p
    lib.ULC_EncodeBlock_VBR.argtypes = [C.POINTER(Enc), f32p, C.POINTER(C.c_int), C.c_float]
    lib.ULC_DecodeBlock.argtypes = [C.POINTER(Dec), f32p, C.c_void_p]
    for ch, bs in ((2, 2048), (1, 1024)):
        rate, nblk = 44100, 14
        pcm = synth_pcm(7, nblk * bs, ch, rate, transient=True, seed=2)
        ref = oracle_encode_debug(pcm, bs, rate, 0, 50.0)
        e = Enc(); e.RateHz = rate; e.nChan = ch; e.BlockSize = bs
        assert lib.ULC_EncoderState_Init(C.byref(e)) == 1
        d = Dec(); d.nChan = ch; d.BlockSize = bs
        assert lib.ULC_DecoderState_Init(C.byref(d)) == 1
        size = C.c_int()
        out = np.zeros(bs * ch, np.float32)
        tf_prev = None
        for k in range(nblk):
            src = np.ascontiguousarray(pcm[k * bs:(k + 1) * bs].reshape(-1))
            p = lib.ULC_EncodeBlock_VBR(C.byref(e), src.ctypes.data_as(f32p), C.byref(size), 50.0)
            assert size.value == ref["bits"][k]
            assert C.string_at(p, size.value // 8) == ref["out"][k, :size.value // 8].tobytes()
            assert e.WindowCtrl == ref["wc"][k], f"call {k}: WindowCtrl {e.WindowCtrl:#x}, oracle {ref['wc'][k]:#x}"
            if k + 1 < nblk:
                assert e.NextWindowCtrl == ref["wc"][k + 1], f"call {k}: NextWindowCtrl {e.NextWindowCtrl:#x}, oracle's next block {ref['wc'][k + 1]:#x}"
            assert e.BlockComplexity == ref["cplx"][k]
            tf = tuple(e.TransientFilter)
            assert all(np.isfinite(tf)) and (k == 0 or tf != tf_prev), "TransientFilter is the running filter state: it moves with the signal"
            tf_prev = tf
            blk = C.string_at(p, size.value // 8) + bytes(16)
            used = lib.ULC_DecodeBlock(C.byref(d), out.ctypes.data_as(f32p), blk)        # bits consumed: nybble-granular (ulcDecodeTool.c:153 rounds up)
            assert 0 < used <= size.value and (used + 7) // 8 * 8 == size.value
            pat = _PATTERNS[(e.WindowCtrl >> 4) & 15]
            last = bs
            while True:
                last = bs >> (pat & 7)
                if last == bs: break                        # ulcDecoder.c:242-245
                pat >>= 4
                if not pat: break
            assert d.LastSubBlockSize == last, f"call {k}: LastSubBlockSize {d.LastSubBlockSize}, expected {last} for WindowCtrl {e.WindowCtrl:#x}"
        lib.ULC_EncoderState_Destroy(C.byref(e)); lib.ULC_DecoderState_Destroy(C.byref(d))


def test_two_files_decoded_by_one_process_share_the_noise_chain():
    """The reference's noise generator is a function-static word (ulcDecoder.c:75-81): ONE xorshift32 chain per process,
    not re-seeded by ULC_DecoderState_Init.  A process that decodes a second file through the drop-in therefore continues
    the first file's chain.  Expected = the oracle run the same way (state handed from one decode to the next); the file
    has noise fill (VBR -50 codes it), so a fresh seed for the second file would differ."""
    import ctypes as C
    import subprocess
    import textwrap
    code = textwrap.dedent(r"""
        import ctypes as C, os, sys
        import numpy as np
        sys.path.insert(0, os.path.join(%r, "tests"))
        from ulc_testlib import synth_pcm, oracle_encode_stream, oracle, ptr, u8p, f32p, i32p
        lib = C.CDLL(os.path.join(%r, "ulc-codec_amd", "libulc_amd.so"))
        class Dec(C.Structure):
            _fields_ = [("nChan", C.c_int), ("BlockSize", C.c_int), ("LastSubBlockSize", C.c_int), ("BufferData", C.c_void_p),
                        ("TransformBuffer", C.c_void_p), ("TransformTemp", C.c_void_p), ("TransformInvLap", C.c_void_p)]
        lib.ULC_DecodeBlock.argtypes = [C.POINTER(Dec), C.POINTER(C.c_float), C.c_void_p]
        orc = oracle()
        orc.orc_decode_stream_seeded.argtypes = [C.c_int, C.c_int, u8p, C.c_int, C.c_int, f32p, i32p, C.POINTER(C.c_uint32)]
        seed = C.c_uint32(1234567)
        fresh_differs = 0
        for f, (ch, bs, nblk) in enumerate(((2, 2048, 6), (1, 1024, 8), (2, 2048, 5))):
            pcm = synth_pcm(20 + f, nblk * bs, ch, 44100, transient=True, seed=f)
            out, bits, wc, cplx = oracle_encode_stream(pcm, bs, 44100, quality=50.0)
            out = np.ascontiguousarray(out)
            before = seed.value
            ref = np.zeros((nblk * bs, ch), np.float32); rb = np.zeros(nblk, np.int32)
            assert orc.orc_decode_stream_seeded(ch, bs, ptr(out, u8p), out.shape[1], nblk, ptr(ref, f32p), ptr(rb, i32p), C.byref(seed)) == 0
            if f > 0:
                fr = np.zeros_like(ref); s2 = C.c_uint32(1234567)
                orc.orc_decode_stream_seeded(ch, bs, ptr(out, u8p), out.shape[1], nblk, ptr(fr, f32p), ptr(rb, i32p), C.byref(s2))
                fresh_differs += int(not np.array_equal(fr.view(np.uint32), ref.view(np.uint32)))
                assert before != 1234567, "the earlier file drew no noise: the test input is wrong"
            st = Dec(); st.nChan = ch; st.BlockSize = bs
            assert lib.ULC_DecoderState_Init(C.byref(st)) == 1
            got = np.zeros((nblk, bs * ch), np.float32)
            for k in range(nblk):
                blk = out[k].tobytes() + bytes(16)
                assert lib.ULC_DecodeBlock(C.byref(st), got[k].ctypes.data_as(C.POINTER(C.c_float)), blk) == rb[k]
            lib.ULC_DecoderState_Destroy(C.byref(st))
            assert np.array_equal(got.reshape(nblk * bs, ch).view(np.uint32), ref.view(np.uint32)), f"file {f}: PCM differs from the oracle's process-wide chain"
        assert fresh_differs == 2, "a fresh seed per file gives the same PCM: the test does not discriminate"
        print("ok")
    """) % (ROOT, ROOT)
    # a fresh interpreter: the process-wide word must start at 1234567, whatever other tests of this session decoded before
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and r.stdout.strip().endswith("ok"), r.stdout + r.stderr
