"""C-ABI boundary without a GPU: libulc_amd.so loads, exports every symbol include/ulc_amd.h
declares, keeps the reference's struct ABI, validates arguments like the reference, fails
loudly (no CPU fallback) when no device exists, and — where /root/reference is mounted —
the reference's own tools compile against the reference's own headers and link against it
unchanged."""
import ctypes as C
import os
import re
import subprocess
import sys
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ulc-codec_amd"))
LIB = os.path.join(ROOT, "ulc-codec_amd", "libulc_amd.so")
REF = "/root/reference"


@pytest.fixture(scope="module")
def lib():
    if not os.path.exists(LIB):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "ulc-codec_amd"), "-j8"], stdout=subprocess.DEVNULL)
    return C.CDLL(LIB)


def test_every_declared_symbol_is_exported(lib):
    hdr = open(os.path.join(ROOT, "include", "ulc_amd.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    names = set(re.findall(r"\b(ULC_\w+|ulcx_\w+)\s*\(", hdr))
    assert len(names) >= 26
    missing = [n for n in sorted(names) if not hasattr(lib, n)]
    assert not missing, missing
    import ulc_amd
    assert set(ulc_amd.EXPORTS) <= names


def test_struct_abi_matches_reference_layout():
    """sizeof/offsets from SURVEY.md §8a T1/T2 (x86-64): 104 and 48 bytes."""
    class Enc(C.Structure):
        _fields_ = [("RateHz", C.c_int), ("nChan", C.c_int), ("BlockSize", C.c_int), ("WindowCtrl", C.c_int), ("NextWindowCtrl", C.c_int),
                    ("BlockComplexity", C.c_float), ("TransientFilter", C.c_float * 3), ("BufferData", C.c_void_p), ("SampleBuffer", C.c_void_p),
                    ("TransformBuffer", C.c_void_p), ("TransformNoise", C.c_void_p), ("TransformFwdLap", C.c_void_p), ("TransformTemp", C.c_void_p),
                    ("TransformIndex", C.c_void_p), ("TransientBuffer", C.c_void_p)]
    class Dec(C.Structure):
        _fields_ = [("nChan", C.c_int), ("BlockSize", C.c_int), ("LastSubBlockSize", C.c_int), ("BufferData", C.c_void_p),
                    ("TransformBuffer", C.c_void_p), ("TransformTemp", C.c_void_p), ("TransformInvLap", C.c_void_p)]
    assert C.sizeof(Enc) == 104 and Enc.BlockComplexity.offset == 20 and Enc.BufferData.offset == 40 and Enc.TransformTemp.offset == 80
    assert C.sizeof(Dec) == 48 and Dec.BufferData.offset == 16 and Dec.TransformInvLap.offset == 40
    # and the C compiler agrees for our own header
    src = '#include <stddef.h>\n#include "ulc_amd.h"\n_Static_assert(sizeof(struct ULC_EncoderState_t)==104,"enc");\n' \
          '_Static_assert(offsetof(struct ULC_EncoderState_t,TransientBuffer)==96,"tb");\n_Static_assert(sizeof(struct ULC_DecoderState_t)==48,"dec");\nint main(void){return 0;}\n'
    p = subprocess.run(["gcc", "-x", "c", "-fsyntax-only", "-I", os.path.join(ROOT, "include"), "-"], input=src.encode(), capture_output=True)
    assert p.returncode == 0, p.stderr.decode()


def test_argument_validation_and_loud_failure_without_gpu(lib):
    lib.ulcx_last_error.restype = C.c_char_p
    h = C.c_void_p()
    # same validation as ulcEncoder.c:32-34 (checked before any device work)
    for (b, c, bs) in [(1, 0, 2048), (1, 256, 2048), (1, 2, 128), (1, 2, 65536), (1, 2, 3000), (0, 2, 2048)]:
        assert lib.ulcx_encoder_create(C.byref(h), 0, b, c, bs, 44100, 1) == -1
        assert lib.ulcx_decoder_create(C.byref(h), 0, b, c, bs, 1) == -1
    # a call without a codec object is refused before anything touches the device (both sample formats)
    for fn in (lib.ulcx_encode_dev, lib.ulcx_encode_dev_pcm16):
        fn.argtypes = [C.c_void_p, C.c_int, C.c_float, C.c_float, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        assert fn(None, 0, 50.0, 0.0, None, 1, None, None, None, None, None) == -1
    for fn in (lib.ulcx_decode_dev, lib.ulcx_decode_dev_pcm16):
        fn.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        assert fn(None, None, 64, 1, None, None, None) == -1
    lib.ulcx_encoder_last_xf_launches.argtypes = [C.c_void_p]
    assert lib.ulcx_encoder_last_xf_launches(None) == 0
    if lib.ulcx_device_count() > 0:
        pytest.skip("GPU present: the no-device path is not reachable here")
    assert lib.ulcx_encoder_create(C.byref(h), 0, 1, 2, 2048, 44100, 1) == -2      # ULCX_ERR_NO_DEVICE, never a CPU fallback
    assert b"device" in lib.ulcx_last_error().lower() or b"hip" in lib.ulcx_last_error().lower()
    assert not h.value


def test_product_does_not_link_or_import_the_oracle():
    out = subprocess.check_output(["ldd", LIB]).decode() + subprocess.check_output(["nm", "-D", LIB]).decode()
    assert "liboracle" not in out and "orc_" not in out
    for dp, _, fs in os.walk(os.path.join(ROOT, "ulc-codec_amd")):
        for f in fs:
            if f.endswith((".c", ".cpp", ".hip", ".h", ".py")) and "build" not in dp:
                txt = open(os.path.join(dp, f), errors="ignore").read()
                assert "ulc_oracle.h" not in txt and "liboracle" not in txt and "import ulc_testlib" not in txt, f


@pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "tools")), reason="reference tree not mounted")
@pytest.mark.parametrize("tool", ["ulcEncodeTool", "ulcDecodeTool"])
def test_reference_tools_link_unchanged(tool, lib, tmp_path):
    """tools/ulcEncodeTool.c / ulcDecodeTool.c compiled from /root/reference against the
    reference's OWN headers, linked against libulc_amd.so instead of libulc + libfourier."""
    exe = tmp_path / tool.lower()
    srcs = [f"{REF}/tools/{tool}.c", f"{REF}/tools/WavIO_Reader.c", f"{REF}/tools/WavIO_Writer.c", f"{REF}/tools/WavIO_Helper.c", f"{REF}/tools/MiniRIFF.c"]
    cmd = ["gcc", "-O2", f"-I{REF}/include", f"-I{REF}/tools", "-o", str(exe)] + srcs + \
          [f"-L{os.path.dirname(LIB)}", "-lulc_amd", f"-Wl,-rpath,{os.path.dirname(LIB)}", "-lm"]
    p = subprocess.run(cmd, capture_output=True)
    assert p.returncode == 0, p.stderr.decode()
    und = subprocess.check_output(["nm", "-u", str(exe)]).decode()
    assert "ULC_" in und and "Fourier_" not in und
    # runs far enough to print its usage text (no GPU needed for that)
    r = subprocess.run([str(exe)], capture_output=True, env=dict(os.environ, LD_LIBRARY_PATH=os.path.dirname(LIB) + ":/opt/rocm/lib"))
    assert b"sage" in r.stdout + r.stderr or r.returncode in (0, 1, 255)


def test_ulc_container_header_roundtrip(lib):
    """24-byte little-endian header, tools/ulc_Helper.h:10-20; RateKbps as ulcEncodeTool.c:173,190."""
    import ulc_amd
    ulc_amd.lib()
    h = ulc_amd.FileHeader(0x32434C55, 2048, 395, 218, 44100, 2, 56, 24)
    buf = (C.c_uint8 * 24)()
    lib.ulcx_ulc_header_pack.argtypes = [C.POINTER(C.c_uint8), C.POINTER(ulc_amd.FileHeader)]
    lib.ulcx_ulc_header_pack(buf, C.byref(h))
    raw = bytes(buf)
    assert raw[:4] == b"ULC2" and raw == bytes(h)                      # same layout as the C struct the tools fwrite
    assert raw[4:8] == (2048).to_bytes(2, "little") + (395).to_bytes(2, "little") and raw[20:] == (24).to_bytes(4, "little")
    g = ulc_amd.FileHeader()
    lib.ulcx_ulc_header_parse.argtypes = [C.POINTER(ulc_amd.FileHeader), C.POINTER(C.c_uint8), C.c_size_t]
    assert lib.ulcx_ulc_header_parse(C.byref(g), buf, 24) == 0 and bytes(g) == raw
    bad = (C.c_uint8 * 24)(*b"RIFF" + bytes(20))
    assert lib.ulcx_ulc_header_parse(C.byref(g), bad, 24) == -1 and lib.ulcx_ulc_header_parse(C.byref(g), buf, 23) == -1
    lib.ulcx_ulc_rate_kbps.argtypes = [C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32]
    assert lib.ulcx_ulc_rate_kbps(70632, 44100, 2048, 218) == round(70632 * 8 * 44100 / 1000 / (2048 * 218))


def test_batched_front_end_is_built_and_prints_usage():
    """ulc-codec_amd/ulcx-tool links against libulc_amd.so only; without arguments it prints its
    usage and touches no GPU."""
    exe = os.path.join(os.path.dirname(LIB), "ulcx-tool")
    assert os.path.exists(exe), "run __graft_entry__.build()"
    r = subprocess.run([exe], capture_output=True, env=dict(os.environ, LD_LIBRARY_PATH=os.path.dirname(LIB) + ":/opt/rocm/lib"))
    assert r.returncode == 1 and b"ulcx-tool encode" in r.stderr
    und = subprocess.check_output(["nm", "-u", exe]).decode()
    assert "ulcx_encode_host" in und and "orc_" not in und


def test_decoder_split_plan_arithmetic(lib):
    """ulcx_dec_split_plan (host arithmetic, round 3): when the synthesis cuts a call's (stream, block) pairs evenly over its
    workgroups instead of giving every stream one.  1536 = resident workgroups of the stereo BlockSize-2048 kernel on an MI355X."""
    f = lib.ulcx_dec_split_plan
    f.argtypes = [C.c_int, C.c_int, C.c_int]; f.restype = C.c_int
    assert f(4096, 32, 1536) == 0 and f(4096, 16, 1536) == 0        # the bench batch: 3 rounds of whole streams against 86 + 1 blocks: not 1.5 x
    assert f(1, 1, 1536) == 0 and f(6, 1, 1536) == 0                  # the drop-in's shape: nothing to cut
    g = f(64, 256, 1536)                                             # few long streams: 16384 pairs, 11 per workgroup
    assert g == 16384 // 11 and g <= 1536
    g = f(16, 512, 1536)
    assert g == 8192 // 8                                            # never fewer than 8 blocks per workgroup (one more is run for the state)
    assert f(3, 40, 1536) == 120 // 8
    for (b, k, r) in [(5, 7, 64), (1000, 9, 100), (2, 3, 8), (1 << 20, 2, 1536)]:
        g = f(b, k, r)
        assert g == 0 or (1 <= g <= r and g != b)
    assert f(0, 4, 16) == 0 and f(4, 0, 16) == 0 and f(4, 4, 0) == 0


def test_decoder_tail_plan_arithmetic(lib):
    """ulcx_dec_tail_plan (host arithmetic, round 5): a batch that runs in rounds of one workgroup per stream keeps its whole
    rounds and cuts only the streams of the partly empty last round, into pieces of 8 blocks (a quarter of a longer call)."""
    f = lib.ulcx_dec_tail_plan
    f.argtypes = [C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int)]; f.restype = C.c_int
    full = C.c_int(-1)
    assert f(4096, 32, 1536, C.byref(full)) == 1024 * 32 // 8 and full.value == 3072     # the bench batch: the last 1024 streams in 4 pieces each
    assert f(2048, 32, 1536, C.byref(full)) == 512 * 4 and full.value == 1536
    assert f(1024, 32, 1536, C.byref(full)) == 1024 * 4 and full.value == 0              # one partly empty round: every stream is in it
    assert f(4096, 256, 1536, C.byref(full)) == 1024 * 4 and full.value == 3072          # a long call: four pieces of 64 blocks
    assert f(4096, 16, 1536, C.byref(full)) == 0 and full.value == 0                     # two pieces per stream do not pay
    assert f(3072, 32, 1536, C.byref(full)) == 0                                         # whole rounds only
    assert f(1536 + 1300, 32, 1536, C.byref(full)) == 0                                  # the last round is more than four fifths full
    assert f(1, 1, 1536, C.byref(full)) == 0 and f(6, 1, 1536, C.byref(full)) == 0      # the drop-in's shape
    assert f(0, 32, 16, None) == 0 and f(4, 0, 16, None) == 0 and f(4, 32, 0, None) == 0
    assert f(100, 33, 64, None) == 36 * 33 // 8
    for (b, k, r) in [(5, 70, 64), (1000, 29, 100), (70, 24, 8), (1 << 20, 40, 1536)]:
        n = f(b, k, r, C.byref(full))
        assert n == 0 or (0 <= full.value < b and full.value % r == 0 and 0 < n <= 4 * r)


def test_block_extent_walk_matches_the_decoder_and_never_overreads(lib):
    """ulcx_block_extent_bytes (host code, what ULC_DecodeBlock stages): the byte count equals what the oracle decoder
    consumed, on hand-assembled streams with every code of the syntax and on encoder output - with every block placed
    flush against an unreadable page, so one byte too many is a fault, not a pass."""
    import mmap
    import numpy as np
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from ulc_testlib import synth_block_stream, synth_pcm, oracle_encode_stream, oracle_decode_stream
    lib.ulcx_block_extent_bytes.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int]
    libc = C.CDLL(None, use_errno=True)
    libc.mprotect.argtypes = [C.c_void_p, C.c_size_t, C.c_int]
    PAGE = mmap.PAGESIZE
    npages = 6
    m = mmap.mmap(-1, (npages + 1) * PAGE)
    base = C.addressof(C.c_char.from_buffer(m))
    assert base % PAGE == 0
    assert libc.mprotect(base + npages * PAGE, PAGE, 0) == 0, "mprotect(PROT_NONE) failed"
    end = base + npages * PAGE
    checked = 0
    cases = []
    for bs, ch in ((512, 2), (2048, 2), (1024, 1), (256, 3)):
        slot = 2 * ch * bs + 16
        for s_ in range(3):
            blocks, bits = synth_block_stream(777 + 13 * s_ + bs, 8, ch, bs, slot)
            cases.append((blocks, bits, ch, bs, slot))
    for bs, ch, rate in ((2048, 2, 44100), (2048, 1, 44100), (4096, 2, 48000)):
        pcm = synth_pcm(3, 8 * bs, ch, rate, transient=True, seed=bs)
        out, bits, wc, cplx = oracle_encode_stream(pcm, bs, rate, quality=60.0)
        cases.append((out, bits, ch, bs, 2 * ch * bs + 16))
    for blocks, bits, ch, bs, slot in cases:
        rc, pcm_, rbits = oracle_decode_stream(np.ascontiguousarray(blocks), ch, bs)
        assert rc == 0
        for k in range(blocks.shape[0]):
            nbytes = (int(rbits[k]) + 7) // 8                    # bits consumed are nybble granular: a half-used byte is read
            assert nbytes <= npages * PAGE
            dst = end - nbytes
            C.memmove(dst, blocks[k].ctypes.data, nbytes)
            got = lib.ulcx_block_extent_bytes(C.c_void_p(dst), ch, bs, slot)
            assert got == nbytes, f"block {k} (bs {bs}, ch {ch}): walk says {got} bytes, the decoder consumed {nbytes}"
            checked += 1
    assert checked > 100
    # a corrupt block (a zero run past the end of its subblock) ends where the reference's decoder gives up, and the bound holds
    bad = np.array([0x00, 0x1F, 0xFF], np.uint8)               # header 0h (N/1), quantizer 0, then 1h,Fh,Fh = 288 zeros... at BlockSize 256
    C.memmove(end - 3, bad.ctypes.data, 3)
    assert lib.ulcx_block_extent_bytes(C.c_void_p(end - 3), 1, 256, 3) <= 3
    libc.mprotect(base + npages * PAGE, PAGE, 3)
    del m
