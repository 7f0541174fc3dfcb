"""The whole-stream cases of the refloop pin (tests/test_oracle_refloop.py, tests/golden/make_refloop_digests.py): the five
shapes of tests/golden/oracle_streams.npz plus one stream of each BASELINE.json configuration, VBR / CBR / ABR between them."""
import ctypes as C
import hashlib
import os
import numpy as np
from ulc_testlib import ORACLE_DIR, synth_pcm, ptr, f32p, i32p, u8p

SITES = ("GetWindowCtrl", "CalculateNoiseLogSpectrum", "CalculatePsychoacoustics", "GetNoiseQ", "GetHFExtParams")
VBR, CBR, ABR = 0, 1, 2
#        tag                 (BlockSize, channels, rate, blocks, stream id, seed, mode, p0, p1)
CASES = {
    # the shapes of oracle_streams.npz (tests/golden/make_golden.py)
    "vbr50_2048s":     (2048, 2, 44100, 20, 21, 9, VBR, 50.0, 0.0),
    "vbr50_2048m":     (2048, 1, 44100, 20, 21, 9, VBR, 50.0, 0.0),
    "cbr64_2048s48k":  (2048, 2, 48000, 20, 21, 9, CBR, 64.0, 0.0),
    "vbr70_4096s":     (4096, 2, 48000, 20, 21, 9, VBR, 70.0, 0.0),
    "vbr90_256m":      (256, 1, 44100, 20, 21, 9, VBR, 90.0, 0.0),
    # BASELINE.json configs[0..4], one stream each
    "cfg0_10s_mono":   (2048, 1, 44100, 216, 100, 1, VBR, 50.0, 0.0),        # 10 s mono 44.1 kHz, VBR -50
    "cfg1_vbr50":      (2048, 2, 44100, 32, 101, 2, VBR, 50.0, 0.0),         # the bench batch's shape: 32 blocks per call
    "cfg3_cbr64_48k":  (2048, 2, 48000, 16, 103, 4, CBR, 64.0, 0.0),
    "cfg3_abr64_48k":  (2048, 2, 48000, 16, 103, 4, ABR, 64.0, 0.35),        # the third rate-control driver (ulcEncoder.c:118-138)
    "cfg4_wswitch":    (4096, 2, 44100, 16, 104, 5, VBR, 50.0, 0.0),         # transient-heavy, BlockSize 4096
    "abr96_1024s":     (1024, 2, 32000, 24, 105, 6, ABR, 96.0, 0.6),
    # the rest of the geometry range the reference accepts (ulcEncoder.c:32-34): every BlockSize, odd and large channel counts
    "vbr60_256s":      (256, 2, 22050, 48, 106, 7, VBR, 60.0, 0.0),
    "cbr48_512m":      (512, 1, 32000, 40, 107, 8, CBR, 48.0, 0.0),
    "vbr40_1024t":     (1024, 3, 48000, 20, 108, 9, VBR, 40.0, 0.0),           # unpaired last channel (no M/S partner)
    "vbr50_8192m":     (8192, 1, 44100, 8, 109, 10, VBR, 50.0, 0.0),
    "cbr128_8192s":    (8192, 2, 48000, 6, 110, 11, CBR, 128.0, 0.0),
    "vbr50_16384s":    (16384, 2, 96000, 5, 111, 12, VBR, 50.0, 0.0),
    "vbr30_32768m":    (32768, 1, 48000, 4, 112, 13, VBR, 30.0, 0.0),
    "vbr70_2048x6":    (2048, 6, 48000, 8, 113, 14, VBR, 70.0, 0.0),           # 5.1: three M/S pairs
    "abr192_4096s":    (4096, 2, 96000, 10, 114, 15, ABR, 192.0, 0.5),
}


def _bind(lib):
    lib.orc_encode_stream_debug.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, f32p, C.c_int, C.c_float, C.c_float,
                                            u8p, C.c_int, i32p, i32p, f32p, f32p, f32p, f32p, i32p, i32p]
    lib.orc_site_reset.argtypes = [C.c_int]
    lib.orc_site_digests.argtypes = [C.POINTER(C.c_uint64), C.POINTER(C.c_int64)]
    return lib


def load_refloop():
    so = os.path.join(ORACLE_DIR, "_ref", "liboracle_refloop.so")
    return _bind(C.CDLL(so)) if os.path.exists(so) else None


def load_pure():
    from ulc_testlib import build_oracle
    return _bind(C.CDLL(build_oracle()))


def run_case(lib, tag):
    bs, ch, rate, nblk, sid, seed, mode, p0, p1 = CASES[tag]
    pcm = synth_pcm(sid, nblk * bs, ch, rate, transient=True, seed=seed)
    slot = 2 * ch * bs + 16
    out = np.zeros((nblk, slot), np.uint8); bits = np.zeros(nblk, np.int32); wc = np.zeros(nblk, np.int32); cplx = np.zeros(nblk, np.float32)
    flat = np.ascontiguousarray(pcm.reshape(-1))
    lib.orc_site_reset(1)
    rc = lib.orc_encode_stream_debug(mode, rate, ch, bs, ptr(flat, f32p), nblk, p0, p1, ptr(out, u8p), slot, ptr(bits, i32p), ptr(wc, i32p),
                                     ptr(cplx, f32p), None, None, None, None, None)
    assert rc == 0, rc
    dig = (C.c_uint64 * 5)(); cnt = (C.c_int64 * 5)()
    lib.orc_site_digests(dig, cnt)
    lib.orc_site_reset(0)
    payload = b"".join(out[k, : bits[k] // 8].tobytes() for k in range(nblk))
    return dict(out=out, bits=bits, wc=wc, cplx=cplx, dig={s: int(dig[i]) for i, s in enumerate(SITES)},
                cnt={s: int(cnt[i]) for i, s in enumerate(SITES)}, sha=hashlib.sha256(payload + bits.tobytes() + wc.tobytes() + cplx.tobytes()).hexdigest(),
                blocks=nblk, bytes=len(payload))
