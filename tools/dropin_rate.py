"""GPU: single-stream rate of the drop-in ABI (one block per call, host pointers: ULC_EncodeBlock_VBR / ULC_DecodeBlock
of libulc_amd.so) beside the C oracle on one host core, same stream.  Numbers for INTEGRATION.md."""
import ctypes as C
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from ulc_testlib import synth_pcm, oracle, ptr, f32p, u8p, i32p

lib = C.CDLL(os.path.join(ROOT, "ulc-codec_amd", "libulc_amd.so"))


class Enc(C.Structure):
    _fields_ = [("RateHz", C.c_int), ("nChan", C.c_int), ("BlockSize", C.c_int), ("WindowCtrl", C.c_int), ("NextWindowCtrl", C.c_int),
                ("BlockComplexity", C.c_float), ("TransientFilter", C.c_float * 3), ("BufferData", C.c_void_p), ("SampleBuffer", C.c_void_p),
                ("TransformBuffer", C.c_void_p), ("TransformNoise", C.c_void_p), ("TransformFwdLap", C.c_void_p), ("TransformTemp", C.c_void_p),
                ("TransformIndex", C.c_void_p), ("TransientBuffer", C.c_void_p)]


class Dec(C.Structure):
    _fields_ = [("nChan", C.c_int), ("BlockSize", C.c_int), ("LastSubBlockSize", C.c_int), ("BufferData", C.c_void_p),
                ("TransformBuffer", C.c_void_p), ("TransformTemp", C.c_void_p), ("TransformInvLap", C.c_void_p)]


lib.ULC_EncodeBlock_VBR.restype = C.c_void_p
lib.ULC_EncodeBlock_VBR.argtypes = [C.POINTER(Enc), f32p, C.POINTER(C.c_int), C.c_float]
lib.ULC_DecodeBlock.argtypes = [C.POINTER(Dec), f32p, C.c_void_p]
for ch in (1, 2):
    bs, rate, nblk = 2048, 44100, 200
    pcm = synth_pcm(1, nblk * bs, ch, rate, transient=True, seed=3)
    e = Enc(); e.RateHz = rate; e.nChan = ch; e.BlockSize = bs
    assert lib.ULC_EncoderState_Init(C.byref(e)) == 1
    d = Dec(); d.nChan = ch; d.BlockSize = bs
    assert lib.ULC_DecoderState_Init(C.byref(d)) == 1
    blocks = []
    size = C.c_int()
    for warm in range(2):
        t0 = time.perf_counter()
        for k in range(nblk):
            src = np.ascontiguousarray(pcm[k * bs:(k + 1) * bs].reshape(-1))
            p = lib.ULC_EncodeBlock_VBR(C.byref(e), ptr(src, f32p), C.byref(size), 50.0)
            if warm: blocks.append(C.string_at(p, size.value // 8) + bytes(16))
        te = time.perf_counter() - t0
    out = np.zeros(bs * ch, np.float32)
    t0 = time.perf_counter()
    for b in blocks:
        lib.ULC_DecodeBlock(C.byref(d), ptr(out, f32p), b)
    td = time.perf_counter() - t0
    orc = oracle(); slot = 2 * ch * bs + 16
    o = np.zeros((nblk, slot), np.uint8); bits = np.zeros(nblk, np.int32); dp = np.zeros(nblk * bs * ch, np.float32)
    flat = np.ascontiguousarray(pcm.reshape(-1))
    t0 = time.perf_counter(); orc.orc_encode_stream_vbr(rate, ch, bs, ptr(flat, f32p), nblk, 50.0, ptr(o, u8p), slot, ptr(bits, i32p), None, None); toe = time.perf_counter() - t0
    t0 = time.perf_counter(); orc.orc_decode_stream(ch, bs, ptr(o, u8p), slot, nblk, ptr(dp, f32p), None); tod = time.perf_counter() - t0
    ms = nblk * bs * ch / 1e6
    print(f"ch={ch} BlockSize={bs}: drop-in encode {te / nblk * 1e3:.3f} ms/block = {ms / te:.2f} Msamples/s, decode {td / nblk * 1e3:.3f} ms/block = {ms / td:.2f} Msamples/s; "
          f"oracle on one core: encode {toe / nblk * 1e3:.3f} ms/block = {ms / toe:.2f} Msamples/s, decode {tod / nblk * 1e3:.3f} ms/block = {ms / tod:.2f} Msamples/s")
    lib.ULC_EncoderState_Destroy(C.byref(e)); lib.ULC_DecoderState_Destroy(C.byref(d))
