"""ctypes binding of libulc_amd.so (include/ulc_amd.h).  Plumbing only: no compute
happens here and there is no fallback — if the shared library or a GPU is missing the
constructors raise."""
import ctypes as C
import os
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# ULC_AMD_LIB: another build of the same library (A/B timing runs, tools/ab_all.sh); the default is the in-tree one
LIB_PATH = os.environ.get("ULC_AMD_LIB") or os.path.join(_HERE, "libulc_amd.so")

MODE_VBR, MODE_CBR, MODE_ABR = 0, 1, 2
_f32p = C.POINTER(C.c_float)
_i32p = C.POINTER(C.c_int32)
_u8p = C.POINTER(C.c_uint8)

EXPORTS = [
    "ULC_EncoderState_Init", "ULC_EncoderState_Destroy", "ULC_EncodeBlock_CBR", "ULC_EncodeBlock_ABR",
    "ULC_EncodeBlock_VBR", "ULC_DecoderState_Init", "ULC_DecoderState_Destroy", "ULC_DecodeBlock",
    "ulcx_last_error", "ulcx_device_count", "ulcx_encoder_create", "ulcx_encoder_destroy", "ulcx_encoder_reset",
    "ulcx_encoder_slot_bytes", "ulcx_encode_dev", "ulcx_encode_dev_pcm16", "ulcx_encode_host", "ulcx_encoder_debug_fetch",
    "ulcx_decoder_create", "ulcx_decoder_destroy", "ulcx_decoder_reset", "ulcx_decode_dev", "ulcx_decode_dev_pcm16", "ulcx_decode_host",
    "ulcx_encoder_last_fallbacks", "ulcx_encoder_debug_force_exact", "ulcx_ulc_header_pack", "ulcx_ulc_header_parse", "ulcx_ulc_rate_kbps",
    "ulcx_pack_streams_dev", "ulcx_decode_packed_dev", "ulcx_decode_packed_host", "ulcx_decoder_upload_payload", "ulcx_decode_resident_host", "ulcx_encoder_stage_ms", "ulcx_encoder_stage_name", "ulcx_encoder_last_xf_launches", "ulcx_decoder_stage_ms", "ulcx_decoder_stage_name", "ulcx_block_extent_bytes", "ulcx_encoder_set_timing", "ulcx_decoder_set_timing", "ulcx_encode_block1", "ulcx_decode_block1", "ulcx_decode_block1_rng", "ulcx_build_rev", "ulcx_dec_split_plan", "ulcx_dec_tail_plan", "ulcx_decoder_last_cut",
]



class FileHeader(C.Structure):
    """tools/ulc_Helper.h:10-20 (24 bytes)."""
    _fields_ = [("Magic", C.c_uint32), ("BlockSize", C.c_uint16), ("MaxBlockSize", C.c_uint16), ("nBlocks", C.c_uint32),
                ("RateHz", C.c_uint32), ("nChan", C.c_uint16), ("RateKbps", C.c_uint16), ("StreamOffs", C.c_uint32)]


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(f"{LIB_PATH} not built: run `make -C ulc-codec_amd` (or __graft_entry__.build())")
        l = C.CDLL(LIB_PATH)
        l.ulcx_last_error.restype = C.c_char_p
        l.ulcx_build_rev.restype = C.c_char_p
        l.ulcx_encoder_create.argtypes = [C.POINTER(C.c_void_p), C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]
        l.ulcx_encoder_destroy.argtypes = [C.c_void_p]
        l.ulcx_encoder_reset.argtypes = [C.c_void_p]
        l.ulcx_encoder_slot_bytes.argtypes = [C.c_void_p]
        l.ulcx_encode_dev.argtypes = [C.c_void_p, C.c_int, C.c_float, C.c_float, C.c_void_p, C.c_int,
                                      C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        l.ulcx_encode_dev_pcm16.argtypes = l.ulcx_encode_dev.argtypes
        l.ulcx_encode_host.argtypes = [C.c_void_p, C.c_int, C.c_float, C.c_float, _f32p, C.c_int, _u8p, _i32p, _i32p, _f32p]
        l.ulcx_encoder_debug_fetch.argtypes = [C.c_void_p, C.c_int, _f32p, _f32p, _f32p, _u8p, _i32p]
        l.ulcx_decoder_create.argtypes = [C.POINTER(C.c_void_p), C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]
        l.ulcx_decoder_destroy.argtypes = [C.c_void_p]
        l.ulcx_decoder_reset.argtypes = [C.c_void_p]
        l.ulcx_decode_dev.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        l.ulcx_decode_dev_pcm16.argtypes = l.ulcx_decode_dev.argtypes
        l.ulcx_decode_host.argtypes = [C.c_void_p, _u8p, C.c_int, C.c_int, _f32p, _i32p]
        l.ulcx_encoder_last_fallbacks.argtypes = [C.c_void_p]
        l.ulcx_decoder_last_cut.argtypes = [C.c_void_p, _i32p, _i32p, _i32p]
        l.ulcx_dec_tail_plan.argtypes = [C.c_int, C.c_int, C.c_int, _i32p]
        l.ulcx_encoder_debug_force_exact.argtypes = [C.c_void_p, C.c_int]
        l.ulcx_decode_packed_dev.argtypes = [C.c_void_p, C.c_void_p, C.c_longlong, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        l.ulcx_decode_packed_host.argtypes = [C.c_void_p, _u8p, C.c_longlong, _i32p, C.c_int, _f32p, _i32p]
        l.ulcx_encoder_set_timing.argtypes = [C.c_void_p, C.c_int]
        l.ulcx_decoder_set_timing.argtypes = [C.c_void_p, C.c_int]
        l.ulcx_decoder_upload_payload.argtypes = [C.c_void_p, _u8p, C.c_longlong, _i32p]
        l.ulcx_decode_resident_host.argtypes = [C.c_void_p, C.c_int, _f32p, _i32p]
        l.ulcx_pack_streams_dev.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_longlong,
                                            C.c_void_p, C.c_void_p, C.c_void_p]
        l.ulcx_ulc_header_pack.argtypes = [_u8p, C.POINTER(FileHeader)]
        l.ulcx_ulc_header_parse.argtypes = [C.POINTER(FileHeader), _u8p, C.c_size_t]
        l.ulcx_ulc_rate_kbps.argtypes = [C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32]
        l.ulcx_encoder_stage_ms.argtypes = [C.c_void_p, _f32p, C.c_int]
        l.ulcx_encoder_last_xf_launches.argtypes = [C.c_void_p]
        l.ulcx_decoder_stage_ms.argtypes = [C.c_void_p, _f32p, C.c_int]
        l.ulcx_block_extent_bytes.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int]
        l.ulcx_encoder_stage_name.restype = C.c_char_p
        l.ulcx_decoder_stage_name.restype = C.c_char_p
        _lib = l
    return _lib


class UlcError(RuntimeError):
    pass


def build_rev():
    """Revision of the sources the loaded library was built from (sha1 prefix, ulc-codec_amd/Makefile)."""
    return lib().ulcx_build_rev().decode()


def _check(rc, what):
    if rc != 0:
        raise UlcError(f"{what} failed ({rc}): {lib().ulcx_last_error().decode()}")


def _p(a, t):
    return a.ctypes.data_as(t) if a is not None else None


class BatchEncoder:
    """B independent streams; encode(pcm[B][K*BS][C]) -> (bytes[B][K][slot], bits[B][K], wc[B][K], cplx[B][K])."""

    def __init__(self, n_streams, n_chan, block_size, rate_hz, max_blocks, device=0):
        self.B, self.C, self.BS, self.rate, self.maxK = n_streams, n_chan, block_size, rate_hz, max_blocks
        self.h = C.c_void_p()
        _check(lib().ulcx_encoder_create(C.byref(self.h), device, n_streams, n_chan, block_size, rate_hz, max_blocks),
               "ulcx_encoder_create")
        self.slot = lib().ulcx_encoder_slot_bytes(self.h)
        self.lastK = 0

    def close(self):
        if self.h:
            lib().ulcx_encoder_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def reset(self):
        _check(lib().ulcx_encoder_reset(self.h), "ulcx_encoder_reset")

    def encode(self, pcm, mode=MODE_VBR, p0=50.0, p1=0.0):
        pcm = np.ascontiguousarray(pcm, dtype=np.float32)
        assert pcm.shape[0] == self.B and pcm.shape[-1] == self.C
        K = pcm.shape[1] // self.BS
        assert pcm.shape[1] == K * self.BS
        out = np.zeros((self.B, K, self.slot), np.uint8)
        bits = np.zeros((self.B, K), np.int32)
        wc = np.zeros((self.B, K), np.int32)
        cplx = np.zeros((self.B, K), np.float32)
        _check(lib().ulcx_encode_host(self.h, mode, p0, p1, _p(pcm, _f32p), K, _p(out, _u8p), _p(bits, _i32p),
                                      _p(wc, _i32p), _p(cplx, _f32p)), "ulcx_encode_host")
        self.lastK = K
        return out, bits, wc, cplx

    def encode_dev(self, d_pcm, n_blocks, d_out, d_bits, d_wc=0, d_cplx=0, mode=MODE_VBR, p0=50.0, p1=0.0, stream=0):
        """Device-pointer path (ints / .data_ptr()); asynchronous on `stream`."""
        _check(lib().ulcx_encode_dev(self.h, mode, p0, p1, d_pcm, n_blocks, d_out, d_bits, d_wc or None, d_cplx or None,
                                     stream or None), "ulcx_encode_dev")
        self.lastK = n_blocks

    def encode_dev_pcm16(self, d_pcm16, n_blocks, d_out, d_bits, d_wc=0, d_cplx=0, mode=MODE_VBR, p0=50.0, p1=0.0, stream=0):
        """PCM16 ingest: d_pcm16 is a device pointer to int16 [B][K][BS][C]; converted on load as tools/WavIO_Helper.c:49-55."""
        _check(lib().ulcx_encode_dev_pcm16(self.h, mode, p0, p1, d_pcm16, n_blocks, d_out, d_bits, d_wc or None, d_cplx or None,
                                           stream or None), "ulcx_encode_dev_pcm16")
        self.lastK = n_blocks

    def debug_fetch(self, K=None):
        K = K or self.lastK
        n = self.C * self.BS
        coef = np.zeros((self.B, K, n), np.float32)
        noise = np.zeros((self.B, K, n), np.float32)
        keys = np.zeros((self.B, K, n), np.float32)
        keep = np.zeros((self.B, K, n), np.uint8)
        nout = np.zeros((self.B, K), np.int32)
        _check(lib().ulcx_encoder_debug_fetch(self.h, K, _p(coef, _f32p), _p(noise, _f32p), _p(keys, _f32p),
                                              _p(keep, _u8p), _p(nout, _i32p)), "ulcx_encoder_debug_fetch")
        return dict(coef=coef, noise=noise, keys=keys, keep=keep, nout=nout)

    def last_fallbacks(self):
        return lib().ulcx_encoder_last_fallbacks(self.h)

    def force_exact(self, every):
        _check(lib().ulcx_encoder_debug_force_exact(self.h, int(every)), "ulcx_encoder_debug_force_exact")

    def xf_launches(self):
        return int(lib().ulcx_encoder_last_xf_launches(self.h))

    def set_timing(self, on):
        _check(lib().ulcx_encoder_set_timing(self.h, int(bool(on))), "ulcx_encoder_set_timing")

    def stage_ms(self):
        ms = np.zeros(32, np.float32)
        n = lib().ulcx_encoder_stage_ms(self.h, _p(ms, _f32p), 32)
        return {lib().ulcx_encoder_stage_name(i).decode(): float(ms[i]) for i in range(n)}


class BatchDecoder:
    def __init__(self, n_streams, n_chan, block_size, max_blocks, device=0):
        self.B, self.C, self.BS, self.maxK = n_streams, n_chan, block_size, max_blocks
        self.h = C.c_void_p()
        _check(lib().ulcx_decoder_create(C.byref(self.h), device, n_streams, n_chan, block_size, max_blocks),
               "ulcx_decoder_create")

    def close(self):
        if self.h:
            lib().ulcx_decoder_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def last_cut(self):
        """(workgroups of the last call's synthesis or 0 = one per stream, leading whole-stream workgroups, resident workgroups)"""
        g, f, r = C.c_int32(0), C.c_int32(0), C.c_int32(0)
        _check(lib().ulcx_decoder_last_cut(self.h, C.byref(g), C.byref(f), C.byref(r)), "ulcx_decoder_last_cut")
        return g.value, f.value, r.value

    def reset(self):
        _check(lib().ulcx_decoder_reset(self.h), "ulcx_decoder_reset")

    def decode(self, blocks):
        """blocks: uint8 [B][K][slot] -> (pcm [B][K*BS][C], bits [B][K])."""
        blocks = np.ascontiguousarray(blocks, dtype=np.uint8)
        B, K, slot = blocks.shape
        assert B == self.B
        pcm = np.zeros((B, K * self.BS, self.C), np.float32)
        bits = np.zeros((B, K), np.int32)
        _check(lib().ulcx_decode_host(self.h, _p(blocks, _u8p), slot, K, _p(pcm, _f32p), _p(bits, _i32p)), "ulcx_decode_host")
        return pcm, bits

    def decode_packed(self, payload, payload_bytes, n_blocks):
        """payload: uint8 [B][stride] contiguous blocks per stream (a .ulc file's data section each);
        continues from where the previous call stopped."""
        payload = np.ascontiguousarray(payload, dtype=np.uint8)
        nbytes = np.ascontiguousarray(payload_bytes, dtype=np.int32)
        B, stride = payload.shape
        pcm = np.zeros((B, n_blocks * self.BS, self.C), np.float32)
        bits = np.zeros((B, n_blocks), np.int32)
        _check(lib().ulcx_decode_packed_host(self.h, _p(payload, _u8p), stride, _p(nbytes, _i32p), n_blocks, _p(pcm, _f32p), _p(bits, _i32p)),
               "ulcx_decode_packed_host")
        return pcm, bits

    def upload_payload(self, payload, payload_bytes):
        """Packed payloads to the device once (rewinds the read positions); decode_resident() then walks them."""
        payload = np.ascontiguousarray(payload, dtype=np.uint8)
        nbytes = np.ascontiguousarray(payload_bytes, dtype=np.int32)
        _check(lib().ulcx_decoder_upload_payload(self.h, _p(payload, _u8p), payload.shape[1], _p(nbytes, _i32p)), "ulcx_decoder_upload_payload")

    def decode_resident(self, n_blocks):
        pcm = np.zeros((self.B, n_blocks * self.BS, self.C), np.float32)
        bits = np.zeros((self.B, n_blocks), np.int32)
        _check(lib().ulcx_decode_resident_host(self.h, n_blocks, _p(pcm, _f32p), _p(bits, _i32p)), "ulcx_decode_resident_host")
        return pcm, bits

    def decode_dev(self, d_in, slot, n_blocks, d_pcm, d_bits, stream=0):
        _check(lib().ulcx_decode_dev(self.h, d_in, slot, n_blocks, d_pcm, d_bits, stream or None), "ulcx_decode_dev")

    def decode_dev_pcm16(self, d_in, slot, n_blocks, d_pcm16, d_bits, stream=0):
        """PCM16 output: d_pcm16 is a device pointer to int16 [B][K][BS][C]; converted on store as tools/WavIO_Helper.c:56-63."""
        _check(lib().ulcx_decode_dev_pcm16(self.h, d_in, slot, n_blocks, d_pcm16, d_bits, stream or None), "ulcx_decode_dev_pcm16")

    def set_timing(self, on):
        _check(lib().ulcx_decoder_set_timing(self.h, int(bool(on))), "ulcx_decoder_set_timing")

    def stage_ms(self):
        ms = np.zeros(8, np.float32)
        n = lib().ulcx_decoder_stage_ms(self.h, _p(ms, _f32p), 8)
        return {lib().ulcx_decoder_stage_name(i).decode(): float(ms[i]) for i in range(n)}
