"""CPU: the oracle built with -mavx2 (-O2, no contraction; and -O3 -mavx2 -mfma -ffp-contract=off) beside the default
-O2 build: are the encoded bytes and decoded samples identical, and what is the single-thread rate?  (DESIGN.md §8 /
bench.py's cpu_baseline.sample refer to this.)  Writes nothing but its report."""
import ctypes as C
import os, subprocess, sys, tempfile, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from ulc_testlib import synth_pcm, ptr, f32p, u8p, i32p

SRC = [os.path.join(ROOT, "oracle", f) for f in ("orc_fourier.c", "orc_encoder.c", "orc_decoder.c")]
BUILDS = {"-O2 (shipped)": ["-O2"], "-O2 -mavx2": ["-O2", "-mavx2"], "-O3 -mavx2 -mfma": ["-O3", "-mavx2", "-mfma"]}
bs, ch, rate, nblk = 2048, 2, 44100, 400
pcm = synth_pcm(1, nblk * bs, ch, rate, transient=True, seed=5)
flat = np.ascontiguousarray(pcm.reshape(-1))
slot = 2 * ch * bs + 16
ref = None
with tempfile.TemporaryDirectory() as td:
    for name, fl in BUILDS.items():
        so = os.path.join(td, "o%d.so" % len(os.listdir(td)))
        subprocess.check_call(["gcc", *fl, "-fPIC", "-ffp-contract=off", "-shared", "-o", so, *SRC, "-lm"])
        o = C.CDLL(so)
        out = np.zeros((nblk, slot), np.uint8); bits = np.zeros(nblk, np.int32); dp = np.zeros(nblk * bs * ch, np.float32)
        te = td_ = 1e9
        for rep in range(3):
            out[:] = 0
            t0 = time.perf_counter(); o.orc_encode_stream_vbr(rate, ch, bs, ptr(flat, f32p), nblk, C.c_float(50.0), ptr(out, u8p), slot, ptr(bits, i32p), None, None); te = min(te, time.perf_counter() - t0)
            t0 = time.perf_counter(); o.orc_decode_stream(ch, bs, ptr(out, u8p), slot, nblk, ptr(dp, f32p), None); td_ = min(td_, time.perf_counter() - t0)
        if ref is None: ref = (out.copy(), bits.copy(), dp.copy())
        same = np.array_equal(out, ref[0]) and np.array_equal(bits, ref[1]) and np.array_equal(dp, ref[2])
        ms = nblk * bs * ch / 1e6
        print(f"{name:20s} encode {ms / te:7.2f} Msamples/s  decode {ms / td_:7.2f} Msamples/s  one thread; output identical to -O2: {same}")
