#!/usr/bin/env python3
"""Randomised decoder sweep on hand-assembled block streams (every code of the syntax, tests/ulc_testlib.py
synth_block_stream) and on corrupted copies of them: GPU vs oracle, PCM and consumed bits.  python tools/fuzz_decode.py [seconds]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ulc-codec_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import ulc_amd
from ulc_testlib import synth_block_stream, oracle_decode_stream

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 3)
t0 = time.time(); n = 0; nblk = 0; ncorrupt = 0
while time.time() - t0 < budget:
    bs = int(rng.choice([256, 512, 1024, 2048, 4096]))
    ch = int(rng.choice([1, 2, 2, 3]))
    B = int(rng.integers(1, 12)); K = int(rng.integers(1, 6)); calls = int(rng.integers(1, 4))
    if rng.random() < 0.25:                       # calls long enough for the synthesis to cut streams into pieces (round 5: ulcx_dec_tail_plan)
        B = int(rng.integers(1, 6)); K = int(rng.integers(24, 41)); calls = int(rng.integers(1, 3))
    slot = 2 * ch * bs + 16
    streams = [synth_block_stream(int(rng.integers(0, 1 << 30)), calls * K, ch, bs, slot)[0] for _ in range(B)]
    blocks = np.stack(streams)
    corrupt = rng.random() < 0.3
    if corrupt:                                   # flip random bytes inside the used part of a few blocks
        for _ in range(int(rng.integers(1, 6))):
            s = int(rng.integers(0, B)); k = int(rng.integers(0, calls * K)); pos = int(rng.integers(0, 64))
            blocks[s, k, pos] ^= np.uint8(rng.integers(1, 256))
        ncorrupt += 1
    dec = ulc_amd.BatchDecoder(B, ch, bs, K)
    got = []; gbits = []
    for c in range(calls):
        p, b = dec.decode(blocks[:, c * K:(c + 1) * K]); got.append(p); gbits.append(b)
    got = np.concatenate(got, axis=1); gbits = np.concatenate(gbits, axis=1)
    dec.close()
    for s in range(B):
        rc, rp, rb = oracle_decode_stream(blocks[s], ch, bs)
        tag = f"bs={bs} ch={ch} B={B} K={K} calls={calls} corrupt={corrupt} stream {s}"
        if rc == 0:
            assert np.array_equal(gbits[s], rb), f"{tag}: bits {gbits[s]} vs {rb}"
            if not np.array_equal(got[s].view(np.uint32), rp.view(np.uint32)):
                d = np.nonzero(got[s].view(np.uint32) != rp.view(np.uint32))
                t = int(d[0][0]); k = t // bs
                np.save(os.path.join(ROOT, "gpurun_out", "fuzz_fail_blocks.npy"), blocks[s])
                hexs = " ".join("%02x" % v for v in blocks[s, k, :48])
                raise AssertionError(f"{tag}: PCM differs first at sample {t} (block {k}, ch {int(d[1][0])}), {len(d[0])} values; bits {rb.tolist()}; block {k} bytes: {hexs}; prev block bytes: {' '.join('%02x' % v for v in blocks[s, max(k-1,0), :32])}")
        else:
            bad = rc - 1                          # first corrupt block: the stream stops there on both sides
            assert np.array_equal(gbits[s, :bad], rb[:bad]) and (gbits[s, bad:] == 0).all(), f"{tag}: corrupt at {bad}: {gbits[s]} vs {rb}"
            assert np.array_equal(got[s, :bad * bs].view(np.uint32), rp[:bad * bs].view(np.uint32)), f"{tag}: PCM before the corrupt block differs"
    n += 1; nblk += B * K * calls
print(f"fuzz_decode: {n} random batches ({ncorrupt} with corrupted bytes), {nblk} blocks, decoder == oracle in {time.time()-t0:.0f} s")
