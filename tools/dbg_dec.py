"""GPU debug: hand-assembled streams through the decoder vs the oracle; prints where they first differ."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ulc-codec_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import ulc_amd
from ulc_testlib import synth_block_stream, oracle_decode_stream_coefs
from spec_decoder import _nybbles

def run(bs, ch, B, K, calls):
    slot = 2 * ch * bs + 16
    streams = [synth_block_stream(1000 + 17 * s + bs, calls * K, ch, bs, slot) for s in range(B)]
    blocks = np.stack([st[0] for st in streams])
    dec = ulc_amd.BatchDecoder(B, ch, bs, K)
    got = np.concatenate([dec.decode(blocks[:, c * K:(c + 1) * K])[0] for c in range(calls)], axis=1)
    nbad = 0
    for s in range(B):
        rc, ref, bits, coefs = oracle_decode_stream_coefs(blocks[s], ch, bs)
        if np.array_equal(got[s].view(np.uint32), ref.view(np.uint32)):
            continue
        nbad += 1
        d = np.flatnonzero((got[s].view(np.uint32) != ref.view(np.uint32)).any(axis=1))
        k = d[0] // bs
        print(f"bs {bs} ch {ch} stream {s}: first diff at sample {d[0]} (block {k}, offset {d[0] % bs}), {len(d)} samples differ, last {d[-1]}")
        for kk in range(max(0, k - 1), min(calls * K, k + 1)):
            ny = _nybbles(blocks[s, kk])[: bits[kk] // 4 + 2]
            print(f"  block {kk}: bits {bits[kk]} header {ny[0]:x} {ny[1]:x}  first nybbles {''.join('%x' % v for v in ny[:48])}")
            for c in range(ch):
                co = coefs[kk].reshape(ch, bs)[c]
                nz = np.flatnonzero(co)
                print(f"    ch {c}: nonzero {len(nz)} first {nz[:4]} last {nz[-4:]}  negzero {int(np.count_nonzero((co == 0) & np.signbit(co)))}")
        if nbad >= 3:
            break
    print(f"bs {bs} ch {ch}: {nbad} bad streams")

for cfg in [(512, 2, 40, 6, 2), (2048, 2, 70, 5, 2), (1024, 1, 33, 7, 1), (256, 3, 20, 4, 2), (4096, 2, 9, 3, 1)]:
    run(*cfg)
