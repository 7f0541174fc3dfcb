"""GPU: decode-only timing of a batch (default the 4096 streams x 16 blocks of rounds 1-2; argv: tag B K) for the library
currently selected (ULC_AMD_LIB)."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ulc-codec_amd")); sys.path.insert(0, ROOT)
import ulc_amd, bench
dev = torch.device("cuda", 0)
B = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
K = int(sys.argv[3]) if len(sys.argv) > 3 else 16
pcm = bench.make_pcm(torch, B, K * 2048, dev, seed=1234, bursts_per_s=float(os.environ.get('DEC_BENCH_BURSTS', '4.0')))      # (DEC_BENCH_BURSTS=0: one burst per stream - nearly no decimated blocks)
enc = ulc_amd.BatchEncoder(B, 2, 2048, 44100, K); dec = ulc_amd.BatchDecoder(B, 2, 2048, K)
slot = enc.slot
d_out = torch.zeros(B * K * slot, dtype=torch.uint8, device=dev); d_bits = torch.zeros(B * K, dtype=torch.int32, device=dev)
d_dec = torch.zeros(B * K * 2048 * 2, dtype=torch.float32, device=dev); d_db = torch.zeros(B * K, dtype=torch.int32, device=dev)
enc.encode_dev(pcm.data_ptr(), K, d_out.data_ptr(), d_bits.data_ptr(), p0=50.0); torch.cuda.synchronize()
acc = {}
for it in range(8):
    dec.decode_dev(d_out.data_ptr(), slot, K, d_dec.data_ptr(), d_db.data_ptr()); torch.cuda.synchronize()
    if it >= 3:
        for k, v in dec.stage_ms().items(): acc[k] = acc.get(k, 0) + v / 5
import hashlib
print(sys.argv[1] if len(sys.argv) > 1 else "", {k: round(v, 3) for k, v in acc.items()}, "sum %.3f" % sum(acc.values()), hashlib.md5(d_dec.cpu().numpy().tobytes()).hexdigest()[:8])
