"""GPU: encode-only timing of the bench batch (4096 streams x K blocks, default 32) for the library currently selected
(ULC_AMD_LIB); no verification, so it also runs the -DULCX_ABLATE builds (ULCX_DBG_SKIP).  Prints the stage intervals."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ulc-codec_amd")); sys.path.insert(0, ROOT)
import ulc_amd, bench
dev = torch.device("cuda", 0)
tag = sys.argv[1] if len(sys.argv) > 1 else ""
K = int(sys.argv[2]) if len(sys.argv) > 2 else 32
B = 4096
pcm = bench.make_pcm(torch, B, K * 2048, dev, seed=1234, bursts_per_s=float(os.environ.get('ENC_BENCH_BURSTS', '4.0')))      # (ENC_BENCH_BURSTS=0: one burst per stream - nearly every block in the steady state)
enc = ulc_amd.BatchEncoder(B, 2, 2048, 44100, K)
slot = enc.slot
d_out = torch.zeros(B * K * slot, dtype=torch.uint8, device=dev); d_bits = torch.zeros(B * K, dtype=torch.int32, device=dev)
acc = {}; tot = 0.0
e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
for it in range(8):
    e0.record()
    enc.encode_dev(pcm.data_ptr(), K, d_out.data_ptr(), d_bits.data_ptr(), p0=50.0)
    e1.record(); torch.cuda.synchronize()
    if it >= 3:
        tot += e0.elapsed_time(e1) / 5
        for k, v in enc.stage_ms().items(): acc[k] = acc.get(k, 0) + v / 5
import hashlib
md5 = hashlib.md5(d_out.cpu().numpy().tobytes() + d_bits.cpu().numpy().tobytes()).hexdigest()[:8]
print(tag, "enc %.3f |" % tot, " ".join("%s %.2f" % (k[2:], v) for k, v in acc.items() if v > 0.05), "| mean bytes %.1f" % (d_bits.float().mean().item() / 8), md5)
