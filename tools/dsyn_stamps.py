"""GPU diagnostic (library built with -DULCX_DSYN_STAMPS): where k_dsyn's stereo fast path spends its cycles."""
import os, sys, ctypes as C
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ulc-codec_amd")); sys.path.insert(0, ROOT)
import ulc_amd, bench
dev = torch.device("cuda", 0)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
K = int(sys.argv[2]) if len(sys.argv) > 2 else 32
NS = 20
pcm = bench.make_pcm(torch, B, K * 2048, dev, seed=1)
enc = ulc_amd.BatchEncoder(B, 2, 2048, 44100, K); dec = ulc_amd.BatchDecoder(B, 2, 2048, K)
slot = enc.slot
d_out = torch.zeros(B * K * slot, dtype=torch.uint8, device=dev); d_bits = torch.zeros(B * K, dtype=torch.int32, device=dev)
d_wc = torch.zeros(B * K, dtype=torch.int32, device=dev)
d_dec = torch.zeros(B * K * 2048 * 2, dtype=torch.float32, device=dev); d_db = torch.zeros(B * K, dtype=torch.int32, device=dev)
enc.encode_dev(pcm.data_ptr(), K, d_out.data_ptr(), d_bits.data_ptr(), d_wc=d_wc.data_ptr(), p0=50.0); torch.cuda.synchronize()
for it in range(3):
    dec.decode_dev(d_out.data_ptr(), slot, K, d_dec.data_ptr(), d_db.data_ptr()); torch.cuda.synchronize()
print("stage ms", dec.stage_ms())
wc = d_wc.cpu().numpy()
print("decimated blocks: %.1f %%" % (100.0 * np.mean((wc & 8) != 0)))
out = np.zeros((B, 2 * NS), np.uint64)
l = ulc_amd.lib(); l.ulcx_decoder_debug_scratch.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t, C.c_int]
rc = l.ulcx_decoder_debug_scratch(dec.h, out.ctypes.data, 4 * 2048 * 4, 2 * NS * 8, B)
names = ["hdr", "unit-seed", "synth-tail", "fft", "barrierA", "post", "barrierB", "dec-time", "scatter", "noise-setup", "noise-draws", "pretw", "zero-fill", "wait-records"] + ["s%d" % i for i in range(14, NS)]
tot = out.astype(np.float64).mean(axis=0)
for w in range(2):
    print("wave", w, "  ".join("%s %.0f" % (names[i], tot[w * NS + i] / K) for i in range(NS) if tot[w * NS + i] > 0), " sum/blk %.0f cycles" % (tot[w * NS: w * NS + NS].sum() / K))
