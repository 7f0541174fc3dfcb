import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ulc-codec_amd")); sys.path.insert(0, ROOT)
import torch, ulc_amd, bench
dev = torch.device("cuda", 0)
B, K, bs, ch = 4096, 16, 2048, 2
pcm = bench.make_pcm(torch, B, K * bs, dev, 1)
enc = ulc_amd.BatchEncoder(B, ch, bs, bench.RATE, K)
out = torch.zeros(B * K * enc.slot, dtype=torch.uint8, device=dev); bits = torch.zeros(B * K, dtype=torch.int32, device=dev)
for it in range(3):
    torch.cuda.synchronize(); t = time.perf_counter()
    enc.encode_dev(pcm.data_ptr(), K, out.data_ptr(), bits.data_ptr(), mode=ulc_amd.MODE_VBR, p0=50.0)
    torch.cuda.synchronize(); dt = time.perf_counter() - t
    print("ms %.2f fallbacks %d of %d" % (dt * 1e3, enc.last_fallbacks(), B * K), {k: round(v, 2) for k, v in enc.stage_ms().items() if v > 0.2})
