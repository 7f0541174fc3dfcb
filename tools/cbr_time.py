import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ulc-codec_amd")); sys.path.insert(0, ROOT)
import torch, ulc_amd, bench
dev = torch.device("cuda", 0)
B, K, bs, ch, rate = 4096, 8, 2048, 2, 48000
bench.RATE = rate
pcm = bench.make_pcm(torch, B, K * bs, dev, 7)
enc = ulc_amd.BatchEncoder(B, ch, bs, rate, K)
out = torch.zeros(B * K * enc.slot, dtype=torch.uint8, device=dev); bits = torch.zeros(B * K, dtype=torch.int32, device=dev)
for mode, p0, name in ((ulc_amd.MODE_VBR, 50.0, "VBR50"), (ulc_amd.MODE_CBR, 64.0, "CBR64")):
    for it in range(3):
        torch.cuda.synchronize(); t = time.perf_counter()
        enc.encode_dev(pcm.data_ptr(), K, out.data_ptr(), bits.data_ptr(), mode=mode, p0=p0)
        torch.cuda.synchronize(); dt = time.perf_counter() - t
    print(name, "ms per %d blocks: %.2f" % (B * K, dt * 1e3), {k: round(v, 2) for k, v in enc.stage_ms().items() if v > 0.3})
