#!/usr/bin/env python3
"""WindowCtrl histogram of the bench workload (how many blocks switch windows)."""
import os, sys, collections
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "ulc-codec_amd"))
import bench, ulc_amd
B, K = 4096, 16
dev = torch.device("cuda", 0)
pcm = bench.make_pcm(torch, B, K * bench.BS, dev, seed=1234)
enc = ulc_amd.BatchEncoder(B, bench.CH, bench.BS, bench.RATE, K)
slot = enc.slot
d_out = torch.zeros(B * K * slot, dtype=torch.uint8, device=dev); d_bits = torch.zeros(B * K, dtype=torch.int32, device=dev)
d_wc = torch.zeros(B * K, dtype=torch.int32, device=dev)
for it in range(3):
    enc.encode_dev(pcm.data_ptr(), K, d_out.data_ptr(), d_bits.data_ptr(), d_wc=d_wc.data_ptr(), mode=ulc_amd.MODE_VBR, p0=50.0)
    torch.cuda.synchronize()
    wc = d_wc.cpu().numpy()
    h = collections.Counter((wc >> 4).tolist())
    print("call", it, "plain 0x10: %.1f%%" % (100.0 * h[1] / wc.size), {hex(k): v for k, v in sorted(h.items())})
